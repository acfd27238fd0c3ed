"""Data path of the reference: `DemandDataset` / `load_data` of src/data/dataloader.py with the
same constructor arguments, item tuple `(clean, noisy, length)` and length rule, feeding the HIP front end.

Host-side Python like the original; the differences are the decoders this image has: .wav files are read with
scipy (PCM normalised to [-1, 1) float32 exactly as torchaudio.load does) instead of torchaudio/sox, and file
names are ordered by an in-module natural sort instead of `natsort`.  Every clip leaves the dataset at exactly
`cut_len` samples (a multiple of hop = 100 for the defaults), which is what `cmgan_enhance` takes.
"""
from __future__ import annotations

import os
import random
import re
from typing import Tuple

import numpy as np
import torch
import torch.utils.data
from torch.utils.data.distributed import DistributedSampler

__all__ = ["DemandDataset", "load_data", "fit_length", "read_wav", "DevicePrefetcher"]


def _natural_key(name: str):
    return [int(t) if t.isdigit() else t.lower() for t in re.split(r"(\d+)", name)]


def read_wav(path: str) -> Tuple[torch.Tensor, int]:
    """(float32 mono samples in [-1, 1), sample rate); channel 0 of multi-channel files."""
    from scipy.io import wavfile
    sr, data = wavfile.read(path)
    if data.ndim > 1:
        data = data[:, 0]
    if data.dtype == np.int16:
        x = data.astype(np.float32) / 32768.0
    elif data.dtype == np.int32:
        x = (data.astype(np.float64) / 2147483648.0).astype(np.float32)
    elif data.dtype == np.uint8:
        x = (data.astype(np.float32) - 128.0) / 128.0
    else:
        x = data.astype(np.float32)
    return torch.from_numpy(np.ascontiguousarray(x)), int(sr)


def fit_length(x: torch.Tensor, cut_len: int, start: int = 0) -> torch.Tensor:
    """The reference's length rule (dataloader.py:33-48): a clip shorter than cut_len is repeated
    cut_len // len times plus its own head; a longer one is cropped to [start, start + cut_len)."""
    n = x.size(-1)
    if n < cut_len:
        reps, rest = divmod(cut_len, n)
        return torch.cat([x] * reps + [x[..., :rest]], dim=-1)
    return x[..., start:start + cut_len]


class DemandDataset(torch.utils.data.Dataset):
    """`<data_dir>/clean/*.wav` paired by name with `<data_dir>/noisy/*.wav`   (dataloader.py:13-50).
    Long clips are cropped at `random.randint(0, length - cut_len)` - the same draw from the same module-level
    RNG as the reference, so `random.seed(s)` reproduces its crops."""

    def __init__(self, data_dir: str, cut_len: int = 16000 * 2):
        self.cut_len = cut_len
        self.clean_dir = os.path.join(data_dir, "clean")
        self.noisy_dir = os.path.join(data_dir, "noisy")
        self.clean_wav_name = sorted(os.listdir(self.clean_dir), key=_natural_key)

    def __len__(self) -> int:
        return len(self.clean_wav_name)

    def __getitem__(self, idx: int):
        name = self.clean_wav_name[idx]
        clean, _ = read_wav(os.path.join(self.clean_dir, name))
        noisy, _ = read_wav(os.path.join(self.noisy_dir, name))
        length = clean.numel()
        if length != noisy.numel():
            raise ValueError(f"{name}: clean and noisy files differ in length")
        start = random.randint(0, length - self.cut_len) if length >= self.cut_len else 0
        return fit_length(clean, self.cut_len, start), fit_length(noisy, self.cut_len, start), length


def load_data(ds_dir: str, batch_size: int, n_cpu: int, cut_len: int):
    """(train_loader, test_loader) over `<ds_dir>/train` and `<ds_dir>/test`   (dataloader.py:53-81): batches of
    `(clean [B, cut_len], noisy [B, cut_len], length [B])`, pinned, one DistributedSampler shard per rank.
    The reference always samples through `DistributedSampler` (default shuffle=True) and needs an initialised
    process group; a single process gets the same behaviour from a one-replica sampler, so BOTH splits are
    shuffled exactly as there (yes, the reference shuffles the test split too).  The sampler is reachable as
    `loader.sampler`; call `loader.sampler.set_epoch(e)` per epoch for a new permutation (the reference never
    does, so it replays one fixed permutation - kept as its default here)."""
    def loader(split: str, drop_last: bool):
        ds = DemandDataset(os.path.join(ds_dir, split), cut_len)
        dist_on = torch.distributed.is_available() and torch.distributed.is_initialized()
        sampler = DistributedSampler(ds) if dist_on else DistributedSampler(ds, num_replicas=1, rank=0)
        return torch.utils.data.DataLoader(dataset=ds, batch_size=batch_size, pin_memory=torch.cuda.is_available(),
                                           shuffle=False, sampler=sampler, drop_last=drop_last, num_workers=n_cpu)

    return loader("train", True), loader("test", False)


class DevicePrefetcher:
    """Feeds a loader's `(clean, noisy, length)` batches to the HIP front end as DEVICE tensors: batch k+1 is copied
    host -> HBM on a side stream (pinned source, `non_blocking`) into the other half of a two-slot ring while the
    kernels of batch k run on the compute stream, so the PCIe leg (4.1 MB per direction at 32 x 2 s) never sits on
    the critical path.  The reference does `batch[0].to(self.gpu_id)` synchronously inside the step
    (src/train.py:179-180, 210-211).

    Yields `(clean, noisy, length)` with clean / noisy float32 [B, cut_len] on `device`; the tensors of a slot are
    reused two batches later, so consume (or clone) a batch before asking for the one after next."""

    def __init__(self, loader, device):
        self.loader = loader
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("DevicePrefetcher feeds GPU kernels: device must be a GPU")
        self.stream = torch.cuda.Stream(device=self.device)
        self._slots = [None, None]

    def __len__(self):
        return len(self.loader)

    def _stage(self, batch, k):
        clean, noisy, length = batch
        slot = self._slots[k]
        if slot is None or slot[0].shape != clean.shape:
            slot = (torch.empty(clean.shape, dtype=torch.float32, device=self.device),
                    torch.empty(noisy.shape, dtype=torch.float32, device=self.device))
            self._slots[k] = slot
        compute = torch.cuda.current_stream(self.device)
        self.stream.wait_stream(compute)            # the slot's previous consumer kernels are ordered before the copy
        with torch.cuda.stream(self.stream):
            slot[0].copy_(clean, non_blocking=True)
            slot[1].copy_(noisy, non_blocking=True)
            ready = torch.cuda.Event()
            ready.record(self.stream)
        return slot[0], slot[1], length, ready

    def __iter__(self):
        it = iter(self.loader)
        k = 0
        try:
            nxt = self._stage(next(it), k)
        except StopIteration:
            return
        while nxt is not None:
            clean, noisy, length, ready = nxt
            k ^= 1
            try:
                nxt = self._stage(next(it), k)      # overlaps with the caller's kernels on the current batch
            except StopIteration:
                nxt = None
            torch.cuda.current_stream(self.device).wait_event(ready)
            yield clean, noisy, length
