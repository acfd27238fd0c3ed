"""Data-parallel launcher glue: one process per GPU, utterance batches sharded by rank,
weights replicated, NO data-path collective (the forward has no cross-sample coupling in
eval mode: InstanceNorm is per sample, BatchNorm1d uses running stats - SURVEY.md 8e).
The only collective is the reduction of per-step scalars (north_star: "a single RCCL
all-reduce over xGMI on the loss scalars"): backend "nccl" on ROCm is RCCL; tests use gloo.
"""
from __future__ import annotations

import os
from typing import Tuple

import torch
import torch.distributed as dist


def init_from_env(backend: str | None = None) -> Tuple[int, int, int]:
    """(rank, local_rank, world).  Single-process when WORLD_SIZE is unset/1."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # CMGAN_DIST_FORCE_INIT=1: build the process group even for ONE rank, so that a single-GPU box can run the real
    # "nccl" (= RCCL) code path end to end - communicator set-up, the loss-scalar / gradient-bucket all-reduces, the
    # buffer broadcasts, the barrier - as a self-test (tests/test_gpu_dist_rccl.py); a 1-rank collective is the identity
    if (world > 1 or _forced()) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local, world


def _forced() -> bool:
    return os.environ.get("CMGAN_DIST_FORCE_INIT", "0") == "1"


def _active() -> bool:
    """Collectives run: a process group exists and spans several ranks (or the single-rank self-test is on)."""
    return dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or _forced())


def shard_bounds(n_items: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous, balanced [lo, hi) slice of n_items for `rank` (first n%world ranks get one more)."""
    base, extra = divmod(n_items, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def shard_batch(batch: torch.Tensor, rank: int, world: int) -> torch.Tensor:
    lo, hi = shard_bounds(batch.size(0), rank, world)
    return batch[lo:hi]


def allreduce_scalars(values: torch.Tensor, op=dist.ReduceOp.SUM) -> torch.Tensor:
    """One small all-reduce (RCCL over xGMI on the GPU box); identity in a single process."""
    if _active():
        dist.all_reduce(values, op=op)
    return values


def gather_shards(local: torch.Tensor, n_items: int) -> torch.Tensor:
    """Reassemble rank-ordered shards of a [n_items, ...] batch on every rank (test helper)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return local
    world = dist.get_world_size()
    sizes = [shard_bounds(n_items, r, world) for r in range(world)]
    mx = max(hi - lo for lo, hi in sizes)
    pad = torch.zeros((mx,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[: local.size(0)] = local
    bufs = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(bufs, pad)
    return torch.cat([b[: hi - lo] for b, (lo, hi) in zip(bufs, sizes)], dim=0)


class FlatBucket:
    """Named tensors as views of ONE flat fp32 buffer - the shape the reference's DDP gives its gradient buckets
    (7.3 MB generator + 0.7 MB discriminator, src/train.py:68-69) and the shape that suits a fully connected xGMI
    mesh: a single all-reduce over the whole bucket instead of one per tensor (latency-bound at this size), and a
    single optimiser launch over the same memory (cmgan_adamw_step).  Works on CPU tensors too (gloo tests).
    Every tensor starts on a 64-byte boundary (the kernels read parameters with 16-byte vector loads); the padding
    floats are zero in the parameter, gradient and moment buckets alike and stay zero under all-reduce and AdamW."""

    ALIGN = 16                              # floats

    def __init__(self, shapes: dict, device="cpu"):
        self.shapes = {k: tuple(v) for k, v in shapes.items()}
        pad = lambda n: (n + self.ALIGN - 1) // self.ALIGN * self.ALIGN
        sizes = {k: int(torch.Size(s).numel()) for k, s in self.shapes.items()}
        self.numel = sum(pad(n) for n in sizes.values())
        self.flat = torch.zeros(self.numel, dtype=torch.float32, device=device)
        self.views, off = {}, 0
        for k, shp in self.shapes.items():
            self.views[k] = self.flat[off:off + sizes[k]].view(shp)
            off += pad(sizes[k])

    def load(self, tensors: dict):
        for k, v in self.views.items():
            v.copy_(tensors[k])
        return self

    def __getitem__(self, key):
        return self.views[key]


def allreduce_mean(flat: torch.Tensor) -> torch.Tensor:
    """Gradient averaging over ranks as ONE collective on a flat bucket (what DDP's bucketed all-reduce computes,
    src/train.py:192,200); identity in a single process."""
    if _active():
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)
        flat /= dist.get_world_size()
    return flat


def _multi() -> bool:
    return _active()


def broadcast_from_rank0(tensors) -> None:
    """In-place broadcast of rank 0's values: what DistributedDataParallel does with the parameters AND buffers at
    construction (src/train.py:68-69) and, with `broadcast_buffers=True` (its default), with the buffers before every
    forward.  `tensors`: a tensor or an iterable of tensors; identity in a single process."""
    if not _multi():
        return
    if isinstance(tensors, torch.Tensor):
        tensors = [tensors]
    # one collective per (device, dtype) group, not one per tensor: the per-step buffer broadcast is ~38 tensors of a few
    # hundred bytes (BatchNorm statistics, spectral-norm vectors) - latency, not bytes
    groups = {}
    for t in tensors:
        groups.setdefault((t.device, t.dtype), []).append(t)
    for group in groups.values():
        if len(group) == 1:
            dist.broadcast(group[0], src=0)
            continue
        flat = torch.cat([t.detach().reshape(-1) for t in group])
        dist.broadcast(flat, src=0)
        off = 0
        for t in group:
            n = t.numel()
            t.detach().copy_(flat[off:off + n].view_as(t))
            off += n


def all_agree(flag: bool, device="cpu") -> bool:
    """True only if `flag` is true on EVERY rank (one MIN all-reduce of a single element).  Used to decide collectively
    whether a step that contains collectives runs: a rank-local decision (e.g. 'PESQ returned no labels here', the
    reference's src/train.py:194) would let one rank skip an all-reduce the others enter."""
    if not _multi():
        return bool(flag)
    t = torch.tensor([1 if flag else 0], dtype=torch.int32, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    return bool(t.item())


def get_rank() -> int:
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0
