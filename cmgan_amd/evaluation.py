"""Mirror of the reference's evaluation script (src/evaluation.py): ``enhance_one_track`` (:12-58, same
argument meaning, same padding / >cut_len chunking rule, same output, every device op a HIP kernel; the
host code is shape logic only) and the directory driver ``evaluation`` (:61-100) that scores the enhanced
tracks with cmgan_amd.metrics."""
from __future__ import annotations

import math
import os
import re

import numpy as np
import torch

from .generator import TSCNet


def chunk_rows(padded_len: int, cut_len: int) -> int:
    """Rows the reference reshapes long audio into (src/evaluation.py:30-34)."""
    if padded_len <= cut_len:
        return 1
    rows = int(math.ceil(padded_len / cut_len))
    while 100 % rows != 0:
        rows += 1
    return rows


@torch.no_grad()
def enhance_one_track(model: TSCNet, noisy: torch.Tensor, cut_len: int = 16000 * 16, n_fft: int = 400,
                      hop: int = 100) -> torch.Tensor:
    """noisy: float32 [1, L] in [-1, 1] on the GPU -> enhanced [L] on the GPU."""
    if noisy.dim() != 2 or noisy.size(0) != 1:
        raise ValueError("expected a mono track shaped [1, L] (torchaudio.load layout)")
    eng = model.engine
    if (eng.cfg.n_fft, eng.cfg.hop) != (n_fft, hop):
        raise ValueError("model was built for a different n_fft / hop")
    noisy = noisy.to(dtype=torch.float32).contiguous()
    length = noisy.size(-1)
    c = eng.rms_scale(noisy)                                               # evaluation.py:21
    padded_len = int(math.ceil(length / 100)) * 100                        # evaluation.py:25-29
    padded = torch.cat([noisy, noisy[:, : padded_len - length]], dim=-1)   # wrap-pad with the clip's head
    rows = chunk_rows(padded_len, cut_len)                                 # evaluation.py:30-34
    batch = padded.reshape(rows, -1).contiguous()
    scale = c.expand(rows).contiguous()
    spec = eng.stft_compress(batch, scale)                                 # evaluation.py:36-39 (x*c fused)
    real, imag = model(spec)                                               # evaluation.py:40
    audio = eng.uncompress_istft(real, imag, scale)                        # evaluation.py:41-51 (/c fused)
    return audio.flatten()[:length]                                        # evaluation.py:52


@torch.no_grad()
def enhance_batch(model: TSCNet, wav: torch.Tensor) -> torch.Tensor:
    """Equal-length clips [B, L] (L % hop == 0), per-row RMS scale (src/train.py:75-79):
    one fused ABI call (cmgan_enhance)."""
    return model.engine.enhance(wav)


# ------------------------------------------------------------------------------------------------
# directory driver (src/evaluation.py:61-100)
# ------------------------------------------------------------------------------------------------
def _natural_key(name: str):
    """natsort-style ordering (p226_2 before p226_10), what the reference gets from natsorted()."""
    return [int(t) if t.isdigit() else t.lower() for t in re.split(r"(\d+)", name)]


def _read_wav(path: str):
    """(float64 samples in [-1, 1), sample rate): soundfile / torchaudio normalisation of PCM files."""
    from scipy.io import wavfile
    sr, data = wavfile.read(path)
    if data.ndim > 1:
        data = data[:, 0]
    if data.dtype == np.int16:
        data = data.astype(np.float64) / 32768.0
    elif data.dtype == np.int32:
        data = data.astype(np.float64) / 2147483648.0
    elif data.dtype == np.uint8:
        data = (data.astype(np.float64) - 128.0) / 128.0
    else:
        data = data.astype(np.float64)
    return data, sr


@torch.no_grad()
def evaluation(model, noisy_dir: str, clean_dir: str, save_tracks: bool = False, saved_dir: str | None = None,
               pesq_fn=None, verbose: bool = True, subtype: str = "PCM_16", allow_missing_pesq: bool = False):
    """Enhance every .wav of `noisy_dir`, score it against the file of the same name in `clean_dir`, and
    return the six averages in the reference's order (pesq, csig, cbak, covl, ssnr, stoi).

    `model` is a loaded cmgan_amd.TSCNet or the path of a reference checkpoint (a state_dict saved by
    src/train.py, loaded exactly like evaluation.py:63-65).  `pesq_fn(fs, clean, enhanced)` overrides the
    `pesq` wheel lookup of cmgan_amd.metrics.  The reference imports `pesq` unconditionally and fails without
    it (compute_metrics.py:7); so does this driver (ImportError) unless `allow_missing_pesq=True`, in which case
    the four PESQ-dependent averages are NaN and a warning says so.  Saved tracks are 16-bit PCM like the
    reference's `sf.write(path, est, sr)` default (`subtype="FLOAT"` keeps float32 samples instead)."""
    from . import metrics
    n_fft = 400
    if isinstance(model, (str, os.PathLike)):
        sd = torch.load(model, map_location="cpu")
        model = TSCNet(num_channel=64, num_features=n_fft // 2 + 1).load_state_dict(sd).eval()
    if subtype not in ("PCM_16", "FLOAT"):
        raise ValueError("subtype must be 'PCM_16' (soundfile's default for .wav) or 'FLOAT'")
    if pesq_fn is None and not metrics.have_pesq():
        if not allow_missing_pesq:
            raise ImportError("the `pesq` package (ITU-T P.862, src/requirements.txt:6) is not installed and no "
                              "pesq_fn was given: PESQ / CSIG / CBAK / COVL cannot be computed.  Pass "
                              "allow_missing_pesq=True to get SSNR and STOI with the others as NaN.")
        import warnings
        warnings.warn("pesq is unavailable: the pesq, csig, cbak and covl averages will be NaN", RuntimeWarning)
    if save_tracks:
        if saved_dir is None:
            raise ValueError("save_tracks needs saved_dir")
        os.makedirs(saved_dir, exist_ok=True)
    names = sorted((f for f in os.listdir(noisy_dir) if f.lower().endswith(".wav")), key=_natural_key)
    if not names:
        raise ValueError(f"no .wav files in {noisy_dir}")
    total = np.zeros(6)
    dev = torch.device(model.engine.device)
    for name in names:
        noisy, sr = _read_wav(os.path.join(noisy_dir, name))
        if sr != 16000:
            raise ValueError(f"{name}: expected 16 kHz audio, got {sr}")           # evaluation.py:19
        est = enhance_one_track(model, torch.from_numpy(noisy).to(dev, torch.float32)[None, :],
                                16000 * 16, n_fft, n_fft // 4).cpu().numpy().astype(np.float64)
        if save_tracks:
            from scipy.io import wavfile
            if subtype == "PCM_16":     # libsndfile's float -> int16 conversion: scale by 2^15, round, clip
                pcm = np.clip(np.rint(est * 32768.0), -32768, 32767).astype(np.int16)
                wavfile.write(os.path.join(saved_dir, name), sr, pcm)
            else:
                wavfile.write(os.path.join(saved_dir, name), sr, est.astype(np.float32))
        clean, sr_c = _read_wav(os.path.join(clean_dir, name))
        if sr_c != 16000:
            raise ValueError(f"{name}: clean file is not 16 kHz")
        mos = None if pesq_fn is None else float(pesq_fn(sr, clean, est))
        total += np.array(metrics.compute_metrics(clean, est, sr, 0, pesq_mos=mos))
    avg = metrics.Scores(*(total / len(names)))
    if verbose:
        print("pesq: ", avg.pesq, "csig: ", avg.csig, "cbak: ", avg.cbak, "covl: ", avg.covl,
              "ssnr: ", avg.ssnr, "stoi: ", avg.stoi)
    return avg
