"""Mirror of ``enhance_one_track`` (reference: src/evaluation.py:12-58) minus file I/O:
same argument meaning, same padding / >cut_len chunking rule, same output, with every
device op a HIP kernel.  The host code below is shape logic only."""
from __future__ import annotations

import math

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
