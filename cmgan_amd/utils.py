"""Mirrors of ``utils.power_compress`` / ``power_uncompress`` (reference: src/utils.py:20-39)
and the two fused front/back-end ops that replace the reference's
``torch.stft -> power_compress -> permute`` and ``permute -> power_uncompress -> torch.istft``
call sequences (src/evaluation.py:36-39, 41-50).  All run as HIP kernels."""
from __future__ import annotations

from typing import Optional

import torch

from .engine import Engine

_engines: dict = {}


def _engine(n_fft: int = 400, hop: int = 100) -> Engine:
    key = (n_fft, hop, torch.cuda.current_device())
    if key not in _engines:
        _engines[key] = Engine(n_fft=n_fft, hop=hop)
    return _engines[key]


def power_compress(x: torch.Tensor) -> torch.Tensor:
    """x[B,F,T,2] -> [B,2,F,T]   (src/utils.py:20-29)"""
    return _engine().power_compress(x)


def power_uncompress(real: torch.Tensor, imag: torch.Tensor) -> torch.Tensor:
    """real, imag [B,1,F,T] -> [B,1,F,T,2]   (src/utils.py:32-39)"""
    return _engine().power_uncompress(real, imag)


def rms_scale(wav: torch.Tensor) -> torch.Tensor:
    """c = sqrt(L / sum x^2) per row   (src/evaluation.py:21)"""
    return _engine().rms_scale(wav)


def stft_compress(wav: torch.Tensor, n_fft: int = 400, hop: int = 100,
                  scale: Optional[torch.Tensor] = None) -> torch.Tensor:
    """wav[B,L] (times scale[B]) -> model input [B,2,T,F]   (src/evaluation.py:36-39)"""
    return _engine(n_fft, hop).stft_compress(wav, scale)


def uncompress_istft(est_real: torch.Tensor, est_imag: torch.Tensor, n_fft: int = 400, hop: int = 100,
                     scale: Optional[torch.Tensor] = None) -> torch.Tensor:
    """model outputs 2 x [B,1,T,F] -> wav[B, hop*(T-1)] (divided by scale[B])   (src/evaluation.py:41-51)"""
    return _engine(n_fft, hop).uncompress_istft(est_real, est_imag, scale)
