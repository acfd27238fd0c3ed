#!/usr/bin/env python3
"""Front / back end alone (stft_compress, uncompress_istft) at B = 32 / 256 for a rocprofv3 --kernel-trace --stats run."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from cmgan_amd.engine import Engine
from cmgan_amd.synth import synthetic_clips
eng = Engine()
for B in (32, 256):
    wav = synthetic_clips(B, 32000, seed=2).cuda()
    c = eng.rms_scale(wav)
    for _ in range(10):
        spec = eng.stft_compress(wav, c)
    re, im = spec[:, 0:1].contiguous(), spec[:, 1:2].contiguous()
    for _ in range(10):
        eng.uncompress_istft(re, im, c)
    torch.cuda.synchronize()
