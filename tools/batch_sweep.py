#!/usr/bin/env python3
"""ms per clip of the full pipeline vs batch size (does the 256 MB Infinity Cache help small batches?)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cmgan_amd import TSCNet
from cmgan_amd.synth import make_state_dict, synthetic_clips

mode = sys.argv[1] if len(sys.argv) > 1 else "f16x3"
model = TSCNet(64, 201, mfma_mode=mode).load_state_dict(make_state_dict(0))
for B in (1, 2, 4, 8, 16, 32, 64):
    wav = synthetic_clips(B, 32000, seed=1).cuda()
    for _ in range(2):
        model.engine.enhance(wav)
    torch.cuda.synchronize()
    n = max(2, 64 // B)
    t0 = time.perf_counter()
    for _ in range(n):
        model.engine.enhance(wav)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    print(f"B={B:3d}  {1e3 * dt:8.2f} ms/step  {1e3 * dt / B:7.3f} ms/clip  {B * 321 / dt:10.0f} frames/s", flush=True)

# front end alone at large batch (north_star quotes an HBM fraction for the STFT at 32 x 32000; the transform is
# launch / latency bound at that size, so the large-batch rate is reported next to it)
eng = model.engine
for B in (32, 256, 1024):
    wav = synthetic_clips(B, 32000, seed=2).cuda()
    c = eng.rms_scale(wav)
    for _ in range(3):
        eng.stft_compress(wav, c)
    torch.cuda.synchronize()
    n = 20
    t0 = time.perf_counter()
    for _ in range(n):
        eng.stft_compress(wav, c)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    nbytes = B * (4 * 32000 + 8 * 201 * 321)
    print(f"stft_compress B={B:4d}  {1e6 * dt:8.1f} us  {nbytes / dt / 1e9:8.1f} GB/s algorithmic = {nbytes / dt / 8e12:.3f} of 8 TB/s", flush=True)
    spec = eng.stft_compress(wav, c)
    re, im = spec[:, 0:1].contiguous(), spec[:, 1:2].contiguous()
    for _ in range(3):
        eng.uncompress_istft(re, im, c)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        eng.uncompress_istft(re, im, c)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    print(f"uncompress_istft B={B:4d}  {1e6 * dt:8.1f} us  {nbytes / dt / 1e9:8.1f} GB/s algorithmic = {nbytes / dt / 8e12:.3f} of 8 TB/s", flush=True)
