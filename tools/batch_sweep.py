#!/usr/bin/env python3
"""ms per clip of the full pipeline vs batch size (does the 256 MB Infinity Cache help small batches?)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cmgan_amd import TSCNet
from oracle.weights import make_state_dict, synthetic_clips

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
