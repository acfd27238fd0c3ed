#!/usr/bin/env python3
"""Max-norm relative error of the loaded library build (CMGAN_HIP_LIB selects a variant) against the CPU oracle on
2 x 2 s synthetic clips and on the three real-recording goldens - used once to quote the error of the
single-product fp16 experiment (cmgan_amd.build variant "x1") next to the shipped f16x3 mode."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from cmgan_amd import TSCNet
from cmgan_amd.evaluation import enhance_batch, enhance_one_track
from cmgan_amd.synth import make_state_dict, synthetic_clips
from oracle import cmgan_oracle as O

sd = make_state_dict(0)
model = TSCNet(64, 201).load_state_dict(sd).eval()
wav = synthetic_clips(2, 32000, seed=6)
got = enhance_batch(model, wav.cuda()).cpu()
want = O.enhance_batch(sd, wav)
res = {"lib": os.environ.get("CMGAN_HIP_LIB", "default"),
       "synthetic_2x2s_rel_err": float((got - want).abs().max() / want.abs().max())}
g = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "tracks.npz"))
for tag in ("a", "b", "silence"):
    noisy = torch.from_numpy(g[f"pcm_{tag}"].astype(np.float32) / 32768.0)[None, :]
    out = enhance_one_track(model, noisy.cuda()).cpu()
    ref = torch.from_numpy(g[f"enhanced_{tag}"])
    res[f"track_{tag}_rel_err"] = float((out - ref).abs().max() / ref.abs().max())
print(json.dumps(res))
