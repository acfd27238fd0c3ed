#!/usr/bin/env python3
"""Golden vectors of BASELINE.json configs[4] at its named size (build container only; ~2.5 min on 8 cores):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_stream_golden.py

  whole        the REFERENCE's own whole-clip output on the 10 s synthetic clip of bench.py's stream leg: its TSCNet /
               power_compress / power_uncompress (oracle/_ref, byte-compiled from /root/reference by oracle/make_ref.py)
               behind the src/evaluation.py:21-53 glue (oracle/ref_runner.enhance_batch)
  stream_40_40 the carried-state streaming contract (oracle/stream_oracle.py: the reference arithmetic with frozen
  stream_40_0  InstanceNorm statistics and windowed TSCBs) on the same clip: 400-frame windows, 40 frames of context,
               40 / 0 frames of look-ahead
stored as float32 in stream10s.npz (the clip itself is regenerated from its seed).  Used by tests/test_gpu_stream_config5.py
so that the GPU suite does not spend two minutes of host time on them."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True

from oracle import ref_runner as R                     # noqa: E402
from oracle import stream_oracle as S                  # noqa: E402
from oracle.weights import make_state_dict, synthetic_clips  # noqa: E402

torch.set_grad_enabled(False)
assert R.available(), "build oracle/_ref first (python -m oracle.make_ref)"
sd = make_state_dict(seed=0, num_features=201)
clip = synthetic_clips(1, 160000, seed=3)
out = {"seed": np.int64(3), "samples": np.int64(160000),
       "whole": R.enhance_batch(R.tscnet(sd), clip)[0].numpy().astype(np.float32)}
for ca, la in ((40, 40), (40, 0)):
    out[f"stream_{ca}_{la}"] = S.enhance_stream(sd, clip, window=400, context=ca, lookahead=la).numpy().astype(np.float32)
    print(ca, la, float(np.abs(out[f"stream_{ca}_{la}"] - out["whole"]).max() / np.abs(out["whole"]).max()), flush=True)
np.savez(os.path.join(HERE, "stream10s.npz"), **out)
print("wrote stream10s.npz")
