#!/usr/bin/env python3
"""Which contractions need three split-f16 products?  (through gpurun)

For every kernel family of the F16MIX mode (include/cmgan_hip.h, CMGAN_MIX_*) - alone, then cumulatively in order of
increasing cost - run the full pipeline wav -> wav with THAT family on one fp16 product and everything else on three,
and record the end-to-end error against the F16X3 output of the same library (which is within 3e-6 of the reference:
tests/test_gpu_parity.py) on the benchmark batch (32 x 2 s synthetic clips) and on the three AudioSamples recordings of
tests/golden/tracks.npz, next to the time of a step.  Prints a table + one JSON line (profiles/r06_mix_ablation.json)."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from cmgan_amd import TSCNet, _lib
from cmgan_amd.synth import make_state_dict, synthetic_clips

dev = torch.device("cuda:0")
sd = make_state_dict(seed=0, num_features=201)
wav = synthetic_clips(32, 32000, seed=0).to(dev)
g = np.load(os.path.join(ROOT, "tests", "golden", "tracks.npz"))
tracks = [torch.from_numpy(g["pcm_" + n].astype(np.float32) / 32768.0)[None][:, :(g["pcm_" + n].size // 100) * 100].contiguous().to(dev)
          for n in ("a", "b", "silence")]


def run(mode, single=None, time_it=True):
    m = TSCNet(64, 201, mfma_mode=mode, mix_single=single).load_state_dict(sd).eval()
    eng = m.engine
    out = eng.enhance_graphed(wav).clone()
    tr = [eng.enhance(t).clone() for t in tracks]
    ms = None
    if time_it:
        for _ in range(3):
            eng.enhance_graphed(wav)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            eng.enhance_graphed(wav)
        torch.cuda.synchronize()
        ms = 1e3 * (time.perf_counter() - t0) / 10
    eng._graphs.clear()
    return out, tr, ms


def rel(a, b):
    return float((a - b).abs().max() / b.abs().max())


ref, ref_tr, ms3 = run("f16x3")
fams = list(_lib.MIX)
rows = []


def record(name, single):
    out, tr, ms = run("f16mix", single)
    e = rel(out, ref)
    et = max(rel(a, b) for a, b in zip(tr, ref_tr))
    rows.append({"single": name, "families": list(single), "rel_err_batch": float(f"{e:.3e}"), "rel_err_tracks_max": float(f"{et:.3e}"),
                 "ms_per_step": round(ms, 3), "ms_saved": round(ms3 - ms, 3)})
    print(f"{name:>40}  batch {e:.2e}  tracks {et:.2e}  {ms:.2f} ms ({ms3 - ms:+.2f})", flush=True)


print(f"{'f16x3 (reference of this table)':>40}  {ms3:.2f} ms", flush=True)
for f in fams:
    record(f, (f,))
order = [r["single"] for r in sorted(rows, key=lambda r: max(r["rel_err_batch"], r["rel_err_tracks_max"]))]
cum = []
for f in order:
    cum.append(f)
    if len(cum) > 1:
        record("+".join(cum), tuple(cum))
out1, tr1, ms1 = run("f16x1")
print(f"{'f16x1 (every family single)':>40}  batch {rel(out1, ref):.2e}  tracks {max(rel(a, b) for a, b in zip(tr1, ref_tr)):.2e}  {ms1:.2f} ms", flush=True)
print(json.dumps({"workload": "32 x 2 s synthetic clips (bench.py's batch) + tests/golden/tracks.npz, wav -> wav, hipGraph replay",
                  "reference": "F16X3 output of the same library (3e-6 from the reference's modules)", "f16x3_ms_per_step": round(ms3, 3),
                  "f16x1": {"rel_err_batch": rel(out1, ref), "ms_per_step": round(ms1, 3)}, "order_by_error": order, "rows": rows}))
