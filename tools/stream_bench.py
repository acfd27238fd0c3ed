#!/usr/bin/env python3
"""BASELINE.json configs[4]: a 10 s 16 kHz clip processed as 400-frame windows from ONE captured hipGraph
(cmgan_amd.streaming.enhance_windows: fixed windows of W samples with C samples of recomputed context, the
per-window contract of DESIGN.md section 8), next to the reference's own long-audio rule (reshape into rows,
evaluation.py:30-34).  Prints one JSON line for profiles/."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cmgan_amd import TSCNet
from cmgan_amd.evaluation import enhance_one_track
from cmgan_amd.streaming import enhance_stream, enhance_windows
from cmgan_amd.synth import make_state_dict, synthetic_clips

model = TSCNet(64, 201).load_state_dict(make_state_dict(0)).eval()
noisy = synthetic_clips(1, 160000, seed=3).cuda()
W, C = 40000, 4000                                   # 400-frame windows, 40 frames of context each side


def timed(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


eng = model.engine
stats = eng.tscnet_forward_stats(eng.stft_compress(noisy[:, :44000], eng.rms_scale(noisy)))[2]     # as bench.py's stream leg
res = {}
import os
def staged(n, la):
    def run():
        old = os.environ.get("CMGAN_STREAM_STAGES")
        os.environ["CMGAN_STREAM_STAGES"] = str(n)
        try:
            return enhance_stream(model, noisy, 400, 40, la, stats=stats, graph=True)
        finally:
            os.environ.pop("CMGAN_STREAM_STAGES") if old is None else os.environ.__setitem__("CMGAN_STREAM_STAGES", old)
    return run
for name, fn in (("carried_state_graph_40_40", lambda: enhance_stream(model, noisy, 400, 40, 40, stats=stats, graph=True)),
                 ("carried_state_2_stages_40_40", staged(2, 40)), ("carried_state_3_stages_40_40", staged(3, 40)),
                 ("carried_state_2_stages_40_0", staged(2, 0)), ("carried_state_3_stages_40_0", staged(3, 0)),
                 ("carried_state_not_pipelined_40_0", lambda: enhance_stream(model, noisy, 400, 40, 0, stats=stats, graph=True, pipeline=False)),
                 ("carried_state_graph_40_0", lambda: enhance_stream(model, noisy, 400, 40, 0, stats=stats, graph=True)),
                 ("carried_state_graph_40_40_not_pipelined", lambda: enhance_stream(model, noisy, 400, 40, 40, stats=stats, graph=True, pipeline=False)),
                 ("carried_state_eager_40_40", lambda: enhance_stream(model, noisy, 400, 40, 40, stats=stats, graph=False)),
                 ("windows_graph_batch1", lambda: enhance_windows(model, noisy, W, C, batch=1, graph=True)),
                 ("windows_graph_batch4", lambda: enhance_windows(model, noisy, W, C, batch=4, graph=True)),
                 ("windows_eager_batch4", lambda: enhance_windows(model, noisy, W, C, batch=4, graph=False)),
                 ("reference_rows_rule_cut40000", lambda: enhance_one_track(model, noisy, cut_len=40000)),
                 ("whole_clip_one_row", lambda: enhance_one_track(model, noisy))):
    dt = timed(fn)
    res[name] = {"ms_per_10s_clip": round(1e3 * dt, 3), "frames_per_s": round(1601 / dt, 1),
                 "real_time_factor": round(10.0 / dt, 1)}
print(json.dumps({"workload": "configs[4]: 10 s 16 kHz clip, 400-frame windows (W=40000 samples, context 4000), "
                              "TSCNet(64,201) random-init, f16x3 mode, 1 x MI355X", "results": res}))
