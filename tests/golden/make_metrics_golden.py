#!/usr/bin/env python3
"""Golden vectors for cmgan_amd.metrics from the REFERENCE's own tool (src/tools/compute_metrics.py).

Run in the build container (needs /root/reference; the GPU box never runs this):
    python tests/golden/make_metrics_golden.py
The reference imports the `pesq` wheel at module load; it is not installed here, so a stub returning a
fixed MOS is registered first - the composite scores are then functions of that constant and of the
reference's own LLR / WSS / segSNR, which is exactly what the port has to reproduce.
"""
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
PESQ_STUB = 2.75
stub = types.ModuleType("pesq")
stub.pesq = lambda fs, ref, deg, mode: PESQ_STUB
sys.modules["pesq"] = stub
sys.path.insert(0, "/root/reference/src")
from tools import compute_metrics as R  # noqa: E402
sys.path.insert(0, os.path.dirname(HERE))
from metrics_signals import CASES, pair  # noqa: E402


def main():
    out = {}
    for fs, n, seed in CASES:
        clean, enhanced = pair(fs, n, seed)
        tag = f"fs{fs}_n{n}"
        out[tag + "_wss"] = R.wss(clean, enhanced, fs)
        out[tag + "_llr"] = R.llr(clean, enhanced, fs)
        snr_all, seg = R.snr(clean, enhanced, fs)
        out[tag + "_snr"], out[tag + "_segsnr"] = np.array(snr_all), seg
        out[tag + "_stoi"] = np.array(R.stoi(clean, enhanced, fs))
        out[tag + "_all"] = np.array(R.compute_metrics(clean, enhanced, fs, 0))
    out["pesq_stub"] = np.array(PESQ_STUB)
    # unequal lengths: the reference trims to the shorter one
    a, b = pair(*CASES[0])
    out["trim_all"] = np.array(R.compute_metrics(a, b[:-123], 16000, 0))
    np.savez_compressed(os.path.join(HERE, "metrics.npz"), **out)
    for k, v in out.items():
        if k.endswith("_all") or k.endswith("_stoi"):
            print(k, v)


if __name__ == "__main__":
    main()
