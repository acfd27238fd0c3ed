"""Per-phase cycle breakdown of attn32_out_x3_kernel (measurement build -DA32_STAMP, see attn32_x3.hip).

    python -c "from cmgan_amd.build import build; build(variant='a32stamp', extra_flags=['-DA32_STAMP'])"
    gpurun -- env CMGAN_HIP_LIB=$PWD/cmgan_amd/lib/variants/a32stamp/libcmgan_hip.so python tools/probes/attn_stamps.py

Runs one ConformerBlock at the two sequence shapes of the B = 32 workload (3232 x 321 time-axis, 10272 x 101
frequency-axis) and prints, per shape, the average cycles a wave spends in each phase."""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cmgan_amd import ConformerBlock, _lib  # noqa: E402
from cmgan_amd.synth import conformer_state_dict  # noqa: E402

# (index in the device array, name, per-chunk phase?) in program order
PH = [(0, "startup", False), (8, "E wait + 1st E q MFMA", True), (1, "rest of E q + window writes", True),
      (9, "window reads", True), (2, "K q MFMAs (K wait)", True), (11, "S results, subtract, max", True),
      (3, "exp2 + sum", True), (10, "1st split + P V (V wait)", True), (4, "rest of P V", True),
      (5, "normalise + stash", False), (6, "barrier", False), (7, "to_out + store", False)]


def main():
    lib = ctypes.CDLL(_lib.LIB_PATH)
    fn = lib.cmgan_dbg_a32_stamps
    fn.argtypes = [ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_int]
    blk = ConformerBlock(dim=64, dim_head=16, heads=4, conv_kernel_size=31, mfma_mode="f16x3")
    blk.load_state_dict(conformer_state_dict(seed=3))
    buf = (ctypes.c_ulonglong * 64)()
    for n, l in ((3232, 321), (10272, 101)):
        x = torch.from_numpy(np.random.default_rng(l).standard_normal((n, l, 64)).astype(np.float32)).cuda()
        blk(x)
        torch.cuda.synchronize()
        fn(buf, 1)
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        ev[0].record()
        blk(x)
        ev[1].record()
        torch.cuda.synchronize()
        fn(buf, 1)
        a = np.array(list(buf), dtype=np.float64)
        b = a[32:] if l < 200 else a[:32]
        waves, chunks = b[16], b[17]
        print(f"--- N={n} L={l}: {int(waves)} waves, {chunks / waves:.2f} chunks/wave, conformer {ev[0].elapsed_time(ev[1]):.3f} ms")
        tot = b[:16].sum()
        for i, name, per_chunk in PH:
            per = b[i] / (chunks if per_chunk else waves)
            print(f"  {name:>28}: {b[i] / waves:9.0f} cyc/wave ({100 * b[i] / tot:5.1f} %)   "
                  f"{per:8.0f} per {'chunk' if per_chunk else 'wave'}")
        print(f"  total {tot / waves:9.0f} cyc/wave")


if __name__ == "__main__":
    main()
