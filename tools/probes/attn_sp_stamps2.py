"""Coarse (per query tile) cycle breakdown of attn_sp_out_x3_kernel (measurement build -DA32_STAMP=2)."""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cmgan_amd import ConformerBlock, _lib  # noqa: E402
from cmgan_amd.synth import conformer_state_dict  # noqa: E402

PH = [(0, "prologue + first front half"), (1, "units of the tiles (reference, bodies)"), (2, "epilogue up to the barrier"),
      (3, "barrier"), (4, "to_out + stores")]


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
        blk(x)
        torch.cuda.synchronize()
        fn(buf, 1)
        a = np.array(list(buf), dtype=np.float64)
        b = a[32:] if l < 200 else a[:32]
        waves, bodies = b[16], b[17]
        tot = b[:16].sum()
        print(f"--- N={n} L={l}: {int(waves)} waves, {bodies / waves:.2f} bodies/wave, {tot / waves:.0f} cycles/wave")
        for i, name in PH:
            print(f"  {name:>40}: {b[i] / waves:9.0f} cyc/wave ({100 * b[i] / tot:5.1f} %)")


if __name__ == "__main__":
    main()
