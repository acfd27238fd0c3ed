"""Per-phase cycle breakdown of dwpw2t_x3_kernel (measurement build -DDT_STAMP: cmgan_amd.build.build(variant="dtstamp",
extra_flags=["-DDT_STAMP"]), CMGAN_HIP_LIB pointing at it)."""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cmgan_amd import ConformerBlock, _lib  # noqa: E402
from cmgan_amd.synth import conformer_state_dict  # noqa: E402

PH = ["prologue (operands, first window)", "prefetch issue", "chunk reads + 4x4x4 MFMAs", "Swish, split, v-tile stores",
      "epilogue fetch issue, kept-half reads", "barrier A", "pointwise product", "store, kept half, new rows",
      "barrier B"]


def main():
    lib = ctypes.CDLL(_lib.LIB_PATH)
    fn = lib.cmgan_dbg_dt_stamps
    fn.argtypes = [ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_int]
    blk = ConformerBlock(dim=64, dim_head=16, heads=4, conv_kernel_size=31, mfma_mode="f16x3")
    blk.load_state_dict(conformer_state_dict(seed=3))
    buf = (ctypes.c_ulonglong * 16)()
    for n, l in ((3232, 321), (10272, 101)):
        x = torch.from_numpy(np.random.default_rng(l).standard_normal((n, l, 64)).astype(np.float32)).cuda()
        blk(x)
        torch.cuda.synchronize()
        fn(buf, 1)
        blk(x)
        torch.cuda.synchronize()
        fn(buf, 1)
        b = np.array(list(buf), dtype=np.float64)
        waves, tiles = b[14], b[15]
        tot = b[:9].sum()
        print(f"--- N={n} L={l}: {int(waves)} waves, {tiles / waves:.2f} tiles/wave, {tot / waves:.0f} cycles/wave, "
              f"{tot / tiles:.0f} cycles per tile")
        for i, name in enumerate(PH):
            per = b[i] / (waves if i == 0 else tiles)
            print(f"  {name:>40}: {per:8.0f} cyc per {'wave' if i == 0 else 'tile'} ({100 * b[i] / tot:5.1f} %)")


if __name__ == "__main__":
    main()
