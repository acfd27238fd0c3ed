"""Where does the fused conv module (tools/probes/convmod_fuse_r4.patch, built as variant `fused`) differ from the two-kernel
path?  Run once per library (CMGAN_HIP_LIB), then with `cmp` to print the error per sequence position."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
OUT = "gpurun_out"


def run(tag):
    from cmgan_amd import ConformerBlock
    from cmgan_amd.synth import conformer_state_dict
    blk = ConformerBlock(dim=64, dim_head=16, heads=4, conv_kernel_size=31, mfma_mode="f16x3")
    blk.load_state_dict(conformer_state_dict(seed=3)).eval()
    for n, l in ((300, 321), (300, 101), (3, 321), (5, 40)):
        x = torch.from_numpy(np.random.default_rng(l + n).standard_normal((n, l, 64)).astype(np.float32)).cuda()
        y = blk(x)
        np.save(f"{OUT}/cm_{tag}_{n}_{l}.npy", y.cpu().numpy())


def cmp():
    for n, l in ((300, 321), (300, 101), (3, 321), (5, 40)):
        a, b = np.load(f"{OUT}/cm_ref_{n}_{l}.npy"), np.load(f"{OUT}/cm_fused_{n}_{l}.npy")
        d = np.abs(a - b)
        scale = np.abs(a).max()
        per_l = d.max(axis=(0, 2)) / scale
        per_n = d.max(axis=(1, 2)) / scale
        bad_l = np.nonzero(per_l > 1e-5)[0]
        bad_n = np.nonzero(per_n > 1e-5)[0]
        print(f"N={n} L={l}: max rel {d.max() / scale:.3e}; positions off: {len(bad_l)} "
              f"{bad_l[:12].tolist()}..{bad_l[-6:].tolist()}; sequences off: {len(bad_n)} {bad_n[:16].tolist()}")
        if len(bad_n):
            k = bad_n[0]
            pl = d[k].max(axis=1) / scale
            print("   first bad sequence", k, "positions", np.nonzero(pl > 1e-5)[0][:24].tolist(),
                  "channels", np.nonzero(d[k].max(axis=0) / scale > 1e-5)[0][:24].tolist())


if __name__ == "__main__":
    cmp() if sys.argv[1] == "cmp" else run(sys.argv[1])
