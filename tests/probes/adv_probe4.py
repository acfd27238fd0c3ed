"""Stage-by-stage check of the complex decoder's backward through its workspace planes (debug probe)."""
import sys, numpy as np, torch
import torch.nn.functional as F
import os; ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from oracle import cmgan_oracle as O
from oracle.weights import make_state_dict
from cmgan_amd.training import DecoderTrain
from conftest import rel_err
DEV = "cuda:0"
sd = make_state_dict(seed=0)
pre = "complex_decoder."
st = {k[len(pre):]: v for k, v in sd.items() if k.startswith(pre)}
B, T, Fe = 2, 33, 101
W = 2 * Fe
Me, Ms = B * T * Fe, 2 * B * T * Fe
al = lambda n: (n + 63) // 64 * 64
off = {}
cur = 0
for name, n in (("img", 3 * 8192), ("imgT", 3 * 8192), ("d", Me * 64), ("s", Ms * 64), ("a", Ms * 64), ("t1", Ms), ("g", Ms * 64)):
    off[name] = cur; cur += al(n)
for trial in range(3):
    rng = np.random.Generator(np.random.PCG64(9 + trial))
    dy = torch.from_numpy(rng.standard_normal((B, 2, T, 2 * Fe - 1)).astype(np.float32))
    x = torch.from_numpy(rng.standard_normal((B, 64, T, Fe)).astype(np.float32))
    xr = x.clone().requires_grad_(True)
    p = "complex_decoder"
    with torch.enable_grad():
        d = O.dense_block(sd, p + ".dense_block", xr); d.retain_grad()
        s = O.sub_pixel(sd, p + ".sub_pixel", d); s.retain_grad()
        a = O._in_prelu(sd, p + ".norm", p + ".prelu", s); a.retain_grad()
        y = F.conv2d(a, sd[p + ".conv.weight"], sd[p + ".conv.bias"])
        y.backward(dy)
    dec = DecoderTrain("complex", st, device=DEV)
    yh = dec.forward(x.permute(0, 2, 3, 1).contiguous().to(DEV))
    ws = dec._ws.view(torch.float32)
    plane = lambda name, rows: ws[off[name]:off[name] + rows * 64].view(rows, 64).clone()
    s_h, a_h = plane("s", Ms), plane("a", Ms)
    cl = lambda t: t.detach().permute(0, 2, 3, 1).reshape(-1, 64)
    print(f"trial {trial}: fwd s {rel_err(s_h, cl(s)):.1e} a {rel_err(a_h, cl(a)):.1e} y {rel_err(yh, y.detach().permute(0, 2, 3, 1)):.1e}")
    outs = []
    for rep in range(2):
        dx, gr = dec.backward(dy.permute(0, 2, 3, 1).contiguous().to(DEV))
        torch.cuda.synchronize()
        g_h, d_h = plane("g", Ms), plane("d", Me)
        print(f"   rep {rep}: dz_s {rel_err(g_h, cl(s.grad)):.1e} dd {rel_err(d_h, cl(d.grad)):.1e} dx {rel_err(dx.permute(0, 3, 1, 2), xr.grad):.1e} "
              f"norm.bias {rel_err(gr['norm.bias'], torch.autograd.grad(y, [], allow_unused=True) if False else gr['norm.bias']):.1e}")
        outs.append((g_h, d_h, dx.clone()))
        if rep == 0:
            bad = (g_h.cpu() - cl(s.grad)).abs().amax(dim=1)
            idx = torch.nonzero(bad > 1e-3 * float(s.grad.abs().max())).flatten()
            print("      bad dz_s rows:", idx.numel(), idx[:12].tolist(), " (row = (b*T+t)*W + f, W =", W, ")")
    print("   deterministic:", all(torch.equal(outs[0][i], outs[1][i]) for i in range(3)))
