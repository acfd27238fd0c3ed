"""CPU model of the index algebra of the discriminator's convolution kernels (cmgan_amd/csrc/disc.hip,
dc_conv_gemm_kernel / dc_conv_wgrad_kernel): the 4 x 4, stride-2, pad-1 convolution of the reference's metric
discriminator (src/models/discriminator.py:33-45) as a GEMM over gathered rows, its data gradient as four parity classes
of 2 x 2 taps, and its weight gradient as a token contraction of im2col rows - each restated with the kernels' own index
formulas in numpy and checked against torch's conv2d and its autograd (the GPU tests hold the kernels' numbers to the
reference modules; this pins the formulas they were written from, including odd input sizes such as 321 x 201)."""
import numpy as np
import pytest
import torch


def gather_fwd(x, to, fo, tap):
    """MODE 0 row element: input (2 to - 1 + kt, 2 fo - 1 + kf), kt = tap >> 2, kf = tap & 3 (zero outside)."""
    ti, fi = 2 * to - 1 + (tap >> 2), 2 * fo - 1 + (tap & 3)
    if 0 <= ti < x.shape[0] and 0 <= fi < x.shape[1]:
        return x[ti, fi]
    return np.zeros(x.shape[2])


@pytest.mark.parametrize("Ti,Fi,Ci,Co", [(9, 7, 2, 3), (8, 10, 3, 2), (21, 13, 1, 2)])
def test_forward_dgrad_and_wgrad_index_maps(Ti, Fi, Ci, Co):
    rng = np.random.default_rng(Ti * 100 + Fi)
    x = rng.standard_normal((Ti, Fi, Ci))                        # channels-last [T, F, C] (axes swapped vs torch's [C, F, T])
    w = rng.standard_normal((Co, Ci, 4, 4))                      # reference layout [co][ci][kh = kf][kw = kt]
    To, Fo = Ti // 2, Fi // 2
    xt = torch.from_numpy(x.transpose(2, 1, 0)[None]).requires_grad_(True)       # [1, C, F, T]
    wt = torch.from_numpy(w).requires_grad_(True)
    y = torch.nn.functional.conv2d(xt, wt, stride=2, padding=1)                   # [1, Co, Fo, To]
    assert y.shape[2:] == (Fo, To)
    dy = rng.standard_normal((To, Fo, Co))
    y.backward(torch.from_numpy(dy.transpose(2, 1, 0)[None]))
    want_y = y.detach().numpy()[0].transpose(2, 1, 0)             # [To, Fo, Co]
    want_dx = xt.grad.numpy()[0].transpose(2, 1, 0)               # [Ti, Fi, Ci]
    want_dw = wt.grad.numpy()                                     # [Co, Ci, kh, kw]

    # the kernels' weight layouts (dc_pack_kernel): tap = kw * 4 + kh with (kt, kf) = (kw, kh)
    wf = np.zeros((16, Ci, Co))                                   # [tap][ci][co]
    for co in range(Co):
        for ci in range(Ci):
            for kh in range(4):
                for kw in range(4):
                    wf[kw * 4 + kh, ci, co] = w[co, ci, kh, kw]
    wb = wf.transpose(0, 2, 1)                                    # [tap][co][ci]

    # forward: out[pos][co] = sum_m xcol[pos][m] wf[m][co], m = tap * Ci + ci
    got_y = np.zeros((To, Fo, Co))
    for to in range(To):
        for fo in range(Fo):
            row = np.concatenate([gather_fwd(x, to, fo, tap) for tap in range(16)])
            got_y[to, fo] = row @ wf.reshape(16 * Ci, Co)
    np.testing.assert_allclose(got_y, want_y, rtol=1e-12, atol=1e-12)

    # data gradient: class (rt, rf), position (a, c) -> input (2a + rt, 2c + rf); taps kt = 1 - rt + 2 jt, kf = 1 - rf + 2 jf
    # from the outputs (a + rt - jt, c + rf - jf); every input position is written by exactly one class
    got_dx = np.full((Ti, Fi, Ci), np.nan)
    for rt in range(2):
        for rf in range(2):
            X, Y = (Ti - rt + 1) // 2, (Fi - rf + 1) // 2
            for a in range(X):
                for c in range(Y):
                    acc = np.zeros(Ci)
                    for j in range(4):
                        jt, jf = j >> 1, j & 1
                        to, fo = a + rt - jt, c + rf - jf
                        if 0 <= to < To and 0 <= fo < Fo:
                            tap = (1 - rt + 2 * jt) * 4 + (1 - rf + 2 * jf)
                            acc += dy[to, fo] @ wb[tap]
                    assert np.isnan(got_dx[2 * a + rt, 2 * c + rf]).all()
                    got_dx[2 * a + rt, 2 * c + rf] = acc
    assert not np.isnan(got_dx).any()
    np.testing.assert_allclose(got_dx, want_dx, rtol=1e-12, atol=1e-12)

    # weight gradient: G[tap][ci][co] = sum_pos xcol[pos][tap, ci] dy[pos][co], scattered back to [co][ci][kh][kw]
    G = np.zeros((16, Ci, Co))
    for to in range(To):
        for fo in range(Fo):
            for tap in range(16):
                G[tap] += np.outer(gather_fwd(x, to, fo, tap), dy[to, fo])
    got_dw = np.zeros_like(w)
    for co in range(Co):
        for ci in range(Ci):
            for kh in range(4):
                for kw in range(4):
                    got_dw[co, ci, kh, kw] = G[kw * 4 + kh, ci, co]
    np.testing.assert_allclose(got_dw, want_dw, rtol=1e-12, atol=1e-12)


def test_wgrad_launch_shapes():
    """rows per block / chunks of dc_conv_wgrad_kernel for the four layers of Discriminator(ndf = 16): a block's
    accumulators are MB * Co = 8192 (first layer: all 32 x 16), thread tiles cover them exactly, and the partial slabs
    fit the DC_SPLIT * maxw floats of the workspace."""
    DC_SPLIT, maxw = 32, 16 * 64 * 128
    for Ci, Co in ((2, 16), (16, 32), (32, 64), (64, 128)):
        Mtot = 16 * Ci
        MB = min(Mtot, 8192 // Co)
        groups = Mtot // MB
        RM, RN = (2, 1) if Ci * Co == 32 else (8, 4)
        assert Mtot % MB == 0 and (MB // RM) * (Co // RN) == 256 and MB <= 256 and Co <= 128
        nw = 16 * Ci * Co
        ns = min(1024 // groups, DC_SPLIT * maxw // nw)
        assert ns >= 1 and ns * nw <= DC_SPLIT * maxw
