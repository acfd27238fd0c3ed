"""CPU model of the FeedForward kernel on 32x32x16 MFMAs (cmgan_amd/csrc/ffn32_x3.hip): a lane-level numpy emulation of
one wave - the v_mfma_f32_32x32x16_f16 operand / accumulator layouts, the PERMUTED channel order a lane keeps
(channel 8 q + 4 hh + r: what a lane loads is what its output accumulators hold, so the residual is their initial
value and nothing ever crosses lanes except the two LayerNorm sums), the W1 / W2 operand images api.hip packs for it,
the hidden-tile -> B-operand hand-over without data movement - checked against the dense formula of
`Scale(0.5, PreNorm(FeedForward))` (conformer.py:54-72, 136-148) with the packer's folds undone.  This is the index
algebra the kernel is written from; the GPU parity tests hold the kernel itself to the reference goldens."""
import numpy as np

lane = np.arange(64)
TOK = lane & 31
HH = lane >> 5


def mfma32(a, b, acc):
    """a, b: [64][8]; lane l feeds A[l & 31][8 (l >> 5) + e] and B[8 (l >> 5) + e][l & 31];
    acc[l][v] = D[8 (v >> 2) + 4 (l >> 5) + (v & 3)][l & 31]."""
    Am, Bm = np.zeros((32, 16)), np.zeros((16, 32))
    for e in range(8):
        Am[TOK, 8 * HH + e] = a[:, e]
        Bm[8 * HH + e, TOK] = b[:, e]
    D = Am @ Bm
    out = acc.copy()
    for v in range(16):
        out[:, v] += D[8 * (v >> 2) + 4 * HH + (v & 3), TOK]
    return out


def chan(q, r, hh):
    """channel a lane of half hh keeps in float4 q (0..7), element r"""
    return 8 * q + 4 * hh + r


def pack_w1(W1):
    """[t 8][kk 4][64 lanes][8]: A operand of hidden tile t, k-step kk; slot e <-> channel chan(2 kk + (e >> 2), e & 3, hh)"""
    img = np.zeros((8, 4, 64, 8))
    for t in range(8):
        for kk in range(4):
            for e in range(8):
                img[t, kk, :, e] = W1[32 * t + TOK, chan(2 * kk + (e >> 2), e & 3, HH)]
    return img


def pack_w2(W2):
    """[u 2][ks 16][64][8]: A operand of output tile u, k-step ks = 2 t + jp; slot e <-> hidden
    32 t + 16 jp + 8 (e >> 2) + 4 hh + (e & 3) = what accumulator register 8 jp + e of hidden tile t holds"""
    img = np.zeros((2, 16, 64, 8))
    for u in range(2):
        for ks in range(16):
            t, jp = ks >> 1, ks & 1
            for e in range(8):
                img[u, ks, :, e] = W2[32 * u + TOK, 32 * t + 16 * jp + 8 * (e >> 2) + 4 * HH + (e & 3)]
    return img


def wave_ffn(x, w1img, b1, w2img, b2, final=None):
    """x: [32 tokens][64] -> y [32][64] as the kernel computes it (no fp16 rounding: layouts only)."""
    xl = np.zeros((64, 8, 4))                                   # lane's 8 float4s
    for q in range(8):
        for r in range(4):
            xl[:, q, r] = x[TOK, chan(q, r, HH)]
    # LayerNorm statistics: 32 values in the lane + the partner half
    s = xl.reshape(64, -1).sum(1)
    mean = (s + s[lane ^ 32]) / 64.0
    d = xl - mean[:, None, None]
    v = (d * d).reshape(64, -1).sum(1)
    rstd = 1.0 / np.sqrt((v + v[lane ^ 32]) / 64.0 + 1e-5)
    xn = d * rstd[:, None, None]
    xb = [np.concatenate([xn[:, 2 * kk], xn[:, 2 * kk + 1]], 1) for kk in range(4)]          # B operands [64][8]
    # output accumulators start from the residual + second bias: register v of tile u <-> channel chan(4 u + (v >> 2), v & 3, hh)
    Y = [np.zeros((64, 16)) for _ in range(2)]
    for u in range(2):
        for v_ in range(16):
            Y[u][:, v_] = xl[:, 4 * u + (v_ >> 2), v_ & 3] + b2[chan(4 * u + (v_ >> 2), v_ & 3, HH)]
    for t in range(8):
        h = np.zeros((64, 16))
        for v_ in range(16):
            h[:, v_] = b1[32 * t + 8 * (v_ >> 2) + 4 * HH + (v_ & 3)]
        for kk in range(4):
            h = mfma32(w1img[t, kk], xb[kk], h)
        a = h / (1.0 + np.exp2(h))                              # swish_scaled on the pre-scaled hidden value
        for jp in range(2):
            p = a[:, 8 * jp:8 * jp + 8]                         # registers 8 jp .. 8 jp + 7 ARE the B operand of k-step 2 t + jp
            for u in range(2):
                Y[u] = mfma32(w2img[u, 2 * t + jp], p, Y[u])
    out = np.zeros((32, 64))
    for u in range(2):
        for v_ in range(16):
            out[TOK, chan(4 * u + (v_ >> 2), v_ & 3, HH)] = Y[u][:, v_]
    return out


def dense(x, W1, b1, W2, b2):
    mean = x.mean(1, keepdims=True)
    xn = (x - mean) / np.sqrt(x.var(1, keepdims=True) + 1e-5)
    h = xn @ W1.T + b1
    a = h / (1.0 + np.exp2(h))
    return x + a @ W2.T + b2


def test_lane_model_of_the_32x32x16_feed_forward_equals_the_dense_formula():
    rng = np.random.default_rng(0)
    x = rng.standard_normal((32, 64))
    W1, b1 = rng.standard_normal((256, 64)) * 0.2, rng.standard_normal(256) * 0.1
    W2, b2 = rng.standard_normal((64, 256)) * 0.1, rng.standard_normal(64) * 0.1
    got = wave_ffn(x, pack_w1(W1), b1, pack_w2(W2), b2)
    want = dense(x, W1, b1, W2, b2)
    assert np.abs(got - want).max() < 1e-12


def test_every_channel_and_hidden_unit_is_owned_exactly_once():
    seen = np.zeros(64, int)
    for q in range(8):
        for r in range(4):
            for hh in range(2):
                seen[chan(q, r, hh)] += 1
    assert (seen == 1).all()
    hid = np.zeros(256, int)
    for t in range(8):
        for jp in range(2):
            for e in range(8):
                for hh in range(2):
                    hid[32 * t + 16 * jp + 8 * (e >> 2) + 4 * hh + (e & 3)] += 1
    assert (hid == 1).all()


def test_the_operand_images_derive_from_the_fragment_major_blob():
    """api.hip builds the images from the fp32 fragment-major weights fm[rb][kb][lane][r] = M[16 rb + (lane & 15)]
    [16 kb + 4 (lane >> 4) + r] (csrc/weights.h): the index map it uses, checked here."""
    rng = np.random.default_rng(1)
    W1 = rng.standard_normal((256, 64))

    def fm_of(M):
        R, K = M.shape
        fm = np.zeros((R // 16, K // 16, 64, 4))
        for rb in range(R // 16):
            for kb in range(K // 16):
                for r in range(4):
                    fm[rb, kb, :, r] = M[16 * rb + (lane & 15), 16 * kb + 4 * (lane >> 4) + r]
        return fm

    def at(fm, row, col):                                        # M[row][col] read back from the fragment-major array
        return fm[row >> 4, col >> 4, (row & 15) + 16 * ((col & 15) >> 2), col & 3]

    fm = fm_of(W1)
    rows = rng.integers(0, 256, 50)
    cols = rng.integers(0, 64, 50)
    assert np.array_equal(at(fm, rows, cols), W1[rows, cols])
