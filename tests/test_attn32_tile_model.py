"""CPU model of the x3 attention on 32x32x16 MFMAs (cmgan_amd/csrc/attn32_x3.hip): a lane-level numpy emulation of
one wave - the v_mfma_f32_32x32x16_f16 operand / accumulator layouts (tools/micro/mfma32_layout), the 32-token Q / K
tile images and the 16-key V group images exactly as qkv32_x3_kernel stores them (through its 32 x 17 transposition
patch), the 96-row distance window with its write / read offsets, the clamped tail tiles, the [V_hi ; V_lo] row
stacking of the P V product and the epilogue's hand-over to the 16x16 to_out fragments - checked against dense Shaw
attention (src/models/conformer.py:100-133).  This is the index algebra the kernel was written from; the GPU parity
tests (tests/test_gpu_parity.py) hold the kernel itself to the reference goldens."""
import numpy as np
import pytest

lane = np.arange(64)
A = lane & 31          # operand row / column
HH = lane >> 5


def mfma32(a, b, acc):
    """a, b: [64 lanes][8 slots]; lane l feeds A[l & 31][8 (l >> 5) + e] and B[8 (l >> 5) + e][l & 31];
    acc[l][v] = D[8 (v >> 2) + 4 (l >> 5) + (v & 3)][l & 31]."""
    Am, Bm = np.zeros((32, 16)), np.zeros((16, 32))
    for e in range(8):
        Am[A, 8 * HH + e] = a[:, e]
        Bm[8 * HH + e, A] = b[:, e]
    D = Am @ Bm
    out = acc.copy()
    for v in range(16):
        out[:, v] += D[8 * (v >> 2) + 4 * HH + (v & 3), A]
    return out


def f16split(x):
    hi = x.astype(np.float16).astype(np.float64)
    return hi, x - hi


def build_images(q, k, v, Lt):
    """Images of one (sequence, head) as qkv32_x3_kernel writes them; q, k, v: [L][16]."""
    L = q.shape[0]
    qimg = np.zeros((Lt, 2, 64, 8))
    kimg = np.zeros((Lt, 2, 64, 8))
    vimg = np.zeros((Lt * 2, 64, 8))
    c16, g16 = lane & 15, lane >> 4
    for it in range(Lt):
        T = np.full((32, 17), np.nan)
        for tb in range(2):
            l = np.minimum(it * 32 + tb * 16 + c16, L - 1)                 # clamped token of 16x16 lane (c, g)
            for which, (src, img) in enumerate(((q, qimg), (k, kimg))):
                acc = np.stack([src[l, 4 * g16 + r] for r in range(4)], 1)   # lane holds d = 4 g + r
                hi, lo = f16split(acc)
                # p = img + ((g >> 1) * 32 + 16 tb + c) * 8 + 4 (g & 1): image lane = (g >> 1) * 32 + 16 tb + c
                il = (g16 >> 1) * 32 + 16 * tb + c16
                for r in range(4):
                    img[it, 0, il, 4 * (g16 & 1) + r] = hi[:, r]
                    img[it, 1, il, 4 * (g16 & 1) + r] = lo[:, r]
            for r in range(4):
                T[16 * tb + c16, 4 * g16 + r] = v[l, 4 * g16 + r]
        for grp in range(2):
            va = np.stack([T[16 * grp + 4 * HH + r, A & 15] for r in range(4)], 1)
            vb = np.stack([T[16 * grp + 8 + 4 * HH + r, A & 15] for r in range(4)], 1)
            val = np.concatenate([va, vb], 1)
            hi, lo = f16split(val)
            vimg[it * 2 + grp] = np.where((A < 16)[:, None], hi, lo)
    return qimg, kimg, vimg


def wave_attention(qimg, kimg, vimg, rel, max_pos, L, it, log):
    """attn32_out_x3_kernel for one wave (one head, query tile `it`); returns O as the kernel stashes it:
    dict (block i, 16x16 lane) -> 4 values."""
    Lt = (L + 31) // 32
    i0 = 32 * it
    wbase = 4 * HH * 32 + A
    rbase = (32 + 4 * HH - A) * 32 + A
    qh, ql = qimg[it, 0], qimg[it, 1]
    rel_hi, rel_lo = f16split(rel)

    def load_e(n, t):
        r = np.clip(i0 - 64 * n + 32 - 32 * t - A, -max_pos, max_pos) + max_pos
        eh = np.stack([rel_hi[r, 8 * HH + e] for e in range(8)], 1)
        el = np.stack([rel_lo[r, 8 * HH + e] for e in range(8)], 1)
        return eh, el

    def load_k(n, jt):
        kt = min(2 * n + jt, Lt - 1)
        return kimg[kt, 0], kimg[kt, 1]

    def load_v(n, g4):
        return vimg[min(4 * n + g4, 2 * Lt - 1)]

    m, run, lsum = np.zeros(64), np.full(64, -np.inf), np.zeros(64)
    o = np.zeros((64, 16))
    nfull, tail = L >> 6, L & 63
    chunks = [(n, 2, True) for n in range(nfull)]
    if tail:
        chunks.append((nfull, 2 if tail > 32 else 1, False))
    R = np.full(96 * 32, np.nan)
    for n, nkt, full in chunks:
        R[:] = np.nan
        for t in range(nkt + 1):
            eh, el = load_e(n, t)
            r = mfma32(eh, qh, np.zeros((64, 16)))
            r = mfma32(eh, ql, r)
            r = mfma32(el, qh, r)
            for v in range(16):
                R[wbase + (32 * t + 8 * (v >> 2) + (v & 3)) * 32] = r[:, v]
        s = []
        for jt in range(nkt):
            sj = np.stack([R[rbase + (32 * jt + 8 * (v >> 2) + (v & 3)) * 32] for v in range(16)], 1)
            assert not np.isnan(sj).any(), "window read outside the rows written"
            kh, kl = load_k(n, jt)
            sj = mfma32(kh, qh, sj)
            sj = mfma32(kh, ql, sj)
            sj = mfma32(kl, qh, sj)
            s.append(sj)
        # softmax (re-reference whenever the band is left; the band itself is exercised on the GPU)
        mx = np.full(64, -np.inf)
        for jt in range(nkt):
            for v in range(16):
                key = 64 * n + 32 * jt + 8 * (v >> 2) + 4 * HH + (v & 3)
                x = s[jt][:, v] - m
                if not full:
                    x = np.where(key < L, x, -np.inf)
                s[jt][:, v] = x
                mx = np.maximum(mx, x)
        mx = np.maximum(mx, mx[lane ^ 32])
        newrun = np.maximum(run, mx)
        if (newrun > 12).any() or (newrun < -4).any():
            log.append("reref")
            alpha = np.where(lsum > 0, np.exp2(-newrun), 1.0)
            for jt in range(nkt):
                s[jt] = np.exp2(s[jt] - newrun[:, None])
            lsum = lsum * alpha
            o = o * alpha[:, None]
            m = m + newrun
            run = np.zeros(64)
        else:
            for jt in range(nkt):
                s[jt] = np.exp2(s[jt])
            run = newrun
        lsum = lsum + sum(sj.sum(1) for sj in s)
        for jt in range(nkt):
            for half in range(2):
                p = s[jt][:, 8 * half:8 * half + 8]
                ph, pl = f16split(p)
                va = load_v(n, 2 * jt + half)
                o = mfma32(va, ph, o)
                o = mfma32(va, pl, o)
    inv = 1.0 / (lsum + lsum[lane ^ 32])
    oa = np.stack([(o[:, r] + o[:, 8 + r]) * inv for r in range(4)], 1)
    ob = np.stack([(o[:, 4 + r] + o[:, 12 + r]) * inv for r in range(4)], 1)
    stash = {}
    for l in range(64):
        i, cq = A[l] >> 4, A[l] & 15
        stash[(i, HH[l] * 16 + cq)] = oa[l]
        stash[(i, (2 + HH[l]) * 16 + cq)] = ob[l]
    return stash


@pytest.mark.parametrize("L,max_pos,scale", [(101, 512, 1.0), (65, 512, 1.0), (70, 20, 1.0), (33, 512, 1.0),
                                             (128, 512, 1.0), (321, 512, 1.0), (96, 40, 6.0)])
def test_attn32_tile_algebra(L, max_pos, scale):
    """101 / 321: the model's sequence lengths (tail chunk of two tiles / of one key); 65: one key in the tail;
    70 / 96 with small max_pos: distances beyond the table are clamped; 128: no tail chunk; scale 6: scores leave
    the (-4, 12] band, so the re-reference path runs."""
    rng = np.random.default_rng(L)
    q = rng.standard_normal((L, 16)) * scale
    k = rng.standard_normal((L, 16))
    v = rng.standard_normal((L, 16))
    rel = rng.standard_normal((2 * max_pos + 1, 16)) * 0.5
    Lt = (L + 31) // 32
    i, j = np.arange(L)[:, None], np.arange(L)[None, :]
    E = rel[np.clip(i - j, -max_pos, max_pos) + max_pos]
    S = q @ k.T + np.einsum("id,ijd->ij", q, E)                          # log2 units (scale folded into q)
    P = np.exp2(S - S.max(1, keepdims=True))
    o_ref = (P / P.sum(1, keepdims=True)) @ v

    qimg, kimg, vimg = build_images(q, k, v, Lt)
    log = []
    for it in range(Lt):
        stash = wave_attention(qimg, kimg, vimg, rel, max_pos, L, it, log)
        for blk in range(2):
            for l16 in range(64):
                c, g = l16 & 15, l16 >> 4
                tok = 32 * it + 16 * blk + c
                if tok < L:
                    np.testing.assert_allclose(stash[(blk, l16)], o_ref[tok, 4 * g:4 * g + 4], rtol=2e-5, atol=2e-5)
    if scale > 1:
        assert "reref" in log
