"""CPU model of the training attention cores (cmgan_amd/csrc/train.hip: at_fwd / at_dq / at_dkv / at_de kernels +
at_window / at_de_scatter): a lane-level numpy emulation of one wave - the 16x16x4 MFMA operand / accumulator layouts of
csrc/common.hip.h, the A-type / row-type fragment loads with their clamped tail offsets, the relative-position window,
the skew and unskew LDS patches (pitches AT_PA / AT_PB / AT_PS with zero pads), the band-product and E-row reuse of
the forward / dq loops, the tile-diagonal dE slabs and their scatter - checked against the dense formulas of Shaw
attention and its gradients (src/models/conformer.py:100-133).  This is the index algebra the kernels were written
from; the GPU parity tests (tests/test_gpu_training.py) hold the kernels themselves to the reference's autograd."""
import numpy as np
import pytest

PA, PB, PS = 20, 36, 48                      # AT_PA, AT_PB, AT_PS
LOG2E = 1.4426950408889634
QS = 0.25 * LOG2E                            # AT_QSCALE
lane = np.arange(64)
c, g = lane & 15, lane >> 4


def mfma(a, b, acc):
    """v_mfma_f32_16x16x4_f32: lane l feeds A[l & 15][l >> 4] and B[l >> 4][l & 15]; acc[l][r] = D[4 (l >> 4) + r][l & 15]."""
    A, B = np.zeros((16, 4)), np.zeros((4, 16))
    A[c, g], B[g, c] = a, b
    D = A @ B
    out = acc.copy()
    for r in range(4):
        out[:, r] += D[4 * g + r, c]
    return out


def at_dot(a, b):
    acc = np.zeros((64, 4))
    for s in range(4):
        acc = mfma(a[:, s], b[:, s], acc)
    return acc


def ldg4(arr, off):
    return np.stack([arr[off + s] for s in range(4)], 1)


def skew_t(eq0, eq1):
    buf = np.full(32 * PA, np.nan)
    for r in range(4):
        buf[(4 * g + r) * PA + c] = eq0[:, r]
        buf[(16 + 4 * g + r) * PA + c] = eq1[:, r]
    return np.stack([buf[(15 + c - 4 * g) * PA + c - r * PA] for r in range(4)], 1)


def skew(qe0, qe1):
    buf = np.full(16 * PB, np.nan)
    for r in range(4):
        buf[(4 * g + r) * PB + c] = qe0[:, r]
        buf[(4 * g + r) * PB + 16 + c] = qe1[:, r]
    return np.stack([buf[4 * g * (PB + 1) + 15 - c + r * (PB + 1)] for r in range(4)], 1)


def zero_pads():
    p = np.full(16 * PS, np.nan)             # NaN = never written: a read outside data + pads would poison the result
    for i in range(512):
        p[(i >> 5) * PS + ((i & 31) if (i & 31) < 16 else 16 + (i & 31))] = 0.0
    return p


@pytest.mark.parametrize("L,max_pos", [(37, 20), (32, 512), (7, 5), (65, 512)])
def test_attention_core_tile_algebra(L, max_pos):
    """37 / 20: ragged last block and distances beyond +-max_pos (clamped table rows share gradient); 32: no ragged
    block; 7: a single ragged block; 65: one row in the last block."""
    rng = np.random.default_rng(L)
    nb, nfull, W = (L + 15) // 16, L >> 4, 16 * ((L + 15) // 16) + 16
    N = 2
    qkv = rng.standard_normal((N * L + 40, 192)).ravel()
    dOf = rng.standard_normal((N * L + 40, 64)).ravel()
    rel = rng.standard_normal((2 * max_pos + 1, 16))
    ewin = np.zeros((2 * W + 1) * 16)                                   # at_window_kernel
    for idx in range(ewin.size):
        dist = min(max((idx >> 4) - W, -max_pos), max_pos)
        ewin[idx] = rel[dist + max_pos, idx & 15]
    n, h = 1, 2
    nh, base = n * 4 + h, n * L
    q = qkv.reshape(-1, 192)[base:base + L, 16 * h:16 * h + 16]
    k = qkv.reshape(-1, 192)[base:base + L, 64 + 16 * h:64 + 16 * h + 16]
    v = qkv.reshape(-1, 192)[base:base + L, 128 + 16 * h:128 + 16 * h + 16]
    dO = dOf.reshape(-1, 64)[base:base + L, 16 * h:16 * h + 16]

    # ---- dense reference ----
    i, j = np.arange(L)[:, None], np.arange(L)[None, :]
    E = rel[np.clip(i - j, -max_pos, max_pos) + max_pos]
    S = (q @ k.T + np.einsum("id,ijd->ij", q, E)) * 0.25
    mx = S.max(1, keepdims=True)
    P = np.exp(S - mx)
    ls = P.sum(1, keepdims=True)
    P /= ls
    lse_ref = (mx + np.log(ls))[:, 0]
    o_ref = P @ v
    Dr = (dO * o_ref).sum(1)
    dS = P * (dO @ v.T - Dr[:, None])
    dq_ref = 0.25 * (dS @ k + np.einsum("ij,ijd->id", dS, E))
    dk_ref, dv_ref = 0.25 * dS.T @ q, P.T @ dO
    drel_ref = np.zeros_like(rel)
    for a in range(L):
        for b in range(L):
            drel_ref[np.clip(a - b, -max_pos, max_pos) + max_pos] += 0.25 * dS[a, b] * q[a]

    lsebuf = np.zeros(N * 4 * L + 40)
    lsebuf[nh * L:nh * L + L] = lse_ref
    Dbuf = np.zeros((N * L + 40) * 4)
    Dbuf.reshape(-1, 4)[base:base + L, h] = Dr
    qh, gh, lh, Dh = base * 192 + 16 * h, base * 64 + 16 * h, nh * L, base * 4 + h
    off_a = lambda R0, stride: np.where(R0 + c < L, c, L - 1 - R0) * stride + 4 * g            # at_off_a
    off_b = lambda R0, stride, r: np.where(R0 + 4 * g + r < L, 4 * g + r, L - 1 - R0) * stride + c
    la, lb, le, lg = c * 192 + 4 * g, 4 * g * 192 + c, c * 16 + 4 * g, g * 16 + c
    la_t = off_a(16 * nfull, 192)
    lb_t = [off_b(16 * nfull, 192, r) for r in range(4)]
    vt = [16 * nfull + 4 * g + r < L for r in range(4)]
    rows4 = lambda I0, TAIL: [np.where(I0 + 4 * g + r < L, 4 * g + r, L - 1 - I0) if TAIL else 4 * g + r for r in range(4)]

    # ---- at_fwd_kernel ----
    o, lse = np.zeros((L, 16)), np.zeros(L)
    for blk in range(nb):
        I0, e_blk = 16 * blk, (16 * blk - 15 + W) * 16
        qa = ldg4(qkv, qh + I0 * 192 + off_a(I0, 192)) * QS
        ot, m, l = np.zeros((64, 4)), np.full(64, -1e30), np.zeros(64)
        eq1 = at_dot(ldg4(ewin, e_blk + 256 + le), qa)
        for jb in range(nb):
            TAIL, kp = jb >= nfull, qh + 64 + jb * 16 * 192
            ka = ldg4(qkv, kp + (la_t if TAIL else la))
            vb = np.stack([qkv[kp + 64 + (lb_t[r] if TAIL else lb + r * 192)] for r in range(4)], 1)
            eq0 = at_dot(ldg4(ewin, e_blk - jb * 256 + le), qa)
            sc = at_dot(ka, qa) + skew_t(eq0, eq1)
            eq1 = eq0
            if TAIL:
                for r in range(4):
                    sc[:, r] = np.where(vt[r], sc[:, r], -1e30)
            mxl = sc.max(1)
            mg = np.array([mxl[c == cc].max() for cc in c])                 # red_g_max
            if (mg > m).any():                                             # the lazy rescale
                mn = np.maximum(m, mg)
                corr = np.exp2(m - mn)
                l, ot, m = l * corr, ot * corr[:, None], mn
            p = np.exp2(sc - m[:, None])
            l = l + p.sum(1)                                               # per lane; the lane groups are summed at the end
            for r in range(4):
                ot = mfma(vb[:, r], p[:, r], ot)
        l = np.array([l[c == cc].sum() for cc in c])                       # red_g_sum
        for ln in range(64):
            if I0 + c[ln] < L:
                o[I0 + c[ln], 4 * g[ln]:4 * g[ln] + 4] = ot[ln] / l[ln]
                lse[I0 + c[ln]] = (m[ln] + np.log2(l[ln])) * 0.6931471805599453
    assert np.abs(o - o_ref).max() < 1e-12 and np.abs(lse - lse_ref).max() < 1e-12

    # ---- at_dq_kernel ----
    dq = np.zeros((L, 16))
    for blk in range(nb):
        I0, e_blk = 16 * blk, (16 * blk - 15 + W) * 16
        ri = np.where(I0 + c < L, I0 + c, L - 1)
        qa, ga = ldg4(qkv, qh + ri * 192 + 4 * g) * QS, ldg4(dOf, (base + ri) * 64 + 16 * h + 4 * g)
        lsec, Di = lsebuf[nh * L + ri] * LOG2E, Dbuf[(base + ri) * 4 + h]
        buf2, acc = zero_pads(), np.zeros((64, 4))
        eq1 = at_dot(ldg4(ewin, e_blk + 256 + le), qa)
        eb_prev = np.stack([ewin[e_blk + 256 + lg + 64 * s] for s in range(4)], 1)
        for jb in range(nb):
            TAIL, kp, ep = jb >= nfull, qh + 64 + jb * 16 * 192, e_blk - jb * 256
            ka, va = ldg4(qkv, kp + (la_t if TAIL else la)), ldg4(qkv, kp + 64 + (la_t if TAIL else la))
            kb = np.stack([qkv[kp + (lb_t[r] if TAIL else lb + r * 192)] for r in range(4)], 1)
            eb_lo = np.stack([ewin[ep + lg + 64 * s] for s in range(4)], 1)
            eq0 = at_dot(ldg4(ewin, ep + le), qa)
            p = np.exp2(at_dot(ka, qa) + skew_t(eq0, eq1) - lsec[:, None])
            eq1 = eq0
            if TAIL:
                for r in range(4):
                    p[:, r] = np.where(vt[r], p[:, r], 0)
            ds = p * (at_dot(va, ga) - Di[:, None])
            for r in range(4):
                acc = mfma(ds[:, r], kb[:, r], acc)
            for s in range(4):
                buf2[c * PS + 16 + 4 * g + s] = ds[:, s]
            for s in range(8):
                acc = mfma(buf2[c * (PS + 1) + 3 - g + 4 * (7 - s)], eb_lo[:, s] if s < 4 else eb_prev[:, s - 4], acc)
            eb_prev = eb_lo
        for ln in range(64):
            for r in range(4):
                if I0 + 4 * g[ln] + r < L:
                    dq[I0 + 4 * g[ln] + r, c[ln]] = acc[ln, r] * 0.25
    assert np.abs(dq - dq_ref).max() < 1e-12

    def tile_pds(qa, ga, ka, va, e0, e1, lser, Drr):                       # at_tile_pds
        p = np.exp2(at_dot(qa, ka) + skew(at_dot(qa, e0), at_dot(qa, e1)) - lser)
        return p, p * (at_dot(ga, va) - Drr)

    def query_block(I0, TAIL):
        qp, gp = qh + I0 * 192, gh + I0 * 64
        qa = ldg4(qkv, qp + (off_a(I0, 192) if TAIL else c * 192 + 4 * g)) * QS
        ga = ldg4(dOf, gp + (off_a(I0, 64) if TAIL else c * 64 + 4 * g))
        rr = rows4(I0, TAIL)
        qb = np.stack([qkv[qp + rr[r] * 192 + c] for r in range(4)], 1)
        gb = np.stack([dOf[gp + rr[r] * 64 + c] for r in range(4)], 1)
        lser = np.stack([lsebuf[lh + I0 + rr[r]] * LOG2E for r in range(4)], 1)
        Drr = np.stack([Dbuf[Dh + (I0 + rr[r]) * 4] for r in range(4)], 1)
        return qa, ga, qb, gb, lser, Drr

    # ---- at_dkv_kernel ----
    dk, dv = np.zeros((L, 16)), np.zeros((L, 16))
    for blk in range(nb):
        J0, e_blk = 16 * blk, (W - 16 * blk - 15) * 16
        rj = J0 * 192 + off_a(J0, 192)
        ka, va = ldg4(qkv, qh + 64 + rj), ldg4(qkv, qh + 128 + rj)
        ak, av = np.zeros((64, 4)), np.zeros((64, 4))
        e0 = ldg4(ewin, e_blk + le)
        for ib in range(nb):
            TAIL, I0 = ib >= nfull, 16 * ib
            qa, ga, qb, gb, lser, Drr = query_block(I0, TAIL)
            e1 = ldg4(ewin, e_blk + ib * 256 + 256 + le)
            p, ds = tile_pds(qa, ga, ka, va, e0, e1, lser, Drr)
            e0 = e1
            if TAIL:
                for r in range(4):
                    bad = I0 + 4 * g + r >= L
                    p[:, r], ds[:, r] = np.where(bad, 0, p[:, r]), np.where(bad, 0, ds[:, r])
            for r in range(4):
                ak, av = mfma(ds[:, r], qb[:, r], ak), mfma(p[:, r], gb[:, r], av)
        for ln in range(64):
            for r in range(4):
                if J0 + 4 * g[ln] + r < L:
                    dk[J0 + 4 * g[ln] + r, c[ln]], dv[J0 + 4 * g[ln] + r, c[ln]] = ak[ln, r] * 0.25, av[ln, r]
    assert np.abs(dk - dk_ref).max() < 1e-12 and np.abs(dv - dv_ref).max() < 1e-12

    # ---- at_de_kernel + at_de_scatter_kernel ----
    slabs = np.full((2 * nb - 1, 32, 16), np.nan)                          # every slab row must be written exactly once
    for t in range(nb):
        buf2 = zero_pads()
        for seg in range(2):
            if seg == 1 and t == 0:
                break
            delta, ntile = (t, nb - t) if seg == 0 else (t - nb, t)
            ib0, jb0 = (t, 0) if seg == 0 else (0, nb - t)
            ep = (16 * delta - 15 + W) * 16
            e0, e1 = ldg4(ewin, ep + le), ldg4(ewin, ep + 256 + le)
            de0, de1 = np.zeros((64, 4)), np.zeros((64, 4))
            for kk in range(ntile):
                TAIL = ib0 + kk >= nfull or jb0 + kk >= nfull
                I0, J0 = 16 * (ib0 + kk), 16 * (jb0 + kk)
                qa, ga, qb, _, lser, Drr = query_block(I0, TAIL)
                kp = qh + 64 + J0 * 192
                oj = off_a(J0, 192) if TAIL else c * 192 + 4 * g
                _, ds = tile_pds(qa, ga, ldg4(qkv, kp + oj), ldg4(qkv, kp + 64 + oj), e0, e1, lser, Drr)
                if TAIL:
                    for r in range(4):
                        ds[:, r] = np.where((I0 + 4 * g + r >= L) | (J0 + c >= L), 0, ds[:, r])
                assert not np.isnan(ds).any()
                for r in range(4):
                    buf2[(4 * g + r) * PS + 16 + c] = ds[:, r]
                for r in range(4):
                    row = 4 * g * (PS + 1) + 15 - c + r * (PS + 1)
                    de0, de1 = mfma(buf2[row + 16], qb[:, r], de0), mfma(buf2[row], qb[:, r], de1)
            for ln in range(64):
                for r in range(4):
                    assert np.isnan(slabs[delta + nb - 1, 4 * g[ln] + r, c[ln]])
                    slabs[delta + nb - 1, 4 * g[ln] + r, c[ln]] = de0[ln, r] * 0.25
                    slabs[delta + nb - 1, 16 + 4 * g[ln] + r, c[ln]] = de1[ln, r] * 0.25
    assert not np.isnan(slabs).any()
    drel = np.zeros_like(rel)
    for e in range(2 * max_pos + 1):
        dist = e - max_pos
        lo = -(L - 1) if dist == -max_pos else dist
        hi = L - 1 if dist == max_pos else dist
        lo, hi = max(lo, -(L - 1)), min(hi, L - 1)
        for dd in range(lo, hi + 1):
            d_lo = max((dd - 15 + 16 * nb + 15) // 16 - nb, -(nb - 1))
            d_hi = min((dd + 15 + 16 * nb) // 16 - nb, nb - 1)
            for delta in range(d_lo, d_hi + 1):
                drel[e] += slabs[delta + nb - 1, dd - (16 * delta - 15)]
    assert np.abs(drel - drel_ref).max() < 1e-12
