"""CPU model of the training attention cores' operand images (cmgan_amd/csrc/train.hip, AT_X3): the (hi, lo) image
written by the producers (at_img4), its conversion to the MFMA operand with four v_perm_b32 (at_row_a: the byte
selectors are emulated from the instruction's definition - D.byte[k] = {S0, S1}.byte[sel.byte[k]], S1 the low dword),
the B-side forms [hi | hi], [lo | lo] (at_b_of) and the two-MFMA product at_mma, which must equal the exact product of
the fp16 splits (all four terms) over the 16 contraction indices (lane group g, slot s)."""
import numpy as np

rng = np.random.default_rng(5)


def split4(v):
    hi = v.astype(np.float16)
    lo = (v - hi.astype(np.float64)).astype(np.float16)
    return hi, lo


def at_img4(v):
    """float4 -> four dwords (hi_i | lo_i << 16)"""
    hi, lo = split4(v)
    return hi.view(np.uint16).astype(np.uint32) | (lo.view(np.uint16).astype(np.uint32) << 16)


def v_perm_b32(s0, s1, sel):
    comb = (np.uint64(s0) << np.uint64(32)) | np.uint64(s1)
    out = 0
    for k in range(4):
        b = (sel >> (8 * k)) & 0xff
        assert b < 8
        out |= int((comb >> np.uint64(8 * b)) & np.uint64(0xff)) << (8 * k)
    return np.uint32(out)


def at_row_a(d):
    """four image dwords -> f16x8 [hi0 hi1 hi2 hi3 | lo0 lo1 lo2 lo3]"""
    w = [v_perm_b32(d[1], d[0], 0x05040100), v_perm_b32(d[3], d[2], 0x05040100),
         v_perm_b32(d[1], d[0], 0x07060302), v_perm_b32(d[3], d[2], 0x07060302)]
    halves = np.array(w, dtype=np.uint32).view(np.uint16)
    return halves.view(np.float16)


def test_image_round_trip_and_operand_forms():
    for _ in range(50):
        v = (1.0 + rng.random(4)) * rng.choice([-1.0, 1.0], 4) * 2.0 ** rng.integers(-3, 7)     # activations: O(1)
        hi, lo = split4(v)
        a = at_row_a(at_img4(v))
        assert np.array_equal(a[:4], hi) and np.array_equal(a[4:], lo)
        # hi + lo reproduces the value to ~2^-22 relative (what "fp32-class" means for these products)
        np.testing.assert_allclose(a[:4].astype(np.float64) + a[4:].astype(np.float64), v, rtol=2.0 ** -21, atol=0)
    # ... as long as the halves stay NORMAL fp16 numbers: a gradient-sized value is a subnormal already in its hi half
    # (step 2^-24) and the pair keeps a handful of bits - the reason dO is stored pre-scaled by an exact power of two
    # (at_scale)
    v = np.full(4, 3.3e-6)
    a = at_row_a(at_img4(v)).astype(np.float64)
    err = abs(a[0] + a[4] - v[0]) / v[0]
    assert err > 2.0 ** -12
    s = 2.0 ** 18                                                 # exact: 3.3e-6 * 2^18 = 0.865
    a = at_row_a(at_img4(v * s)).astype(np.float64)
    assert abs((a[0] + a[4]) / s - v[0]) / v[0] < 2.0 ** -21


def test_two_mfmas_give_all_four_split_terms():
    """at_mma(a, b) = mfma(A, [bh | bh]) + mfma(A, [bl | bl]) with A = [ah | al]: per lane pair the 8-slot dot products are
    ah.bh + al.bh and ah.bl + al.bl, i.e. (ah + al).(bh + bl) summed over the four lane groups g"""
    X = rng.standard_normal((16, 16))                             # row i: the 16 contraction values of output row i
    Y = rng.standard_normal((16, 16))
    A = np.zeros((64, 8))
    Bhh = np.zeros((64, 8))
    Bll = np.zeros((64, 8))
    for l in range(64):
        c, g = l & 15, l >> 4
        a = at_row_a(at_img4(X[c, 4 * g:4 * g + 4])).astype(np.float64)
        b = at_row_a(at_img4(Y[c, 4 * g:4 * g + 4])).astype(np.float64)
        A[l] = a
        Bhh[l] = np.concatenate([b[:4], b[:4]])                   # at_b_of: shuffles of the A-form
        Bll[l] = np.concatenate([b[4:], b[4:]])
    D = np.zeros((16, 16))
    for g in range(4):
        rows = slice(16 * g, 16 * g + 16)
        D += A[rows] @ Bhh[rows].T + A[rows] @ Bll[rows].T
    Xs = sum(p.astype(np.float64) for p in split4(X))
    Ys = sum(p.astype(np.float64) for p in split4(Y))
    np.testing.assert_allclose(D, Xs @ Ys.T, rtol=0, atol=1e-12)  # exactly the product of the split operands
    np.testing.assert_allclose(D, X @ Y.T, rtol=0, atol=5e-6)     # and fp32-class against the true product
