"""CPU model of the operand conventions the split-f16 training kernels share (cmgan_amd/csrc/train_x3.hip):
  * pack_x3_kernel's weight image  [R/16][K/32][hi | lo][64 lanes][8 halfs],
    lane (c, g) slot e  <->  W[16 rb + c][32 m + 16 (e >> 2) + 4 g + (e & 3)]
  * the activation operand of k32 block m = split8(block 2m, block 2m + 1) of the chain layout, in which lane (c, g)
    holds features 16 j + 4 g + r (r = 0 .. 3) of token c for every 16-feature block j
  * lin_acc_x3: out block ob accumulates, over m, the three split products of image fragment (ob, m) with operand m,
    and v_mfma_f32_16x16x32_f16 contracts slot e of lane group g on the A side with slot e of lane group g on the B side.
Emulated in numpy with exact fp16 splits and checked against W @ x: the index algebra every x3 kernel of the training
path (FeedForward, conv module, attention projections, their backwards with transposed images) was written from."""
import numpy as np
import pytest

lane = np.arange(64)
C, G = lane & 15, lane >> 4


def split(x):
    hi = x.astype(np.float16)
    lo = (x - hi.astype(np.float64)).astype(np.float16)
    return hi.astype(np.float64), lo.astype(np.float64)


def pack_x3(W, transpose=False):
    """image[rb][m][hi|lo][lane][e] of W (or of W^T) exactly as pack_x3_kernel indexes it"""
    Wm = W.T if transpose else W
    R, K = Wm.shape
    img = np.zeros((R // 16, K // 32, 2, 64, 8))
    for rb in range(R // 16):
        for m in range(K // 32):
            for e in range(8):
                col = 32 * m + 16 * (e >> 2) + 4 * G + (e & 3)
                hi, lo = split(Wm[16 * rb + C, col])
                img[rb, m, 0, :, e], img[rb, m, 1, :, e] = hi, lo
    return img


def chain_operand(x, m):
    """split8(block 2m, block 2m + 1) for a tile x [16 tokens][K]: lane (c, g) slot e <-> x[c][32 m + 16 (e >> 2) + 4 g + (e & 3)]"""
    v = np.zeros((64, 8))
    for e in range(8):
        v[:, e] = x[C, 32 * m + 16 * (e >> 2) + 4 * G + (e & 3)]
    return split(v)


def mfma_16x16x32(a, b):
    """D[row = A-side lane & 15][col = B-side lane & 15] = sum over (g, e) of a[lane (row, g)][e] * b[lane (col, g)][e];
    returned in the accumulator layout: lane (c, g) register r holds D[4 g + r][c]"""
    D = np.zeros((16, 16))
    for g in range(4):
        D += a[16 * g:16 * g + 16] @ b[16 * g:16 * g + 16].T
    out = np.zeros((64, 4))
    for r in range(4):
        out[:, r] = D[4 * G + r, C]
    return out


def lin_acc_x3(img, ob, operands):
    acc = np.zeros((64, 4))
    for m, (bh, bl) in enumerate(operands):
        ah, al = img[ob, m, 0], img[ob, m, 1]
        acc += mfma_16x16x32(ah, bh) + mfma_16x16x32(ah, bl) + mfma_16x16x32(al, bh)
    return acc


@pytest.mark.parametrize("R,K,transpose", [(256, 64, False), (64, 256, False), (64, 256, True), (192, 64, False),
                                           (64, 192, True), (64, 128, False), (128, 64, True)])
def test_image_times_chain_operand_is_the_matrix_product(R, K, transpose):
    """the shapes of the training path: FeedForward W1 / W2 / W2^T / W1^T, [to_q ; to_kv] and its transpose, pw2 and pw2^T"""
    rng = np.random.default_rng(R * 1000 + K + transpose)
    W = rng.standard_normal((K, R) if transpose else (R, K))     # the stored parameter; transpose = its W^T image
    x = rng.standard_normal((16, K))
    img = pack_x3(W, transpose)
    ops = [chain_operand(x, m) for m in range(K // 32)]
    want = x @ (W if transpose else W.T)                         # [16 tokens][R]
    for ob in range(R // 16):
        acc = lin_acc_x3(img, ob, ops)
        for r in range(4):
            # accumulator: lane (c, g) register r = output feature 16 ob + 4 g + r of token c - the chain layout again,
            # so a kernel's result feeds the next product as split8 operands without any data movement
            np.testing.assert_allclose(acc[:, r], want[C, 16 * ob + 4 * G + r], rtol=0, atol=2e-5)
