"""Lane-level numpy model of stft_fft400_kernel / istft_fft400_kernel (cmgan_amd/csrc/stft.hip): the 400-point real DFT as a
16 x 25 factorisation - pass 1 = 25-point DFTs (5 x 5) of x[16 m + r] per residue r, twiddle W400^{r k'}, pass 2 = 16-point DFTs
(4 x 4) per channel k' = 0..12, bins above N/2 taken as conjugates of the missing channels.  The model uses the kernel's own
butterflies, index maps and table layouts (tools/gen_fft_tables.py) and must reproduce numpy's rfft / irfft; it also checks
that every one of the 201 bins is produced exactly once."""
import numpy as np

N = 400
W = lambda n, e: np.exp(-2j * np.pi * (e % n) / n)
C1, C2, S1, S2 = np.cos(2 * np.pi / 5), np.cos(4 * np.pi / 5), np.sin(2 * np.pi / 5), np.sin(4 * np.pi / 5)


def dft5(x, inv=False):
    rot = (lambda z: 1j * z) if inv else (lambda z: -1j * z)
    t1, t2, t3, t4 = x[1] + x[4], x[2] + x[3], x[1] - x[4], x[2] - x[3]
    a1, a2 = x[0] + C1 * t1 + C2 * t2, x[0] + C2 * t1 + C1 * t2
    r1, r2 = rot(S1 * t3 + S2 * t4), rot(S2 * t3 - S1 * t4)
    return np.array([x[0] + t1 + t2, a1 + r1, a2 + r2, a2 - r2, a1 - r1])


def dft4(x, inv=False):
    rot = (lambda z: 1j * z) if inv else (lambda z: -1j * z)
    t0, t1, t2, t3 = x[0] + x[2], x[0] - x[2], x[1] + x[3], x[1] - x[3]
    return np.array([t0 + t2, t1 + rot(t3), t0 - t2, t1 - rot(t3)])


def forward(x):
    H = np.zeros((16, 13), complex)
    for r in range(16):                                       # pass 1: lane r
        v = x[16 * np.arange(25) + r].astype(complex)
        C = np.array([dft5(v[5 * np.arange(5) + b]) for b in range(5)])          # C[b][c]
        for c in range(5):
            y = dft5(np.array([C[b][c] * W(25, b * c) for b in range(5)]))       # y[d] = G[c + 5 d]
            for d in range(3):
                if c + 5 * d < 13:
                    H[r][c + 5 * d] = y[d] * W(400, r * (c + 5 * d))
    X, cnt = np.zeros(201, complex), np.zeros(201, int)
    for kp in range(13):                                      # pass 2: lane k'
        inner = np.array([dft4(H[4 * np.arange(4) + q, kp]) for q in range(4)])  # inner[q][s]
        for s in range(4):
            y = dft4(np.array([inner[q][s] * W(16, q * s) for q in range(4)]))   # y[u] = Y[s + 4 u]
            for u in range(4):
                k = kp + 25 * (s + 4 * u)
                if k <= 200:
                    X[k] = y[u]; cnt[k] += 1
                elif kp != 0:
                    X[400 - k] = np.conj(y[u]); cnt[400 - k] += 1
    return X, cnt


def inverse(Y):
    """Y[0..200] -> x[0..399] = irfft(Y, 400) by the mirrored passes (imaginary parts of DC / Nyquist ignored)."""
    Y = Y.copy()
    Y[0], Y[200] = Y[0].real, Y[200].real
    Hh = np.zeros((16, 13), complex)
    for kp in range(13):                                      # pass A: lane k': inverse 16-point DFT over k1
        ch = np.zeros(16, complex)
        for k1 in range(16):
            k = kp + 25 * k1
            ch[k1] = Y[k] if k <= 200 else np.conj(Y[400 - k])
        inner = np.array([dft4(ch[4 * np.arange(4) + q], inv=True) for q in range(4)])          # over p: k1 = 4 p + q
        for s in range(4):
            y = dft4(np.array([inner[q][s] * np.conj(W(16, q * s)) for q in range(4)]), inv=True)   # y[u] = h[s + 4 u]
            for u in range(4):
                r = s + 4 * u
                Hh[r][kp] = y[u] * np.conj(W(400, r * kp))
    x = np.zeros(400)
    for r in range(16):                                       # pass B: lane r: inverse 25-point DFT, Hermitian input
        full = np.zeros(25, complex)
        full[:13] = Hh[r]
        full[13:] = np.conj(Hh[r][12:0:-1])
        C = np.array([dft5(full[5 * np.arange(5) + b], inv=True) for b in range(5)])           # k' = 5 a + b
        for c in range(5):
            y = dft5(np.array([C[b][c] * np.conj(W(25, b * c)) for b in range(5)]), inv=True)  # y[d] -> m = c + 5 d
            for d in range(5):
                x[16 * (c + 5 * d) + r] = y[d].real / N
    return x


def test_forward_passes_reproduce_rfft_and_cover_every_bin_once():
    rng = np.random.default_rng(0)
    for _ in range(3):
        x = rng.standard_normal(N)
        X, cnt = forward(x)
        assert (cnt == 1).all()
        ref = np.fft.rfft(x)
        assert np.abs(X - ref).max() < 1e-12 * np.abs(ref).max()


def test_inverse_passes_reproduce_irfft():
    rng = np.random.default_rng(1)
    Y = rng.standard_normal(201) + 1j * rng.standard_normal(201)
    x = inverse(Y)
    ref = np.fft.irfft(Y, N)                                   # numpy also ignores the imaginary parts of DC / Nyquist
    assert np.abs(x - ref).max() < 1e-12 * np.abs(ref).max()
    assert np.abs(inverse(np.fft.rfft(ref)) - ref).max() < 1e-12


def test_generated_tables_match_the_model():
    import os
    import re
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "cmgan_amd", "csrc", "stft_fft_tables.h")
    text = open(path).read()
    for name, want in (("fft_tw400", [W(400, r * k) for r in range(16) for k in range(13)]),
                       ("fft_tw25", [W(25, b * c) for b in range(5) for c in range(5)]),
                       ("fft_tw16", [W(16, q * s) for q in range(4) for s in range(4)])):
        body = re.search(name + r"\[\d+\] = \{(.*?)\};", text, re.S).group(1)
        vals = np.array([float(v.rstrip("f")) for v in re.findall(r"-?[\d.]+(?:e-?\d+)?f", body)]).reshape(-1, 2)
        got = vals[:, 0] + 1j * vals[:, 1]
        assert got.shape == (len(want),) and np.abs(got - np.array(want)).max() < 1e-7
