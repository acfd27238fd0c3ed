"""Deterministic speech-shaped test signals shared by tests/golden/make_metrics_golden.py (which runs the
REFERENCE's metric tool on them) and tests/test_metrics.py (which runs cmgan_amd.metrics on the same
signals), so the golden file only has to store the reference's outputs."""
import numpy as np

CASES = ((16000, 40000, 1), (16000, 23017, 2), (8000, 20000, 3))     # (fs, samples, seed)


def speechlike(n: int, fs: int, seed: int) -> np.ndarray:
    """Voiced bursts (harmonic stacks with vibrato), unvoiced noise bursts and near-silent gaps on a small
    noise floor, at int16-like amplitude (what scipy.io.wavfile hands the reference)."""
    g = np.random.Generator(np.random.PCG64(seed))
    t = np.arange(n) / fs
    x = np.zeros(n)
    pos = 0
    while pos < n:
        dur = int(fs * g.uniform(0.08, 0.35))
        kind = int(g.integers(0, 4))
        seg = slice(pos, min(n, pos + dur))
        tt = t[seg]
        if kind <= 1:                                   # voiced
            f0 = g.uniform(90, 220) * (1 + 0.03 * np.sin(2 * np.pi * 5 * tt))
            ph = 2 * np.pi * np.cumsum(f0) / fs
            tilt = g.uniform(0.8, 1.6)
            s = sum(np.sin(k * ph) / k ** tilt for k in range(1, 25)) * np.hanning(tt.size)
        elif kind == 2:                                 # unvoiced
            s = g.standard_normal(tt.size) * np.hanning(tt.size) * 0.3
        else:                                           # pause
            s = np.zeros(tt.size)
        x[seg] = s
        pos += dur
    x = x / np.max(np.abs(x)) * 9000.0
    return x + 4.0 * g.standard_normal(n)               # recording noise floor: no exactly-silent frame


def pair(fs: int, n: int, seed: int):
    clean = speechlike(n, fs, seed)
    g = np.random.Generator(np.random.PCG64(100 + seed))
    enhanced = 0.97 * clean + 45.0 * g.standard_normal(n) + 0.03 * np.roll(clean, 37)
    return clean, enhanced
