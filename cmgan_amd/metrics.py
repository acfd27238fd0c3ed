"""Objective speech-quality metrics of the reference's evaluation harness, host side.

Replaces `src/tools/compute_metrics.py` (the Python transcription of Loizou's `compute_metrics.m` the
reference scores with, `src/evaluation.py:86-87`): weighted spectral slope (WSS, :80-274), log-likelihood
ratio (LLR, :277-347), segmental SNR (:350-397), STOI (:400-596) and the composite CSIG / CBAK / COVL
(:62-72).  Same definitions, frame bookkeeping and constants (including the tool's quirks: the peak
search's off-by-one, STOI's one-sample-early silence test), but written frame-parallel: every measure
works on a [frames, samples] view of the signals where the reference loops over frames in Python.
Agreement with the reference tool on shared signals is 1e-9 (tests/test_metrics.py).

PESQ itself is ITU-T P.862 code the reference takes from the `pesq` wheel (:7, :59); it is imported when
present and otherwise has to be supplied by the caller (`pesq_mos=`) - the composite scores depend on it.
Everything is float64 numpy on the CPU; this module never touches the GPU library.
"""
from __future__ import annotations

import math
from typing import Callable, NamedTuple, Optional

import numpy as np
from scipy import signal as _sig

__all__ = ["wss", "llr", "segmental_snr", "stoi", "compute_metrics", "Scores"]

_EPS = float(np.spacing(1.0))

# critical-band centre frequencies / bandwidths (Hz) of the WSS filterbank   compute_metrics.py:102-158
_CB_CENTRE = np.array([50.0, 120.0, 190.0, 260.0, 330.0, 400.0, 470.0, 540.0, 617.372, 703.378, 798.717,
                       904.128, 1020.38, 1148.30, 1288.72, 1442.54, 1610.70, 1794.16, 1993.93, 2211.08,
                       2446.71, 2701.97, 2978.04, 3276.17, 3597.63])
_CB_WIDTH = np.array([70.0, 70.0, 70.0, 70.0, 70.0, 70.0, 70.0, 77.3724, 86.0056, 95.3398, 105.411, 116.256,
                      127.914, 140.423, 153.823, 168.154, 183.457, 199.776, 217.153, 235.631, 255.255,
                      276.072, 298.126, 321.465, 346.136])


class Scores(NamedTuple):
    """Order of the reference's return tuple (compute_metrics.py:77)."""
    pesq: float
    csig: float
    cbak: float
    covl: float
    ssnr: float
    stoi: float


def _pair(clean, processed):
    a = np.asarray(clean, dtype=np.float64).reshape(-1)
    b = np.asarray(processed, dtype=np.float64).reshape(-1)
    if a.size != b.size:
        raise ValueError("clean and processed signals must have the same length")
    return a, b


def _analysis_frames(x: np.ndarray, win: int, hop: int, count: int) -> np.ndarray:
    """[count, win] view of x (frame f starts at f * hop), times the reference's 'Hanning' taper
    0.5 (1 - cos(2 pi n / (win + 1))), n = 1..win   (compute_metrics.py:183-186, :303-305, :375-377)."""
    taper = 0.5 * (1.0 - np.cos(2.0 * math.pi * np.arange(1, win + 1) / (win + 1)))
    view = np.lib.stride_tricks.sliding_window_view(x, win)[::hop][:count]
    return view * taper


def _win_hop(fs: int):
    win = int(np.round(30 * fs / 1000))          # 30 ms
    return win, win // 4


# ------------------------------------------------------------------------------------------ WSS
def _nearest_peak(energy: np.ndarray, slope: np.ndarray) -> np.ndarray:
    """For every band b < nb-1 the energy of the spectral peak its slope points to: walk right while the
    slope stays positive, otherwise left while it stays non-positive   (compute_metrics.py:216-240).
    energy [F, nb], slope [F, nb-1]  ->  [F, nb-1]."""
    F, ns = slope.shape
    rising = slope > 0
    # right walk: index of the first band >= b whose slope is not positive (ns when none), minus one
    stop_r = np.full((F, ns + 1), ns, dtype=np.int64)
    for b in range(ns - 1, -1, -1):
        stop_r[:, b] = np.where(rising[:, b], stop_r[:, b + 1], b)
    # left walk: index of the first band <= b whose slope is positive (-1 when none), plus one
    stop_l = np.full((F, ns + 1), -1, dtype=np.int64)           # column b + 1 holds the answer for band b
    for b in range(ns):
        stop_l[:, b + 1] = np.where(rising[:, b], b, stop_l[:, b])
    rows = np.arange(F)[:, None]
    right = energy[rows, stop_r[:, :ns] - 1]
    left = energy[rows, stop_l[:, 1:] + 1]
    return np.where(rising, right, left)


def wss(clean, processed, fs: int) -> np.ndarray:
    """Per-frame weighted spectral slope distance (Klatt)   compute_metrics.py:80-274."""
    x, y = _pair(clean, processed)
    win, hop = _win_hop(fs)
    nframes = int(x.size / hop - win / hop)
    nfft = int(2 ** math.ceil(math.log2(2 * win)))
    half = nfft // 2
    nb = _CB_CENTRE.size
    # Gaussian critical-band filters, equal area, cut at -30 dB
    bins = np.arange(half)
    f0 = np.floor(_CB_CENTRE / (fs / 2) * half)[:, None]
    bw = (_CB_WIDTH / (fs / 2) * half)[:, None]
    gain = (np.log(_CB_WIDTH[0]) - np.log(_CB_WIDTH))[:, None]
    bank = np.exp(-11.0 * ((bins[None, :] - f0) / bw) ** 2 + gain)
    bank[bank <= math.exp(-30.0 / (2.0 * 2.303))] = 0.0

    def band_db(sig):
        frames = _analysis_frames(sig, win, hop, nframes) / 32768.0
        power = np.abs(np.fft.fft(frames, nfft, axis=1)[:, :half]) ** 2
        return 10.0 * np.log10(np.maximum(power @ bank.T, 1e-10))

    ex, ey = band_db(x), band_db(y)
    sx, sy = np.diff(ex, axis=1), np.diff(ey, axis=1)
    kmax, kloc = 20.0, 1.0

    def weights(e, s):
        peak = _nearest_peak(e, s)
        e1 = e[:, : nb - 1]
        return (kmax / (kmax + e.max(axis=1, keepdims=True) - e1)) * (kloc / (kloc + peak - e1))

    w = 0.5 * (weights(ex, sx) + weights(ey, sy))
    return np.sum(w * (sx - sy) ** 2, axis=1) / np.sum(w, axis=1)


# ------------------------------------------------------------------------------------------ LLR
def _lpc(frames: np.ndarray, order: int):
    """Autocorrelation lags R [F, order+1] and LPC polynomial [1, -a_1 .. -a_p] [F, order+1] by the
    Levinson-Durbin recursion, all frames at once   (compute_metrics.py:321-347)."""
    F, n = frames.shape
    R = np.stack([np.einsum("fi,fi->f", frames[:, : n - k], frames[:, k:]) for k in range(order + 1)], axis=1)
    a = np.zeros((F, order))
    err = R[:, 0].copy()
    for i in range(order):
        acc = np.einsum("fj,fj->f", a[:, :i], R[:, i:0:-1]) if i else 0.0
        k = (R[:, i + 1] - acc) / err
        if i:
            a[:, :i] = a[:, :i] - k[:, None] * a[:, i - 1 :: -1][:, :i]
        a[:, i] = k
        err = (1.0 - k * k) * err
    return R, np.concatenate([np.ones((F, 1)), -a], axis=1)


def llr(clean, processed, fs: int) -> np.ndarray:
    """Per-frame log-likelihood ratio of the LPC models   compute_metrics.py:277-318."""
    x, y = _pair(clean, processed)
    win, hop = _win_hop(fs)
    nframes = int((x.size - win) / hop)
    order = 10 if fs < 10000 else 16
    Rx, Ax = _lpc(_analysis_frames(x, win, hop, nframes), order)
    _, Ay = _lpc(_analysis_frames(y, win, hop, nframes), order)
    lag = np.abs(np.arange(order + 1)[:, None] - np.arange(order + 1)[None, :])
    T = Rx[:, lag]                                               # [F, p+1, p+1] Toeplitz of the clean lags
    num = np.einsum("fi,fij,fj->f", Ay, T, Ay)
    den = np.einsum("fi,fij,fj->f", Ax, T, Ax)
    return np.log(num / den)


# ------------------------------------------------------------------------------------------ SNR
def segmental_snr(clean, processed, fs: int):
    """(overall SNR dB, per-frame segmental SNR clipped to [-10, 35] dB)   compute_metrics.py:350-397."""
    x, y = _pair(clean, processed)
    win, hop = _win_hop(fs)
    nframes = int(x.size / hop - win / hop)
    overall = 10.0 * np.log10(np.sum(x * x) / np.sum((x - y) ** 2))
    fx, fy = _analysis_frames(x, win, hop, nframes), _analysis_frames(y, win, hop, nframes)
    seg = 10.0 * np.log10(np.sum(fx * fx, axis=1) / (np.sum((fx - fy) ** 2, axis=1) + _EPS) + _EPS)
    return overall, np.clip(seg, -10.0, 35.0)


# ------------------------------------------------------------------------------------------ STOI
def _third_octave_bands(fs: int, nfft: int, nbands: int, first_centre: float) -> np.ndarray:
    """0/1 matrix [bands, nfft/2+1] of the one-third-octave bands   compute_metrics.py:474-524."""
    f = np.linspace(0, fs, nfft + 1)[: nfft // 2 + 1]
    k = np.arange(nbands)
    lo = first_centre * 2.0 ** ((2 * k - 1) / 6.0)               # geometric means of neighbouring centres
    hi = first_centre * 2.0 ** ((2 * k + 1) / 6.0)
    A = np.zeros((nbands, f.size))
    for i in range(nbands):
        a, b = int(np.argmin((f - lo[i]) ** 2)), int(np.argmin((f - hi[i]) ** 2))
        A[i, a:b] = 1.0
    width = A.sum(axis=1)
    keep = [i for i in range(nbands - 1) if width[i + 1] >= width[i] and width[i + 1] != 0]
    return A[: keep[-1] + 2]


def _hann_inner(n: int) -> np.ndarray:
    return _sig.windows.hann(n + 2)[1 : n + 1]


def _drop_silent_frames(x: np.ndarray, y: np.ndarray, dyn_range: float, n: int, hop: int):
    """Overlap-add reconstruction of x and y from the frames whose clean energy is within dyn_range dB of
    the loudest one.  The energy test reads every frame ONE SAMPLE EARLY (index -1 wraps to the last
    sample for the first frame), exactly as the reference does   (compute_metrics.py:551-585)."""
    starts = np.arange(0, x.size - n, hop)
    w = _hann_inner(n)
    idx = (starts[:, None] - 1 + np.arange(n)[None, :]) % x.size
    level = 20.0 * np.log10(np.linalg.norm(x[idx] * w, axis=1) / math.sqrt(n))
    keep = np.flatnonzero((level - level.max() + dyn_range) > 0)
    xs, ys = np.zeros(x.size), np.zeros(y.size)
    grab = starts[keep][:, None] + np.arange(n)[None, :]
    put = starts[: keep.size][:, None] + np.arange(n)[None, :]
    np.add.at(xs, put, x[grab] * w)
    np.add.at(ys, put, y[grab] * w)
    end = int(put[-1, -1]) + 1
    return xs[:end], ys[:end]


def _band_envelopes(x: np.ndarray, n: int, nfft: int, bands: np.ndarray) -> np.ndarray:
    """sqrt of the band energies of the Hann-windowed short-time DFT (hop n/2), [bands, frames]; scaled
    like scipy.signal.stft (division by the window sum), which the reference calls   (:527-548, :435-440)."""
    hop = n // 2
    count = int((x.size - n) / hop)
    w = _hann_inner(n)
    frames = np.lib.stride_tricks.sliding_window_view(x, n)[::hop][:count] * w
    spec = np.fft.rfft(frames, nfft, axis=1) / w.sum()
    return np.sqrt(bands @ (np.abs(spec.T) ** 2))


def stoi(clean, processed, fs: int) -> float:
    """Short-time objective intelligibility (Taal et al. 2011)   compute_metrics.py:400-471."""
    x, y = _pair(clean, processed)
    fs_i, n, nfft, nb, seg, beta, dyn = 10000, 256, 512, 15, 30, -15.0, 40.0
    bands = _third_octave_bands(fs_i, nfft, nb, 150.0)
    if fs != fs_i:
        x, y = _sig.resample_poly(x, fs_i, fs), _sig.resample_poly(y, fs_i, fs)
    x, y = _drop_silent_frames(x, y, dyn, n, n // 2)
    X, Y = _band_envelopes(x, n, nfft, bands), _band_envelopes(y, n, nfft, bands)
    clip = 10.0 ** (-beta / 20.0)
    # every run of `seg` consecutive frames, all bands at once: [bands, windows, seg]
    Xs = np.lib.stride_tricks.sliding_window_view(X, seg, axis=1)
    Ys = np.lib.stride_tricks.sliding_window_view(Y, seg, axis=1)
    alpha = np.sqrt(np.sum(Xs * Xs, axis=2, keepdims=True) / np.sum(Ys * Ys, axis=2, keepdims=True))
    Yc = np.minimum(Ys * alpha, Xs * (1.0 + clip))
    xn = Xs - Xs.mean(axis=2, keepdims=True)
    yn = Yc - Yc.mean(axis=2, keepdims=True)
    xn = xn / np.linalg.norm(xn, axis=2, keepdims=True)
    yn = yn / np.linalg.norm(yn, axis=2, keepdims=True)
    corr = np.sum(xn * yn, axis=(0, 2)) / nb                     # the reference divides by J = 15 (:468)
    return float(corr.mean())


# ------------------------------------------------------------------------------------------ composite
def _default_pesq() -> Optional[Callable]:
    try:
        from pesq import pesq as _pesq                           # the wheel the reference uses
    except Exception:
        return None
    return lambda fs, ref, deg: float(_pesq(fs, ref, deg, "wb"))


def have_pesq() -> bool:
    """True when the `pesq` wheel the reference scores with (compute_metrics.py:7) can be imported."""
    return _default_pesq() is not None


def compute_metrics(clean, enhanced, fs: int, path: int = 0, *, pesq_mos: Optional[float] = None) -> Scores:
    """`pesq, csig, cbak, covl, ssnr, stoi = compute_metrics(clean, enhanced, Fs, path)` with the reference's
    argument meaning (compute_metrics.py:26-77): `path=1` reads two .wav files, `path=0` takes arrays.
    PESQ comes from `pesq_mos` when given, else from the `pesq` package; without either the PESQ-dependent
    entries are NaN (SSNR and STOI are still exact)."""
    if path == 1:
        from scipy.io import wavfile
        fs1, a = wavfile.read(clean)
        fs2, b = wavfile.read(enhanced)
        if fs1 != fs2:
            raise ValueError("The two files do not match!")
        fs = fs1
    else:
        a, b = clean, enhanced
    a = np.asarray(a, dtype=np.float64).reshape(-1)
    b = np.asarray(b, dtype=np.float64).reshape(-1)
    if a.size != b.size:                                         # the reference trims and adds eps (:41-44)
        n = min(a.size, b.size)
        a, b = a[:n] + _EPS, b[:n] + _EPS
    trim = 0.95

    def trimmed_mean(v):
        v = np.sort(v)
        return float(np.mean(v[: round(v.size * trim)]))

    wss_d = trimmed_mean(wss(a, b, fs))
    llr_d = trimmed_mean(llr(a, b, fs))
    _, seg = segmental_snr(a, b, fs)
    ssnr = float(np.mean(seg))
    if pesq_mos is None:
        fn = _default_pesq()
        pesq_mos = fn(fs, a, b) if fn is not None else float("nan")
    lim = lambda v: v if math.isnan(v) else min(5.0, max(1.0, v))
    csig = lim(3.093 - 1.029 * llr_d + 0.603 * pesq_mos - 0.009 * wss_d)
    cbak = lim(1.634 + 0.478 * pesq_mos - 0.007 * wss_d + 0.063 * ssnr)
    covl = lim(1.594 + 0.805 * pesq_mos - 0.512 * llr_d - 0.007 * wss_d)
    return Scores(float(pesq_mos), csig, cbak, covl, ssnr, stoi(a, b, fs))
