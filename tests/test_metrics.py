"""cmgan_amd.metrics against the reference tool's own outputs (tests/golden/metrics.npz,
made by tests/golden/make_metrics_golden.py from src/tools/compute_metrics.py) on shared deterministic
signals.  CPU only; float64 throughout, so the bar is round-off (1e-9), not a tolerance on the metric."""
import math
import os

import numpy as np
import pytest

from cmgan_amd import metrics as M
from metrics_signals import CASES, pair

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "metrics.npz"))
PESQ = float(G["pesq_stub"])


def close(a, b, tol=1e-9):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape
    return float(np.max(np.abs(a - b) / (1.0 + np.abs(b)))) < tol


@pytest.mark.parametrize("fs,n,seed", CASES)
def test_per_frame_measures_match_the_reference_tool(fs, n, seed):
    clean, enh = pair(fs, n, seed)
    tag = f"fs{fs}_n{n}"
    assert close(M.wss(clean, enh, fs), G[tag + "_wss"])
    assert close(M.llr(clean, enh, fs), G[tag + "_llr"])
    overall, seg = M.segmental_snr(clean, enh, fs)
    assert close(overall, G[tag + "_snr"]) and close(seg, G[tag + "_segsnr"])
    assert close(M.stoi(clean, enh, fs), G[tag + "_stoi"], 1e-8)


@pytest.mark.parametrize("fs,n,seed", CASES)
def test_compute_metrics_tuple_matches_reference_order_and_values(fs, n, seed):
    clean, enh = pair(fs, n, seed)
    got = M.compute_metrics(clean, enh, fs, 0, pesq_mos=PESQ)
    assert got._fields == ("pesq", "csig", "cbak", "covl", "ssnr", "stoi")
    assert close(np.array(got), G[f"fs{fs}_n{n}_all"], 1e-8)


def test_unequal_lengths_are_trimmed_like_the_reference():
    clean, enh = pair(*CASES[0])
    assert close(np.array(M.compute_metrics(clean, enh[:-123], 16000, 0, pesq_mos=PESQ)), G["trim_all"], 1e-8)


def test_wav_path_mode_and_missing_pesq(tmp_path):
    from scipy.io import wavfile
    clean, enh = pair(*CASES[1])
    a, b = str(tmp_path / "clean.wav"), str(tmp_path / "enh.wav")
    wavfile.write(a, 16000, clean.astype(np.int16))
    wavfile.write(b, 16000, enh.astype(np.int16))
    got = M.compute_metrics(a, b, 0, 1, pesq_mos=PESQ)
    want = M.compute_metrics(clean.astype(np.int16), enh.astype(np.int16), 16000, 0, pesq_mos=PESQ)
    assert close(np.array(got), np.array(want))
    try:
        import pesq  # noqa: F401
    except Exception:
        bare = M.compute_metrics(clean, enh, 16000, 0)
        assert math.isnan(bare.pesq) and math.isnan(bare.csig) and math.isnan(bare.covl)
        assert close([bare.ssnr, bare.stoi], [want.ssnr, want.stoi], 1e-2)     # int16 rounding only


def test_identical_signals_score_perfectly():
    clean, _ = pair(*CASES[2])
    assert M.stoi(clean, clean, 8000) > 0.999999
    assert float(np.max(np.abs(M.llr(clean, clean, 8000)))) < 1e-12
    assert float(np.max(M.wss(clean, clean, 8000))) < 1e-12
    assert np.all(M.segmental_snr(clean, clean * (1 + 1e-9), 8000)[1] == 35.0)


def test_length_mismatch_in_single_measures_raises():
    clean, enh = pair(*CASES[2])
    with pytest.raises(ValueError):
        M.wss(clean, enh[:-1], 8000)


# ------------------------------------------------------------------ the reference's OWN logged answers
KA = np.load(os.path.join(os.path.dirname(__file__), "golden", "metrics_known_answers.npz"))


@pytest.mark.parametrize("name", [str(n) for n in KA["names"]])
def test_logged_scores_of_the_reference_tool_on_its_shipped_tracks(name, tmp_path):
    """src/tools/Noisy_metrics_results/python_noisy_metrics.log lines of three `AudioSamples` noisy / clean pairs
    (fixture: tests/golden/make_metrics_known_answers.py): CSIG, CBAK, COVL, SSNR and STOI to the log's 6 decimals,
    with the logged PESQ as the (third-party) input.  The log was made on int16-scale samples (path = 1)."""
    pesq_mos, *want = [float(v) for v in KA[f"log_{name}"]]
    clean, noisy = KA[f"clean_{name}"], KA[f"noisy_{name}"]
    got = M.compute_metrics(clean, noisy, 16000, 0, pesq_mos=pesq_mos)
    assert max(abs(a - b) for a, b in zip(got[1:], want)) < 1e-6, (got, want)
    # and through the wav-file form the log was made with
    from scipy.io import wavfile
    a, b = str(tmp_path / "c.wav"), str(tmp_path / "n.wav")
    wavfile.write(a, 16000, clean)
    wavfile.write(b, 16000, noisy)
    got1 = M.compute_metrics(a, b, 0, 1, pesq_mos=pesq_mos)
    assert max(abs(a_ - b_) for a_, b_ in zip(got1[1:], want)) < 1e-6


def test_all_shipped_tracks_against_the_log_when_the_reference_tree_is_here():
    ref = "/root/reference/AudioSamples"
    if not os.path.isdir(ref):
        pytest.skip("no /root/reference here")
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
    from make_metrics_known_answers import logged
    from scipy.io import wavfile
    log, worst = logged(), 0.0
    names = sorted(os.path.splitext(f)[0] for f in os.listdir(os.path.join(ref, "noisy")))
    assert len(names) == 25
    for n in names:
        got = M.compute_metrics(os.path.join(ref, "clean", n + ".wav"), os.path.join(ref, "noisy", n + ".wav"), 0, 1,
                                pesq_mos=log[n][0])
        worst = max(worst, max(abs(a - b) for a, b in zip(got[1:], log[n][1:])))
    assert worst < 1e-6, worst
