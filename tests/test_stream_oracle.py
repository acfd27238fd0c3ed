"""The carried-state streaming contract on the CPU oracle (oracle/stream_oracle.py): the properties the GPU path is
then held to bit for bit (tests/test_gpu_parity.py::test_stream_*).  CPU only."""
import numpy as np
import pytest
import torch

from conftest import rel_err
from oracle import cmgan_oracle as O
from oracle import stream_oracle as S
from oracle.weights import make_state_dict, synthetic_clips


@pytest.fixture(scope="module")
def sd():
    return make_state_dict(seed=0, num_features=201)


@pytest.fixture(scope="module")
def spec():
    wav = synthetic_clips(1, 6000, seed=31)                  # T = 61 frames
    return O.stft_compress(wav * O.rms_scale(wav)[:, None])


def test_replaying_a_clips_own_statistics_is_the_reference_forward(sd, spec):
    stats = S.calibrate(sd, spec)
    assert len(stats) == 16                                  # encoder 2 + 4, each decoder 4 + 1
    want = O.tscnet_forward(sd, spec)
    with O.norm_stats("replay", stats):
        got = O.tscnet_forward(sd, spec)
    assert rel_err(got[0], want[0]) < 1e-6 and rel_err(got[1], want[1]) < 1e-6


def test_with_frozen_statistics_the_encoder_and_the_decoders_are_causal_with_15_frames_of_memory(sd, spec):
    """Frame t of their outputs depends on input frames t - 15 .. t only: any slice with 15 frames of history in
    front reproduces the whole-clip frames (fp32 summation noise of torch's conv kernels at different sizes aside);
    with 14 frames it does not."""
    stats = S.calibrate(sd, spec)
    whole = S.encoder_frozen(sd, spec, stats)
    for lo, hi in ((20, 45), (15, 61), (33, 34)):
        part = S.encoder_frozen(sd, spec[:, :, lo - S.HIST:hi], stats)[:, :, S.HIST:]
        assert rel_err(part, whole[:, :, lo:hi]) < 2e-6, (lo, hi)
    short = S.encoder_frozen(sd, spec[:, :, 20 - 14:45], stats)[:, :, 14:]
    assert rel_err(short[:, :, :1], whole[:, :, 20:21]) > 1e-4
    h = torch.from_numpy(np.random.Generator(np.random.PCG64(3)).standard_normal((1, 64, 61, 101)).astype(np.float32))
    wr, wi = S.decoders_frozen(sd, h, spec, stats)
    pr, pi = S.decoders_frozen(sd, h[:, :, 25 - S.HIST:50], spec[:, :, 25 - S.HIST:50], stats)
    assert rel_err(pr[:, :, S.HIST:], wr[:, :, 25:50]) < 2e-6 and rel_err(pi[:, :, S.HIST:], wi[:, :, 25:50]) < 2e-6
    # and the unfrozen modules are NOT causal: the same slice under its own statistics differs
    own = O.dense_encoder(sd, torch.cat([torch.sqrt(spec[:, 0:1] ** 2 + spec[:, 1:2] ** 2), spec], 1)[:, :, 5:45])
    assert rel_err(own[:, :, 15:], whole[:, :, 20:45]) > 1e-3


def test_one_window_covering_the_clip_is_the_reference_forward(sd, spec):
    """window >= T: one step, calibration on the whole clip = TSCNet.forward itself."""
    stats = S.calibrate(sd, spec)
    got = S.stream_forward(sd, spec, stats, window=64, context=8, lookahead=8)
    want = O.tscnet_forward(sd, spec)
    assert rel_err(got[0], want[0]) < 1e-6 and rel_err(got[1], want[1]) < 1e-6


def test_stream_steps_equal_the_whole_clip_pass_where_state_is_exact_and_stay_close_elsewhere(sd, spec):
    """With context and look-ahead covering the whole clip the TSCBs see every frame in every step, so the stream equals
    the whole-clip pass under the same statistics; with short context it is the windowed-attention approximation."""
    stats = S.calibrate(sd, spec)
    with O.norm_stats("replay", stats):
        want = O.tscnet_forward(sd, spec)
    got = S.stream_forward(sd, spec, stats, window=16, context=64, lookahead=64)
    assert rel_err(got[0], want[0]) < 5e-6 and rel_err(got[1], want[1]) < 5e-6
    approx = S.stream_forward(sd, spec, stats, window=16, context=8, lookahead=4)
    assert 1e-4 < rel_err(approx[0], want[0]) < 0.5
    assert torch.isfinite(approx[0]).all() and approx[0].shape == want[0].shape


def test_config5_goldens_are_consistent_with_the_published_costs():
    """tests/golden/stream10s.npz (reference whole-clip output + stream-oracle outputs at the named configs[4] shape) and
    tests/golden/stream_cost_oracle.json (the approximation-cost bands of tests/test_gpu_stream_config5.py) come from two
    scripts: the costs recomputed from the arrays must be the published ones."""
    import json
    import os
    import numpy as np
    from conftest import GOLDEN, load_golden
    from cmgan_amd import metrics as M
    g = load_golden("stream10s.npz")
    costs = json.load(open(os.path.join(GOLDEN, "stream_cost_oracle.json")))
    ref = g["whole"].double().numpy()
    assert ref.shape == (160000,) and int(g["seed"]) == 3
    for ca, la in ((40, 40), (40, 0)):
        a = g[f"stream_{ca}_{la}"].double().numpy()
        d = a - ref
        want = costs[f"synthetic10s_w400_c{ca}_l{la}"]
        assert abs(np.abs(d).max() / np.abs(ref).max() - want["rel_max"]) < 1e-4          # (whole: _ref here, the port there: 1e-6 apart)
        assert abs(np.sqrt((d * d).mean() / (ref * ref).mean()) - want["rel_rms"]) < 1e-4
        assert abs(float(np.mean(M.segmental_snr(ref, a, 16000)[1])) - want["ssnr"]) < 0.02
        assert abs(M.stoi(ref, a, 16000) - want["stoi"]) < 1e-5
    # a one-window stream IS the whole clip: the three 2 s recordings at window 400 cost nothing
    for name in ("a", "b", "silence"):
        assert costs[f"track_{name}_w400_c40_l40"]["rel_max"] < 2e-6
