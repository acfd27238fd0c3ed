"""GPU parity tests: the HIP path, called through the C ABI (ctypes), against
(a) the committed golden fixtures produced by the reference's own modules and
(b) the CPU oracle on seeded inputs, plus size-independent properties at
BASELINE.json's full sizes.  Tolerance: north_star's gate is 1e-3 relative fp32
(max|a-b| / max|b|); stage tests assert a tighter 2e-4 so drift is caught early."""
import numpy as np
import pytest
import torch

from conftest import assert_close, load_golden, rel_err
from oracle import cmgan_oracle as O
from oracle.weights import conformer_state_dict, make_state_dict, synthetic_clips

pytestmark = pytest.mark.gpu

GATE = 1e-3      # north_star: "within 1e-3 rel fp32"
STAGE = 2e-4     # what the fp32 kernels are actually held to
DEV = "cuda:0"


def _report(name, err):
    print(f"[parity] {name}: rel_err = {err:.3e}")
    return err


def _check(name, got, want, gate=GATE, rtol=1e-3, atol_rel=2e-5):
    """max-norm relative error under `gate` AND elementwise |got-want| <= atol_rel*max|want| + rtol*|want|."""
    assert _report(name, rel_err(got, want)) < gate, name
    assert_close(got, want, rtol=rtol, atol_rel=atol_rel, name=name)


@pytest.fixture(scope="module")
def sd():
    return make_state_dict(seed=0, num_features=201)


MODES = ["f16x3", "f32"]     # both matrix modes of the library are held to the same tolerances

_CPU_CACHE = {}


def _once(key, fn):
    """A CPU-side expected value (oracle / reference modules) computed ONCE per session: the tests that use the
    mode-parametrised `model` fixture run for both matrix modes against the same expectation, and the 2 s / 6 s / 10 s
    oracle passes are what the suite's wall time is made of."""
    if key not in _CPU_CACHE:
        _CPU_CACHE[key] = fn()
    return _CPU_CACHE[key]


@pytest.fixture(scope="module", params=MODES)
def model(request, sd):
    from cmgan_amd import TSCNet
    m = TSCNet(num_channel=64, num_features=201, mfma_mode=request.param).cuda().load_state_dict(sd).eval()
    print(f"[mode] TSCNet mfma_mode={m.engine.mfma_mode}")
    return m


@pytest.fixture(scope="module", params=MODES)
def conf(request):
    from cmgan_amd import ConformerBlock
    blk = ConformerBlock(dim=64, dim_head=16, heads=4, conv_kernel_size=31, attn_dropout=0.2, ff_dropout=0.2,
                         mfma_mode=request.param)
    print(f"[mode] ConformerBlock mfma_mode={blk.engine.mfma_mode}")
    return blk.load_state_dict(conformer_state_dict(seed=3)).eval()


# ------------------------------------------------------------------ conventions
def test_native_library_is_loaded_and_mfma_convention_holds(model):
    import cmgan_amd._lib as L
    assert L._lib is not None and "libcmgan_hip.so" in L.LIB_PATH
    assert model.engine.selftest_mfma() < 1e-5
    assert _report("x3 split-product MFMA self-test (abs err on O(10) dot products)", model.engine.selftest_mfma_x3()) < 2e-5


# ------------------------------------------------------------------ front / back end
def test_stft_compress_matches_reference_golden(model):
    g = load_golden("stft.npz")
    eng = model.engine
    spec = eng.stft_compress(g["wav"].to(DEV))                 # [B,2,T,F]
    want = g["compressed"].permute(0, 1, 3, 2)                 # reference layout [B,2,F,T] -> [B,2,T,F]
    assert _report("stft_compress vs golden", rel_err(spec, want)) < 1e-5


def test_uncompress_istft_matches_reference_golden(model):
    g = load_golden("stft.npz")
    comp = g["compressed"].permute(0, 1, 3, 2).contiguous().to(DEV)
    wav = model.engine.uncompress_istft(comp[:, 0:1].contiguous(), comp[:, 1:2].contiguous())
    assert _report("uncompress_istft vs golden", rel_err(wav, g["istft"])) < 1e-5


@pytest.mark.parametrize("gain", [1e-6, 30.0])
def test_uncompress_istft_is_amplitude_independent(model, gain):
    """The split-f16 inverse transform rescales each tile by a power of two: spectra whose uncompressed
    magnitudes are ~1e-20 or ~1e5 must come out as accurately as O(1) ones."""
    g = load_golden("stft.npz")
    comp = (g["compressed"].permute(0, 1, 3, 2).contiguous() * gain)
    want = O.uncompress_istft(comp[:, 0:1], comp[:, 1:2])
    got = model.engine.uncompress_istft(comp[:, 0:1].contiguous().to(DEV), comp[:, 1:2].contiguous().to(DEV))
    assert torch.isfinite(got).all()
    assert _report(f"uncompress_istft at gain {gain:g} vs oracle", rel_err(got, want)) < 2e-5


def test_power_compress_uncompress_standalone_match_golden():
    from cmgan_amd.utils import power_compress, power_uncompress
    g = load_golden("stft.npz")
    comp = power_compress(g["spec"].to(DEV))
    assert _report("power_compress", rel_err(comp, g["compressed"])) < 1e-5
    unc = power_uncompress(comp[:, 0:1].contiguous(), comp[:, 1:2].contiguous())
    assert _report("power_uncompress", rel_err(unc, g["uncompressed"])) < 1e-5


def test_rms_scale_and_scaled_stft(model):
    wav = synthetic_clips(3, 1600, seed=4)
    c = model.engine.rms_scale(wav.to(DEV))
    assert rel_err(c, O.rms_scale(wav)) < 1e-6
    spec = model.engine.stft_compress(wav.to(DEV), c)
    want = O.stft_compress(wav * O.rms_scale(wav)[:, None])
    # the dense fp32 DFT has sqrt(K)-type summation error (~4e-6 abs on O(10) bins); |X|^-0.7 compression
    # amplifies it on near-silent bins, hence 5e-5 rather than the 1e-5 an FFT reaches - gate is 1e-3
    assert _report("scaled stft_compress vs oracle", rel_err(spec, want)) < 5e-5


def test_stft_compress_tiny_amplitudes_stay_finite_and_accurate(model):
    """|X|^2 down in the fp32 denormal range: the log2/exp2 power law must not flush it to 0 or inf."""
    wav = synthetic_clips(2, 3200, seed=6) * 1e-19
    spec = model.engine.stft_compress(wav.to(DEV))
    want = O.stft_compress(wav)
    assert torch.isfinite(spec).all() and torch.count_nonzero(spec) > 0.9 * spec.numel()
    assert _report("stft_compress at 1e-19 amplitude vs oracle", rel_err(spec, want)) < 1e-3


def test_stft_istft_round_trip_full_size(model):
    """BASELINE config 2 shape: 32 x 32000 samples -> [32,2,321,201] -> back (identity)."""
    wav = synthetic_clips(32, 32000, seed=5).to(DEV)
    spec = model.engine.stft_compress(wav)
    assert spec.shape == (32, 2, 321, 201)
    back = model.engine.uncompress_istft(spec[:, 0:1].contiguous(), spec[:, 1:2].contiguous())
    assert back.shape == (32, 32000)
    assert _report("stft->istft round trip 32x32000", rel_err(back, wav)) < 2e-5


def test_silent_input_gives_zeros_not_nans(model):
    spec = model.engine.stft_compress(torch.zeros(1, 800, device=DEV))
    assert torch.count_nonzero(spec) == 0 and torch.isfinite(spec).all()


# ------------------------------------------------------------------ conformer
def test_conformer_stages_match_reference_golden(conf):
    g = load_golden("conformer.npz")
    y, taps = conf.forward_with_taps(g["x"].to(DEV))
    for i, name in enumerate(("ff1", "attn", "conv", "ff2")):
        assert _report(f"conformer.{name} vs golden", rel_err(taps[i], g[name])) < STAGE, name
    assert _report("conformer.out vs golden", rel_err(y, g["out"])) < STAGE


@pytest.mark.parametrize("n,l", [(5, 101), (3, 321), (2, 16), (2, 17), (1, 1), (7, 64), (4, 65), (3, 37), (2, 45), (1, 129),
                                 (2, 301), (2, 109)])
def test_conformer_matches_oracle_over_lengths(conf, n, l):
    """ragged block edges: L below / at / above the 16-token and 64-key tile sizes; every compile-time-tail instantiation of
    the attention (L % 64 = 1: 1, 65, 129, 321; 37: 37, 101; 45: 45, 109, 301) as a single chunk, as the last of two and of
    several, next to run-time tails (16, 17)."""
    csd = conformer_state_dict(seed=3)
    x = torch.from_numpy(np.random.Generator(np.random.PCG64(100 + l)).standard_normal((n, l, 64)).astype(np.float32))
    st = {}
    want = O.conformer_block(csd, "", x, st)
    y, taps = conf.forward_with_taps(x.to(DEV))
    for i, name in enumerate(("ff1", "attn", "conv", "ff2")):
        assert _report(f"conformer[{n}x{l}].{name}", rel_err(taps[i], st[name])) < STAGE, name
    assert _report(f"conformer[{n}x{l}].out", rel_err(y, want)) < STAGE


def test_conformer_attention_mask_matches_reference_golden(conf):
    """ConformerBlock.forward(x, mask) (conformer.py:216-217, 113-126): ragged lengths and an arbitrary pattern with
    masked query rows (uniform attention) and a chunk whose keys are all masked for some queries."""
    g = load_golden("conformer_mask.npz")
    y = conf(g["x"].to(DEV), mask=g["mask"].bool().to(DEV))
    _check("conformer(x, mask) vs golden", y, g["out"], gate=STAGE)
    # the unmasked call is untouched by the masked variant.  In F32 mode an all-True mask equals no mask bit for bit
    # (one kernel, the mask compiled in or out); in F16X3 mode the unmasked call runs the software-pipelined kernel
    # (reference level folded into the E q accumulator, other summation order of the denominator) and a masked call
    # the un-pipelined one: equal to rounding
    y0 = conf(g["x"].to(DEV))
    y1 = conf(g["x"].to(DEV), mask=torch.ones(4, 83, dtype=torch.bool, device=DEV))
    if conf.engine.mfma_mode == "f32":
        assert torch.equal(y1, y0)
    else:
        assert _report("conformer(x, all-True mask) vs conformer(x)", rel_err(y1, y0)) < 1e-6


@pytest.mark.parametrize("n,l,seed", [(3, 321, 1), (5, 101, 2), (2, 130, 3)])
def test_conformer_attention_mask_over_lengths_vs_oracle(conf, n, l, seed):
    """random masks at the path's own sequence lengths, incl. sequences whose first 64-key chunk is fully masked."""
    csd = conformer_state_dict(seed=3)
    rng = np.random.Generator(np.random.PCG64(500 + seed))
    x = torch.from_numpy(rng.standard_normal((n, l, 64)).astype(np.float32))
    mask = torch.from_numpy(rng.random((n, l)) > 0.3)
    mask[0, :min(l, 70)] = False                      # dead first chunk for the kept queries of sequence 0
    mask[0, -1] = True
    want = O.conformer_block(csd, "", x, mask=mask)
    y = conf(x.to(DEV), mask=mask.to(DEV))
    assert _report(f"conformer[{n}x{l}](x, mask)", rel_err(y, want)) < STAGE


def test_conformer_rel_pos_clamp_beyond_512(conf):
    """n = 600 > max_pos_emb: distances saturate at +-512 (conformer.py:108)."""
    g = load_golden("attention_long.npz")
    csd = conformer_state_dict(seed=3)
    st = {}
    want = O.conformer_block(csd, "", g["x"], st)
    y, taps = conf.forward_with_taps(g["x"].to(DEV))
    assert _report("conformer[n=600].attn", rel_err(taps[1], st["attn"])) < STAGE
    assert _report("conformer[n=600].out", rel_err(y, want)) < STAGE


def test_oversized_sequences_are_rejected_before_any_launch(conf):
    """The conv-module kernel addresses the rows of a sequence with 32-bit byte offsets: a call whose sequences would
    overflow them fails loudly at the C ABI (CMGAN_E_BADARG + a message), before the workspace is even looked at."""
    eng = conf.engine
    x = torch.zeros(64, device=DEV)
    rc = eng.lib.cmgan_conformer_forward(eng._h, 0, x.data_ptr(), 1, 1 << 23, x.data_ptr(), None, None, 0, None)
    assert rc < 0
    if eng.mfma_mode == "f32":          # the fp32 kernels have no such limit: the call gets as far as the (missing) workspace
        assert b"too large" not in eng.lib.cmgan_last_error(eng._h)
        rc = eng.lib.cmgan_conformer_forward(eng._h, 0, x.data_ptr(), 1 << 12, 1 << 19, x.data_ptr(), None, None, 0, None)
    assert rc == -1
    assert b"too large" in eng.lib.cmgan_last_error(eng._h)


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("gain", [40.0, 0.02])
def test_attention_softmax_rereference_branch(mode, gain):
    """The F16X3 attention keeps a stale softmax reference and only re-references when a chunk's
    scores drift by more than 2^12 from it; random weights never take that branch, so force it:
    to_q scaled x40 makes |scores| >> 12 (peaky softmax, positive and negative drifts), x0.02 keeps
    everything on the common path.  Both must match the oracle (a rare branch needs its own test)."""
    from cmgan_amd import ConformerBlock
    csd = dict(conformer_state_dict(seed=3))
    csd["attn.fn.to_q.weight"] = csd["attn.fn.to_q.weight"] * gain
    blk = ConformerBlock(dim=64, dim_head=16, heads=4, conv_kernel_size=31, mfma_mode=mode).load_state_dict(csd)
    x = torch.from_numpy(np.random.Generator(np.random.PCG64(77)).standard_normal((3, 200, 64)).astype(np.float32))
    st = {}
    want = O.conformer_block(csd, "", x, st)
    y, taps = blk.forward_with_taps(x.to(DEV))
    assert _report(f"attn re-reference gain={gain} [{mode}]", rel_err(taps[1], st["attn"])) < STAGE
    assert _report(f"conformer out gain={gain} [{mode}]", rel_err(y, want)) < STAGE


X1_GATE = 3e-3     # the opt-in single-product mode's own band (test_f16x1_mode_error_bands); never the parity gate


def test_f16x1_kernels_on_edge_shapes_mask_clamp_and_rereference():
    """The F16X1 mode is a SECOND, separately compiled copy of every conformer / attention kernel (other register
    allocation, FFN_POST_WAVES = 8): the ragged shapes (L = 1, 16, 17, 64, 65), the masked attention path, the > 512
    relative-position clamp and the forced softmax re-reference branch run on those copies too, each against the
    oracle at the mode's own gate - an indexing error shows as O(1), the mode's rounding as < 1e-3."""
    from cmgan_amd import ConformerBlock
    csd = conformer_state_dict(seed=3)
    mk = lambda sd_: ConformerBlock(dim=64, dim_head=16, heads=4, conv_kernel_size=31, mfma_mode="f16x1").load_state_dict(sd_).eval()
    blk = mk(csd)
    assert blk.engine.mfma_mode == "f16x1"
    for n, l in [(5, 101), (3, 321), (2, 16), (2, 17), (1, 1), (7, 64), (4, 65)]:
        x = torch.from_numpy(np.random.Generator(np.random.PCG64(100 + l)).standard_normal((n, l, 64)).astype(np.float32))
        st = {}
        want = O.conformer_block(csd, "", x, st)
        y, taps = blk.forward_with_taps(x.to(DEV))
        for i, name in enumerate(("ff1", "attn", "conv", "ff2")):
            assert _report(f"x1 conformer[{n}x{l}].{name}", rel_err(taps[i], st[name])) < X1_GATE, (n, l, name)
        assert _report(f"x1 conformer[{n}x{l}].out", rel_err(y, want)) < X1_GATE
    g = load_golden("conformer_mask.npz")
    assert _report("x1 conformer(x, mask) vs golden", rel_err(blk(g["x"].to(DEV), mask=g["mask"].bool().to(DEV)), g["out"])) < X1_GATE
    for n, l, seed in [(3, 321, 1), (5, 101, 2), (2, 130, 3)]:
        rng = np.random.Generator(np.random.PCG64(500 + seed))
        x = torch.from_numpy(rng.standard_normal((n, l, 64)).astype(np.float32))
        mask = torch.from_numpy(rng.random((n, l)) > 0.3)
        mask[0, :min(l, 70)] = False
        mask[0, -1] = True
        want = O.conformer_block(csd, "", x, mask=mask)
        assert _report(f"x1 conformer[{n}x{l}](x, mask)", rel_err(blk(x.to(DEV), mask=mask.to(DEV)), want)) < X1_GATE
    g = load_golden("attention_long.npz")
    st = {}
    want = O.conformer_block(csd, "", g["x"], st)
    y, taps = blk.forward_with_taps(g["x"].to(DEV))
    assert _report("x1 conformer[n=600].attn (clamp)", rel_err(taps[1], st["attn"])) < X1_GATE
    assert _report("x1 conformer[n=600].out", rel_err(y, want)) < X1_GATE
    for gain in (40.0, 0.02):
        c2 = dict(csd)
        c2["attn.fn.to_q.weight"] = c2["attn.fn.to_q.weight"] * gain
        b2 = mk(c2)
        x = torch.from_numpy(np.random.Generator(np.random.PCG64(77)).standard_normal((3, 200, 64)).astype(np.float32))
        st = {}
        want = O.conformer_block(c2, "", x, st)
        y, taps = b2.forward_with_taps(x.to(DEV))
        # a x40 score gain makes the softmax peaky: fp16 rounding of q / k moves scores by ~|s| 2^-11, so the band widens
        gate = X1_GATE if gain < 1 else 3e-2
        assert _report(f"x1 attn re-reference gain={gain}", rel_err(taps[1], st["attn"])) < gate
        assert _report(f"x1 conformer out gain={gain}", rel_err(y, want)) < gate


# ------------------------------------------------------------------ generator
def test_tscnet_stages_match_reference_golden(model):
    g = load_golden("tscnet.npz")
    real, imag, st = model.forward_with_taps(g["x"].to(DEV))
    for name in ("encoder", "tscb1", "tscb4", "mask", "complex"):
        assert _report(f"tscnet.{name} vs golden", rel_err(st[name], g[name])) < STAGE, name
    assert _report("tscnet.real vs golden", rel_err(real, g["real"])) < STAGE
    assert _report("tscnet.imag vs golden", rel_err(imag, g["imag"])) < STAGE


def test_tscnet_matches_oracle_on_a_2s_clip(model, sd):
    """full T = 321 frames (2 s @ 16 kHz), B = 2: every tile loop runs its real trip count."""
    wav = synthetic_clips(2, 32000, seed=6)
    x = O.stft_compress(wav * O.rms_scale(wav)[:, None])
    def oracle():
        st = {}
        wr, wi = O.tscnet_forward(sd, x, st)
        return st, wr, wi
    st, wr, wi = _once("tscnet 2x321", oracle)
    real, imag, got = model.forward_with_taps(x.to(DEV))
    for name in ("encoder", "tscb1", "tscb2", "tscb3", "tscb4", "mask", "complex"):
        assert _report(f"tscnet[2x321].{name}", rel_err(got[name], st[name])) < GATE, name
    _check("tscnet[2x321].real", real, wr)
    _check("tscnet[2x321].imag", imag, wi)


@pytest.mark.parametrize("mode", MODES)
def test_tscnet_48k_variant_matches_reference_golden(mode):
    from cmgan_amd import TSCNet
    g = load_golden("tscnet48.npz")
    m48 = TSCNet(64, 601, mfma_mode=mode).load_state_dict(make_state_dict(seed=5, num_features=601))
    real, imag = m48(g["x"].to(DEV))
    assert _report("tscnet48.real vs golden", rel_err(real, g["real"])) < STAGE
    assert _report("tscnet48.imag vs golden", rel_err(imag, g["imag"])) < STAGE


def test_batch_rows_are_independent_and_bit_reproducible(model):
    """BASELINE config 2 size (B = 32 x 2 s).  The forward has no cross-sample coupling, and all
    reductions run in a fixed order, so (a) two runs are bit-identical and (b) any shard of the
    batch computed alone equals the same rows of the full batch bit-for-bit - the property the
    data-parallel sharding relies on."""
    wav = synthetic_clips(32, 32000, seed=7).to(DEV)
    out = model.engine.enhance(wav)
    assert out.shape == (32, 32000) and torch.isfinite(out).all()
    again = model.engine.enhance(wav)
    assert torch.equal(out, again)
    shard = model.engine.enhance(wav[8:16].contiguous())
    assert torch.equal(shard, out[8:16])
    one = model.engine.enhance(wav[31:32].contiguous())
    assert torch.equal(one, out[31:32])


def test_hipgraph_replay_is_bit_identical_to_eager(model):
    """cmgan_enhance never allocates or synchronises, so the whole forward is capturable."""
    wav = synthetic_clips(4, 8000, seed=12).to(DEV)
    eager = model.engine.enhance(wav).clone()
    g1 = model.engine.enhance_graphed(wav).clone()
    assert torch.equal(eager, g1)
    wav2 = synthetic_clips(4, 8000, seed=13).to(DEV)
    g2 = model.engine.enhance_graphed(wav2).clone()            # replay with new data
    assert torch.equal(g2, model.engine.enhance(wav2))
    assert not torch.equal(g1, g2)


def test_branched_form_is_bit_identical(model):
    """cmgan_enhance_branched: part-batch branches on as many streams (fork / join by events), eager and as parallel
    paths of one captured hipGraph, any branch count / start offset, uneven splits: the rows are independent, so nothing
    may change."""
    eng = model.engine
    for B in (5, 2):
        wav = synthetic_clips(B, 8000, seed=40 + B).to(DEV)
        want = eng.enhance(wav).clone()
        for branches, offset in ((2, 0), (2, 7), (2, 10 ** 6), (3, 0), (4, 5), (8, 0)):
            assert torch.equal(eng.enhance_branched(wav, branches, offset), want), (B, branches, offset)
        for branches, offset in ((2, 0), (2, 30), (3, 0), (4, 2)):
            g = eng.enhance_graphed(wav, branches=branches, offset=offset).clone()
            assert torch.equal(g, want), (B, branches, offset)
            wav2 = synthetic_clips(B, 8000, seed=50 + B).to(DEV)
            g2 = eng.enhance_graphed(wav2, branches=branches, offset=offset).clone()      # replay with new data
            assert torch.equal(g2, eng.enhance(wav2))
    one = synthetic_clips(1, 8000, seed=60).to(DEV)
    assert torch.equal(eng.enhance_graphed(one, branches=2), eng.enhance(one))     # B = 1: the one-stream form
    with pytest.raises(ValueError):
        eng.enhance_graphed(one, branches=9)
    eng.set_profiling(True)
    try:
        with pytest.raises(Exception):
            eng.enhance_branched(wav)
    finally:
        eng.set_profiling(False)


# ------------------------------------------------------------------ pipeline
def test_enhance_one_track_matches_reference_golden_ragged_and_chunked(model):
    from cmgan_amd.evaluation import enhance_one_track
    g = load_golden("pipeline.npz")
    out = enhance_one_track(model, g["noisy"].to(DEV))
    assert out.shape == (2350,)
    _check("enhance_one_track vs golden", out, g["enhanced"])
    out_c = enhance_one_track(model, g["noisy"].to(DEV), cut_len=int(g["cut_len_chunked"]))
    _check("enhance_one_track (chunked) vs golden", out_c, g["enhanced_chunked"])


def test_enhance_batch_matches_oracle(model, sd):
    from cmgan_amd.evaluation import enhance_batch
    wav = synthetic_clips(2, 8000, seed=8)
    got = enhance_batch(model, wav.to(DEV))
    _check("enhance_batch vs oracle", got, O.enhance_batch(sd, wav))


@pytest.mark.parametrize("b,l", [(1, 300), (1, 800), (3, 1600), (5, 4100 - 100), (2, 6400)])
def test_enhance_batch_small_and_odd_shapes(model, sd, b, l):
    """T = L / 100 + 1 from 4 frames up: conv tiles that are mostly padding, time sequences shorter than one
    64-key attention chunk, the 64-frame STFT tile with a single live frame group, batch 1."""
    from cmgan_amd.evaluation import enhance_batch
    wav = synthetic_clips(b, l, seed=50 + b)
    got = enhance_batch(model, wav.to(DEV))
    assert got.shape == (b, l)
    assert _report(f"enhance_batch [{b} x {l}] vs oracle", rel_err(got, O.enhance_batch(sd, wav))) < GATE


def test_long_track_runs_unchunked_like_the_reference(model, sd):
    """6 s < 16 s: one row of T = 961 frames (distances beyond 512 saturate in the time conformer)."""
    from cmgan_amd.evaluation import enhance_one_track
    noisy = synthetic_clips(1, 96000, seed=9)
    got = enhance_one_track(model, noisy.to(DEV))
    assert _report("enhance 6 s track vs oracle", rel_err(got, _once("6 s track", lambda: O.enhance(sd, noisy)))) < GATE


def test_config5_ten_second_clip_in_400_frame_windows(model, sd):
    """BASELINE.json configs[4]: a 10 s clip processed as 400-frame windows.  The reference has no state
    carry; its own mechanism for windows is the reshape-to-rows rule of
    evaluation.py:30-34, so parity is defined per window: cut_len = 400 frames x hop = 40000 samples ->
    4 independent rows of 401 frames each."""
    from cmgan_amd.evaluation import enhance_one_track
    noisy = synthetic_clips(1, 160000, seed=21)
    got = enhance_one_track(model, noisy.to(DEV), cut_len=40000)
    want = _once("10 s clip as 4 rows", lambda: O.enhance(sd, noisy, cut_len=40000))
    assert got.shape == (160000,)
    assert _report("10 s clip, 4 x 400-frame windows vs oracle", rel_err(got, want)) < GATE


def test_evaluation_directory_driver_scores_like_the_oracle_path(model, sd, tmp_path):
    """src/evaluation.py:61-100 end to end: wav files in, six averaged scores out.  The same files enhanced by
    the CPU oracle and scored by the same metric code must give the same averages (the enhanced audio agrees to
    ~1e-6, the metrics are smooth in it)."""
    import numpy as np
    from scipy.io import wavfile
    from cmgan_amd import metrics
    from cmgan_amd.evaluation import evaluation
    noisy_dir, clean_dir, out_dir = tmp_path / "noisy", tmp_path / "clean", tmp_path / "enh"
    noisy_dir.mkdir(); clean_dir.mkdir()
    names = ["p1_10.wav", "p1_2.wav"]                                    # natural order: p1_2 before p1_10
    want = np.zeros(6)
    for i, name in enumerate(names):
        clean = (synthetic_clips(1, 16000 + 700 * i, seed=30 + i)[0].numpy() * 0.2)
        noisy = clean + 0.05 * synthetic_clips(1, clean.size, seed=40 + i)[0].numpy()
        wavfile.write(str(noisy_dir / name), 16000, np.round(noisy * 32767).astype(np.int16))
        wavfile.write(str(clean_dir / name), 16000, np.round(clean * 32767).astype(np.int16))
        nz = np.round(noisy * 32767).astype(np.int16).astype(np.float64) / 32768.0
        cl = np.round(clean * 32767).astype(np.int16).astype(np.float64) / 32768.0
        est = O.enhance(sd, torch.from_numpy(nz).float()[None, :]).numpy().astype(np.float64)
        want += np.array(metrics.compute_metrics(cl, est, 16000, 0, pesq_mos=2.0))
    got = evaluation(model, str(noisy_dir), str(clean_dir), True, str(out_dir),
                     pesq_fn=lambda fs, a, b: 2.0, verbose=False)
    assert sorted(p.name for p in out_dir.iterdir()) == sorted(names)
    assert _report("evaluation() averages vs oracle-enhanced", rel_err(torch.tensor(got), torch.tensor(want / 2))) < 1e-3


@pytest.mark.parametrize("graph", [False, True])
def test_windowed_inference_matches_the_per_window_contract(model, sd, graph):
    """cmgan_amd.streaming.enhance_windows: window k = samples [kW - C, (k+1)W + C) of the file-scaled signal,
    enhanced like one reference row, central W samples kept; hipGraph replay must not change a bit."""
    from cmgan_amd.streaming import enhance_windows
    W, C, L = 8000, 1600, 21700                                     # 3 windows, the last one partly zero padding
    noisy = synthetic_clips(1, L, seed=60)
    got = enhance_windows(model, noisy.to(DEV), window=W, context=C, batch=2, graph=graph)
    c = O.rms_scale(noisy)
    padded = torch.zeros(3 * W + 2 * C)
    padded[C:C + L] = noisy[0] * c
    rows = padded.unfold(0, W + 2 * C, W)
    est = O.uncompress_istft(*O.tscnet_forward(sd, O.stft_compress(rows)))
    want = (est[:, C:C + W].reshape(-1)[:L] / c)
    assert got.shape == (L,)
    assert _report(f"enhance_windows (graph={graph}) vs oracle windows", rel_err(got, want)) < GATE
    if graph:
        eager = enhance_windows(model, noisy.to(DEV), window=W, context=C, batch=2, graph=False)
        assert torch.equal(got, eager)


def test_windowed_inference_without_lookahead(model, sd):
    """lookahead = 0 (causal chunks: the rows the oracle sees end at (k+1) W, so window k uses nothing beyond its own
    end): the same per-window contract."""
    from cmgan_amd.streaming import enhance_windows
    W, C, L = 8000, 2400, 19000
    noisy = synthetic_clips(1, L, seed=61)
    got = enhance_windows(model, noisy.to(DEV), window=W, context=C, batch=2, graph=True, lookahead=0)
    c = O.rms_scale(noisy)
    padded = torch.zeros(3 * W + C)
    padded[C:C + L] = noisy[0] * c
    rows = padded.unfold(0, W + C, W)
    est = O.uncompress_istft(*O.tscnet_forward(sd, O.stft_compress(rows)))
    want = est[:, C:C + W].reshape(-1)[:L] / c
    assert _report("enhance_windows (lookahead 0) vs oracle windows", rel_err(got, want)) < GATE


# ------------------------------------------------------------------ real recordings, full-size rows, 48 kHz
@pytest.mark.parametrize("tag", ["a", "b", "silence"])
def test_real_recordings_match_reference_golden(model, tag):
    """AudioSamples/noisy tracks of the reference repo (speech + DEMAND noise, 2.1 s) and a clean track with
    0.9 s of gated digital silence, enhanced by the reference's own modules (tests/golden/make_golden.py):
    near-zero bins through |X|^-0.7, sparse InstanceNorm planes, the fp16 split ranges on real audio."""
    from cmgan_amd.evaluation import enhance_one_track
    g = load_golden("tracks.npz")
    noisy = (g[f"pcm_{tag}"].float() / 32768.0)[None, :]
    out = enhance_one_track(model, noisy.to(DEV))
    assert torch.isfinite(out).all()
    # Elementwise floor per track = what the golden itself is good for: the reference's fp32 output sits 4.8e-6 ('a') /
    # 1.3e-6 ('b') / 2.4e-5 ('silence': |X|^-0.7 on near-silent bins is ill-conditioned) of the peak away from the
    # fp64 evaluation of the same pipeline (oracle in fp64, measured in the build container).  'a' and 'b' therefore
    # use the 2e-5 of the synthetic clips; 'silence' gets 6e-5 = 2.5 x its golden's own fp32 noise.
    floor = {"a": 2e-5, "b": 2e-5, "silence": 6e-5}[tag]
    _check(f"real track '{tag}' vs reference golden", out, g[f"enhanced_{tag}"], atol_rel=floor)


def test_f16x1_mode_error_bands(sd):
    """mfma_mode="f16x1" (CMGAN_MFMA_F16X1: one fp16 product per contraction in the TSCNet body - BASELINE configs[1]'s
    "bf16"-class throughput mode; STFT / ISTFT stay on the split products) is an OPT-IN reduced-precision mode, and
    this test pins what it costs: 6e-4 .. 9e-4 of the peak on synthetic clips AND on the three real recordings (measured
    on MI355X; the 4.7e-2 of round 2's all-single-product probe on the track with digital silence came from the FRONT end,
    |X|^-0.7 on near-silent bins, which stays fp32-class here).  That is inside the 1e-3 gate but with no margin - the
    default mode sits at 4e-6 on the same inputs, asserted next to it - so the bands below are two-sided: a build
    that silently became more (or less) accurate than a single fp16 product changes them."""
    from cmgan_amd import TSCNet
    from cmgan_amd.evaluation import enhance_one_track
    m1 = TSCNet(num_channel=64, num_features=201, mfma_mode="f16x1").cuda().load_state_dict(sd).eval()
    m3 = TSCNet(num_channel=64, num_features=201, mfma_mode="f16x3").cuda().load_state_dict(sd).eval()
    assert m1.engine.mfma_mode == "f16x1"
    wav = synthetic_clips(2, 32000, seed=5)
    want = _once("f16 modes: 2 x 2 s clips", lambda: O.enhance_batch(sd, wav))
    e1 = _report("f16x1: 2 x 2 s synthetic clips vs oracle", rel_err(m1.engine.enhance(wav.to(DEV)), want))
    e3 = _report("f16x3: the same clips vs oracle", rel_err(m3.engine.enhance(wav.to(DEV)), want))
    assert e3 < 2e-5 and 20 * e3 < e1 < 3e-3
    g = load_golden("tracks.npz")
    for tag in ("a", "b", "silence"):
        noisy = (g[f"pcm_{tag}"].float() / 32768.0)[None, :]
        out = enhance_one_track(m1, noisy.to(DEV))
        assert torch.isfinite(out).all()
        e = _report(f"f16x1: real track '{tag}' vs reference golden", rel_err(out, g[f"enhanced_{tag}"]))
        assert 1e-4 < e < 3e-3, (tag, e)


def test_f16mix_mode_error_band(sd):
    """mfma_mode="f16mix" (CMGAN_MFMA_F16MIX with the shipped preset: every conformer kernel family on ONE fp16 product,
    the dense / sub-pixel convs on three): the reduced-precision mode WITH margin.  The ablation (tools/mix_ablation.py,
    profiles/r06_mix_ablation.json) shows where F16X1's 6e-4 .. 1.5e-3 comes from - the dilated dense convs alone - so
    this mode must sit at <= 2e-4 of the peak (5 x inside north_star's 1e-3 gate) on synthetic clips and on the three
    real recordings, and - two-sided - well above the default mode's error: a preset that silently fell back to three
    products (or to one everywhere) changes the band.  Also: single_mask plumbing - the empty family set IS f16x3, the
    full set IS f16x1, bit for bit."""
    from cmgan_amd import TSCNet
    from cmgan_amd.engine import F16MIX_PRESET
    from cmgan_amd.evaluation import enhance_one_track
    mk = lambda **kw: TSCNet(num_channel=64, num_features=201, **kw).cuda().load_state_dict(sd).eval()
    mm, m3, m1 = mk(mfma_mode="f16mix"), mk(mfma_mode="f16x3"), mk(mfma_mode="f16x1")
    assert mm.engine.mfma_mode == "f16mix" and set(mm.engine.mix_single) == set(F16MIX_PRESET) and "conv" not in F16MIX_PRESET
    wav = synthetic_clips(2, 32000, seed=5)
    want = _once("f16 modes: 2 x 2 s clips", lambda: O.enhance_batch(sd, wav))
    dw = wav.to(DEV)
    em = _report("f16mix: 2 x 2 s synthetic clips vs oracle", rel_err(mm.engine.enhance(dw), want))
    e3 = _report("f16x3: the same clips vs oracle", rel_err(m3.engine.enhance(dw), want))
    e1 = _report("f16x1: the same clips vs oracle", rel_err(m1.engine.enhance(dw), want))
    assert e3 < 2e-5 and 3 * e3 < em < 2e-4 and em < 0.5 * e1
    g = load_golden("tracks.npz")
    for tag in ("a", "b", "silence"):
        noisy = (g[f"pcm_{tag}"].float() / 32768.0)[None, :]
        out = enhance_one_track(mm, noisy.to(DEV))
        assert torch.isfinite(out).all()
        e = _report(f"f16mix: real track '{tag}' vs reference golden", rel_err(out, g[f"enhanced_{tag}"]))
        assert 5e-6 < e < 2e-4, (tag, e)
    none = mk(mfma_mode="f16mix", mix_single=())
    every = mk(mfma_mode="f16mix", mix_single=("conv", "ff1", "ff2", "qkv", "attn", "pw1", "dwpw2"))
    assert torch.equal(none.engine.enhance(dw), m3.engine.enhance(dw))
    assert torch.equal(every.engine.enhance(dw), m1.engine.enhance(dw))
    one = mk(mfma_mode="f16mix", mix_single=("attn",))                  # a genuinely mixed table (stage by stage from two builds)
    assert 3 * e3 < rel_err(one.engine.enhance(dw), want) < 2e-4
    with pytest.raises(ValueError):
        mk(mfma_mode="f16mix", mix_single=("nope",))
    with pytest.raises(ValueError):
        mk(mfma_mode="f16x3", mix_single=("attn",))


def test_one_row_of_the_full_config2_batch_matches_the_oracle_directly(model, sd):
    """BASELINE configs[1] shape (B = 32 x 2 s): row 17 of the batch against the CPU oracle run on that clip
    alone - a direct check at the benchmark size, not only shard == full by transitivity."""
    wav = synthetic_clips(32, 32000, seed=7)
    out = model.engine.enhance(wav.to(DEV))
    want = _once("config-2 row 17", lambda: O.enhance_batch(sd, wav[17:18]))
    _check("config-2 batch, row 17 vs oracle", out[17:18], want)


def test_full_size_clips_match_the_references_own_modules(model, sd):
    """T = 321 (2 s): the HIP path against oracle/_ref - the reference repo's OWN TSCNet / power_compress /
    power_uncompress (bytecode built from /root/reference by oracle/make_ref.py; travels to the GPU box) behind the
    src/evaluation.py:21-53 glue - not against the port.  Five rows spread over the benchmark batch plus a ragged
    track through enhance_one_track."""
    from oracle import ref_runner as R
    from cmgan_amd.evaluation import enhance_one_track
    if not R.available():
        pytest.skip("oracle/_ref not built")
    ref = R.tscnet(sd)
    wav = synthetic_clips(32, 32000, seed=7)
    out = model.engine.enhance(wav.to(DEV))
    for row in (0, 5, 11, 23, 31):                    # (row 17: the oracle test above)
        _check(f"config-2 batch, row {row} vs the reference modules", out[row:row + 1],
               _once(f"_ref row {row}", lambda: R.enhance_batch(ref, wav[row:row + 1])))
    noisy = synthetic_clips(1, 32000 + 1234, seed=9)
    _check("ragged 2.08 s track vs the reference's enhance_one_track glue",
           enhance_one_track(model, noisy.to(DEV)).flatten(), _once("_ref ragged track", lambda: R.enhance(ref, noisy)))


@pytest.mark.parametrize("mode", MODES)
def test_48k_front_and_back_end_match_the_oracle(mode):
    """n_fft = 1200 / hop = 300 (BASELINE configs[3]): the dense fp32 DFT kernels (no folded image at this size)."""
    from cmgan_amd.engine import Engine
    eng = Engine(n_fft=1200, hop=300, mfma_mode=mode)
    wav = synthetic_clips(2, 6000, seed=31)
    c = eng.rms_scale(wav.to(DEV))
    spec = eng.stft_compress(wav.to(DEV), c)
    want = O.stft_compress(wav * O.rms_scale(wav)[:, None], 1200, 300)
    assert spec.shape == (2, 2, 21, 601)
    assert _report(f"48k stft_compress [{mode}] vs oracle", rel_err(spec, want)) < 5e-5
    back = eng.uncompress_istft(spec[:, 0:1].contiguous(), spec[:, 1:2].contiguous(), c)
    assert back.shape == (2, 6000)
    assert _report(f"48k stft->istft round trip [{mode}]", rel_err(back, wav)) < 2e-5
    # torch.stft semantics for a length that is not a whole number of hops: 1 + L // hop frames
    odd = synthetic_clips(1, 2500, seed=32)
    sp_odd = eng.stft_compress(odd.to(DEV))
    assert sp_odd.shape == (1, 2, 9, 601)
    assert _report(f"48k stft_compress L=2500 [{mode}]", rel_err(sp_odd, O.stft_compress(odd, 1200, 300))) < 5e-5


@pytest.mark.parametrize("mode", MODES)
def test_48k_pipeline_matches_reference_golden(mode):
    """configs[3] wav -> wav through enhance_one_track: ragged, chunked (>cut_len rows) and the reference's
    shorter-output quirk when the 100-sample padding is not a multiple of hop = 300."""
    from cmgan_amd import TSCNet
    from cmgan_amd.evaluation import enhance_one_track
    g = load_golden("pipeline48.npz")
    m48 = TSCNet(64, 601, mfma_mode=mode).load_state_dict(make_state_dict(seed=5, num_features=601)).eval()
    assert (m48.engine.cfg.n_fft, m48.engine.cfg.hop) == (1200, 300)
    out = enhance_one_track(m48, g["noisy"].to(DEV), cut_len=48000 * 16, n_fft=1200, hop=300)
    _check(f"48k enhance_one_track [{mode}] vs golden", out, g["enhanced"])
    out_c = enhance_one_track(m48, g["noisy"].to(DEV), cut_len=int(g["cut_len_chunked"]), n_fft=1200, hop=300)
    _check(f"48k enhance_one_track chunked [{mode}] vs golden", out_c, g["enhanced_chunked"])
    out_s = enhance_one_track(m48, g["noisy_short"].to(DEV), cut_len=48000 * 16, n_fft=1200, hop=300)
    assert out_s.shape == (2400,)
    _check(f"48k enhance_one_track short output [{mode}] vs golden", out_s, g["enhanced_short"])


_ORACLE_48K = {}


@pytest.mark.parametrize("mode", MODES)
def test_48k_full_size_clip_matches_the_oracle(mode):
    """configs[3] at full size: one 2 s 48 kHz clip -> T = 321 frames x F = 601 bins (F' = 301: 19-block
    frequency sequences, 4 x the 16 kHz activation volume), both matrix modes (the oracle runs once)."""
    from cmgan_amd import TSCNet
    from cmgan_amd.evaluation import enhance_batch
    sd48 = make_state_dict(seed=5, num_features=601)
    m48 = TSCNet(64, 601, mfma_mode=mode).load_state_dict(sd48).eval()
    wav = synthetic_clips(1, 96000, seed=33)
    got = enhance_batch(m48, wav.to(DEV))
    assert got.shape == (1, 96000)
    if "want" not in _ORACLE_48K:
        _ORACLE_48K["want"] = O.enhance_batch(sd48, wav, 1200, 300)
    _check(f"48k full-size clip (T=321, F=601) [{mode}] vs oracle", got, _ORACLE_48K["want"])


# ------------------------------------------------------------------ loss terms, weight reload
def test_loss_terms_match_the_reference_formulas(model):
    """Trainer.calculate_generator_loss (train.py:124-151) without the GAN term: loss_ri, loss_mag, time_loss."""
    import torch.nn.functional as Fn
    gen = torch.Generator().manual_seed(5)
    B, T, F, L = 3, 17, 201, 1600
    er, ei = torch.randn(B, 1, T, F, generator=gen), torch.randn(B, 1, T, F, generator=gen)
    cs = torch.randn(B, 2, T, F, generator=gen)
    ea, ca = torch.randn(B, L, generator=gen), torch.randn(B, L, generator=gen)
    cr, ci = cs[:, 0:1], cs[:, 1:2]
    want = torch.stack([Fn.mse_loss(er, cr) + Fn.mse_loss(ei, ci),
                        Fn.mse_loss(torch.sqrt(er ** 2 + ei ** 2), torch.sqrt(cr ** 2 + ci ** 2)),
                        torch.mean(torch.abs(ea - ca)), Fn.mse_loss(ea, ca)])
    eng = model.engine
    got = eng.loss_terms(er.to(DEV), ei.to(DEV), cs.to(DEV), ea.to(DEV), ca.to(DEV)).cpu()
    assert_close(got, want, rtol=2e-6, atol_rel=0.0, name="loss terms")
    again = eng.loss_terms(er.to(DEV), ei.to(DEV), cs.to(DEV), ea.to(DEV), ca.to(DEV)).cpu()
    assert torch.equal(got, again)                                   # fixed-order reduction
    only_time = eng.loss_terms(est_audio=ea.to(DEV), clean_audio=ca.to(DEV)).cpu()
    assert only_time[0] == 0 and only_time[1] == 0 and torch.equal(only_time[2:], got[2:])
    with pytest.raises(ValueError):
        eng.loss_terms(est_real=er.to(DEV))


def test_reloading_weights_invalidates_captured_graphs(sd):
    """A second load_state_dict re-allocates the device weight buffers; graphs captured before it hold stale
    pointers and must be re-captured: same model object, two checkpoints, graph path == eager path."""
    from cmgan_amd import TSCNet
    from cmgan_amd.streaming import enhance_windows
    sd_b = make_state_dict(seed=9, num_features=201)
    m = TSCNet(64, 201).load_state_dict(sd).eval()
    wav = synthetic_clips(2, 4000, seed=14).to(DEV)
    long = synthetic_clips(1, 9000, seed=15).to(DEV)
    a_graph = m.engine.enhance_graphed(wav).clone()
    a_win = enhance_windows(m, long, window=4000, context=800, batch=2, graph=True).clone()
    assert torch.equal(a_graph, m.engine.enhance(wav))
    m.load_state_dict(sd_b)
    assert not m.engine._graphs and not m.engine._row_graphs
    b_eager = m.engine.enhance(wav).clone()
    b_graph = m.engine.enhance_graphed(wav).clone()
    assert torch.equal(b_graph, b_eager) and not torch.equal(b_graph, a_graph)
    b_win = enhance_windows(m, long, window=4000, context=800, batch=2, graph=True)
    assert torch.equal(b_win, enhance_windows(m, long, window=4000, context=800, batch=2, graph=False))
    assert not torch.equal(b_win, a_win)
    _check("reloaded weights vs oracle", b_eager, O.enhance_batch(sd_b, wav.cpu()))


def test_state_dict_validation_is_complete(sd):
    from cmgan_amd import TSCNet
    m = TSCNet(64, 201)
    bad = dict(sd)
    del bad["TSCB_3.freq_conformer.conv.net.5.running_var"]          # a key the old 12-key check never looked at
    with pytest.raises(KeyError):
        m.load_state_dict(bad)
    extra = dict(sd)
    extra["module.dense_encoder.conv_1.0.weight"] = sd["dense_encoder.conv_1.0.weight"]
    with pytest.raises(KeyError):
        m.load_state_dict(extra)                                      # strict (default): unexpected key
    m.load_state_dict(extra, strict=False)                            # tolerated when asked
    wrong = dict(sd)
    wrong["TSCB_1.time_conformer.ff1.fn.fn.net.0.weight"] = torch.zeros(256, 32)
    with pytest.raises(ValueError):
        m.load_state_dict(wrong)
    with pytest.raises(RuntimeError):
        m.cuda("cpu")
    assert m.to(torch.device("cuda:0")) is m and m.cuda(0) is m
    with pytest.raises(TypeError):
        m.to(torch.float16)


# ------------------------------------------------------------------ error behaviour
def test_argument_errors_surface_as_exceptions(model):
    from cmgan_amd._lib import CmganError
    with pytest.raises(ValueError):
        model(torch.zeros(1, 2, 4, 200, device=DEV))                 # wrong F
    with pytest.raises(RuntimeError):
        model(torch.zeros(1, 2, 4, 201))                             # CPU tensor: no CPU path
    with pytest.raises(CmganError):
        model.engine.enhance(torch.zeros(1, 850, device=DEV))        # fused wav -> wav needs whole hops
    with pytest.raises(CmganError):
        model.engine.stft_compress(torch.zeros(1, 100, device=DEV))  # L <= n_fft/2 (reflect pad)


def test_unloaded_model_refuses_to_run():
    from cmgan_amd import TSCNet
    m = TSCNet(64, 201)
    with pytest.raises(RuntimeError, match="no weights"):
        m(torch.zeros(1, 2, 4, 201, device=DEV))
    with pytest.raises(KeyError):
        m.load_state_dict({"dense_encoder.conv_1.0.weight": torch.zeros(64, 3, 1, 1)})


# ------------------------------------------------------------------ streaming with carried state
def _stream_setup(model, L=6000, seed=31):
    wav = synthetic_clips(1, L, seed=seed)
    spec = O.stft_compress(wav * O.rms_scale(wav)[:, None])
    return wav, spec


def test_stream_frozen_statistics_of_the_clip_itself_reproduce_the_forward_bit_for_bit(model):
    """cmgan_tscnet_forward_stats: stats_out of an unfrozen call, fed back as frozen_stats, skips every in_finalize /
    mask_stats launch and must give the SAME bits; the unfrozen call is cmgan_tscnet_forward."""
    eng = model.engine
    _, spec = _stream_setup(model)
    x = spec.to(DEV)
    r0, i0 = eng.tscnet_forward(x)
    r1, i1, blob = eng.tscnet_forward_stats(x)
    assert torch.equal(r0, r1) and torch.equal(i0, i1) and blob.numel() == eng.stats_floats(1)
    r2, i2, blob2 = eng.tscnet_forward_stats(x, frozen=blob)
    assert torch.equal(r2, r0) and torch.equal(i2, i0) and torch.equal(blob2, blob)
    # other statistics -> another result (the blob really is what normalises)
    r3, _, _ = eng.tscnet_forward_stats(x, frozen=blob * 1.01)
    assert not torch.equal(r3, r0)


def test_stream_encoder_and_decoder_state_carry_is_exact(model):
    """The N3 exactness claim: under frozen statistics frame t of the dense encoder / of the decoders depends on input
    frames t - 15 .. t only, so ANY slice with 15 frames of history in front reproduces the whole-clip frames BIT FOR
    BIT (same kernels, same per-output summation order, other tile positions) - and the whole-clip frozen pass is the
    plain forward (previous test).  14 frames of history are not enough."""
    eng = model.engine
    _, spec = _stream_setup(model)
    x = spec.to(DEV)
    real, imag, st = eng.tscnet_forward(x, taps=True)
    _, _, blob = eng.tscnet_forward_stats(x)
    whole = eng.stream_encoder(x, blob)                                        # [1,T,F',64]
    assert torch.equal(whole.permute(0, 3, 1, 2), st["encoder"])               # = the forward's own encoder output
    H = 15
    for lo, hi in ((20, 45), (15, 61), (33, 34), (40, 61)):
        part = eng.stream_encoder(x[:, :, lo - H:hi].contiguous(), blob)
        assert torch.equal(part[:, H:], whole[:, lo:hi]), (lo, hi)
    short = eng.stream_encoder(x[:, :, 20 - 14:45].contiguous(), blob)
    assert not torch.equal(short[:, 14:15], whole[:, 20:21])
    # decoders: whole-clip TSCB output in, the forward's own outputs out; slices with history equal them
    h = st["tscb4"].permute(0, 2, 3, 1).contiguous()                           # [1,T,F',64]
    wr, wi = eng.stream_decoder(h, x, blob)
    assert torch.equal(wr, real) and torch.equal(wi, imag)
    for lo, hi in ((25, 50), (15, 61), (30, 31)):
        pr, pi = eng.stream_decoder(h[:, lo - H:hi].contiguous(), x[:, :, lo - H:hi].contiguous(), blob)
        assert torch.equal(pr[:, :, H:], real[:, :, lo:hi]) and torch.equal(pi[:, :, H:], imag[:, :, lo:hi]), (lo, hi)
    # the TSCB slice = the forward's TSCBs
    xt = whole.clone()
    eng.stream_tscb(xt)
    assert torch.equal(xt.permute(0, 3, 1, 2), st["tscb4"])


@pytest.mark.parametrize("graph", [False, True])
def test_enhance_stream_matches_the_stream_oracle_and_graph_replay_equals_eager(model, sd, graph):
    """cmgan_amd.streaming.enhance_stream (carried encoder / decoder state, cached encoder outputs as attention
    context, frozen statistics calibrated on the first window) against oracle/stream_oracle.py on a 1.6 s clip in
    40-frame windows: five steps of three shapes (first, steady, last)."""
    from cmgan_amd.streaming import enhance_stream
    from oracle import stream_oracle as S
    wav = synthetic_clips(1, 16000, seed=33)
    want = S.enhance_stream(sd, wav, window=40, context=12, lookahead=8)
    got = enhance_stream(model, wav.to(DEV), window=40, context=12, lookahead=8, graph=graph)
    _check(f"enhance_stream (graph={graph}) vs stream oracle", got, want, gate=STAGE)
    if graph:
        again = enhance_stream(model, wav.to(DEV), window=40, context=12, lookahead=8, graph=True)      # cached graphs
        eager = enhance_stream(model, wav.to(DEV), window=40, context=12, lookahead=8, graph=False)
        assert torch.equal(again, got) and torch.equal(eager, got)


def test_enhance_stream_with_one_window_is_the_plain_forward(model):
    """window >= T, calibration on the whole clip: one step whose frozen statistics are the clip's own = enhance()."""
    from cmgan_amd.streaming import enhance_stream
    wav = synthetic_clips(1, 6000, seed=35).to(DEV)
    got = enhance_stream(model, wav, window=64, context=8, lookahead=8, graph=False)
    want = model.engine.enhance(wav)[0]
    assert torch.equal(got, want)


def test_streaming_enhancer_on_arbitrary_chunks_equals_enhance_stream(model):
    """Sample-level live front end (StreamingEnhancer.push / flush): incremental STFT blocks with guard frames, network
    steps as soon as a window's look-ahead is covered, samples emitted once their four frames exist - fed with ragged
    chunk sizes it returns, piece by piece, the samples enhance_stream computes from the whole signal with the same
    statistics and scale.  Equal to rounding, not to the bit: the split-f16 STFT / ISTFT scale every 64-frame TILE by the
    power of two of its own maximum, and a block of frames cut at another place puts a frame into another tile (1.5e-6)."""
    from cmgan_amd.streaming import StreamingEnhancer, enhance_stream
    eng = model.engine
    wav = synthetic_clips(1, 16000, seed=37).to(DEV)
    c = eng.rms_scale(wav)
    stats = eng.tscnet_forward_stats(eng.stft_compress(wav[:, :4800].contiguous(), c))[2]
    want = enhance_stream(model, wav, window=40, context=12, lookahead=8, stats=stats, graph=False)
    for sizes in ((1600,) * 10, (137, 3333, 50, 4000, 1, 2479, 6000), (16000,)):
        live = StreamingEnhancer(model, stats, window=40, context=12, lookahead=8, scale=c, graph=False)
        out, pos, got_before_end = [], 0, 0
        for n in sizes:
            out.append(live.push(wav[0, pos:pos + n]))
            pos += n
            got_before_end += out[-1].numel()
        assert pos == 16000
        out.append(live.flush())
        got = torch.cat(out)
        assert got.shape == want.shape
        assert _report(f"StreamingEnhancer in {len(sizes)} chunks vs enhance_stream", rel_err(got, want)) < 1e-5, sizes
        if len(sizes) > 1:
            assert got_before_end > 0                     # samples really come out while the stream is still running


def test_the_16x16x32_feed_forward_kernel_behind_CMGAN_FFN32_0_still_matches_the_goldens():
    """ffn32_x3_kernel (32x32x16 MFMAs) is the default FeedForward; ffn_x3_kernel stays in the library as the A/B partner
    (CMGAN_FFN32=0, read once per process): run the conformer / TSCNet stage goldens on it in a fresh process."""
    import os
    import subprocess
    import sys
    env = dict(os.environ, CMGAN_FFN32="0")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(root, "tests", "test_gpu_parity.py"), "-q", "-x", "-p",
                        "no:cacheprovider", "-k", "conformer_stages_match_reference_golden or tscnet_stages_match_reference_golden"],
                       env=env, cwd=root, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-1000:]
    assert "4 passed" in r.stdout, r.stdout[-500:]
