"""GPU tests of the training-step slice: the train-mode FeedForward branch and its backward on the
HIP kernels against (a) the reference module's own autograd (tests/golden/ffn_train.npz, made by make_golden.py)
and (b) autograd through the CPU oracle at a larger, ragged token count; plus the weighted generator loss."""
import numpy as np
import pytest
import torch

from conftest import assert_close, load_golden, rel_err
from oracle import cmgan_oracle as O
from oracle.weights import conformer_state_dict

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
GRAD_TOL = 1e-4          # verdict r1: "dL/dx and dL/dW of one FeedForward against torch autograd at 1e-3"
KEYS = ("fn.norm.weight", "fn.norm.bias", "fn.fn.net.0.weight", "fn.fn.net.0.bias", "fn.fn.net.3.weight",
        "fn.fn.net.3.bias")


def _report(name, err):
    print(f"[parity] {name}: rel_err = {err:.3e}")
    return err


@pytest.fixture(scope="module")
def ff():
    from cmgan_amd.training import FeedForwardTrain
    csd = conformer_state_dict(seed=3)
    return FeedForwardTrain({k: csd["ff1." + k] for k in KEYS}, dropout=0.2)


@pytest.mark.parametrize("masked", [True, False])
def test_feed_forward_train_matches_reference_autograd(ff, masked):
    g = load_golden("ffn_train.npz")
    m1 = g["mask1"].to(DEV) if masked else None
    m2 = g["mask2"].to(DEV) if masked else None
    y = ff.forward(g["x"].to(DEV), m1, m2)
    assert _report(f"ffn train forward (masked={masked})", rel_err(y, g["y" if masked else "y_nomask"])) < GRAD_TOL
    dx, grads = ff.backward(g["x"].to(DEV), g["dy"].to(DEV), m1, m2)
    assert _report(f"ffn train dL/dx (masked={masked})", rel_err(dx, g["dx" if masked else "dx_nomask"])) < GRAD_TOL
    pre = "" if masked else "nomask_"
    for k in KEYS:
        want = g[pre + k.replace(".", "_")]
        assert _report(f"ffn train dL/d[{k}] (masked={masked})", rel_err(grads[k], want)) < GRAD_TOL, k
        assert_close(grads[k], want, rtol=1e-3, atol_rel=1e-4, name=k)


def test_feed_forward_train_ragged_token_count_and_determinism(ff):
    """M = 1000 tokens (not a multiple of 16 or 64): padding tokens of the last block must not leak into the
    weight gradients; masks drawn by the module; two runs bit-identical (fixed-order split-K reductions)."""
    csd = conformer_state_dict(seed=3)
    rng = np.random.Generator(np.random.PCG64(5))
    x = torch.from_numpy(rng.standard_normal((1000, 64)).astype(np.float32))
    dy = torch.from_numpy(rng.standard_normal((1000, 64)).astype(np.float32))
    gen = torch.Generator(device=DEV).manual_seed(3)
    m1, m2 = ff.masks(1000, gen)                          # byte keep flags; F.dropout's arithmetic is flag / (1 - p)
    assert m1.dtype == torch.uint8 and set(torch.unique(m1).tolist()) == {0, 1} and m2.shape == (1000, 64)
    assert 0.7 < float(m1.float().mean()) < 0.9
    f1, f2 = m1.float().cpu() * 1.25, m2.float().cpu() * 1.25
    leaf = {"ff1." + k: csd["ff1." + k].clone().requires_grad_(True) for k in KEYS}
    xr = x.clone().requires_grad_(True)
    with torch.enable_grad():
        want_y = O.feed_forward_train(leaf, "ff1", xr, f1, f2)
        want_y.backward(dy)
    y = ff.forward(x.to(DEV), m1, m2)
    assert _report("ffn train forward M=1000", rel_err(y, want_y.detach())) < GRAD_TOL
    dx, grads = ff.backward(x.to(DEV), dy.to(DEV), m1, m2)
    first = {k: v.clone() for k, v in grads.items()}
    assert _report("ffn train dL/dx M=1000", rel_err(dx, xr.grad)) < GRAD_TOL
    for k in KEYS:
        assert _report(f"ffn train dL/d[{k}] M=1000", rel_err(grads[k], leaf["ff1." + k].grad)) < GRAD_TOL, k
    dx2, grads2 = ff.backward(x.to(DEV), dy.to(DEV), m1, m2)
    assert torch.equal(dx, dx2) and all(torch.equal(first[k], grads2[k]) for k in KEYS)


def test_feed_forward_backward_fits_its_compact_workspace_with_ragged_tiles(ff):
    """The fused backward (train_x3.hip ffn_train_bwd_aw_x3_kernel, the default) keeps dh [M,256] and per-tile partial rows
    only: `cmgan_ffn_train_workspace_bytes` is ~270 floats per token, not 768 (train.hip ffn_ws_compact).  Its dh stores of a
    ragged last tile must be DROPPED by the buffer descriptor's range check (which sees the per-lane offset only, not the
    scalar tile offset).  Run forward + backward inside a workspace of EXACTLY the advertised size with a guard band behind
    it, at token counts whose last tile is ragged (M = 1000: 8 of 32 valid; M = 33: 1 of 32): the guard stays intact and the
    gradients equal those computed in a roomy workspace bit for bit."""
    import os
    for M in (1000, 33):
        rng = np.random.Generator(np.random.PCG64(11 + M))
        x = torch.from_numpy(rng.standard_normal((M, 64)).astype(np.float32)).to(DEV)
        dy = torch.from_numpy(rng.standard_normal((M, 64)).astype(np.float32)).to(DEV)
        m1, m2 = ff.masks(M, torch.Generator(device=DEV).manual_seed(4))
        ff._ws = None
        need = ff._workspace(M).numel()                    # bytes, as the library advertises them
        if os.environ.get("CMGAN_FFN_BWD_FUSED", "1") != "0":
            assert need < 4 * (65536 + 400 * M + 2 * 256 * 16384 + 4 * 512 * 256) + 4096, need      # compact: < 400 floats / token
        roomy = torch.zeros(need + (1 << 22), dtype=torch.uint8, device=DEV)
        ff._ws = roomy
        ff.forward(x, m1, m2)
        dx_ref, g_ref = ff.backward(x, dy, m1, m2)
        g_ref = {k: v.clone() for k, v in g_ref.items()}
        tight = torch.empty(need + 65536, dtype=torch.uint8, device=DEV)
        tight[need:].fill_(0xA5)
        ff._ws = tight
        ff.forward(x, m1, m2)
        dx, grads = ff.backward(x, dy, m1, m2)
        torch.cuda.synchronize()
        assert bool((tight[need:] == 0xA5).all()), f"M = {M}: the workspace was overrun"
        assert torch.equal(dx, dx_ref) and all(torch.equal(grads[k], g_ref[k]) for k in KEYS), M
    ff._ws = None


def test_feed_forward_backward_is_exactly_scale_equivariant(ff):
    """Gradients arrive with any magnitude (dL/dy of a mean loss is ~1 / numel), far below what an fp16 split can
    represent.  The split-product kernels therefore scale every gradient operand by an exact power of two (per tile /
    per running contraction, train_x3.hip), so the backward of 2^-40 dy must be BIT-IDENTICAL to 2^-40 times the
    backward of dy - input gradient and all six parameter gradients - including across a re-reference of the weight
    gradient's running scale (the second half of the tokens is 2^24 times larger than the first)."""
    rng = np.random.Generator(np.random.PCG64(15))
    x = torch.from_numpy(rng.standard_normal((1000, 64)).astype(np.float32)).to(DEV)
    dy = torch.from_numpy(rng.standard_normal((1000, 64)).astype(np.float32)).to(DEV)
    dy[500:] *= 2.0 ** 24
    gen = torch.Generator(device=DEV).manual_seed(4)
    m1, m2 = ff.masks(1000, gen)
    ff.forward(x, m1, m2)
    dx, grads = ff.backward(x, dy, m1, m2)
    ref = {k: v.clone() for k, v in grads.items()}
    dx = dx.clone()
    k2 = 2.0 ** -40
    dx_s, grads_s = ff.backward(x, dy * k2, m1, m2)
    assert torch.isfinite(dx_s).all() and float(dx_s.abs().max()) > 0
    assert torch.equal(dx_s, dx * k2)
    for k in KEYS:
        assert torch.equal(grads_s[k], ref[k] * k2), k


def test_eval_arithmetic_of_the_train_kernel_matches_the_inference_path(ff):
    """masks = None is the eval forward: 0.5 FF(LN(x)) must agree with the oracle's eval feed_forward."""
    csd = conformer_state_dict(seed=3)
    x = torch.from_numpy(np.random.Generator(np.random.PCG64(6)).standard_normal((5, 33, 64)).astype(np.float32))
    y = ff.forward(x.to(DEV))
    assert _report("ffn train kernel, no dropout, vs eval oracle", rel_err(y, O.feed_forward(csd, "ff1", x))) < 1e-5


def test_weighted_generator_loss(ff):
    """loss = 0.1 loss_ri + 0.9 loss_mag + 0.2 time_loss (train.py:28,143-148, GAN term excluded)."""
    import torch.nn.functional as Fn
    from cmgan_amd.training import generator_loss_terms
    gen = torch.Generator().manual_seed(8)
    er, ei = torch.randn(2, 1, 9, 201, generator=gen), torch.randn(2, 1, 9, 201, generator=gen)
    cs = torch.randn(2, 2, 9, 201, generator=gen)
    ea, ca = torch.randn(2, 800, generator=gen), torch.randn(2, 800, generator=gen)
    loss, terms = generator_loss_terms(ff.engine, er.to(DEV), ei.to(DEV), cs.to(DEV), ea.to(DEV), ca.to(DEV))
    cr, ci = cs[:, 0:1], cs[:, 1:2]
    want = (0.1 * (Fn.mse_loss(er, cr) + Fn.mse_loss(ei, ci))
            + 0.9 * Fn.mse_loss(torch.sqrt(er ** 2 + ei ** 2), torch.sqrt(cr ** 2 + ci ** 2))
            + 0.2 * torch.mean(torch.abs(ea - ca)))
    assert abs(float(loss) - float(want)) < 2e-6 * abs(float(want))
    assert terms.shape == (4,)


def test_validation_step_matches_reference_golden_and_oracle():
    """Generator half of Trainer.test_step through the HIP path: losses of the reference's own modules (golden)."""
    from cmgan_amd import TSCNet
    from cmgan_amd.training import forward_generator_step, validation_step
    from oracle.weights import make_state_dict
    g = load_golden("valstep.npz")
    model = TSCNet(64, 201).load_state_dict(make_state_dict(seed=0)).eval()
    out = forward_generator_step(model, g["clean"].to(DEV), g["noisy"].to(DEV))
    assert _report("validation est_audio vs golden", rel_err(out["est_audio"], g["est_audio"])) < 1e-3
    loss, terms = validation_step(model, g["clean"].to(DEV), g["noisy"].to(DEV))
    for got, key in ((loss, "loss"), (terms[0], "loss_ri"), (terms[1], "loss_mag"), (terms[2], "time_loss")):
        err = abs(float(got) - float(g[key])) / abs(float(g[key]))
        assert _report(f"validation {key} vs golden", err) < 1e-4, key


def test_data_path_feeds_device_batches_to_the_validation_step(tmp_path):
    """SURVEY.md N4 wired end to end: wav pairs on disk -> DemandDataset / load_data (sharded sampler) ->
    DevicePrefetcher (pinned, side-stream H2D into a two-slot ring) -> validation_step on the HIP kernels;
    every batch's losses equal the CPU oracle's on the same host batch."""
    import os
    from scipy.io import wavfile
    from cmgan_amd import TSCNet
    from cmgan_amd.data import DevicePrefetcher, load_data
    from cmgan_amd.training import validation_step
    from oracle.weights import make_state_dict, synthetic_clips
    sd = make_state_dict(seed=0)
    model = TSCNet(64, 201).load_state_dict(sd).eval()
    for split, n in (("train", 5), ("test", 5)):
        for sub in ("clean", "noisy"):
            os.makedirs(tmp_path / split / sub)
        for i in range(n):
            clean = synthetic_clips(1, 900 + 450 * i, seed=70 + i)[0].numpy() * 0.3
            noisy = clean + 0.1 * synthetic_clips(1, clean.size, seed=80 + i)[0].numpy()
            for sub, sig in (("clean", clean), ("noisy", noisy)):
                wavfile.write(str(tmp_path / split / sub / f"p_{i}.wav"), 16000, np.round(sig * 32767).astype(np.int16))
    _, test_loader = load_data(str(tmp_path), batch_size=2, n_cpu=0, cut_len=1600)
    import random
    random.seed(123)                                                   # clips longer than cut_len are cropped at random.randint
    host = [(c.clone(), n.clone()) for c, n, _ in test_loader]        # same sampler order + same crops on the second pass
    random.seed(123)
    seen = 0
    for (clean_d, noisy_d, length), (clean_h, noisy_h) in zip(DevicePrefetcher(test_loader, DEV), host):
        assert clean_d.is_cuda and clean_d.shape == clean_h.shape and torch.equal(clean_d.cpu(), clean_h)
        loss, terms = validation_step(model, clean_d, noisy_d)
        want = O.generator_loss(O.forward_generator_step(sd, clean_h, noisy_h), clean_h)
        assert abs(float(loss) - float(want[0])) < 1e-4 * abs(float(want[0]))
        seen += clean_d.size(0)
    assert seen == 5                                                     # drop_last=False keeps the odd batch


def test_three_adamw_steps_of_the_feed_forward_branch_match_torch():
    """forward (dropout masks) -> backward -> gradient bucket -> AdamW, three times, on the HIP kernels, against
    autograd through the oracle + torch.optim.AdamW on the CPU (train.py:63, 191-193: lr 5e-4, torch defaults)."""
    from cmgan_amd.training import AdamW, FeedForwardTrain, step_lr
    assert step_lr(0) == 5e-4 and step_lr(29) == 5e-4 and step_lr(30) == 2.5e-4 and step_lr(95) == 5e-4 / 8
    csd = conformer_state_dict(seed=3)
    ffm = FeedForwardTrain({k: csd["ff1." + k] for k in KEYS}, dropout=0.2)
    opt = AdamW(ffm.engine, ffm.param_bucket, ffm.grad_bucket, lr=5e-4)
    leaf = {"ff1." + k: csd["ff1." + k].clone().requires_grad_(True) for k in KEYS}
    ref_opt = torch.optim.AdamW([leaf["ff1." + k] for k in KEYS], lr=5e-4)
    rng = np.random.Generator(np.random.PCG64(9))
    gen = torch.Generator(device=DEV).manual_seed(11)
    for it in range(3):
        x = torch.from_numpy(rng.standard_normal((200, 64)).astype(np.float32))
        tgt = torch.from_numpy(rng.standard_normal((200, 64)).astype(np.float32))
        m1, m2 = ffm.masks(200, gen)
        y = ffm.forward(x.to(DEV), m1, m2)
        dy = (2.0 / y.numel()) * (y - tgt.to(DEV))                       # d/dy of mean((y - tgt)^2)
        ffm.backward(x.to(DEV), dy, m1, m2)
        ffm.allreduce_gradients()                                        # identity in one process
        opt.step(step_lr(0))
        ref_opt.zero_grad()
        with torch.enable_grad():
            yr = O.feed_forward_train(leaf, "ff1", x, m1.float().cpu() * 1.25, m2.float().cpu() * 1.25)
            torch.mean((yr - tgt) ** 2).backward()
        ref_opt.step()
    for k in KEYS:
        # Adam divides every gradient element by its own running magnitude, so an absolute gradient error of 1e-6 of the
        # tensor's maximum (the split-f16 products: 2^-21 each) becomes a RELATIVE update error of 1e-3 on an element
        # whose gradient is a thousand times below that maximum: 3 steps x lr 5e-4 x 1e-3 = 1.5e-6 absolute, i.e. 1e-5
        # of the largest parameter (the fp32-MFMA build, -DTRAIN_X3=0, sits at 1e-6 here).  The second bar - the update
        # itself is resolved to 2 % - is unchanged.
        assert _report(f"param {k} after 3 AdamW steps", rel_err(ffm.params[k], leaf["ff1." + k].detach())) < 5e-5, k
        # the update itself (3 x lr = 1.5e-3 per element at most) is resolved, not just the unchanged bulk
        moved = (leaf["ff1." + k].detach() - csd["ff1." + k]).abs().max()
        assert float((ffm.params[k].cpu() - leaf["ff1." + k].detach()).abs().max()) < 2e-2 * float(moved), k


CM_KEYS = ("net.0.weight", "net.0.bias", "net.2.weight", "net.2.bias", "net.4.conv.weight", "net.4.conv.bias",
           "net.5.weight", "net.5.bias", "net.7.weight", "net.7.bias")
CM_ALL = CM_KEYS + ("net.5.running_mean", "net.5.running_var")


def test_conv_module_train_matches_reference_autograd():
    """ConformerConvModule in TRAIN mode (BatchNorm1d on batch statistics + running-stat update): forward, dL/dx and
    all ten parameter gradients against the reference module's own torch autograd (tests/golden/convmod_train.npz)."""
    from cmgan_amd.training import ConvModuleTrain
    g = load_golden("convmod_train.npz")
    csd = conformer_state_dict(seed=3)
    cm = ConvModuleTrain({k: csd["conv." + k] for k in CM_ALL})
    y = cm.forward(g["x"].to(DEV))
    assert _report("conv module train forward", rel_err(y, g["y"])) < GRAD_TOL
    assert _report("BatchNorm running_mean", rel_err(cm.running_mean, g["running_mean"])) < 1e-5
    assert _report("BatchNorm running_var", rel_err(cm.running_var, g["running_var"])) < 1e-5
    dx, grads = cm.backward(g["x"].to(DEV), g["dy"].to(DEV))
    assert _report("conv module dL/dx", rel_err(dx, g["dx"])) < GRAD_TOL
    for k in CM_KEYS:
        want = g["grad_" + k.replace(".", "_")]
        if k == "net.4.conv.bias":
            # a bias in front of a batch-statistics BatchNorm has an exactly zero gradient (sum of dd over the batch
            # vanishes): both torch and the HIP path return rounding noise, compared on the scale of dL/d(bn bias)
            scale = float(g["grad_net_5_bias"].abs().max())
            assert float(grads[k].abs().max()) < 1e-4 * scale and float(want.abs().max()) < 1e-4 * scale
            continue
        assert _report(f"conv module dL/d[{k}]", rel_err(grads[k], want)) < GRAD_TOL, k
        assert_close(grads[k], want, rtol=1e-3, atol_rel=2e-4, name=k)


def test_conv_module_train_longer_sequences_vs_oracle_autograd():
    """N = 5 sequences of L = 101 (a frequency-axis shape: 4 depthwise tiles, the last one ragged)."""
    from cmgan_amd.training import ConvModuleTrain
    csd = conformer_state_dict(seed=3)
    cm = ConvModuleTrain({k: csd["conv." + k] for k in CM_ALL})
    rng = np.random.Generator(np.random.PCG64(13))
    x = torch.from_numpy(rng.standard_normal((5, 101, 64)).astype(np.float32))
    dy = torch.from_numpy(rng.standard_normal((5, 101, 64)).astype(np.float32))
    leaf = {"conv." + k: csd["conv." + k].clone().requires_grad_(True) for k in CM_KEYS}
    xr = x.clone().requires_grad_(True)
    with torch.enable_grad():
        want = O.conv_module_train(leaf, "conv", xr)
        want.backward(dy)
    y = cm.forward(x.to(DEV), update_running_stats=False)
    assert _report("conv module train forward [5x101]", rel_err(y, want.detach())) < GRAD_TOL
    dx, grads = cm.backward(x.to(DEV), dy.to(DEV))
    assert _report("conv module dL/dx [5x101]", rel_err(dx, xr.grad)) < GRAD_TOL
    for k in CM_KEYS:
        if k == "net.4.conv.bias":                                      # exactly zero in exact arithmetic (see above)
            assert float(grads[k].abs().max()) < 1e-4 * float(leaf["conv.net.5.bias"].grad.abs().max())
            continue
        assert _report(f"conv module dL/d[{k}] [5x101]", rel_err(grads[k], leaf["conv." + k].grad)) < GRAD_TOL, k
    dx2, grads2 = cm.backward(x.to(DEV), dy.to(DEV))                    # second backward on the same saved state:
    assert _report("repeat backward dL/dx", rel_err(dx2, dx)) < 1e-6    # (ddn is overwritten in place, so the saved
    with pytest.raises(RuntimeError):                                   #  forward activations must be intact)
        cm.backward(x[:2].to(DEV), dy[:2].to(DEV))


AT_KEYS = ("norm.weight", "norm.bias", "fn.to_q.weight", "fn.to_kv.weight", "fn.to_out.weight", "fn.to_out.bias",
           "fn.rel_pos_emb.weight")


def test_attention_train_matches_reference_autograd():
    """PreNorm(Attention) in TRAIN mode (Shaw relative positions, dropout keep-mask on the to_out output): forward,
    dL/dx and all seven parameter gradients incl. the embedding table vs the reference module's torch autograd."""
    from cmgan_amd.training import AttentionTrain
    g = load_golden("attn_train.npz")
    csd = conformer_state_dict(seed=3)
    at = AttentionTrain({k: csd["attn." + k] for k in AT_KEYS})
    y = at.forward(g["x"].to(DEV), g["mask"].to(DEV))
    assert _report("attention train forward", rel_err(y, g["y"])) < GRAD_TOL
    dx, grads = at.backward(g["x"].to(DEV), g["dy"].to(DEV), g["mask"].to(DEV))
    assert _report("attention dL/dx", rel_err(dx, g["dx"])) < GRAD_TOL
    for k in AT_KEYS:
        want = g["grad_" + k.replace(".", "_")]
        assert _report(f"attention dL/d[{k}]", rel_err(grads[k], want)) < GRAD_TOL, k
        assert_close(grads[k], want, rtol=1e-3, atol_rel=2e-4, name=k)


def test_fused_residual_pointers_equal_a_separate_add(ff):
    """`x = branch(x) + x` (conformer.py:216-219) and its backward ride on each branch's last kernel through the optional
    residual / dresidual pointers: same values as the un-fused call plus a torch add (one extra fp32 rounding at most),
    identical parameter gradients, on a ragged token count."""
    from cmgan_amd.training import AttentionTrain, ConvModuleTrain
    csd = conformer_state_dict(seed=3)
    rng = np.random.Generator(np.random.PCG64(11))
    t = lambda *s: torch.from_numpy(rng.standard_normal(s).astype(np.float32)).to(DEV)
    x, r, dy, dr = t(3, 37, 64), t(3, 37, 64), t(3, 37, 64), t(3, 37, 64)
    close = lambda a, b: float((a - b).abs().max()) <= 2e-6 * float(b.abs().max())
    at = AttentionTrain({k: csd["attn." + k] for k in AT_KEYS})
    cm = ConvModuleTrain({k: csd["conv." + k] for k in CM_ALL})
    m1, m2 = ff.masks(3 * 37, torch.Generator(device=DEV).manual_seed(1))
    am = at.mask(3, 37, torch.Generator(device=DEV).manual_seed(2))
    cases = (("ffn", lambda **k: ff.forward(x, m1, m2, **k), lambda **k: ff.backward(x, dy, m1, m2, **k)),
             ("attn", lambda **k: at.forward(x, am, **k), lambda **k: at.backward(x, dy, am, **k)),
             ("conv", lambda **k: cm.forward(x, update_running_stats=False, **k), lambda **k: cm.backward(x, dy, **k)))
    for name, fwd, bwd in cases:
        plain = fwd()
        assert close(fwd(residual=r), plain + r), name
        dx0, g0 = bwd()
        g0 = {k: v.clone() for k, v in g0.items()}
        dx1, g1 = bwd(dresidual=dr)
        assert close(dx1, dx0 + dr), name
        assert all(torch.equal(g0[k], g1[k]) for k in g0), name
    with pytest.raises(ValueError, match="residual"):
        ff.forward(x, m1, m2, residual=r[:, :5])


@pytest.mark.parametrize("bwd", ["cores", "fused"])
@pytest.mark.parametrize("n,l", [(2, 321), (3, 101), (1, 512), (2, 7), (1, 65), (1, 600), (2, 32), (1, 352)])
def test_attention_train_full_length_sequences_vs_oracle_autograd(n, l, bwd, monkeypatch):
    """bwd = "fused": the one-kernel backward (at_bwd_fused_kernel: wrapped-diagonal ownership, LDS accumulators; taken
    for L <= 352, CMGAN_ATTN_BWD=fused) against the same gradients as the three cores - 32 and 352 are an even number
    of blocks (padded Latin square), 512 and 600 fall back to the cores.
    The two sequence lengths of the 2 s training clip (time axis T = 321, frequency axis F' = 101), a multiple of the
    16-row tile (512: no ragged block anywhere), two short ones (7: a single ragged block; 65: one row in the last
    block) and one longer than max_pos_emb (600: distances beyond +-512 share the end rows of the embedding table,
    in the scores and in its gradient); no dropout."""
    from cmgan_amd.training import AttentionTrain
    monkeypatch.setenv("CMGAN_ATTN_BWD", bwd)
    csd = conformer_state_dict(seed=3)
    at = AttentionTrain({k: csd["attn." + k] for k in AT_KEYS})
    rng = np.random.Generator(np.random.PCG64(17 + l))
    x = torch.from_numpy(rng.standard_normal((n, l, 64)).astype(np.float32))
    dy = torch.from_numpy(rng.standard_normal((n, l, 64)).astype(np.float32))
    leaf = {"attn." + k: csd["attn." + k].clone().requires_grad_(True) for k in AT_KEYS}
    xr = x.clone().requires_grad_(True)
    with torch.enable_grad():
        want = O.attention_train(leaf, "attn", xr)
        want.backward(dy)
    y = at.forward(x.to(DEV))
    assert _report(f"attention train forward [{n}x{l}]", rel_err(y, want.detach())) < GRAD_TOL
    dx, grads = at.backward(x.to(DEV), dy.to(DEV))
    assert _report(f"attention dL/dx [{n}x{l}]", rel_err(dx, xr.grad)) < GRAD_TOL
    for k in AT_KEYS:
        assert _report(f"attention dL/d[{k}] [{n}x{l}]", rel_err(grads[k], leaf["attn." + k].grad)) < GRAD_TOL, k
    with pytest.raises(ValueError):
        at.forward(torch.zeros(1, 4097, 64, device=DEV))


def test_whole_conformer_block_trains_like_the_reference():
    """The reference ConformerBlock in TRAIN mode end to end on the HIP kernels: forward, dL/dx and all 31 parameter
    gradients (one flat bucket) against the module's own torch autograd (tests/golden/block_train.npz); then one
    AdamW step over the whole bucket against torch.optim.AdamW."""
    from cmgan_amd.training import AdamW, ConformerBlockTrain
    g = load_golden("block_train.npz")
    csd = conformer_state_dict(seed=3)
    blk = ConformerBlockTrain(csd)
    assert blk.param_bucket.numel == sum(v.numel() for k, v in csd.items() if v.dtype == torch.float32 and "running" not in k)
    masks = {k[5:]: g[k].to(DEV) for k in g if k.startswith("mask_")}
    y = blk.forward(g["x"].to(DEV), masks)
    assert _report("conformer block train forward", rel_err(y, g["y"])) < GRAD_TOL
    dx, grads = blk.backward(g["dy"].to(DEV))
    assert _report("conformer block dL/dx", rel_err(dx, g["dx"])) < GRAD_TOL
    worst = 0.0
    for k, got in grads.items():
        want = g["grad_" + k.replace(".", "_")]
        if k == "conv.net.4.conv.bias":                                   # exactly zero behind batch-stat BatchNorm
            assert float(got.abs().max()) < 1e-4 * float(g["grad_conv_net_5_bias"].abs().max())
            continue
        err = rel_err(got, want)
        worst = max(worst, err)
        assert err < GRAD_TOL, (k, err)
    _report(f"conformer block: worst of {len(grads)} parameter gradients", worst)
    # one optimiser step over the whole bucket
    opt = AdamW(blk.engine, blk.param_bucket, blk.grad_bucket, lr=5e-4)
    leaf = {k: csd[k].clone().requires_grad_(True) for k in grads}
    for k, v in leaf.items():
        v.grad = g["grad_" + k.replace(".", "_")].clone()
    torch.optim.AdamW(list(leaf.values()), lr=5e-4).step()
    opt.step()
    for k in ("ff1.fn.fn.net.0.weight", "attn.fn.rel_pos_emb.weight", "conv.net.4.conv.weight", "post_norm.bias"):
        assert _report(f"after AdamW: {k}", rel_err(blk.params[k], leaf[k].detach())) < 1e-5, k


def test_tscb_trains_like_the_reference():
    """One two-stage conformer block of the generator (generator.py:72-99) in TRAIN mode: time conformer per
    (b, f'), frequency conformer per (b, t), both residuals and the layout flips on the HIP kernels, channels-last
    [B, T, F', 64]; forward, dL/dx and representative parameter gradients vs the reference TSCB's torch autograd."""
    from cmgan_amd.training import TSCBTrain
    from oracle.weights import make_state_dict
    g = load_golden("tscb_train.npz")
    sd = make_state_dict(seed=0)
    blk = TSCBTrain({k[len("TSCB_1."):]: v for k, v in sd.items() if k.startswith("TSCB_1.")})
    mt = {k[len("mask_time_"):]: g[k].to(DEV) for k in g if k.startswith("mask_time_")}
    mf = {k[len("mask_freq_"):]: g[k].to(DEV) for k in g if k.startswith("mask_freq_")}
    x_cl = g["x"].permute(0, 2, 3, 1).contiguous()                        # NCHW -> channels-last
    y = blk.forward(x_cl.to(DEV), mt, mf)
    assert _report("TSCB train forward", rel_err(y.permute(0, 3, 1, 2), g["y"])) < GRAD_TOL
    dx = blk.backward(g["dy"].permute(0, 2, 3, 1).contiguous().to(DEV))
    assert _report("TSCB dL/dx", rel_err(dx.permute(0, 3, 1, 2), g["dx"])) < GRAD_TOL
    for k in g:
        if k.startswith("grad_"):
            ax = "time" if k.startswith("grad_time") else "freq"
            grads = getattr(blk, ax).grads
            name = next(n for n in grads if n.replace(".", "_") == k[len(f"grad_{ax}_conformer_"):])
            assert _report(f"TSCB dL/d[{ax}.{name}]", rel_err(grads[name], g[k])) < GRAD_TOL, k


def _dense_check(name, blk, g_or_leaf, got_grads, key_fn, scale_key):
    for k, got in got_grads.items():
        want = g_or_leaf(k)
        if k.startswith("conv") and k.endswith(".bias"):                 # in front of an InstanceNorm: exactly zero
            assert float(got.abs().max()) < 1e-4 * float(g_or_leaf(scale_key).abs().max()), k
            continue
        assert _report(f"{name} dL/d[{k}]", rel_err(got, want)) < GRAD_TOL, k


def test_dense_block_train_matches_reference_autograd():
    """DilatedDenseNet (generator.py:6-47): dilated (2,3) convs over the growing concat (slot views, newest-first
    channel map), InstanceNorm2d, PReLU - forward, dL/dx and all twenty parameter gradients vs the reference module's
    torch autograd, channels-last on the HIP side."""
    from cmgan_amd.training import DenseBlockTrain
    from oracle.weights import make_state_dict
    g = load_golden("dense_train.npz")
    sd = make_state_dict(seed=0)
    pre = "dense_encoder.dilated_dense."
    blk = DenseBlockTrain({k[len(pre):]: v for k, v in sd.items() if k.startswith(pre)})
    x = g["x"].permute(0, 2, 3, 1).contiguous()
    y = blk.forward(x.to(DEV))
    assert _report("dense block train forward", rel_err(y.permute(0, 3, 1, 2), g["y"])) < GRAD_TOL
    dx, grads = blk.backward(x.to(DEV), g["dy"].permute(0, 2, 3, 1).contiguous().to(DEV))
    assert _report("dense block dL/dx", rel_err(dx.permute(0, 3, 1, 2), g["dx"])) < GRAD_TOL
    _dense_check("dense block", blk, lambda k: g["grad_" + k.replace(".", "_")], grads, None, "norm1.bias")


def test_dense_block_train_decoder_shape_vs_oracle_autograd():
    """a decoder instance at a ragged plane (T = 21 > 2 * 8 + ..., F = 37): every dilation sees real history and the
    16-position waves straddle rows and clips."""
    from cmgan_amd.training import DenseBlockTrain
    from oracle.weights import make_state_dict
    sd = make_state_dict(seed=0)
    pre = "mask_decoder.dense_block."
    blk = DenseBlockTrain({k[len(pre):]: v for k, v in sd.items() if k.startswith(pre)})
    rng = np.random.Generator(np.random.PCG64(23))
    x = torch.from_numpy(rng.standard_normal((3, 64, 21, 37)).astype(np.float32))
    dy = torch.from_numpy(rng.standard_normal((3, 64, 21, 37)).astype(np.float32))
    leaf = {k: v.clone().requires_grad_(True) for k, v in sd.items() if k.startswith(pre)}
    sdx = dict(sd)
    sdx.update(leaf)
    xr = x.clone().requires_grad_(True)
    with torch.enable_grad():
        want = O.dense_block(sdx, pre[:-1], xr)
        want.backward(dy)
    y = blk.forward(x.permute(0, 2, 3, 1).contiguous().to(DEV))
    assert _report("dense block train forward [3x21x37]", rel_err(y.permute(0, 3, 1, 2), want.detach())) < GRAD_TOL
    dx, grads = blk.backward(x.permute(0, 2, 3, 1).contiguous().to(DEV), dy.permute(0, 2, 3, 1).contiguous().to(DEV))
    assert _report("dense block dL/dx [3x21x37]", rel_err(dx.permute(0, 3, 1, 2), xr.grad)) < GRAD_TOL
    _dense_check("dense block [3x21x37]", blk, lambda k: leaf[pre + k].grad, grads, None, "norm1.bias")


# ---- encoder / decoders / whole generator ---------------------------------------------------------------------------
def _zero_or_close(name, k, got, want, scale, zero_patterns):
    import re
    if any(re.search(z, k) for z in zero_patterns):       # a bias in front of an InstanceNorm: zero up to rounding
        assert float((got.cpu() - want).abs().max()) < 1e-4 * max(scale, float(want.abs().max())), k
        return
    assert _report(f"{name} dL/d[{k}]", rel_err(got, want)) < GRAD_TOL, k


def test_dense_encoder_train_matches_reference_autograd():
    """DenseEncoder (generator.py:50-69): 1x1 conv from (|X|, re, im), IN + PReLU, the dilated dense block, the
    stride-2 (1,3) conv, IN + PReLU - forward and all thirty parameter gradients vs the reference module's autograd."""
    from cmgan_amd.training import DenseEncoderTrain
    from oracle.weights import make_state_dict
    g = load_golden("encoder_train.npz")
    sd = make_state_dict(seed=0)
    pre = "dense_encoder."
    enc = DenseEncoderTrain({k[len(pre):]: v for k, v in sd.items() if k.startswith(pre)})
    y = enc.forward(g["x"].permute(0, 2, 3, 1).contiguous().to(DEV))
    assert _report("encoder train forward", rel_err(y.permute(0, 3, 1, 2), g["y"])) < GRAD_TOL
    grads = enc.backward(g["dy"].permute(0, 2, 3, 1).contiguous().to(DEV))
    scale = float(g["grad_conv_1_1_bias"].abs().max())
    for k, got in grads.items():
        _zero_or_close("encoder", k, got, g["grad_" + k.replace(".", "_")], scale, [r"\.0\.bias$", r"\.conv\d\.bias$"])


@pytest.mark.parametrize("kind", ["mask", "complex"])
def test_decoder_train_matches_reference_autograd(kind):
    """MaskDecoder / ComplexDecoder (generator.py:121-156): dense block, sub-pixel conv (pixel shuffle as an index
    map), the (1,2) tail conv and the IN / PReLU heads - forward, dL/dx and every parameter gradient."""
    from cmgan_amd.training import DecoderTrain
    from oracle.weights import make_state_dict
    g = load_golden("maskdec_train.npz" if kind == "mask" else "complexdec_train.npz")
    sd = make_state_dict(seed=0)
    pre = kind + "_decoder."
    st = {k[len(pre):]: v for k, v in sd.items() if k.startswith(pre)}
    if kind == "mask":
        st["prelu_out.weight"] = st["prelu_out.weight"][:21].clone()
    dec = DecoderTrain(kind, st, num_features=21)
    y = dec.forward(g["x"].permute(0, 2, 3, 1).contiguous().to(DEV))
    y_ref = g["y"][:, 0] if kind == "mask" else g["y"].permute(0, 2, 3, 1)
    assert _report(f"{kind} decoder train forward", rel_err(y, y_ref)) < GRAD_TOL
    dy = g["dy"][:, 0].contiguous() if kind == "mask" else g["dy"].permute(0, 2, 3, 1).contiguous()
    dx, grads = dec.backward(dy.to(DEV))
    assert _report(f"{kind} decoder dL/dx", rel_err(dx.permute(0, 3, 1, 2), g["dx"])) < GRAD_TOL
    scale = float(g["grad_norm_bias"].abs().max())
    zero = [r"\.conv\d\.bias$", r"^conv_1\.bias$"] if kind == "mask" else [r"\.conv\d\.bias$"]
    for k, got in grads.items():
        _zero_or_close(f"{kind} decoder", k, got, g["grad_" + k.replace(".", "_")], scale, zero)


@pytest.mark.parametrize("kind", ["mask", "complex"])
def test_decoder_train_ragged_plane_vs_oracle_autograd(kind):
    """B = 3, T = 21, F' = 37 (F = 73): 16-position waves straddle rows and clips in every kernel of the heads."""
    from cmgan_amd.training import DecoderTrain
    from oracle.weights import make_state_dict
    sd = make_state_dict(seed=0)
    pre = kind + "_decoder."
    st = {k: v.clone() for k, v in sd.items() if k.startswith(pre)}
    if kind == "mask":
        st[pre + "prelu_out.weight"] = st[pre + "prelu_out.weight"][:73].clone()
    rng = np.random.Generator(np.random.PCG64(29))
    x = torch.from_numpy(rng.standard_normal((3, 64, 21, 37)).astype(np.float32))
    dy = torch.from_numpy(rng.standard_normal((3, 1 if kind == "mask" else 2, 21, 73)).astype(np.float32))
    leaf = {k: v.clone().requires_grad_(True) for k, v in st.items()}
    sdx = dict(sd)
    sdx.update(leaf)
    xr = x.clone().requires_grad_(True)
    with torch.enable_grad():
        want = (O.mask_decoder if kind == "mask" else O.complex_decoder)(sdx, xr)
        want.backward(dy)
    dec = DecoderTrain(kind, {k[len(pre):]: v for k, v in st.items()}, num_features=73)
    y = dec.forward(x.permute(0, 2, 3, 1).contiguous().to(DEV))
    y_ref = want.detach()[:, 0] if kind == "mask" else want.detach().permute(0, 2, 3, 1)
    assert _report(f"{kind} decoder forward [3x21x37]", rel_err(y, y_ref)) < GRAD_TOL
    dyd = dy[:, 0].contiguous() if kind == "mask" else dy.permute(0, 2, 3, 1).contiguous()
    dx, grads = dec.backward(dyd.to(DEV))
    assert _report(f"{kind} decoder dL/dx [3x21x37]", rel_err(dx.permute(0, 3, 1, 2), xr.grad)) < GRAD_TOL
    scale = float(leaf[pre + "norm.bias"].grad.abs().max())
    zero = [r"\.conv\d\.bias$", r"^conv_1\.bias$"] if kind == "mask" else [r"\.conv\d\.bias$"]
    for k, got in grads.items():
        _zero_or_close(f"{kind} decoder [3x21x37]", k, got, leaf[pre + k].grad, scale, zero)


def test_loss_backward_matches_reference_autograd():
    """dL/d est_real, dL/d est_imag of 0.1 loss_ri + 0.9 loss_mag + 0.2 time_loss, incl. the adjoint of torch.istft and
    of power_uncompress, at the reference's own network outputs (generator_step.npz)."""
    from cmgan_amd.engine import Engine
    from cmgan_amd._lib import check
    g = load_golden("generator_step.npz")
    eng = Engine(device=DEV)
    clean, noisy = g["clean"].to(DEV), g["noisy"].to(DEV)
    c = eng.rms_scale(noisy)
    clean_spec = eng.stft_compress(clean, c)
    er, ei = g["est_real"].to(DEV).contiguous(), g["est_imag"].to(DEV).contiguous()
    audio = eng.uncompress_istft(er, ei)
    d_real, d_imag = torch.empty_like(er), torch.empty_like(ei)
    B, _, T, F = er.shape
    check(eng._h, eng.lib.cmgan_loss_backward(eng._h, er.data_ptr(), ei.data_ptr(), clean_spec.data_ptr(), B, T,
                                              audio.data_ptr(), clean.data_ptr(), 0.1, 0.9, 0.2, d_real.data_ptr(),
                                              d_imag.data_ptr(), eng._stream()))
    assert _report("loss backward d_real", rel_err(d_real, g["d_real"])) < GRAD_TOL
    assert _report("loss backward d_imag", rel_err(d_imag, g["d_imag"])) < GRAD_TOL
    # spectral terms only (no audio pointers) vs autograd through the oracle's loss
    erl, eil = g["est_real"].clone().requires_grad_(True), g["est_imag"].clone().requires_grad_(True)
    with torch.enable_grad():
        out = {"est_real": erl, "est_imag": eil, "clean_spec": clean_spec.cpu(), "est_audio": torch.zeros(2, 800)}
        loss, *_ = O.generator_loss(out, torch.zeros(2, 800), (0.3, 0.7, 0.0))
        loss.backward()
    check(eng._h, eng.lib.cmgan_loss_backward(eng._h, er.data_ptr(), ei.data_ptr(), clean_spec.data_ptr(), B, T, None, None,
                                              0.3, 0.7, 0.0, d_real.data_ptr(), d_imag.data_ptr(), eng._stream()))
    assert _report("loss backward (spectral only) d_real", rel_err(d_real, erl.grad)) < GRAD_TOL
    assert _report("loss backward (spectral only) d_imag", rel_err(d_imag, eil.grad)) < GRAD_TOL


def test_generator_train_step_matches_the_reference_trainer():
    """One optimisation step of the reference trainer's generator half without the metric discriminator
    (generator_step.npz: reference TSCNet in train mode, the 40 dropout masks of synthetic_dropout_masks(77),
    loss, loss.backward(), AdamW.step(), second forward): losses, network outputs, all 335 parameter-gradient
    digests, the loss after the update and a BatchNorm's running statistics."""
    from cmgan_amd.synth import sample_indices, synthetic_dropout_masks
    from cmgan_amd.training import AdamW, GeneratorTrain, generator_train_step
    from oracle.weights import make_state_dict
    g = load_golden("generator_step.npz")
    sd = make_state_dict(seed=0)
    gen = GeneratorTrain(sd, device=DEV)
    assert len(gen.params) == 335
    masks = [tuple({k: torch.from_numpy(v).to(DEV) for k, v in d.items()} for d in pair)
             for pair in synthetic_dropout_masks(77, 2, 9, 101)]
    clean, noisy = g["clean"].to(DEV), g["noisy"].to(DEV)

    # forward alone first: the network outputs
    eng = gen.engine
    spec = eng.stft_compress(noisy, eng.rms_scale(noisy))
    er, ei = gen.forward(spec, masks)
    assert _report("generator train forward est_real", rel_err(er, g["est_real"])) < GRAD_TOL
    assert _report("generator train forward est_imag", rel_err(ei, g["est_imag"])) < GRAD_TOL
    gen2 = GeneratorTrain(sd, engine=eng)                   # fresh buffers / buckets for the real step
    opt = AdamW(eng, gen2.param_bucket, gen2.grad_bucket, lr=5e-4)
    loss, terms = generator_train_step(gen2, opt, clean, noisy, masks=masks)
    assert abs(float(loss) - float(g["loss"])) < 1e-4 * abs(float(g["loss"]))
    assert _report("generator step loss terms", rel_err(terms[:3], g["terms"])) < GRAD_TOL
    keys = [k[len("gsmp_"):] for k in g if k.startswith("gsmp_")]
    assert set(keys) == set(gen2.grads)
    scale = max(float(g["gl2_" + k]) for k in keys)
    # The whole-network gradient is NOT smooth in its inputs (PReLU / L1-loss kinks flip with rounding): the reference's
    # own fp32 autograd differs from its fp64 autograd by 2e-4 (median over tensors, relative to each tensor's max)
    # and 1.1e-2 (worst tensor) on this fixture - test_oracle_golden.py::test_generator_gradient_noise_floor measures
    # it.  Every module is pinned at 1e-6 on its own above; here the bar is that noise floor.
    errs = []
    for k in keys:
        got = gen2.grads[k].reshape(-1)
        l2 = float(g["gl2_" + k])
        want = g["gsmp_" + k]
        smp = got[torch.from_numpy(sample_indices(got.numel())).to(DEV)].cpu()
        d = float((smp - want).abs().max())
        assert d < 3e-2 * float(want.abs().max()) + 1e-6 * scale, (k, d, l2)
        assert abs(float(got.double().norm()) - l2) < 3e-2 * l2 + 1e-6 * scale, k
        if l2 > 1e-4 * scale:
            errs.append(d / float(want.abs().max()))
    errs = np.array(errs)
    _report("generator step parameter gradients: median error relative to each tensor's max", float(np.median(errs)))
    _report("generator step parameter gradients: worst tensor", float(errs.max()))
    assert float(np.median(errs)) < 2e-3
    # the update happened: a second step with the same data and masks starts from the reference's second loss
    loss2, terms2 = generator_train_step(gen2, opt, clean, noisy, masks=masks)
    assert _report("loss after one AdamW step", abs(float(loss2) - float(g["loss2"])) / abs(float(g["loss2"]))) < 2e-3
    # BatchNorm buffers: two train-mode forwards happened on each side (momentum-0.1 updates on batch statistics)
    out_sd = gen2.state_dict()
    assert len(out_sd) == 359
    assert int(out_sd["TSCB_2.freq_conformer.conv.net.5.num_batches_tracked"]) == 102
    assert _report("BatchNorm running_mean after two steps",
                   rel_err(out_sd["TSCB_2.freq_conformer.conv.net.5.running_mean"], g["bn_mean"])) < 1e-3
    assert _report("BatchNorm running_var after two steps",
                   rel_err(out_sd["TSCB_2.freq_conformer.conv.net.5.running_var"], g["bn_var"])) < 1e-3


def _whole_step_vs_oracle(sd, clean, noisy, npm, tag, bar, n_fft=400, hop=100, loss_weights=(0.1, 0.9, 0.2),
                          allow_cancelling=()):
    """One generator optimisation step on the HIP path against autograd through the oracle on the same state dict,
    clips and dropout masks: loss terms, network outputs, output gradients and ALL parameter gradients, each tensor
    held to `bar` relative to its own maximum (worst tensor reported)."""
    from cmgan_amd.engine import Engine
    from cmgan_amd.training import AdamW, GeneratorTrain, generator_train_step
    tm = lambda dev=None: [tuple({k: (torch.from_numpy(v) if dev is None else torch.from_numpy(v).to(dev))
                                  for k, v in d.items()} for d in pair) for pair in npm]
    want = O.generator_step_gradients(sd, clean, noisy, tm(), loss_weights=loss_weights, n_fft=n_fft, hop=hop)
    gen = GeneratorTrain(sd, device=DEV) if n_fft == 400 else GeneratorTrain(sd, engine=Engine(n_fft=n_fft, hop=hop, device=DEV))
    opt = AdamW(gen.engine, gen.param_bucket, gen.grad_bucket, lr=5e-4)
    loss, terms = generator_train_step(gen, opt, clean.to(DEV), noisy.to(DEV), loss_weights=loss_weights, masks=tm(DEV),
                                       update=False)
    assert _report(f"{tag}: loss", abs(float(loss) - float(want["loss"])) / abs(float(want["loss"]))) < 1e-4
    assert _report(f"{tag}: loss terms", rel_err(terms[:3], want["terms"])) < 1e-4
    scale = max(float(v.abs().max()) for v in want["grads"].values())
    assert set(want["grads"]) == set(gen.grads)
    # Every tensor relative to its OWN maximum.  The only exceptions are tensors whose gradient is mathematically zero
    # - the 24 conv biases that sit directly in front of an InstanceNorm / BatchNorm, which removes them: rounding noise
    # on both sides, no relative error to speak of - or vanishing (a relative-position table of which a short sequence
    # touches few rows): recognised by a maximum below 1e-6 of the largest gradient and held to an ABSOLUTE error
    # below that same 1e-6 of the largest gradient.
    # A third kind shows up at B = 4 x T = 321: NEARLY cancelling sums - the sub-pixel conv bias (two sub-pixel biases per
    # channel, of which the InstanceNorm behind the pixel shuffle removes the mean) and the dense block's last norm bias in
    # the mask decoder, maxima 6e-4 of the largest gradient - whose fp32 column sums over 4 x 321 x 202 positions differ
    # between any two summation orders by 1e-4 of that small maximum (the oracle's own fp32 autograd is 6.5e-5 from its
    # fp64 one on it at B = 2): a tensor whose error is below the bar only in ABSOLUTE terms (1e-6 of the largest
    # gradient, the zero-gradient floor) passes if it is also within 1e-2 of its own maximum.  OPT-IN: only the tensors the
    # caller names in `allow_cancelling` (the B = 4 test names exactly those two); every other caller keeps the strict rule.
    FLOOR = 1e-6
    rel, small, cancel = [], [], []
    for k, w in want["grads"].items():
        d, mx = float((gen.grads[k].cpu() - w).abs().max()), float(w.abs().max())
        if mx < FLOOR * scale:
            small.append((d / (FLOOR * scale), k))
        elif k in allow_cancelling and d / mx >= bar and d < FLOOR * scale and d / mx < 1e-2:
            cancel.append((d / mx, k))
        else:
            rel.append((d / mx, k))
    rel.sort(reverse=True)
    small.sort(reverse=True)
    for e, k in rel[:3]:
        _report(f"{tag}: {k}", e)
    for e, k in cancel:
        _report(f"{tag}: {k} (cancelling sum, absolute error below 1e-6 of the largest gradient)", e)
    _report(f"{tag}: worst of {len(rel)} gradient tensors, each relative to its own max", rel[0][0])
    if small:
        _report(f"{tag}: worst of {len(small)} zero-gradient tensors, absolute error in units of 1e-6 of the largest "
                f"gradient ({small[0][1]})", small[0][0])
    assert len(small) <= 32, small
    assert rel[0][0] < bar, rel[0]
    assert not small or small[0][0] < 1.0, small[0]


def test_generator_step_of_the_kink_free_twin_holds_every_gradient_tensor_to_the_gate():
    """The whole training pipeline (STFT, TSCNet in train mode, losses, backward through every module) on the state dict
    whose PReLU slopes are 1 - 1e-3 i / n (cmgan_amd.synth.kink_free_twin): without kinks the gradient is well defined
    to rounding (test_oracle_golden.py: fp32 vs fp64 autograd agree to 1e-5 on the worst tensor), so EVERY one of the
    335 gradient tensors is held to 1e-4 of its own maximum (measured 1.1e-5; 3.5e-5 at T = 321), ten times inside the
    1e-3 gate - an indexing error in a rarely hit path anywhere upstream of the loss
    cannot hide in kink noise here (it can in the default-slope fixtures below, whose bar is the measured noise floor;
    the PReLU derivative itself is pinned by the per-module tests on the default slopes)."""
    from cmgan_amd.synth import kink_free_twin, synthetic_dropout_masks
    from oracle.weights import make_state_dict
    g = load_golden("generator_step.npz")
    _whole_step_vs_oracle(kink_free_twin(make_state_dict(seed=0)), g["clean"], g["noisy"],
                          synthetic_dropout_masks(77, 2, 9, 101), "kink-free twin, B = 2 x T = 9", 1e-4)


def test_adamw_trajectory_of_the_whole_generator_stays_with_torch_over_four_steps():
    """Four optimisation steps of the WHOLE generator (forward in train mode with fixed dropout masks, the three loss terms,
    backward through every module, gradient bucket, AdamW) on the HIP path against autograd through the oracle +
    torch.optim.AdamW on the CPU, on the kink-free twin (src/train.py:63, 185-193: lr 5e-4, torch defaults).  Adam divides
    each gradient element by its own running magnitude, so the split-f16 products' absolute gradient error (1e-6 of a
    tensor's maximum) becomes a relative UPDATE error of up to 1e-3 on elements far below that maximum; this bounds what
    that does to the trajectory: the loss of every step, every parameter after four steps relative to the largest
    parameter of its tensor, and the update itself resolved to a few per cent on every tensor that moved."""
    from cmgan_amd.synth import kink_free_twin, synthetic_dropout_masks
    from cmgan_amd.training import AdamW, GeneratorTrain, generator_train_step
    from oracle.weights import make_state_dict
    g = load_golden("generator_step.npz")
    sd = kink_free_twin(make_state_dict(seed=0))
    clean, noisy = g["clean"], g["noisy"]
    npm = synthetic_dropout_masks(77, 2, 9, 101)
    tm = lambda dev=None: [tuple({k: (torch.from_numpy(v) if dev is None else torch.from_numpy(v).to(dev))
                                  for k, v in d.items()} for d in pair) for pair in npm]
    steps, lr = 4, 5e-4
    gen = GeneratorTrain(sd, device=DEV)
    opt = AdamW(gen.engine, gen.param_bucket, gen.grad_bucket, lr=lr)
    masks_dev = tm(DEV)
    got_losses = [float(generator_train_step(gen, opt, clean.to(DEV), noisy.to(DEV), masks=masks_dev)[0]) for _ in range(steps)]

    ref = {k: v.clone() for k, v in sd.items()}
    keys = [k for k, v in sd.items() if v.is_floating_point() and "running_" not in k]
    leaves = {k: ref[k].clone().requires_grad_(True) for k in keys}
    ref_opt = torch.optim.AdamW([leaves[k] for k in keys], lr=lr)
    want_losses = []
    for _ in range(steps):
        cur = dict(ref)
        cur.update({k: v.detach() for k, v in leaves.items()})
        w = O.generator_step_gradients(cur, clean, noisy, tm())
        want_losses.append(float(w["loss"]))
        ref_opt.zero_grad()
        for k in keys:
            leaves[k].grad = w["grads"][k] if w["grads"][k] is not None else torch.zeros_like(leaves[k])
        ref_opt.step()
    for i, (a, b) in enumerate(zip(got_losses, want_losses)):
        assert _report(f"AdamW trajectory: loss of step {i}", abs(a - b) / abs(b)) < 2e-5, (i, a, b)
    assert want_losses[-1] < want_losses[0]                      # the trajectory goes somewhere
    # Tensors whose gradient is mathematically zero (the 24 conv biases in front of a normalisation) or vanishing get
    # rounding noise on BOTH sides, and Adam turns noise of the size of its eps into steps of up to lr with a random
    # sign: they are only held to the bound of that random walk.  Recognised as in _whole_step_vs_oracle, on the
    # first step's oracle gradients.
    first = O.generator_step_gradients(sd, clean, noisy, tm())["grads"]
    gscale = max(float(v.abs().max()) for v in first.values() if v is not None)
    # Everywhere else the same mechanism acts on single ELEMENTS: an element whose gradient happens to lie below the
    # rounding noise of its tensor (1e-6 of the tensor's maximum; a 16 k-element weight has a few) takes steps of
    # arbitrary sign on either side - the reference's own fp32 and fp64 trajectories differ there too.  So the maximum
    # over elements is only held to the random-walk bound (measured: 0.5 lr after four steps on the worst tensor), and
    # the trajectory is judged by the RMS error of each tensor against the RMS of its own movement.
    worst_rms, worst_max, nzero = (0.0, ""), (0.0, ""), 0
    for k in keys:
        got, want = gen.params[k].cpu().double(), leaves[k].detach().double()
        d = got - want
        dmax = float(d.abs().max())
        if first[k] is None or float(first[k].abs().max()) < 1e-6 * gscale:
            nzero += 1
            assert dmax <= 2 * steps * lr * 1.01, (k, dmax)
            continue
        assert dmax <= 2 * steps * lr * 1.01, (k, dmax)
        worst_max = max(worst_max, (dmax / lr, k))
        move = float((want - sd[k].double()).pow(2).mean().sqrt())
        if move > 0.1 * lr:                                      # a tensor Adam really moved (its step is ~lr per element)
            worst_rms = max(worst_rms, (float(d.pow(2).mean().sqrt()) / move, k))
    assert nzero <= 32, nzero
    _report(f"AdamW trajectory: worst RMS parameter error after {steps} steps relative to the tensor's RMS movement "
            f"({worst_rms[1]})", worst_rms[0])
    _report(f"AdamW trajectory: largest single-element difference in units of lr ({worst_max[1]})", worst_max[0])
    assert worst_rms[0] < 1e-2, worst_rms                       # measured 1.0e-3; losses 2e-6; largest element 0.33 lr


def test_generator_step_at_full_length_T321_kink_free_twin_vs_oracle_autograd():
    """The same at the benchmark's clip length: one 2 s clip, T = 321 frames (21-block time sequences, the 321-row
    attention, every tile-edge path of the training kernels at their real sizes), all 335 gradient tensors."""
    from cmgan_amd.synth import kink_free_twin, synthetic_clips, synthetic_dropout_masks
    from oracle.weights import make_state_dict
    clean = synthetic_clips(1, 32000, seed=41)
    noisy = clean + 0.3 * synthetic_clips(1, 32000, seed=42)
    _whole_step_vs_oracle(kink_free_twin(make_state_dict(seed=0)), clean, noisy,
                          synthetic_dropout_masks(91, 1, 321, 101), "kink-free twin, B = 1 x T = 321", 1e-4)


@pytest.mark.slow            # 125 s of CPU autograd at B = 4 x T = 321; the B = 1 x T = 321 twin above stays in the default tier
def test_generator_step_at_batch_4_x_T321_kink_free_twin_vs_oracle_autograd():
    """Four 2 s clips (T = 321): the largest whole step a CPU autograd oracle finishes in minutes (the bench's 32 clips per
    GPU do not fit one) - multi-clip BatchNorm batch statistics over 4 x 321 x 101 positions, several clips per XCD in the
    per-XCD block orders of the training kernels, clip strides in every byte offset - all 335 gradient tensors at 1e-4.
    The time-domain L1 term is switched off here (weight 0): with 4 x 32 000 samples some |est - clean| falls inside the
    forward's rounding error (this draw: 5.8e-6 at sample 1032 of clip 1, measured with tests/probes/b4_probe.py), the
    sign of that one sample flips and the output gradient of its four frames moves by 5e-3 - the one kink the twin
    does not remove.  The L1 term's own backward is pinned at T = 321 by the one-clip test above (no such sample there)
    and at T = 33 by test_full_loss_gradient_at_the_network_output_vs_oracle_autograd."""
    from cmgan_amd.synth import kink_free_twin, synthetic_clips, synthetic_dropout_masks
    from oracle.weights import make_state_dict
    clean = synthetic_clips(4, 32000, seed=43)
    noisy = clean + 0.3 * synthetic_clips(4, 32000, seed=44)
    _whole_step_vs_oracle(kink_free_twin(make_state_dict(seed=0)), clean, noisy,
                          synthetic_dropout_masks(92, 4, 321, 101), "kink-free twin, B = 4 x T = 321, no L1 term", 1e-4,
                          loss_weights=(0.1, 0.9, 0.0),
                          allow_cancelling=("mask_decoder.sub_pixel.conv.bias", "mask_decoder.dense_block.norm4.bias"))


def test_generator_step_at_48_khz_kink_free_twin_holds_every_gradient_tensor_to_the_gate():
    """BASELINE configs[3] (n_fft 1200 / hop 300, F = 601, F' = 301) on the kink-free twin: the default-slope 48 kHz test
    below can only be held to the kink noise, this one holds all 335 tensors to 1e-4 of their own maximum."""
    from cmgan_amd.synth import kink_free_twin, synthetic_dropout_masks
    from oracle.weights import make_state_dict, synthetic_clips
    B, L = 1, 2400
    clean = synthetic_clips(B, L, seed=61)
    noisy = clean + 0.3 * synthetic_clips(B, L, seed=62)
    _whole_step_vs_oracle(kink_free_twin(make_state_dict(seed=0, num_features=601)), clean, noisy,
                          synthetic_dropout_masks(81, B, L // 300 + 1, 301), "kink-free twin, 48 kHz, B = 1 x T = 9", 1e-4,
                          n_fft=1200, hop=300)


def test_adversarial_step_of_the_kink_free_twins_holds_every_gradient_tensor_to_the_gate():
    """Trainer.train_step (train.py:173-205) on kink-free twins of BOTH networks (generator PReLU slopes ~ 1, and the
    discriminator's five PReLUs likewise): the generator gradient of the FULL loss incl. the 0.05 x metric-discriminator
    term (through cmgan_mag_pair_backward and the discriminator's input gradient) and the discriminator gradient of
    mse(D(clean, clean), 1) + mse(D(clean, est), labels) (two slots, the second accumulated) against autograd through the
    oracle - every one of the 335 + 22 tensors at 1e-4 of its own maximum, so an indexing error in the GAN-gradient path
    cannot hide in kink noise (the default-slope adversarial test below is held to that noise).  The discriminator's
    max-pool stays (its argmax is stable under rounding except at exact ties)."""
    from cmgan_amd.synth import discriminator_state_dict, kink_free_twin, synthetic_clips, synthetic_dropout_masks
    from cmgan_amd.training import AdamW, DiscriminatorTrain, GeneratorTrain, adversarial_train_step
    from oracle.weights import make_state_dict
    import torch.nn.functional as F
    sd = kink_free_twin(make_state_dict(seed=0))
    dsd = dict(discriminator_state_dict(0))
    for k in ("layers.2.weight", "layers.5.weight", "layers.8.weight", "layers.11.weight", "layers.16.weight"):
        n = dsd[k].numel()
        dsd[k] = (1.0 - 1e-3 * torch.arange(n, dtype=torch.float64) / n).to(dsd[k].dtype).reshape(dsd[k].shape)
    B, L, T = 2, 3200, 33
    clean = synthetic_clips(B, L, seed=71) * 0.5
    noisy = clean + 0.2 * synthetic_clips(B, L, seed=72)
    pesq = torch.tensor([0.41, 0.63])
    npm = synthetic_dropout_masks(83, B, T, 101)
    drs = np.random.RandomState(84)
    dmk = [torch.from_numpy((drs.random_sample((B, 64)) >= 0.3).astype(np.float32) / np.float32(0.7)) for _ in range(3)]
    tm = lambda dev=None: [tuple({k: (torch.from_numpy(v) if dev is None else torch.from_numpy(v).to(dev))
                                  for k, v in d.items()} for d in pair) for pair in npm]
    # ---- oracle: generator half, then the discriminator half on the buffers the generator half left ----
    wg = O.adversarial_generator_gradients(sd, dsd, clean, noisy, tm(), dmk[0])
    dleaf = {k: v.detach().clone().requires_grad_(True) for k, v in dsd.items()
             if v.is_floating_point() and not k.endswith(("_u", "_v"))}
    dsx = dict(dsd)
    dsx.update(wg["disc_buffers"])
    dsx.update(dleaf)
    cs = wg["clean_spec"]
    clean_mag = torch.sqrt(cs[:, 0:1] ** 2 + cs[:, 1:2] ** 2).permute(0, 1, 3, 2)
    est_mag = torch.sqrt(wg["est_real"] ** 2 + wg["est_imag"] ** 2).permute(0, 1, 3, 2)
    with torch.enable_grad():
        s_enh, new1 = O.discriminator(dsx, clean_mag, est_mag, dmk[1], train=True)
        dsx.update(new1)
        s_max, _ = O.discriminator(dsx, clean_mag, clean_mag, dmk[2], train=True)
        loss_d = F.mse_loss(s_max.flatten(), torch.ones(B)) + F.mse_loss(s_enh.flatten(), pesq)
        loss_d.backward()
    # ---- HIP path ----
    gen = GeneratorTrain(sd, device=DEV)
    disc = DiscriminatorTrain(dsd, engine=gen.engine)
    opt_g = AdamW(gen.engine, gen.param_bucket, gen.grad_bucket, lr=5e-4)
    opt_d = AdamW(gen.engine, disc.param_bucket, disc.grad_bucket, lr=1e-3)
    loss, terms, gan, got_d = adversarial_train_step(gen, disc, opt_g, opt_d, clean.to(DEV), noisy.to(DEV), pesq.to(DEV),
                                                     masks=tm(DEV), disc_masks=[m.to(DEV) for m in dmk], update=False)
    assert _report("twin adversarial step: generator loss", abs(float(loss) - float(wg["loss"])) / float(wg["loss"])) < 1e-4
    assert _report("twin adversarial step: gen_loss_GAN", abs(float(gan) - float(wg["gan"])) / float(wg["gan"])) < 1e-4
    assert _report("twin adversarial step: discriminator loss", abs(float(got_d) - float(loss_d.detach())) / float(loss_d.detach())) < 1e-4
    scale = max(float(v.abs().max()) for v in wg["grads"].values())
    rel, small = [], []
    for k, w in wg["grads"].items():
        d, mx = float((gen.grads[k].cpu() - w).abs().max()), float(w.abs().max())
        (rel if mx >= 1e-6 * scale else small).append((d / max(mx, 1e-6 * scale), k))
    rel.sort(reverse=True)
    small.sort(reverse=True)
    _report(f"twin adversarial step: worst of {len(rel)} generator gradient tensors ({rel[0][1]})", rel[0][0])
    assert rel[0][0] < 1e-4, rel[:3]
    assert len(small) <= 32 and (not small or small[0][0] < 1.0), small[:3]
    worst = (0.0, "")
    dscale = max(float(leaf.grad.abs().max()) for leaf in dleaf.values())
    for k, leaf in dleaf.items():                          # (conv biases in front of an InstanceNorm: zero gradient)
        w = leaf.grad
        e = float((disc.grads[k].cpu() - w).abs().max()) / max(float(w.abs().max()), 1e-6 * dscale)
        worst = max(worst, (e, k))
    _report(f"twin adversarial step: worst of {len(dleaf)} discriminator gradient tensors ({worst[1]})", worst[0])
    assert set(dleaf) == set(disc.grads)
    assert worst[0] < 1e-4, worst


# ---- metric discriminator + the full adversarial step -----------------------------------------------------------------
def test_discriminator_matches_reference_autograd():
    """Discriminator(ndf=16) (discriminator.py:29-64) in train mode: spectral-norm power iteration (u / v buffers), the
    four strided convs + InstanceNorm + PReLU, max pool, the two spectral-norm Linear layers, Dropout mask,
    LearnableSigmoid - score, updated buffers, dL/dx, dL/dy and all 22 parameter gradients vs the reference module's
    autograd; then the eval-mode score with the updated buffers."""
    from cmgan_amd.synth import discriminator_state_dict
    from cmgan_amd.training import DiscriminatorTrain
    g = load_golden("disc_train.npz")
    disc = DiscriminatorTrain(discriminator_state_dict(0), device=DEV)
    xy = torch.stack([g["x"][:, 0].permute(0, 2, 1), g["y"][:, 0].permute(0, 2, 1)], dim=-1).contiguous().to(DEV)
    mask = g["mask"].to(DEV)
    score = disc.forward(xy, mask, train=True)
    assert _report("discriminator score (train mode)", rel_err(score, g["score"][:, 0])) < GRAD_TOL
    for k, v in disc.buffers.items():
        assert _report(f"power iteration {k}", rel_err(v, g["new_" + k.replace(".", "_")])) < GRAD_TOL, k
    dxy = disc.backward(g["dscore"][:, 0].contiguous().to(DEV))
    assert _report("discriminator dL/dx", rel_err(dxy[..., 0].permute(0, 2, 1), g["dx"][:, 0])) < GRAD_TOL
    assert _report("discriminator dL/dy", rel_err(dxy[..., 1].permute(0, 2, 1), g["dy"][:, 0])) < GRAD_TOL
    for k, got in disc.grads.items():
        assert _report(f"discriminator dL/d[{k}]", rel_err(got, g["grad_" + k.replace(".", "_")])) < GRAD_TOL, k
    score_eval = disc.forward(xy, None, train=False)
    assert _report("discriminator score (eval mode)", rel_err(score_eval, g["score_eval"][:, 0])) < GRAD_TOL
    sd = disc.state_dict()
    assert len(sd) == 34


def test_adversarial_train_step_matches_the_reference_trainer():
    """Trainer.train_step (train.py:173-205) with given PESQ labels: generator loss incl. the metric-discriminator term,
    its gradients (digests of all 335 tensors), AdamW on the generator, the discriminator's two scores and loss, all 22
    discriminator gradients, AdamW on the discriminator, and the generator loss of a second step with both updated."""
    from cmgan_amd.synth import discriminator_state_dict, sample_indices, synthetic_dropout_masks
    from cmgan_amd.training import AdamW, DiscriminatorTrain, GeneratorTrain, adversarial_train_step
    from oracle.weights import make_state_dict
    g = load_golden("adversarial_step.npz")
    gen = GeneratorTrain(make_state_dict(seed=0), device=DEV)
    eng = gen.engine
    disc = DiscriminatorTrain(discriminator_state_dict(0), engine=eng)
    opt_g = AdamW(eng, gen.param_bucket, gen.grad_bucket, lr=5e-4)
    opt_d = AdamW(eng, disc.param_bucket, disc.grad_bucket, lr=1e-3)
    masks = [tuple({k: torch.from_numpy(v).to(DEV) for k, v in d.items()} for d in pair)
             for pair in synthetic_dropout_masks(78, 2, 33, 101)]
    drs = np.random.RandomState(79)
    dmasks = [torch.from_numpy((drs.random_sample((2, 64)) >= 0.3).astype(np.float32) / np.float32(0.7)).to(DEV)
              for _ in range(6)]
    clean, noisy, pesq = g["clean"].to(DEV), g["noisy"].to(DEV), g["pesq"].to(DEV)
    loss, terms, gan, loss_d = adversarial_train_step(gen, disc, opt_g, opt_d, clean, noisy, pesq, masks=masks,
                                                      disc_masks=dmasks[:3])
    assert _report("generator loss incl. the GAN term", abs(float(loss) - float(g["loss"])) / float(g["loss"])) < GRAD_TOL
    assert _report("gen_loss_GAN", abs(float(gan) - float(g["gan"])) / float(g["gan"])) < GRAD_TOL
    assert _report("discriminator loss", abs(float(loss_d) - float(g["loss_d"])) / float(g["loss_d"])) < 1e-3
    # generator gradients (noise floor of the whole-network gradient: see test_generator_train_step...)
    keys = [k[len("gsmp_"):] for k in g if k.startswith("gsmp_")]
    scale = max(float(g["gl2_" + k]) for k in keys)
    errs = []
    for k in keys:
        got = gen.grads[k].reshape(-1)
        want = g["gsmp_" + k]
        smp = got[torch.from_numpy(sample_indices(got.numel())).to(DEV)].cpu()
        d = float((smp - want).abs().max())
        assert d < 3e-2 * float(want.abs().max()) + 1e-6 * scale, (k, d)
        if float(g["gl2_" + k]) > 1e-4 * scale:
            errs.append(d / float(want.abs().max()))
    groups = {}
    for k in keys:
        want = g["gsmp_" + k]
        if float(g["gl2_" + k]) > 1e-4 * scale:
            got = gen.grads[k].reshape(-1)
            smp = got[torch.from_numpy(sample_indices(got.numel())).to(DEV)].cpu()
            groups.setdefault(k.split(".")[0], []).append(float((smp - want).abs().max()) / float(want.abs().max()))
    for name, v in groups.items():
        _report(f"adversarial step: gradient error, median over the tensors of {name}", float(np.median(v)))
    _report("adversarial step: generator gradients, median error relative to each tensor's max", float(np.median(errs)))
    # Measured with tests/probes/adv_probe4.py: the HIP forward is 1e-6 from torch's, which is enough to put ~1 of the
    # 853 k InstanceNorm outputs in front of a decoder's PReLU on the other side of zero.  That single element changes
    # dL/ds there by O(1) and every gradient UPSTREAM of it by 1e-4 .. 1e-3 of the tensor's max (seen here: the complex
    # decoder's head, so everything but the mask decoder sits at 3.6e-4; the mask decoder, which shares all kernels,
    # at 3.7e-6).  The bar is therefore the kink noise, with the per-module tests at 1e-6 carrying the exactness claim.
    assert float(np.median(errs)) < 2e-3 and min(float(np.median(v)) for v in groups.values()) < 2e-5
    # discriminator gradients: two graphs (D(clean, est), D(clean, clean)) after the generator update moved nothing in D
    worst = 0.0
    for k, got in disc.grads.items():
        e = rel_err(got, g["dgrad_" + k.replace(".", "_")])
        worst = max(worst, e)
        assert e < 2e-3, (k, e)
    _report("adversarial step: worst discriminator gradient", worst)
    # second step: both networks updated
    loss2, _, gan2, _ = adversarial_train_step(gen, disc, opt_g, opt_d, clean, noisy, None, masks=masks,
                                               disc_masks=dmasks[3:])
    assert _report("generator loss after both updates", abs(float(loss2) - float(g["loss2"])) / float(g["loss2"])) < 2e-3
    assert _report("gen_loss_GAN after both updates", abs(float(gan2) - float(g["gan2"])) / float(g["gan2"])) < 2e-2


def test_full_loss_gradient_at_the_network_output_vs_oracle_autograd():
    """dL/d est_real, dL/d est_imag of the FULL generator loss (RI + magnitude + time + 0.05 x metric discriminator) from
    the HIP pieces (cmgan_loss_backward, discriminator backward, cmgan_mag_pair_backward) against autograd through the
    oracle, at T = 33."""
    from cmgan_amd._lib import check
    from cmgan_amd.synth import discriminator_state_dict, synthetic_dropout_masks
    from cmgan_amd.training import DiscriminatorTrain, GeneratorTrain, generator_loss_terms
    from oracle.weights import make_state_dict
    g = load_golden("adversarial_step.npz")
    sd, dsd = make_state_dict(seed=0), discriminator_state_dict(0)
    np_masks = synthetic_dropout_masks(78, 2, 33, 101)
    drs = np.random.RandomState(79)
    dmask = torch.from_numpy((drs.random_sample((2, 64)) >= 0.3).astype(np.float32) / np.float32(0.7))
    want = O.adversarial_generator_gradients(
        sd, dsd, g["clean"], g["noisy"],
        [tuple({k: torch.from_numpy(v) for k, v in d.items()} for d in pair) for pair in np_masks], dmask)
    gen = GeneratorTrain(sd, device=DEV)
    eng = gen.engine
    disc = DiscriminatorTrain(dsd, engine=eng)
    clean, noisy = g["clean"].to(DEV), g["noisy"].to(DEV)
    c = eng.rms_scale(noisy)
    clean_spec = eng.stft_compress(clean, c)
    # the oracle's own network outputs as the operating point: isolates the loss / discriminator path
    er, ei = want["est_real"].to(DEV).contiguous(), want["est_imag"].to(DEV).contiguous()
    audio = eng.uncompress_istft(er, ei)
    B, _, T, F = er.shape
    d_real, d_imag = torch.empty_like(er), torch.empty_like(ei)
    check(eng._h, eng.lib.cmgan_loss_backward(eng._h, er.data_ptr(), ei.data_ptr(), clean_spec.data_ptr(), B, T,
                                              audio.data_ptr(), clean.data_ptr(), 0.1, 0.9, 0.2, d_real.data_ptr(),
                                              d_imag.data_ptr(), eng._stream()))
    xy = disc.pair(clean_spec, er, ei)
    score = disc.forward(xy, dmask.to(DEV), train=True)
    gan, dscore = disc.score_mse(score, None, scale=0.05)
    assert _report("gen_loss_GAN vs oracle", abs(float(gan) - float(want["gan"])) / float(want["gan"])) < GRAD_TOL
    dxy = disc.backward(dscore)
    check(eng._h, eng.lib.cmgan_mag_pair_backward(eng._h, er.data_ptr(), ei.data_ptr(), dxy.data_ptr(), B, T, 1.0,
                                                  d_real.data_ptr(), d_imag.data_ptr(), eng._stream()))
    assert _report("full-loss d_real vs oracle autograd", rel_err(d_real, want["d_real"])) < GRAD_TOL
    assert _report("full-loss d_imag vs oracle autograd", rel_err(d_imag, want["d_imag"])) < GRAD_TOL
    for k, v in disc.buffers.items():
        assert rel_err(v, want["disc_buffers"][k]) < GRAD_TOL, k


def test_graphed_train_step_is_bit_identical_to_the_eager_step():
    """The adversarial step captured as two hipGraphs (forward/backward, optimisers; device-resident AdamW step count)
    and replayed twice leaves exactly the parameters the eager step leaves (dropout off so that both see the same
    arithmetic), and the BatchNorm counters advance per replay."""
    from cmgan_amd.synth import discriminator_state_dict, synthetic_clips
    from cmgan_amd.training import (AdamW, DiscriminatorTrain, GeneratorTrain, GraphedTrainStep,
                                    adversarial_train_step)
    from oracle.weights import make_state_dict
    sd, dsd = make_state_dict(seed=0), discriminator_state_dict(0)
    B, L = 2, 3200
    clean = synthetic_clips(B, L, seed=5).to(DEV)
    noisy = (clean + 0.3 * synthetic_clips(B, L, seed=6).to(DEV)).contiguous()
    pesq = torch.tensor([0.4, 0.7], device=DEV)

    def fresh():
        gen = GeneratorTrain(sd, device=DEV)
        disc = DiscriminatorTrain(dsd, engine=gen.engine)
        return (gen, disc, AdamW(gen.engine, gen.param_bucket, gen.grad_bucket, lr=5e-4),
                AdamW(gen.engine, disc.param_bucket, disc.grad_bucket, lr=1e-3))
    gen, disc, og, od = fresh()
    for _ in range(2):
        want = adversarial_train_step(gen, disc, og, od, clean, noisy, pesq, masks=None, disc_masks=None)
    gen2, disc2, og2, od2 = fresh()
    step = GraphedTrainStep(gen2, og2, B, L, disc2, od2, dropout=False)
    assert og2.t == 0 and gen2.blocks[0].time.conv.num_batches_tracked == 100     # capturing executed nothing
    for _ in range(2):
        got = step(clean, noisy, pesq)
    assert torch.equal(gen.param_bucket.flat, gen2.param_bucket.flat)
    assert torch.equal(disc.param_bucket.flat, disc2.param_bucket.flat)
    assert float(want[0]) == float(got[0]) and float(want[3]) == float(got[3])
    assert og2.t == 2 and od2.t == 2 and gen2.blocks[3].freq.conv.num_batches_tracked == 102
    # labels computed by a callable from THIS step's est_audio, where the reference calls batch_pesq (train.py:156-162):
    # the eager step and the replay (graph A1 -> host callable -> graph A2) see the same audio and agree bit for bit;
    # a step whose labels are missing updates the generator only
    seen = []

    def labels(clean_cut, est_audio):
        seen.append(float(est_audio.abs().sum()))
        return torch.sigmoid(est_audio.abs().mean(1) * 40.0)
    want3 = adversarial_train_step(gen, disc, og, od, clean, noisy, labels, masks=None, disc_masks=None)
    got3 = step(clean, noisy, labels)
    assert seen[0] == seen[1] and float(want3[3]) == float(got3[3])
    assert torch.equal(gen.param_bucket.flat, gen2.param_bucket.flat)
    assert torch.equal(disc.param_bucket.flat, disc2.param_bucket.flat)
    before = disc2.param_bucket.flat.clone()
    got4 = step(clean, noisy, None)
    assert got4[3] is None and od2.t == 3 and og2.t == 4 and torch.equal(before, disc2.param_bucket.flat)
    # masks drawn inside the graph differ from replay to replay
    step_d = GraphedTrainStep(gen2, og2, B, L, disc2, od2)
    l1 = float(step_d(clean, noisy, pesq)[0])
    l2 = float(step_d(clean, noisy, pesq)[0])
    assert l1 != l2


def test_generator_train_step_at_48_khz_vs_oracle_autograd():
    """BASELINE configs[3] shape for the training path: n_fft 1200 / hop 300 (F = 601, F' = 301), one clip of 8 hops:
    outputs, loss terms, the output gradients incl. the ISTFT adjoint at N = 1200, and all parameter gradients against
    autograd through the oracle (dropout masks fixed)."""
    from cmgan_amd.engine import Engine
    from cmgan_amd.synth import synthetic_clips, synthetic_dropout_masks
    from cmgan_amd.training import AdamW, GeneratorTrain, generator_train_step
    from oracle.weights import make_state_dict
    sd = make_state_dict(seed=0, num_features=601)
    B, L = 1, 2400
    T, Fe = L // 300 + 1, 301
    clean = synthetic_clips(B, L, seed=61)
    noisy = clean + 0.3 * synthetic_clips(B, L, seed=62)
    npm = synthetic_dropout_masks(81, B, T, Fe)
    want = O.generator_step_gradients(
        sd, clean, noisy, [tuple({k: torch.from_numpy(v) for k, v in d.items()} for d in pair) for pair in npm],
        n_fft=1200, hop=300)
    gen = GeneratorTrain(sd, engine=Engine(n_fft=1200, hop=300, device=DEV))
    opt = AdamW(gen.engine, gen.param_bucket, gen.grad_bucket, lr=5e-4)
    masks = [tuple({k: torch.from_numpy(v).to(DEV) for k, v in d.items()} for d in pair) for pair in npm]
    eng = gen.engine
    nz = noisy.to(DEV)
    er, ei = gen.forward(eng.stft_compress(nz, eng.rms_scale(nz)), masks)
    assert _report("48 kHz train forward est_real", rel_err(er, want["est_real"])) < GRAD_TOL
    assert _report("48 kHz train forward est_imag", rel_err(ei, want["est_imag"])) < GRAD_TOL
    gen2 = GeneratorTrain(sd, engine=eng)
    opt = AdamW(eng, gen2.param_bucket, gen2.grad_bucket, lr=5e-4)
    loss, terms = generator_train_step(gen2, opt, clean.to(DEV), nz, masks=masks)
    assert _report("48 kHz loss", abs(float(loss) - float(want["loss"])) / float(want["loss"])) < GRAD_TOL
    assert _report("48 kHz loss terms", rel_err(terms[:3], want["terms"])) < GRAD_TOL
    scale = max(float(v.abs().max()) for v in want["grads"].values())
    groups = {}
    for k, v in want["grads"].items():
        den = float(v.abs().max())
        if den > 1e-6 * scale:
            e = float((gen2.grads[k].cpu() - v).abs().max()) / den
            assert e < 3e-2, (k, e)
            groups.setdefault(k.split(".")[0], []).append(e)
    for name, v in groups.items():
        _report(f"48 kHz gradient error, median over the tensors of {name}", float(np.median(v)))
    allv = [e for v in groups.values() for e in v]
    assert float(np.median(allv)) < 2e-3 and min(float(np.median(v)) for v in groups.values()) < 2e-5


def test_trainer_runs_an_epoch_like_the_reference_trainer(tmp_path):
    """`Trainer` (train.py:47-275) end to end on a tiny VCTK-DEMAND-shaped directory: data loader -> device prefetch ->
    adversarial train steps (stub PESQ labels in place of the absent wheel) -> eval-mode validation on the CURRENT
    parameters through the inference kernels -> checkpoint; the checkpoint is the reference's 359-entry generator
    state_dict and loads into the inference model, whose output then equals the oracle's on those weights."""
    import os
    from scipy.io import wavfile
    from cmgan_amd import TSCNet
    from cmgan_amd.data import load_data
    from cmgan_amd.synth import discriminator_state_dict
    from cmgan_amd.training import Trainer, adversarial_train_step
    from oracle.weights import make_state_dict, synthetic_clips
    for split, n in (("train", 4), ("test", 2)):
        for sub in ("clean", "noisy"):
            os.makedirs(tmp_path / split / sub)
        for i in range(n):
            clean = synthetic_clips(1, 3200, seed=90 + i)[0].numpy() * 0.3
            noisy = clean + 0.1 * synthetic_clips(1, clean.size, seed=95 + i)[0].numpy()
            for sub, sig in (("clean", clean), ("noisy", noisy)):
                wavfile.write(str(tmp_path / split / sub / f"p_{i}.wav"), 16000, np.round(sig * 32767).astype(np.int16))
    train_ds, test_ds = load_data(str(tmp_path), batch_size=2, n_cpu=0, cut_len=3200)
    calls = []

    def fake_pesq(clean, est):                       # (PESQ - 1) / 3.5 stand-in: deterministic, on the device
        calls.append(tuple(est.shape))
        return torch.full((clean.shape[0],), 0.6, device=clean.device)
    logs = []
    tr = Trainer(train_ds, test_ds, make_state_dict(seed=0), discriminator_state_dict(0), device=DEV, pesq_fn=fake_pesq,
                 log_interval=1, log=logs.append)
    p0 = tr.gen.param_bucket.flat.clone()
    d0 = tr.disc.param_bucket.flat.clone()
    hist = tr.train(1, save_model_dir=str(tmp_path / "ckpt"))
    assert len(hist) == 1 and np.isfinite(hist[0]) and tr.epoch == 1
    assert tr.optimizer.t == 2 and tr.optimizer_disc.t == 2                  # 4 training clips / batch 2
    assert not torch.equal(p0, tr.gen.param_bucket.flat) and not torch.equal(d0, tr.disc.param_bucket.flat)
    assert len(calls) == 2 + 1 and any("Epoch 0, Step 2" in m for m in logs) and any("Generator loss" in m for m in logs)
    saved = [f for f in os.listdir(tmp_path / "ckpt") if f.startswith("CMGAN_epoch_0_")]
    assert len(saved) == 1
    ck = torch.load(str(tmp_path / "ckpt" / saved[0]))
    assert len(ck) == 359 and int(ck["TSCB_1.time_conformer.conv.net.5.num_batches_tracked"]) == 102
    # the checkpoint drives the inference model; parity of that model with the oracle on the TRAINED weights
    model = TSCNet(64, 201).load_state_dict(ck).eval()
    wav = synthetic_clips(1, 3200, seed=7)
    from cmgan_amd.evaluation import enhance_one_track
    got = enhance_one_track(model, wav.to(DEV))
    want = O.enhance(ck, wav)
    assert _report("trained checkpoint: inference vs oracle", rel_err(got, want)) < 1e-4
    # resume path: save -> one more step -> restore -> the same step again lands on bit-identical parameters, which
    # needs the AdamW moments / step counts, the buffers AND the dropout-mask stream offsets to be carried
    st = tr.resume_state()
    assert st["epoch"] == 1 and all(v > 0 for v in st["mask_rng_offsets"].values())
    clean = (synthetic_clips(2, 3200, seed=70) * 0.3).to(DEV)
    noisy = clean + 0.1 * synthetic_clips(2, 3200, seed=71).to(DEV)
    tr.train_step(clean, noisy)
    p1, d1 = tr.gen.param_bucket.flat.clone(), tr.disc.param_bucket.flat.clone()
    tr.epoch = 7
    tr.load_resume_state(st)
    assert tr.epoch == 1 and tr.optimizer.t == 2 and tr.gen.mask_rng_offsets() == st["mask_rng_offsets"]
    tr.train_step(clean, noisy)
    assert torch.equal(tr.gen.param_bucket.flat, p1) and torch.equal(tr.disc.param_bucket.flat, d1)


def _philox4x32_10_np(ctr_lo, ctr_hi, half, seed):
    """numpy Philox4x32-10 (Salmon et al. 2011; the generator of torch's CUDA dropout) for counters (ctr, half, 0)."""
    M0, M1, W0, W1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57), 0x9E3779B9, 0xBB67AE85
    c = [ctr_lo.astype(np.uint64), ctr_hi.astype(np.uint64), np.full_like(ctr_lo, half, dtype=np.uint64),
         np.zeros_like(ctr_lo, dtype=np.uint64)]
    k0, k1 = seed & 0xffffffff, (seed >> 32) & 0xffffffff
    for _ in range(10):
        p0, p1 = M0 * c[0], M1 * c[2]
        hi0, lo0, hi1, lo1 = p0 >> np.uint64(32), p0 & np.uint64(0xffffffff), p1 >> np.uint64(32), p1 & np.uint64(0xffffffff)
        c = [hi1 ^ c[1] ^ np.uint64(k0), lo1, hi0 ^ c[3] ^ np.uint64(k1), lo0]
        k0, k1 = (k0 + W0) & 0xffffffff, (k1 + W1) & 0xffffffff
    return c


def test_dropout_keep_masks_are_philox_draws_with_a_device_side_offset():
    """cmgan_dropout_masks: bytes = (16-bit uniform < round(keep * 65536)) of Philox4x32-10 at counters offset + group, the
    offset living in device memory - checked bit for bit against a numpy Philox, for the advance of the offset between
    two calls (what makes the draw replayable inside a captured graph), and for the keep rate."""
    from cmgan_amd.engine import Engine
    from cmgan_amd._lib import check
    eng = Engine(device=DEV)
    seed, keep, n = 0x1234567890ABCDEF & ((1 << 63) - 1), 0.8, 16 * 4096
    state = torch.tensor([seed, 7], dtype=torch.int64, device=DEV)
    bufs = []
    for _ in range(2):
        buf = torch.empty(n, dtype=torch.uint8, device=DEV)
        check(eng._h, eng.lib.cmgan_dropout_masks(eng._h, buf.data_ptr(), n, keep, state.data_ptr(), eng._stream()))
        bufs.append(buf.cpu().numpy())
    assert state.cpu().tolist() == [seed, 7 + 2 * n // 16]
    thresh = int(keep * 65536 + 0.5)
    for call, got in enumerate(bufs):
        ctr = np.arange(n // 16, dtype=np.uint64) + np.uint64(7 + call * (n // 16))
        want = np.zeros((n // 16, 16), dtype=np.uint8)
        for half in range(2):
            c = _philox4x32_10_np(ctr & np.uint64(0xffffffff), ctr >> np.uint64(32), half, seed)
            for j in range(2):
                for w, word in enumerate((c[2 * j], c[2 * j + 1])):
                    base = 8 * half + 4 * j + 2 * w
                    want[:, base] = (word & np.uint64(0xffff)) < thresh
                    want[:, base + 1] = (word >> np.uint64(16)) < thresh
        assert np.array_equal(got.reshape(-1, 16), want), call
    rate = float(np.concatenate(bufs).mean())
    assert abs(rate - keep) < 4 * np.sqrt(keep * (1 - keep) / (2 * n))
    assert not np.array_equal(bufs[0], bufs[1])
    with pytest.raises(RuntimeError):
        check(eng._h, eng.lib.cmgan_dropout_masks(eng._h, buf.data_ptr(), 24, keep, state.data_ptr(), eng._stream()))


def test_generator_masks_come_from_the_library_generator_and_restart_with_the_model():
    """GeneratorTrain.masks: one draw for all forty masks; a fresh model with the same generator seed repeats the stream,
    the same model continues it."""
    from cmgan_amd.training import GeneratorTrain
    from oracle.weights import make_state_dict
    sd = make_state_dict(seed=0)
    g1 = GeneratorTrain(sd, device=DEV)
    a = g1.masks(1, 5, torch.Generator(device=DEV).manual_seed(5))
    b = g1.masks(1, 5, torch.Generator(device=DEV).manual_seed(5))
    g2 = GeneratorTrain(sd, engine=g1.engine)
    c = g2.masks(1, 5, torch.Generator(device=DEV).manual_seed(5))
    first = lambda m: m[0][0]["ff1_1"]
    assert torch.equal(first(a), first(c)) and not torch.equal(first(a), first(b))
    assert first(a).dtype == torch.uint8 and set(first(a).unique().tolist()) <= {0, 1}
    assert 0.7 < float(first(a).float().mean()) < 0.9


def test_mask_stream_offsets_survive_a_resume_and_reset_keeps_the_device_state_in_place():
    """`mask_rng_offsets` / `set_mask_rng_offsets` are what `Trainer.resume_state` carries: a model restored to the saved
    offsets draws the masks the original would have drawn next (not the first ones again); `reset_mask_rng` restarts the
    stream at offset 0 WITHOUT replacing the device {seed, offset} tensor a captured graph points at."""
    from cmgan_amd.training import GeneratorTrain
    from oracle.weights import make_state_dict
    sd = make_state_dict(seed=0)
    g1 = GeneratorTrain(sd, device=DEV)
    gen = lambda: torch.Generator(device=DEV).manual_seed(5)
    first = lambda m: m[0][0]["ff1_1"].clone()
    a = first(g1.masks(1, 5, gen()))
    saved = g1.mask_rng_offsets()
    assert len(saved) == 1 and next(iter(saved.values())) > 0
    b = first(g1.masks(1, 5, gen()))                          # what the run draws next
    g2 = GeneratorTrain(sd, engine=g1.engine)                 # the restarted process
    g2.set_mask_rng_offsets(saved)
    assert torch.equal(first(g2.masks(1, 5, gen())), b)
    assert g2.mask_rng_offsets() == g1.mask_rng_offsets()
    state = g1.mask_rng_state(gen())
    ptr = state.data_ptr()
    g1.reset_mask_rng()
    assert g1.mask_rng_state(gen()).data_ptr() == ptr and g1.mask_rng_offsets() == {k: 0 for k in saved}
    assert torch.equal(first(g1.masks(1, 5, gen())), a)       # the stream starts over
    g1.set_mask_rng_offsets(saved)                            # existing stream: restored in place too
    assert g1.mask_rng_state(gen()).data_ptr() == ptr
    assert torch.equal(first(g1.masks(1, 5, gen())), b)
