"""Deterministic random-init generator weights in the reference's state_dict format, and
synthetic noisy clips of the benchmark shape (BASELINE.json: "synthetic 16 kHz noisy clips",
random-init weights of the architecture - there is no network for checkpoints or datasets).

Lives in the package (not under oracle/) so that bench.py's GPU leg and the examples never
import test infrastructure; oracle/weights.py re-exports it for the tests.

The reference checkpoint is absent (``.MISSING_LARGE_BLOBS``), so parity runs on
seeded random weights.  torch's default initialisers are not guaranteed stable
across versions, so the 359 tensors of ``TSCNet(num_channel=64, num_features=F)``
(key names and shapes: /root/reference/src/models/generator.py:159-172,
/root/reference/src/models/conformer.py:75-214) are drawn from
a numpy PCG64 stream instead; the same seed gives the same bytes on any box.

Distributions follow the torch defaults in spirit (uniform +-1/sqrt(fan_in) for
conv/linear, N(0,1) embedding) but norm gains/biases, PReLU slopes and the
BatchNorm running statistics are randomised too, so every affine path, the
BN fold and the per-frequency PReLU are exercised.
"""
from __future__ import annotations

import numpy as np
import torch

C = 64            # num_channel (generator.py:160)
HEADS = 4         # generator.py:78
DIM_HEAD = 16     # generator.py:77
FF_MULT = 4       # conformer.py:189
CONV_EXP = 2      # conformer.py:190
CONV_K = 31       # generator.py:79
MAX_POS = 512     # conformer.py:76


def _to_torch(v: np.ndarray) -> torch.Tensor:
    return torch.tensor(v.item()) if v.ndim == 0 else torch.from_numpy(np.ascontiguousarray(v))


class _Stream:
    def __init__(self, seed: int):
        self.rng = np.random.Generator(np.random.PCG64(seed))

    def uniform(self, shape, bound):
        return self.rng.uniform(-bound, bound, size=shape).astype(np.float32)

    def normal(self, shape, mean=0.0, std=1.0):
        return (mean + std * self.rng.standard_normal(size=shape)).astype(np.float32)


def _conv(s, sd, name, cout, cin, kh, kw):
    bound = 1.0 / np.sqrt(cin * kh * kw)
    sd[name + ".weight"] = s.uniform((cout, cin, kh, kw), bound)
    sd[name + ".bias"] = s.uniform((cout,), bound)


def _linear(s, sd, name, out_f, in_f, bias=True, conv1d=False):
    bound = 1.0 / np.sqrt(in_f)
    shape = (out_f, in_f, 1) if conv1d else (out_f, in_f)
    sd[name + ".weight"] = s.uniform(shape, bound)
    if bias:
        sd[name + ".bias"] = s.uniform((out_f,), bound)


def _norm(s, sd, name, n):
    sd[name + ".weight"] = s.normal((n,), 1.0, 0.1)
    sd[name + ".bias"] = s.normal((n,), 0.0, 0.1)


def _prelu(s, sd, name, n, init=0.25):
    sd[name + ".weight"] = s.normal((n,), init, 0.05)


def _dense_block(s, sd, prefix):
    for i in range(1, 5):
        _conv(s, sd, f"{prefix}.conv{i}", C, C * i, 2, 3)
        _norm(s, sd, f"{prefix}.norm{i}", C)
        _prelu(s, sd, f"{prefix}.prelu{i}", C)


def _conformer(s, sd, p):
    inner = C * CONV_EXP
    for ff in ("ff1", "ff2"):
        _linear(s, sd, f"{p}.{ff}.fn.fn.net.0", C * FF_MULT, C)
        _linear(s, sd, f"{p}.{ff}.fn.fn.net.3", C, C * FF_MULT)
        _norm(s, sd, f"{p}.{ff}.fn.norm", C)
    _linear(s, sd, f"{p}.attn.fn.to_q", HEADS * DIM_HEAD, C, bias=False)
    _linear(s, sd, f"{p}.attn.fn.to_kv", 2 * HEADS * DIM_HEAD, C, bias=False)
    _linear(s, sd, f"{p}.attn.fn.to_out", C, HEADS * DIM_HEAD)
    sd[f"{p}.attn.fn.rel_pos_emb.weight"] = s.normal((2 * MAX_POS + 1, DIM_HEAD), 0.0, 1.0)
    _norm(s, sd, f"{p}.attn.norm", C)
    _norm(s, sd, f"{p}.conv.net.0", C)
    _linear(s, sd, f"{p}.conv.net.2", inner * 2, C, conv1d=True)
    bound = 1.0 / np.sqrt(CONV_K)
    sd[f"{p}.conv.net.4.conv.weight"] = s.uniform((inner, 1, CONV_K), bound)
    sd[f"{p}.conv.net.4.conv.bias"] = s.uniform((inner,), bound)
    _norm(s, sd, f"{p}.conv.net.5", inner)
    sd[f"{p}.conv.net.5.running_mean"] = s.normal((inner,), 0.0, 0.2)
    sd[f"{p}.conv.net.5.running_var"] = s.rng.uniform(0.5, 2.0, size=(inner,)).astype(np.float32)
    sd[f"{p}.conv.net.5.num_batches_tracked"] = np.array(100, dtype=np.int64)
    _linear(s, sd, f"{p}.conv.net.7", C, inner, conv1d=True)
    _norm(s, sd, f"{p}.post_norm", C)


def conformer_state_dict(seed: int = 0) -> dict:
    """State dict of one standalone ``ConformerBlock(dim=64, dim_head=16, heads=4,
    conv_kernel_size=31)`` (keys without a prefix)."""
    s = _Stream(seed)
    sd: dict = {}
    _conformer(s, sd, "X")
    return {k[2:]: _to_torch(v) for k, v in sd.items()}


def make_state_dict(seed: int = 0, num_features: int = 201) -> dict:
    """All 359 entries of ``TSCNet(64, num_features).state_dict()``."""
    s = _Stream(seed)
    sd: dict = {}
    _conv(s, sd, "dense_encoder.conv_1.0", C, 3, 1, 1)
    _norm(s, sd, "dense_encoder.conv_1.1", C)
    _prelu(s, sd, "dense_encoder.conv_1.2", C)
    _dense_block(s, sd, "dense_encoder.dilated_dense")
    _conv(s, sd, "dense_encoder.conv_2.0", C, C, 1, 3)
    _norm(s, sd, "dense_encoder.conv_2.1", C)
    _prelu(s, sd, "dense_encoder.conv_2.2", C)
    for b in range(1, 5):
        for axis in ("time", "freq"):
            _conformer(s, sd, f"TSCB_{b}.{axis}_conformer")
    _dense_block(s, sd, "mask_decoder.dense_block")
    _conv(s, sd, "mask_decoder.sub_pixel.conv", 2 * C, C, 1, 3)
    _conv(s, sd, "mask_decoder.conv_1", 1, C, 1, 2)
    _norm(s, sd, "mask_decoder.norm", 1)
    _prelu(s, sd, "mask_decoder.prelu", 1)
    _conv(s, sd, "mask_decoder.final_conv", 1, 1, 1, 1)
    _prelu(s, sd, "mask_decoder.prelu_out", num_features, init=-0.25)
    _dense_block(s, sd, "complex_decoder.dense_block")
    _conv(s, sd, "complex_decoder.sub_pixel.conv", 2 * C, C, 1, 3)
    _prelu(s, sd, "complex_decoder.prelu", C)
    _norm(s, sd, "complex_decoder.norm", C)
    _conv(s, sd, "complex_decoder.conv", 2, C, 1, 2)
    return {k: _to_torch(v) for k, v in sd.items()}


def synthetic_clips(batch: int, length: int, seed: int = 0) -> torch.Tensor:
    """``0.1 * N(0,1)`` noisy clips, float32 [batch, length]."""
    rng = np.random.Generator(np.random.PCG64(1000 + seed))
    return torch.from_numpy((0.1 * rng.standard_normal((batch, length))).astype(np.float32))


_SHAPES: dict = {}


def state_dict_shapes(num_features: int = 201) -> dict:
    """{key: shape} of all 359 entries of ``TSCNet(64, num_features).state_dict()``; what
    ``cmgan_amd.TSCNet.load_state_dict`` validates an incoming checkpoint against."""
    if num_features not in _SHAPES:
        _SHAPES[num_features] = {k: tuple(v.shape) for k, v in make_state_dict(0, num_features).items()}
    return _SHAPES[num_features]


def synthetic_dropout_masks(seed: int, B: int, T: int, Fe: int, p: float = 0.2, blocks: int = 4) -> list:
    """Deterministic keep-masks (numpy float32, entries 0 or 1/(1-p)) for every Dropout of `blocks` two-stage conformer
    blocks on a [B, T, Fe] grid: [(time, freq)] * blocks, each a dict ff1_1 / ff1_2 / attn / ff2_1 / ff2_2 of
    [N, L, C] arrays (time: N = B Fe, L = T; freq: N = B T, L = Fe).  numpy's legacy RandomState stream is frozen,
    so the fixture generator and the tests draw the same masks without storing them."""
    rs = np.random.RandomState(seed)
    out = []
    for _ in range(blocks):
        pair = []
        for n, l in ((B * Fe, T), (B * T, Fe)):
            d = {}
            for name, c in (("ff1_1", 256), ("ff1_2", 64), ("attn", 64), ("ff2_1", 256), ("ff2_2", 64)):
                d[name] = ((rs.random_sample((n, l, c)) >= p).astype(np.float32) / np.float32(1.0 - p)).astype(np.float32)
            pair.append(d)
        out.append(tuple(pair))
    return out


def kink_free_twin(sd: dict, eps: float = 1e-3) -> dict:
    """The same generator state dict with every PReLU slope moved to 1 - eps * (i / n) (channel i of n): PReLU is then
    linear up to eps, so the loss is a smooth function of every activation and the whole-network gradient is well
    defined to rounding - with the default slopes (0.25) a single InstanceNorm output that lands on the other side of
    zero changes every upstream gradient by 1e-4 .. 1e-2 of its maximum, in the reference's own fp32 autograd as in any
    re-implementation (tests/test_oracle_golden.py measures both).  Used by the whole-step parity tests, which can
    then hold EVERY gradient tensor of the full pipeline to the 1e-3 gate; the PReLU derivative itself is pinned by
    the per-module tests on the default slopes.  The slopes differ per channel, so their indexing stays observable."""
    import torch
    out = dict(sd)
    for k, v in sd.items():
        if ("prelu" in k and k.endswith(".weight")) or k.endswith("conv_1.2.weight") or k.endswith("conv_2.2.weight"):
            n = v.numel()
            out[k] = (1.0 - eps * torch.arange(n, dtype=torch.float64) / n).to(v.dtype).reshape(v.shape)
    return out


def sample_indices(numel: int, k: int = 256, seed: int = 1234) -> np.ndarray:
    """k fixed flat indices into a tensor of `numel` elements (all of them when numel <= k)."""
    if numel <= k:
        return np.arange(numel)
    return np.sort(np.random.RandomState(seed + numel % 9973).choice(numel, k, replace=False))


def discriminator_state_dict(seed: int = 0, ndf: int = 16) -> dict:
    """State dict of the reference metric discriminator `Discriminator(ndf=16)` (src/models/discriminator.py:29-64):
    four spectral-norm 4x4 stride-2 convs (weight_orig + the power-iteration vectors weight_u / weight_v, both unit
    norm like torch initialises them), InstanceNorm2d(affine) + PReLU after each, two spectral-norm Linear layers,
    PReLU, LearnableSigmoid.  Keys as `Discriminator.state_dict()` returns them (34 entries)."""
    s = _Stream(seed + 7919)
    sd: dict = {}

    def unit(n):
        v = s.normal((n,), 0.0, 1.0).astype(np.float64)
        return (v / max(float(np.linalg.norm(v)), 1e-12)).astype(np.float32)

    chans = (2, ndf, 2 * ndf, 4 * ndf, 8 * ndf)
    for i in range(4):
        cin, cout = chans[i], chans[i + 1]
        bound = 1.0 / np.sqrt(cin * 16)
        sd[f"layers.{3 * i}.weight_orig"] = s.uniform((cout, cin, 4, 4), bound)
        sd[f"layers.{3 * i}.weight_u"] = unit(cout)
        sd[f"layers.{3 * i}.weight_v"] = unit(cin * 16)
        _norm(s, sd, f"layers.{3 * i + 1}", cout)
        _prelu(s, sd, f"layers.{3 * i + 2}", cout)
    for idx, (out_f, in_f) in ((14, (4 * ndf, 8 * ndf)), (17, (1, 4 * ndf))):
        bound = 1.0 / np.sqrt(in_f)
        sd[f"layers.{idx}.weight_orig"] = s.uniform((out_f, in_f), bound)
        sd[f"layers.{idx}.bias"] = s.uniform((out_f,), bound)
        sd[f"layers.{idx}.weight_u"] = unit(out_f)
        sd[f"layers.{idx}.weight_v"] = unit(in_f)
    _prelu(s, sd, "layers.16", 4 * ndf)
    sd["layers.18.slope"] = s.normal((1,), 1.0, 0.05)
    return {k: _to_torch(v) for k, v in sd.items()}
