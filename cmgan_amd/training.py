"""The reference's training step (src/train.py:72-205) on the HIP path: every module of the generator
and the metric discriminator with forward + backward, the losses and their gradients, AdamW and the data-parallel
gradient mean.  Not here: the epoch loop, logging, checkpoint files and `batch_pesq` (CPU code of the absent `pesq`
wheel - PESQ labels are an input).

* `generator_loss_terms`   - the non-adversarial terms of Trainer.calculate_generator_loss (train.py:124-151) as one
                             deterministic device reduction; the per-rank scalars a data-parallel step all-reduces.
* `forward_generator_step` - Trainer.forward_generator_step (train.py:72-122) in eval mode: per-row RMS scale, STFT +
                             power compression of the noisy AND the clean batch, TSCNet, inverse transform.
* `validation_step`        - the generator half of Trainer.test_step (train.py:207-220): the above + the loss terms
                             + ONE all-reduce of the scalars across ranks (RCCL on the GPU box).
* `AdamW`, `step_lr`       - torch.optim.AdamW (train.py:63-66) as one HIP launch over a flat parameter bucket, and the
                             StepLR(30, 0.5) schedule (train.py:248-253); `FeedForwardTrain.allreduce_gradients()` is
                             the data-parallel gradient mean as ONE all-reduce over the same bucket (train.py:192).
* `DenseBlockTrain`        - `DilatedDenseNet` (generator.py:6-47; encoder and both decoders) in train mode on
                             channels-last [B, T, F, 64]: dilated (2,3) convs over slot views (no concat), InstanceNorm2d,
                             PReLU; forward + full backward (twenty parameter gradients).
* `TSCBTrain`              - one two-stage conformer block of the generator (generator.py:72-99) in TRAIN mode on
                             channels-last activations [B, T, F', 64]: time conformer per (b, f'), frequency conformer
                             per (b, t), both residuals, the layout flips as a HIP kernel.
* `ConformerBlockTrain`    - the WHOLE reference ConformerBlock (conformer.py:182-222) in TRAIN mode from the pieces below
                             plus HIP residual adds and the post_norm LayerNorm: forward, backward, all 31 parameter
                             gradients in ONE flat bucket (one all-reduce, one AdamW launch).
* `AttentionTrain`         - PreNorm(Attention) (conformer.py:54-72, 75-133) in TRAIN mode: Shaw relative-position
                             attention with the output Dropout as a keep-mask, forward + full backward incl. the
                             relative-position embedding gradient.
* `ConvModuleTrain`        - the ConformerConvModule (conformer.py:151-176) in TRAIN mode: BatchNorm1d on batch
                             statistics with running-stat update, forward + full backward (ten parameter gradients).
* `FeedForwardTrain`       - a ConformerBlock's `Scale(0.5, PreNorm(dim, FeedForward(dim, mult=4, dropout)))` branch
                             (conformer.py:54-72, 136-148, 211-212) in TRAIN mode: forward with the two Dropout layers
                             as explicit keep-masks, and the full backward (dL/dx and all six parameter gradients).

* `DenseEncoderTrain`,
  `DecoderTrain`           - DenseEncoder (generator.py:50-69), MaskDecoder / ComplexDecoder (generator.py:121-156) in
                             train mode: strided / sub-pixel row convs on the fp32 MFMA chain, the (1,2) tail convs,
                             InstanceNorm2d + PReLU heads, forward + full backward.
* `GeneratorTrain`         - the WHOLE TSCNet (generator.py:159-201) in train mode from the pieces above: forward,
                             backward, every learnable parameter and its gradient a view of ONE flat bucket.
* `generator_train_step`   - one optimisation step of the generator on the non-adversarial loss (train.py:72-151,
                             185-193 without the metric-discriminator term): STFT, forward, ISTFT, loss, loss gradient
                             (incl. the ISTFT adjoint), backward, ONE gradient all-reduce, ONE AdamW launch.

* `DiscriminatorTrain`     - the metric discriminator (src/models/discriminator.py:29-64) with spectral-norm power
                             iteration, forward + backward (csrc/disc.hip).
* `adversarial_train_step` - Trainer.train_step (train.py:173-205): the generator step incl. the 0.05 * gen_loss_GAN
                             term through the discriminator, then the discriminator step on GIVEN PESQ labels
                             (batch_pesq itself is CPU code from the absent `pesq` wheel).

Everything numerical runs in libcmgan_hip (csrc/train.hip, csrc/disc.hip); this module only owns parameter tensors,
draws the dropout masks with torch's generator (plumbing) and passes pointers.
"""
from __future__ import annotations

import ctypes
from typing import Dict, Optional, Tuple

import torch

from ._lib import (AttnParams, ConvModParams, DecoderParams, DenseParams, DiscParams, EncoderParams, FfnParams,
                   check)
from .dist import FlatBucket, all_agree, allreduce_mean, broadcast_from_rank0, get_rank
from .engine import Engine

__all__ = ["Trainer", "batch_pesq", "GraphedTrainStep", "DiscriminatorTrain", "adversarial_train_step", "GeneratorTrain", "generator_train_step", "DenseEncoderTrain", "DecoderTrain", "DenseBlockTrain", "TSCBTrain", "ConformerBlockTrain", "FeedForwardTrain", "ConvModuleTrain", "AttentionTrain", "AdamW", "step_lr", "generator_loss_terms", "dropout_mask", "forward_generator_step",
           "validation_step"]

_KEYS = ("fn.norm.weight", "fn.norm.bias", "fn.fn.net.0.weight", "fn.fn.net.0.bias",
         "fn.fn.net.3.weight", "fn.fn.net.3.bias")
_FIELDS = ("ln_weight", "ln_bias", "w1", "b1", "w2", "b2")
_SHAPES = ((64,), (64,), (256, 64), (256,), (64, 256), (64,))


def dropout_mask(shape, p: float, device, generator: Optional[torch.Generator] = None) -> Optional[torch.Tensor]:
    """Keep-mask of nn.Dropout(p) in train mode as BYTES: Bernoulli(1 - p) flags, uint8 (None when p == 0).  The kernels
    multiply kept values by 1 / (1 - p) themselves; bytes because the masks are the step's largest per-token traffic."""
    if p <= 0.0:
        return None
    return torch.empty(shape, dtype=torch.uint8, device=device).bernoulli_(1.0 - p, generator=generator)


def _same_shape(eng, t: Optional[torch.Tensor], like: torch.Tensor, name: str) -> Optional[int]:
    """Device pointer of the optional fused residual `t` (must have `like`'s element count, fp32, contiguous) or None."""
    if t is None:
        return None
    if t.numel() != like.numel() or not t.is_contiguous():
        raise ValueError(f"{name}: expected a contiguous tensor of {tuple(like.shape)}")
    return eng._in(t, name).data_ptr()


def _keep_mask(mask: Optional[torch.Tensor], shape, p: float, device, name: str) -> Optional[torch.Tensor]:
    """The byte keep-mask the kernels take, from either form a caller may hold: uint8 / bool keep flags, or the float
    mask of F.dropout's arithmetic (entries 0 or 1 / (1 - p), as the reference-derived fixtures store them)."""
    if mask is None:
        return None
    if mask.device != torch.device(device):
        raise ValueError(f"{name} is on {mask.device}, the engine on {device}")
    if mask.is_floating_point():
        if p <= 0.0:
            raise ValueError(f"{name}: a float dropout mask needs the module's dropout probability to be > 0")
        kept = mask[mask != 0]
        if kept.numel() and float((kept - 1.0 / (1.0 - p)).abs().max()) > 1e-5:
            raise ValueError(f"{name}: float mask entries must be 0 or 1/(1-p) = {1.0 / (1.0 - p):.6f}")
        mask = mask != 0
    return mask.reshape(shape).to(torch.uint8).contiguous()


def generator_loss_terms(engine: Engine, est_real, est_imag, clean_spec, est_audio, clean_audio,
                         loss_weights=(0.1, 0.9, 0.2)) -> Tuple[torch.Tensor, torch.Tensor]:
    """(weighted loss without the GAN term, float32[4] = {loss_ri, loss_mag, time_loss, time_mse}) on the device.
    loss = w0 loss_ri + w1 loss_mag + w2 time_loss   (train.py:143-148; default weights train.py:28)."""
    terms = engine.loss_terms(est_real, est_imag, clean_spec, est_audio, clean_audio)
    return (terms * _loss_w(loss_weights, terms.device)).sum(), terms


_LOSS_W: Dict[tuple, torch.Tensor] = {}


def _loss_w(loss_weights, device) -> torch.Tensor:
    """[w_ri, w_mag, w_time, 0] on the device, created once per (weights, device): no host -> device copy inside a
    step, which keeps the step graph-capturable."""
    key = (tuple(float(x) for x in loss_weights[:3]), str(device))
    w = _LOSS_W.get(key)
    if w is None:
        w = _LOSS_W[key] = torch.tensor(list(key[0]) + [0.0], dtype=torch.float32, device=device)
    return w


def _buckets(shapes: Dict[str, tuple], state, views, device):
    """(param_bucket, grad_bucket, params, grads): own FlatBuckets loaded from `state`, or the given views."""
    if views is not None:
        params, grads = views
        for k, shp in shapes.items():
            if tuple(params[k].shape) != tuple(shp) or tuple(grads[k].shape) != tuple(shp):
                raise ValueError(f"{k}: view shape {tuple(params[k].shape)}, expected {tuple(shp)}")
        return None, None, params, grads
    for k, shp in shapes.items():
        if tuple(state[k].shape) != tuple(shp):
            raise ValueError(f"{k}: shape {tuple(state[k].shape)}, expected {tuple(shp)}")
    pb = FlatBucket(shapes, device).load({k: state[k].detach().to(device, torch.float32) for k in shapes})
    gb = FlatBucket(shapes, device)
    return pb, gb, pb.views, gb.views


def _subviews(views, prefix: str):
    """The (params, grads) view dictionaries of the sub-module `prefix`, keys relative to it."""
    n = len(prefix) + 1
    params, grads = views
    return ({k[n:]: v for k, v in params.items() if k.startswith(prefix + ".")},
            {k[n:]: v for k, v in grads.items() if k.startswith(prefix + ".")})


class FeedForwardTrain:
    """ff1 / ff2 branch of one ConformerBlock in train mode on the HIP kernels.

    `state` holds the branch's six tensors under the reference's key names relative to `ff1.` / `ff2.`
    (`fn.norm.weight`, `fn.fn.net.0.weight`, ...), e.g. a slice of `TSCNet.state_dict()`."""

    SHAPES = dict(zip(_KEYS, _SHAPES))

    def __init__(self, state: Optional[Dict[str, torch.Tensor]] = None, dropout: float = 0.2,
                 engine: Optional[Engine] = None, device=None, views=None):
        """`views` = (params, grads): dicts of tensors that are views of a larger FlatBucket (ConformerBlockTrain)."""
        self.engine = engine if engine is not None else Engine(device=device)
        self.p = float(dropout)
        # parameters and gradients live in two flat buckets (views per tensor): one collective, one optimiser launch
        self.param_bucket, self.grad_bucket, self.params, self.grads = _buckets(self.SHAPES, state, views,
                                                                                self.engine.device)
        self._ws: Optional[torch.Tensor] = None

    def _struct(self, tensors: Dict[str, torch.Tensor]) -> FfnParams:
        s = FfnParams()
        for key, field in zip(_KEYS, _FIELDS):
            setattr(s, field, tensors[key].data_ptr())
        return s

    def _workspace(self, M: int) -> torch.Tensor:
        need = self.engine.lib.cmgan_ffn_train_workspace_bytes(self.engine._h, M)
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty(need, dtype=torch.uint8, device=self.engine.device)
        return self._ws

    def _scale(self) -> float:
        return 1.0 / (1.0 - self.p) if self.p > 0.0 else 1.0

    def masks(self, M: int, generator: Optional[torch.Generator] = None):
        """Fresh keep-masks for the two Dropout layers (conformer.py:142,144)."""
        dev = self.engine.device
        return dropout_mask((M, 256), self.p, dev, generator), dropout_mask((M, 64), self.p, dev, generator)

    def forward(self, x: torch.Tensor, mask1: Optional[torch.Tensor] = None,
                mask2: Optional[torch.Tensor] = None, residual: Optional[torch.Tensor] = None) -> torch.Tensor:
        """x [..., 64] -> 0.5 * FeedForward(LayerNorm(x)) with the given dropout masks, plus `residual` (same shape)
        when given - the `+ x` of conformer.py:216 fused into the kernel's final store."""
        eng = self.engine
        shape = x.shape
        x2 = eng._in(x.reshape(-1, 64), "x")
        M = x2.size(0)
        r = _same_shape(eng, residual, x2, "residual")
        m1 = _keep_mask(mask1, (M, 256), self.p, eng.device, "mask1")
        m2 = _keep_mask(mask2, (M, 64), self.p, eng.device, "mask2")
        y = torch.empty_like(x2)
        ws = self._workspace(M)
        p = self._struct(self.params)
        with torch.cuda.device(eng.device):
            check(eng._h, eng.lib.cmgan_ffn_train_forward(
                eng._h, x2.data_ptr(), M, ctypes.byref(p), m1.data_ptr() if m1 is not None else None,
                m2.data_ptr() if m2 is not None else None, self._scale(), r, y.data_ptr(), ws.data_ptr(), ws.numel(),
                eng._stream()))
        return y.reshape(shape)

    def backward(self, x: torch.Tensor, dy: torch.Tensor, mask1: Optional[torch.Tensor] = None,
                 mask2: Optional[torch.Tensor] = None, dresidual: Optional[torch.Tensor] = None
                 ) -> Tuple[torch.Tensor, Dict[str, torch.Tensor]]:
        """(dL/dx of the branch [+ dresidual], {key: dL/dparam}) for upstream gradient dy; masks must be the forward's.
        `dresidual=dy` gives the gradient of `ff(x) + x` in one pass."""
        eng = self.engine
        shape = x.shape
        x2, dy2 = eng._in(x.reshape(-1, 64), "x"), eng._in(dy.reshape(-1, 64), "dy")
        M = x2.size(0)
        r = _same_shape(eng, dresidual, x2, "dresidual")
        m1 = _keep_mask(mask1, (M, 256), self.p, eng.device, "mask1")
        m2 = _keep_mask(mask2, (M, 64), self.p, eng.device, "mask2")
        dx = torch.empty_like(x2)
        ws = self._workspace(M)
        p, g = self._struct(self.params), self._struct(self.grads)
        with torch.cuda.device(eng.device):
            check(eng._h, eng.lib.cmgan_ffn_train_backward(
                eng._h, x2.data_ptr(), dy2.data_ptr(), M, ctypes.byref(p), m1.data_ptr() if m1 is not None else None,
                m2.data_ptr() if m2 is not None else None, self._scale(), r, dx.data_ptr(), ctypes.byref(g),
                ws.data_ptr(), ws.numel(), eng._stream()))
        return dx.reshape(shape), self.grads

    def allreduce_gradients(self) -> torch.Tensor:
        """Mean of the six gradients over the data-parallel ranks: ONE all-reduce (RCCL) over the flat bucket."""
        if self.grad_bucket is None:
            raise RuntimeError("this module's gradients are views of its parent's bucket: all-reduce the parent")
        return allreduce_mean(self.grad_bucket.flat)


def step_lr(epoch: int, init_lr: float = 5e-4, decay_epoch: int = 30, gamma: float = 0.5) -> float:
    """torch.optim.lr_scheduler.StepLR(optimizer, step_size=decay_epoch, gamma=0.5) stepped once per epoch
    (src/train.py:17-19, 248-253, 275): the learning rate in force during `epoch` (0-based)."""
    return init_lr * gamma ** (epoch // decay_epoch)


class AdamW:
    """torch.optim.AdamW(params, lr=init_lr) of src/train.py:63 on a FlatBucket pair: one launch updates every
    parameter of the bucket.  Defaults are torch's (betas (0.9, 0.999), eps 1e-8, weight_decay 1e-2).  The learning
    rate and the update count live in device memory (`state`), so the launch carries no step-dependent host scalars
    and a captured training step can be replayed (`GraphedTrainStep`); `set_lr` is the schedule's hook."""

    def __init__(self, engine: Engine, params: FlatBucket, grads: FlatBucket, lr: float = 5e-4,
                 betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 1e-2):
        if params.numel != grads.numel:
            raise ValueError("parameter and gradient buckets differ in size")
        self.engine, self.params, self.grads = engine, params, grads
        self.lr, self.betas, self.eps, self.weight_decay = float(lr), betas, eps, weight_decay
        self.exp_avg = torch.zeros_like(params.flat)
        self.exp_avg_sq = torch.zeros_like(params.flat)
        self.state = torch.tensor([self.lr, 0.0], dtype=torch.float32, device=engine.device)   # lr, update count

    @property
    def t(self) -> int:
        return int(self.state[1].item())

    def set_lr(self, lr: float):
        """Not capturable (host -> device copy): call it between replays, like StepLR.step() between epochs."""
        if float(lr) != self.lr:
            self.lr = float(lr)
            self.state[0:1].copy_(torch.tensor([self.lr], dtype=torch.float32))

    def step(self, lr: Optional[float] = None):
        eng = self.engine
        if lr is not None:
            self.set_lr(lr)
        p, g = eng._in(self.params.flat, "params"), eng._in(self.grads.flat, "grads")
        with torch.cuda.device(eng.device):
            check(eng._h, eng.lib.cmgan_adamw_step_dev(eng._h, p.data_ptr(), g.data_ptr(), self.exp_avg.data_ptr(),
                                                       self.exp_avg_sq.data_ptr(), self.params.numel,
                                                       self.state.data_ptr(), float(self.betas[0]), float(self.betas[1]),
                                                       float(self.eps), float(self.weight_decay), eng._stream()))


@torch.no_grad()
def forward_generator_step(model, clean: torch.Tensor, noisy: torch.Tensor) -> Dict[str, torch.Tensor]:
    """Trainer.forward_generator_step (src/train.py:72-122), eval mode.  clean, noisy: float32 [B, L] on the GPU
    (L a multiple of hop).  Everything is a HIP kernel; the x c scaling is fused into the STFT."""
    eng = model.engine
    c = eng.rms_scale(noisy)                                    # train.py:75: c of the NOISY rows scales both
    noisy_spec = eng.stft_compress(noisy, c)                    # train.py:76-94, 95 (model layout [B,2,T,F])
    clean_spec = eng.stft_compress(clean, c)                    # train.py:88-98
    est_real, est_imag = model(noisy_spec)                      # train.py:99
    est_audio = eng.uncompress_istft(est_real, est_imag)        # train.py:104-112 (stays in the scaled domain)
    return {"est_real": est_real, "est_imag": est_imag, "clean_spec": clean_spec, "est_audio": est_audio,
            "clean": clean}                                     # train.py:218: the RAW clean batch


@torch.no_grad()
def validation_step(model, clean: torch.Tensor, noisy: torch.Tensor, loss_weights=(0.1, 0.9, 0.2),
                    reduce: bool = True) -> Tuple[torch.Tensor, torch.Tensor]:
    """Generator half of Trainer.test_step (train.py:207-220): (weighted loss, float32[4] terms), summed over ranks
    with ONE all-reduce when a process group is initialised (divide by world size for the mean).  The adversarial
    term needs the metric discriminator (src/models/discriminator.py), which is outside this build."""
    from . import dist as cdist
    out = forward_generator_step(model, clean, noisy)
    loss, terms = generator_loss_terms(model.engine, out["est_real"], out["est_imag"], out["clean_spec"],
                                       out["est_audio"], out["clean"], loss_weights)
    if reduce:
        packed = torch.cat([loss.reshape(1), terms])
        cdist.allreduce_scalars(packed)
        loss, terms = packed[0], packed[1:]
    return loss, terms


_CM_KEYS = ("net.0.weight", "net.0.bias", "net.2.weight", "net.2.bias", "net.4.conv.weight", "net.4.conv.bias",
            "net.5.weight", "net.5.bias", "net.7.weight", "net.7.bias")
_CM_FIELDS = ("ln_weight", "ln_bias", "pw1_weight", "pw1_bias", "dw_weight", "dw_bias", "bn_weight", "bn_bias",
              "pw2_weight", "pw2_bias")
_CM_SHAPES = ((64,), (64,), (256, 64, 1), (256,), (128, 1, 31), (128,), (128,), (128,), (64, 128, 1), (64,))


class ConvModuleTrain:
    """`conv` branch of one ConformerBlock in train mode on the HIP kernels (csrc/train.hip).

    `state` holds the module's tensors under the reference's key names relative to `conv.` (`net.0.weight`,
    `net.4.conv.weight`, `net.5.running_mean`, ...).  BatchNorm1d uses the statistics of the batch and updates
    `running_mean` / `running_var` like torch (momentum 0.1).  `backward` must follow the `forward` of the same x."""

    SHAPES = dict(zip(_CM_KEYS, _CM_SHAPES))

    def __init__(self, state: Dict[str, torch.Tensor], engine: Optional[Engine] = None, device=None, views=None):
        self.engine = engine if engine is not None else Engine(device=device)
        dev = self.engine.device
        self.param_bucket, self.grad_bucket, self.params, self.grads = _buckets(self.SHAPES, state, views, dev)
        self.running_mean = state["net.5.running_mean"].detach().to(dev, torch.float32).clone()
        self.running_var = state["net.5.running_var"].detach().to(dev, torch.float32).clone()
        self.num_batches_tracked = int(state.get("net.5.num_batches_tracked", 0))
        self._ws: Optional[torch.Tensor] = None
        self._shape = None

    def _struct(self, tensors) -> ConvModParams:
        s = ConvModParams()
        for key, field in zip(_CM_KEYS, _CM_FIELDS):
            setattr(s, field, tensors[key].data_ptr())
        return s

    def _workspace(self, N: int, L: int) -> torch.Tensor:
        need = self.engine.lib.cmgan_convmod_train_workspace_bytes(self.engine._h, N, L)
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty(need, dtype=torch.uint8, device=self.engine.device)
        return self._ws

    def forward(self, x: torch.Tensor, update_running_stats: bool = True,
                residual: Optional[torch.Tensor] = None) -> torch.Tensor:
        """x [N, L, 64] -> ConformerConvModule(x) in train mode [+ residual, fused: conformer.py:218]."""
        eng = self.engine
        x = eng._in(x, "x")
        r = _same_shape(eng, residual, x, "residual")
        N, L, C = x.shape
        if C != 64:
            raise ValueError("conformer dim must be 64")
        ws = self._workspace(N, L)
        y = torch.empty_like(x)
        p = self._struct(self.params)
        rm = self.running_mean.data_ptr() if update_running_stats else None
        rv = self.running_var.data_ptr() if update_running_stats else None
        with torch.cuda.device(eng.device):
            check(eng._h, eng.lib.cmgan_convmod_train_forward(eng._h, x.data_ptr(), N, L, ctypes.byref(p), rm, rv,
                                                              r, y.data_ptr(), ws.data_ptr(), ws.numel(), eng._stream()))
        self._shape = (N, L)
        if update_running_stats:
            self.num_batches_tracked += 1
        return y

    def backward(self, x: torch.Tensor, dy: torch.Tensor, dresidual: Optional[torch.Tensor] = None
                 ) -> Tuple[torch.Tensor, Dict[str, torch.Tensor]]:
        """(dL/dx [+ dresidual], {key: dL/dparam}) for upstream gradient dy; uses the activations the last forward
        left behind."""
        eng = self.engine
        x, dy = eng._in(x, "x"), eng._in(dy, "dy")
        r = _same_shape(eng, dresidual, x, "dresidual")
        N, L, _ = x.shape
        if self._shape != (N, L) or dy.shape != x.shape:
            raise RuntimeError("backward() needs the forward() of the same [N, L, 64] input first")
        ws = self._workspace(N, L)
        dx = torch.empty_like(x)
        p, g = self._struct(self.params), self._struct(self.grads)
        with torch.cuda.device(eng.device):
            check(eng._h, eng.lib.cmgan_convmod_train_backward(eng._h, x.data_ptr(), dy.data_ptr(), N, L, ctypes.byref(p),
                                                               r, dx.data_ptr(), ctypes.byref(g), ws.data_ptr(),
                                                               ws.numel(), eng._stream()))
        return dx, self.grads

    def allreduce_gradients(self) -> torch.Tensor:
        return allreduce_mean(self.grad_bucket.flat)


_AT_KEYS = ("norm.weight", "norm.bias", "fn.to_q.weight", "fn.to_kv.weight", "fn.to_out.weight", "fn.to_out.bias",
            "fn.rel_pos_emb.weight")
_AT_FIELDS = ("ln_weight", "ln_bias", "to_q_weight", "to_kv_weight", "to_out_weight", "to_out_bias", "rel_pos_emb")


class AttentionTrain:
    """`attn` branch (PreNorm + Attention) of one ConformerBlock in train mode on the HIP kernels (csrc/train.hip).
    `state` holds the branch's tensors under the reference's key names relative to `attn.`.  Sequences of up to 4096
    positions; `backward` must follow the `forward` of the same x (it reads q|k|v, O and the row log-sum-exp)."""

    @staticmethod
    def shapes(max_pos_emb: int = 512) -> Dict[str, tuple]:
        return {"norm.weight": (64,), "norm.bias": (64,), "fn.to_q.weight": (64, 64), "fn.to_kv.weight": (128, 64),
                "fn.to_out.weight": (64, 64), "fn.to_out.bias": (64,),
                "fn.rel_pos_emb.weight": (2 * max_pos_emb + 1, 16)}

    def __init__(self, state: Optional[Dict[str, torch.Tensor]] = None, dropout: float = 0.2,
                 engine: Optional[Engine] = None, device=None, views=None):
        self.engine = engine if engine is not None else Engine(device=device)
        self.p = float(dropout)
        self.param_bucket, self.grad_bucket, self.params, self.grads = _buckets(
            self.shapes(self.engine.cfg.max_pos_emb), state, views, self.engine.device)
        self._ws: Optional[torch.Tensor] = None
        self._shape = None

    def _struct(self, tensors) -> AttnParams:
        s = AttnParams()
        for key, field in zip(_AT_KEYS, _AT_FIELDS):
            setattr(s, field, tensors[key].data_ptr())
        return s

    def _workspace(self, N: int, L: int) -> torch.Tensor:
        need = self.engine.lib.cmgan_attn_train_workspace_bytes(self.engine._h, N, L)
        if need == 0:
            raise ValueError(f"unsupported attention shape N={N}, L={L} (L <= 4096)")
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty(need, dtype=torch.uint8, device=self.engine.device)
        return self._ws

    def mask(self, N: int, L: int, generator: Optional[torch.Generator] = None):
        return dropout_mask((N, L, 64), self.p, self.engine.device, generator)

    def forward(self, x: torch.Tensor, mask: Optional[torch.Tensor] = None,
                residual: Optional[torch.Tensor] = None) -> torch.Tensor:
        eng = self.engine
        x = eng._in(x, "x")
        r = _same_shape(eng, residual, x, "residual")
        N, L, C = x.shape
        if C != 64:
            raise ValueError("conformer dim must be 64")
        m = _keep_mask(mask, (N, L, 64), self.p, eng.device, "mask")
        ws = self._workspace(N, L)
        y = torch.empty_like(x)
        p = self._struct(self.params)
        scale = 1.0 / (1.0 - self.p) if self.p > 0.0 else 1.0
        with torch.cuda.device(eng.device):
            check(eng._h, eng.lib.cmgan_attn_train_forward(eng._h, x.data_ptr(), N, L, ctypes.byref(p),
                                                           m.data_ptr() if m is not None else None, scale, r,
                                                           y.data_ptr(), ws.data_ptr(), ws.numel(), eng._stream()))
        self._shape = (N, L)
        return y

    def backward(self, x: torch.Tensor, dy: torch.Tensor, mask: Optional[torch.Tensor] = None,
                 dresidual: Optional[torch.Tensor] = None):
        eng = self.engine
        x, dy = eng._in(x, "x"), eng._in(dy, "dy")
        r = _same_shape(eng, dresidual, x, "dresidual")
        N, L, _ = x.shape
        if self._shape != (N, L) or dy.shape != x.shape:
            raise RuntimeError("backward() needs the forward() of the same [N, L, 64] input first")
        m = _keep_mask(mask, (N, L, 64), self.p, eng.device, "mask")
        ws = self._workspace(N, L)
        dx = torch.empty_like(x)
        p, g = self._struct(self.params), self._struct(self.grads)
        scale = 1.0 / (1.0 - self.p) if self.p > 0.0 else 1.0
        with torch.cuda.device(eng.device):
            check(eng._h, eng.lib.cmgan_attn_train_backward(eng._h, x.data_ptr(), dy.data_ptr(), N, L, ctypes.byref(p),
                                                            m.data_ptr() if m is not None else None, scale, r,
                                                            dx.data_ptr(), ctypes.byref(g), ws.data_ptr(), ws.numel(),
                                                            eng._stream()))
        return dx, self.grads

    def allreduce_gradients(self) -> torch.Tensor:
        return allreduce_mean(self.grad_bucket.flat)


class ConformerBlockTrain:
    """`models.conformer.ConformerBlock(dim=64, dim_head=16, heads=4, conv_kernel_size=31, attn_dropout, ff_dropout)`
    in TRAIN mode on the HIP kernels:  x -> ff1(x)+x -> attn(.)+. -> conv(.)+. -> ff2(.)+. -> post_norm
    (conformer.py:216-222).  Dropout layers are explicit keep-masks (`masks()` draws them), BatchNorm1d runs on batch
    statistics.  All 31 parameters (and their gradients) are views of one FlatBucket: `allreduce_gradients()` is one
    collective, `AdamW(block.engine, block.param_bucket, block.grad_bucket)` one optimiser launch."""

    def __init__(self, state: Dict[str, torch.Tensor], attn_dropout: float = 0.2, ff_dropout: float = 0.2,
                 engine: Optional[Engine] = None, device=None, views=None):
        self.engine = eng = engine if engine is not None else Engine(device=device)
        self.param_bucket, self.grad_bucket, self.params, self.grads = _buckets(
            self.shapes(eng.cfg.max_pos_emb), state, views, eng.device)

        views = lambda prefix: _subviews((self.params, self.grads), prefix)
        self.ff1 = FeedForwardTrain(dropout=ff_dropout, engine=eng, views=views("ff1"))
        self.attn = AttentionTrain(dropout=attn_dropout, engine=eng, views=views("attn"))
        self.conv = ConvModuleTrain({k[5:]: v for k, v in state.items() if k.startswith("conv.")}, engine=eng,
                                    views=views("conv"))
        self.ff2 = FeedForwardTrain(dropout=ff_dropout, engine=eng, views=views("ff2"))
        self._saved = None
        self._lnws: Optional[torch.Tensor] = None

    @staticmethod
    def shapes(max_pos_emb: int = 512) -> Dict[str, tuple]:
        sub = {"ff1": FeedForwardTrain.SHAPES, "attn": AttentionTrain.shapes(max_pos_emb),
               "conv": ConvModuleTrain.SHAPES, "ff2": FeedForwardTrain.SHAPES,
               "post_norm": {"weight": (64,), "bias": (64,)}}
        return {f"{p}.{k}": shp for p, d in sub.items() for k, shp in d.items()}

    def masks(self, N: int, L: int, generator: Optional[torch.Generator] = None) -> Dict[str, Optional[torch.Tensor]]:
        a1, a2 = self.ff1.masks(N * L, generator)
        at = self.attn.mask(N, L, generator)
        b1, b2 = self.ff2.masks(N * L, generator)
        return {"ff1_1": a1, "ff1_2": a2, "attn": at, "ff2_1": b1, "ff2_2": b2}

    def _add(self, a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
        eng = self.engine
        out = torch.empty_like(a)
        check(eng._h, eng.lib.cmgan_add(eng._h, a.data_ptr(), b.data_ptr(), out.data_ptr(), a.numel(), eng._stream()))
        return out

    def forward(self, x: torch.Tensor, masks: Optional[Dict[str, Optional[torch.Tensor]]] = None,
                residual: Optional[torch.Tensor] = None) -> torch.Tensor:
        """ConformerBlock(x) [+ residual, fused into the post_norm store: the `+ x` of generator.py:95,97]."""
        eng = self.engine
        m = masks or {}
        x0 = eng._in(x, "x")
        r = _same_shape(eng, residual, x0, "residual")
        with torch.cuda.device(eng.device):
            x1 = self.ff1.forward(x0, m.get("ff1_1"), m.get("ff1_2"), residual=x0)      # the `+ x` of :216-219 is
            x2 = self.attn.forward(x1, m.get("attn"), residual=x1)                       # fused into each branch's
            x3 = self.conv.forward(x2, residual=x2)                                      # last kernel
            x4 = self.ff2.forward(x3, m.get("ff2_1"), m.get("ff2_2"), residual=x3)
            y = torch.empty_like(x4)
            check(eng._h, eng.lib.cmgan_layernorm_train_forward(eng._h, x4.data_ptr(), x4.numel() // 64,
                                                                self.params["post_norm.weight"].data_ptr(),
                                                                self.params["post_norm.bias"].data_ptr(), r,
                                                                y.data_ptr(), eng._stream()))
        self._saved = (x0, x1, x2, x3, x4, m)
        return y

    def backward(self, dy: torch.Tensor) -> Tuple[torch.Tensor, Dict[str, torch.Tensor]]:
        """(dL/dx, {key: dL/dparam} = views of the flat gradient bucket) for the last forward()."""
        if self._saved is None:
            raise RuntimeError("backward() needs a forward() first")
        eng = self.engine
        x0, x1, x2, x3, x4, m = self._saved
        dy = eng._in(dy, "dy")
        M = x4.numel() // 64
        need = eng.lib.cmgan_layernorm_train_workspace_bytes(eng._h, M)
        if self._lnws is None or self._lnws.numel() < need:
            self._lnws = torch.empty(need, dtype=torch.uint8, device=eng.device)
        with torch.cuda.device(eng.device):
            d4 = torch.empty_like(x4)
            check(eng._h, eng.lib.cmgan_layernorm_train_backward(
                eng._h, x4.data_ptr(), dy.data_ptr(), M, self.params["post_norm.weight"].data_ptr(),
                self.params["post_norm.bias"].data_ptr(), d4.data_ptr(), self.grads["post_norm.weight"].data_ptr(),
                self.grads["post_norm.bias"].data_ptr(), self._lnws.data_ptr(), self._lnws.numel(), eng._stream()))
            d3 = self.ff2.backward(x3, d4, m.get("ff2_1"), m.get("ff2_2"), dresidual=d4)[0]
            d2 = self.conv.backward(x2, d3, dresidual=d3)[0]
            d1 = self.attn.backward(x1, d2, m.get("attn"), dresidual=d2)[0]
            d0 = self.ff1.backward(x0, d1, m.get("ff1_1"), m.get("ff1_2"), dresidual=d1)[0]
        return d0, self.grads

    def allreduce_gradients(self) -> torch.Tensor:
        if self.grad_bucket is None:
            raise RuntimeError("this module's gradients are views of its parent's bucket: all-reduce the parent")
        return allreduce_mean(self.grad_bucket.flat)


class TSCBTrain:
    """`models.generator.TSCB(num_channel=64)` (generator.py:72-99) in TRAIN mode on the HIP kernels.  Activations are
    channels-last `[B, T, F', 64]` (the layout of the inference path); the reference's NCHW `[B, 64, T, F']` tensor is
    `x.permute(0, 2, 3, 1)`.  `state` = the block's slice of the generator state_dict (`time_conformer.*`,
    `freq_conformer.*`)."""

    def __init__(self, state: Dict[str, torch.Tensor], engine: Optional[Engine] = None, device=None, views=None):
        self.engine = eng = engine if engine is not None else Engine(device=device)
        sub = lambda p: {k[len(p) + 1:]: v for k, v in state.items() if k.startswith(p + ".")}
        vw = lambda p: None if views is None else _subviews(views, p)
        self.time = ConformerBlockTrain(sub("time_conformer"), engine=eng, views=vw("time_conformer"))
        self.freq = ConformerBlockTrain(sub("freq_conformer"), engine=eng, views=vw("freq_conformer"))
        self._shape = None

    @staticmethod
    def shapes(max_pos_emb: int = 512) -> Dict[str, tuple]:
        blk = ConformerBlockTrain.shapes(max_pos_emb)
        return {f"{p}.{k}": shp for p in ("time_conformer", "freq_conformer") for k, shp in blk.items()}

    def masks(self, B: int, T: int, F2: int, generator: Optional[torch.Generator] = None):
        return self.time.masks(B * F2, T, generator), self.freq.masks(B * T, F2, generator)

    def _swap(self, x: torch.Tensor, B: int, A: int, C: int, add: Optional[torch.Tensor] = None) -> torch.Tensor:
        """[B, A, C, 64] -> [B, C, A, 64] of x (+ add, same layout as x: a residual riding on the flip)."""
        eng = self.engine
        out = torch.empty(B, C, A, 64, dtype=torch.float32, device=x.device)
        check(eng._h, eng.lib.cmgan_swap_axes(eng._h, x.data_ptr(), add.data_ptr() if add is not None else None,
                                              out.data_ptr(), B, A, C, eng._stream()))
        return out

    def forward(self, x: torch.Tensor, masks_time=None, masks_freq=None) -> torch.Tensor:
        eng = self.engine
        x = eng._in(x, "x")
        B, T, F2, C = x.shape
        if C != 64:
            raise ValueError("expected channels-last [B, T, F', 64]")
        with torch.cuda.device(eng.device):
            xt = self._swap(x, B, T, F2).view(B * F2, T, 64)                       # generator.py:94
            yt = self.time.forward(xt, masks_time)                                 # :95, its `+ x_t` rides on
            xf = self._swap(yt, B, F2, T, add=xt).view(B * T, F2, 64)              # the flip of :96
            xf = self.freq.forward(xf, masks_freq, residual=xf)                    # :97
        self._shape = (B, T, F2)
        return xf.view(B, T, F2, 64)

    def backward(self, dy: torch.Tensor) -> torch.Tensor:
        """dL/dx [B, T, F', 64] for the last forward(); parameter gradients land in time.grads / freq.grads."""
        if self._shape is None:
            raise RuntimeError("backward() needs a forward() first")
        eng = self.engine
        B, T, F2 = self._shape
        dy = eng._in(dy, "dy").view(B * T, F2, 64)
        with torch.cuda.device(eng.device):
            dxt2 = self._swap(self.freq.backward(dy)[0], B, T, F2, add=dy).view(B * F2, T, 64)
            return self._swap(self.time.backward(dxt2)[0], B, F2, T, add=dxt2)


class DenseBlockTrain:
    """`models.generator.DilatedDenseNet(depth=4, in_channels=64)` (generator.py:6-47) in train mode on the HIP kernels.
    Activations are channels-last `[B, T, F, 64]`.  `state` = the block's tensors under the reference's key names
    (`conv1.weight` ... `prelu4.weight`), e.g. the `dense_encoder.dilated_dense.` slice of the generator state_dict."""

    SHAPES = {**{f"conv{i}.weight": (64, 64 * i, 2, 3) for i in range(1, 5)},
              **{f"conv{i}.bias": (64,) for i in range(1, 5)},
              **{f"norm{i}.weight": (64,) for i in range(1, 5)}, **{f"norm{i}.bias": (64,) for i in range(1, 5)},
              **{f"prelu{i}.weight": (64,) for i in range(1, 5)}}

    def __init__(self, state: Optional[Dict[str, torch.Tensor]] = None, engine: Optional[Engine] = None, device=None,
                 views=None):
        self.engine = engine if engine is not None else Engine(device=device)
        self.param_bucket, self.grad_bucket, self.params, self.grads = _buckets(self.SHAPES, state, views,
                                                                                self.engine.device)
        self._ws: Optional[torch.Tensor] = None
        self._shape = None

    def _struct(self, tensors) -> DenseParams:
        s = DenseParams()
        for i in range(4):
            s.conv_weight[i] = tensors[f"conv{i + 1}.weight"].data_ptr()
            s.conv_bias[i] = tensors[f"conv{i + 1}.bias"].data_ptr()
            s.norm_weight[i] = tensors[f"norm{i + 1}.weight"].data_ptr()
            s.norm_bias[i] = tensors[f"norm{i + 1}.bias"].data_ptr()
            s.prelu_weight[i] = tensors[f"prelu{i + 1}.weight"].data_ptr()
        return s

    def _workspace(self, B: int, T: int, F: int) -> torch.Tensor:
        need = self.engine.lib.cmgan_dense_train_workspace_bytes(self.engine._h, B, T, F)
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty(need, dtype=torch.uint8, device=self.engine.device)
        return self._ws

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        eng = self.engine
        x = eng._in(x, "x")
        B, T, F, C = x.shape
        if C != 64:
            raise ValueError("expected channels-last [B, T, F, 64]")
        ws = self._workspace(B, T, F)
        y = torch.empty_like(x)
        p = self._struct(self.params)
        with torch.cuda.device(eng.device):
            check(eng._h, eng.lib.cmgan_dense_train_forward(eng._h, x.data_ptr(), B, T, F, ctypes.byref(p), y.data_ptr(),
                                                            ws.data_ptr(), ws.numel(), eng._stream()))
        self._shape = (B, T, F)
        return y

    def backward(self, x: torch.Tensor, dy: torch.Tensor):
        eng = self.engine
        x, dy = eng._in(x, "x"), eng._in(dy, "dy")
        B, T, F, _ = x.shape
        if self._shape != (B, T, F) or dy.shape != x.shape:
            raise RuntimeError("backward() needs the forward() of the same [B, T, F, 64] input first")
        ws = self._workspace(B, T, F)
        dx = torch.empty_like(x)
        p, g = self._struct(self.params), self._struct(self.grads)
        with torch.cuda.device(eng.device):
            check(eng._h, eng.lib.cmgan_dense_train_backward(eng._h, x.data_ptr(), dy.data_ptr(), B, T, F, ctypes.byref(p),
                                                             dx.data_ptr(), ctypes.byref(g), ws.data_ptr(), ws.numel(),
                                                             eng._stream()))
        return dx, self.grads

    def allreduce_gradients(self) -> torch.Tensor:
        if self.grad_bucket is None:
            raise RuntimeError("this module's gradients are views of its parent's bucket: all-reduce the parent")
        return allreduce_mean(self.grad_bucket.flat)


def _dense_struct(tensors, prefix: str) -> DenseParams:
    s = DenseParams()
    for i in range(4):
        s.conv_weight[i] = tensors[f"{prefix}conv{i + 1}.weight"].data_ptr()
        s.conv_bias[i] = tensors[f"{prefix}conv{i + 1}.bias"].data_ptr()
        s.norm_weight[i] = tensors[f"{prefix}norm{i + 1}.weight"].data_ptr()
        s.norm_bias[i] = tensors[f"{prefix}norm{i + 1}.bias"].data_ptr()
        s.prelu_weight[i] = tensors[f"{prefix}prelu{i + 1}.weight"].data_ptr()
    return s


class _WsModule:
    """Common plumbing of the encoder / decoder wrappers: buckets (own or views) and a grow-only workspace."""

    def __init__(self, shapes, state, engine, device, views):
        self.engine = engine if engine is not None else Engine(device=device)
        self.param_bucket, self.grad_bucket, self.params, self.grads = _buckets(shapes, state, views, self.engine.device)
        self._ws: Optional[torch.Tensor] = None
        self._shape = None

    def _grow(self, need: int) -> torch.Tensor:
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty(need, dtype=torch.uint8, device=self.engine.device)
        return self._ws

    def allreduce_gradients(self) -> torch.Tensor:
        if self.grad_bucket is None:
            raise RuntimeError("this module's gradients are views of its parent's bucket: all-reduce the parent")
        return allreduce_mean(self.grad_bucket.flat)


class DenseEncoderTrain(_WsModule):
    """`models.generator.DenseEncoder(in_channel=3, channels=64)` (generator.py:50-69) in train mode on the HIP
    kernels.  Input `[B, T, F, 3]` channels-last (|spec|, re, im), output `[B, T, F', 64]`, F' = (F - 1) // 2 + 1.
    `state`: the `dense_encoder.` slice of the generator state_dict.  The input carries no gradient."""

    SHAPES = {"conv_1.0.weight": (64, 3, 1, 1), "conv_1.0.bias": (64,), "conv_1.1.weight": (64,), "conv_1.1.bias": (64,),
              "conv_1.2.weight": (64,),
              **{f"dilated_dense.{k}": v for k, v in DenseBlockTrain.SHAPES.items()},
              "conv_2.0.weight": (64, 64, 1, 3), "conv_2.0.bias": (64,), "conv_2.1.weight": (64,), "conv_2.1.bias": (64,),
              "conv_2.2.weight": (64,)}

    def __init__(self, state=None, engine: Optional[Engine] = None, device=None, views=None):
        super().__init__(self.SHAPES, state, engine, device, views)

    def _struct(self, t) -> EncoderParams:
        s = EncoderParams()
        for i, pre in ((1, "conv_1"), (2, "conv_2")):
            setattr(s, f"conv{i}_weight", t[pre + ".0.weight"].data_ptr())
            setattr(s, f"conv{i}_bias", t[pre + ".0.bias"].data_ptr())
            setattr(s, f"norm{i}_weight", t[pre + ".1.weight"].data_ptr())
            setattr(s, f"norm{i}_bias", t[pre + ".1.bias"].data_ptr())
            setattr(s, f"prelu{i}_weight", t[pre + ".2.weight"].data_ptr())
        s.dense = _dense_struct(t, "dilated_dense.")
        return s

    def forward(self, xin: torch.Tensor) -> torch.Tensor:
        eng = self.engine
        xin = eng._in(xin, "xin")
        B, T, F, C = xin.shape
        if C != 3:
            raise ValueError("expected channels-last [B, T, F, 3]")
        ws = self._grow(eng.lib.cmgan_encoder_train_workspace_bytes(eng._h, B, T, F))
        y = torch.empty(B, T, (F - 1) // 2 + 1, 64, dtype=torch.float32, device=eng.device)
        p = self._struct(self.params)
        with torch.cuda.device(eng.device):
            check(eng._h, eng.lib.cmgan_encoder_train_forward(eng._h, xin.data_ptr(), B, T, F, ctypes.byref(p), y.data_ptr(),
                                                              ws.data_ptr(), ws.numel(), eng._stream()))
        self._shape = (B, T, F)
        self._xin = xin
        return y

    def backward(self, dy: torch.Tensor) -> Dict[str, torch.Tensor]:
        if self._shape is None:
            raise RuntimeError("backward() needs a forward() first")
        eng = self.engine
        B, T, F = self._shape
        dy = eng._in(dy, "dy")
        if tuple(dy.shape) != (B, T, (F - 1) // 2 + 1, 64):
            raise ValueError("dy does not match the forward() output")
        p, g = self._struct(self.params), self._struct(self.grads)
        with torch.cuda.device(eng.device):
            check(eng._h, eng.lib.cmgan_encoder_train_backward(eng._h, self._xin.data_ptr(), dy.data_ptr(), B, T, F,
                                                               ctypes.byref(p), ctypes.byref(g), self._ws.data_ptr(),
                                                               self._ws.numel(), eng._stream()))
        return self.grads


class DecoderTrain(_WsModule):
    """`MaskDecoder(num_features)` (kind "mask", generator.py:121-138) or `ComplexDecoder()` (kind "complex",
    generator.py:141-156) in train mode on the HIP kernels.  Input `[B, T, F', 64]` channels-last; output `[B, T, F]`
    (mask) or `[B, T, F, 2]` (complex), F = 2 F' - 1.  `state`: the `mask_decoder.` / `complex_decoder.` slice."""

    KINDS = {"mask": 0, "complex": 1}

    @staticmethod
    def shapes(kind: str, num_features: int = 201) -> Dict[str, tuple]:
        d = {f"dense_block.{k}": v for k, v in DenseBlockTrain.SHAPES.items()}
        d.update({"sub_pixel.conv.weight": (128, 64, 1, 3), "sub_pixel.conv.bias": (128,)})
        if kind == "mask":
            d.update({"conv_1.weight": (1, 64, 1, 2), "conv_1.bias": (1,), "norm.weight": (1,), "norm.bias": (1,),
                      "prelu.weight": (1,), "final_conv.weight": (1, 1, 1, 1), "final_conv.bias": (1,),
                      "prelu_out.weight": (num_features,)})
        else:
            d.update({"prelu.weight": (64,), "norm.weight": (64,), "norm.bias": (64,), "conv.weight": (2, 64, 1, 2),
                      "conv.bias": (2,)})
        return d

    def __init__(self, kind: str, state=None, num_features: Optional[int] = None, engine: Optional[Engine] = None,
                 device=None, views=None):
        if kind not in self.KINDS:
            raise ValueError("kind must be 'mask' or 'complex'")
        eng = engine if engine is not None else Engine(device=device)
        self.kind = kind
        self.num_features = int(num_features if num_features is not None else eng.cfg.num_features)
        super().__init__(self.shapes(kind, self.num_features), state, eng, device, views)

    def _struct(self, t) -> DecoderParams:
        s = DecoderParams()
        s.dense = _dense_struct(t, "dense_block.")
        s.sub_pixel_weight = t["sub_pixel.conv.weight"].data_ptr()
        s.sub_pixel_bias = t["sub_pixel.conv.bias"].data_ptr()
        conv = "conv_1" if self.kind == "mask" else "conv"
        s.conv_weight, s.conv_bias = t[conv + ".weight"].data_ptr(), t[conv + ".bias"].data_ptr()
        s.norm_weight, s.norm_bias = t["norm.weight"].data_ptr(), t["norm.bias"].data_ptr()
        s.prelu_weight = t["prelu.weight"].data_ptr()
        if self.kind == "mask":
            s.final_weight, s.final_bias = t["final_conv.weight"].data_ptr(), t["final_conv.bias"].data_ptr()
            s.prelu_out_weight = t["prelu_out.weight"].data_ptr()
        return s

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        eng = self.engine
        x = eng._in(x, "x")
        B, T, Fe, C = x.shape
        if C != 64:
            raise ValueError("expected channels-last [B, T, F', 64]")
        F = 2 * Fe - 1
        if self.kind == "mask" and F != self.num_features:
            raise ValueError(f"2 F' - 1 = {F} does not match num_features = {self.num_features} (prelu_out)")
        ws = self._grow(eng.lib.cmgan_decoder_train_workspace_bytes(eng._h, B, T, Fe))
        out = torch.empty((B, T, F) if self.kind == "mask" else (B, T, F, 2), dtype=torch.float32, device=eng.device)
        p = self._struct(self.params)
        with torch.cuda.device(eng.device):
            check(eng._h, eng.lib.cmgan_decoder_train_forward(eng._h, self.KINDS[self.kind], x.data_ptr(), B, T, Fe,
                                                              ctypes.byref(p), out.data_ptr(), ws.data_ptr(), ws.numel(),
                                                              eng._stream()))
        self._shape = (B, T, Fe)
        self._x = x
        return out

    def backward(self, dout: torch.Tensor) -> Tuple[torch.Tensor, Dict[str, torch.Tensor]]:
        if self._shape is None:
            raise RuntimeError("backward() needs a forward() first")
        eng = self.engine
        B, T, Fe = self._shape
        F = 2 * Fe - 1
        dout = eng._in(dout, "dout")
        if tuple(dout.shape) != ((B, T, F) if self.kind == "mask" else (B, T, F, 2)):
            raise ValueError("dout does not match the forward() output")
        dx = torch.empty_like(self._x)
        p, g = self._struct(self.params), self._struct(self.grads)
        with torch.cuda.device(eng.device):
            check(eng._h, eng.lib.cmgan_decoder_train_backward(eng._h, self.KINDS[self.kind], self._x.data_ptr(),
                                                               dout.data_ptr(), B, T, Fe, ctypes.byref(p), dx.data_ptr(),
                                                               ctypes.byref(g), self._ws.data_ptr(), self._ws.numel(),
                                                               eng._stream()))
        return dx, self.grads


class GeneratorTrain:
    """`models.generator.TSCNet(num_channel=64, num_features)` (generator.py:159-201) in TRAIN mode on the HIP kernels.

    `state` is the reference generator state_dict.  Every learnable tensor (`named_parameters()` of the reference
    module) and its gradient is a view of ONE flat fp32 bucket (`param_bucket` / `grad_bucket`): the data-parallel
    gradient mean is one all-reduce, the optimiser one launch.  BatchNorm running statistics are buffers of the conv
    modules and are exported by `state_dict()`.  `forward` takes the compressed noisy spectrogram `[B, 2, T, F]`
    (`Engine.stft_compress`) and returns `est_real, est_imag [B, 1, T, F]` like the reference module."""

    BLOCKS = ("TSCB_1", "TSCB_2", "TSCB_3", "TSCB_4")

    def __init__(self, state: Dict[str, torch.Tensor], engine: Optional[Engine] = None, device=None):
        self.engine = eng = engine if engine is not None else Engine(device=device)
        F = eng.cfg.num_features
        shapes: Dict[str, tuple] = {}
        shapes.update({f"dense_encoder.{k}": v for k, v in DenseEncoderTrain.SHAPES.items()})
        for b in self.BLOCKS:
            shapes.update({f"{b}.{k}": v for k, v in TSCBTrain.shapes(eng.cfg.max_pos_emb).items()})
        shapes.update({f"mask_decoder.{k}": v for k, v in DecoderTrain.shapes("mask", F).items()})
        shapes.update({f"complex_decoder.{k}": v for k, v in DecoderTrain.shapes("complex", F).items()})
        self.param_bucket, self.grad_bucket, self.params, self.grads = _buckets(shapes, state, None, eng.device)
        views = (self.params, self.grads)
        sub = lambda p: {k[len(p) + 1:]: v for k, v in state.items() if k.startswith(p + ".")}
        self.dense_encoder = DenseEncoderTrain(engine=eng, views=_subviews(views, "dense_encoder"))
        self.blocks = [TSCBTrain(sub(b), engine=eng, views=_subviews(views, b)) for b in self.BLOCKS]
        self.mask_decoder = DecoderTrain("mask", engine=eng, views=_subviews(views, "mask_decoder"))
        self.complex_decoder = DecoderTrain("complex", engine=eng, views=_subviews(views, "complex_decoder"))
        self._saved = None
        self._mask_rng: Dict[int, torch.Tensor] = {}            # seed -> device {seed, offset} of the keep-mask generator

    def _draw_keep_bytes(self, nbytes: int, p: float, generator: Optional[torch.Generator]) -> torch.Tensor:
        """`nbytes` Bernoulli(1 - p) keep flags from the library's Philox4x32-10 kernel (cmgan_dropout_masks): ONE launch at
        the HBM rate instead of torch's bernoulli_ (4.7 ms for the 5.8 GB of masks of a 32-clip step).  The stream is
        scoped to (this model, the generator's seed): {seed, offset} live in device memory and the offset advances on the
        device, so the draw is replayable inside a captured graph.  A generator with a NEW seed starts a new stream; the
        same seed continues the one this model already has (reset_mask_rng() restarts them)."""
        eng = self.engine
        state = self.mask_rng_state(generator)
        n16 = (nbytes + 15) // 16 * 16
        buf = torch.empty(n16, dtype=torch.uint8, device=eng.device)
        with torch.cuda.device(eng.device):              # (like every other native call: the launch goes to eng's device)
            check(eng._h, eng.lib.cmgan_dropout_masks(eng._h, buf.data_ptr(), n16, 1.0 - p, state.data_ptr(),
                                                      eng._stream()))
        return buf[:nbytes]

    def mask_rng_state(self, generator: Optional[torch.Generator] = None) -> torch.Tensor:
        """The device {seed, offset} pair of the stream `generator` selects (created on first use: a host -> device copy,
        so a graph capture calls this BEFORE it starts)."""
        seed = (generator.initial_seed() if generator is not None else torch.initial_seed()) & ((1 << 63) - 1)
        state = self._mask_rng.get(seed)
        if state is None:
            state = torch.tensor([seed, 0], dtype=torch.int64, device=self.engine.device)
            self._mask_rng[seed] = state
        return state

    def reset_mask_rng(self) -> None:
        """Restart every keep-mask stream of this model at offset 0.  The device tensors are kept and zeroed IN PLACE: a
        captured GraphedTrainStep holds their raw pointers, and dropping them would leave its replays reading (and
        advancing) memory the allocator has handed to someone else."""
        for state in self._mask_rng.values():
            state[1] = 0

    def mask_rng_offsets(self) -> Dict[int, int]:
        """{seed: offset} of the keep-mask streams (host copy) - save next to the optimiser state, restore with
        `set_mask_rng_offsets`, so that a resumed run continues the mask sequence instead of repeating it from 0."""
        return {seed: int(state[1]) for seed, state in self._mask_rng.items()}

    def set_mask_rng_offsets(self, offsets: Dict[int, int]) -> None:
        for seed, off in offsets.items():
            state = self._mask_rng.get(int(seed))
            if state is None:
                self._mask_rng[int(seed)] = torch.tensor([int(seed), int(off)], dtype=torch.int64, device=self.engine.device)
            else:
                state[1] = int(off)

    def masks(self, B: int, T: int, generator: Optional[torch.Generator] = None):
        """Keep-masks of every Dropout of the four TSCBs for a [B, 2, T, F] input: [(time, freq)] * 4."""
        Fe = (self.engine.cfg.num_features - 1) // 2 + 1
        ps = {m.p for blk in self.blocks for c in (blk.time, blk.freq) for m in (c.ff1, c.attn, c.ff2)}
        if len(ps) != 1 or next(iter(ps)) <= 0.0:
            return [blk.masks(B, T, Fe, generator) for blk in self.blocks]
        # all forty byte masks from ONE Bernoulli draw (one launch instead of 160): views of a single buffer
        p = next(iter(ps))
        widths = (("ff1_1", 256), ("ff1_2", 64), ("attn", 64), ("ff2_1", 256), ("ff2_2", 64))
        tokens = B * T * Fe
        per_axis = tokens * sum(w for _, w in widths)
        buf = self._draw_keep_bytes(len(self.blocks) * 2 * per_axis, p, generator)
        out, off = [], 0
        for _ in self.blocks:
            pair = []
            for n, l in ((B * Fe, T), (B * T, Fe)):
                d = {}
                for name, w in widths:
                    d[name] = buf[off:off + tokens * w].view(n, l, w)
                    off += tokens * w
                pair.append(d)
            out.append(tuple(pair))
        return out

    def forward(self, spec: torch.Tensor, masks=None) -> Tuple[torch.Tensor, torch.Tensor]:
        eng = self.engine
        spec = eng._in(spec, "spec")
        B, two, T, F = spec.shape
        if two != 2 or F != eng.cfg.num_features:
            raise ValueError(f"expected [B, 2, T, {eng.cfg.num_features}]")
        with torch.cuda.device(eng.device):
            xin = torch.empty(B, T, F, 3, dtype=torch.float32, device=eng.device)
            check(eng._h, eng.lib.cmgan_tscnet_prologue(eng._h, spec.data_ptr(), B, T, xin.data_ptr(), eng._stream()))
            x = self.dense_encoder.forward(xin)                                     # generator.py:181
            for i, blk in enumerate(self.blocks):                                   # :182-185
                mt, mf = masks[i] if masks is not None else (None, None)
                x = blk.forward(x, mt, mf)
            mask = self.mask_decoder.forward(x)                                     # :187
            cplx = self.complex_decoder.forward(x)                                  # :190
            est_real = torch.empty(B, 1, T, F, dtype=torch.float32, device=eng.device)
            est_imag = torch.empty_like(est_real)
            check(eng._h, eng.lib.cmgan_tscnet_epilogue_forward(eng._h, spec.data_ptr(), mask.data_ptr(), cplx.data_ptr(),
                                                                B, T, est_real.data_ptr(), est_imag.data_ptr(),
                                                                eng._stream()))    # :188-199
        self._saved = (spec, B, T, F)
        return est_real, est_imag

    def backward(self, d_real: torch.Tensor, d_imag: torch.Tensor) -> Dict[str, torch.Tensor]:
        """All parameter gradients (views of `grad_bucket`) for the last forward()."""
        if self._saved is None:
            raise RuntimeError("backward() needs a forward() first")
        eng = self.engine
        spec, B, T, F = self._saved
        d_real, d_imag = eng._in(d_real, "d_real"), eng._in(d_imag, "d_imag")
        if d_real.numel() != B * T * F or d_imag.numel() != B * T * F:
            raise ValueError("gradients do not match the forward() outputs")
        with torch.cuda.device(eng.device):
            dmask = torch.empty(B, T, F, dtype=torch.float32, device=eng.device)
            dcplx = torch.empty(B, T, F, 2, dtype=torch.float32, device=eng.device)
            check(eng._h, eng.lib.cmgan_tscnet_epilogue_backward(eng._h, spec.data_ptr(), d_real.data_ptr(),
                                                                 d_imag.data_ptr(), B, T, dmask.data_ptr(),
                                                                 dcplx.data_ptr(), eng._stream()))
            dx_m, _ = self.mask_decoder.backward(dmask)
            dx_c, _ = self.complex_decoder.backward(dcplx)
            dx = self.blocks[0].time._add(dx_m, dx_c)
            for blk in reversed(self.blocks):
                dx = blk.backward(dx)
            self.dense_encoder.backward(dx)
        return self.grads

    def state_dict(self) -> Dict[str, torch.Tensor]:
        """The reference generator state_dict (copies): parameters + the BatchNorm buffers of the eight conv modules."""
        out = {k: v.detach().clone() for k, v in self.params.items()}
        for name, blk in zip(self.BLOCKS, self.blocks):
            for ax, conf in (("time_conformer", blk.time), ("freq_conformer", blk.freq)):
                out[f"{name}.{ax}.conv.net.5.running_mean"] = conf.conv.running_mean.clone()
                out[f"{name}.{ax}.conv.net.5.running_var"] = conf.conv.running_var.clone()
                out[f"{name}.{ax}.conv.net.5.num_batches_tracked"] = torch.tensor(
                    conf.conv.num_batches_tracked, dtype=torch.int64)
        return out

    def allreduce_gradients(self) -> torch.Tensor:
        return allreduce_mean(self.grad_bucket.flat)


def generator_train_step(gen: GeneratorTrain, optimizer: "AdamW", clean: torch.Tensor, noisy: torch.Tensor,
                         loss_weights=(0.1, 0.9, 0.2), generator: Optional[torch.Generator] = None, masks="draw",
                         lr: Optional[float] = None, update: bool = True) -> Tuple[torch.Tensor, torch.Tensor]:
    """One optimisation step of the generator on the non-adversarial loss - Trainer.train_step's generator half
    (train.py:153-193) with forward_generator_step (train.py:72-122) and calculate_generator_loss (train.py:124-151)
    minus the metric-discriminator term: returns (loss, float32[4] terms) of THIS rank before the update.
    clean, noisy: float32 [B, L] on the GPU, L a multiple of hop.  `masks="draw"` draws fresh dropout masks from
    `generator`; pass `gen.masks(...)`-shaped masks to fix them, or None to disable dropout.  `update=False` stops
    after the backward pass (gradients in the bucket; no all-reduce, no optimiser launch)."""
    eng = gen.engine
    clean, noisy = eng._in(clean, "clean"), eng._in(noisy, "noisy")
    B = noisy.shape[0]
    c = eng.rms_scale(noisy)                                    # train.py:75
    noisy_spec = eng.stft_compress(noisy, c)                    # train.py:76-95
    clean_spec = eng.stft_compress(clean, c)                    # train.py:88-98
    T = noisy_spec.shape[2]
    if isinstance(masks, str):
        masks = gen.masks(B, T, generator)
    est_real, est_imag = gen.forward(noisy_spec, masks)         # train.py:100
    est_audio = eng.uncompress_istft(est_real, est_imag)        # train.py:105-112
    La = est_audio.shape[-1]
    clean_cut = clean[:, :La].contiguous()                      # train.py:187: the time term sees the RAW clean batch
    loss, terms = generator_loss_terms(eng, est_real, est_imag, clean_spec, est_audio, clean_cut, loss_weights)
    d_real, d_imag = torch.empty_like(est_real), torch.empty_like(est_imag)
    with torch.cuda.device(eng.device):
        check(eng._h, eng.lib.cmgan_loss_backward(eng._h, est_real.data_ptr(), est_imag.data_ptr(), clean_spec.data_ptr(),
                                                  B, T, est_audio.data_ptr(), clean_cut.data_ptr(),
                                                  float(loss_weights[0]), float(loss_weights[1]), float(loss_weights[2]),
                                                  d_real.data_ptr(), d_imag.data_ptr(), eng._stream()))
    gen.backward(d_real, d_imag)                                # train.py:190 (loss.backward())
    if update:                                                  # False: stop after the backward (GraphedTrainStep)
        gen.allreduce_gradients()                               # DDP's gradient mean, one collective
        optimizer.step(lr)                                      # train.py:191
    return loss, terms


_DISC_CH = (2, 16, 32, 64, 128)


class DiscriminatorTrain:
    """`models.discriminator.Discriminator(ndf=16)` (discriminator.py:29-64) on the HIP kernels of csrc/disc.hip.

    `state` is the module's state_dict (34 entries: `layers.N.weight_orig / weight_u / weight_v` of the six spectral
    norms, InstanceNorm / PReLU / bias / slope tensors).  The 22 learnable tensors and their gradients are views of one
    flat bucket each; the power-iteration vectors are buffers, updated in place by every train-mode forward like
    torch's hook does.  Inputs are `xy [B, T, F, 2] = (|clean|, |est|)` from `pair()`.  Up to `slots` forwards can be
    alive at once (the discriminator loss backpropagates through two of them, train.py:163-170)."""

    @staticmethod
    def shapes() -> Dict[str, tuple]:
        d: Dict[str, tuple] = {}
        for i in range(4):
            d[f"layers.{3 * i}.weight_orig"] = (_DISC_CH[i + 1], _DISC_CH[i], 4, 4)
            d[f"layers.{3 * i + 1}.weight"] = (_DISC_CH[i + 1],)
            d[f"layers.{3 * i + 1}.bias"] = (_DISC_CH[i + 1],)
            d[f"layers.{3 * i + 2}.weight"] = (_DISC_CH[i + 1],)
        d.update({"layers.14.weight_orig": (64, 128), "layers.14.bias": (64,), "layers.16.weight": (64,),
                  "layers.17.weight_orig": (1, 64), "layers.17.bias": (1,), "layers.18.slope": (1,)})
        return d

    _SN = ("layers.0", "layers.3", "layers.6", "layers.9", "layers.14", "layers.17")

    def __init__(self, state: Dict[str, torch.Tensor], engine: Optional[Engine] = None, device=None, dropout: float = 0.3,
                 slots: int = 2):
        self.engine = eng = engine if engine is not None else Engine(device=device)
        self.dropout = float(dropout)
        self.param_bucket, self.grad_bucket, self.params, self.grads = _buckets(self.shapes(), state, None, eng.device)
        self._scratch = FlatBucket(self.shapes(), eng.device)          # second graph's gradients before accumulation
        self.buffers = {}
        for p in self._SN:
            for s in ("weight_u", "weight_v"):
                self.buffers[f"{p}.{s}"] = state[f"{p}.{s}"].detach().to(eng.device, torch.float32).clone().contiguous()
        self._ws = [None] * slots
        self._saved = [None] * slots

    def _struct(self, t, with_buffers: bool = True) -> DiscParams:
        s = DiscParams()
        for i in range(4):
            s.conv_weight_orig[i] = t[f"layers.{3 * i}.weight_orig"].data_ptr()
            s.norm_weight[i] = t[f"layers.{3 * i + 1}.weight"].data_ptr()
            s.norm_bias[i] = t[f"layers.{3 * i + 1}.bias"].data_ptr()
            s.prelu_weight[i] = t[f"layers.{3 * i + 2}.weight"].data_ptr()
            if with_buffers:
                s.conv_u[i] = self.buffers[f"layers.{3 * i}.weight_u"].data_ptr()
                s.conv_v[i] = self.buffers[f"layers.{3 * i}.weight_v"].data_ptr()
        s.fc1_weight_orig, s.fc1_bias = t["layers.14.weight_orig"].data_ptr(), t["layers.14.bias"].data_ptr()
        s.prelu5_weight = t["layers.16.weight"].data_ptr()
        s.fc2_weight_orig, s.fc2_bias = t["layers.17.weight_orig"].data_ptr(), t["layers.17.bias"].data_ptr()
        s.slope = t["layers.18.slope"].data_ptr()
        if with_buffers:
            s.fc1_u, s.fc1_v = self.buffers["layers.14.weight_u"].data_ptr(), self.buffers["layers.14.weight_v"].data_ptr()
            s.fc2_u, s.fc2_v = self.buffers["layers.17.weight_u"].data_ptr(), self.buffers["layers.17.weight_v"].data_ptr()
        return s

    def mask(self, B: int, generator: Optional[torch.Generator] = None) -> Optional[torch.Tensor]:
        """Float keep-mask [B, 64] (0 or 1 / 0.7) of the Dropout(0.3): 256 bytes per clip, not worth a byte form."""
        keep = dropout_mask((B, 64), self.dropout, self.engine.device, generator)
        return None if keep is None else keep.to(torch.float32) / (1.0 - self.dropout)

    def pair(self, clean_spec: torch.Tensor, est_real: Optional[torch.Tensor] = None,
             est_imag: Optional[torch.Tensor] = None) -> torch.Tensor:
        """xy [B, T, F, 2] = (|clean_spec|, |est|); without est: (|clean|, |clean|) (train.py:102-103, 166)."""
        eng = self.engine
        clean_spec = eng._in(clean_spec, "clean_spec")
        B, _, T, F = clean_spec.shape
        xy = torch.empty(B, T, F, 2, dtype=torch.float32, device=eng.device)
        er = eng._in(est_real, "est_real").data_ptr() if est_real is not None else None
        ei = eng._in(est_imag, "est_imag").data_ptr() if est_imag is not None else None
        with torch.cuda.device(eng.device):
            check(eng._h, eng.lib.cmgan_mag_pair(eng._h, clean_spec.data_ptr(), er, ei, B, T, xy.data_ptr(), eng._stream()))
        return xy

    def forward(self, xy: torch.Tensor, mask: Optional[torch.Tensor] = None, train: bool = True, slot: int = 0) -> torch.Tensor:
        eng = self.engine
        xy = eng._in(xy, "xy")
        B, T, F, two = xy.shape
        if two != 2 or F != eng.cfg.num_features:
            raise ValueError(f"expected [B, T, {eng.cfg.num_features}, 2]")
        need = eng.lib.cmgan_disc_workspace_bytes(eng._h, B, T)
        if need == 0:
            raise ValueError("the discriminator needs T >= 16 frames and F >= 16 bins")
        if self._ws[slot] is None or self._ws[slot].numel() < need:
            self._ws[slot] = torch.empty(need, dtype=torch.uint8, device=eng.device)
        ws = self._ws[slot]
        score = torch.empty(B, dtype=torch.float32, device=eng.device)
        mask = eng._in(mask, "mask") if mask is not None else None        # float32, contiguous; kept for the backward
        if mask is not None and tuple(mask.shape) != (B, 64):
            raise ValueError(f"mask must be [B, 64] = [{B}, 64]")
        mp = mask.data_ptr() if mask is not None else None
        p = self._struct(self.params)
        with torch.cuda.device(eng.device):
            check(eng._h, eng.lib.cmgan_disc_forward(eng._h, xy.data_ptr(), B, T, ctypes.byref(p), mp, 1 if train else 0,
                                                     score.data_ptr(), ws.data_ptr(), ws.numel(), eng._stream()))
        self._saved[slot] = (xy, mask, B, T)
        return score

    def backward(self, dscore: torch.Tensor, slot: int = 0, need_input_grad: bool = True,
                 accumulate: bool = False) -> Optional[torch.Tensor]:
        """dL/dxy (or None) for the forward of `slot`; parameter gradients are written to (or, with `accumulate`,
        added to) the gradient bucket."""
        if self._saved[slot] is None:
            raise RuntimeError("backward() needs the forward() of the same slot first")
        eng = self.engine
        xy, mask, B, T = self._saved[slot]
        dscore = eng._in(dscore, "dscore")
        dxy = torch.empty_like(xy) if need_input_grad else None
        target = self._scratch.views if accumulate else self.grads
        p, g = self._struct(self.params), self._struct(target, with_buffers=False)
        mp = mask.data_ptr() if mask is not None else None
        ws = self._ws[slot]
        with torch.cuda.device(eng.device):
            check(eng._h, eng.lib.cmgan_disc_backward(eng._h, xy.data_ptr(), dscore.data_ptr(), B, T, ctypes.byref(p), mp,
                                                      dxy.data_ptr() if dxy is not None else None, ctypes.byref(g),
                                                      ws.data_ptr(), ws.numel(), eng._stream()))
            if accumulate:
                gf, sf = self.grad_bucket.flat, self._scratch.flat
                check(eng._h, eng.lib.cmgan_add(eng._h, gf.data_ptr(), sf.data_ptr(), gf.data_ptr(), gf.numel(), eng._stream()))
        return dxy

    def score_mse(self, score: torch.Tensor, target: Optional[torch.Tensor] = None, scale: float = 1.0):
        """(mean((score - target)^2), scale * d/dscore); target None = ones (train.py:129-131, 168-170)."""
        eng = self.engine
        B = score.numel()
        loss = torch.empty(1, dtype=torch.float32, device=eng.device)
        dscore = torch.empty(B, dtype=torch.float32, device=eng.device)
        tp = eng._in(target, "target").data_ptr() if target is not None else None
        with torch.cuda.device(eng.device):
            check(eng._h, eng.lib.cmgan_score_mse(eng._h, score.data_ptr(), tp, B, float(scale), loss.data_ptr(),
                                                  dscore.data_ptr(), eng._stream()))
        return loss[0], dscore

    def state_dict(self) -> Dict[str, torch.Tensor]:
        out = {k: v.detach().clone() for k, v in self.params.items()}
        out.update({k: v.clone() for k, v in self.buffers.items()})
        return out

    def allreduce_gradients(self) -> torch.Tensor:
        return allreduce_mean(self.grad_bucket.flat)


def _adversarial_generator_half(gen: GeneratorTrain, disc: DiscriminatorTrain, clean: torch.Tensor, noisy: torch.Tensor,
                                loss_weights, generator, masks, disc_masks):
    """train.py:100-151, 188-190: STFTs, train-mode generator forward, the full generator loss (RI, magnitude, time and
    the 0.05 x metric-discriminator term) and its backward.  Leaves the generator's gradient bucket filled and returns
    (loss, terms, gan, ctx); ctx carries what the discriminator half needs (est is 'detached' there: only values)."""
    eng = gen.engine
    clean, noisy = eng._in(clean, "clean"), eng._in(noisy, "noisy")
    B = noisy.shape[0]
    c = eng.rms_scale(noisy)
    noisy_spec = eng.stft_compress(noisy, c)
    clean_spec = eng.stft_compress(clean, c)
    T = noisy_spec.shape[2]
    if isinstance(masks, str):
        masks = gen.masks(B, T, generator)
    if isinstance(disc_masks, str):
        disc_masks = [disc.mask(B, generator) for _ in range(3)]
    elif disc_masks is None:
        disc_masks = [None, None, None]
    est_real, est_imag = gen.forward(noisy_spec, masks)                       # train.py:100
    est_audio = eng.uncompress_istft(est_real, est_imag)                      # train.py:105-112
    La = est_audio.shape[-1]
    clean_cut = clean[:, :La].contiguous()
    base, terms = generator_loss_terms(eng, est_real, est_imag, clean_spec, est_audio, clean_cut, loss_weights[:3])
    xy = disc.pair(clean_spec, est_real, est_imag)                            # (clean_mag, est_mag), train.py:102-103
    score = disc.forward(xy, disc_masks[0], train=True, slot=0)               # train.py:126-128
    gan, dscore = disc.score_mse(score, None, scale=float(loss_weights[3]))    # train.py:129-131
    loss = base + float(loss_weights[3]) * gan
    d_real, d_imag = torch.empty_like(est_real), torch.empty_like(est_imag)
    with torch.cuda.device(eng.device):
        check(eng._h, eng.lib.cmgan_loss_backward(eng._h, est_real.data_ptr(), est_imag.data_ptr(), clean_spec.data_ptr(),
                                                  B, T, est_audio.data_ptr(), clean_cut.data_ptr(),
                                                  float(loss_weights[0]), float(loss_weights[1]), float(loss_weights[2]),
                                                  d_real.data_ptr(), d_imag.data_ptr(), eng._stream()))
        dxy = disc.backward(dscore, slot=0, need_input_grad=True)
        check(eng._h, eng.lib.cmgan_mag_pair_backward(eng._h, est_real.data_ptr(), est_imag.data_ptr(), dxy.data_ptr(), B, T,
                                                      1.0, d_real.data_ptr(), d_imag.data_ptr(), eng._stream()))
    gen.backward(d_real, d_imag)                                              # train.py:190
    ctx = dict(xy=xy, clean_spec=clean_spec, est_audio=est_audio, clean_cut=clean_cut, disc_masks=disc_masks)
    return loss, terms, gan, ctx


def _adversarial_discriminator_half(disc: DiscriminatorTrain, ctx, pesq_score: torch.Tensor):
    """train.py:153-171, 194-199: D(clean, est.detach()) against the PESQ labels + D(clean, clean) against 1, and the
    backward of their sum.  Leaves the discriminator's gradient bucket filled; returns the discriminator loss."""
    dm = ctx["disc_masks"]
    s_enh = disc.forward(ctx["xy"], dm[1], train=True, slot=0)                # D(clean, est.detach()), :163-165
    s_max = disc.forward(disc.pair(ctx["clean_spec"]), dm[2], train=True, slot=1)   # D(clean, clean), :166-167
    l_max, d_max = disc.score_mse(s_max, None)
    l_enh, d_enh = disc.score_mse(s_enh, pesq_score)
    disc.backward(d_enh, slot=0, need_input_grad=False)
    disc.backward(d_max, slot=1, need_input_grad=False, accumulate=True)
    return l_max + l_enh                                                      # :168-170


def _agree_on_labels(pesq_score, device):
    """The labels if EVERY rank has them, else None on every rank (see adversarial_train_step)."""
    return pesq_score if all_agree(pesq_score is not None, device) else None


def adversarial_train_step(gen: GeneratorTrain, disc: DiscriminatorTrain, opt_g: "AdamW", opt_d: "AdamW",
                           clean: torch.Tensor, noisy: torch.Tensor, pesq_score: Optional[torch.Tensor],
                           loss_weights=(0.1, 0.9, 0.2, 0.05), generator: Optional[torch.Generator] = None,
                           masks="draw", disc_masks="draw", lr: Optional[float] = None, update: bool = True):
    """Trainer.train_step (train.py:173-205): the generator step on the FULL loss (train.py:124-151: RI, magnitude,
    time and 0.05 x metric-discriminator terms), then the discriminator step (train.py:153-171) on `pesq_score`
    [B] = (PESQ - 1) / 3.5 of (clean, est_audio) - the labels discriminator.batch_pesq computes on the CPU; None (a
    silent clip made PESQ fail) skips the discriminator update like the reference; a callable
    `pesq_score(clean [B, La], est_audio [B, La]) -> tensor | None` is evaluated where the reference calls batch_pesq
    (after the generator update).  Returns
    (generator loss, float32[4] terms, gen_loss_GAN, discriminator loss or None) of this rank before the updates.
    `update=False` leaves both gradient buckets filled and skips the all-reduces and optimiser launches; note that the
    reference updates the generator BEFORE the discriminator forwards, which does not change the discriminator's
    inputs (est is detached), so deferring both updates gives the same gradients.
    Multi-rank: whether the discriminator step (which contains a collective) runs is decided by ALL ranks together - a
    batch whose labels are missing on any rank is skipped on every rank (one MIN all-reduce of a flag); a rank-local
    decision would let one rank's next generator all-reduce pair with the others' discriminator all-reduce (the
    reference has this hazard through DDP, train.py:194-201)."""
    eng = gen.engine
    loss, terms, gan, ctx = _adversarial_generator_half(gen, disc, clean, noisy, loss_weights, generator, masks, disc_masks)
    if update:
        gen.allreduce_gradients()
        opt_g.step(lr)                                                        # train.py:191
    loss_d = None
    if callable(pesq_score):                                                  # discriminator.batch_pesq's place in the step
        pesq_score = pesq_score(ctx["clean_cut"], ctx["est_audio"])
    if update:
        pesq_score = _agree_on_labels(pesq_score, eng.device)
    if pesq_score is not None:                                                # train.py:194-201
        loss_d = _adversarial_discriminator_half(disc, ctx, pesq_score)
        if update:
            disc.allreduce_gradients()
            opt_d.step(None if lr is None else 2.0 * lr)                      # train.py:64-66: twice the generator's rate
    return loss, terms, gan, loss_d


class GraphedTrainStep:
    """A training step captured once as hipGraphs and replayed: the eager step is bound by its ~1 500 kernel launches
    (62 of 72 ms at batch 4 are spent enqueueing), a replay by the kernels.

    The collectives and the host-side PESQ stay OUTSIDE the graphs, in the reference's order (train.py:173-205):
      graph A1  STFTs, train-mode generator forward, the full generator loss incl. the D(clean, est) term, backward
      -> generator gradient all-reduce, graph B_g (generator AdamW)
      -> PESQ labels of THIS step's est_audio: `pesq_score` may be a tensor or - like `adversarial_train_step` - a
         callable `fn(clean_cut, est_audio)` evaluated here, after A1 has produced est_audio (the reference computes the
         labels from the current forward, train.py:156-162); None / a failed batch on ANY rank skips the rest
      graph A2  the two discriminator forwards + backward on those labels
      -> discriminator gradient all-reduce, graph B_d (discriminator AdamW).
    Inputs are copied into static buffers; dropout masks are drawn INSIDE the graphs by the library's Philox kernel, whose
    {seed, offset} state lives in device memory and advances on the device, so every replay sees fresh masks (the
    discriminator head's small mask still comes from torch's graph-safe generator).  `dropout=False` captures the step without dropout
    (deterministic: used by the parity test against the eager step).  Without a discriminator the step is
    `generator_train_step` (graph A1 + B_g)."""

    def __init__(self, gen: GeneratorTrain, opt_g: AdamW, batch: int, length: int, disc: Optional[DiscriminatorTrain] = None,
                 opt_d: Optional[AdamW] = None, loss_weights=(0.1, 0.9, 0.2, 0.05), dropout: bool = True):
        if (disc is None) != (opt_d is None):
            raise ValueError("pass both the discriminator and its optimiser, or neither")
        self.gen, self.disc, self.opt_g, self.opt_d = gen, disc, opt_g, opt_d
        eng = gen.engine
        dev = eng.device
        self.clean = torch.zeros(batch, length, dtype=torch.float32, device=dev)
        self.noisy = torch.ones(batch, length, dtype=torch.float32, device=dev)       # non-zero: the RMS scale divides
        self.pesq = torch.full((batch,), 0.5, dtype=torch.float32, device=dev) if disc is not None else None
        w3 = tuple(float(x) for x in loss_weights[:3])
        _loss_w(w3, dev)                                                # host -> device copies happen before capture
        self._convs = [c.conv for blk in gen.blocks for c in (blk.time, blk.freq)]
        masks = "draw" if dropout else None
        gen.mask_rng_state(None)                                        # the keep-mask stream's device state exists
        torch.cuda.synchronize(dev)
        self.graph_a, self.graph_b = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
        self.graph_a2 = self.graph_bd = None
        self._ctx = None
        with torch.cuda.graph(self.graph_a):                            # ends where the all-reduce / optimiser starts
            if disc is None:
                self.out = generator_train_step(gen, opt_g, self.clean, self.noisy, w3, masks=masks, update=False)
            else:
                loss, terms, gan, self._ctx = _adversarial_generator_half(gen, disc, self.clean, self.noisy, loss_weights,
                                                                          None, masks, masks)
                self.out = (loss, terms, gan)
        with torch.cuda.graph(self.graph_b):
            opt_g.step()
        if disc is not None:
            self.graph_a2, self.graph_bd = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph_a2):
                self.loss_d = _adversarial_discriminator_half(disc, self._ctx, self.pesq)
            with torch.cuda.graph(self.graph_bd):
                opt_d.step()
        # capturing executes nothing: the BatchNorm update counters advanced by the traced forward are rolled back
        for c in self._convs:
            c.num_batches_tracked -= 1

    def __call__(self, clean: torch.Tensor, noisy: torch.Tensor, pesq_score=None):
        self.clean.copy_(clean)
        self.noisy.copy_(noisy)
        self.graph_a.replay()
        for c in self._convs:
            c.num_batches_tracked += 1
        self.gen.allreduce_gradients()
        self.graph_b.replay()
        if self.disc is None:
            return self.out
        if callable(pesq_score):                                        # labels of THIS step's est_audio
            pesq_score = pesq_score(self._ctx["clean_cut"], self._ctx["est_audio"])
        pesq_score = _agree_on_labels(pesq_score, self.gen.engine.device)
        if pesq_score is None:
            return (*self.out, None)
        self.pesq.copy_(pesq_score)
        self.graph_a2.replay()
        self.disc.allreduce_gradients()
        self.graph_bd.replay()
        return (*self.out, self.loss_d)


def batch_pesq(clean: torch.Tensor, est: torch.Tensor, sr: int = 16000) -> Optional[torch.Tensor]:
    """discriminator.batch_pesq (src/models/discriminator.py:9-26): wide-band PESQ of every (clean, estimate) pair on the
    CPU, normalised to (PESQ - 1) / 3.5 on the inputs' device; None when any pair fails (silent clips).  Needs the
    `pesq` wheel the reference depends on: without it there are no labels, and None is returned (the discriminator
    update is then skipped, exactly what the reference does for a failed batch)."""
    from .metrics import _default_pesq
    fn = _default_pesq()
    if fn is None:
        return None
    scores = []
    for c, e in zip(clean.detach().cpu().numpy(), est.detach().cpu().numpy()):
        try:
            scores.append(float(fn(sr, c, e)))
        except Exception:                                      # noqa: BLE001 - "error can happen due to silent period"
            return None
    return ((torch.tensor(scores, dtype=torch.float32) - 1.0) / 3.5).to(clean.device)


class Trainer:
    """`Trainer` of src/train.py:47-275 on the HIP path: `train_step`, `test_step`, `test`, `train` with the reference's
    optimisers (AdamW 5e-4 / 1e-3), StepLR(decay_epoch, 0.5) for both, per-epoch validation and the generator checkpoint
    `CMGAN_epoch_<e>_<loss>` (a plain `state_dict`, loadable by the reference's evaluation.py and by `cmgan_amd.TSCNet`).
    One instance per process / GPU (train.py:277-297 spawns them); gradients are averaged over the initialised
    process group.  `train_ds` / `test_ds` are the loaders of `cmgan_amd.data.load_data`; `pesq_fn` defaults to
    `batch_pesq` (labels need the `pesq` wheel; without it every discriminator update is skipped)."""

    def __init__(self, train_ds, test_ds, generator_state: Dict[str, torch.Tensor],
                 discriminator_state: Dict[str, torch.Tensor], device=None, init_lr: float = 5e-4, decay_epoch: int = 30,
                 loss_weights=(0.1, 0.9, 0.2, 0.05), n_fft: int = 400, hop: int = 100, pesq_fn=batch_pesq,
                 log_interval: int = 500, log=print):
        from .generator import TSCNet
        self.train_ds, self.test_ds = train_ds, test_ds
        self.engine = eng = Engine(n_fft=n_fft, hop=hop, device=device)
        self.gen = GeneratorTrain(generator_state, engine=eng)                            # train.py:52
        self.disc = DiscriminatorTrain(discriminator_state, engine=eng)                   # train.py:58
        self.optimizer = AdamW(eng, self.gen.param_bucket, self.gen.grad_bucket, lr=init_lr)           # train.py:63
        self.optimizer_disc = AdamW(eng, self.disc.param_bucket, self.disc.grad_bucket, lr=2 * init_lr)    # :64-66
        self.init_lr, self.decay_epoch, self.loss_weights = init_lr, decay_epoch, tuple(loss_weights)
        self.pesq_fn, self.log_interval, self.log = pesq_fn, log_interval, log
        self.epoch = 0
        # DistributedDataParallel's start-up semantics (train.py:68-69): every rank continues from rank 0's parameters
        # and buffers, whatever state dict it was handed
        broadcast_from_rank0([self.gen.param_bucket.flat, self.disc.param_bucket.flat])
        self.sync_buffers()
        # eval-mode twin of the generator for test(): the inference path on the CURRENT parameters
        self._eval_model = TSCNet(64, eng.cfg.num_features, n_fft=n_fft, hop=hop, device=eng.device)

    def _batches(self, loader):
        from .data import DevicePrefetcher
        return DevicePrefetcher(loader, self.engine.device)

    def buffers(self) -> list:
        """Every non-parameter state tensor of the two networks: the BatchNorm1d running statistics of the sixteen
        conv modules and the power-iteration vectors of the six spectral norms."""
        out = []
        for blk in self.gen.blocks:
            for c in (blk.time, blk.freq):
                out += [c.conv.running_mean, c.conv.running_var]
        out += [self.disc.buffers[k] for k in sorted(self.disc.buffers)]
        return out

    def sync_buffers(self):
        """DDP(broadcast_buffers=True) - the reference's setting, train.py:68-69 - overwrites every rank's buffers with
        rank 0's before each forward, so the spectral-norm u / v (hence the effective discriminator weights W / sigma)
        and the running statistics are the same on all ranks.  ~20 small broadcasts; identity in a single process."""
        broadcast_from_rank0(self.buffers())

    def train_step(self, clean: torch.Tensor, noisy: torch.Tensor) -> Tuple[float, float]:
        """train.py:173-205: (generator loss, discriminator loss or 0.0 when PESQ gave no labels)."""
        lr = step_lr(self.epoch, self.init_lr, self.decay_epoch)
        self.sync_buffers()
        loss, _, _, loss_d = adversarial_train_step(self.gen, self.disc, self.optimizer, self.optimizer_disc, clean, noisy,
                                                    self.pesq_fn, self.loss_weights, lr=lr)
        return float(loss), (float(loss_d) if loss_d is not None else 0.0)

    @torch.no_grad()
    def test_step(self, clean: torch.Tensor, noisy: torch.Tensor) -> Tuple[float, float]:
        """train.py:207-227 in eval mode: the full generator loss (incl. the metric term) and the discriminator loss."""
        eng = self.engine
        out = forward_generator_step(self._eval_model, clean, noisy)
        La = out["est_audio"].shape[-1]
        clean_cut = clean[:, :La].contiguous()
        base, _ = generator_loss_terms(eng, out["est_real"], out["est_imag"], out["clean_spec"], out["est_audio"], clean_cut,
                                       self.loss_weights[:3])
        xy = self.disc.pair(out["clean_spec"], out["est_real"], out["est_imag"])
        gan, _ = self.disc.score_mse(self.disc.forward(xy, None, train=False, slot=0), None)
        loss = float(base) + self.loss_weights[3] * float(gan)
        labels = self.pesq_fn(clean_cut, out["est_audio"]) if self.pesq_fn is not None else None
        loss_d = 0.0
        if labels is not None:
            s_enh = self.disc.forward(xy, None, train=False, slot=0)
            s_max = self.disc.forward(self.disc.pair(out["clean_spec"]), None, train=False, slot=1)
            loss_d = float(self.disc.score_mse(s_max, None)[0]) + float(self.disc.score_mse(s_enh, labels)[0])
        return loss, loss_d

    def test(self) -> float:
        """train.py:229-245: mean generator loss over the validation set with the networks in eval mode."""
        self._eval_model.load_state_dict(self.gen.state_dict()).eval()
        gen_total = disc_total = 0.0
        steps = 0
        for clean, noisy, _ in self._batches(self.test_ds):
            loss, loss_d = self.test_step(clean, noisy)
            gen_total, disc_total, steps = gen_total + loss, disc_total + loss_d, steps + 1
        steps = max(steps, 1)
        self.log(f"GPU: {self.engine.device}, Generator loss: {gen_total / steps}, Discriminator loss: {disc_total / steps}")
        return gen_total / steps

    def resume_state(self) -> Dict[str, object]:
        """Everything a restarted run needs to CONTINUE this one bit for bit (the reference only saves the generator
        state_dict, train.py:268-274, and so restarts its optimisers and dropout streams): both parameter buckets, both
        AdamW states, the non-parameter buffers, the epoch counter and the keep-mask stream offsets of the generator
        (`GeneratorTrain.mask_rng_offsets`) - without the last a resumed run would repeat the mask sequence from 0."""
        opt = lambda o: {"exp_avg": o.exp_avg.cpu().clone(), "exp_avg_sq": o.exp_avg_sq.cpu().clone(),
                         "state": o.state.cpu().clone()}
        return {"gen_params": self.gen.param_bucket.flat.cpu().clone(),
                "disc_params": self.disc.param_bucket.flat.cpu().clone(),
                "opt_gen": opt(self.optimizer), "opt_disc": opt(self.optimizer_disc),
                "buffers": [b.cpu().clone() for b in self.buffers()],
                "epoch": int(self.epoch), "mask_rng_offsets": self.gen.mask_rng_offsets(),
                # the discriminator head's small Dropout mask still comes from torch's CUDA generator
                "torch_cuda_rng": torch.cuda.get_rng_state(self.engine.device)}

    def load_resume_state(self, state: Dict[str, object]) -> None:
        """In place (captured graphs keep their pointers): the inverse of `resume_state`."""
        self.gen.param_bucket.flat.copy_(state["gen_params"])
        self.disc.param_bucket.flat.copy_(state["disc_params"])
        for o, st in ((self.optimizer, state["opt_gen"]), (self.optimizer_disc, state["opt_disc"])):
            o.exp_avg.copy_(st["exp_avg"])
            o.exp_avg_sq.copy_(st["exp_avg_sq"])
            o.state.copy_(st["state"])
            o.lr = float(st["state"][0])
        for b, v in zip(self.buffers(), state["buffers"]):
            b.copy_(v)
        self.epoch = int(state["epoch"])
        self.gen.set_mask_rng_offsets(state["mask_rng_offsets"])
        torch.cuda.set_rng_state(state["torch_cuda_rng"], self.engine.device)

    def train(self, epochs: int, save_model_dir: Optional[str] = None, rank: Optional[int] = None) -> list:
        """train.py:247-275.  Returns the per-epoch validation losses; rank 0 (of the initialised process group unless
        `rank` is given, like the reference's `gpu_id == 0`) writes the generator checkpoints."""
        import os
        if rank is None:
            rank = get_rank()
        history = []
        for _ in range(epochs):
            for idx, (clean, noisy, _) in enumerate(self._batches(self.train_ds)):
                loss, loss_d = self.train_step(clean, noisy)
                if (idx + 1) % self.log_interval == 0:
                    self.log(f"GPU: {self.engine.device}, Epoch {self.epoch}, Step {idx + 1}, loss: {loss}, disc_loss: {loss_d}")
            gen_loss = self.test()
            history.append(gen_loss)
            if save_model_dir is not None and rank == 0:
                os.makedirs(save_model_dir, exist_ok=True)
                torch.save({k: v.cpu() for k, v in self.gen.state_dict().items()},
                           os.path.join(save_model_dir, "CMGAN_epoch_" + str(self.epoch) + "_" + str(gen_loss)[:5]))
            self.epoch += 1                                                     # scheduler_G.step(); scheduler_D.step()
        return history
