"""Drop-in for ``models.generator.TSCNet`` (reference: src/models/generator.py:159-196).

Same constructor arguments, ``load_state_dict`` consumes the reference's own
generator state_dict, and ``forward(x[B,2,T,F]) -> (real[B,1,T,F], imag[B,1,T,F])``
runs the hand-written HIP kernels through the C ABI.  Inference only (the reference's
``evaluation.py`` path: eval mode, no_grad) - there are no parameters on the Python side.
"""
from __future__ import annotations

import torch

from . import packer
from .engine import Engine
from .synth import state_dict_shapes


class TSCNet:
    def __init__(self, num_channel: int = 64, num_features: int = 201, *, n_fft: int | None = None,
                 hop: int | None = None, device=None, mfma_mode: str | None = None, mix_single=None):
        if num_channel != 64:
            raise ValueError("the HIP kernels are specialised for num_channel=64 (generator.py:160)")
        n_fft = n_fft if n_fft is not None else 2 * (num_features - 1)
        hop = hop if hop is not None else n_fft // 4            # evaluation.py:78
        self.num_channel, self.num_features = num_channel, num_features
        self.engine = Engine(n_fft=n_fft, hop=hop, num_features=num_features, device=device, mfma_mode=mfma_mode,
                             mix_single=mix_single)

    # nn.Module look-alikes so evaluation-style code runs unchanged
    def _check_device(self, device):
        if device is None:
            return
        dev = torch.device(device) if not isinstance(device, int) else torch.device("cuda", device)
        if dev.type != "cuda":
            raise RuntimeError("cmgan_amd has no CPU path: the model lives on the GPU it was constructed on")
        if dev.index is not None and dev != self.engine.device:
            raise RuntimeError(f"this model's engine is bound to {self.engine.device}; construct "
                               f"TSCNet(..., device='{dev}') instead of moving it (one handle = one device)")

    def cuda(self, device=None):
        self._check_device(device)
        return self

    def to(self, *args, **kwargs):
        for a in list(args) + [kwargs.get("device"), kwargs.get("dtype")]:
            if isinstance(a, torch.dtype):
                if a != torch.float32:
                    raise TypeError("the HIP path stores and accumulates in float32 only")
            elif a is not None and not isinstance(a, bool):
                self._check_device(a)
        return self

    def eval(self):
        return self

    def train(self, mode: bool = True):
        if mode:
            raise NotImplementedError(
                "this module is the inference path (packed split-f16 weights, no autograd); the train-mode generator - "
                "Dropout masks, BatchNorm1d batch statistics, every module's backward, AdamW (src/train.py:72-193) - is "
                "cmgan_amd.training.GeneratorTrain / generator_train_step")
        return self

    def load_state_dict(self, state_dict: dict, strict: bool = True):
        """Consumes the reference generator state_dict (src/evaluation.py:63-64; 359 entries).
        Every expected tensor must be present with the reference's shape (there are no Python-side parameters
        to fall back on, so a missing key is fatal even with strict=False); strict=True also rejects keys the
        reference model does not have, like nn.Module.load_state_dict."""
        expected = state_dict_shapes(self.num_features)
        missing = [k for k in expected if k not in state_dict]
        if missing:
            raise KeyError(f"state_dict is missing {len(missing)} of {len(expected)} keys, e.g. {missing[:3]}")
        unexpected = [k for k in state_dict if k not in expected]
        if unexpected and strict:
            raise KeyError(f"state_dict has {len(unexpected)} unexpected keys, e.g. {unexpected[:3]} "
                           "(a DDP checkpoint? strip the 'module.' prefix, or pass strict=False)")
        for k, shape in expected.items():
            got = tuple(state_dict[k].shape)
            if got != shape:
                what = (f"prelu_out has {state_dict[k].numel()} slopes, model was built for {self.num_features}"
                        if k == "mask_decoder.prelu_out.weight" else f"{k}: shape {got}, expected {shape}")
                raise ValueError(what)
        self.engine.load_blob(packer.pack_state_dict({k: state_dict[k] for k in expected}))
        return self

    @torch.no_grad()
    def forward(self, x: torch.Tensor):
        return self.engine.tscnet_forward(x)

    __call__ = forward

    def forward_with_taps(self, x: torch.Tensor):
        """(real, imag, {encoder, tscb1..4, mask, complex}) - NCHW like the reference modules."""
        return self.engine.tscnet_forward(x, taps=True)
