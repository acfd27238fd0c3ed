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


class TSCNet:
    def __init__(self, num_channel: int = 64, num_features: int = 201, *, n_fft: int | None = None,
                 hop: int | None = None, device=None, mfma_mode: str | None = None):
        if num_channel != 64:
            raise ValueError("the HIP kernels are specialised for num_channel=64 (generator.py:160)")
        n_fft = n_fft if n_fft is not None else 2 * (num_features - 1)
        hop = hop if hop is not None else n_fft // 4            # evaluation.py:78
        self.num_channel, self.num_features = num_channel, num_features
        self.engine = Engine(n_fft=n_fft, hop=hop, num_features=num_features, device=device, mfma_mode=mfma_mode)

    # nn.Module look-alikes so evaluation-style code runs unchanged
    def cuda(self, *a, **k):
        return self

    def to(self, *a, **k):
        return self

    def eval(self):
        return self

    def train(self, mode: bool = True):
        if mode:
            raise NotImplementedError("cmgan_amd implements the inference (eval) forward path only")
        return self

    def load_state_dict(self, state_dict: dict, strict: bool = True):
        missing = [k for k in _expected_keys() if k not in state_dict]
        if missing:
            raise KeyError(f"state_dict is missing {len(missing)} keys, e.g. {missing[:3]}")
        pout = state_dict["mask_decoder.prelu_out.weight"]
        if pout.numel() != self.num_features:
            raise ValueError(f"prelu_out has {pout.numel()} slopes, model was built for {self.num_features}")
        self.engine.load_blob(packer.pack_state_dict(state_dict))
        return self

    @torch.no_grad()
    def forward(self, x: torch.Tensor):
        return self.engine.tscnet_forward(x)

    __call__ = forward

    def forward_with_taps(self, x: torch.Tensor):
        """(real, imag, {encoder, tscb1..4, mask, complex}) - NCHW like the reference modules."""
        return self.engine.tscnet_forward(x, taps=True)


def _expected_keys():
    keys = ["dense_encoder.conv_1.0.weight", "dense_encoder.conv_2.0.weight",
            "mask_decoder.prelu_out.weight", "complex_decoder.conv.weight"]
    for b in range(1, 5):
        for ax in ("time", "freq"):
            keys.append(f"TSCB_{b}.{ax}_conformer.attn.fn.rel_pos_emb.weight")
    return keys
