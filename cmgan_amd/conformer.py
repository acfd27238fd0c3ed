"""Drop-in for ``models.conformer.ConformerBlock`` (reference: src/models/conformer.py:182-222)
in eval mode: ``forward(x[N,L,64], mask=None) -> [N,L,64]`` on the HIP kernels.  Only the configuration
CMGAN instantiates (generator.py:75-90) is supported."""
from __future__ import annotations

import torch

from . import packer
from .engine import Engine


class ConformerBlock:
    def __init__(self, *, dim=64, dim_head=16, heads=4, ff_mult=4, conv_expansion_factor=2,
                 conv_kernel_size=31, attn_dropout=0.0, ff_dropout=0.0, conv_dropout=0.0, device=None,
                 mfma_mode=None, mix_single=None):
        if (dim, dim_head, heads, ff_mult, conv_expansion_factor, conv_kernel_size) != (64, 16, 4, 4, 2, 31):
            raise ValueError("HIP kernels are specialised for dim=64, dim_head=16, heads=4, ff_mult=4, "
                             "conv_expansion_factor=2, conv_kernel_size=31")
        self.engine = Engine(device=device, mfma_mode=mfma_mode, mix_single=mix_single)     # dropouts are identity in eval mode

    def eval(self):
        return self

    def cuda(self, *a, **k):
        return self

    def load_state_dict(self, state_dict: dict, strict: bool = True):
        self.engine.load_blob(packer.pack_conformer_state_dict(state_dict, slot=0))
        return self

    @torch.no_grad()
    def forward(self, x: torch.Tensor, mask=None):
        """x [N, L, 64]; mask [N, L] bool as in the reference (conformer.py:216-217, 113-126): it only gates the
        attention scores (pairs of two kept positions keep theirs), every other sub-module sees all rows."""
        return self.engine.conformer_forward(0, x, mask=mask)

    __call__ = forward

    def forward_with_taps(self, x: torch.Tensor):
        """(out, taps[4,N,L,64]) - residual stream after ff1 / attn / conv / ff2."""
        return self.engine.conformer_forward(0, x, taps=True)
