"""Chunked inference over a long waveform with ONE captured hipGraph (BASELINE.json configs[4]).

The reference has no streaming mode: its own mechanism for long audio is the reshape-to-rows rule of
`evaluation.py:30-34` (cmgan_amd.evaluation.enhance_one_track), where every row is enhanced independently and
the RMS scale is that of the whole file.  This module adds the fixed-window variant a streaming front end
needs, with a numerical contract that can be stated against the reference:

    window k covers samples [k*W - C, (k+1)*W + C) of the RMS-scaled signal (zero padded outside the file),
    is enhanced exactly like one reference row of that length, and contributes its central W samples.

`W` = window, `C` = context (look-back / look-ahead "state", re-computed rather than cached: the conformers
attend over the whole window, so a KV cache would change the result, which is why the contract is per window).
`lookahead` (default = C) is the right-hand context alone: `lookahead=0` is the causal-chunk form - window k then
needs no sample beyond (k+1) W, so the algorithmic latency is one window - under the same per-window contract
(window k covers [k W - C, (k+1) W + lookahead)).
All windows have the same shape [1, W + 2C], so the ~250 kernel launches are captured once into a hipGraph
(`Engine.enhance_graphed`-style) and replayed per window; several windows can be batched per replay.
"""
from __future__ import annotations

import math

import torch

from .generator import TSCNet

__all__ = ["enhance_windows"]


@torch.no_grad()
def enhance_windows(model: TSCNet, noisy: torch.Tensor, window: int = 40000, context: int = 4000,
                    batch: int = 4, graph: bool = True, lookahead: int | None = None) -> torch.Tensor:
    """noisy: float32 [1, L] on the GPU -> enhanced [L].  window, context and lookahead must be multiples of hop;
    lookahead (right-hand context) defaults to `context`, 0 = no sample beyond the window's end is used."""
    if noisy.dim() != 2 or noisy.size(0) != 1:
        raise ValueError("expected a mono track shaped [1, L]")
    eng = model.engine
    hop = eng.cfg.hop
    ahead = context if lookahead is None else lookahead
    if window <= 0 or window % hop or context < 0 or context % hop or ahead < 0 or ahead % hop:
        raise ValueError("window, context and lookahead must be non-negative multiples of hop")
    noisy = noisy.to(dtype=torch.float32).contiguous()
    L = noisy.size(-1)
    c = eng.rms_scale(noisy)                                   # file-level scale, as evaluation.py:21
    nwin = int(math.ceil(L / window))
    span = window + context + ahead
    padded = torch.zeros(nwin * window + context + ahead, device=noisy.device, dtype=torch.float32)
    padded[context:context + L] = noisy[0] * c
    rows = padded.unfold(0, span, window).contiguous()         # [nwin, span], window k starts at k*W - C
    out = torch.empty(nwin, window, device=noisy.device, dtype=torch.float32)
    for k0 in range(0, nwin, batch):
        blk = rows[k0:k0 + batch]
        if blk.size(0) < batch:                                 # keep ONE graph shape: pad the last group
            blk = torch.cat([blk, blk.new_zeros(batch - blk.size(0), span)])
        est = _enhance_rows(model, blk.contiguous(), graph)
        n = min(batch, nwin - k0)
        out[k0:k0 + n] = est[:n, context:context + window]
    return out.reshape(-1)[:L] / c


def _enhance_rows(model: TSCNet, rows: torch.Tensor, graph: bool) -> torch.Tensor:
    """One reference row pipeline (stft -> compress -> TSCNet -> uncompress -> istft) on already scaled rows."""
    eng = model.engine
    if not graph:
        spec = eng.stft_compress(rows)
        real, imag = model(spec)
        return eng.uncompress_istft(real, imag)
    key = ("rows",) + tuple(rows.shape)
    cache = eng._row_graphs
    ent = cache.get(key)
    if ent is None or ent[3] != eng._ws_token():
        g_in = torch.empty_like(rows)
        g_in.copy_(rows)
        side = torch.cuda.Stream(device=rows.device)
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):                           # warm-up outside capture (allocates the workspace)
            spec = eng.stft_compress(g_in)
            real, imag = model(spec)
            eng.uncompress_istft(real, imag)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            spec = eng.stft_compress(g_in)
            real, imag = model(spec)
            g_out = eng.uncompress_istft(real, imag)
        ent = (g, g_in, g_out, eng._ws_token())
        cache[key] = ent
    g, g_in, g_out, _ = ent
    g_in.copy_(rows)
    g.replay()
    return g_out
