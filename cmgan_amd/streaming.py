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
import os

import torch

from .generator import TSCNet

__all__ = ["enhance_windows", "enhance_stream", "StreamState", "StreamingEnhancer", "HIST_FRAMES"]


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
        while len(cache) >= 8:                                  # one entry per (batch, span): a few configurations, oldest out
            cache.pop(next(iter(cache)))
        cache[key] = ent
    g, g_in, g_out, _ = ent
    g_in.copy_(rows)
    g.replay()
    return g_out


# ======================================================================================================================
# Streaming with CARRIED state (BASELINE.json configs[4]: "chunked 400-frame windows with KV/state carry, hipGraph-captured")
# ======================================================================================================================
# Contract (include/cmgan_hip.h, "Streaming"; CPU restatement: oracle/stream_oracle.py).  What couples the frames of
# the dense encoder and of the two decoders in the reference is only InstanceNorm2d's statistics over all of T; the
# convolutions themselves are causal in time with a receptive field of 1 + 2 + 4 + 8 = 15 frames back
# (generator.py:16-20, 39-47).  With the statistics FROZEN (a calibration pass, then held) their state carries
# EXACTLY: window k feeds them 15 frames of input history in front of its new frames and drops the first 15 outputs -
# every encoder / decoder frame is computed once, bit-identical to the whole-clip pass under the same statistics.
# The four TSCBs attend over the whole window in both directions (no exact cache exists for them): window k runs them
# on [context | window | look-ahead] frames of encoder outputs that are CACHED (context) or fresh, and keeps the window.
#
#   step k:  encoder   spec frames [e0 - 15, e1)        -> keep [e0, e1),  e1 = min((k + 1) W + La, T): each frame once
#            (the step with e1 = T is the last one and emits every frame from k W on)
#            TSCBs     encoder outputs [k W - Ca, e1)    -> keep [k W, (k + 1) W)
#            decoders  [15 kept TSCB frames of step k - 1 | the kept frames] + their spec frames -> keep the W new ones
#
# STFT / ISTFT are frame-local (a frame sees 400 samples, a sample 4 frames) and run over the clip's frames before /
# after the steps; the steps are what is captured into hipGraphs (one per step shape: first, steady, last).
HIST_FRAMES = 15


class StreamCursor:
    """The host-side bookkeeping of a carried-state stream: which frames the device buffers hold, and for every step
    the integer offsets its captured graph bakes in.  Pure Python (no torch): tests/test_stream_plan.py replays it on
    arrays of frame NUMBERS and checks that every stage sees exactly the frames the contract names.

    Buffers (see _StreamSlot): S holds spectrogram frames [spec_lo, e1), E encoder outputs [enc_lo, e1), the decoder
    input is [h_dec history frames | n_keep kept frames] ending at frame lo + n_keep."""

    def __init__(self, window: int, context: int, lookahead: int):
        self.W, self.Ca, self.La = window, context, lookahead
        self.k = self.e1 = self.spec_lo = self.enc_lo = self.h_dec = self.prev_len = 0

    def plan(self, n_new: int, last: bool) -> dict:
        """Offsets of step k given `n_new` new frames (frames [e1, e1 + n_new)); raises like StreamState.step."""
        W, Ca, H = self.W, self.Ca, HIST_FRAMES
        k, e0 = self.k, self.e1
        e1 = e0 + n_new
        want = (k + 1) * W + self.La
        if (e1 < want and not last) or e1 > want:
            raise ValueError(f"step {k} takes the frames up to {want} (fewer only at the end of the clip), got up to {e1}")
        lo = k * W                                             # first frame this step emits
        n_keep = e1 - lo if last else min(W, e1 - lo)          # the clip's last step also emits what is left past its window
        if n_keep <= 0:
            raise ValueError("no frame left to emit")
        a0 = max(lo - Ca, 0)                                   # TSCB frames [a0, e1): cached context + fresh
        nxt_lo = lo + W
        h_dec = self.h_dec
        return dict(k=k, e0=e0, e1=e1, lo=lo, a0=a0, n_new=n_new, n_keep=n_keep, last=bool(last), par=k & 1,
                    h_enc=min(H, e0),                          # history the encoder can see (0 at the start of the clip)
                    n_tail=e0 - self.spec_lo,                  # frames in S before the new ones
                    n_ctx=e0 - a0,                             # frames in E before the fresh ones (= enc_lo == a0)
                    keep_lo=lo - a0,                           # first kept frame inside the TSCB span
                    h_dec=h_dec, dec_lo=lo - h_dec - self.spec_lo,             # decoder frames [lo - h_dec, lo + n_keep) in S
                    spec_drop=max(max(nxt_lo - H, 0) - self.spec_lo, 0),       # frames S / E shed after the step
                    enc_drop=max(max(nxt_lo - Ca, 0) - self.enc_lo, 0),
                    h_dec_next=min(H, h_dec + n_keep), prev_len=self.prev_len)

    def commit(self, p: dict):
        if not p["last"]:
            self.spec_lo += p["spec_drop"]
            self.enc_lo += p["enc_drop"]
            self.h_dec = p["h_dec_next"]
        self.prev_len = p["h_dec"] + p["n_keep"]
        self.k, self.e1 = p["k"] + 1, p["e1"]


class _StreamSlot:
    """Device-resident state of one stream configuration (B, window, context, look-ahead) plus the hipGraphs of its step
    shapes.  The graphs read AND update the state buffers themselves (a replay is the whole step: no host-side cat /
    clone / contiguous, no allocation), so the buffers must outlive any one clip: the slot lives on the Engine and the
    next StreamState of the same configuration re-uses buffers and graphs.

        S  [B,2,cs,F]    spectrogram frames [s_lo, e0) from position 0 on; a step appends its new frames
        E  [B,ce,F',64]  cached encoder outputs of frames [a0, e0); a step appends the fresh ones
        D  [B,cd,F',64]  decoder input: [the last <= 15 kept TSCB frames of the previous step | this step's kept frames]

    Valid lengths are host integers (StreamState); every step shape bakes its own offsets into its graph."""
    MAX_GRAPHS = 6                                           # first / steady / last shapes of a clip length, with spares

    def __init__(self, eng, B: int, W: int, Ca: int, La: int):
        dev, F, F2 = eng.device, eng.F, (eng.F + 1) // 2
        n_max = W + La                                       # most frames a step takes (the first one)
        self.S = torch.empty(B, 2, HIST_FRAMES + La + n_max, F, device=dev)
        self.E = torch.empty(B, Ca + La + n_max, F2, 64, device=dev)
        self.D = torch.empty(B, HIST_FRAMES + n_max, F2, 64, device=dev)
        self.stats = torch.empty(eng.stats_floats(B), device=dev)
        self.graphs: dict = {}                               # step signature -> (graph, spec_in, out_real, out_imag, token); LRU
        self.owner = None                                    # weakref of the StreamState using the buffers
        # pipelined form (StreamState.step_pipelined): three stages on three streams - encoder of step k + 2 | TSCBs of step
        # k + 1 | decoders of step k - so what crosses a stage boundary is double-buffered by step parity and the outer
        # stages get workspaces of their own
        self.n_max, self.B = n_max, B
        self.ENC = self.D2 = self.SD = self.ws_enc = self.ws_dec = self.enc_stream = self.dec_stream = None
        self.stage_graphs: dict = {}                         # (stage, signature) -> (graph, outputs, token); LRU
        self.spec_in: dict = {}                              # n_new -> static input of the encoder stage
        self.ev = {"front": [None, None], "mid": [None, None], "dec": [None, None]}   # per stage and parity: last run done

    def pipeline_buffers(self, eng):
        if self.D2 is None:
            dev, F, F2 = eng.device, eng.F, (eng.F + 1) // 2
            nd = HIST_FRAMES + self.n_max
            self.ENC = [torch.empty(self.B, self.n_max, F2, 64, device=dev) for _ in range(2)]     # fresh encoder outputs
            self.D2 = [torch.empty(self.B, nd, F2, 64, device=dev) for _ in range(2)]              # decoder input
            self.SD = [torch.empty(self.B, 2, nd, F, device=dev) for _ in range(2)]                # its spectrogram frames
            self.ws_enc = torch.empty(eng.workspace_bytes(self.B, nd), dtype=torch.uint8, device=dev)
            self.ws_dec = torch.empty(eng.workspace_bytes(self.B, nd), dtype=torch.uint8, device=dev)
            self.enc_stream = torch.cuda.Stream(device=dev)
            self.dec_stream = torch.cuda.Stream(device=dev)


class StreamState:
    """Carried state of one stream (or B streams in lock-step): the statistics blob, the encoder-output cache, the
    decoder's input history and the spectrogram frames they belong to.  `step(spec_frames)` is frame-level.
    graph=True: the state lives in an Engine-owned _StreamSlot and every step is ONE hipGraph replay."""

    def __init__(self, model: TSCNet, stats: torch.Tensor, B: int, window: int, context: int, lookahead: int,
                 graph: bool = True):
        if window <= 0 or context < 0 or lookahead < 0:
            raise ValueError("window must be positive, context / lookahead non-negative (frames)")
        self.model, self.eng, self.stats, self.B = model, model.engine, stats, B
        self.W, self.Ca, self.La, self.graph = window, context, lookahead, graph
        self.F, self.F2 = self.eng.F, (self.eng.F + 1) // 2
        dev = self.eng.device
        self.cur = StreamCursor(window, context, lookahead)     # steps done, frames held, per-step offsets
        #: stages of step_pipelined that get their own stream: 2 = [encoder + TSCBs] | decoders, 3 = encoder | TSCBs | decoders
        self.pipeline_stages = 3 if os.environ.get("CMGAN_STREAM_STAGES", "2") == "3" else 2
        if graph:
            self.slot = self._claim_slot()
            self.slot.stats.copy_(self.eng._in(stats, "stats"))
            self.stats = self.slot.stats                               # (what the graphs' kernel arguments point at)
        else:
            self.enc = torch.empty(B, 0, self.F2, 64, device=dev)          # cached encoder outputs
            self.spec_tail = torch.empty(B, 2, 0, self.F, device=dev)       # spec frames [spec_lo, ...) still needed
            self.dec_hist = None             # kept TSCB outputs of the last HIST_FRAMES frames before k W

    k = property(lambda self: self.cur.k)                       # steps done
    e1 = property(lambda self: self.cur.e1)                     # frames fed so far

    def _claim_slot(self) -> _StreamSlot:
        import weakref
        slots = self.eng._stream_slots
        key = (self.B, self.W, self.Ca, self.La)
        slot = slots.get(key)
        if slot is not None and slot.owner is not None and slot.owner() is not None and slot.owner() is not self:
            slot = _StreamSlot(self.eng, *key)                         # another live stream holds the cached one: a private slot
        elif slot is None:
            slot = _StreamSlot(self.eng, *key)
            while len(slots) >= 4:                                     # a handful of configurations per engine, oldest out
                slots.pop(next(iter(slots)))
            slots[key] = slot
        slot.owner = weakref.ref(self)
        return slot

    # ---- one step on explicit tensors (eager form) ----
    def _run(self, spec_enc, enc_ctx, dec_hist, spec_dec, n_new_enc, keep_lo, n_keep):
        eng = self.eng
        x_new = eng.stream_encoder(spec_enc, self.stats)[:, spec_enc.size(2) - n_new_enc:]      # drop the history outputs
        x = torch.cat([enc_ctx, x_new], dim=1).contiguous()
        eng.stream_tscb(x)
        kept = x[:, keep_lo:keep_lo + n_keep]
        xin = kept if dec_hist is None else torch.cat([dec_hist, kept], dim=1)
        real, imag = eng.stream_decoder(xin.contiguous(), spec_dec, self.stats)
        return x_new, kept, real[:, :, xin.size(1) - n_keep:], imag[:, :, xin.size(1) - n_keep:]

    # ---- one step as ONE graph replay over the slot's state buffers ----
    def _graph_body(self, sl: _StreamSlot, spec_in, sig):
        """The captured work of a step shape.  sig = (n_tail, h_enc, n_new, n_ctx, keep_lo, n_keep, h_dec, dec_lo,
        spec_drop, enc_drop, h_dec_next, last): all host integers, baked into the graph."""
        n_tail, h_enc, n_new, n_ctx, keep_lo, n_keep, h_dec, dec_lo, spec_drop, enc_drop, h_dec_next, last = sig
        eng, S, E, D = self.eng, sl.S, sl.E, sl.D
        S[:, :, n_tail:n_tail + n_new].copy_(spec_in)
        enc = eng.stream_encoder(S[:, :, n_tail - h_enc:n_tail + n_new].contiguous(), sl.stats)
        E[:, n_ctx:n_ctx + n_new].copy_(enc[:, h_enc:])                # the history outputs are dropped
        x = E[:, :n_ctx + n_new].clone()                               # the TSCBs work in place: the cache keeps the encoder's
        eng.stream_tscb(x)
        D[:, h_dec:h_dec + n_keep].copy_(x[:, keep_lo:keep_lo + n_keep])
        real, imag = eng.stream_decoder(D[:, :h_dec + n_keep].contiguous(),
                                        S[:, :, dec_lo:dec_lo + h_dec + n_keep].contiguous(), sl.stats)
        if not last:                                                   # carry: what the next step's three stages look back at
            for buf, dim, lo, hi in ((S, 2, spec_drop, n_tail + n_new), (E, 1, enc_drop, n_ctx + n_new),
                                     (D, 1, h_dec + n_keep - h_dec_next, h_dec + n_keep)):
                if lo > 0 and hi > lo:
                    tmp = buf.narrow(dim, lo, hi - lo).clone()         # (source and destination may overlap)
                    buf.narrow(dim, 0, hi - lo).copy_(tmp)
        return real[:, :, h_dec:], imag[:, :, h_dec:]

    def _step_graphed(self, spec_new, sig):
        sl, eng = self.slot, self.eng
        token = eng._ws_token()
        ent = sl.graphs.pop(sig, None)
        if ent is not None and ent[4] != token:
            ent = None
        if ent is None:
            spec_in = spec_new.contiguous().clone()
            # warm-up outside capture on COPIES of the state (a step updates it): allocator pools, workspace growth
            saved = (sl.S.clone(), sl.E.clone(), sl.D.clone())
            side = torch.cuda.Stream(device=eng.device)
            side.wait_stream(torch.cuda.current_stream(eng.device))
            with torch.cuda.stream(side):
                self._graph_body(sl, spec_in, sig)
            torch.cuda.current_stream(eng.device).wait_stream(side)
            torch.cuda.synchronize(eng.device)
            token = eng._ws_token()                                    # (the warm-up may have grown the workspace)
            for dst, src in zip((sl.S, sl.E, sl.D), saved):
                dst.copy_(src)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                out_r, out_i = self._graph_body(sl, spec_in, sig)
            for dst, src in zip((sl.S, sl.E, sl.D), saved):            # (capture does not execute, but keep the rule simple)
                dst.copy_(src)
            del saved
            ent = (g, spec_in, out_r, out_i, token)
        sl.graphs[sig] = ent                                           # most recently used last
        while len(sl.graphs) > _StreamSlot.MAX_GRAPHS:
            sl.graphs.pop(next(iter(sl.graphs)))
        g, spec_in, out_r, out_i, _ = ent
        spec_in.copy_(spec_new)
        g.replay()
        return out_r, out_i              # static buffers of this shape's graph: valid until its next replay

    # ---- pipelined form: encoder of step k + 2 | TSCBs of step k + 1 | decoders of step k, on three streams ----
    def _front_body(self, sl: _StreamSlot, spec_in, sig):
        """Stage 1 (needs spectrogram frames only): append the new frames, encoder -> ENC[parity], the decoder's frames ->
        SD[parity], carry of the spectrogram tail."""
        n_tail, h_enc, n_new, dec_lo, n_dec, spec_drop, last, par = sig
        S = sl.S
        S[:, :, n_tail:n_tail + n_new].copy_(spec_in)
        enc = self.eng.stream_encoder(S[:, :, n_tail - h_enc:n_tail + n_new].contiguous(), sl.stats, ws=sl.ws_enc)
        sl.ENC[par][:, :n_new].copy_(enc[:, h_enc:])                   # the history outputs are dropped
        sl.SD[par][:, :, :n_dec].copy_(S[:, :, dec_lo:dec_lo + n_dec])
        if not last and spec_drop > 0:
            tmp = S[:, :, spec_drop:n_tail + n_new].clone()            # (source and destination may overlap)
            S[:, :, :n_tail + n_new - spec_drop].copy_(tmp)

    def _mid_body(self, sl: _StreamSlot, sig):
        """Stage 2: cached context + fresh encoder outputs -> TSCBs -> the decoder input D2[parity], carry of the cache."""
        n_ctx, n_new, keep_lo, n_keep, h_dec, enc_drop, last, par, prev_len = sig
        E, D, Dp = sl.E, sl.D2[par], sl.D2[par ^ 1]
        E[:, n_ctx:n_ctx + n_new].copy_(sl.ENC[par][:, :n_new])
        x = E[:, :n_ctx + n_new].clone()                               # the TSCBs work in place: the cache keeps the encoder's
        self.eng.stream_tscb(x)
        if h_dec:
            D[:, :h_dec].copy_(Dp[:, prev_len - h_dec:prev_len])       # the previous step's last kept frames (other parity: read only)
        D[:, h_dec:h_dec + n_keep].copy_(x[:, keep_lo:keep_lo + n_keep])
        if not last and enc_drop > 0:
            tmp = E[:, enc_drop:n_ctx + n_new].clone()
            E[:, :n_ctx + n_new - enc_drop].copy_(tmp)

    def _dec_body(self, sl: _StreamSlot, par, h_dec, n_keep):
        """Stage 3: both decoders on [history | kept frames]."""
        real, imag = self.eng.stream_decoder(sl.D2[par][:, :h_dec + n_keep].contiguous(),
                                             sl.SD[par][:, :, :h_dec + n_keep].contiguous(), sl.stats, ws=sl.ws_dec)
        return real[:, :, h_dec:], imag[:, :, h_dec:]

    def _captured(self, sl: _StreamSlot, key, body, stream, restore=()):
        """LRU-cached hipGraph of one stage (warm-up outside capture; `restore`: state a run of the body modifies)."""
        eng = self.eng
        token = eng._ws_token()
        ent = sl.stage_graphs.pop(key, None)
        if ent is not None and ent[2] != token:
            ent = None
        if ent is None:
            torch.cuda.synchronize(eng.device)                         # the other stages are idle while this one is captured
            saved = [t.clone() for t in restore]
            side = torch.cuda.Stream(device=eng.device)
            side.wait_stream(stream)
            with torch.cuda.stream(side):
                body()
            stream.wait_stream(side)
            torch.cuda.synchronize(eng.device)
            for dst, src in zip(restore, saved):
                dst.copy_(src)
            torch.cuda.synchronize(eng.device)
            g = torch.cuda.CUDAGraph()
            # (a graph cannot be captured ON the default stream - the caller's, for the middle stage; torch's own capture
            # stream does it then; replaying on the default stream is fine)
            cap = None if stream == torch.cuda.default_stream(eng.device) else stream
            with torch.cuda.graph(g, stream=cap):
                outs = body()
            for dst, src in zip(restore, saved):                       # (capture does not execute, but keep the rule simple)
                dst.copy_(src)
            torch.cuda.synchronize(eng.device)
            ent = (g, outs, eng._ws_token())
        sl.stage_graphs[key] = ent                                     # most recently used last
        while len(sl.stage_graphs) > 3 * _StreamSlot.MAX_GRAPHS:
            sl.stage_graphs.pop(next(iter(sl.stage_graphs)))
        return ent

    @torch.no_grad()
    def step_pipelined(self, spec_new: torch.Tensor, out, last: bool = False):
        """step() for a driver that has the next windows' frames at hand (offline enhance_stream; a live front end that is
        fed faster than real time).  A step is three captured stages on three streams - encoder | TSCBs | decoders - tied
        by events, so that the encoder of step k + 2 and the decoders of step k run beside the TSCBs of step k + 1: at one
        clip per step no stage fills the chip by itself.  What crosses a stage boundary is double-buffered by step parity;
        the same kernels run on the same values as in step(): the result is bit-identical.  `out` = (real, imag)
        [B,1,>=w,F] are written on the decoder stream - call finish_pipeline() before reading them elsewhere."""
        if not self.graph:
            raise RuntimeError("the pipelined form replays captured graphs: graph=True")
        spec_new = self.eng._in(spec_new, "spec_new")
        p = self.cur.plan(spec_new.size(2), last)
        n_new, n_keep, h_dec, par, last = p["n_new"], p["n_keep"], p["h_dec"], p["par"], p["last"]
        sl, eng = self.slot, self.eng
        if sl.owner is None or sl.owner() is not self:
            raise RuntimeError("this stream's state buffers were claimed by a newer StreamState of the same configuration")
        sl.pipeline_buffers(eng)
        cur, ds = torch.cuda.current_stream(eng.device), sl.dec_stream
        es = sl.enc_stream if self.pipeline_stages == 3 else cur       # 2 stages: the encoder stays in front of the TSCBs on one stream
        ev = sl.ev

        def after(stream, *events):
            for e in events:
                if e is not None:
                    stream.wait_event(e)

        def mark(stream):
            e = torch.cuda.Event()
            e.record(stream)
            return e

        # ---- stage 1 on the encoder stream: ENC / SD of this parity are free once step k - 2's TSCBs / decoders are done ----
        spec_in = sl.spec_in.get(n_new)
        if spec_in is None:
            spec_in = sl.spec_in[n_new] = torch.empty_like(spec_new)
        if es is not cur:
            es.wait_stream(cur)                                        # (spec_new was produced on the caller's stream)
        after(es, ev["mid"][par], ev["dec"][par])
        fsig = (p["n_tail"], p["h_enc"], n_new, p["dec_lo"], h_dec + n_keep, p["spec_drop"], last, par)
        ent = self._captured(sl, ("front",) + fsig, lambda: self._front_body(sl, spec_in, fsig), es, restore=(sl.S,))
        with torch.cuda.stream(es):
            spec_in.copy_(spec_new)
            ent[0].replay()
        ev["front"][par] = mark(es)
        # ---- stage 2 on the caller's stream: D2 of this parity is free once step k - 2's decoders are done ----
        after(cur, ev["front"][par], ev["dec"][par])
        msig = (p["n_ctx"], n_new, p["keep_lo"], n_keep, h_dec, p["enc_drop"], last, par, p["prev_len"])
        ent = self._captured(sl, ("mid",) + msig, lambda: self._mid_body(sl, msig), cur, restore=(sl.E,))
        ent[0].replay()
        ev["mid"][par] = mark(cur)
        # ---- stage 3 on the decoder stream ----
        after(ds, ev["mid"][par], ev["front"][par])
        ent = self._captured(sl, ("dec", par, h_dec, n_keep), lambda: self._dec_body(sl, par, h_dec, n_keep), ds)
        with torch.cuda.stream(ds):
            ent[0].replay()
            r, i = ent[1]
            out[0][:, :, :n_keep].copy_(r)
            out[1][:, :, :n_keep].copy_(i)
        ev["dec"][par] = mark(ds)
        self.cur.commit(p)
        return n_keep

    def finish_pipeline(self):
        """Make the current stream wait for the stages still running on the other two."""
        sl = self.slot
        if sl.dec_stream is not None:
            cur = torch.cuda.current_stream(self.eng.device)
            cur.wait_stream(sl.dec_stream)
            cur.wait_stream(sl.enc_stream)

    @torch.no_grad()
    def step(self, spec_new: torch.Tensor, last: bool = False, out=None):
        """spec_new: [B,2,n,F] = the spectrogram frames that arrived since the previous step (frames [e1, e1 + n)).
        Step k needs frames up to (k + 1) W + La (fewer only when `last`: the clip ended).  Returns (est_real, est_imag)
        [B,1,w,F] for frames [k W, k W + w), w = W - or, when `last`, all that is left of the clip (at most W + La: the
        step whose look-ahead reaches the clip's end is the last one).  out = (real, imag) [B,1,>=w,F] tensors: the
        result is written into their first w frames instead (views of them are returned)."""
        p = self.cur.plan(spec_new.size(2), last)
        n_new, n_keep, h_enc = p["n_new"], p["n_keep"], p["h_enc"]
        if self.graph:
            if self.slot.owner is None or self.slot.owner() is not self:
                raise RuntimeError("this stream's state buffers were claimed by a newer StreamState of the same configuration")
            sig = (p["n_tail"], h_enc, n_new, p["n_ctx"], p["keep_lo"], n_keep, p["h_dec"], p["dec_lo"], p["spec_drop"],
                   p["enc_drop"], p["h_dec_next"], p["last"])
            real, imag = self._step_graphed(self.eng._in(spec_new, "spec_new"), sig)
            self.cur.commit(p)
            if out is not None:
                out[0][:, :, :n_keep].copy_(real)
                out[1][:, :, :n_keep].copy_(imag)
                return out[0][:, :, :n_keep], out[1][:, :, :n_keep]
            return real.clone(), imag.clone()                  # (not views of a graph's static outputs)
        # eager form: the same bookkeeping on tensors that are cut and concatenated on the host
        c = self.cur
        self.spec_tail = torch.cat([self.spec_tail, spec_new], dim=2)
        spec_enc = self.spec_tail[:, :, p["n_tail"] - h_enc:p["n_tail"] + n_new].contiguous()
        enc_ctx = self.enc[:, :p["n_ctx"]].contiguous()
        h_dec = p["h_dec"]
        spec_dec = self.spec_tail[:, :, p["dec_lo"]:p["dec_lo"] + h_dec + n_keep].contiguous()
        x_new, kept, real, imag = self._run(spec_enc, enc_ctx, self.dec_hist, spec_dec, n_new, p["keep_lo"], n_keep)
        # carry: encoder outputs from the next step's context start on, the last H kept TSCB frames, the spec frames both need
        self.enc = torch.cat([self.enc, x_new], dim=1)
        if not p["last"]:
            if p["enc_drop"] > 0:
                self.enc = self.enc[:, p["enc_drop"]:].contiguous()
            if p["spec_drop"] > 0:
                self.spec_tail = self.spec_tail[:, :, p["spec_drop"]:].contiguous()
        hist = kept if self.dec_hist is None else torch.cat([self.dec_hist, kept], dim=1)
        self.dec_hist = hist[:, -p["h_dec_next"]:].clone()                 # the frames just before the next window
        c.commit(p)
        if out is not None:
            out[0][:, :, :n_keep].copy_(real)
            out[1][:, :, :n_keep].copy_(imag)
            return out[0][:, :, :n_keep], out[1][:, :, :n_keep]
        return real, imag


@torch.no_grad()
def enhance_stream(model: TSCNet, noisy: torch.Tensor, window: int = 400, context: int = 40, lookahead: int = 40,
                   stats: torch.Tensor | None = None, calib_frames: int | None = None, graph: bool = True,
                   pipeline: bool | None = None) -> torch.Tensor:
    """noisy: float32 [1, L] on the GPU -> enhanced [L'] (L' = hop * (T - 1), T = L // hop + 1, like one reference row).
    window / context / lookahead are in FRAMES.  `stats`: a frozen statistics blob (Engine.tscnet_forward_stats); by
    default the clip's first `calib_frames` (default window + lookahead) frames calibrate it."""
    if noisy.dim() != 2 or noisy.size(0) != 1:
        raise ValueError("expected a mono track shaped [1, L]")
    eng = model.engine
    noisy = noisy.to(dtype=torch.float32).contiguous()
    c = eng.rms_scale(noisy)                                   # file-level scale, as evaluation.py:21
    spec = eng.stft_compress(noisy, c)                         # [1,2,T,F]: frame-local, see the contract above
    T = spec.size(2)
    if stats is None:
        n = min(T, calib_frames if calib_frames is not None else window + lookahead)
        stats = eng.tscnet_forward_stats(spec[:, :, :n].contiguous())[2]
    st = StreamState(model, stats, 1, window, context, lookahead, graph)
    real = torch.empty(1, 1, T, eng.F, device=noisy.device)
    imag = torch.empty_like(real)
    k, fed = 0, 0
    pipeline = graph if pipeline is None else pipeline         # (the whole clip is at hand: decoders of step k beside step k + 1)
    while fed < T:
        upto = min((k + 1) * window + lookahead, T)
        o = (real[:, :, k * window:], imag[:, :, k * window:])
        if pipeline:
            st.step_pipelined(spec[:, :, fed:upto], o, last=upto == T)
        else:
            st.step(spec[:, :, fed:upto], last=upto == T, out=o)
        fed, k = upto, k + 1
    if pipeline:
        st.finish_pipeline()
    return (eng.uncompress_istft(real, imag) / c[:, None]).reshape(-1)


class StreamingEnhancer:
    """Sample-level front end of `StreamState` for live audio: `push(samples)` takes whatever arrived (any chunk size)
    and returns the enhanced samples that are final by then; `flush()` ends the stream.

    Everything outside the network is frame-local and therefore incremental WITHOUT approximation: STFT frame t reads
    samples [100 t - 200, 100 t + 200) (reflect padding only at the true ends of the stream), output sample s sums the
    four frames floor((s - 200) / 100) + 1 .. floor((s + 200) / 100); so frames are transformed as their samples arrive
    (a block with two guard frames on each side), network steps run as soon as a window's look-ahead is covered, and
    sample s is emitted once frame floor(s / 100) + 2 has been estimated.  Algorithmic latency = window + look-ahead
    frames + 3 frames.  `scale` = the RMS normalisation factor c (evaluation.py:21-23): a live stream cannot know its
    file-level RMS, so it is a parameter (e.g. from the calibration stretch); the output is divided by it again.
    The result equals `enhance_stream` on the whole signal with the same statistics and scale to rounding (1.5e-6: the
    split-f16 transforms scale each 64-frame tile by its own power of two, and the blocks tile the frames differently)."""

    def __init__(self, model: TSCNet, stats: torch.Tensor, window: int = 400, context: int = 40, lookahead: int = 40,
                 scale: float | torch.Tensor = 1.0, graph: bool = True):
        self.eng = model.engine
        self.hop, self.n_fft = self.eng.cfg.hop, self.eng.cfg.n_fft
        if self.n_fft != 4 * self.hop:
            raise ValueError("StreamingEnhancer assumes n_fft = 4 hop (400 / 100, 1200 / 300)")
        dev = self.eng.device
        self.c = torch.as_tensor(scale, dtype=torch.float32, device=dev).reshape(1)
        self.state = StreamState(model, stats, 1, window, context, lookahead, graph)
        self.W, self.La = window, lookahead
        self.samples = torch.empty(0, device=dev)      # received samples from index `s_lo` on
        self.s_lo = 0
        self.n_in = 0                                  # samples received
        self.t_spec = 0                                # spectrogram frames computed
        self.spec_pend = torch.empty(1, 2, 0, self.eng.F, device=dev)   # computed, not yet fed to the network
        self.est_r = torch.empty(1, 1, 0, self.eng.F, device=dev)       # estimated frames from `f_lo` on
        self.est_i = torch.empty(1, 1, 0, self.eng.F, device=dev)
        self.f_lo = 0
        self.t_est = 0                                 # estimated frames
        self.n_out = 0                                 # samples emitted
        self.closed = False

    # ---- STFT of the frames whose samples have arrived ----
    def _advance_spec(self, final: bool):
        hop = self.hop
        t_b = (self.n_in // hop + 1) if final else (self.n_in - 2 * hop) // hop + 1      # frames [0, t_b) are computable
        t_b = max(t_b, 0)
        if t_b <= self.t_spec:
            return
        t_a = self.t_spec
        g0 = max(t_a - 2, 0)                                            # two guard frames on the left (none at the start)
        lo = g0 * hop
        hi = self.n_in if final else (t_b + 1) * hop                    # ... and the samples of two on the right
        if hi - lo <= self.n_fft // 2:                                  # (the very first samples: not enough for one block yet)
            return
        x = self.samples[lo - self.s_lo:hi - self.s_lo].reshape(1, -1).contiguous()
        spec = self.eng.stft_compress(x, self.c)
        self.spec_pend = torch.cat([self.spec_pend, spec[:, :, t_a - g0:t_b - g0]], dim=2)
        self.t_spec = t_b
        keep = max((t_b - 2) * hop - 2 * hop, 0)                        # samples the next block still needs
        if keep > self.s_lo:
            self.samples, self.s_lo = self.samples[keep - self.s_lo:].contiguous(), keep

    # ---- network steps for every window whose look-ahead is covered ----
    def _advance_net(self, final: bool):
        st = self.state
        while True:
            want = (st.k + 1) * self.W + self.La
            have = st.e1 + self.spec_pend.size(2)
            # a step runs once at least one frame BEYOND its look-ahead exists (then it is not the stream's last step) or
            # the stream has ended (then the step that takes the remaining frames is the last one and also emits what
            # is left past its window): exactly enhance_stream's rule `last = (upto == T)`
            if have > want:
                n, last = want - st.e1, False
            elif final and self.spec_pend.size(2) > 0:
                n, last = self.spec_pend.size(2), True
            else:
                return
            r, i = st.step(self.spec_pend[:, :, :n].contiguous(), last=last)
            self.spec_pend = self.spec_pend[:, :, n:]
            self.est_r, self.est_i = torch.cat([self.est_r, r], dim=2), torch.cat([self.est_i, i], dim=2)
            self.t_est += r.size(2)
            if last:
                return

    # ---- ISTFT of the samples whose four frames exist ----
    def _advance_out(self, final: bool) -> torch.Tensor:
        hop = self.hop
        end = (self.t_est - 1) * hop if final else (self.t_est - 2) * hop      # samples [n_out, end) are final
        if end <= self.n_out or self.t_est - self.f_lo < 2:
            return torch.empty(0, device=self.eng.device)
        f0 = max(self.n_out // hop - 2, 0)                                      # two guard frames on the left
        r = self.est_r[:, :, f0 - self.f_lo:].contiguous()
        i = self.est_i[:, :, f0 - self.f_lo:].contiguous()
        wav = self.eng.uncompress_istft(r, i)[0]                                # samples [f0 hop, (t_est - 1) hop)
        out = wav[self.n_out - f0 * hop:end - f0 * hop] / self.c
        self.n_out = end
        keep = max(end // hop - 2, 0)
        if keep > self.f_lo:
            self.est_r, self.est_i = self.est_r[:, :, keep - self.f_lo:], self.est_i[:, :, keep - self.f_lo:]
            self.f_lo = keep
        return out

    @torch.no_grad()
    def push(self, samples: torch.Tensor) -> torch.Tensor:
        """samples: float32 GPU tensor, any shape (flattened) -> the enhanced samples that became final (maybe none)."""
        if self.closed:
            raise RuntimeError("stream already flushed")
        x = samples.reshape(-1).to(device=self.eng.device, dtype=torch.float32)
        self.samples = torch.cat([self.samples, x])
        self.n_in += x.numel()
        self._advance_spec(False)
        self._advance_net(False)
        return self._advance_out(False)

    @torch.no_grad()
    def flush(self) -> torch.Tensor:
        """End of the stream (its length must be a multiple of hop, like one reference row): the remaining samples."""
        if self.closed:
            return torch.empty(0, device=self.eng.device)
        if self.n_in % self.hop or self.n_in <= self.n_fft // 2:
            raise ValueError("a stream must end on a multiple of hop and be longer than n_fft / 2")
        self.closed = True
        self._advance_spec(True)
        self._advance_net(True)
        return self._advance_out(True)
