"""Thin host wrapper over one libcmgan_hip handle.

PyTorch is used only for device memory and streams: every method takes CUDA (ROCm)
``torch.Tensor``s, passes their ``data_ptr()`` through the C ABI, and enqueues on
``torch.cuda.current_stream()``.  No arithmetic happens in Python.
"""
from __future__ import annotations

import ctypes
from typing import Optional

import numpy as np
import torch

from . import _lib
from ._lib import Config, KernelTime, Taps, check


def _f32c(t: torch.Tensor, name: str, device: Optional[torch.device] = None) -> torch.Tensor:
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{name}: expected a torch.Tensor")
    if not t.is_cuda:
        raise RuntimeError(f"{name}: tensor must live on the GPU (cmgan_amd has no CPU path)")
    if t.dtype != torch.float32:
        raise TypeError(f"{name}: expected float32, got {t.dtype}")
    if device is not None and t.device != device:
        raise RuntimeError(f"{name}: tensor is on {t.device} but this Engine is bound to {device} "
                           "(one handle = one device; build one Engine per GPU)")
    return t.contiguous()


def _on_device(fn):
    """Run an Engine method with the Engine's device current, so the launch stream, the workspace and the
    handle's own buffers all belong to the same GPU even when another device is current in the caller."""
    import functools

    @functools.wraps(fn)
    def wrapper(self, *a, **k):
        with torch.cuda.device(self.device):
            return fn(self, *a, **k)
    return wrapper


#: kernel families (names of _lib.MIX) that run ONE fp16 product in mfma_mode="f16mix".  The ablation
#: (tools/mix_ablation.py -> profiles/r06_mix_ablation.json, 32 x 2 s clips + three real recordings, error vs the F16X3
#: output): alone, every conformer family costs 1e-5 .. 7e-5 and the dense / sub-pixel convs 6e-4 .. 1.4e-3 - F16X1's whole
#: error is the convs.  All six conformer families together: 5.9e-5 (batch) / 1.14e-4 (recordings) for 18.1 vs 21.1 ms.
F16MIX_PRESET = ("ff1", "ff2", "qkv", "attn", "pw1", "dwpw2")


class Engine:
    """One handle = one device = one set of weights.  Not re-entrant: the workspace is shared by all calls,
    so use an Engine from ONE stream at a time (the stream current on its device when a method is called)."""

    def __init__(self, n_fft: int = 400, hop: int = 100, num_features: Optional[int] = None,
                 num_tscb: int = 4, max_pos_emb: int = 512, device: Optional[torch.device] = None,
                 mfma_mode: Optional[str] = None, mix_single=None):
        """mfma_mode: "f16x3" (default; fp32-accurate split products on the f16 matrix pipe), "f32" (bit-exact fp32
        MFMA), "f16x1" (REDUCED precision, opt-in: one fp16 product per contraction in the TSCNet body, 6e-4 .. 9e-4 of
        the peak vs the reference - inside the 1e-3 gate without margin, 200x the default mode's error) or "f16mix"
        (REDUCED precision, opt-in, WITH margin: only the kernel families in `mix_single` - names of _lib.MIX, default
        F16MIX_PRESET - run one product, the rest three; <= 2e-4 end to end) - see include/cmgan_hip.h."""
        self.lib = _lib.load()
        if not torch.cuda.is_available():
            raise RuntimeError("cmgan_amd needs a ROCm GPU: torch.cuda.is_available() is False")
        self.device = torch.device(device if device is not None else f"cuda:{torch.cuda.current_device()}")
        if self.device.type != "cuda":
            raise RuntimeError(f"cmgan_amd has no CPU path: device must be a GPU, got {self.device}")
        if self.device.index is None:
            self.device = torch.device(f"cuda:{torch.cuda.current_device()}")
        cfg = _lib.default_config()
        cfg.n_fft, cfg.hop = n_fft, hop
        cfg.num_features = num_features if num_features is not None else n_fft // 2 + 1
        cfg.num_tscb, cfg.max_pos_emb = num_tscb, max_pos_emb
        if mfma_mode is not None:
            modes = {"f32": _lib.MFMA_F32, "f16x3": _lib.MFMA_F16X3, "f16x1": _lib.MFMA_F16X1, "f16mix": _lib.MFMA_F16MIX}
            if mfma_mode not in modes:
                raise ValueError(f"mfma_mode must be one of {sorted(modes)}")
            cfg.mfma_mode = modes[mfma_mode]
        if mix_single is not None and cfg.mfma_mode != _lib.MFMA_F16MIX:
            raise ValueError("mix_single only applies to mfma_mode='f16mix'")
        if cfg.mfma_mode == _lib.MFMA_F16MIX:
            fams = F16MIX_PRESET if mix_single is None else tuple(mix_single)
            unknown = [f for f in fams if f not in _lib.MIX]
            if unknown:
                raise ValueError(f"unknown kernel families {unknown}: choose from {sorted(_lib.MIX)}")
            cfg.single_mask = sum(_lib.MIX[f] for f in set(fams))
            self.mix_single = tuple(sorted(set(fams)))
        self.mfma_mode = {_lib.MFMA_F32: "f32", _lib.MFMA_F16X3: "f16x3", _lib.MFMA_F16X1: "f16x1",
                          _lib.MFMA_F16MIX: "f16mix"}[cfg.mfma_mode]
        self.cfg = cfg
        self._h = ctypes.c_void_p()
        with torch.cuda.device(self.device):
            rc = self.lib.cmgan_create(ctypes.byref(self._h), ctypes.byref(cfg))
        if rc != 0:
            msg = self.lib.cmgan_last_error(None)
            raise _lib.CmganError(rc, msg.decode() if msg else "?")
        self._ws: Optional[torch.Tensor] = None
        self._cws: Optional[torch.Tensor] = None
        self._bws: Optional[torch.Tensor] = None          # workspace of the two-branch form (enhance_graphed(branches=2))
        self._graphs: dict = {}          # enhance_graphed: input shape -> (graph, in, out, token)
        self._row_graphs: dict = {}      # streaming.enhance_windows: row shape -> (graph, in, out, token)
        self._stream_slots: dict = {}    # streaming.StreamState(graph=True): (B, W, Ca, La) -> state buffers + step graphs
        self.weights_loaded = False

    def __del__(self):
        h = getattr(self, "_h", None)
        if h is not None and h.value:
            try:
                self.lib.cmgan_destroy(h)
            except Exception:
                pass
            self._h = ctypes.c_void_p()

    # ---- weights -------------------------------------------------------------------
    def load_blob(self, blob: np.ndarray):
        blob = np.ascontiguousarray(blob, dtype=np.uint8)
        # cmgan_load_weights re-allocates the device weight buffers: every captured graph holds kernel arguments
        # pointing into the old ones, so they are dropped BEFORE the swap (a failed load keeps the old weights,
        # and the graphs are simply re-captured on next use)
        self._graphs.clear()
        self._row_graphs.clear()
        self._stream_slots.clear()
        with torch.cuda.device(self.device):
            check(self._h, self.lib.cmgan_load_weights(self._h, blob.ctypes.data_as(ctypes.c_void_p), blob.nbytes))
        self.weights_loaded = True

    def _stream(self) -> int:
        return torch.cuda.current_stream(self.device).cuda_stream

    def _in(self, t: torch.Tensor, name: str) -> torch.Tensor:
        return _f32c(t, name, self.device)

    # ---- workspace -----------------------------------------------------------------
    def _ws_token(self):
        """Identity of what a captured graph's kernel arguments point into: the workspace allocation and the
        weight buffers (cmgan_weights_generation).  A graph whose token differs is stale and is re-captured."""
        ws = self._ws
        return (ws.data_ptr() if ws is not None else 0, ws.numel() if ws is not None else 0,
                int(self.lib.cmgan_weights_generation(self._h)))

    def _workspace(self, B: int, T: int) -> torch.Tensor:
        need = self.lib.cmgan_workspace_bytes(self._h, B, T)
        if need == 0:
            raise ValueError(f"bad batch/frames ({B}, {T})")
        if self._ws is None or self._ws.numel() < need:
            # graphs captured on the old allocation are stale AND keep it (multi-GB) alive: evict them all
            self._graphs = {k: v for k, v in self._graphs.items() if k[1] != 1}
            self._row_graphs.clear()
            for slot in self._stream_slots.values():
                slot.graphs.clear()
                slot.stage_graphs.clear()
            self._ws = None
            self._ws = torch.empty(need, dtype=torch.uint8, device=self.device)
        return self._ws

    def workspace_bytes(self, B: int, T: int) -> int:
        return int(self.lib.cmgan_workspace_bytes(self._h, B, T))

    def _conf_workspace(self, N: int, L: int) -> torch.Tensor:
        need = self.lib.cmgan_conformer_workspace_bytes(self._h, N, L)
        if self._cws is None or self._cws.numel() < need:
            self._cws = None
            self._cws = torch.empty(need, dtype=torch.uint8, device=self.device)
        return self._cws

    @property
    def F(self) -> int:
        return self.cfg.num_features

    def num_frames(self, L: int) -> int:
        return L // self.cfg.hop + 1

    # ---- front / back end ----------------------------------------------------------
    @_on_device
    def rms_scale(self, wav: torch.Tensor) -> torch.Tensor:
        wav = self._in(wav, "wav")
        B, L = wav.shape
        out = torch.empty(B, dtype=torch.float32, device=wav.device)
        check(self._h, self.lib.cmgan_rms_scale(self._h, wav.data_ptr(), B, L, out.data_ptr(), self._stream()))
        return out

    @_on_device
    def stft_compress(self, wav: torch.Tensor, scale: Optional[torch.Tensor] = None) -> torch.Tensor:
        wav = self._in(wav, "wav")
        B, L = wav.shape
        sp = self._in(scale, "scale").data_ptr() if scale is not None else None
        out = torch.empty(B, 2, self.num_frames(L), self.F, dtype=torch.float32, device=wav.device)
        check(self._h, self.lib.cmgan_stft_compress(self._h, wav.data_ptr(), sp, B, L, out.data_ptr(), self._stream()))
        return out

    @_on_device
    def uncompress_istft(self, real: torch.Tensor, imag: torch.Tensor,
                         scale: Optional[torch.Tensor] = None) -> torch.Tensor:
        real, imag = self._in(real, "real"), self._in(imag, "imag")
        B, _, T, F = real.shape
        if F != self.F or imag.shape != real.shape:
            raise ValueError(f"expected 2 x [B,1,T,{self.F}], got {tuple(real.shape)} / {tuple(imag.shape)}")
        ws = self._workspace(B, T)
        sp = self._in(scale, "scale").data_ptr() if scale is not None else None
        out = torch.empty(B, self.cfg.hop * (T - 1), dtype=torch.float32, device=real.device)
        check(self._h, self.lib.cmgan_uncompress_istft(self._h, real.data_ptr(), imag.data_ptr(), sp, B, T,
                                                       out.data_ptr(), ws.data_ptr(), ws.numel(), self._stream()))
        return out

    @_on_device
    def power_compress(self, x: torch.Tensor) -> torch.Tensor:
        x = self._in(x, "x")
        B, F, T, two = x.shape
        assert two == 2
        y = torch.empty(B, 2, F, T, dtype=torch.float32, device=x.device)
        check(self._h, self.lib.cmgan_power_compress(self._h, x.data_ptr(), B, F, T, y.data_ptr(), self._stream()))
        return y

    @_on_device
    def power_uncompress(self, real: torch.Tensor, imag: torch.Tensor) -> torch.Tensor:
        real, imag = self._in(real, "real"), self._in(imag, "imag")
        B, one, F, T = real.shape
        y = torch.empty(B, 1, F, T, 2, dtype=torch.float32, device=real.device)
        check(self._h, self.lib.cmgan_power_uncompress(self._h, real.data_ptr(), imag.data_ptr(), B, F, T,
                                                       y.data_ptr(), self._stream()))
        return y

    # ---- model ---------------------------------------------------------------------
    def _need_weights(self):
        if not self.weights_loaded:
            raise RuntimeError("no weights loaded: call load_state_dict() first")

    @_on_device
    def tscnet_forward(self, x: torch.Tensor, taps: bool = False):
        self._need_weights()
        x = self._in(x, "x")
        B, two, T, F = x.shape
        if two != 2 or F != self.F:
            raise ValueError(f"expected [B,2,T,{self.F}], got {tuple(x.shape)}")
        ws = self._workspace(B, T)
        real = torch.empty(B, 1, T, F, dtype=torch.float32, device=x.device)
        imag = torch.empty_like(real)
        if not taps:
            check(self._h, self.lib.cmgan_tscnet_forward(self._h, x.data_ptr(), B, T, real.data_ptr(),
                                                         imag.data_ptr(), ws.data_ptr(), ws.numel(), self._stream()))
            return real, imag
        F2 = (F + 1) // 2
        st = {"encoder": torch.empty(B, 64, T, F2, dtype=torch.float32, device=x.device),
              "mask": torch.empty(B, 1, T, F, dtype=torch.float32, device=x.device),
              "complex": torch.empty(B, 2, T, F, dtype=torch.float32, device=x.device)}
        tp = Taps()
        tp.encoder_dev = st["encoder"].data_ptr()
        for k in range(self.cfg.num_tscb):
            st[f"tscb{k + 1}"] = torch.empty(B, 64, T, F2, dtype=torch.float32, device=x.device)
            tp.tscb_dev[k] = st[f"tscb{k + 1}"].data_ptr()
        tp.mask_dev, tp.complex_dev = st["mask"].data_ptr(), st["complex"].data_ptr()
        check(self._h, self.lib.cmgan_tscnet_forward_taps(self._h, x.data_ptr(), B, T, real.data_ptr(),
                                                          imag.data_ptr(), ctypes.byref(tp), ws.data_ptr(),
                                                          ws.numel(), self._stream()))
        return real, imag, st

    # ---- frozen-statistics / sliced forms (streaming with carried state; include/cmgan_hip.h, "Streaming") ----
    def stats_floats(self, B: int) -> int:
        return int(self.lib.cmgan_stats_floats(self._h, B))

    @_on_device
    def tscnet_forward_stats(self, x: torch.Tensor, frozen: Optional[torch.Tensor] = None, export: bool = True):
        """TSCNet.forward with every InstanceNorm on the `frozen` statistics blob (None = the tensor's own, i.e. the
        reference's arithmetic).  Returns (real, imag, blob of the statistics this call used or None)."""
        self._need_weights()
        x = self._in(x, "x")
        B, two, T, F = x.shape
        if two != 2 or F != self.F:
            raise ValueError(f"expected [B,2,T,{self.F}], got {tuple(x.shape)}")
        ws = self._workspace(B, T)
        real = torch.empty(B, 1, T, F, dtype=torch.float32, device=x.device)
        imag = torch.empty_like(real)
        fz = self._stats_arg(frozen, B)
        out = torch.empty(self.stats_floats(B), dtype=torch.float32, device=x.device) if export else None
        check(self._h, self.lib.cmgan_tscnet_forward_stats(self._h, x.data_ptr(), B, T, real.data_ptr(), imag.data_ptr(),
                                                           fz, out.data_ptr() if export else None, ws.data_ptr(),
                                                           ws.numel(), self._stream()))
        return real, imag, out

    def _stats_arg(self, stats: Optional[torch.Tensor], B: int):
        if stats is None:
            return None
        stats = self._in(stats, "stats")
        if stats.numel() != self.stats_floats(B):
            raise ValueError(f"statistics blob has {stats.numel()} floats, a batch of {B} rows needs {self.stats_floats(B)}")
        return stats.data_ptr()

    def _own_ws(self, ws: Optional[torch.Tensor], B: int, T: int) -> torch.Tensor:
        if ws is None:
            return self._workspace(B, T)
        if not ws.is_cuda or ws.dtype != torch.uint8 or ws.numel() < self.workspace_bytes(B, T):
            raise ValueError("ws must be a uint8 GPU tensor of at least workspace_bytes(B, T) bytes")
        return ws

    @_on_device
    def stream_encoder(self, spec: torch.Tensor, stats: torch.Tensor, out: Optional[torch.Tensor] = None,
                       ws: Optional[torch.Tensor] = None) -> torch.Tensor:
        """dense_encoder under frozen statistics: spec [B,2,T,F] -> x [B,T,F',64] (channels-last).  ws: a workspace of the
        caller's instead of the engine's (lets the call run on another stream beside a step's TSCBs)."""
        self._need_weights()
        spec = self._in(spec, "spec")
        B, two, T, F = spec.shape
        if two != 2 or F != self.F:
            raise ValueError(f"expected [B,2,T,{self.F}], got {tuple(spec.shape)}")
        ws = self._own_ws(ws, B, T)
        want = (B, T, (F + 1) // 2, 64)
        if out is None:
            out = torch.empty(want, dtype=torch.float32, device=spec.device)
        elif not out.is_contiguous() or tuple(out.shape) != want:
            raise ValueError(f"out must be a contiguous [B,T,F',64] = {want} tensor (the kernel writes it in place), "
                             f"got {tuple(out.shape)}{'' if out.is_contiguous() else ' (non-contiguous)'}")
        check(self._h, self.lib.cmgan_stream_encoder(self._h, spec.data_ptr(), B, T, self._stats_arg(stats, B),
                                                     self._in(out, "out").data_ptr(), ws.data_ptr(), ws.numel(), self._stream()))
        return out

    @_on_device
    def stream_tscb(self, x: torch.Tensor) -> torch.Tensor:
        """TSCB_1..4 on x [B,T,F',64], IN PLACE (returns x)."""
        self._need_weights()
        if not x.is_contiguous():
            raise ValueError("stream_tscb works in place: x must be contiguous")
        x = self._in(x, "x")
        B, T, F2, C = x.shape
        if C != 64 or F2 != (self.F + 1) // 2:
            raise ValueError(f"expected [B,T,{(self.F + 1) // 2},64], got {tuple(x.shape)}")
        ws = self._workspace(B, T)
        check(self._h, self.lib.cmgan_stream_tscb(self._h, x.data_ptr(), B, T, ws.data_ptr(), ws.numel(), self._stream()))
        return x

    @_on_device
    def stream_decoder(self, x: torch.Tensor, spec: torch.Tensor, stats: torch.Tensor, ws: Optional[torch.Tensor] = None):
        """mask + complex decoder + recombination under frozen statistics: x [B,T,F',64], spec [B,2,T,F] ->
        (est_real, est_imag) [B,1,T,F].  ws: a workspace of the caller's (uint8, >= workspace_bytes(B, T)) instead of the
        engine's - what lets a decoder call run on another stream BESIDE the next step's encoder / TSCBs."""
        self._need_weights()
        x, spec = self._in(x, "x"), self._in(spec, "spec")
        B, T, F2, C = x.shape
        if tuple(spec.shape) != (B, 2, T, self.F) or C != 64 or F2 != (self.F + 1) // 2:
            raise ValueError(f"expected x [B,T,{(self.F + 1) // 2},64] and spec [B,2,T,{self.F}]")
        ws = self._own_ws(ws, B, T)
        real = torch.empty(B, 1, T, self.F, dtype=torch.float32, device=x.device)
        imag = torch.empty_like(real)
        check(self._h, self.lib.cmgan_stream_decoder(self._h, x.data_ptr(), spec.data_ptr(), B, T, self._stats_arg(stats, B),
                                                     real.data_ptr(), imag.data_ptr(), ws.data_ptr(), ws.numel(),
                                                     self._stream()))
        return real, imag

    @_on_device
    def conformer_forward(self, index: int, x: torch.Tensor, taps: bool = False, mask: Optional[torch.Tensor] = None):
        """mask: [N, L] bool (or any integer / float tensor, non-zero = keep) - ConformerBlock.forward(x, mask)."""
        self._need_weights()
        x = self._in(x, "x")
        N, L, C = x.shape
        if C != 64:
            raise ValueError("conformer dim must be 64")
        ws = self._conf_workspace(N, L)
        y = torch.empty_like(x)
        tp = torch.empty(4, N, L, 64, dtype=torch.float32, device=x.device) if taps else None
        if mask is None:
            check(self._h, self.lib.cmgan_conformer_forward(self._h, index, x.data_ptr(), N, L, y.data_ptr(),
                                                            tp.data_ptr() if taps else None, ws.data_ptr(),
                                                            ws.numel(), self._stream()))
        else:
            if tuple(mask.shape) != (N, L):
                raise ValueError(f"mask must be [N, L] = [{N}, {L}], got {tuple(mask.shape)}")
            if mask.device != x.device:
                raise ValueError(f"mask is on {mask.device}, x on {x.device}")
            mk = (mask != 0).to(torch.uint8).contiguous()
            check(self._h, self.lib.cmgan_conformer_forward_masked(self._h, index, x.data_ptr(), N, L, mk.data_ptr(),
                                                                   y.data_ptr(), tp.data_ptr() if taps else None,
                                                                   ws.data_ptr(), ws.numel(), self._stream()))
        return (y, tp) if taps else y

    @_on_device
    def enhance(self, wav: torch.Tensor) -> torch.Tensor:
        """wav[B,L] -> enhanced[B,L]: the whole device pipeline in one ABI call."""
        self._need_weights()
        wav = self._in(wav, "wav")
        B, L = wav.shape
        ws = self._workspace(B, self.num_frames(L))
        out = torch.empty_like(wav)
        check(self._h, self.lib.cmgan_enhance(self._h, wav.data_ptr(), B, L, out.data_ptr(), ws.data_ptr(),
                                              ws.numel(), self._stream()))
        return out

    # ---- hipGraph replay of the whole pipeline -------------------------------------------
    #: enhance_graphed's defaults (overridable per call and through CMGAN_BRANCHES / CMGAN_BRANCH_OFFSET): part-batch
    #: branches the captured graph holds, and the launches branch i issues before branch i + 1 starts.  Measured in one
    #: session at 32 x 2 s (profiles/r06_branch_sweep.txt): 1 branch 21.7 ms, 2 branches together 21.0 - 21.1 ms (-3 %),
    #: offsets of 12 / 24 / 60 launches 21.45 / 21.7 / 22.3 ms: what a second branch buys is the other branch's
    #: workgroups in every kernel's ramp-up and drain, not unlike kernels side by side.
    BRANCHES = 2
    BRANCH_OFFSET = 0
    MAX_BRANCHES = 8

    def _enhance_call(self, wav, out, B, L, branches: int, offset: int):
        if branches == 1:
            ws = self._workspace(B, self.num_frames(L))
            check(self._h, self.lib.cmgan_enhance(self._h, wav.data_ptr(), B, L, out.data_ptr(), ws.data_ptr(),
                                                  ws.numel(), self._stream()))
        else:
            ws = self._branched_workspace(B, self.num_frames(L), branches)
            check(self._h, self.lib.cmgan_enhance_branched(self._h, wav.data_ptr(), B, L, out.data_ptr(), ws.data_ptr(),
                                                           ws.numel(), self._stream(), branches, offset))

    def _branched_workspace(self, B: int, T: int, branches: int) -> torch.Tensor:
        need = self.lib.cmgan_workspace_bytes_branched(self._h, B, T, branches)
        if need == 0:
            raise ValueError(f"bad batch / frames / branches ({B}, {T}, {branches})")
        if self._bws is None or self._bws.numel() < need:
            self._graphs = {k: v for k, v in self._graphs.items() if k[1] == 1}
            self._bws = None
            self._bws = torch.empty(need, dtype=torch.uint8, device=self.device)
        return self._bws

    def _branch_args(self, B: int, branches: Optional[int], offset: Optional[int]):
        import os
        if branches is None:
            branches = int(os.environ.get("CMGAN_BRANCHES", self.BRANCHES))
        if offset is None:
            offset = int(os.environ.get("CMGAN_BRANCH_OFFSET", self.BRANCH_OFFSET))
        if not 1 <= branches <= self.MAX_BRANCHES:
            raise ValueError(f"branches must be 1 .. {self.MAX_BRANCHES}")
        if offset < 0:
            raise ValueError("branch offset must be >= 0")
        branches = min(branches, B)
        return branches, (offset if branches > 1 else 0)

    @_on_device
    def enhance_branched(self, wav: torch.Tensor, branches: int = 2, offset: int = 0) -> torch.Tensor:
        """enhance() as `branches` part-batch branches on as many streams (cmgan_enhance_branched); same result bit for bit."""
        self._need_weights()
        wav = self._in(wav, "wav")
        B, L = wav.shape
        out = torch.empty_like(wav)
        branches, offset = self._branch_args(B, branches, offset)
        self._enhance_call(wav, out, B, L, branches, offset)
        return out

    @_on_device
    def enhance_graphed(self, wav: torch.Tensor, branches: Optional[int] = None, offset: Optional[int] = None) -> torch.Tensor:
        """Same result as enhance(), but the ~250 kernel launches of cmgan_enhance are captured once
        per input shape into a hipGraph (torch.cuda.CUDAGraph on the capture stream) and replayed:
        the C ABI never allocates or synchronises, so it is capturable as is.  The returned tensor is
        a static buffer that the next call with the same shape overwrites.
        branches > 1: the graph holds that many part-batch branches (cmgan_enhance_branched) as parallel paths the GPU
        overlaps; the result is the same bit for bit (rows are independent).  Defaults: BRANCHES / BRANCH_OFFSET."""
        self._need_weights()
        wav = self._in(wav, "wav")
        B, L = wav.shape
        branches, offset = self._branch_args(B, branches, offset)
        key = (tuple(wav.shape), branches, offset)
        # the workspace may grow (and evict every graph that points into it) first
        T = self.num_frames(L)
        ws = self._workspace(B, T) if branches == 1 else self._branched_workspace(B, T, branches)
        token = (ws.data_ptr(), ws.numel(), int(self.lib.cmgan_weights_generation(self._h)))
        ent = self._graphs.get(key)
        if ent is not None and ent[3] != token:                 # workspace or weights changed since capture
            ent = None
        if ent is None:
            g_in, g_out = torch.empty_like(wav), torch.empty_like(wav)
            g_in.copy_(wav)
            side = torch.cuda.Stream(device=self.device)
            side.wait_stream(torch.cuda.current_stream(self.device))
            with torch.cuda.stream(side):                       # warm-up outside capture
                self._enhance_call(g_in, g_out, B, L, branches, offset)
            torch.cuda.current_stream(self.device).wait_stream(side)
            torch.cuda.synchronize(self.device)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                self._enhance_call(g_in, g_out, B, L, branches, offset)
            ent = (graph, g_in, g_out, token)
            self._graphs[key] = ent
        graph, g_in, g_out, _ = ent
        g_in.copy_(wav)
        graph.replay()
        return g_out

    # ---- training / validation step pieces (src/train.py) ----------------------------------
    @_on_device
    def loss_terms(self, est_real=None, est_imag=None, clean_spec=None, est_audio=None, clean_audio=None,
                   out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """{loss_ri, loss_mag, time_loss, time_mse} of Trainer.calculate_generator_loss (train.py:124-151) as a
        float32[4] device tensor (deterministic two-pass reduction).  Spectral operands in the model layout:
        est_real/est_imag [B,1,T,F], clean_spec [B,2,T,F]; audio [B,L].  Either group may be omitted."""
        spec = [est_real, est_imag, clean_spec]
        audio = [est_audio, clean_audio]
        if any(t is None for t in spec) and not all(t is None for t in spec):
            raise ValueError("spectral terms need est_real, est_imag and clean_spec together")
        if any(t is None for t in audio) and not all(t is None for t in audio):
            raise ValueError("the time term needs est_audio and clean_audio together")
        B = T = L = 0
        ptr = [None] * 5
        if spec[0] is not None:
            er, ei, cs = (self._in(t, n) for t, n in zip(spec, ("est_real", "est_imag", "clean_spec")))
            B, _, T, F = er.shape
            if F != self.F or ei.shape != er.shape or tuple(cs.shape) != (B, 2, T, F):
                raise ValueError(f"expected est [B,1,T,{self.F}] x2 and clean_spec [B,2,T,{self.F}]")
            ptr[0:3] = [er.data_ptr(), ei.data_ptr(), cs.data_ptr()]
        if audio[0] is not None:
            ea, ca = self._in(audio[0], "est_audio"), self._in(audio[1], "clean_audio")
            if ea.shape != ca.shape or ea.dim() != 2 or (B and ea.size(0) != B):
                raise ValueError("est_audio / clean_audio must both be [B, L]")
            B, L = ea.shape
            ptr[3:5] = [ea.data_ptr(), ca.data_ptr()]
        if out is None:
            out = torch.empty(4, dtype=torch.float32, device=self.device)
        check(self._h, self.lib.cmgan_loss_terms(self._h, ptr[0], ptr[1], ptr[2], B, T, ptr[3], ptr[4], L,
                                                 self._in(out, "out").data_ptr(), self._stream()))
        return out

    # ---- diagnostics ---------------------------------------------------------------
    @_on_device
    def selftest_mfma(self) -> float:
        err = ctypes.c_float()
        check(self._h, self.lib.cmgan_selftest_mfma(self._h, ctypes.byref(err)))
        return float(err.value)

    @_on_device
    def selftest_mfma_x3(self) -> float:
        err = ctypes.c_float()
        check(self._h, self.lib.cmgan_selftest_mfma_x3(self._h, ctypes.byref(err)))
        return float(err.value)

    def release_workspaces(self):
        """Drop the inference workspaces and every captured graph that points into them (they are re-created on demand):
        ~6 GB each at 32 x 2 s.  bench.py calls this before its training leg."""
        self._graphs.clear()
        self._row_graphs.clear()
        self._stream_slots.clear()
        self._ws = self._bws = self._cws = None

    def set_profiling(self, on: bool):
        check(self._h, self.lib.cmgan_set_profiling(self._h, 1 if on else 0))

    def profile(self):
        """[(kernel name, ms)] of the most recent forward (profiling must be on)."""
        cap = 1024
        buf = (KernelTime * cap)()
        n = self.lib.cmgan_profile_read(self._h, buf, cap)
        return [(buf[i].name.decode(), float(buf[i].ms)) for i in range(n)]
