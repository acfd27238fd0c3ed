"""CPU evaluation of the PACKED weight blob, following the HIP kernels' restructured
arithmetic step by step (folded LayerNorms / BatchNorm / scales, slot-ordered dense
blocks, normalise-on-load InstanceNorm, per-position tail projections summed across
neighbours, mask*x + complex).  Test infrastructure: it lets the CPU suite prove that
packer.py + the algebraic restructuring reproduce the oracle before any kernel runs,
so a GPU mismatch can only come from kernel indexing.
"""
from __future__ import annotations

import struct

import numpy as np
import torch
import torch.nn.functional as F

from cmgan_amd import packer as P

EPS = 1e-5


def parse_blob(blob: np.ndarray) -> dict:
    raw = blob.tobytes()
    magic, ver, n, payload = struct.unpack_from("<4I", raw, 0)
    assert magic == P.MAGIC and ver == P.VERSION
    data = np.frombuffer(raw, dtype=np.float32, offset=16 + 16 * n, count=payload)
    out = {}
    for i in range(n):
        wid, off, cnt, _ = struct.unpack_from("<4I", raw, 16 + 16 * i)
        assert off % 64 == 0
        out[wid] = torch.from_numpy(data[off:off + cnt].copy())
    return out


def unfm(flat: torch.Tensor, rows: int, cols: int) -> torch.Tensor:
    return flat.reshape(rows // 16, cols // 16, 4, 16, 4).permute(0, 3, 1, 2, 4).reshape(rows, cols)


def unconv(flat: torch.Tensor, cout: int, ci: int, nt: int) -> torch.Tensor:
    """[ci/16][nt*3][cout/16][64][4] -> W[cout, ci, nt, 3] (slot-ordered ci)."""
    w = torch.zeros(cout, ci, nt, 3)
    per = cout * 16
    idx = 0
    for chunk in range(ci // 16):
        for kt in range(nt):
            for kf in range(3):
                w[:, 16 * chunk:16 * chunk + 16, kt, kf] = unfm(flat[idx:idx + per], cout, 16)
                idx += per
    return w


def _xhat(x):
    mu = x.mean(-1, keepdim=True)
    var = ((x - mu) ** 2).mean(-1, keepdim=True)
    return (x - mu) * torch.rsqrt(var + EPS)


def _swish(x):
    return x * torch.sigmoid(x)


class BlobModel:
    def __init__(self, blob: np.ndarray, max_pos: int = 512):
        self.w = parse_blob(blob)
        self.max_pos = max_pos

    def g(self, group, item):
        return self.w[P.wid(group, item)]

    # ---- conformer --------------------------------------------------------------
    def ffn(self, grp, x, first: bool):
        iw1, ib1, iw2, ib2 = ((P.CF_FF1_W1, P.CF_FF1_B1, P.CF_FF1_W2, P.CF_FF1_B2) if first else
                              (P.CF_FF2_W1, P.CF_FF2_B1, P.CF_FF2_W2, P.CF_FF2_B2))
        w1, w2 = unfm(self.g(grp, iw1), 256, 64), unfm(self.g(grp, iw2), 64, 256)
        hp = _xhat(x) @ w1.t() + self.g(grp, ib1)        # h' = -log2(e) h  (scale folded by the packer)
        return x + (hp / (1.0 + torch.exp2(hp))) @ w2.t() + self.g(grp, ib2)

    def attn(self, grp, x):
        n = x.shape[1]
        wqkv = unfm(self.g(grp, P.CF_QKV_W), 192, 64)
        qkv = _xhat(x) @ wqkv.t() + self.g(grp, P.CF_QKV_B)
        q, k, v = [t.reshape(x.shape[0], n, 4, 16).transpose(1, 2) for t in qkv.split(64, dim=-1)]
        emb = self.g(grp, P.CF_REL).reshape(-1, 16)
        idx = torch.arange(n)
        rel = (idx[:, None] - idx[None, :]).clamp(-self.max_pos, self.max_pos) + self.max_pos
        s = q @ k.transpose(-1, -2) + torch.gather(q @ emb.t(), -1, rel.expand(x.shape[0], 4, n, n))
        # scores are in log2 units (log2(e) folded into the q projection)
        o = (torch.softmax(s * float(np.log(2.0)), -1) @ v).transpose(1, 2).reshape(x.shape[0], n, 64)
        return x + o @ unfm(self.g(grp, P.CF_WO), 64, 64).t() + self.g(grp, P.CF_BO)

    def convmod(self, grp, x):
        h = _xhat(x) @ unfm(self.g(grp, P.CF_PW1_W), 256, 64).t() + self.g(grp, P.CF_PW1_B)
        u = h[..., :128] * torch.sigmoid(h[..., 128:])
        taps = self.g(grp, P.CF_DW_W).reshape(31, 128)
        up = F.pad(u, (0, 0, 15, 15))
        v = sum(taps[t] * up[:, t:t + u.shape[1]] for t in range(31)) + self.g(grp, P.CF_DW_B)
        v = _swish(v)
        return x + v @ unfm(self.g(grp, P.CF_PW2_W), 64, 128).t() + self.g(grp, P.CF_PW2_B)

    def conformer(self, index, x, stages=None):
        grp = P.G_CONF0 + index
        x0 = x
        x = self.ffn(grp, x, True)
        if stages is not None: stages["ff1"] = x
        x = self.attn(grp, x)
        if stages is not None: stages["attn"] = x
        x = self.convmod(grp, x)
        if stages is not None: stages["conv"] = x
        x = self.ffn(grp, x, False)
        if stages is not None: stages["ff2"] = x
        gb = self.g(grp, P.CF_POST_GB)
        return _xhat(x) * gb[:64] + gb[64:], x0

    # ---- convs (channels-last [B,T,F,C]) -----------------------------------------
    @staticmethod
    def _stats(raw, gb):
        mu = raw.mean(dim=(1, 2), keepdim=True)
        var = (raw * raw).mean(dim=(1, 2), keepdim=True) - mu * mu
        sc = gb[:64] * torch.rsqrt(var + EPS)
        return sc, gb[64:] - mu * sc

    @staticmethod
    def _load(raw, sc, sh, alpha):
        if sc is None:
            return raw
        y = raw * sc + sh
        return torch.where(y >= 0, y, alpha * y)

    @staticmethod
    def _conv(inp_cl, w, bias, dil=1):
        """inp_cl [B,T,F,Ci]; w [Co,Ci,NT,3]; causal in time (top pad), same in freq."""
        x = inp_cl.permute(0, 3, 1, 2)
        nt = w.shape[2]
        x = F.pad(x, (1, 1, dil * (nt - 1), 0))
        return F.conv2d(x, w, bias, dilation=(dil, 1)).permute(0, 2, 3, 1)

    def dense_block(self, grp, x0, x0_norm):
        slots = [(x0,) + x0_norm]
        for i in range(4):
            inp = torch.cat([self._load(*s) for s in slots], dim=-1)
            w = unconv(self.g(grp, i * 4 + 0), 64, 64 * (i + 1), 2)
            raw = self._conv(inp, w, self.g(grp, i * 4 + 1), dil=2 ** i)
            sc, sh = self._stats(raw, self.g(grp, i * 4 + 2))
            slots.append((raw, sc, sh, self.g(grp, i * 4 + 3)))
        return slots[-1]

    def forward(self, spec, stages=None):
        re, im = spec[:, 0], spec[:, 1]                                  # [B,T,F]
        mag = torch.sqrt(re * re + im * im)
        c1 = self.g(P.G_ENC, P.ENC_C1_W).reshape(4, 64)
        raw = mag[..., None] * c1[0] + re[..., None] * c1[1] + im[..., None] * c1[2] + c1[3]
        sc, sh = self._stats(raw, self.g(P.G_ENC, P.ENC_C1_GB))
        last = self.dense_block(P.G_DB_E, raw, (sc, sh, self.g(P.G_ENC, P.ENC_C1_PRELU)))
        w2 = unconv(self.g(P.G_ENC, P.ENC_C2_W), 64, 64, 1)
        raw = self._conv(self._load(*last), w2, self.g(P.G_ENC, P.ENC_C2_BIAS))[:, :, 0::2]
        sc, sh = self._stats(raw, self.g(P.G_ENC, P.ENC_C2_GB))
        x = self._load(raw, sc, sh, self.g(P.G_ENC, P.ENC_C2_PRELU))      # [B,T,F2,64]
        if stages is not None: stages["encoder"] = x.permute(0, 3, 1, 2)
        B, T, F2, _ = x.shape
        for k in range(4):
            xt = x.permute(0, 2, 1, 3).reshape(B * F2, T, 64)
            y, x0 = self.conformer(2 * k, xt)
            x = (y + x0).reshape(B, F2, T, 64).permute(0, 2, 1, 3)
            xf = x.reshape(B * T, F2, 64)
            y, x0 = self.conformer(2 * k + 1, xf)
            x = (y + x0).reshape(B, T, F2, 64)
            if stages is not None: stages[f"tscb{k + 1}"] = x.permute(0, 3, 1, 2)

        def subpixel(last, wkey, bkey, grp):
            w = unconv(self.g(grp, wkey), 128, 64, 1)
            y = self._conv(self._load(*last), w, self.g(grp, bkey))       # [B,T,F2,128]
            return torch.stack([y[..., :64], y[..., 64:]], dim=3).reshape(B, T, 2 * F2, 64)

        # mask decoder
        sp = subpixel(self.dense_block(P.G_DB_M, x, (None, None, None)), P.MK_SP_W, P.MK_SP_BIAS, P.G_MASK)
        d = sp @ unfm(self.g(P.G_MASK, P.MK_TAIL_W), 16, 64)[:4].t()       # [B,T,W,4]
        sca = self.g(P.G_MASK, P.MK_SCALARS)
        m = d[:, :, :-1, 0] + d[:, :, 1:, 1] + sca[0]
        mu = m.mean(dim=(1, 2), keepdim=True)
        var = (m * m).mean(dim=(1, 2), keepdim=True) - mu * mu
        m = (m - mu) * torch.rsqrt(var + EPS) * sca[1] + sca[2]
        m = torch.where(m >= 0, m, sca[3] * m)
        m = m * sca[4] + sca[5]
        pout = self.g(P.G_MASK, P.MK_PRELU_OUT)
        mask = torch.where(m >= 0, m, pout * m)
        # complex decoder
        sp = subpixel(self.dense_block(P.G_DB_C, x, (None, None, None)), P.CX_SP_W, P.CX_SP_BIAS, P.G_CPLX)
        sc, sh = self._stats(sp, self.g(P.G_CPLX, P.CX_GB))
        spn = self._load(sp, sc, sh, self.g(P.G_CPLX, P.CX_PRELU))
        d = spn @ unfm(self.g(P.G_CPLX, P.CX_TAIL_W), 16, 64)[:4].t()
        cb = self.g(P.G_CPLX, P.CX_BIAS)
        c0 = d[:, :, :-1, 0] + d[:, :, 1:, 1] + cb[0]
        c1_ = d[:, :, :-1, 2] + d[:, :, 1:, 3] + cb[1]
        if stages is not None:
            stages["mask"] = mask[:, None]
            stages["complex"] = torch.stack([c0, c1_], dim=1)
        return (mask * re + c0)[:, None], (mask * im + c1_)[:, None]
