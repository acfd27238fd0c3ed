"""state_dict -> packed weight blob for libcmgan_hip.so (layout: csrc/weights.h).

Consumes the reference generator ``state_dict`` (the 359-entry dict that
``src/evaluation.py:63-64`` loads; key names in SURVEY.md App. C) and produces the
host blob handed to ``cmgan_load_weights``.  All algebraic folds are done here once,
in float64, then rounded to fp32:

* LayerNorm affine of every PreNorm folded into the following Linear / pointwise conv
  (``W' = W diag(gamma)``, ``b' = b + W beta``)           conformer.py:54-72,161-163
* ``Scale(0.5)`` folded into the second FeedForward Linear   conformer.py:211-212
* FeedForward Swish evaluated on ``h' = -log2(e) h``: first Linear scaled by -log2(e), second by -ln 2
* attention ``scale = dim_head**-0.5`` folded into ``to_q``  conformer.py:80,103,110
  together with log2(e), so the kernels' softmax is a bare exp2
* eval-mode ``BatchNorm1d`` folded into the depthwise conv    conformer.py:165-168
* dense-block input channels reordered from the reference's newest-first concat
  (generator.py:46) to slot order (block input first)
* every MFMA operand stored fragment-major (see weights.h)
"""
from __future__ import annotations

import struct

import numpy as np

MAGIC = 0x42474D43
VERSION = 1
EPS = 1e-5

G_ENC, G_DB_E, G_DB_M, G_DB_C, G_MASK, G_CPLX, G_CONF0 = 0, 1, 2, 3, 4, 5, 8
(ENC_C1_W, ENC_C1_GB, ENC_C1_PRELU, ENC_C2_W, ENC_C2_BIAS, ENC_C2_GB, ENC_C2_PRELU) = range(7)
(MK_SP_W, MK_SP_BIAS, MK_TAIL_W, MK_SCALARS, MK_PRELU_OUT) = range(5)
(CX_SP_W, CX_SP_BIAS, CX_GB, CX_PRELU, CX_TAIL_W, CX_BIAS) = range(6)
(CF_FF1_W1, CF_FF1_B1, CF_FF1_W2, CF_FF1_B2, CF_QKV_W, CF_QKV_B, CF_WO, CF_BO, CF_REL, CF_PW1_W,
 CF_PW1_B, CF_DW_W, CF_DW_B, CF_PW2_W, CF_PW2_B, CF_FF2_W1, CF_FF2_B1, CF_FF2_W2, CF_FF2_B2,
 CF_POST_GB) = range(20)


def wid(group: int, item: int) -> int:
    return group * 64 + item


def _np(t) -> np.ndarray:
    if hasattr(t, "detach"):
        t = t.detach().cpu().numpy()
    return np.asarray(t, dtype=np.float64)


def fm(mat: np.ndarray) -> np.ndarray:
    """Fragment-major packing: out[rb][kb][lane][r] = M[16rb + (lane&15)][16kb + 4(lane>>4) + r]."""
    rows, cols = mat.shape
    assert rows % 16 == 0 and cols % 16 == 0, mat.shape
    return mat.reshape(rows // 16, 16, cols // 16, 4, 4).transpose(0, 2, 3, 1, 4).reshape(-1)


def fm_pad_rows(mat: np.ndarray) -> np.ndarray:
    rows, cols = mat.shape
    out = np.zeros((16 * ((rows + 15) // 16), cols))
    out[:rows] = mat
    return fm(out)


def conv_fm(w: np.ndarray) -> np.ndarray:
    """[COUT, CI, NT, 3] (slot-ordered CI) -> [CI/16][NT*3][COUT/16][64][4]."""
    cout, ci, nt, kf = w.shape
    assert kf == 3 and ci % 16 == 0
    parts = []
    for chunk in range(ci // 16):
        for kt in range(nt):
            for k in range(3):
                parts.append(fm(w[:, 16 * chunk:16 * chunk + 16, kt, k]))
    return np.concatenate(parts)


def _gb(sd, name):
    return np.concatenate([_np(sd[name + ".weight"]), _np(sd[name + ".bias"])])


def _fold_ln(w, b, gamma, beta):
    """Linear(LN(x)) with LN = xhat*gamma + beta  ->  W' xhat + b'."""
    w2 = w * gamma[None, :]
    b2 = w @ beta + (b if b is not None else 0.0)
    return w2, b2


def _dense_block(sd, prefix, group, out):
    for i in range(4):                      # reference conv{i+1}
        w = _np(sd[f"{prefix}.conv{i + 1}.weight"])        # [64, 64(i+1), 2, 3], newest-first channels
        nblk = i + 1
        slot = np.concatenate([w[:, 64 * (nblk - 1 - s):64 * (nblk - s)] for s in range(nblk)], axis=1)
        out[wid(group, i * 4 + 0)] = conv_fm(slot)
        out[wid(group, i * 4 + 1)] = _np(sd[f"{prefix}.conv{i + 1}.bias"])
        out[wid(group, i * 4 + 2)] = _gb(sd, f"{prefix}.norm{i + 1}")
        out[wid(group, i * 4 + 3)] = _np(sd[f"{prefix}.prelu{i + 1}.weight"])


def _conformer(sd, p, group, out, heads=4, dim_head=16):
    pre = p + "." if p else ""
    for ff, (iw1, ib1, iw2, ib2) in (("ff1", (CF_FF1_W1, CF_FF1_B1, CF_FF1_W2, CF_FF1_B2)),
                                     ("ff2", (CF_FF2_W1, CF_FF2_B1, CF_FF2_W2, CF_FF2_B2))):
        g, bt = _np(sd[f"{pre}{ff}.fn.norm.weight"]), _np(sd[f"{pre}{ff}.fn.norm.bias"])
        w1, b1 = _fold_ln(_np(sd[f"{pre}{ff}.fn.fn.net.0.weight"]), _np(sd[f"{pre}{ff}.fn.fn.net.0.bias"]), g, bt)
        # Swish(h) = h / (1 + e^-h) = -ln2 * h' / (1 + 2^h') with h' = -log2(e) * h: the first Linear is
        # stored pre-multiplied by -log2(e) and the second by -ln2, so the kernel's activation is
        # h' * rcp(1 + exp2(h')) - one multiply per hidden value fewer (conformer.py:25-27,141-147)
        k1 = np.float32(-np.log2(np.e))
        k2 = np.float32(-np.log(2.0) * 0.5)
        out[wid(group, iw1)] = fm(k1 * w1)
        out[wid(group, ib1)] = k1 * b1
        out[wid(group, iw2)] = fm(k2 * _np(sd[f"{pre}{ff}.fn.fn.net.3.weight"]))
        out[wid(group, ib2)] = 0.5 * _np(sd[f"{pre}{ff}.fn.fn.net.3.bias"])
    scale = dim_head ** -0.5 * np.log2(np.e)      # attention scale, and scores in log2 units (kernels use exp2)
    wq = scale * _np(sd[f"{pre}attn.fn.to_q.weight"])
    wkv = _np(sd[f"{pre}attn.fn.to_kv.weight"])
    g, bt = _np(sd[f"{pre}attn.norm.weight"]), _np(sd[f"{pre}attn.norm.bias"])
    wqkv, bqkv = _fold_ln(np.concatenate([wq, wkv], axis=0), None, g, bt)
    out[wid(group, CF_QKV_W)] = fm(wqkv)
    out[wid(group, CF_QKV_B)] = bqkv
    out[wid(group, CF_WO)] = fm(_np(sd[f"{pre}attn.fn.to_out.weight"]))
    out[wid(group, CF_BO)] = _np(sd[f"{pre}attn.fn.to_out.bias"])
    out[wid(group, CF_REL)] = _np(sd[f"{pre}attn.fn.rel_pos_emb.weight"]).reshape(-1)
    g, bt = _np(sd[f"{pre}conv.net.0.weight"]), _np(sd[f"{pre}conv.net.0.bias"])
    w1, b1 = _fold_ln(_np(sd[f"{pre}conv.net.2.weight"])[:, :, 0], _np(sd[f"{pre}conv.net.2.bias"]), g, bt)
    out[wid(group, CF_PW1_W)] = fm(w1)
    out[wid(group, CF_PW1_B)] = b1
    dw = _np(sd[f"{pre}conv.net.4.conv.weight"])[:, 0, :]                     # [128, 31]
    db = _np(sd[f"{pre}conv.net.4.conv.bias"])
    s = _np(sd[f"{pre}conv.net.5.weight"]) / np.sqrt(_np(sd[f"{pre}conv.net.5.running_var"]) + EPS)
    out[wid(group, CF_DW_W)] = np.ascontiguousarray((dw * s[:, None]).T).reshape(-1)   # [31][128]
    out[wid(group, CF_DW_B)] = (db - _np(sd[f"{pre}conv.net.5.running_mean"])) * s + _np(sd[f"{pre}conv.net.5.bias"])
    out[wid(group, CF_PW2_W)] = fm(_np(sd[f"{pre}conv.net.7.weight"])[:, :, 0])
    out[wid(group, CF_PW2_B)] = _np(sd[f"{pre}conv.net.7.bias"])
    out[wid(group, CF_POST_GB)] = _gb(sd, f"{pre}post_norm")


def _assemble(entries: dict) -> np.ndarray:
    ids = sorted(entries)
    offs, cur = [], 0
    arrs = []
    for i in ids:
        a = np.ascontiguousarray(np.asarray(entries[i], dtype=np.float64).reshape(-1)).astype(np.float32)
        arrs.append(a)
        offs.append(cur)
        cur += (a.size + 63) // 64 * 64
    payload = np.zeros(cur, dtype=np.float32)
    for a, o in zip(arrs, offs):
        payload[o:o + a.size] = a
    head = struct.pack("<4I", MAGIC, VERSION, len(ids), cur)
    directory = b"".join(struct.pack("<4I", i, o, a.size, 0) for i, o, a in zip(ids, offs, arrs))
    return np.frombuffer(head + directory + payload.tobytes(), dtype=np.uint8).copy()


def pack_state_dict(sd: dict, num_tscb: int = 4) -> np.ndarray:
    """Full generator ``state_dict`` -> blob (uint8 array)."""
    e: dict = {}
    c1 = _np(sd["dense_encoder.conv_1.0.weight"])[:, :, 0, 0]                 # [64, 3]: mag, re, im
    e[wid(G_ENC, ENC_C1_W)] = np.concatenate([c1[:, 0], c1[:, 1], c1[:, 2], _np(sd["dense_encoder.conv_1.0.bias"])])
    e[wid(G_ENC, ENC_C1_GB)] = _gb(sd, "dense_encoder.conv_1.1")
    e[wid(G_ENC, ENC_C1_PRELU)] = _np(sd["dense_encoder.conv_1.2.weight"])
    e[wid(G_ENC, ENC_C2_W)] = conv_fm(_np(sd["dense_encoder.conv_2.0.weight"]))
    e[wid(G_ENC, ENC_C2_BIAS)] = _np(sd["dense_encoder.conv_2.0.bias"])
    e[wid(G_ENC, ENC_C2_GB)] = _gb(sd, "dense_encoder.conv_2.1")
    e[wid(G_ENC, ENC_C2_PRELU)] = _np(sd["dense_encoder.conv_2.2.weight"])
    _dense_block(sd, "dense_encoder.dilated_dense", G_DB_E, e)
    _dense_block(sd, "mask_decoder.dense_block", G_DB_M, e)
    _dense_block(sd, "complex_decoder.dense_block", G_DB_C, e)
    for k in range(num_tscb):
        _conformer(sd, f"TSCB_{k + 1}.time_conformer", G_CONF0 + 2 * k, e)
        _conformer(sd, f"TSCB_{k + 1}.freq_conformer", G_CONF0 + 2 * k + 1, e)
    # mask decoder tail (generator.py:126-131)
    e[wid(G_MASK, MK_SP_W)] = conv_fm(_np(sd["mask_decoder.sub_pixel.conv.weight"]))
    e[wid(G_MASK, MK_SP_BIAS)] = _np(sd["mask_decoder.sub_pixel.conv.bias"])
    mw = _np(sd["mask_decoder.conv_1.weight"])                                 # [1, 64, 1, 2]
    e[wid(G_MASK, MK_TAIL_W)] = fm_pad_rows(np.stack([mw[0, :, 0, 0], mw[0, :, 0, 1]]))
    e[wid(G_MASK, MK_SCALARS)] = np.array([
        _np(sd["mask_decoder.conv_1.bias"])[0], _np(sd["mask_decoder.norm.weight"])[0],
        _np(sd["mask_decoder.norm.bias"])[0], _np(sd["mask_decoder.prelu.weight"])[0],
        _np(sd["mask_decoder.final_conv.weight"]).reshape(-1)[0], _np(sd["mask_decoder.final_conv.bias"])[0],
        0.0, 0.0])
    e[wid(G_MASK, MK_PRELU_OUT)] = _np(sd["mask_decoder.prelu_out.weight"])
    # complex decoder tail (generator.py:146-149)
    e[wid(G_CPLX, CX_SP_W)] = conv_fm(_np(sd["complex_decoder.sub_pixel.conv.weight"]))
    e[wid(G_CPLX, CX_SP_BIAS)] = _np(sd["complex_decoder.sub_pixel.conv.bias"])
    e[wid(G_CPLX, CX_GB)] = _gb(sd, "complex_decoder.norm")
    e[wid(G_CPLX, CX_PRELU)] = _np(sd["complex_decoder.prelu.weight"])
    cw = _np(sd["complex_decoder.conv.weight"])                                # [2, 64, 1, 2]
    e[wid(G_CPLX, CX_TAIL_W)] = fm_pad_rows(np.stack([cw[0, :, 0, 0], cw[0, :, 0, 1], cw[1, :, 0, 0], cw[1, :, 0, 1]]))
    e[wid(G_CPLX, CX_BIAS)] = _np(sd["complex_decoder.conv.bias"])
    return _assemble(e)


def pack_conformer_state_dict(sd: dict, slot: int = 0) -> np.ndarray:
    """State dict of one stand-alone ``ConformerBlock`` (un-prefixed keys) -> blob with
    only conformer slot ``slot`` populated."""
    e: dict = {}
    _conformer(sd, "", G_CONF0 + slot, e)
    return _assemble(e)


REQUIRED_KEYS_HINT = "dense_encoder.conv_1.0.weight"
