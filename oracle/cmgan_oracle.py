"""CPU restatement of the CMGAN generator forward path (torch, fp32, functional).

TEST INFRASTRUCTURE (see oracle/__init__.py).  Parity status: PINNED against the
reference's own modules run in the build container (tests/golden/*.npz, made by
tests/golden/make_golden.py from /root/reference/src; checked by
tests/test_oracle_golden.py).  The reference ships no golden vectors or tests of
its own for this path.

Every function takes the reference ``state_dict`` (``sd``) plus a key prefix and
cites the reference lines it restates.  Paths are relative to /root/reference/.
The code is written functionally (no nn.Module tree), with the layout flips,
concatenations and the rel-pos gather re-expressed, so it doubles as the
executable specification the HIP kernels are written against.
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F

EPS = 1e-5  # LayerNorm / InstanceNorm2d / BatchNorm1d default eps


# --------------------------------------------------------------------------- #
# front / back end: src/evaluation.py:21-53, src/utils.py:20-39
# --------------------------------------------------------------------------- #
def rms_scale(wav: torch.Tensor) -> torch.Tensor:
    """c = sqrt(L / sum(x^2)) per row (src/evaluation.py:21, src/train.py:75-79)."""
    return torch.sqrt(wav.size(-1) / torch.sum(wav ** 2.0, dim=-1))


def stft(wav: torch.Tensor, n_fft: int = 400, hop: int = 100) -> torch.Tensor:
    """[B, L] -> [B, F, T, 2] with the reference's torch.stft settings
    (src/evaluation.py:36-38): periodic Hamming window, center/reflect pad,
    one-sided, un-normalised.  torch 2.x needs return_complex=True; the legacy
    real view is restored with view_as_real."""
    win = torch.hamming_window(n_fft, dtype=wav.dtype)
    spec = torch.stft(wav, n_fft, hop, window=win, onesided=True, return_complex=True)
    return torch.view_as_real(spec)


def power_compress(spec: torch.Tensor) -> torch.Tensor:
    """[B, F, T, 2] -> [B, 2, F, T]; mag**0.3 with the phase kept (src/utils.py:20-29)."""
    re, im = spec[..., 0], spec[..., 1]
    mag = torch.sqrt(re * re + im * im)
    phase = torch.atan2(im, re)
    m = mag ** 0.3
    return torch.stack([m * torch.cos(phase), m * torch.sin(phase)], dim=1)


def power_uncompress(real: torch.Tensor, imag: torch.Tensor) -> torch.Tensor:
    """[B, 1, F, T] x2 -> [B, 1, F, T, 2]; mag**(1/0.3) (src/utils.py:32-39)."""
    mag = torch.sqrt(real * real + imag * imag)
    phase = torch.atan2(imag, real)
    m = mag ** (1.0 / 0.3)
    return torch.stack([m * torch.cos(phase), m * torch.sin(phase)], dim=-1)


def istft(spec: torch.Tensor, n_fft: int = 400, hop: int = 100) -> torch.Tensor:
    """[B, F, T, 2] -> [B, hop*(T-1)] (src/evaluation.py:44-50)."""
    win = torch.hamming_window(n_fft, dtype=spec.dtype)
    return torch.istft(torch.view_as_complex(spec.contiguous()), n_fft, hop,
                       window=win, onesided=True)


def stft_compress(wav: torch.Tensor, n_fft: int = 400, hop: int = 100) -> torch.Tensor:
    """[B, L] -> model input [B, 2, T, F] (src/evaluation.py:36-39)."""
    return power_compress(stft(wav, n_fft, hop)).permute(0, 1, 3, 2).contiguous()


def uncompress_istft(est_real: torch.Tensor, est_imag: torch.Tensor,
                     n_fft: int = 400, hop: int = 100) -> torch.Tensor:
    """model outputs 2x[B, 1, T, F] -> [B, hop*(T-1)] (src/evaluation.py:41-50)."""
    r, i = est_real.permute(0, 1, 3, 2), est_imag.permute(0, 1, 3, 2)
    return istft(power_uncompress(r, i).squeeze(1), n_fft, hop)


# --------------------------------------------------------------------------- #
# conformer: src/models/conformer.py
# --------------------------------------------------------------------------- #
def layer_norm(sd, p, x):
    """nn.LayerNorm(64) (conformer.py:68,161,214)."""
    return F.layer_norm(x, (x.shape[-1],), sd[p + ".weight"], sd[p + ".bias"], EPS)


def feed_forward(sd, p, x):
    """Scale(0.5, PreNorm(FeedForward)) (conformer.py:136-148, 54-72, 211-212).
    ``p`` is e.g. 'TSCB_1.time_conformer.ff1'."""
    h = layer_norm(sd, p + ".fn.norm", x)
    h = F.linear(h, sd[p + ".fn.fn.net.0.weight"], sd[p + ".fn.fn.net.0.bias"])
    h = h * torch.sigmoid(h)                                  # Swish, conformer.py:25-27
    h = F.linear(h, sd[p + ".fn.fn.net.3.weight"], sd[p + ".fn.fn.net.3.bias"])
    return 0.5 * h


def feed_forward_train(sd, p, x, mask1=None, mask2=None):
    """TRAIN-mode Scale(0.5, PreNorm(dim, FeedForward(dim, mult=4, dropout))) (conformer.py:54-72, 136-148,
    211-212): the two nn.Dropout layers (conformer.py:142, 144) as explicit keep-masks (entries 0 or 1/(1-p);
    None = no dropout).  Differentiable: torch autograd through this function is the gradient oracle of
    cmgan_amd.training.FeedForwardTrain.  `p` is the branch prefix, e.g. "ff1" (keys p + ".fn.norm.weight", ...)."""
    h = F.layer_norm(x, (x.shape[-1],), sd[p + ".fn.norm.weight"], sd[p + ".fn.norm.bias"], 1e-5)
    h = F.linear(h, sd[p + ".fn.fn.net.0.weight"], sd[p + ".fn.fn.net.0.bias"])
    h = h * torch.sigmoid(h)                                   # Swish (conformer.py:25-27)
    if mask1 is not None:
        h = h * mask1
    h = F.linear(h, sd[p + ".fn.fn.net.3.weight"], sd[p + ".fn.fn.net.3.bias"])
    if mask2 is not None:
        h = h * mask2
    return 0.5 * h


def attention(sd, p, x, heads: int = 4, max_pos: int = 512, mask=None):
    """PreNorm(Attention) with Shaw relative positions (conformer.py:75-133).
    `mask` [b, n] bool (conformer.py:113-126): pairs (i, j) with mask_i AND mask_j keep their score, every other
    score becomes -finfo.max - so a masked QUERY row attends uniformly to all n keys, like in the reference.
    bias[i,j] = q_i . E[clamp(i-j, +-512) + 512]; the reference materialises
    E[dist] as [n, n, d]; here q E^T is formed once ([.., n, 1025]) and gathered
    along the relative index - the same sums, a Toeplitz read."""
    n = x.shape[-2]
    h = layer_norm(sd, p + ".norm", x)
    q = F.linear(h, sd[p + ".fn.to_q.weight"])
    kv = F.linear(h, sd[p + ".fn.to_kv.weight"])
    k, v = kv[..., : kv.shape[-1] // 2], kv[..., kv.shape[-1] // 2:]
    d = q.shape[-1] // heads
    split = lambda t: t.reshape(t.shape[0], n, heads, d).transpose(1, 2)   # b h n d
    q, k, v = split(q), split(k), split(v)
    scale = d ** -0.5
    dots = torch.matmul(q, k.transpose(-1, -2)) * scale
    emb = sd[p + ".fn.rel_pos_emb.weight"]                                  # [1025, d]
    idx = torch.arange(n)
    rel = (idx[:, None] - idx[None, :]).clamp(-max_pos, max_pos) + max_pos   # [n, n]
    qe = torch.matmul(q, emb.t())                                           # b h n 1025
    pos = torch.gather(qe, -1, rel.expand(q.shape[0], heads, n, n)) * scale
    dots = dots + pos
    if mask is not None:
        pair = mask[:, None, :, None] & mask[:, None, None, :]
        dots = dots.masked_fill(~pair, -torch.finfo(dots.dtype).max)
    attn = torch.softmax(dots, dim=-1)
    out = torch.matmul(attn, v).transpose(1, 2).reshape(x.shape[0], n, heads * d)
    return F.linear(out, sd[p + ".fn.to_out.weight"], sd[p + ".fn.to_out.bias"])


def attention_train(sd, p, x, mask=None, heads: int = 4, max_pos: int = 512):
    """PreNorm(dim, Attention) in TRAIN mode (conformer.py:54-72, 100-133): identical to `layer_norm` + `attention`
    except for the nn.Dropout on the to_out output (conformer.py:133), given as a keep-mask [N, L, 64] (None = no
    dropout).  Differentiable: autograd through it is the gradient oracle of cmgan_amd.training.AttentionTrain.
    `p` is the branch prefix, e.g. "attn" (keys p + ".norm.weight", p + ".fn.to_q.weight", ...)."""
    out = attention(sd, p, x, heads, max_pos)                  # includes the PreNorm LayerNorm
    return out * mask if mask is not None else out


def conv_module(sd, p, x, kernel: int = 31):
    """ConformerConvModule, eval mode (conformer.py:151-176, 30-48)."""
    h = layer_norm(sd, p + ".net.0", x).transpose(1, 2)                    # b c n
    h = F.conv1d(h, sd[p + ".net.2.weight"], sd[p + ".net.2.bias"])
    a, g = h.chunk(2, dim=1)
    h = a * torch.sigmoid(g)                                               # GLU
    pad = kernel // 2
    h = F.pad(h, (pad, pad - (kernel + 1) % 2))
    h = F.conv1d(h, sd[p + ".net.4.conv.weight"], sd[p + ".net.4.conv.bias"],
                 groups=h.shape[1])
    h = F.batch_norm(h, sd[p + ".net.5.running_mean"], sd[p + ".net.5.running_var"],
                     sd[p + ".net.5.weight"], sd[p + ".net.5.bias"], False, 0.1, EPS)
    h = h * torch.sigmoid(h)
    h = F.conv1d(h, sd[p + ".net.7.weight"], sd[p + ".net.7.bias"])
    return h.transpose(1, 2)


def conv_module_train(sd, p, x, running: dict | None = None, kernel: int = 31):
    """ConformerConvModule in TRAIN mode (conformer.py:151-176): BatchNorm1d(128) normalises with the statistics of
    the batch (biased variance over all N*L positions, eps 1e-5) and, when `running` = {"mean", "var"} is given,
    updates the running statistics in place with momentum 0.1 and the unbiased variance (torch semantics); the
    module's Dropout has p = conv_dropout = 0 (conformer.py:193, generator.py:75-90).  Differentiable: autograd
    through this function is the gradient oracle of cmgan_amd.training.ConvModuleTrain."""
    h = layer_norm(sd, p + ".net.0", x).transpose(1, 2)                   # [N,64,L]
    h = F.conv1d(h, sd[p + ".net.2.weight"], sd[p + ".net.2.bias"])      # pointwise 64 -> 256
    a, g = h.chunk(2, dim=1)
    h = a * torch.sigmoid(g)                                              # GLU(dim=1)
    pad = kernel // 2
    h = F.pad(h, (pad, pad - (kernel + 1) % 2))
    h = F.conv1d(h, sd[p + ".net.4.conv.weight"], sd[p + ".net.4.conv.bias"], groups=h.shape[1])
    rm = running["mean"] if running is not None else None
    rv = running["var"] if running is not None else None
    h = F.batch_norm(h, rm, rv, sd[p + ".net.5.weight"], sd[p + ".net.5.bias"], True, 0.1, EPS)
    h = h * torch.sigmoid(h)
    h = F.conv1d(h, sd[p + ".net.7.weight"], sd[p + ".net.7.bias"])
    return h.transpose(1, 2)


def conformer_block(sd, p, x, stages: dict | None = None, mask=None):
    """ConformerBlock.forward (conformer.py:216-222).  x: [N, L, 64].
    ``stages`` (optional dict) receives the residual stream after each sub-module."""
    pre = (p + ".") if p else ""
    x = feed_forward(sd, pre + "ff1", x) + x
    if stages is not None: stages["ff1"] = x
    x = attention(sd, pre + "attn", x, mask=mask) + x
    if stages is not None: stages["attn"] = x
    x = conv_module(sd, pre + "conv", x) + x
    if stages is not None: stages["conv"] = x
    x = feed_forward(sd, pre + "ff2", x) + x
    if stages is not None: stages["ff2"] = x
    return layer_norm(sd, pre + "post_norm", x)


# --------------------------------------------------------------------------- #
# generator: src/models/generator.py
# --------------------------------------------------------------------------- #
def conformer_block_train(sd, p, x, masks: dict | None = None, running: dict | None = None):
    """ConformerBlock.forward in TRAIN mode (conformer.py:216-222): the five nn.Dropout layers as keep-masks
    `masks` = {"ff1_1" [..,256], "ff1_2" [..,64], "attn" [..,64], "ff2_1", "ff2_2"} (missing / None = no dropout),
    BatchNorm1d on batch statistics.  Differentiable (gradient oracle of cmgan_amd.training.ConformerBlockTrain)."""
    pre = (p + ".") if p else ""
    m = masks or {}
    x = feed_forward_train(sd, pre + "ff1", x, m.get("ff1_1"), m.get("ff1_2")) + x
    x = attention_train(sd, pre + "attn", x, m.get("attn")) + x
    x = conv_module_train(sd, pre + "conv", x, running) + x
    x = feed_forward_train(sd, pre + "ff2", x, m.get("ff2_1"), m.get("ff2_2")) + x
    return layer_norm(sd, pre + "post_norm", x)


def tscb_train(sd, p, x, masks_time: dict | None = None, masks_freq: dict | None = None):
    """TSCB.forward in TRAIN mode (generator.py:92-99).  x: NCHW [B, 64, T, F'] like the reference."""
    b, c, t, f = x.shape
    x_t = x.permute(0, 3, 2, 1).contiguous().view(b * f, t, c)
    x_t = conformer_block_train(sd, p + ".time_conformer", x_t, masks_time) + x_t
    x_f = x_t.view(b, f, t, c).permute(0, 2, 1, 3).contiguous().view(b * t, f, c)
    x_f = conformer_block_train(sd, p + ".freq_conformer", x_f, masks_freq) + x_f
    return x_f.view(b, t, f, c).permute(0, 3, 1, 2)


class norm_stats:
    """Context manager for FROZEN InstanceNorm statistics (the carried-state streaming contract of
    include/cmgan_hip.h, "Streaming"; oracle/stream_oracle.py).  `norm_stats("record", d)`: every InstanceNorm2d of the
    generator (generator.py:35,55,61,128,148) stores the per-(b, c) mean and biased variance it used in d[<its
    state-dict prefix>]; `norm_stats("replay", d)`: it normalises with the stored pair instead of its input's own -
    BatchNorm-in-eval-mode arithmetic, under which the dense encoder and both decoders are exactly time-causal."""
    active = None

    def __init__(self, mode: str, store: dict):
        assert mode in ("record", "replay")
        self.mode, self.store = mode, store

    def __enter__(self):
        self.prev, norm_stats.active = norm_stats.active, self
        return self.store

    def __exit__(self, *exc):
        norm_stats.active = self.prev


def _instance_norm(x, w, b, key):
    ctx = norm_stats.active
    if ctx is None:
        return F.instance_norm(x, weight=w, bias=b, eps=EPS)
    if ctx.mode == "record":
        ctx.store[key] = (x.mean(dim=(2, 3), keepdim=True), x.var(dim=(2, 3), unbiased=False, keepdim=True))
        return F.instance_norm(x, weight=w, bias=b, eps=EPS)
    mean, var = ctx.store[key]
    return (x - mean) / torch.sqrt(var + EPS) * w.view(1, -1, 1, 1) + b.view(1, -1, 1, 1)


def _in_prelu(sd, norm, prelu, x):
    x = _instance_norm(x, sd[norm + ".weight"], sd[norm + ".bias"], norm)
    return F.prelu(x, sd[prelu + ".weight"])


def dense_block(sd, p, x):
    """DilatedDenseNet.forward (generator.py:39-47): 4 x {pad(top=dil, l/r=1),
    conv(2x3, dilation (dil,1)), InstanceNorm, PReLU, cat newest-first}."""
    skip = x
    out = x
    for i in range(1, 5):
        dil = 2 ** (i - 1)
        out = F.pad(skip, (1, 1, dil, 0))
        out = F.conv2d(out, sd[f"{p}.conv{i}.weight"], sd[f"{p}.conv{i}.bias"], dilation=(dil, 1))
        out = _in_prelu(sd, f"{p}.norm{i}", f"{p}.prelu{i}", out)
        skip = torch.cat([out, skip], dim=1)
    return out


def dense_encoder(sd, x_in):
    """DenseEncoder.forward (generator.py:65-69).  [B,3,T,F] -> [B,64,T,F']."""
    p = "dense_encoder"
    x = F.conv2d(x_in, sd[p + ".conv_1.0.weight"], sd[p + ".conv_1.0.bias"])
    x = _in_prelu(sd, p + ".conv_1.1", p + ".conv_1.2", x)
    x = dense_block(sd, p + ".dilated_dense", x)
    x = F.conv2d(x, sd[p + ".conv_2.0.weight"], sd[p + ".conv_2.0.bias"], stride=(1, 2), padding=(0, 1))
    return _in_prelu(sd, p + ".conv_2.1", p + ".conv_2.2", x)


def tscb(sd, p, x):
    """TSCB.forward (generator.py:92-99): time conformer over T for every (b,f),
    then frequency conformer over F' for every (b,t), each with an outer residual."""
    b, c, t, f = x.shape
    xt = x.permute(0, 3, 2, 1).reshape(b * f, t, c)
    xt = conformer_block(sd, p + ".time_conformer", xt) + xt
    xf = xt.reshape(b, f, t, c).permute(0, 2, 1, 3).reshape(b * t, f, c)
    xf = conformer_block(sd, p + ".freq_conformer", xf) + xf
    return xf.reshape(b, t, f, c).permute(0, 3, 1, 2)


def sub_pixel(sd, p, x, r: int = 2):
    """SPConvTranspose2d (generator.py:112-119): out[b,c,t,2f+k] = conv[b,64k+c,t,f]."""
    y = F.conv2d(F.pad(x, (1, 1, 0, 0)), sd[p + ".conv.weight"], sd[p + ".conv.bias"])
    b, ch, t, w = y.shape
    return y.reshape(b, r, ch // r, t, w).permute(0, 2, 3, 4, 1).reshape(b, ch // r, t, w * r)


def mask_decoder(sd, x):
    """MaskDecoder.forward (generator.py:133-139).  [B,64,T,F'] -> [B,1,T,F]."""
    p = "mask_decoder"
    x = dense_block(sd, p + ".dense_block", x)
    x = sub_pixel(sd, p + ".sub_pixel", x)
    x = F.conv2d(x, sd[p + ".conv_1.weight"], sd[p + ".conv_1.bias"])
    x = _in_prelu(sd, p + ".norm", p + ".prelu", x)
    x = F.conv2d(x, sd[p + ".final_conv.weight"], sd[p + ".final_conv.bias"])
    a = sd[p + ".prelu_out.weight"].view(1, 1, 1, -1)          # slope per frequency bin
    return torch.where(x >= 0, x, a * x)


def complex_decoder(sd, x):
    """ComplexDecoder.forward (generator.py:151-156).  [B,64,T,F'] -> [B,2,T,F]."""
    p = "complex_decoder"
    x = dense_block(sd, p + ".dense_block", x)
    x = sub_pixel(sd, p + ".sub_pixel", x)
    x = _in_prelu(sd, p + ".norm", p + ".prelu", x)
    return F.conv2d(x, sd[p + ".conv.weight"], sd[p + ".conv.bias"])


def tscnet_forward(sd, x, stages: dict | None = None):
    """TSCNet.forward (generator.py:174-196).  x: [B,2,T,F] -> 2 x [B,1,T,F]."""
    re, im = x[:, 0:1], x[:, 1:2]
    mag = torch.sqrt(re * re + im * im)
    phase = torch.atan2(im, re)
    h = dense_encoder(sd, torch.cat([mag, x], dim=1))
    if stages is not None: stages["encoder"] = h
    for b in range(1, 5):
        h = tscb(sd, f"TSCB_{b}", h)
        if stages is not None: stages[f"tscb{b}"] = h
    mask = mask_decoder(sd, h)
    cplx = complex_decoder(sd, h)
    if stages is not None:
        stages["mask"] = mask
        stages["complex"] = cplx
    out_mag = mask * mag
    real = out_mag * torch.cos(phase) + cplx[:, 0:1]
    imag = out_mag * torch.sin(phase) + cplx[:, 1:2]
    return real, imag


# --------------------------------------------------------------------------- #
# whole pipeline: src/evaluation.py:12-58 (enhance_one_track, minus file I/O)
# --------------------------------------------------------------------------- #
def chunk_rows(padded_len: int, cut_len: int) -> int:
    """Batch rows for long audio (evaluation.py:30-34)."""
    if padded_len <= cut_len:
        return 1
    rows = int(math.ceil(padded_len / cut_len))
    while 100 % rows != 0:
        rows += 1
    return rows


@torch.no_grad()
def enhance(sd, noisy: torch.Tensor, cut_len: int = 16000 * 16, n_fft: int = 400, hop: int = 100):
    """noisy: [1, L] float32 in [-1, 1] -> enhanced [L] (evaluation.py:21-53)."""
    c = rms_scale(noisy)
    noisy = noisy * c[:, None]
    length = noisy.size(-1)
    padded = int(math.ceil(length / 100)) * 100
    noisy = torch.cat([noisy, noisy[:, : padded - length]], dim=-1)
    rows = chunk_rows(padded, cut_len)
    if rows > 1:
        noisy = noisy.reshape(rows, -1)
    spec = stft_compress(noisy, n_fft, hop)
    real, imag = tscnet_forward(sd, spec)
    audio = uncompress_istft(real, imag, n_fft, hop) / c
    return audio.flatten()[:length]


@torch.no_grad()
def enhance_batch(sd, wav: torch.Tensor, n_fft: int = 400, hop: int = 100):
    """Batched fixed-length form used by the benchmark: per-row RMS scale
    (train.py:75-79), STFT, generator, ISTFT, un-scale.  wav: [B, L], L % hop == 0."""
    c = rms_scale(wav)
    spec = stft_compress(wav * c[:, None], n_fft, hop)
    real, imag = tscnet_forward(sd, spec)
    return uncompress_istft(real, imag, n_fft, hop) / c[:, None]


# --------------------------------------------------------------------------- #
# generator half of the training / validation step: src/train.py:72-151
# --------------------------------------------------------------------------- #
@torch.no_grad()
def forward_generator_step(sd, clean: torch.Tensor, noisy: torch.Tensor, n_fft: int = 400, hop: int = 100):
    """Trainer.forward_generator_step (train.py:72-122), eval mode.  clean, noisy: [B, L]."""
    c = rms_scale(noisy)                                        # train.py:75 (of the NOISY rows)
    noisy_s, clean_s = noisy * c[:, None], clean * c[:, None]  # train.py:76-79
    noisy_spec = stft_compress(noisy_s, n_fft, hop)            # [B,2,T,F]  (train.py:81-94)
    clean_spec = stft_compress(clean_s, n_fft, hop)            # model layout; the reference keeps [B,2,F,T]
    est_real, est_imag = tscnet_forward(sd, noisy_spec)        # train.py:99
    est_audio = uncompress_istft(est_real, est_imag, n_fft, hop)   # train.py:104-112 (NOT divided by c)
    return {"est_real": est_real, "est_imag": est_imag, "clean_spec": clean_spec, "est_audio": est_audio}


def generator_loss(out: dict, clean: torch.Tensor, loss_weights=(0.1, 0.9, 0.2)):
    """calculate_generator_loss without the GAN term (train.py:124-151): (loss, loss_ri, loss_mag, time_loss).
    `clean` is the RAW batch (train.py:218), while est_audio lives in the RMS-scaled domain - a reference quirk."""
    cr, ci = out["clean_spec"][:, 0:1], out["clean_spec"][:, 1:2]
    er, ei = out["est_real"], out["est_imag"]
    loss_mag = F.mse_loss(torch.sqrt(er ** 2 + ei ** 2), torch.sqrt(cr ** 2 + ci ** 2))
    loss_ri = F.mse_loss(er, cr) + F.mse_loss(ei, ci)
    time_loss = torch.mean(torch.abs(out["est_audio"] - clean))
    loss = loss_weights[0] * loss_ri + loss_weights[1] * loss_mag + loss_weights[2] * time_loss
    return loss, loss_ri, loss_mag, time_loss


def tscnet_forward_train(sd, x, masks=None):
    """TSCNet.forward in TRAIN mode (generator.py:176-201): `masks` = [(time, freq)] * 4 keep-mask dictionaries of
    the four TSCBs (None = no dropout).  Differentiable."""
    re, im = x[:, 0:1], x[:, 1:2]
    mag = torch.sqrt(re * re + im * im)
    phase = torch.atan2(im, re)
    h = dense_encoder(sd, torch.cat([mag, x], dim=1))
    for b in range(1, 5):
        mt, mf = masks[b - 1] if masks is not None else (None, None)
        h = tscb_train(sd, f"TSCB_{b}", h, mt, mf)
    mask = mask_decoder(sd, h)
    cplx = complex_decoder(sd, h)
    out_mag = mask * mag
    return out_mag * torch.cos(phase) + cplx[:, 0:1], out_mag * torch.sin(phase) + cplx[:, 1:2]


def generator_step_gradients(sd, clean, noisy, masks=None, loss_weights=(0.1, 0.9, 0.2), n_fft: int = 400,
                             hop: int = 100):
    """The generator half of Trainer.train_step without the metric discriminator (train.py:72-151, 185-190):
    forward in train mode, the three non-adversarial loss terms, autograd.  Returns a dict with `loss`, `terms`
    (loss_ri, loss_mag, time_loss), `est_real`, `est_imag`, their gradients `d_real`, `d_imag`, and `grads`
    {key: dL/dparam} for every floating-point tensor of `sd` that the loss depends on."""
    leaf = {k: v.detach().clone().requires_grad_(True) for k, v in sd.items()
            if v.is_floating_point() and "running_" not in k}
    sdx = dict(sd)
    sdx.update(leaf)
    with torch.enable_grad():
        c = rms_scale(noisy)
        noisy_spec = stft_compress(noisy * c[:, None], n_fft, hop)
        clean_spec = stft_compress(clean * c[:, None], n_fft, hop)
        er, ei = tscnet_forward_train(sdx, noisy_spec, masks)
        er.retain_grad(); ei.retain_grad()
        audio = uncompress_istft(er, ei, n_fft, hop)
        out = {"est_real": er, "est_imag": ei, "clean_spec": clean_spec, "est_audio": audio}
        loss, l_ri, l_mag, l_time = generator_loss(out, clean[:, :audio.shape[-1]], loss_weights)
        loss.backward()
    return {"loss": loss.detach(), "terms": torch.stack([l_ri, l_mag, l_time]).detach(), "est_real": er.detach(),
            "est_imag": ei.detach(), "d_real": er.grad.detach(), "d_imag": ei.grad.detach(),
            "grads": {k: v.grad for k, v in leaf.items()}}


# --------------------------------------------------------------------------- #
# metric discriminator: src/models/discriminator.py:29-64
# --------------------------------------------------------------------------- #
def spectral_normed(w_orig, u, v, update: bool, eps: float = 1e-12):
    """torch.nn.utils.spectral_norm's weight (one power iteration per training-mode forward, the iteration itself
    outside the autograd graph): returns (w_orig / sigma, u', v').  discriminator.py:33-58."""
    wm = w_orig.reshape(w_orig.shape[0], -1)
    with torch.no_grad():
        if update:
            v = F.normalize(torch.mv(wm.t(), u), dim=0, eps=eps)
            u = F.normalize(torch.mv(wm, v), dim=0, eps=eps)
    sigma = torch.dot(u, torch.mv(wm, v))
    return w_orig / sigma, u, v


def discriminator(dsd, x, y, mask=None, train: bool = True):
    """Discriminator(ndf=16).forward(x, y) (discriminator.py:61-64).  x, y: [B,1,F,T] magnitudes.  `mask` [B,64] is
    the keep-mask of the Dropout(0.3) (None = no dropout).  Returns (score [B,1], {key: updated u / v})."""
    h = torch.cat([x, y], dim=1)
    new = {}
    for i in range(4):
        p = f"layers.{3 * i}"
        w, u, v = spectral_normed(dsd[p + ".weight_orig"], dsd[p + ".weight_u"], dsd[p + ".weight_v"], train)
        new[p + ".weight_u"], new[p + ".weight_v"] = u, v
        h = F.conv2d(h, w, None, stride=(2, 2), padding=(1, 1))
        h = F.instance_norm(h, weight=dsd[f"layers.{3 * i + 1}.weight"], bias=dsd[f"layers.{3 * i + 1}.bias"], eps=EPS)
        h = F.prelu(h, dsd[f"layers.{3 * i + 2}.weight"])
    h = torch.amax(h, dim=(2, 3))
    w, u, v = spectral_normed(dsd["layers.14.weight_orig"], dsd["layers.14.weight_u"], dsd["layers.14.weight_v"], train)
    new["layers.14.weight_u"], new["layers.14.weight_v"] = u, v
    h = F.linear(h, w, dsd["layers.14.bias"])
    if mask is not None:
        h = h * mask
    h = F.prelu(h, dsd["layers.16.weight"])
    w, u, v = spectral_normed(dsd["layers.17.weight_orig"], dsd["layers.17.weight_u"], dsd["layers.17.weight_v"], train)
    new["layers.17.weight_u"], new["layers.17.weight_v"] = u, v
    h = F.linear(h, w, dsd["layers.17.bias"])
    return torch.sigmoid(dsd["layers.18.slope"] * h), new


def adversarial_generator_gradients(sd, dsd, clean, noisy, masks=None, disc_mask=None, loss_weights=(0.1, 0.9, 0.2, 0.05),
                                    n_fft: int = 400, hop: int = 100):
    """Generator half of Trainer.train_step with the metric-discriminator term (train.py:72-151, 185-190): like
    generator_step_gradients plus `gan` = mse(D(clean_mag, est_mag), 1) through `discriminator` (one power iteration).
    Returns loss, gan, est_*, d_real / d_imag, generator grads and the updated u / v of the discriminator."""
    leaf = {k: v.detach().clone().requires_grad_(True) for k, v in sd.items()
            if v.is_floating_point() and "running_" not in k}
    sdx = dict(sd)
    sdx.update(leaf)
    with torch.enable_grad():
        c = rms_scale(noisy)
        noisy_spec = stft_compress(noisy * c[:, None], n_fft, hop)
        clean_spec = stft_compress(clean * c[:, None], n_fft, hop)
        er, ei = tscnet_forward_train(sdx, noisy_spec, masks)
        er.retain_grad(); ei.retain_grad()
        audio = uncompress_istft(er, ei, n_fft, hop)
        out = {"est_real": er, "est_imag": ei, "clean_spec": clean_spec, "est_audio": audio}
        base, l_ri, l_mag, l_time = generator_loss(out, clean[:, :audio.shape[-1]], loss_weights[:3])
        est_mag = torch.sqrt(er ** 2 + ei ** 2).permute(0, 1, 3, 2)
        clean_mag = torch.sqrt(clean_spec[:, 0:1] ** 2 + clean_spec[:, 1:2] ** 2).permute(0, 1, 3, 2)
        score, new = discriminator(dsd, clean_mag, est_mag, disc_mask, train=True)
        gan = F.mse_loss(score.flatten(), torch.ones(score.shape[0], dtype=score.dtype))
        loss = base + loss_weights[3] * gan
        loss.backward()
    return {"loss": loss.detach(), "gan": gan.detach(), "est_real": er.detach(), "est_imag": ei.detach(),
            "d_real": er.grad.detach(), "d_imag": ei.grad.detach(), "grads": {k: v.grad for k, v in leaf.items()},
            "disc_buffers": new, "clean_spec": clean_spec}
