#!/usr/bin/env python3
"""Generate the golden fixtures in this directory from the REFERENCE implementation.

Run in the build container only (needs /root/reference, which does not exist on
the GPU box):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

It imports the reference's own ``models.generator`` / ``models.conformer`` /
``utils`` from /root/reference/src, loads the deterministic weights from
``oracle/weights.py`` into them (strict), runs them on small seeded inputs on
CPU and stores inputs + outputs as ``*.npz``.  ``evaluation.py`` itself cannot
be imported (torchaudio/natsort/soundfile/pesq are absent and it parses
sys.argv), so its 20-line glue (src/evaluation.py:21-53) is driven here through
the reference's ``power_compress`` / ``power_uncompress`` / ``TSCNet`` with the
torch>=2 adapters for ``torch.stft`` / ``torch.istft``.
"""
import json
import math
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference/src")
sys.dont_write_bytecode = True

from models.conformer import ConformerBlock            # noqa: E402  (reference)
from models.generator import TSCNet                    # noqa: E402  (reference)
import utils as ref_utils                              # noqa: E402  (reference)

from oracle.weights import conformer_state_dict, make_state_dict, synthetic_clips  # noqa: E402

torch.manual_seed(0)
torch.set_grad_enabled(False)


def rnd(shape, seed, scale=1.0):
    rng = np.random.Generator(np.random.PCG64(seed))
    return torch.from_numpy((scale * rng.standard_normal(shape)).astype(np.float32))


def save(name, **arrs):
    out = {k: (v.numpy() if isinstance(v, torch.Tensor) else np.asarray(v)) for k, v in arrs.items()}
    np.savez_compressed(os.path.join(HERE, name), **out)
    print(f"{name}: " + ", ".join(f"{k}{tuple(v.shape)}" for k, v in out.items()))


def ref_stft(x, n_fft=400, hop=100):
    return torch.view_as_real(torch.stft(x, n_fft, hop, window=torch.hamming_window(n_fft),
                                         onesided=True, return_complex=True))


def ref_istft(spec, n_fft=400, hop=100):
    return torch.istft(torch.view_as_complex(spec.contiguous()), n_fft, hop,
                       window=torch.hamming_window(n_fft), onesided=True)


def ref_enhance(model, noisy, cut_len, n_fft=400, hop=100):
    """src/evaluation.py:21-53 driven through the reference's own functions."""
    c = torch.sqrt(noisy.size(-1) / torch.sum((noisy ** 2.0), dim=-1))
    noisy = torch.transpose(noisy, 0, 1)
    noisy = torch.transpose(noisy * c, 0, 1)
    length = noisy.size(-1)
    frame_num = int(np.ceil(length / 100))
    padded_len = frame_num * 100
    padding_len = padded_len - length
    noisy = torch.cat([noisy, noisy[:, :padding_len]], dim=-1)
    if padded_len > cut_len:
        batch_size = int(np.ceil(padded_len / cut_len))
        while 100 % batch_size != 0:
            batch_size += 1
        noisy = torch.reshape(noisy, (batch_size, -1))
    noisy_spec = ref_stft(noisy, n_fft, hop)
    noisy_spec = ref_utils.power_compress(noisy_spec).permute(0, 1, 3, 2)
    est_real, est_imag = model(noisy_spec)
    est_real, est_imag = est_real.permute(0, 1, 3, 2), est_imag.permute(0, 1, 3, 2)
    est_spec_uncompress = ref_utils.power_uncompress(est_real, est_imag).squeeze(1)
    est_audio = ref_istft(est_spec_uncompress, n_fft, hop)
    est_audio = est_audio / c
    return torch.flatten(est_audio)[:length]


def main():
    # -- key/shape manifest of the reference state_dict ------------------------
    model = TSCNet(num_channel=64, num_features=201).eval()
    manifest = {k: list(v.shape) for k, v in model.state_dict().items()}
    with open(os.path.join(HERE, "state_dict_manifest.json"), "w") as f:
        json.dump(manifest, f, indent=0, sort_keys=True)
    sd = make_state_dict(seed=0, num_features=201)
    model.load_state_dict(sd, strict=True)

    # -- 1. front/back end ------------------------------------------------------
    wav = synthetic_clips(2, 800, seed=1)
    spec = ref_stft(wav)                                   # [2,201,9,2]
    comp = ref_utils.power_compress(spec)                  # [2,2,201,9]
    unc = ref_utils.power_uncompress(comp[:, 0:1], comp[:, 1:2])   # [2,1,201,9,2]
    back = ref_istft(unc.squeeze(1))
    save("stft.npz", wav=wav, spec=spec, compressed=comp, uncompressed=unc, istft=back)

    # -- 2. one conformer block, sub-module by sub-module ----------------------
    csd = conformer_state_dict(seed=3)
    blk = ConformerBlock(dim=64, dim_head=16, heads=4, conv_kernel_size=31,
                         attn_dropout=0.2, ff_dropout=0.2).eval()
    blk.load_state_dict(csd, strict=True)
    x = rnd((3, 37, 64), 11)
    s1 = blk.ff1(x) + x
    s2 = blk.attn(s1) + s1
    s3 = blk.conv(s2) + s2
    s4 = blk.ff2(s3) + s3
    out = blk.post_norm(s4)
    assert torch.equal(out, blk(x))
    save("conformer.npz", x=x, ff1=s1, attn=s2, conv=s3, ff2=s4, out=out)

    # -- 2b. ConformerBlock.forward(x, mask) (conformer.py:216-217, 113-126): ragged lengths as a [b, n] bool mask, plus
    #        an arbitrary (non-prefix) mask; masked query rows attend uniformly, as the reference's masked_fill makes them
    xm = rnd((4, 83, 64), 14)
    mk = torch.zeros(4, 83, dtype=torch.bool)
    for row, n_valid in enumerate((83, 40, 7)):
        mk[row, :n_valid] = True
    mk[3] = torch.rand(83, generator=torch.Generator().manual_seed(15)) > 0.4
    mk[3, :70] &= (torch.arange(70) % 64 != 63)              # ... and a fully masked stretch inside one 64-key chunk
    mk[3, 64:83] = False
    save("conformer_mask.npz", x=xm, mask=mk.to(torch.uint8), out=blk(xm, mask=mk), attn=blk.attn(blk.ff1(xm) + xm, mask=mk))

    # -- 3. attention beyond max_pos_emb (clamp active, n = 600) ---------------
    xl = rnd((1, 600, 64), 12)
    save("attention_long.npz", x=xl, out=blk.attn(xl))

    # -- 4. TSCNet on a short spectrogram, with stage taps ---------------------
    xin = rnd((2, 2, 9, 201), 13, 0.5)
    mag = torch.sqrt(xin[:, 0:1] ** 2 + xin[:, 1:2] ** 2)
    enc = model.dense_encoder(torch.cat([mag, xin], dim=1))
    t1 = model.TSCB_1(enc)
    t4 = model.TSCB_4(model.TSCB_3(model.TSCB_2(t1)))
    mask = model.mask_decoder(t4)
    cplx = model.complex_decoder(t4)
    real, imag = model(xin)
    save("tscnet.npz", x=xin, encoder=enc, tscb1=t1.contiguous(), tscb4=t4.contiguous(),
         mask=mask, complex=cplx, real=real, imag=imag)

    # -- 5. whole pipeline: ragged length, and the >cut_len chunking rule ------
    noisy = synthetic_clips(1, 2350, seed=2)
    save("pipeline.npz", noisy=noisy, enhanced=ref_enhance(model, noisy, 16000 * 16),
         enhanced_chunked=ref_enhance(model, noisy, 1000), cut_len_chunked=np.int64(1000))

    # -- 6. 48 kHz variant: n_fft 1200 / hop 300, F = 601 ----------------------
    m48 = TSCNet(num_channel=64, num_features=601).eval()
    sd48 = make_state_dict(seed=5, num_features=601)
    m48.load_state_dict(sd48, strict=True)
    x48 = rnd((1, 2, 5, 601), 14, 0.5)
    r48, i48 = m48(x48)
    save("tscnet48.npz", x=x48, real=r48, imag=i48)

    # -- 7. 48 kHz variant end to end (wav -> wav, n_fft 1200 / hop 300): ragged, chunked, and a length whose
    #       100-sample padding (evaluation.py:25) is NOT a multiple of hop = 300: torch.stft then yields
    #       1 + padded // 300 frames and torch.istft returns 300 * (T - 1) samples, i.e. a SHORTER track
    n48 = synthetic_clips(1, 4150, seed=22)                   # padded 4200 = 14 hops
    n48s = synthetic_clips(1, 2450, seed=23)                  # padded 2500: T = 9, output 2400 samples
    save("pipeline48.npz", noisy=n48, enhanced=ref_enhance(m48, n48, 48000 * 16, 1200, 300),
         enhanced_chunked=ref_enhance(m48, n48, 2100, 1200, 300), cut_len_chunked=np.int64(2100),
         noisy_short=n48s, enhanced_short=ref_enhance(m48, n48s, 48000 * 16, 1200, 300))

    # -- 8. real recordings: two noisy VoiceBank+DEMAND test tracks that ship with the reference
    #       (AudioSamples/noisy, int16 PCM / 32768 as torchaudio.load does) and a clean track with
    #       0.3 s + 0.6 s of exact digital silence gated into it (zero bins through |X|^-0.7, sparse
    #       InstanceNorm planes, the split-f16 operand ranges)
    from scipy.io import wavfile
    tracks = {}
    for tag, rel in (("a", "noisy/p232_170.wav"), ("b", "noisy/p257_054.wav"), ("silence", "clean/p232_052.wav")):
        sr, pcm = wavfile.read(os.path.join("/root/reference/AudioSamples", rel))
        assert sr == 16000 and pcm.dtype == np.int16
        pcm = pcm.copy()
        if tag == "silence":
            pcm[:4800] = 0
            pcm[14000:23600] = 0
        x = torch.from_numpy(pcm.astype(np.float32) / 32768.0)[None, :]
        tracks[f"pcm_{tag}"] = pcm
        tracks[f"enhanced_{tag}"] = ref_enhance(model, x, 16000 * 16)
    save("tracks.npz", **tracks)

    # -- 9. training-mode FeedForward branch with its gradients (SURVEY.md N2 slice): the reference ConformerBlock's
    #       ff1 = Scale(0.5, PreNorm(dim, FeedForward(dim, mult=4, dropout=0.2))) in TRAIN mode, its two nn.Dropout
    #       layers replaced by multiplications with explicit keep-masks (same arithmetic as F.dropout: x * mask,
    #       mask in {0, 1/(1-p)}), differentiated by torch autograd
    class _Mask(torch.nn.Module):
        def __init__(self, m):
            super().__init__()
            self.m = m

        def forward(self, t):
            return t * self.m

    with torch.enable_grad():
        blk_t = ConformerBlock(dim=64, dim_head=16, heads=4, conv_kernel_size=31, attn_dropout=0.2, ff_dropout=0.2)
        blk_t.load_state_dict(csd, strict=True)
        blk_t.train()
        ff = blk_t.ff1
        Mtok, pdrop = 3 * 37, 0.2
        gen = torch.Generator().manual_seed(17)
        m1 = (torch.rand(3, 37, 256, generator=gen) >= pdrop).float() / (1 - pdrop)
        m2 = (torch.rand(3, 37, 64, generator=gen) >= pdrop).float() / (1 - pdrop)
        ff.fn.fn.net[2] = _Mask(m1)
        ff.fn.fn.net[4] = _Mask(m2)
        xt = rnd((3, 37, 64), 21).requires_grad_(True)
        dy = rnd((3, 37, 64), 22)
        yt = ff(xt)
        yt.backward(dy)
        grads = {k.replace(".", "_"): v.grad.detach() for k, v in ff.named_parameters()}
        # and without dropout (masks of ones = eval arithmetic), for the mask = NULL path
        ff.fn.fn.net[2] = _Mask(torch.ones(()))
        ff.fn.fn.net[4] = _Mask(torch.ones(()))
        for v in ff.parameters():
            v.grad = None
        x0 = xt.detach().clone().requires_grad_(True)
        y0 = ff(x0)
        y0.backward(dy)
        grads0 = {"nomask_" + k.replace(".", "_"): v.grad.detach() for k, v in ff.named_parameters()}
    # -- 10. generator half of Trainer.test_step (train.py:72-151, 207-220) on a tiny batch: per-row RMS scale of
    #        noisy AND clean by the noisy row's c, STFT + power_compress of both, TSCNet, losses (GAN term excluded:
    #        the metric discriminator is outside this build).  Note the reference compares est_audio (scaled domain,
    #        never divided by c) with the RAW clean batch (train.py:139-141, 218).
    import torch.nn.functional as Fn
    vclean, vnoisy = synthetic_clips(2, 1600, seed=41) * 0.5, synthetic_clips(2, 1600, seed=42)
    vc = torch.sqrt(vnoisy.size(-1) / torch.sum((vnoisy ** 2.0), dim=-1))
    n_s = torch.transpose(torch.transpose(vnoisy, 0, 1) * vc, 0, 1)
    c_s = torch.transpose(torch.transpose(vclean, 0, 1) * vc, 0, 1)
    n_spec = ref_utils.power_compress(ref_stft(n_s)).permute(0, 1, 3, 2)
    c_spec = ref_utils.power_compress(ref_stft(c_s))
    c_real, c_imag = c_spec[:, 0, :, :].unsqueeze(1), c_spec[:, 1, :, :].unsqueeze(1)
    e_real, e_imag = model(n_spec)
    e_real, e_imag = e_real.permute(0, 1, 3, 2), e_imag.permute(0, 1, 3, 2)
    e_mag = torch.sqrt(e_real ** 2 + e_imag ** 2)
    c_mag = torch.sqrt(c_real ** 2 + c_imag ** 2)
    e_audio = ref_istft(ref_utils.power_uncompress(e_real, e_imag).squeeze(1))
    l_mag = Fn.mse_loss(e_mag, c_mag)
    l_ri = Fn.mse_loss(e_real, c_real) + Fn.mse_loss(e_imag, c_imag)
    l_time = torch.mean(torch.abs(e_audio - vclean))
    save("valstep.npz", clean=vclean, noisy=vnoisy, est_audio=e_audio, loss_ri=l_ri, loss_mag=l_mag, time_loss=l_time,
         loss=0.1 * l_ri + 0.9 * l_mag + 0.2 * l_time)

    # -- 11. ConformerConvModule in TRAIN mode (BatchNorm1d on batch statistics, running stats updated with momentum
    #        0.1; conv_dropout = 0 so its Dropout is the identity) with autograd gradients        conformer.py:151-176
    with torch.enable_grad():
        blk_c = ConformerBlock(dim=64, dim_head=16, heads=4, conv_kernel_size=31, attn_dropout=0.2, ff_dropout=0.2)
        blk_c.load_state_dict(csd, strict=True)
        blk_c.train()
        cm = blk_c.conv
        xc = rnd((3, 37, 64), 31).requires_grad_(True)
        dyc = rnd((3, 37, 64), 32)
        yc = cm(xc)
        yc.backward(dyc)
        cgr = {"grad_" + k.replace(".", "_"): v.grad.detach() for k, v in cm.named_parameters()}
        bn = cm.net[5]
    save("convmod_train.npz", x=xc.detach(), dy=dyc, y=yc.detach(), dx=xc.grad.detach(),
         running_mean=bn.running_mean.detach().clone(), running_var=bn.running_var.detach().clone(), **cgr)

    # -- 12. PreNorm(Attention) in TRAIN mode: Shaw relative-position attention with nn.Dropout on the to_out output
    #        (conformer.py:100-133, 54-72) as an explicit keep-mask; gradients of x and of every parameter incl. the
    #        relative-position embedding table
    with torch.enable_grad():
        blk_a = ConformerBlock(dim=64, dim_head=16, heads=4, conv_kernel_size=31, attn_dropout=0.2, ff_dropout=0.2)
        blk_a.load_state_dict(csd, strict=True)
        blk_a.train()
        at = blk_a.attn
        gen_a = torch.Generator().manual_seed(19)
        ma = (torch.rand(3, 37, 64, generator=gen_a) >= 0.2).float() / 0.8
        at.fn.dropout = _Mask(ma)
        xa = rnd((3, 37, 64), 41).requires_grad_(True)
        dya = rnd((3, 37, 64), 42)
        ya = at(xa)
        ya.backward(dya)
        agr = {"grad_" + k.replace(".", "_"): v.grad.detach() for k, v in at.named_parameters()}
    save("attn_train.npz", x=xa.detach(), dy=dya, mask=ma, y=ya.detach(), dx=xa.grad.detach(), **agr)

    # -- 13. the WHOLE ConformerBlock in TRAIN mode (conformer.py:216-222): every Dropout an explicit keep-mask,
    #        BatchNorm1d on batch statistics; autograd gradients of x and of all 31 parameters
    with torch.enable_grad():
        blk_w = ConformerBlock(dim=64, dim_head=16, heads=4, conv_kernel_size=31, attn_dropout=0.2, ff_dropout=0.2)
        blk_w.load_state_dict(csd, strict=True)
        blk_w.train()
        gen_w = torch.Generator().manual_seed(23)
        km = lambda c: (torch.rand(2, 53, c, generator=gen_w) >= 0.2).float() / 0.8
        wm = {"ff1_1": km(256), "ff1_2": km(64), "attn": km(64), "ff2_1": km(256), "ff2_2": km(64)}
        blk_w.ff1.fn.fn.net[2], blk_w.ff1.fn.fn.net[4] = _Mask(wm["ff1_1"]), _Mask(wm["ff1_2"])
        blk_w.attn.fn.dropout = _Mask(wm["attn"])
        blk_w.ff2.fn.fn.net[2], blk_w.ff2.fn.fn.net[4] = _Mask(wm["ff2_1"]), _Mask(wm["ff2_2"])
        xw = rnd((2, 53, 64), 51).requires_grad_(True)
        dyw = rnd((2, 53, 64), 52)
        yw = blk_w(xw)
        yw.backward(dyw)
        wgr = {"grad_" + k.replace(".", "_"): v.grad.detach() for k, v in blk_w.named_parameters()}
    save("block_train.npz", x=xw.detach(), dy=dyw, y=yw.detach(), dx=xw.grad.detach(),
         **{"mask_" + k: v for k, v in wm.items()}, **wgr)

    # -- 14. one TSCB (generator.py:72-99) in TRAIN mode: time conformer over T per (b, f') + residual, frequency
    #        conformer over F' per (b, t) + residual; ten keep-masks; input / output NCHW [b, 64, t, f'] as in the model
    from models.generator import TSCB
    with torch.enable_grad():
        tscb = TSCB(num_channel=64)
        tsd = {k[len("TSCB_1."):]: v for k, v in sd.items() if k.startswith("TSCB_1.")}
        tscb.load_state_dict(tsd, strict=True)
        tscb.train()
        bT, tT, fT = 2, 19, 11
        gen_t = torch.Generator().manual_seed(29)
        tm = {}
        for ax, (nn_, ll) in (("time", (bT * fT, tT)), ("freq", (bT * tT, fT))):
            conf = getattr(tscb, ax + "_conformer")
            kmt = lambda c: (torch.rand(nn_, ll, c, generator=gen_t) >= 0.2).float() / 0.8
            mk = {"ff1_1": kmt(256), "ff1_2": kmt(64), "attn": kmt(64), "ff2_1": kmt(256), "ff2_2": kmt(64)}
            conf.ff1.fn.fn.net[2], conf.ff1.fn.fn.net[4] = _Mask(mk["ff1_1"]), _Mask(mk["ff1_2"])
            conf.attn.fn.dropout = _Mask(mk["attn"])
            conf.ff2.fn.fn.net[2], conf.ff2.fn.fn.net[4] = _Mask(mk["ff2_1"]), _Mask(mk["ff2_2"])
            tm.update({f"mask_{ax}_{k}": v for k, v in mk.items()})
        xs = rnd((bT, 64, tT, fT), 61).requires_grad_(True)
        dys = rnd((bT, 64, tT, fT), 62)
        ys = tscb(xs)
        ys.backward(dys)
        # a few representative parameter gradients (the full set is checked per block by block_train.npz)
        pick = ("time_conformer.ff1.fn.fn.net.0.weight", "time_conformer.attn.fn.rel_pos_emb.weight",
                "time_conformer.conv.net.4.conv.weight", "freq_conformer.attn.fn.to_kv.weight",
                "freq_conformer.conv.net.5.weight", "freq_conformer.post_norm.weight")
        named = dict(tscb.named_parameters())
        tgr = {"grad_" + k.replace(".", "_"): named[k].grad.detach() for k in pick}
    save("tscb_train.npz", x=xs.detach(), dy=dys, y=ys.detach().contiguous(), dx=xs.grad.detach(), **tm, **tgr)

    # -- 15. DilatedDenseNet (generator.py:6-47; the encoder's instance) with autograd gradients: dilated (2,3) convs
    #        over a growing concat, InstanceNorm2d(affine) (identical in train and eval), PReLU; T > 8 so that the
    #        dilation-8 layer sees real history
    from models.generator import DilatedDenseNet
    with torch.enable_grad():
        ddn = DilatedDenseNet(depth=4, in_channels=64)
        dsd = {k[len("dense_encoder.dilated_dense."):]: v for k, v in sd.items()
               if k.startswith("dense_encoder.dilated_dense.")}
        ddn.load_state_dict(dsd, strict=True)
        ddn.train()
        xd = rnd((2, 64, 13, 11), 71).requires_grad_(True)
        dyd = rnd((2, 64, 13, 11), 72)
        yd = ddn(xd)
        yd.backward(dyd)
        dgr = {"grad_" + k.replace(".", "_"): v.grad.detach() for k, v in ddn.named_parameters()}
    save("dense_train.npz", x=xd.detach(), dy=dyd, y=yd.detach(), dx=xd.grad.detach(), **dgr)

    # -- 16. DenseEncoder (generator.py:50-69): conv_1 (1x1, 3 -> 64) + IN + PReLU, the dilated dense block, conv_2
    #        ((1,3), stride (1,2), padding (0,1)) + IN + PReLU, with autograd gradients of all parameters
    from models.generator import DenseEncoder
    with torch.enable_grad():
        enc = DenseEncoder(in_channel=3, channels=64)
        esd = {k[len("dense_encoder."):]: v for k, v in sd.items() if k.startswith("dense_encoder.")}
        enc.load_state_dict(esd, strict=True)
        enc.train()
        xe = rnd((2, 3, 13, 21), 81)                       # [mag, re, im] planes, F = 21 -> F' = 11
        dye = rnd((2, 64, 13, 11), 82)
        ye = enc(xe)
        ye.backward(dye)
        egr = {"grad_" + k.replace(".", "_"): v.grad.detach() for k, v in enc.named_parameters()}
    save("encoder_train.npz", x=xe, dy=dye, y=ye.detach(), **egr)

    # -- 17/18. MaskDecoder(num_features = 21) and ComplexDecoder (generator.py:121-156) with autograd gradients:
    #        dilated dense block, sub-pixel conv (pad, (1,3) conv to 128 channels, pixel shuffle x2 along F), heads
    from models.generator import ComplexDecoder, MaskDecoder
    with torch.enable_grad():
        mdec = MaskDecoder(num_features=21, num_channel=64, out_channel=1)
        msd = {k[len("mask_decoder."):]: v for k, v in sd.items() if k.startswith("mask_decoder.")}
        msd["prelu_out.weight"] = msd["prelu_out.weight"][:21].clone()
        mdec.load_state_dict(msd, strict=True)
        mdec.train()
        xm = rnd((2, 64, 13, 11), 91).requires_grad_(True)
        dym = rnd((2, 1, 13, 21), 92)
        ym = mdec(xm)
        ym.backward(dym)
        mgr = {"grad_" + k.replace(".", "_"): v.grad.detach() for k, v in mdec.named_parameters()}
    save("maskdec_train.npz", x=xm.detach(), dy=dym, y=ym.detach(), dx=xm.grad.detach(), **mgr)
    with torch.enable_grad():
        cdec = ComplexDecoder(num_channel=64)
        cdec.load_state_dict({k[len("complex_decoder."):]: v for k, v in sd.items() if k.startswith("complex_decoder.")},
                             strict=True)
        cdec.train()
        xc = rnd((2, 64, 13, 11), 93).requires_grad_(True)
        dyc = rnd((2, 2, 13, 21), 94)
        yc = cdec(xc)
        yc.backward(dyc)
        cgr = {"grad_" + k.replace(".", "_"): v.grad.detach() for k, v in cdec.named_parameters()}
    save("complexdec_train.npz", x=xc.detach(), dy=dyc, y=yc.detach(), dx=xc.grad.detach(), **cgr)

    # -- 19. one generator optimisation step of the reference trainer without the metric discriminator:
    #        Trainer.forward_generator_step (train.py:72-122) on the reference TSCNet in TRAIN mode (all 40 Dropout
    #        layers replaced by the deterministic keep-masks of cmgan_amd.synth.synthetic_dropout_masks),
    #        loss = 0.1 loss_ri + 0.9 loss_mag + 0.2 time_loss (train.py:133-148, the RAW clean batch in the time
    #        term as train.py:187 does), loss.backward(), AdamW(lr 5e-4).step(), and the loss of a second forward.
    #        Stored: the losses, est_real / est_imag and their gradients, per-parameter gradient digests (sum, l2,
    #        and up to 256 fixed samples - the full set is 1.8 M floats), the running statistics of one BatchNorm.
    from cmgan_amd.synth import sample_indices, synthetic_dropout_masks
    with torch.enable_grad():
        gmodel = TSCNet(num_channel=64, num_features=201)
        gmodel.load_state_dict(sd, strict=True)
        gmodel.train()
        Bg, Lg = 2, 800
        Tg, Feg = Lg // 100 + 1, 101
        gm = synthetic_dropout_masks(77, Bg, Tg, Feg)
        for bi, name in enumerate(("TSCB_1", "TSCB_2", "TSCB_3", "TSCB_4")):
            for ai, ax in enumerate(("time_conformer", "freq_conformer")):
                conf = getattr(getattr(gmodel, name), ax)
                mk = {k: torch.from_numpy(v) for k, v in gm[bi][ai].items()}
                conf.ff1.fn.fn.net[2], conf.ff1.fn.fn.net[4] = _Mask(mk["ff1_1"]), _Mask(mk["ff1_2"])
                conf.attn.fn.dropout = _Mask(mk["attn"])
                conf.ff2.fn.fn.net[2], conf.ff2.fn.fn.net[4] = _Mask(mk["ff2_1"]), _Mask(mk["ff2_2"])
        gclean = synthetic_clips(Bg, Lg, seed=31)
        gnoisy = gclean + 0.3 * synthetic_clips(Bg, Lg, seed=32)
        win = torch.hamming_window(400)

        def gen_step():
            c = torch.sqrt(gnoisy.size(-1) / torch.sum(gnoisy ** 2.0, dim=-1))
            noisy_s, clean_s = gnoisy * c[:, None], gclean * c[:, None]
            nspec = torch.view_as_real(torch.stft(noisy_s, 400, 100, window=win, onesided=True, return_complex=True))
            cspec = torch.view_as_real(torch.stft(clean_s, 400, 100, window=win, onesided=True, return_complex=True))
            nspec = ref_utils.power_compress(nspec).permute(0, 1, 3, 2)
            cspec = ref_utils.power_compress(cspec)
            clean_real, clean_imag = cspec[:, 0:1], cspec[:, 1:2]
            er0, ei0 = gmodel(nspec)                                   # [B,1,T,F]
            if er0.requires_grad:
                er0.retain_grad(); ei0.retain_grad()
            er, ei = er0.permute(0, 1, 3, 2), ei0.permute(0, 1, 3, 2)
            est_mag = torch.sqrt(er ** 2 + ei ** 2)
            clean_mag = torch.sqrt(clean_real ** 2 + clean_imag ** 2)
            unc = ref_utils.power_uncompress(er, ei).squeeze(1)
            est_audio = torch.istft(torch.view_as_complex(unc.contiguous()), 400, 100, window=win, onesided=True)
            F_ = torch.nn.functional
            loss_mag = F_.mse_loss(est_mag, clean_mag)
            loss_ri = F_.mse_loss(er, clean_real) + F_.mse_loss(ei, clean_imag)
            time_loss = torch.mean(torch.abs(est_audio - gclean))
            loss = 0.1 * loss_ri + 0.9 * loss_mag + 0.2 * time_loss
            return loss, (loss_ri, loss_mag, time_loss), er0, ei0

        opt = torch.optim.AdamW(gmodel.parameters(), lr=5e-4)
        loss1, terms1, er0, ei0 = gen_step()
        opt.zero_grad()
        loss1.backward()
        digest = {}
        for k, v in gmodel.named_parameters():
            gflat = v.grad.detach().reshape(-1)
            idx = torch.from_numpy(sample_indices(gflat.numel()))
            digest["gsum_" + k] = gflat.double().sum().float()
            digest["gl2_" + k] = gflat.double().norm().float()
            digest["gsmp_" + k] = gflat[idx].clone()
        d_er, d_ei = er0.grad.detach().clone(), ei0.grad.detach().clone()
        opt.step()
        with torch.no_grad():
            loss2, terms2, _, _ = gen_step()
        bn = gmodel.TSCB_2.freq_conformer.conv.net[5]
    save("generator_step.npz", clean=gclean, noisy=gnoisy, loss=loss1.detach(), terms=torch.stack(terms1).detach(),
         est_real=er0.detach(), est_imag=ei0.detach(), d_real=d_er, d_imag=d_ei, loss2=loss2.detach(),
         terms2=torch.stack(terms2).detach(), bn_mean=bn.running_mean.detach().clone(),
         bn_var=bn.running_var.detach().clone(), **digest)

    # -- 20. the metric discriminator (src/models/discriminator.py:29-64) in TRAIN mode: one power iteration of every
    #        spectral norm per forward (u / v buffers updated), Dropout(0.3) as an explicit keep-mask, autograd
    #        gradients of the score with respect to both inputs and all parameters; plus the eval-mode score afterwards.
    #        The module imports the `pesq` wheel (absent here) at the top for batch_pesq only: a stub module stands in.
    import types
    sys.modules.setdefault("pesq", types.SimpleNamespace(pesq=None))
    from models.discriminator import Discriminator
    from cmgan_amd.synth import discriminator_state_dict
    dsd = discriminator_state_dict(0)
    with torch.enable_grad():
        disc = Discriminator(ndf=16)
        disc.load_state_dict(dsd, strict=True)
        xdm = rnd((2, 1, 201, 33), 101).abs()
        ydm = rnd((2, 1, 201, 33), 102).abs()
        disc.train()
        mdm = (torch.rand(2, 64, generator=torch.Generator().manual_seed(103)) >= 0.3).float() / 0.7
        disc.layers[15] = _Mask(mdm)
        xdl, ydl = xdm.clone().requires_grad_(True), ydm.clone().requires_grad_(True)
        score = disc(xdl, ydl)
        dscore = rnd((2, 1), 104)
        score.backward(dscore)
        dgrads = {"grad_" + k.replace(".", "_"): v.grad.detach() for k, v in disc.named_parameters()}
        duv = {"new_" + k.replace(".", "_"): v.detach().clone() for k, v in disc.state_dict().items()
               if k.endswith("_u") or k.endswith("_v")}
        disc.eval()                                   # eval mode: no power iteration (the UPDATED u / v), no dropout
        disc.layers[15] = torch.nn.Identity()
        with torch.no_grad():
            score_eval = disc(xdm, ydm)
    save("disc_train.npz", x=xdm, y=ydm, mask=mdm, score_eval=score_eval, score=score.detach(), dscore=dscore,
         dx=xdl.grad.detach(), dy=ydl.grad.detach(), **dgrads, **duv)

    # -- 21. one FULL adversarial training step of the reference trainer (train.py:153-205) at B = 2, L = 3200 (T = 33):
    #        generator loss incl. 0.05 * gen_loss_GAN through the discriminator, AdamW(5e-4) on the generator, then the
    #        discriminator loss mse(D(clean, clean), 1) + mse(D(clean, est.detach()), pesq) with GIVEN normalised PESQ
    #        labels (the pesq wheel is absent; the labels are data as far as the device step is concerned),
    #        AdamW(1e-3) on the discriminator; and the generator loss of a second step.  Dropout masks: generator
    #        synthetic_dropout_masks(78), discriminator RandomState(79) keep-masks [3 calls][B, 64].
    with torch.enable_grad():
        amodel = TSCNet(num_channel=64, num_features=201)
        amodel.load_state_dict(sd, strict=True)
        amodel.train()
        adisc = Discriminator(ndf=16)
        adisc.load_state_dict(dsd, strict=True)
        adisc.train()
        Ba, La = 2, 3200
        Ta = La // 100 + 1
        am = synthetic_dropout_masks(78, Ba, Ta, 101)
        for bi, name in enumerate(("TSCB_1", "TSCB_2", "TSCB_3", "TSCB_4")):
            for ai, ax in enumerate(("time_conformer", "freq_conformer")):
                conf = getattr(getattr(amodel, name), ax)
                mk = {k: torch.from_numpy(v) for k, v in am[bi][ai].items()}
                conf.ff1.fn.fn.net[2], conf.ff1.fn.fn.net[4] = _Mask(mk["ff1_1"]), _Mask(mk["ff1_2"])
                conf.attn.fn.dropout = _Mask(mk["attn"])
                conf.ff2.fn.fn.net[2], conf.ff2.fn.fn.net[4] = _Mask(mk["ff2_1"]), _Mask(mk["ff2_2"])
        drs = np.random.RandomState(79)
        dmasks = [torch.from_numpy(((drs.random_sample((Ba, 64)) >= 0.3).astype(np.float32) / np.float32(0.7)))
                  for _ in range(6)]
        aclean = synthetic_clips(Ba, La, seed=41)
        anoisy = aclean + 0.3 * synthetic_clips(Ba, La, seed=42)
        pesq_lab = torch.tensor([0.45, 0.8])
        ones = torch.ones(Ba)
        F_ = torch.nn.functional
        dcall = [0]

        def D(a, b_):
            adisc.layers[15] = _Mask(dmasks[dcall[0]])
            dcall[0] += 1
            return adisc(a, b_)

        def fwd_gen():
            c = torch.sqrt(anoisy.size(-1) / torch.sum(anoisy ** 2.0, dim=-1))
            noisy_s, clean_s = anoisy * c[:, None], aclean * c[:, None]
            nspec = torch.view_as_real(torch.stft(noisy_s, 400, 100, window=win, onesided=True, return_complex=True))
            cspec = torch.view_as_real(torch.stft(clean_s, 400, 100, window=win, onesided=True, return_complex=True))
            nspec = ref_utils.power_compress(nspec).permute(0, 1, 3, 2)
            cspec = ref_utils.power_compress(cspec)
            clean_real, clean_imag = cspec[:, 0:1], cspec[:, 1:2]
            er0, ei0 = amodel(nspec)
            er, ei = er0.permute(0, 1, 3, 2), ei0.permute(0, 1, 3, 2)
            est_mag = torch.sqrt(er ** 2 + ei ** 2)
            clean_mag = torch.sqrt(clean_real ** 2 + clean_imag ** 2)
            unc = ref_utils.power_uncompress(er, ei).squeeze(1)
            est_audio = torch.istft(torch.view_as_complex(unc.contiguous()), 400, 100, window=win, onesided=True)
            gan = F_.mse_loss(D(clean_mag, est_mag).flatten(), ones)
            loss_mag = F_.mse_loss(est_mag, clean_mag)
            loss_ri = F_.mse_loss(er, clean_real) + F_.mse_loss(ei, clean_imag)
            time_loss = torch.mean(torch.abs(est_audio - aclean))
            loss = 0.1 * loss_ri + 0.9 * loss_mag + 0.2 * time_loss + 0.05 * gan
            return loss, gan, est_mag, clean_mag

        opt_g = torch.optim.AdamW(amodel.parameters(), lr=5e-4)
        opt_d = torch.optim.AdamW(adisc.parameters(), lr=1e-3)
        loss_g, gan_g, est_mag, clean_mag = fwd_gen()
        opt_g.zero_grad()
        loss_g.backward()
        adig = {}
        for k, v in amodel.named_parameters():
            gflat = v.grad.detach().reshape(-1)
            idx = torch.from_numpy(sample_indices(gflat.numel()))
            adig["gl2_" + k] = gflat.double().norm().float()
            adig["gsmp_" + k] = gflat[idx].clone()
        opt_g.step()
        p_enh = D(clean_mag.detach(), est_mag.detach())
        p_max = D(clean_mag.detach(), clean_mag.detach())
        loss_d = F_.mse_loss(p_max.flatten(), ones) + F_.mse_loss(p_enh.flatten(), pesq_lab)
        opt_d.zero_grad()
        loss_d.backward()
        ddig = {"dgrad_" + k.replace(".", "_"): v.grad.detach().clone() for k, v in adisc.named_parameters()}
        opt_d.step()
        with torch.no_grad():
            loss_g2, gan_g2, _, _ = fwd_gen()
    save("adversarial_step.npz", clean=aclean, noisy=anoisy, pesq=pesq_lab, loss=loss_g.detach(), gan=gan_g.detach(),
         loss_d=loss_d.detach(), p_enh=p_enh.detach(), p_max=p_max.detach(), loss2=loss_g2.detach(),
         gan2=gan_g2.detach(), **adig, **ddig)

    save("ffn_train.npz", x=xt.detach(), dy=dy, mask1=m1, mask2=m2, y=yt.detach(), dx=xt.grad.detach(),
         y_nomask=y0.detach(), dx_nomask=x0.grad.detach(), **grads, **grads0)


if __name__ == "__main__":
    main()
