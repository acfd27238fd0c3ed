"""CPU restatement of the carried-state streaming contract (cmgan_amd/streaming.py::enhance_stream; include/cmgan_hip.h,
"Streaming").  TEST INFRASTRUCTURE ONLY.

The reference has no streaming mode; what it does have is time-causal dilated convs in
the dense encoder and both decoders (generator.py:16-20, 39-47) whose frames are coupled ONLY by InstanceNorm2d's
statistics over all of T (generator.py:35,55,61,128,148).  The contract freezes those statistics (`cmgan_oracle.norm_stats`:
the reference modules' own arithmetic with the mean / variance of a calibration pass held, as BatchNorm does in eval
mode) and then carries state exactly where it exists:

    encoder, decoders : 15 frames of INPUT history in front of the new frames, first 15 outputs dropped (receptive
                        field 1 + 2 + 4 + 8 frames back) - equal to the whole-clip pass under the same statistics;
    TSCBs             : bidirectional attention + a 31-tap depthwise conv over time: run on [context | window |
                        look-ahead] frames of cached / fresh encoder outputs, the window's frames kept.
"""
import torch

from . import cmgan_oracle as O

HIST = 15


@torch.no_grad()
def calibrate(sd, spec):
    """InstanceNorm statistics of TSCNet.forward on `spec` [B,2,T,F] (the frozen set of a stream)."""
    with O.norm_stats("record", {}) as st:
        O.tscnet_forward(sd, spec)
    return st


def _encoder(sd, spec):
    re, im = spec[:, 0:1], spec[:, 1:2]
    mag = torch.sqrt(re * re + im * im)
    return O.dense_encoder(sd, torch.cat([mag, spec], dim=1))


def _decoders(sd, h, spec):
    """mask / complex decoders + recombination (generator.py:187-194) on h [B,64,T,F'], spec [B,2,T,F]."""
    re, im = spec[:, 0:1], spec[:, 1:2]
    mag = torch.sqrt(re * re + im * im)
    phase = torch.atan2(im, re)
    mask = O.mask_decoder(sd, h)
    cplx = O.complex_decoder(sd, h)
    out_mag = mask * mag
    return out_mag * torch.cos(phase) + cplx[:, 0:1], out_mag * torch.sin(phase) + cplx[:, 1:2]


@torch.no_grad()
def encoder_frozen(sd, spec, stats):
    with O.norm_stats("replay", stats):
        return _encoder(sd, spec)


@torch.no_grad()
def decoders_frozen(sd, h, spec, stats):
    with O.norm_stats("replay", stats):
        return _decoders(sd, h, spec)


@torch.no_grad()
def stream_forward(sd, spec, stats, window: int, context: int, lookahead: int):
    """spec [B,2,T,F] -> (est_real, est_imag) [B,1,T,F] by the step rule of the module header (frames)."""
    T = spec.size(2)
    real = torch.empty(spec.size(0), 1, T, spec.size(3))
    imag = torch.empty_like(real)
    enc = None                                   # encoder outputs of frames [0, e1)  ([B,64,t,F'])
    kept_all = None                              # kept TSCB outputs of frames [0, k W)
    e1, k = 0, 0
    with O.norm_stats("replay", stats):
        while e1 < T:
            lo, e0 = k * window, e1
            e1 = min((k + 1) * window + lookahead, T)
            h0 = max(e0 - HIST, 0)
            new = _encoder(sd, spec[:, :, h0:e1])[:, :, e0 - h0:]
            enc = new if enc is None else torch.cat([enc, new], dim=2)
            a0 = max(lo - context, 0)
            x = enc[:, :, a0:e1]
            for b in range(1, 5):
                x = O.tscb(sd, f"TSCB_{b}", x)
            n_keep = e1 - lo if e1 == T else min(window, e1 - lo)      # the last step emits what is left
            kept = x[:, :, lo - a0:lo - a0 + n_keep]
            kept_all = kept if kept_all is None else torch.cat([kept_all, kept], dim=2)
            d0 = max(lo - HIST, 0)
            r, i = _decoders(sd, kept_all[:, :, d0:lo + n_keep], spec[:, :, d0:lo + n_keep])
            real[:, :, lo:lo + n_keep], imag[:, :, lo:lo + n_keep] = r[:, :, lo - d0:], i[:, :, lo - d0:]
            k += 1
    return real, imag


@torch.no_grad()
def enhance_stream(sd, noisy, window: int = 400, context: int = 40, lookahead: int = 40, calib_frames=None,
                   n_fft: int = 400, hop: int = 100):
    """noisy [1, L] -> enhanced [hop (T - 1)]: file-level RMS scale, whole-clip STFT (frame-local), the steps, ISTFT."""
    c = O.rms_scale(noisy)
    spec = O.stft_compress(noisy * c[:, None], n_fft, hop)
    T = spec.size(2)
    n = min(T, calib_frames if calib_frames is not None else window + lookahead)
    stats = calibrate(sd, spec[:, :, :n])
    real, imag = stream_forward(sd, spec, stats, window, context, lookahead)
    return (O.uncompress_istft(real, imag, n_fft, hop) / c[:, None]).reshape(-1)
