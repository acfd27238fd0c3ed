"""ctypes binding of libcmgan_hip.so (C ABI: include/cmgan_hip.h).

There is no fallback: if the shared library is missing or does not export a
declared symbol, importing the binding raises - the HIP kernels ARE the product.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import POINTER, Structure, c_char_p, c_float, c_int, c_int32, c_longlong, c_size_t, c_void_p

HERE = os.path.dirname(os.path.abspath(__file__))
# CMGAN_HIP_LIB selects an alternative BUILD of the same HIP library (cmgan_amd.build variants, used for
# same-session A/B timing); it is never a fallback: a missing file is a hard error either way.
LIB_PATH = os.environ.get("CMGAN_HIP_LIB") or os.path.join(HERE, "lib", "libcmgan_hip.so")

OK = 0
ABI_VERSION = 6
MFMA_F32, MFMA_F16X3, MFMA_F16X1, MFMA_F16MIX = 0, 1, 2, 3
# CMGAN_MIX_* bits of Config.single_mask (include/cmgan_hip.h)
MIX = {"conv": 1, "ff1": 2, "ff2": 4, "qkv": 8, "attn": 16, "pw1": 32, "dwpw2": 64}
MIX_ALL = 127


class CmganError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"libcmgan_hip error {code}: {msg}")
        self.code = code


class Config(Structure):
    _fields_ = [(n, c_int32) for n in ("n_fft", "hop", "num_features", "num_channel", "num_tscb",
                                       "heads", "dim_head", "conv_kernel", "max_pos_emb", "mfma_mode", "single_mask")]


class Taps(Structure):
    _fields_ = [("encoder_dev", c_void_p), ("tscb_dev", c_void_p * 4), ("mask_dev", c_void_p),
                ("complex_dev", c_void_p)]


class FfnParams(Structure):
    _fields_ = [(n, c_void_p) for n in ("ln_weight", "ln_bias", "w1", "b1", "w2", "b2")]


class ConvModParams(Structure):
    _fields_ = [(n, c_void_p) for n in ("ln_weight", "ln_bias", "pw1_weight", "pw1_bias", "dw_weight", "dw_bias",
                                        "bn_weight", "bn_bias", "pw2_weight", "pw2_bias")]


class AttnParams(Structure):
    _fields_ = [(n, c_void_p) for n in ("ln_weight", "ln_bias", "to_q_weight", "to_kv_weight", "to_out_weight",
                                        "to_out_bias", "rel_pos_emb")]


class DenseParams(Structure):
    _fields_ = [(n, c_void_p * 4) for n in ("conv_weight", "conv_bias", "norm_weight", "norm_bias", "prelu_weight")]


class EncoderParams(Structure):
    _fields_ = ([(n, c_void_p) for n in ("conv1_weight", "conv1_bias", "norm1_weight", "norm1_bias", "prelu1_weight")]
                + [("dense", DenseParams)]
                + [(n, c_void_p) for n in ("conv2_weight", "conv2_bias", "norm2_weight", "norm2_bias", "prelu2_weight")])


class DecoderParams(Structure):
    _fields_ = ([("dense", DenseParams)]
                + [(n, c_void_p) for n in ("sub_pixel_weight", "sub_pixel_bias", "conv_weight", "conv_bias",
                                           "norm_weight", "norm_bias", "prelu_weight", "final_weight", "final_bias",
                                           "prelu_out_weight")])


class DiscParams(Structure):
    _fields_ = ([(n, c_void_p * 4) for n in ("conv_weight_orig", "conv_u", "conv_v", "norm_weight", "norm_bias",
                                             "prelu_weight")]
                + [(n, c_void_p) for n in ("fc1_weight_orig", "fc1_bias", "fc1_u", "fc1_v", "prelu5_weight",
                                           "fc2_weight_orig", "fc2_bias", "fc2_u", "fc2_v", "slope")])


class KernelTime(Structure):
    _fields_ = [("name", c_char_p), ("ms", c_float)]


# name -> (restype, argtypes); must list every symbol include/cmgan_hip.h declares
SIGNATURES = {
    "cmgan_default_config": (None, [POINTER(Config)]),
    "cmgan_abi_version": (c_int, []),
    "cmgan_create": (c_int, [POINTER(c_void_p), POINTER(Config)]),
    "cmgan_destroy": (None, [c_void_p]),
    "cmgan_last_error": (c_char_p, [c_void_p]),
    "cmgan_load_weights": (c_int, [c_void_p, c_void_p, c_size_t]),
    "cmgan_weights_generation": (c_int, [c_void_p]),
    "cmgan_loss_terms": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_int,
                                 c_void_p, c_void_p]),
    "cmgan_ffn_train_workspace_bytes": (c_size_t, [c_void_p, c_longlong]),
    "cmgan_ffn_train_forward": (c_int, [c_void_p, c_void_p, c_longlong, POINTER(FfnParams), c_void_p, c_void_p, c_float,
                                        c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "cmgan_ffn_train_backward": (c_int, [c_void_p, c_void_p, c_void_p, c_longlong, POINTER(FfnParams), c_void_p,
                                         c_void_p, c_float, c_void_p, c_void_p, POINTER(FfnParams), c_void_p, c_size_t,
                                         c_void_p]),
    "cmgan_convmod_train_workspace_bytes": (c_size_t, [c_void_p, c_int, c_int]),
    "cmgan_convmod_train_forward": (c_int, [c_void_p, c_void_p, c_int, c_int, POINTER(ConvModParams), c_void_p,
                                            c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "cmgan_convmod_train_backward": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, POINTER(ConvModParams),
                                             c_void_p, c_void_p, POINTER(ConvModParams), c_void_p, c_size_t, c_void_p]),
    "cmgan_attn_train_workspace_bytes": (c_size_t, [c_void_p, c_int, c_int]),
    "cmgan_attn_train_forward": (c_int, [c_void_p, c_void_p, c_int, c_int, POINTER(AttnParams), c_void_p, c_float,
                                         c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "cmgan_attn_train_backward": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, POINTER(AttnParams), c_void_p,
                                          c_float, c_void_p, c_void_p, POINTER(AttnParams), c_void_p, c_size_t, c_void_p]),
    "cmgan_swap_axes": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "cmgan_add": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_longlong, c_void_p]),
    "cmgan_dropout_masks": (c_int, [c_void_p, c_void_p, c_longlong, c_float, c_void_p, c_void_p]),
    "cmgan_layernorm_train_workspace_bytes": (c_size_t, [c_void_p, c_longlong]),
    "cmgan_layernorm_train_forward": (c_int, [c_void_p, c_void_p, c_longlong, c_void_p, c_void_p, c_void_p, c_void_p,
                                              c_void_p]),
    "cmgan_layernorm_train_backward": (c_int, [c_void_p, c_void_p, c_void_p, c_longlong, c_void_p, c_void_p, c_void_p,
                                               c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "cmgan_dense_train_workspace_bytes": (c_size_t, [c_void_p, c_int, c_int, c_int]),
    "cmgan_dense_train_forward": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, POINTER(DenseParams), c_void_p,
                                          c_void_p, c_size_t, c_void_p]),
    "cmgan_dense_train_backward": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, POINTER(DenseParams),
                                           c_void_p, POINTER(DenseParams), c_void_p, c_size_t, c_void_p]),
    "cmgan_encoder_train_workspace_bytes": (c_size_t, [c_void_p, c_int, c_int, c_int]),
    "cmgan_encoder_train_forward": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, POINTER(EncoderParams), c_void_p,
                                            c_void_p, c_size_t, c_void_p]),
    "cmgan_encoder_train_backward": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, POINTER(EncoderParams),
                                             POINTER(EncoderParams), c_void_p, c_size_t, c_void_p]),
    "cmgan_decoder_train_workspace_bytes": (c_size_t, [c_void_p, c_int, c_int, c_int]),
    "cmgan_decoder_train_forward": (c_int, [c_void_p, c_int, c_void_p, c_int, c_int, c_int, POINTER(DecoderParams),
                                            c_void_p, c_void_p, c_size_t, c_void_p]),
    "cmgan_decoder_train_backward": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_int, c_int, c_int,
                                             POINTER(DecoderParams), c_void_p, POINTER(DecoderParams), c_void_p,
                                             c_size_t, c_void_p]),
    "cmgan_tscnet_prologue": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p]),
    "cmgan_tscnet_epilogue_forward": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p,
                                              c_void_p]),
    "cmgan_tscnet_epilogue_backward": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p,
                                               c_void_p, c_void_p]),
    "cmgan_loss_backward": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_float,
                                    c_float, c_float, c_void_p, c_void_p, c_void_p]),
    "cmgan_disc_workspace_bytes": (c_size_t, [c_void_p, c_int, c_int]),
    "cmgan_disc_forward": (c_int, [c_void_p, c_void_p, c_int, c_int, POINTER(DiscParams), c_void_p, c_int, c_void_p,
                                   c_void_p, c_size_t, c_void_p]),
    "cmgan_disc_backward": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, POINTER(DiscParams), c_void_p, c_void_p,
                                    POINTER(DiscParams), c_void_p, c_size_t, c_void_p]),
    "cmgan_mag_pair": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p]),
    "cmgan_mag_pair_backward": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_float, c_void_p,
                                        c_void_p, c_void_p]),
    "cmgan_score_mse": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_float, c_void_p, c_void_p, c_void_p]),
    "cmgan_adamw_step": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_longlong, c_float, c_float,
                                 c_float, c_float, c_float, c_int, c_void_p]),
    "cmgan_adamw_step_dev": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_longlong, c_void_p, c_float,
                                     c_float, c_float, c_float, c_void_p]),
    "cmgan_workspace_bytes": (c_size_t, [c_void_p, c_int, c_int]),
    "cmgan_rms_scale": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p]),
    "cmgan_num_frames": (c_int, [c_void_p, c_int]),
    "cmgan_stft_compress": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p]),
    "cmgan_tscnet_forward": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "cmgan_uncompress_istft": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_size_t, c_void_p]),
    "cmgan_enhance": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_size_t, c_void_p]),
    "cmgan_enhance_branched": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_size_t, c_void_p, c_int, c_int]),
    "cmgan_workspace_bytes_branched": (c_size_t, [c_void_p, c_int, c_int, c_int]),
    "cmgan_power_compress": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]),
    "cmgan_power_uncompress": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]),
    "cmgan_conformer_workspace_bytes": (c_size_t, [c_void_p, c_int, c_int]),
    "cmgan_conformer_forward": (c_int, [c_void_p, c_int, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "cmgan_conformer_forward_masked": (c_int, [c_void_p, c_int, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p,
                                               c_void_p, c_size_t, c_void_p]),
    "cmgan_tscnet_forward_taps": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, POINTER(Taps), c_void_p, c_size_t, c_void_p]),
    "cmgan_stats_floats": (c_size_t, [c_void_p, c_int]),
    "cmgan_tscnet_forward_stats": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                           c_size_t, c_void_p]),
    "cmgan_stream_encoder": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "cmgan_stream_tscb": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_size_t, c_void_p]),
    "cmgan_stream_decoder": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p,
                                     c_size_t, c_void_p]),
    "cmgan_selftest_mfma": (c_int, [c_void_p, POINTER(c_float)]),
    "cmgan_selftest_mfma_x3": (c_int, [c_void_p, POINTER(c_float)]),
    "cmgan_set_profiling": (c_int, [c_void_p, c_int]),
    "cmgan_profile_read": (c_int, [c_void_p, POINTER(KernelTime), c_int]),
}

_lib = None


def load() -> ctypes.CDLL:
    """Load the library once; raise loudly if it is absent or incomplete."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} not found: build the HIP extension first (python -m cmgan_amd.build). "
            "cmgan_amd has no CPU / eager fallback by design.")
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if a declared symbol is missing
        fn.restype = res
        fn.argtypes = args
    got = lib.cmgan_abi_version()
    if got != ABI_VERSION:
        raise RuntimeError(f"libcmgan_hip ABI version {got}, binding expects {ABI_VERSION}")
    _lib = lib
    return lib


def default_config() -> Config:
    cfg = Config()
    load().cmgan_default_config(ctypes.byref(cfg))
    return cfg


def check(handle, rc: int):
    if rc != OK:
        msg = load().cmgan_last_error(handle)
        raise CmganError(rc, msg.decode() if msg else "?")
