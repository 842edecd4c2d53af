"""ctypes binding of libf5hip.so (include/f5_hip.h).  There is NO fallback: if the library is missing or a call fails
this module raises -- the product path never computes on the CPU or through PyTorch ops."""
from __future__ import annotations

import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libf5hip.so")

F5_OPT_QK_RMSNORM, F5_OPT_LONG_SKIP, F5_OPT_TEXT_AVG_UPSAMPLE = 1, 2, 4
F5_PREC_F32, F5_PREC_BF16, F5_PREC_F16, F5_PREC_F16X3, F5_PREC_F16P = 0, 1, 2, 3, 4
PRECISIONS = {"f32": F5_PREC_F32, "fp32": F5_PREC_F32, "bf16": F5_PREC_BF16, "f16": F5_PREC_F16, "fp16": F5_PREC_F16,
              "f16x3": F5_PREC_F16X3, "f16p": F5_PREC_F16P, "parity": F5_PREC_F16P}
F5_BACKBONE_DIT, F5_BACKBONE_UNETT = 0, 1
ACT_NONE, ACT_GELU_TANH, ACT_GELU_ERF, ACT_SILU, ACT_MISH = 0, 1, 2, 3, 4
PROFILE_CLASSES = ("gemm", "attention", "layernorm", "convpos", "misc", "text_encoder", "time_adaln")


class F5Error(RuntimeError):
    pass


class f5_config(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "backbone", "precision", "dim", "depth", "heads", "dim_head", "ff_dim", "text_dim", "conv_layers",
        "pe_attn_head", "text_mask_padding", "attn_mask_enabled", "text_num_embeds", "mel_dim", "max_pos", "options")] + \
        [("reserved", C.c_int32 * 4)]


class f5_bigvgan_config(C.Structure):
    _fields_ = [("num_mels", C.c_int32), ("upsample_initial_channel", C.c_int32), ("num_upsamples", C.c_int32),
                ("upsample_rates", C.c_int32 * 8), ("upsample_kernel_sizes", C.c_int32 * 8), ("num_kernels", C.c_int32),
                ("resblock_kernel_sizes", C.c_int32 * 4), ("num_dilations", C.c_int32), ("resblock_dilations", C.c_int32 * 4),
                ("use_tanh_at_final", C.c_int32), ("use_bias_at_final", C.c_int32), ("precision", C.c_int32),
                ("reserved", C.c_int32 * 3)]


class f5_vocos_config(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("input_channels", "dim", "intermediate_dim", "num_layers", "n_fft",
                                         "hop_length")] + [("reserved", C.c_int32 * 4)]


_p = C.c_void_p
_i = C.c_int32
_f = C.c_float

# name -> (restype, argtypes); mirrors include/f5_hip.h exactly (tests/test_abi.py checks the export list)
SIGNATURES = {
    "f5_last_error": (C.c_char_p, []),
    "f5_version": (C.c_char_p, []),
    "f5_create": (_i, [C.POINTER(f5_config), C.POINTER(_p)]),
    "f5_destroy": (_i, [_p]),
    "f5_load_weight": (_i, [_p, C.c_char_p, _p, C.POINTER(C.c_int64), _i, _p]),
    "f5_finalize": (_i, [_p, _p]),
    "f5_text_embed": (_i, [_p, _p, _i, _i, C.POINTER(_i), _i, _i, _p, _p]),
    "f5_dit_forward": (_i, [_p, _p, _p, _p, _i, C.POINTER(_f), C.POINTER(_i), _i, _i, _i, _i, _i, _p, _p]),
    "f5_sample": (_i, [_p, _p, _i, _p, _p, _p, _i, C.POINTER(_f), _i, _f, C.POINTER(_i), _i, _i, _p, _p, _p]),
    "f5_reserve": (_i, [_p, _i, _i, _i]),
    "f5_vocos_create": (_i, [C.POINTER(f5_vocos_config), C.POINTER(_p)]),
    "f5_vocos_destroy": (_i, [_p]),
    "f5_vocos_load_weight": (_i, [_p, C.c_char_p, _p, C.POINTER(C.c_int64), _i, _p]),
    "f5_vocos_finalize": (_i, [_p, _p]),
    "f5_vocos_decode": (_i, [_p, _p, _i, _i, _p, _p]),
    "f5_vocos_decode_strided": (_i, [_p, _p, _i, _i, C.c_int64, C.c_int64, C.c_int64, _p, _p]),
    "f5_bigvgan_create": (_i, [C.POINTER(f5_bigvgan_config), C.POINTER(_p)]),
    "f5_bigvgan_destroy": (_i, [_p]),
    "f5_bigvgan_load_weight": (_i, [_p, C.c_char_p, _p, C.POINTER(C.c_int64), _i, _p]),
    "f5_bigvgan_finalize": (_i, [_p, _p]),
    "f5_bigvgan_forward": (_i, [_p, _p, _i, _i, C.c_int64, C.c_int64, C.c_int64, _p, _p]),
    "f5_mel_create": (_i, [_i, _i, _i, C.POINTER(_p)]),
    "f5_mel_destroy": (_i, [_p]),
    "f5_mel_load": (_i, [_p, C.c_char_p, _p, C.POINTER(C.c_int64), _i, _p]),
    "f5_mel_forward": (_i, [_p, _p, _i, _i, _p, _p]),
    "f5_mel_forward_ex": (_i, [_p, _p, _i, _i, _i, _f, _p, _p]),
    "f5k_gemm": (_i, [_i, _p, _p, _p, _i, _p, _i, _i, _i, _i, _i, _p]),
    "f5k_attention": (_i, [_i, _p, _p, _p, C.POINTER(_i), _p, _i, _i, _i, _p]),
    "f5k_convpos": (_i, [_i, _p, _p, _p, _p, C.POINTER(_i), _p, _i, _i, _i, _p]),
    "f5k_layernorm_mod": (_i, [_p, _p, _p, _p, _i, _i, _i, _f, _p]),
    "f5k_gemm_time": (_i, [_i, _i, _i, _i, _i, _i, _i, C.POINTER(_f), _p]),
    "f5_profile_enable": (_i, [_p, _i]),
    "f5_profile_read": (_i, [_p, C.POINTER(_f), C.POINTER(_i), C.POINTER(C.c_double), _i]),
}

_lib = None


def load() -> C.CDLL:
    """Loads the engine library; raises F5Error (never falls back) when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise F5Error(f"{LIB_PATH} is missing: build it with `python korean-f5-tts_amd/build.py` "
                      "(or __graft_entry__.build()); there is no CPU / PyTorch fallback for the F5-TTS hot path")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc: int, what: str = ""):
    if rc != 0:
        msg = load().f5_last_error().decode(errors="replace")
        raise F5Error(f"{what or 'libf5hip'} failed (code {rc}): {msg}")


def int_array(vals):
    if vals is None:
        return None
    return (C.c_int32 * len(vals))(*[int(v) for v in vals])


def float_array(vals):
    return (C.c_float * len(vals))(*[float(v) for v in vals])


def shape_array(shape):
    return (C.c_int64 * max(len(shape), 1))(*[int(s) for s in shape])
