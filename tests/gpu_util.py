"""Helpers for the -m gpu tests: call the kernel-level C entry points on torch device tensors."""
import ctypes as C

import torch

from f5_tts_amd import _lib

DEV = "cuda:0"


def _p(t):
    return C.c_void_p(0 if t is None else t.data_ptr())


def _s():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def prec_id(name):
    return _lib.PRECISIONS[name]


def k_gemm(prec, A, W, bias=None, act=0, tile=(0, 0)):
    lib = _lib.load()
    M, K = A.shape
    N = W.shape[0]
    out = torch.empty(M, N, device=A.device, dtype=torch.float32)
    _lib.check(lib.f5k_gemm(prec_id(prec), _p(A), _p(W), _p(bias), act, _p(out), M, N, K, tile[0], tile[1], _s()), "f5k_gemm")
    return out


def k_attention(prec, q, k, v, lens=None):
    lib = _lib.load()
    Bp, H, N, _ = q.shape
    out = torch.empty(Bp, N, H * 64, device=q.device, dtype=torch.float32)
    _lib.check(lib.f5k_attention(prec_id(prec), _p(q), _p(k), _p(v), _lib.int_array(lens), _p(out), Bp, H, N, _s()),
               "f5k_attention")
    return out


def k_convpos(prec, x, w, bias, res=None, lens=None):
    lib = _lib.load()
    Bp, N, D = x.shape
    y = torch.empty_like(x)
    _lib.check(lib.f5k_convpos(prec_id(prec), _p(x), _p(w), _p(bias), _p(res), _lib.int_array(lens), _p(y), Bp, N, D, _s()),
               "f5k_convpos")
    return y


def k_layernorm_mod(x, scale, shift, rows_per_batch, eps=1e-6):
    lib = _lib.load()
    R, D = x.shape
    out = torch.empty_like(x)
    _lib.check(lib.f5k_layernorm_mod(_p(x), _p(scale), _p(shift), _p(out), R, D, rows_per_batch, eps, _s()), "f5k_layernorm_mod")
    return out
