"""Split-operand GEMM (F5_PREC_F16X3, gemm2.h MODE 3): error against a float64 product next to the f32 and f16 kernels, every
tile configuration, then the rate at the shapes of C2 (M = 2048) and C3 (M = 16384).
usage: python tools/split_check.py"""
import ctypes as C
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from gpu_util import k_gemm
import importlib
_lib = importlib.import_module("korean-f5-tts_amd._lib")

g = torch.Generator().manual_seed(0)
for (M, N, K) in ((2048, 1024, 1024), (700, 520, 1024), (2048, 3072, 1024), (300, 100, 2048)):
    A = torch.randn(M, K, generator=g).cuda()
    A[:, :7] *= 300.0          # a few large columns (FF hidden outliers)
    W = (torch.randn(N, K, generator=g) / K ** 0.5).cuda()
    b = torch.randn(N, generator=g).cuda()
    ref = (A.double() @ W.double().T + b.double())
    sc = ref.abs().max().item()
    row = []
    for prec in ("f32", "f16x3", "f16"):
        for cfg in (8, 9, 2, 10, 13):
            out = k_gemm(prec, A, W, b, tile=(-cfg, 0))
            row.append(f"{prec}[{cfg}] {((out.double() - ref).abs().max().item() / sc):.1e}")
    print(f"{M}x{N}x{K}: " + "  ".join(row), flush=True)

fn = _lib.load().f5k_gemm_time
fn.restype = C.c_int32
fn.argtypes = [C.c_int32] * 7 + [C.POINTER(C.c_float), C.c_void_p]
s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
for M in (2048, 16384):
    for name, n, k in (("qkv", 3072, 1024), ("out", 1024, 1024), ("ff1", 2048, 1024), ("ff2", 1024, 2048)):
        row = []
        for prec, pn in ((0, "f32"), (3, "f16x3"), (2, "f16")):
            us = C.c_float(0)
            rc = fn(prec, M, n, k, 0, 0, 30, C.byref(us), s)
            row.append(f"{pn} {us.value:7.1f} us ({2.0 * M * n * k / us.value / 1e6:6.1f} TF/s)" if rc == 0 else f"{pn} ERR")
        print(f"M={M} {name}: " + "   ".join(row), flush=True)
