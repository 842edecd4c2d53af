"""Tile sweep of the split-operand GEMM (F5_PREC_F16X3) at the C2 shapes: which configuration wins per projection."""
import ctypes as C
import os, sys, importlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
_lib = importlib.import_module("korean-f5-tts_amd._lib")
fn = _lib.load().f5k_gemm_time
fn.restype = C.c_int32
fn.argtypes = [C.c_int32] * 7 + [C.POINTER(C.c_float), C.c_void_p]
s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
for M in (2048, 4096):
    for name, n, k in (("qkv", 3072, 1024), ("out", 1024, 1024), ("ff1", 2048, 1024), ("ff2", 1024, 2048)):
        row = []
        for cfg in (0, 8, 9, 2, 10, 13):
            us = C.c_float(0)
            rc = fn(3, M, n, k, -cfg, 0, 30, C.byref(us), s)
            row.append(f"[{cfg if cfg else 'auto'}] {us.value:6.1f}" if rc == 0 else f"[{cfg}] ERR")
        print(f"M={M} {name}: " + "  ".join(row), flush=True)
