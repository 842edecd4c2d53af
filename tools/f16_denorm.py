"""Does v_mfma_f32_16x16x32_f16 keep f16 SUBNORMAL inputs?  (The split-operand f32 GEMM stores the low halves of small values
as f16 subnormals.)  A = 2^-20 (f16 subnormal), W = 2^10, K = 64: the exact result is 64 * 2^-10 = 0.0625; 0 if flushed."""
import ctypes as C
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from f5_tts_amd import _lib

lib = _lib.load()
fn = lib.f5k_gemm
fn.restype = C.c_int32
fn.argtypes = [C.c_int32] + [C.c_void_p] * 3 + [C.c_int32, C.c_void_p] + [C.c_int32] * 5 + [C.c_void_p]
s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
for e in (-15, -20, -24):
    A = torch.full((64, 64), 2.0 ** e, device="cuda:0")
    W = torch.full((64, 64), 2.0 ** 10, device="cuda:0")
    out = torch.zeros(64, 64, device="cuda:0")
    rc = fn(2, A.data_ptr(), W.data_ptr(), None, 0, out.data_ptr(), 64, 64, 64, 0, 0, s)
    torch.cuda.synchronize()
    print(f"A = 2^{e}: rc {rc}, out[0,0] = {out[0, 0].item():.6g}, exact {64 * 2.0 ** (e + 10):.6g}")
