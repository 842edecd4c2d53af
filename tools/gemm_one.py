"""Runs one GEMM shape/config a few times (for rocprofv3 --pmc runs)."""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from f5_tts_amd import _lib
lib = _lib.load()
fn = lib.f5x_gemm2
fn.restype = C.c_int32
fn.argtypes = [C.c_int32] + [C.c_void_p] * 3 + [C.c_int32, C.c_void_p] + [C.c_int32] * 5 + [C.POINTER(C.c_float), C.c_void_p]
m, n, k, cfg = (int(x) for x in sys.argv[1:5])
prec = int(sys.argv[5]) if len(sys.argv) > 5 else 1
dev = "cuda:0"
A = torch.randn(m, k, device=dev); W = torch.randn(n, k, device=dev) / k ** 0.5; out = torch.zeros(m, n, device=dev)
us = C.c_float(0)
s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
rc = fn(prec, A.data_ptr(), W.data_ptr(), None, 0, out.data_ptr(), m, n, k, cfg, 10, C.byref(us), s)
print("rc", rc, "us", us.value, "TF", 2.0 * m * n * k / max(us.value, 1e-9) / 1e6)
