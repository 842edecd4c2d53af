import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from f5_tts_amd import _lib
lib = _lib.load()
fn = lib.f5x_gemm2
fn.restype = C.c_int32
fn.argtypes = [C.c_int32] + [C.c_void_p] * 3 + [C.c_int32, C.c_void_p] + [C.c_int32] * 5 + [C.POINTER(C.c_float), C.c_void_p]
dev = "cuda:0"
s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
lib.f5x_set_cold_weights(1)
for name, m, n, k, cfg in (("ff1", 2048, 2048, 1024, 2), ("qkv", 2048, 3072, 1024, 2), ("out", 2048, 1024, 1024, 9)):
    A = torch.randn(m, k, device=dev); W = torch.randn(n, k, device=dev) / k ** 0.5; b = torch.randn(n, device=dev)
    out = torch.zeros(m, n, device=dev)
    for obf in (0, 1):
        lib.f5x_set_out_bf16(obf)
        for act in (0, 1):
            for bias in (None, b):
                us = C.c_float(0)
                fn(1, A.data_ptr(), W.data_ptr(), bias.data_ptr() if bias is not None else None, act, out.data_ptr(), m, n, k, cfg, 64, C.byref(us), s)
                print(f"{name} cfg{cfg} out={'bf16' if obf else 'f32'} act={act} bias={bias is not None}: {us.value:6.1f}us", flush=True)
