"""In-loop rate of the GEMM configs: long K amortises launch + epilogue; bf16 outputs as the engine's epilogues write.
usage: python tools/gemm3_rate.py cfgs [prec]"""
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
lib.f5x_set_out_bf16(1)
prec = int(sys.argv[2]) if len(sys.argv) > 2 else 1
def run(m, n, k, cfg):
    A = torch.randn(m, k, device=dev); W = torch.randn(n, k, device=dev) / k ** 0.5; out = torch.zeros(m, n, device=dev)
    us = C.c_float(0)
    rc = fn(prec, A.data_ptr(), W.data_ptr(), None, 0, out.data_ptr(), m, n, k, cfg, 20, C.byref(us), s)
    assert rc == 0, lib.f5_last_error()
    return us.value
cfgs = [int(c) for c in sys.argv[1].split(",")]
for cfg in cfgs:
    row = []
    for (m, n, k) in ((16384, 1024, 1024), (16384, 2048, 1024), (16384, 3072, 1024), (16384, 1024, 2048), (16384, 2048, 8192), (16384, 1024, 8192),
                      (4096, 1024, 8192), (65536, 1024, 1024), (2048, 1024, 1024), (2048, 3072, 1024), (2048, 2048, 1024), (2048, 1024, 2048)):
        us = run(m, n, k, cfg)
        row.append(f"{m}x{n}x{k}: {us:7.1f}us {2.0*m*n*k/us/1e6:6.0f}TF")
    print("cfg", cfg, " | ".join(row), flush=True)
