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
def run(m, n, k, cfg, prec=1):
    A = torch.randn(m, k, device=dev); W = torch.randn(n, k, device=dev) / k ** 0.5; out = torch.zeros(m, n, device=dev)
    us = C.c_float(0)
    rc = fn(prec, A.data_ptr(), W.data_ptr(), None, 0, out.data_ptr(), m, n, k, cfg, 20, C.byref(us), s)
    return us.value
cfgs = [int(c) for c in sys.argv[1].split(",")] if len(sys.argv) > 1 else [2, 8, 9]
for cfg in cfgs:
    print("cfg", cfg)
    for k in (1024, 8192):
        us = run(2048, 2048, k, cfg); print(f"  K={k:5d}: {us:7.1f}us {2.0*2048*2048*k/us/1e6:7.1f}TF")
    for m in ((2048, 16384, 65536) if cfg < 100 else ()):
        us = run(m, 2048, 1024, cfg); print(f"  M={m:5d}: {us:7.1f}us {2.0*m*2048*1024/us/1e6:7.1f}TF")
