"""Times the GEMM shapes of one DiT Base step (M = 2B*N rows) for every tile config through f5k_gemm_time."""
import ctypes as C
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from f5_tts_amd import _lib

lib = _lib.load()
torch.cuda.init()
s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
M = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
shapes = [("qkv", M, 3072, 1024), ("out", M, 1024, 1024), ("ff1", M, 2048, 1024), ("ff2", M, 1024, 2048), ("inproj", M, 1024, 712)]
for prec, pname in ((1, "bf16"), (0, "f32")):
    for name, m, n, k in shapes:
        row = []
        for tm, tn in ((128, 128), (128, 64), (64, 64), (0, 0)):
            us = C.c_float()
            _lib.check(lib.f5k_gemm_time(prec, m, n, k, tm, tn, 50, C.byref(us), s))
            row.append(f"{tm}x{tn}: {us.value:7.1f}us {2.0*m*n*k/us.value/1e6:7.1f}TF")
        print(pname, name, (m, n, k), " | ".join(row), flush=True)
