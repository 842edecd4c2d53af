import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from f5_tts_amd import _lib
lib = _lib.load()
torch.cuda.init()
s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
names = ["LN alone", "GEMM alone", "LN->GEMM (data edge)", "LN'->GEMM (no edge)", "GEMM a -> GEMM b", "GEMM a -> GEMM a", "GEMM + QKV epilogue alone", "GEMM + bias store alone"]
shapes = [(2048, 2048, 1024, 2), (2048, 3072, 1024, 2)] if len(sys.argv) < 2 else [(2048, 3072, 1024, int(c)) for c in sys.argv[1].split(',')]
for (m, n, k, cfg) in shapes:
    r = (C.c_float * 8)()
    rc = lib.f5x_pair_time(m, n, k, cfg, 200, r, s)
    print(f"{m}x{n}x{k} cfg{cfg}: " + "; ".join(f"{nm} {r[i]:.1f}us" for i, nm in enumerate(names)), flush=True)
