"""Times the bf16 attention kernel (variants 3 = exact running max, 7 = lazy reference) at several sequence lengths under
rocprofv3 --kernel-trace --stats: per-launch time vs number of 64-key tiles separates the fixed cost from the per-tile cost."""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from f5_tts_amd import _lib
from gpu_util import k_attention
lib = _lib.load()
Bp, H = 2, 16
for N in (128, 256, 512, 1024, 2048):
    q, k, v = (torch.randn(Bp, H, N, 64, device="cuda:0") for _ in range(3))
    # blocks = Bp*H*N/128: keep the chip equally full by trading heads for length
    for var in ((7,) if len(sys.argv) < 2 else (3, 7)):
        lib.f5x_set_attn_variant(var)
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for _ in range(2):
            out = k_attention("bf16", q, k, v)
        torch.cuda.synchronize()
        print("N", N, "variant", var, "mean|out|", out.abs().mean().item(), flush=True)
