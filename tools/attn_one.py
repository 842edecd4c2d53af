"""Runs the attention kernel a few times through f5k_attention (for rocprofv3 runs) and prints timing."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from gpu_util import k_attention
Bp, H, N = (int(x) for x in sys.argv[1:4])
prec = sys.argv[4] if len(sys.argv) > 4 else "bf16"
q, k, v = (torch.randn(Bp, H, N, 64, device="cuda:0") for _ in range(3))
for _ in range(5):
    out = k_attention(prec, q, k, v)
torch.cuda.synchronize()
print("ok", out.float().abs().mean().item())
