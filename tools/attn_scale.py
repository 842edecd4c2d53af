"""Attention kernel time at N = 1024 as the number of workgroups grows (Bp*H*8): separates per-CU cost from chip-wide
contention.  Run under rocprofv3 --kernel-trace and read the attn2 rows of the trace."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from gpu_util import k_attention
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
for bh in (4, 8, 16, 32, 64, 128):
    q, k, v = (torch.randn(1, bh, N, 64, device="cuda:0") for _ in range(3))
    for _ in range(2):
        out = k_attention("bf16", q, k, v)
    torch.cuda.synchronize()
    print("bh", bh, out.abs().mean().item(), flush=True)
