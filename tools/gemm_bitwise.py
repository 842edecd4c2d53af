"""Every GEMM configuration must produce bit-identical results (same K order per output element)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from gpu_util import k_gemm
g = torch.Generator().manual_seed(0)
for (M, N, K) in ((8192, 1024, 1024), (8192, 2048, 1024), (2048, 1024, 2048), (700, 520, 1024)):
    A = torch.randn(M, K, generator=g).cuda(); W = (torch.randn(N, K, generator=g) / K ** 0.5).cuda(); b = torch.randn(N, generator=g).cuda()
    for prec in ("bf16", "f32"):
        ref = k_gemm(prec, A, W, b, tile=(-2, 0))
        for cfg in (8, 9, 10, 13, 20):
            out = k_gemm(prec, A, W, b, tile=(-cfg, 0))
            d = (out - ref).abs().max().item()
            print(prec, (M, N, K), "cfg", cfg, "vs cfg 2: max diff", d, "nonzero frac", (out != ref).float().mean().item())
