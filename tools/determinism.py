"""Bitwise run-to-run determinism of each kernel and of sample() (race detector)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
import f5_tts_amd as P
from gpu_util import k_attention, k_gemm, k_convpos, k_layernorm_mod
dev = "cuda:0"
g = torch.Generator().manual_seed(0)
def rep(name, fn, n=30):
    ref = fn()
    bad = 0
    worst = 0.0
    for _ in range(n):
        o = fn()
        if not torch.equal(o, ref):
            bad += 1
            worst = max(worst, (o - ref).abs().max().item())
    print(f"{name}: {bad}/{n} runs differ (max |diff| {worst:.3e})", flush=True)
q, k, v = (torch.randn(2, 16, 1024, 64, generator=g).to(dev) for _ in range(3))
rep("attention bf16 N=1024", lambda: k_attention("bf16", q, k, v))
rep("attention f32 N=1024", lambda: k_attention("f32", q, k, v), 5)
q2, k2, v2 = (torch.randn(2, 4, 777, 64, generator=g).to(dev) for _ in range(3))
rep("attention bf16 N=777", lambda: k_attention("bf16", q2, k2, v2))
for (m, n, kk) in ((2048, 3072, 1024), (2048, 1024, 2048), (2048, 2048, 1024), (2048, 1024, 1024)):
    A = torch.randn(m, kk, generator=g).to(dev); W = (torch.randn(n, kk, generator=g) / kk ** 0.5).to(dev)
    for t in ((0, 0), (-2, 0), (-8, 0), (-9, 0)):
        rep(f"gemm bf16 {m}x{n}x{kk} cfg{t}", lambda: k_gemm("bf16", A, W, None, tile=t), 20)
x = torch.randn(2, 1024, 1024, generator=g).to(dev); w = (torch.randn(1024, 64, 31, generator=g) * 0.02).to(dev); b = torch.randn(1024, generator=g).to(dev)
rep("convpos bf16", lambda: k_convpos("bf16", x, w, b, x), 10)
arch = P.config.F5TTS_BASE
tr = P.DiT(**arch, text_num_embeds=2546, mel_dim=100, precision="bf16").init_synthetic()
model = P.CFM(transformer=tr).to(dev)
cond = torch.randn(1, 256, 100, generator=g); text = torch.randint(0, 2545, (1, 150), generator=g)
def fwd():
    return tr(x=cond.new_zeros(1, 1024, 100).to(dev) + 0.1, cond=torch.zeros(1, 1024, 100, device=dev), text=text, time=torch.tensor(0.3), cfg_infer=True)
rep("dit forward bf16", fwd, 10)
rep("sample NFE=4", lambda: model.sample(cond, text, 1024, steps=4, cfg_strength=2.0, sway_sampling_coef=-1.0, seed=0)[0], 8)
