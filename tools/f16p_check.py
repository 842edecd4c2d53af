import sys, os, time, json
sys.path.insert(0, '/root/repo')
import torch
torch.set_num_threads(1)
import f5_tts_amd as P
dev = torch.device("cuda:0"); nv = P.config.VOCAB_SIZE + 1
N, ref, nfe = 1024, 256, 16
g = torch.Generator().manual_seed(1)
cond = torch.randn(1, ref, 100, generator=g).to(dev)
text = torch.randint(1, nv - 2, (1, round(0.15 * N)), generator=g)
kw = dict(steps=nfe, cfg_strength=2.0, sway_sampling_coef=-1.0, seed=0)
def build(prec):
    tr = P.DiT(**P.config.F5TTS_BASE, text_num_embeds=nv, mel_dim=100, precision=prec).init_synthetic(seed=0)
    return P.CFM(transformer=tr, mel_spec_module=P.mel.MelSpec()).to(dev)
def run(m):
    out, traj = m.sample(cond, text, N, **kw)
    for _ in range(2): m.sample(cond, text, N, **kw)
    torch.cuda.synchronize(); a = time.perf_counter()
    for _ in range(5): m.sample(cond, text, N, **kw)
    torch.cuda.synchronize()
    return out, traj, (time.perf_counter() - a) / 5 * 1e3
ro, rt, ms = run(build("f32"))
for prec in ("f16p", "f16", "f16x3", "bf16"):
    o, t, ms = run(build(prec))
    print(prec, "traj Linf %.3e mel Linf %.3e  sample %.2f ms" % (float((t - rt).abs().max()), float((o[:, ref:] - ro[:, ref:]).abs().max()), ms), flush=True)
