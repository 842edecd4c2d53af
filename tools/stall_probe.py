"""Where do the occasional 50-150 ms sample() calls come from?  Times host preparation, the enqueue (f5_sample) and the
wait separately over many calls and prints the outliers."""
import sys, os, time, gc
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import f5_tts_amd as P
dev = torch.device("cuda:0")
nv = P.config.VOCAB_SIZE + 1
tr = P.DiT(**P.config.F5TTS_BASE, text_num_embeds=nv, mel_dim=100, precision="bf16").init_synthetic(seed=0)
model = P.CFM(transformer=tr, mel_spec_module=P.mel.MelSpec()).to(dev)
g = torch.Generator().manual_seed(1)
cond = torch.randn(1, 256, 100, generator=g).to(dev)
text = torch.randint(1, nv - 2, (1, 150), generator=g)
eng = tr.engine()
orig = eng.sample
rec = {}
def timed_sample(*a, **k):
    t0 = time.perf_counter()
    r = orig(*a, **k)
    rec["enq"] = time.perf_counter() - t0
    return r
eng.sample = timed_sample
kw = dict(steps=16, cfg_strength=2.0, sway_sampling_coef=-1.0, seed=0)
for _ in range(4):
    model.sample(cond, text, 1024, **kw)
torch.cuda.synchronize()
mode = sys.argv[1] if len(sys.argv) > 1 else "plain"
if mode == "nogc":
    gc.disable()
if mode == "t1":
    torch.set_num_threads(1)
rows = []
for i in range(80):
    t0 = time.perf_counter()
    out, _ = model.sample(cond, text, 1024, **kw)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    rows.append(((t1 - t0) * 1e3, rec["enq"] * 1e3, (t2 - t1) * 1e3))
tot = sorted(r[0] + r[2] for r in rows)
print(mode, "median total %.1f ms, p90 %.1f, max %.1f" % (tot[len(tot) // 2], tot[int(len(tot) * 0.9)], tot[-1]))
for i, (host, enq, wait) in enumerate(rows):
    if host + wait > 45:
        print("  call %d: host %.1f ms (of which f5_sample enqueue %.1f), wait %.1f" % (i, host, enq, wait))
