"""Which contraction classes of the F5_PREC_F16X3 engine need the three split-f16 products?  

For every subset listed below, the f16x3 engine is built with F5_X3_ABLATE=<mask>: the classes in the mask run with the plain
f16 product (hi x hi only: both operands rounded to f16, f32 accumulate, everything else -- activations, residual, norms,
softmax statistics -- as in f16x3), i.e. the CHEAPEST a class could ever be made.  Measured at C2 size (Base, B=1, 256 + 768
frames, NFE=16, cfg 2, sway -1): trajectory / generated-mel L-inf against the exact-f32 engine (itself pinned to the CPU oracle
at 6.6e-6 at this size, tests/test_configs_gpu.py).  north_star's bar is 1e-3; with a 2x margin a mix must stay below 5e-4.

    python tools/x3_ablate.py            (on the GPU box; ~1 minute)
    python tools/x3_ablate.py unett      the E2-TTS UNetT Base instead (B=2; no AdaLN gates: its blocks' own products dominate; bit 512 =
                                         the skip projections)
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

torch.set_num_threads(1)
import f5_tts_amd as P  # noqa: E402

UNETT = len(sys.argv) > 1 and sys.argv[1] == "unett"
CLASSES = {"qkv": 1, "attn_qk": 2, "attn_pv": 4, "out": 8, "ff1": 16, "ff2": 32, "in_proj": 64, "conv_pos": 128, "proj_out": 256}
if UNETT:
    CLASSES["skip_proj"] = 512
dev = torch.device("cuda:0")
nv = P.config.VOCAB_SIZE + 1
N, ref, nfe = 1024, 256, 16
BB = 2 if UNETT else 1
g = torch.Generator().manual_seed(1)
cond = torch.randn(BB, ref, 100, generator=g).to(dev)
text = torch.randint(1, nv - 2, (BB, round(0.15 * N)), generator=g)
kw = dict(steps=nfe, cfg_strength=2.0, sway_sampling_coef=-1.0, seed=0)


def build(prec, mask=0):
    os.environ["F5_X3_ABLATE"] = str(mask)
    os.environ["F5_X3_ATTN_SPLIT"] = "1"      # (the product default is plain-f16 attention products; here the mask decides)
    if UNETT:
        tr = P.UNetT(**P.config.E2TTS_BASE, text_num_embeds=nv, mel_dim=100, precision=prec).init_synthetic(seed=0)
    else:
        tr = P.DiT(**P.config.F5TTS_BASE, text_num_embeds=nv, mel_dim=100, precision=prec).init_synthetic(seed=0)
    m = P.CFM(transformer=tr, mel_spec_module=P.mel.MelSpec()).to(dev)
    tr.engine()          # f5_create reads the environment
    return m


def run(m, timed=True):
    out, traj = m.sample(cond, text, N, **kw)
    ms = None
    if timed:
        for _ in range(2):
            m.sample(cond, text, N, **kw)
        torch.cuda.synchronize()
        a = time.perf_counter()
        for _ in range(3):
            m.sample(cond, text, N, **kw)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - a) / 3 * 1e3
    return out, traj, ms


ref_out, ref_traj, ms32 = run(build("f32"))
rows = [{"mix": "f32 (reference)", "mask": None, "traj_linf": 0.0, "mel_linf": 0.0, "ms_sample": ms32}]
for prec in ("f16", "bf16"):
    o, t, ms = run(build(prec))
    rows.append({"mix": prec + " engine (16-bit activations too)", "mask": None, "traj_linf": float((t - ref_traj).abs().max()),
                 "mel_linf": float((o[:, ref:] - ref_out[:, ref:]).abs().max()), "ms_sample": ms})
masks = [("f16x3 (all classes split)", 0)]
masks += [("f16 products in: " + k, v) for k, v in CLASSES.items()]
if UNETT:
    masks += [("f16 products in: qkv + attention", 7), ("f16 products in: qkv + attention + ff1", 23), ("f16 products in: out + ff2 + skip (the residual writers)", 8 + 32 + 512)]
masks += [("f16 products in: attn_qk + attn_pv", 6), ("f16 products in: all four block GEMMs", 57), ("f16 products in: the blocks (GEMMs + attention)", 63),
          ("f16 products in: in_proj + conv_pos + proj_out (the I/O layers)", 448), ("f16 products in: blocks + in_proj", 63 + 64),
          ("f16 products in: blocks + conv_pos", 63 + 128), ("f16 products in: blocks + proj_out", 63 + 256), ("f16 products in: everything", 511)]
for name, mask in masks:
    o, t, ms = run(build("f16x3", mask))
    rows.append({"mix": name, "mask": mask, "traj_linf": float((t - ref_traj).abs().max()),
                 "mel_linf": float((o[:, ref:] - ref_out[:, ref:]).abs().max()),
                 "ms_sample": ms, "note": "time includes the diagnostic lo-zeroing passes; not a speed number" if mask else None})
    print(json.dumps(rows[-1]), flush=True)
print(json.dumps({"x3_ablation_c2": rows, "state_magnitude": float(ref_traj.abs().max())}))
