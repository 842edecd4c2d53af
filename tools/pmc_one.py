"""ONE eager sample() of a benchmark workload for a rocprofv3 --pmc pass (tools/pmc_pass.sh), with a stderr line after every phase.

    rocprofv3 --pmc FETCH_SIZE --output-format csv -d OUT -- python tools/pmc_one.py c2

Round 2's PMC passes ran `bench.py` under the counter service: ~30k dispatches plus the capture / instantiate / replay of a
~40k-node HIP graph, printing nothing until the last line -- three of those passes stalled until their time limit and the log
could not say where.  This script is the minimum that dispatches every per-block kernel of the workload: no HIP graph
(F5_HIP_GRAPH=0: the first call of a signature is eager anyway), no warm-up, no vocoder, ~2.7k dispatches for c2 and ~1.4k for
c3chunk, and it says on stderr where it is, so a stalled log names the phase.

Workloads (F5-TTS Base, f16p -- bench.py's precision -- unless F5_PMC_PREC is set, synthetic weights seed 0):
  c2       B=1, 256 + 768 frames, NFE=16, cfg 2                    (2,048 rows per backbone call: the latency-bound GEMM shapes)
  c3chunk  B=16 x 1024 frames, NFE=4, cfg 2  = ONE 32,768-row chunk (the many-row kernels: gemm_pp_kernel, attn2 at 16 x 2 x 16 heads)
"""
import os
import sys
import time

os.environ["F5_HIP_GRAPH"] = "0"
os.environ.setdefault("F5_TRACE", "1")
T0 = time.time()


def mark(msg):
    sys.stderr.write("[pmc_one %7.1fs] %s\n" % (time.time() - T0, msg))
    sys.stderr.flush()


mark("start")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

mark("import torch done")
torch.set_num_threads(1)
import f5_tts_amd as P  # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "c2"
prec = os.environ.get("F5_PMC_PREC", "f16p")
dev = torch.device("cuda:0")
torch.cuda.set_device(0)
torch.zeros(1, device=dev)
torch.cuda.synchronize()
mark("GPU initialised")
nv = P.config.VOCAB_SIZE + 1
tr = P.DiT(**P.config.F5TTS_BASE, text_num_embeds=nv, mel_dim=100, precision=prec).init_synthetic(seed=0)
model = P.CFM(transformer=tr, mel_spec_module=P.mel.MelSpec()).to(dev)
tr.engine()
torch.cuda.synchronize()
mark("engine created, weights uploaded and finalised")
g = torch.Generator().manual_seed(1)
if wl == "c2":
    B, N, ref, nfe = 1, 1024, 256, 16
elif wl == "c3chunk":
    B, N, ref, nfe = 16, 1024, 256, 4
else:
    raise SystemExit("workload must be c2 or c3chunk")
cond = torch.randn(B, ref, 100, generator=g).to(dev)
text = torch.randint(1, nv - 2, (B, round(0.15 * N)), generator=g)
dur = N if B == 1 else torch.full((B,), N, dtype=torch.long)
lens = None if B == 1 else torch.full((B,), ref, dtype=torch.long)
tr.engine().reserve(B, N, nfe)
torch.cuda.synchronize()
mark("arena reserved; launching ONE eager sample() (%s: B=%d N=%d NFE=%d %s)" % (wl, B, N, nfe, prec))
out, _ = model.sample(cond, text, dur, lens=lens, steps=nfe, cfg_strength=2.0, sway_sampling_coef=-1.0, seed=0)
mark("sample() enqueued")
torch.cuda.synchronize()
mark("sample() complete on the GPU")
assert torch.isfinite(out).all()
mark("done")
