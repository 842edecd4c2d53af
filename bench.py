#!/usr/bin/env python3
"""bench.py -- RTF of the F5-TTS hot path (CFM.sample -> DiT x NFE -> Vocos) on N MI355X GPUs of one node.

    python bench.py --gpus 1 --steps 10 --warmup 3
    python bench.py --gpus N ...        (N > 1 outside torchrun: starts its own N ranks as a child torch.distributed.run)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One "step" = one utterance through the whole path on every rank: sample() (text encoder, time/AdaLN precompute,
NFE Euler steps with CFG over the DiT backbone) + Vocos decode of the generated frames, and for N > 1 one RCCL
all_gather of the generated mel.  Workload (BASELINE.json configs[1], "C2"): F5-TTS Base, 16-bit MFMA operands
(default precision f16p: f16 operands in the 22 blocks, split-f16 input / output layers -- within north_star's 1e-3 of the CPU path,
which bf16, BASELINE's wording, misses 17-fold; bf16 is timed beside it in `precisions`), batch 1, prompt 256 frames, total 1024 frames (768 generated = 8.192 s of 24 kHz audio), NFE=16 (EPSS grid),
cfg_strength 2.0, sway -1; synthetic random-init weights (seed 0) and synthetic inputs, all resident in HBM before the
timed region.  value = generated audio seconds of ALL ranks / wall seconds (the metric as BASELINE.json words it:
audio_sec / wall_sec, higher is better); rtf_wall_over_audio (the reference's own convention, benchmark.py:457) is
its inverse.  `value` scales weakly: every rank synthesises its own utterance per step.

Beside it, every line carries
  precisions  the same C2 step timed at every operand precision (f32 = exact-f32 MFMA; f16x3; f16p; f16; bf16) with
              the generated-mel L-inf of that precision against the f32 engine on the same inputs (the f32 engine itself
              is pinned against the CPU oracle at this size by tests/test_configs_gpu.py): each speed number sits with
              its own accuracy;
  parity_precision / value_at_parity   the fastest precision of that record that stays within north_star's 1e-3, and its rate;
  c3          BASELINE.json configs[2]: B=32 variable-length utterances padded to 1024 frames, NFE=32, both attn_mask_enabled values;
  c4          BASELINE.json configs[3]: the 256-utterance synthetic set (lengths seed 1234, NFE=16)
              sharded over the N ranks by dist.dp_sample (contiguous slices of the length-sorted list with equal padded cost,
              frame-budget batches, ONE all_gather of the generated mel): wall of the whole job, audio-s per wall-s -- STRONG
              scaling (total work fixed as N grows; north_star's ">= 6x at 8 GPUs" is c4.value at N=8 over c4.value at N=1);
              at N=1 also shard8_wall_sec / predicted_scaling_8: the 8 shards of an 8-rank run timed one by one on this GPU.
"""
from __future__ import annotations

import argparse
import glob
import json
import os
import statistics
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402

MFMA_PEAK_TFLOPS = {"bf16": 2500.0, "f16": 2500.0, "f16p": 2500.0, "f32": 157.3, "f16x3": 2500.0 / 3}   # (f16x3: three f16 MFMAs per product)  # dense, /opt/skills/guides/MI355X_MICROARCH.md
TRAFFIC_FILE = os.path.join("profiles", "r03_pmc_traffic.json")


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--precision", default="parity", choices=["parity", "f16p", "bf16", "f16", "f16x3", "f32"],
                    help="parity (default) = the fastest precision inside north_star's 1e-3 for the workload's backbone: f16p for the DiT "
                         "workloads (c2 / c3), f16x3 for c5's UNetT.  f16p: f16 MFMA operands in the 22 blocks, split-f16 (f32-level) input / output layers -- the fastest "
                         "precision that meets north_star's 1e-3 mel L-inf (1.6e-4 at this size); bf16: BASELINE's wording of C2, 17x over that bar")
    ap.add_argument("--nfe", type=int, default=16)
    ap.add_argument("--frames", type=int, default=1024)
    ap.add_argument("--ref-frames", type=int, default=256)
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--workload", default="c2", choices=["c2", "c3", "c5"],
                    help="c2: B=1 N=1024 NFE=16 (default, the metric's config); c3: B=32 variable-length padded NFE=32; "
                         "c5: E2-TTS UNetT B=8 NFE=16")
    ap.add_argument("--attn-mask", action="store_true", help="c3: run with attn_mask_enabled=True (padded keys masked)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profile", action="store_true")
    ap.add_argument("--vocoder-precision", default="f16x3", choices=["f32", "f16x3"],
                    help="c5: BigVGAN GEMM precision (f16x3 = split-f16 products, f32-level results)")
    ap.add_argument("--no-precisions", action="store_true", help="skip the per-precision record")
    ap.add_argument("--no-c4", action="store_true", help="skip the 256-utterance data-parallel job")
    ap.add_argument("--no-c3", action="store_true", help="skip the c3 record (B=32 variable-length, NFE=32) of the default c2 run")
    ap.add_argument("--no-shard8", action="store_true",
                    help="c4 at N=1: skip timing the 8 shards of partition(durs, 8) one by one (predicted_scaling_8)")
    ap.add_argument("--c4-utts", type=int, default=256)
    ap.add_argument("--c4-warm-passes", type=int, default=2,
                    help="untimed passes of the c4 job before the timed one (first: eager, second: HIP-graph capture)")
    ap.add_argument("--setup-runs", type=int, default=3, help="untimed engine-initialisation runs before the warm-up")
    ap.add_argument("--rehearse-one-gpu", action="store_true",
                    help="multi-rank rehearsal on a ONE-GPU box: every rank uses cuda:0 and the collectives run on gloo "
                         "through host copies (RCCL refuses two ranks on one device); never used for reported numbers")
    return ap.parse_args()


def dit_flops_per_seq_forward(N, D=1024, depth=22, F=2048, Dt=512, mel=100):
    """algorithmic FLOPs of one sequence-forward of N tokens through DiT Base."""
    per_block = 2 * N * D * 3 * D + 2 * N * D * D + 4 * N * N * D + 2 * 2 * N * D * F
    embed = 2 * N * (2 * mel + Dt) * D + 2 * (2 * N * D * (D // 16) * 31) + 2 * N * D * mel
    return depth * per_block + embed


def make_inputs(P, args, rank, workload=None, B=None):
    """SURVEY.md section 8(d) synthetic inputs.  Returns cond [B, ref_max, 100], text [B, nt], durations, ref lens."""
    g = torch.Generator().manual_seed(1 + rank)
    workload = workload or args.workload
    B = B or args.batch
    if workload == "c3":
        gl = torch.Generator().manual_seed(1234 + rank)
        durs = [args.frames] + [int(x) for x in torch.randint(384, args.frames + 1, (B - 1,), generator=gl)]
    else:
        durs = [args.frames] * B
    refs = [d // 4 for d in durs] if workload == "c3" else [args.ref_frames] * B
    cond = torch.zeros(B, max(refs), 100)
    for i, r in enumerate(refs):
        cond[i, :r] = torch.randn(r, 100, generator=g)
    nts = [round(0.15 * d) for d in durs]
    text = torch.full((B, max(nts)), -1, dtype=torch.long)
    for i, n in enumerate(nts):
        text[i, :n] = torch.randint(1, P.config.VOCAB_SIZE - 1, (n,), generator=g)
    return cond, text, durs, refs


def make_c4_job(P, n_utts):
    """BASELINE.json configs[3] / SURVEY.md 8(d): n_utts utterances, N_i ~ U{384..1024} (seed 1234; the first is 1024),
    prompt = N_i // 4 frames of N(0,1) (seed 1), text = round(0.15 N_i) ids.  Identical on every rank."""
    gl = torch.Generator().manual_seed(1234)
    durs = [1024] + [int(x) for x in torch.randint(384, 1025, (n_utts - 1,), generator=gl)]
    g = torch.Generator().manual_seed(1)
    conds = [torch.randn(d // 4, 100, generator=g) for d in durs]
    texts = [torch.randint(1, P.config.VOCAB_SIZE - 1, (round(0.15 * d),), generator=g) for d in durs]
    return conds, texts, durs


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(P, args, sd, vsd, cond, text):
    """The CPU oracle (a port: the reference itself cannot travel to the GPU box) on a bounded sample of the same
    workload (1 warm-up + 3 timed runs, median): each run times a 1-step and a 2-step sample() of the C2
    utterance -- their difference is one Euler step (2 DiT forwards with CFG; the per-step cost does not depend on
    the step), the 1-step run minus one step is the per-utterance fixed part (text encoder, time MLP) -- and the Vocos
    decode of the generated frames; scaled to NFE steps."""
    from oracle import f5_oracle as O

    arch = P.config.F5TTS_BASE
    # a one-GPU box owns a 16-core share of the host (more threads than that only oversubscribe it)
    cores = int(os.environ.get("F5_CPU_THREADS", min(16, len(os.sched_getaffinity(0)))))
    torch.set_num_threads(cores)
    N = args.frames
    gen = N - args.ref_frames
    kw = dict(cfg_strength=2.0, sway_sampling_coef=-1.0, seed=0, use_epss=False)
    walls, steps_s, fixed_s, voc_s = [], [], [], []
    with torch.no_grad():
        out, _ = O.sample(sd, arch, cond[:1], text[:1], N, steps=1, **kw)       # warm-up (page-in, thread pool, allocator)
        for _ in range(3):
            t0 = time.perf_counter()
            O.sample(sd, arch, cond[:1], text[:1], N, steps=1, **kw)
            t1 = time.perf_counter()
            out, _ = O.sample(sd, arch, cond[:1], text[:1], N, steps=2, **kw)
            t2 = time.perf_counter()
            O.vocos_decode(vsd, out[:, args.ref_frames:].permute(0, 2, 1))
            t3 = time.perf_counter()
            per_step = (t2 - t1) - (t1 - t0)
            fixed = max((t1 - t0) - per_step, 0.0)
            steps_s.append(per_step)
            fixed_s.append(fixed)
            voc_s.append(t3 - t2)
            walls.append(fixed + per_step * args.nfe + (t3 - t2))
    wall = statistics.median(walls)
    audio = gen * 256 / 24000
    return {"value": audio / wall, "unit": "audio_sec/wall_sec", "cores": cores, "kind": "port", "cpu_model": cpu_model(),
            "wall_sec_scaled": wall, "wall_sec_scaled_runs": walls, "rtf_wall_over_audio": wall / audio,
            "sample": f"CPU oracle (PyTorch fp32, {cores} threads) on the C2 utterance, 1 warm-up + 3 timed runs (median): per run a "
                      f"1-step and a 2-step sample() (difference = one Euler step with CFG: {statistics.median(steps_s):.2f} s, "
                      f"scaled to NFE={args.nfe}; per-utterance fixed part {statistics.median(fixed_s):.2f} s) and the Vocos "
                      f"decode of {gen} frames ({statistics.median(voc_s):.2f} s)"}


class ClockWatch:
    """Shader clock and socket power of THIS rank's GPU while a region runs, read from the amdgpu hwmon files (sysfs, read-only, no
    child process): `with ClockWatch(dev) as cw: ...; cw.summary()`.  Informational -- the MI355X lowers its clock under sustained
    matrix load, so a roofline fraction priced at the nominal 2.4 GHz understates what the kernels do per cycle.  Returns None
    where the files are missing or the card cannot be matched by PCI address."""

    def __init__(self, dev_index, period=0.02):
        self.period, self.samples, self.dir, self.cap = period, [], None, None
        self._stop = threading.Event()
        self._thr = None
        try:
            pr = torch.cuda.get_device_properties(dev_index)
            bdf = "%04x:%02x:%02x.0" % (getattr(pr, "pci_domain_id", 0), pr.pci_bus_id, pr.pci_device_id)
            for c in sorted(glob.glob("/sys/class/drm/card*/device")):
                if os.path.realpath(c).endswith(bdf):
                    hw = sorted(glob.glob(os.path.join(c, "hwmon", "hwmon*")))
                    if hw and os.path.exists(os.path.join(hw[0], "freq1_input")):
                        self.dir = hw[0]
            if self.dir and os.path.exists(os.path.join(self.dir, "power1_cap")):
                self.cap = int(open(os.path.join(self.dir, "power1_cap")).read()) / 1e6
        except Exception:
            self.dir = None

    def _read(self, name):
        try:
            return int(open(os.path.join(self.dir, name)).read())
        except Exception:
            return None

    def _run(self):
        while not self._stop.is_set():
            f, pw = self._read("freq1_input"), self._read("power1_input")
            if f is not None:
                self.samples.append((f / 1e6, pw / 1e6 if pw is not None else None))
            self._stop.wait(self.period)

    def __enter__(self):
        if self.dir:
            self._thr = threading.Thread(target=self._run, daemon=True)
            self._thr.start()
        return self

    def __exit__(self, *a):
        self._stop.set()
        if self._thr:
            self._thr.join()
        return False

    def summary(self):
        if not self.samples:
            return None
        fs = [a for a, _ in self.samples]
        ps = [b for _, b in self.samples if b is not None]
        return {"sclk_mhz_avg": round(sum(fs) / len(fs), 1), "sclk_mhz_min": round(min(fs), 1), "sclk_mhz_nominal": 2400,
                "power_w_avg": round(sum(ps) / len(ps), 1) if ps else None, "power_cap_w": self.cap, "samples": len(fs),
                "source": "amdgpu hwmon freq1_input / power1_input, sampled every %d ms during the timed region" % int(self.period * 1e3)}


def free_port() -> int:
    import socket
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        return so.getsockname()[1]


def self_launch(args) -> int:
    """`python bench.py --gpus N` (N > 1) outside torchrun: start the N ranks as a CHILD torch.distributed.run (one process per
    GPU over RCCL, the way the reference's harness is started: runtime/triton_trtllm/run.sh:81,103, eval_infer_batch.py:28) and
    return its status.  Runs before anything in this process has touched the GPU (device_count() does not initialise it)."""
    import subprocess
    have = torch.cuda.device_count()
    if not args.rehearse_one_gpu and args.gpus > have:
        print(f"bench.py: --gpus {args.gpus} but only {have} GPU(s) are visible", file=sys.stderr)
        return 2
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__), *sys.argv[1:]]
    print("bench.py: launching " + " ".join(cmd), file=sys.stderr)
    return subprocess.call(cmd, env=env)


def main():
    args = parse()
    if args.precision == "parity":
        args.precision = "f16x3" if args.workload == "c5" else "f16p"
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(args))
    if args.workload == "c3":
        args.batch, args.nfe = (32 if args.batch == 1 else args.batch), (32 if args.nfe == 16 else args.nfe)
    if args.workload == "c5":
        args.batch = 8 if args.batch == 1 else args.batch
    # The GPU path has no parallel host work, and torch's default intra-op pool is one thread per LOGICAL CPU of the
    # host (256 on the GPU boxes, of which a job owns a 16-CPU share; x8 ranks on a node): every tiny CPU op of a step
    # (noise draw, pads) then wakes an oversubscribed OpenMP team whose spinning starves the HIP submission thread.
    # Measured on a loaded box (tools/stall_probe.py): p90 / max per utterance 99.9 / 198.7 ms with the default pool,
    # 33.9 / 37.3 ms with one thread (median 33-36 ms either way).  cpu_baseline() sets its own thread count later.
    P_threads = int(os.environ.get("F5_HOST_THREADS", "1"))
    torch.set_num_threads(P_threads)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not args.rehearse_one_gpu and local_rank >= torch.cuda.device_count():
        raise SystemExit(f"LOCAL_RANK {local_rank} but only {torch.cuda.device_count()} GPU(s) are visible")
    if args.rehearse_one_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device(f"cuda:{local_rank}")
    if world > 1:
        import torch.distributed as dist
        if args.rehearse_one_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    import f5_tts_amd as P

    nv = P.config.VOCAB_SIZE + 1  # load_model: text_num_embeds = vocab_size + 1 (utils_infer.py:313-317)

    def build_model(precision, attn_mask=None):
        if args.workload == "c5":
            tr_ = P.UNetT(**P.config.E2TTS_BASE, text_num_embeds=nv, mel_dim=100, precision=precision).init_synthetic(seed=0)
        else:
            arch = dict(P.config.F5TTS_BASE, attn_mask_enabled=bool(args.attn_mask if attn_mask is None else attn_mask))
            tr_ = P.DiT(**arch, text_num_embeds=nv, mel_dim=100, precision=precision).init_synthetic(seed=0)
        return tr_, P.CFM(transformer=tr_, mel_spec_module=P.mel.MelSpec(mel_spec_type="bigvgan" if args.workload == "c5" else "vocos")).to(dev)

    tr, model = build_model(args.precision)
    if args.workload == "c5":   # BASELINE config 5: E2-TTS + BigVGAN (the reference calls it as vocoder(mel), utils_infer.py:705)
        _bv = P.BigVGAN(P.config.BIGVGAN_V2_24K, precision=args.vocoder_precision).init_synthetic(seed=1).to(dev)
        voc = type("BigVGANDecode", (), {"decode": staticmethod(lambda mel: _bv(mel)), "state_dict": _bv.state_dict})()
    else:
        voc = P.Vocos(P.config.VOCOS_24K).init_synthetic(seed=1).to(dev)
    cond_cpu, text_cpu, durs, refs = make_inputs(P, args, rank)
    # the prompt mel is resident in HBM; the text ids stay on the host (the reference's API takes list[str]: their
    # length feeds host-side duration arithmetic, cfm.py:125-141, and a device copy would force a D2H sync per call)
    cond, text = cond_cpu.to(dev), text_cpu
    eng = tr.engine()
    eng.reserve(args.batch, args.frames, args.nfe)
    N, ref, B = args.frames, args.ref_frames, args.batch
    gen = N - ref
    gen_frames_total = sum(d - r for d, r in zip(durs, refs))
    kw = dict(steps=args.nfe, cfg_strength=2.0, sway_sampling_coef=-1.0, seed=0)
    uniform = len(set(durs)) == 1 and len(set(refs)) == 1
    dur_t = N if uniform else torch.tensor(durs)
    lens_t = None if uniform else torch.tensor(refs)
    gather_buf = None
    if world > 1:
        gather_buf = torch.empty(world * B, N, 100, device=dev)

    def step(m=None):
        out, _traj = (m or model).sample(cond, text, dur_t, lens=lens_t, **kw)
        if uniform:
            wav = voc.decode(out[:, ref:, :].permute(0, 2, 1))
        else:  # per item, as the reference's harness does (eval_infer_batch.py:202-206): own prompt / total length
            wav = [voc.decode(out[i:i + 1, refs[i]:durs[i], :].permute(0, 2, 1)) for i in range(B)][-1]
        if world > 1:
            if args.rehearse_one_gpu:   # gloo has no device collectives: same call shape on host copies
                host = torch.empty(gather_buf.shape, dtype=gather_buf.dtype)
                dist.all_gather_into_tensor(host, out.contiguous().cpu())
                gather_buf.copy_(host)
            else:
                dist.all_gather_into_tensor(gather_buf, out)
        return out, wav

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(x):
        if world == 1:
            return x
        tt = torch.tensor([x], device="cpu" if args.rehearse_one_gpu else dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        return float(tt.item())

    # engine setup (untimed, before the W warm-up steps): first-use work that is not part of a step -- arena growth,
    # hipFuncSetAttribute on every kernel instantiation, RCCL communicator creation, clock ramp from the idle state
    for _ in range(args.setup_runs):
        step()
    barrier()
    for _ in range(args.warmup):
        out, wav = step()
    barrier()
    with ClockWatch(dev.index) as cw:
        t0 = time.perf_counter()
        for _ in range(args.steps):
            out, wav = step()
        barrier()
        elapsed = max_over_ranks(time.perf_counter() - t0)
    assert torch.isfinite(out).all() and torch.isfinite(wav).all()

    audio_per_step = gen_frames_total * 256 / 24000
    value = audio_per_step * args.steps * world / elapsed
    ms_per_step = elapsed / args.steps * 1e3

    result = {
        "metric": "RTF (audio_sec/wall_sec) F5-TTS Base NFE=%d batch=%d" % (args.nfe, B),
        "value": value, "unit": "audio_sec/wall_sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": {"f16p": "f16"}.get(args.precision, args.precision), "precision_mode": args.precision, "data": "synthetic",
        "rtf_wall_over_audio": 1.0 / value * world,  # per-utterance RTF in the reference's convention (wall / audio)
        "config": {"workload": {"c2": "C2: F5-TTS Base, 1 utterance/rank/step, prompt %d + generated %d mel frames, NFE=%d EPSS, "
                                      "cfg 2.0, sway -1, Vocos decode, mel all_gather when n_gpus>1" % (ref, gen, args.nfe),
                                "c3": "C3: F5-TTS Base, %d variable-length utterances/rank/step padded to %d frames (prompt = len/4), "
                                      "NFE=%d, cfg 2.0, sway -1, attn_mask_enabled=%s, per-item Vocos decode" % (B, N, args.nfe, bool(args.attn_mask)),
                                "c5": "C5: E2-TTS UNetT Base, %d utterances/rank/step, prompt %d + generated %d frames, NFE=%d, cfg 2.0, "
                                      "sway -1, BigVGAN v2 decode (24 kHz, 100 band, 256x; %s; parity unpinned)" % (B, ref, gen, args.nfe, args.vocoder_precision)}[args.workload],
                   "global_batch": B * world, "frames": N, "generated_audio_sec_per_step": audio_per_step * world,
                   "parallelism": "dp%d" % world, "weights": "synthetic random-init seed 0"},
    }
    if args.workload == "c5":
        result["parity_precision"] = "f16x3"   # UNetT has no AdaLN gates: its block products dominate the error
        result["parity_note"] = ("E2-TTS UNetT meets north_star's 1e-3 only with split-f16 products in every block (--precision f16x3: "
                                 "~610-640 ms per step on one MI355X, profiles/r03_o_bench_c5_f16x3.json; f16p 5.0e-3, bf16 3e-2 at Base size)")
    result["clocks"] = cw.summary()   # rank 0's GPU during the timed region (None where sysfs does not offer it)
    if world > 1:
        result["rccl_ranks"] = dist.get_world_size()
        result["collective_backend"] = dist.get_backend()
        try:
            result["rccl_version"] = ".".join(str(v) for v in torch.cuda.nccl.version())
        except Exception as ex:
            result["rccl_version"] = repr(ex)

    # ---- diagnostic split of one step (after the timed region, not part of the metric).  The host side of a step is
    # measured UNTHROTTLED: the engine's pinned staging ring holds 8 calls, so a host that is more than 8 utterances ahead
    # of the GPU waits there (back-pressure, an idle wait -- in a long timed region enqueue time per step therefore tends
    # to the GPU time per step and says nothing about host cost).  Here: 4 steps enqueued from an idle stream.
    phases = {}
    torch.cuda.synchronize()
    rec = {}
    orig_sample, orig_decode = eng.sample, voc.decode

    def t_sample(*a, **k):
        a0 = time.perf_counter()
        r = orig_sample(*a, **k)
        rec["engine_sample_call"] = rec.get("engine_sample_call", 0.0) + time.perf_counter() - a0
        return r

    def t_decode(*a, **k):
        a0 = time.perf_counter()
        r = orig_decode(*a, **k)
        rec["vocos_decode_call"] = rec.get("vocos_decode_call", 0.0) + time.perf_counter() - a0
        return r

    eng.sample, voc.decode = t_sample, t_decode
    # (C5: the BigVGAN decode of one step is ~2,700 launches; several steps overrun the stream's command queue and the
    #  host blocks on it -- again back-pressure, not host cost -- so one step is enqueued there)
    nrep = 1 if args.workload == "c5" else 4
    a0 = time.perf_counter()
    for _ in range(nrep):
        step()
    a1 = time.perf_counter()
    torch.cuda.synchronize()
    eng.sample, voc.decode = orig_sample, orig_decode
    phases["host_enqueue_ms_per_step"] = (a1 - a0) / nrep * 1e3
    phases["host_ms_per_step_in"] = {k: v / nrep * 1e3 for k, v in rec.items()}   # rest = sample() argument handling + noise draw
    ps, pv = [], []
    for _ in range(3):
        torch.cuda.synchronize()
        a = time.perf_counter()
        o_, _ = model.sample(cond, text, dur_t, lens=lens_t, **kw)
        torch.cuda.synchronize()
        b = time.perf_counter()
        if uniform:
            voc.decode(o_[:, ref:, :].permute(0, 2, 1))
        torch.cuda.synchronize()
        c = time.perf_counter()
        ps.append((b - a) * 1e3)
        pv.append((c - b) * 1e3)
    phases["sample_ms_min_max"] = [min(ps), max(ps)]
    phases["vocos_ms_min_max"] = [min(pv), max(pv)]
    result["phases"] = phases

    if rank == 0:
        # ---- roofline of the dominant kernel class (MFMA GEMMs), per-launch HIP events on the launch stream, in a
        # dedicated pass over the same workload right after the timed region
        if not args.no_profile:
            eng.profile(True)
            model.sample(cond, text, dur_t, lens=lens_t, **kw)
            torch.cuda.synchronize()
            prof = eng.profile_read()
            eng.profile(False)
            gm = prof["gemm"]
            peak = MFMA_PEAK_TFLOPS[args.precision]
            ach = gm["flops"] / (gm["ms"] * 1e-3) / 1e12 if gm["ms"] > 0 else 0.0
            traffic, traffic_source = None, None
            tfile = os.path.join(ROOT, TRAFFIC_FILE)
            if args.workload == "c2" and args.precision in ("bf16", "f16", "f16p") and os.path.exists(tfile):
                # HBM-side bytes per launch of this kernel class: NOT measured in this run (PMC counters need rocprofv3
                # around the process); collected with rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes over this
                # workload (tools/pmc_pass.sh over tools/pmc_one.py) and committed under profiles/; refreshed whenever the GEMM changes
                tj = json.load(open(tfile))
                traffic = tj["traffic_bytes_per_launch"]
                traffic_source = "%s (%s)" % (TRAFFIC_FILE, tj.get("collected", "rocprofv3 --pmc passes of this command"))
            result["roofline"] = {
                "bound": "mfma", "kernel": "gemm_tn_glds_kernel (all DiT projections / FFN, fused epilogues)",
                "achieved": ach, "peak": peak, "unit": "TFLOP/s", "frac": ach / peak, "traffic": traffic,
                "traffic_source": traffic_source,
                "launches": gm["launches"], "avg_launch_us": gm["ms"] * 1e3 / max(gm["launches"], 1),
                "flops_per_launch_avg": gm["flops"] / max(gm["launches"], 1),
                "note": "durations are per-launch HIP event pairs on an eager pass over the same workload: each includes its launch "
                        "boundary (the classes sum to more than ms_per_step), so `achieved` is a LOWER bound; rocprofv3's kernel-only "
                        "averages of the same command are in profiles/ (r03_k_rocprof_kernel_stats_c2_summary.txt)",
            }
            result["kernel_classes"] = {
                k: {"ms": round(v["ms"], 3), "launches": v["launches"],
                    "tflops": (v["flops"] / (v["ms"] * 1e-3) / 1e12) if v["ms"] > 0 and v["flops"] > 0 else None}
                for k, v in prof.items()}
            if args.workload != "c5":  # algorithmic FLOPs on VALID tokens only, so padding waste shows as lost efficiency
                total_fl = 2 * args.nfe * sum(dit_flops_per_seq_forward(d) for d in durs)
                result["whole_path_tflops"] = total_fl / (ms_per_step * 1e-3) / 1e12

    # ---- the same step at every operand precision, each with its own accuracy (rank 0 of a 1-GPU run, C2 only)
    if world == 1 and args.workload == "c2" and not args.no_precisions:
        ksteps = max(1, min(args.steps, 5))
        _, ref_model = (tr, model) if args.precision == "f32" else build_model("f32")
        ref_out, ref_traj = ref_model.sample(cond, text, dur_t, lens=lens_t, **kw)
        torch.cuda.synchronize()
        precs = {}
        for prec in ("f32", "f16x3", "f16p", "f16", "bf16"):
            m = model if prec == args.precision else (ref_model if prec == "f32" else build_model(prec)[1])
            for _ in range(3):   # arena growth, graph capture, clocks
                step(m)
            torch.cuda.synchronize()
            a = time.perf_counter()
            for _ in range(ksteps):
                o_, _w = step(m)
            torch.cuda.synchronize()
            ms = (time.perf_counter() - a) / ksteps * 1e3
            o2, t2 = m.sample(cond, text, dur_t, lens=lens_t, **kw)
            precs[prec] = {"ms_per_step": ms, "audio_sec_per_wall_sec": audio_per_step / (ms * 1e-3),
                           "rtf_wall_over_audio": ms * 1e-3 / audio_per_step,
                           "mel_linf_vs_f32_engine": float((o2[:, ref:] - ref_out[:, ref:]).abs().max()),
                           "traj_linf_vs_f32_engine": float((t2 - ref_traj).abs().max()), "steps_timed": ksteps}
            del m
        precs["note"] = ("north_star tolerance: 1e-3 mel L-inf against the reference CPU path; the f32 engine is pinned against the "
                         "CPU oracle at this exact size (N=1024, NFE=16) at <= 1e-5 by tests/test_configs_gpu.py; state magnitude ~8")
        result["precisions"] = precs
        # the speed claim that comes WITH the parity claim: the fastest precision whose generated mel stays within north_star's
        # 1e-3 of the f32 engine here (and of the CPU oracle in tests/test_configs_gpu.py) -- `value` above is the BASELINE
        # config's dtype (bf16), which does not meet that bar
        ok = [(v["ms_per_step"], k) for k, v in precs.items() if isinstance(v, dict) and v["traj_linf_vs_f32_engine"] < 5e-4]
        if ok:
            ms_p, k_p = min(ok)
            result["parity_precision"] = k_p
            result["value_at_parity"] = audio_per_step / (ms_p * 1e-3)
            result["ms_per_step_at_parity"] = ms_p
            result["rtf_wall_over_audio_at_parity"] = ms_p * 1e-3 / audio_per_step
        del ref_model

    # ---- C3 (BASELINE.json configs[2]; the reference's batch workload, benchmark.py:413-424: steps=32, cfg 2, sway -1): B=32
    # variable-length utterances padded to 1024 frames, NFE=32, with the default attn_mask_enabled=False (pad rows computed, the
    # reference's behaviour) and with attn_mask_enabled=True (the engine runs the valid rows only, RowPack)
    if world == 1 and args.workload == "c2" and not args.no_c3:
        cond3_cpu, text3, durs3, refs3 = make_inputs(P, args, rank, workload="c3", B=32)
        cond3 = cond3_cpu.to(dev)
        kw3 = dict(steps=32, cfg_strength=2.0, sway_sampling_coef=-1.0, seed=0)
        audio3 = sum(d - r for d, r in zip(durs3, refs3)) * 256 / 24000
        fl3 = 2 * 32 * sum(dit_flops_per_seq_forward(d) for d in durs3)
        rec3 = {"workload": "C3: F5-TTS Base %s, 32 variable-length utterances (N_i ~ U{384..1024} seed 1234, prompt N_i/4) padded to "
                            "1024 frames, NFE=32 EPSS, cfg 2.0, sway -1; one sample() call per step, no vocoder" % args.precision,
                "unit": "audio_sec/wall_sec", "generated_audio_sec_per_step": audio3, "steps_timed": 2,
                "valid_frames": sum(durs3), "padded_frames": 32 * max(durs3)}
        for mask in (False, True):
            m3 = model if (mask == bool(args.attn_mask)) else build_model(args.precision, attn_mask=mask)[1]

            def step3():
                return m3.sample(cond3, text3, torch.tensor(durs3), lens=torch.tensor(refs3), **kw3)[0]

            for _ in range(3):   # arena growth (eager), graph capture, first replay
                o3 = step3()
            torch.cuda.synchronize()
            with ClockWatch(dev.index) as cw3:
                a = time.perf_counter()
                for _ in range(2):
                    o3 = step3()
                torch.cuda.synchronize()
                ms3 = (time.perf_counter() - a) / 2 * 1e3
            assert torch.isfinite(o3).all()
            rec3["attn_mask_enabled=%s" % mask] = {
                "ms_per_step": ms3, "value": audio3 / (ms3 * 1e-3), "rtf_wall_over_audio": ms3 * 1e-3 / audio3, "clocks": cw3.summary(),
                "whole_path_tflops_valid_tokens": fl3 / (ms3 * 1e-3) / 1e12,
                "rows": "valid rows only (RowPack)" if mask else "all padded rows (bug-compatible with the reference's unmasked softmax)"}
            del m3, o3
        result["c3"] = rec3
        del cond3

    # ---- C4: the 256-utterance job sharded data-parallel over the ranks (strong scaling), one all_gather of the mel
    if args.workload == "c2" and not args.no_c4:
        from f5_tts_amd import dist as D
        conds4, texts4, durs4 = make_c4_job(P, args.c4_utts)
        conds4 = [c.to(dev) for c in conds4]          # prompts resident in HBM; text ids stay on the host (see above)
        gen4 = sum(d - d // 4 for d in durs4)
        kw4 = dict(steps=16, cfg_strength=2.0, sway_sampling_coef=-1.0, seed=0)

        def job():
            return D.dp_sample(model, conds4, texts4, durs4, batch_size=32, device=dev,
                               collective_on_host=args.rehearse_one_gpu, **kw4)

        for _ in range(args.c4_warm_passes):
            job()
        barrier()
        with ClockWatch(dev.index) as cw4:
            t0 = time.perf_counter()
            mels, _lens = job()
            barrier()
            el4 = max_over_ranks(time.perf_counter() - t0)
        assert torch.isfinite(mels).all() and mels.shape == (args.c4_utts, max(durs4), 100)
        shards4 = D.partition(durs4, world, 32)
        audio4 = gen4 * 256 / 24000
        valid4 = sum(D.utterance_cost(d) for d in durs4)
        result["c4"] = {
            "workload": "C4: %d synthetic utterances (N_i ~ U{384..1024} seed 1234, prompt N_i/4), F5-TTS Base %s, NFE=16 EPSS, cfg 2.0, "
                        "sway -1, sharded over %d rank(s) by dist.dp_sample (contiguous slices of the length-sorted list with equal "
                        "padded cost, frame-budget batches of <= 32 utterances / %d frames), ONE all_gather of the generated mel; "
                        "no vocoder" % (args.c4_utts, args.precision, world, D.MAX_BATCH_FRAMES),
            "value": audio4 / el4, "unit": "audio_sec/wall_sec", "scaling": "strong", "n_gpus": world, "wall_sec": el4, "clocks": cw4.summary(),
            "generated_audio_sec": audio4, "utterances_per_rank": [len(s) for s in shards4],
            "batches_per_rank": [len(D.batches_of(s, durs4, 32)) for s in shards4],
            "padded_over_ideal_cost_per_rank": [round(D.padded_cost(s, durs4, 32) / (valid4 / world), 4) for s in shards4],
            "all_gather_bytes_per_rank": max(len(s) for s in shards4) * max(durs4) * 100 * 4,
            "whole_path_tflops": 2 * 16 * sum(dit_flops_per_seq_forward(d) for d in durs4) / el4 / 1e12,
        }
        del mels
        # ---- what 8 GPUs would do, measured on this one: rank r's shard of partition(durs, 8) run alone (same warm passes:
        # eager, capture, then the timed replay), r = 0..7.  predicted_scaling_8 = this job's wall / the slowest shard's wall:
        # everything but the all_gather (8 x ~19 MB over xGMI, ~1 ms) and host contention between the 8 rank processes.
        if world == 1 and not args.no_shard8:
            shards8 = D.partition(durs4, 8, 32)
            per_rank8 = max(len(s) for s in shards8)
            walls8 = []
            for sh in shards8:
                def one():
                    return D.run_shard(model, conds4, texts4, durs4, sh, per_rank=per_rank8, batch_size=32, device=dev, **kw4)
                for _ in range(args.c4_warm_passes):
                    one()
                torch.cuda.synchronize()
                a = time.perf_counter()
                lb = one()
                torch.cuda.synchronize()
                walls8.append(time.perf_counter() - a)
                del lb
            result["c4"].update({
                "shard8_wall_sec": [round(x, 4) for x in walls8], "shard8_utterances": [len(s) for s in shards8],
                "shard8_batches": [[len(b) for b in D.batches_of(s, durs4, 32)] for s in shards8],
                "predicted_scaling_8": el4 / max(walls8),
                "predicted_scaling_8_note": "wall_sec of this 1-GPU job / max(shard8_wall_sec): each of the 8 shards of "
                                            "partition(durs, 8) timed alone on this GPU after the same warm passes; excludes the "
                                            "single all_gather (~19 MB per rank) and host contention between rank processes"})

    if rank == 0:
        if world == 1 and not args.no_cpu_baseline and args.workload == "c2":
            try:
                result["cpu_baseline"] = cpu_baseline(P, args, tr.state_dict(), voc.state_dict(), cond_cpu, text_cpu)
            except Exception as ex:  # the baseline is a report, never a reason to lose the bench line
                result["cpu_baseline"] = {"value": None, "error": repr(ex)}
        print(json.dumps(result))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
