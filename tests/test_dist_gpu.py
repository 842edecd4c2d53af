"""GPU tests of the data-parallel driver with the REAL engine (BASELINE config C4's code path) and of the RCCL call path.

A one-GPU box cannot run two RCCL ranks (RCCL refuses two ranks on one device), so
  * test A runs dist.dp_sample at world 1 through the HIP engine: the partition, the frame-budget batches, the masked
    store into the slot buffer and the final reorder all run on device tensors, and every returned mel is compared
    bit for bit with model.sample() of its own batch and, on its valid frames, with the CPU oracle (1e-3);
  * test B initialises a world-1 "nccl" process group on cuda:0 and runs the job's ONE collective,
    all_gather_into_tensor of a [slots, N, 100] device buffer -- librccl.so is loaded and a communicator is created in
    GPUTEST, not first in the driver's 8-GPU run;
  * test C runs dp_sample INSIDE that world-1 nccl group (dist.is_initialized() path).
The N > 1 bookkeeping (slots of unequal shards, empty shards, exactly one collective) is tests/test_dist_cpu.py (gloo)."""
import os
import socket

import pytest
import torch

pytestmark = pytest.mark.gpu

import f5_tts_amd as P  # noqa: E402
from f5_tts_amd import dist as D  # noqa: E402
from oracle import f5_oracle as O  # noqa: E402

DEV = "cuda:0"
NV = 40
KW = dict(steps=6, cfg_strength=2.0, sway_sampling_coef=-1.0, seed=5)


def make_job(n=12, seed=11):
    g = torch.Generator().manual_seed(seed)
    durs = [int(x) for x in torch.randint(40, 131, (n,), generator=g)]
    if n > 7:
        durs[3] = durs[7]                               # a tie in the sort key
    conds = [torch.randn(d // 4, 100, generator=g) for d in durs]
    texts = [torch.randint(1, NV - 1, (max(2, round(0.15 * d)),), generator=g) for d in durs]
    return conds, texts, durs


def build(prec="f32"):
    arch = P.config.F5TTS_TINY
    sd = P.weights.synthetic_state_dict(P.weights.dit_param_shapes(arch, NV))
    tr = P.DiT(**arch, text_num_embeds=NV, mel_dim=100, precision=prec)
    tr.load_state_dict(sd)
    return sd, arch, P.CFM(transformer=tr, mel_spec_module=P.mel.MelSpec()).to(DEV)


def check_job(model, sd, arch, mels, conds, texts, durs, batch_size, max_batch_frames, oracle=True):
    assert mels.shape == (len(durs), max(durs), 100) and mels.device.type == "cuda"
    shard = D.partition(durs, 1, batch_size, max_batch_frames)[0]
    batches = D.batches_of(shard, durs, batch_size, max_batch_frames)
    assert sorted(i for b in batches for i in b) == list(range(len(durs)))
    worst = 0.0
    for b in batches:
        cond = torch.nn.utils.rnn.pad_sequence([conds[u] for u in b], batch_first=True)
        text = torch.nn.utils.rnn.pad_sequence([texts[u] for u in b], batch_first=True, padding_value=-1)
        dur = torch.tensor([durs[u] for u in b])
        lens = torch.tensor([conds[u].shape[0] for u in b])
        own, _ = model.sample(cond.to(DEV), text, dur, lens=lens, **KW)
        ref = O.sample(sd, arch, cond, text, dur, lens=lens, **KW)[0] if oracle else None
        for k, u in enumerate(b):
            d = durs[u]
            assert torch.equal(mels[u, :d], own[k, :d]), f"utterance {u}: dp_sample differs from sample() of its own batch"
            assert not mels[u, d:].any(), f"utterance {u}: frames past its length must be zero"
            if oracle:
                worst = max(worst, (mels[u, :d].cpu() - ref[k, :d]).abs().max().item())
    return worst


@pytest.mark.parametrize("batch_size,max_batch_frames", [(4, 1 << 30), (32, 300)])
def test_dp_sample_world1_real_engine_vs_own_batches_and_oracle(batch_size, max_batch_frames):
    sd, arch, model = build("f32")
    conds, texts, durs = make_job()
    conds_dev = [c.to(DEV) for c in conds]
    mels, lens = D.dp_sample(model, conds_dev, texts, durs, batch_size=batch_size, max_batch_frames=max_batch_frames, device=DEV, **KW)
    assert lens == durs
    worst = check_job(model, sd, arch, mels, conds, texts, durs, batch_size, max_batch_frames)
    print(f"[dp_sample world 1, batch_size {batch_size}, frame budget {max_batch_frames}] mel Linf vs CPU oracle on valid frames: {worst:.3e}")
    assert worst < 1e-3
    # graph replay of the same job (third pass) gives the same bits
    for _ in range(2):
        again, _ = D.dp_sample(model, conds_dev, texts, durs, batch_size=batch_size, max_batch_frames=max_batch_frames, device=DEV, **KW)
    assert torch.equal(again, mels)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.fixture(scope="module")
def nccl_world1():
    import torch.distributed as dist
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), RANK="0", WORLD_SIZE="1")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device(DEV))
    yield dist
    dist.destroy_process_group()
    for k in ("MASTER_ADDR", "MASTER_PORT", "RANK", "WORLD_SIZE"):
        os.environ.pop(k, None)


def test_rccl_all_gather_into_tensor_world1(nccl_world1):
    dist = nccl_world1
    assert dist.get_backend() == "nccl" and dist.get_world_size() == 1
    g = torch.Generator().manual_seed(0)
    local = torch.randn(3, 257, 100, generator=g).to(DEV)
    flat = torch.empty(1 * 3, 257, 100, device=DEV)
    dist.all_gather_into_tensor(flat, local)            # the job's one collective (dist.py::dp_sample), through librccl
    torch.cuda.synchronize()
    assert torch.equal(flat, local)
    t = torch.tensor([1.25], device=DEV, dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)            # bench.py::max_over_ranks
    dist.barrier()
    assert t.item() == 1.25
    ver = torch.cuda.nccl.version()
    print(f"[rccl] version {ver}")
    mapped = [ln.split()[-1] for ln in open("/proc/self/maps") if "librccl" in ln]
    assert mapped, "librccl.so is not mapped into the process"


def test_dp_sample_inside_nccl_group_world1(nccl_world1):
    sd, arch, model = build("f32")
    conds, texts, durs = make_job(7, seed=2)
    conds_dev = [c.to(DEV) for c in conds]
    mels, _ = D.dp_sample(model, conds_dev, texts, durs, batch_size=3, device=DEV, **KW)
    check_job(model, sd, arch, mels, conds, texts, durs, 3, D.MAX_BATCH_FRAMES, oracle=False)
