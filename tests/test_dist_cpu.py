"""CPU, world_size 2 over gloo: the data-parallel driver (partition -> per-rank sample() -> ONE all_gather -> reorder).
The per-rank sampler is a deterministic stand-in (there is no GPU here); what is under test is the N > 1 path's
bookkeeping and its single collective."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import f5_tts_amd as P
from f5_tts_amd import dist as D


class FakeModel:
    """sample() returns a mel that encodes (utterance content, frame index) so that misplaced rows are detected."""
    device = torch.device("cpu")
    calls = 0

    def sample(self, cond, text, duration, *, lens=None, **kw):
        FakeModel.calls += 1
        B, N = cond.shape[0], int(duration.max())
        out = torch.zeros(B, N, cond.shape[-1])
        for b in range(B):
            key = float(text[b][text[b] >= 0].sum())
            out[b, :int(duration[b])] = key + torch.arange(int(duration[b]))[:, None] * 1e-3
        return out, None


def make_job(n=11, seed=3):
    g = torch.Generator().manual_seed(seed)
    durs = [int(x) for x in torch.randint(20, 90, (n,), generator=g)]
    conds = [torch.randn(d // 4, 8, generator=g) for d in durs]
    texts = [torch.randint(0, 50, (5 + i % 4,), generator=g) for i in range(n)]
    return conds, texts, durs


def expected(conds, texts, durs):
    n_max = max(durs)
    out = torch.zeros(len(durs), n_max, 8)
    for i, d in enumerate(durs):
        out[i, :d] = float(texts[i].sum()) + torch.arange(d)[:, None] * 1e-3
    return out


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    calls = {"n": 0}
    orig = dist.all_gather_into_tensor

    def counting(*a, **k):
        calls["n"] += 1
        return orig(*a, **k)

    dist.all_gather_into_tensor = counting
    conds, texts, durs = make_job()
    mels, lens = D.dp_sample(FakeModel(), conds, texts, durs, batch_size=3, device="cpu")
    ok = torch.allclose(mels, expected(conds, texts, durs)) and lens == durs and calls["n"] == 1
    q.put((rank, bool(ok), calls["n"]))
    dist.destroy_process_group()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_dp_sample_world2_gloo_single_all_gather():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(r[0] for r in res) == [0, 1]
    assert all(r[1] for r in res), res
    assert all(r[2] == 1 for r in res), "exactly one collective on the data path"


def test_partition_is_balanced_and_complete():
    g = torch.Generator().manual_seed(1234)
    durs = [int(x) for x in torch.randint(384, 1025, (256,), generator=g)]  # BASELINE config C4 lengths
    shards = D.partition(durs, 8)
    assert sorted(i for s in shards for i in s) == list(range(256))
    assert all(len(s) == 32 for s in shards)
    loads = [sum(D.utterance_cost(durs[i]) for i in s) for s in shards]
    assert max(loads) / min(loads) < 1.01
    assert D.partition(durs, 8) == shards  # deterministic: every rank derives the same assignment


def test_world1_no_collective():
    conds, texts, durs = make_job(5)
    mels, _ = D.dp_sample(FakeModel(), conds, texts, durs, batch_size=2, device="cpu")
    assert torch.allclose(mels, expected(conds, texts, durs))
