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


def _c4_durations():
    g = torch.Generator().manual_seed(1234)
    return [1024] + [int(x) for x in torch.randint(384, 1025, (255,), generator=g)]   # bench.py::make_c4_job (BASELINE C4)


def test_partition_is_complete_contiguous_and_deterministic():
    durs = _c4_durations()
    order = sorted(range(256), key=lambda i: (-durs[i], i))
    for world in (1, 2, 3, 4, 8):
        shards = D.partition(durs, world)
        assert len(shards) == world
        assert [i for s in shards for i in s] == order          # contiguous slices of the length-sorted list, nothing lost
        assert D.partition(durs, world) == shards               # every rank derives the same assignment
        for s in shards:                                        # a rank's batches are consecutive runs of its shard
            assert [i for b in D.batches_of(s, durs, 32) for i in b] == s
            for b in D.batches_of(s, durs, 32):
                assert len(b) <= 32 and (len(b) == 1 or len(b) * durs[b[0]] <= D.MAX_BATCH_FRAMES)


def test_partition_padded_cost_is_balanced():
    """What a rank pays is the PADDED cost of its batches (attn_mask_enabled=False computes pad rows): the slowest rank
    must stay within 10 % of an ideal split of the valid work (round 2's LPT deal: 1.51 at world 8); with the GEMMs' whole
    rounds of tiles counted as well (D.batch_cost, what partition() balances) within 13 %, and the ranks within 10 % of each other."""
    durs = _c4_durations()
    total_valid = sum(D.utterance_cost(d) for d in durs)
    for world in (1, 2, 4, 8):
        shards = D.partition(durs, world)
        padded = [sum(len(b) * D.utterance_cost(durs[b[0]]) for b in D.batches_of(s, durs, 32)) for s in shards]
        assert max(padded) <= 1.10 * total_valid / world, (world, max(padded) / (total_valid / world))
        quant = [D.padded_cost(s, durs, 32) for s in shards]
        assert max(quant) <= 1.13 * total_valid / world, (world, max(quant) / (total_valid / world))
        assert min(quant) >= 0.90 * max(quant), (world, min(quant) / max(quant))
        # per batch: padded frames / valid frames
        for s in shards:
            for b in D.batches_of(s, durs, 32):
                assert len(b) * durs[b[0]] <= 1.12 * sum(durs[i] for i in b)
    # the unit of batch_cost: one full round of tiles = 8 utterances x 1,024 frames with CFG
    assert D.batch_cost(8, 1024) == 8 * D.utterance_cost(1024)
    assert D.batch_cost(11, 1024) > 11 * D.utterance_cost(1024) * 1.15      # 22,528 rows: 1.4 rounds of the N = 1024 GEMMs run as 2


def test_partition_edge_cases():
    assert D.partition([], 4) == [[], [], [], []]
    assert D.partition([50, 70], 4) == [[1], [0], [], []]                    # more ranks than utterances: empty shards
    assert D.partition([100] * 7, 2) in ([[0, 1, 2, 3], [4, 5, 6]], [[0, 1, 2], [3, 4, 5, 6]])
    one = D.partition([30, 20, 90, 40], 1)
    assert one == [[2, 3, 0, 1]]
    assert D.batches_of([2, 3, 0, 1], [30, 20, 90, 40], batch_size=3, max_batch_frames=100) == [[2], [3, 0], [1]]
    # frame budget: a batch never exceeds max_batch_frames padded frames unless it is a single utterance
    bs = D.batches_of(list(range(6)), [60, 60, 60, 60, 60, 60], batch_size=32, max_batch_frames=130)
    assert [len(b) for b in bs] == [2, 2, 2]


def _worker_sparse(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    conds, texts, durs = make_job(2)                          # 2 utterances on 3 ranks: rank 2's shard is empty
    mels, _ = D.dp_sample(FakeModel(), conds, texts, durs, batch_size=3, device="cpu")
    q.put((rank, bool(torch.allclose(mels, expected(conds, texts, durs))), 1))
    dist.destroy_process_group()


def test_dp_sample_world3_with_an_empty_shard():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_sparse, args=(r, 3, port, q)) for r in range(3)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert all(r[1] for r in res), res


def test_world1_no_collective():
    conds, texts, durs = make_job(5)
    mels, _ = D.dp_sample(FakeModel(), conds, texts, durs, batch_size=2, device="cpu")
    assert torch.allclose(mels, expected(conds, texts, durs))


def test_bench_refuses_more_ranks_than_gpus():
    """`python bench.py --gpus N` outside torchrun starts its own N ranks -- and must fail loudly, before touching a GPU, when fewer
    than N devices are visible (round 2 printed a one-GPU number with `n_gpus: 1` and rc 0).  No GPU here: --gpus 2 exits 2."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    if torch.cuda.device_count() >= 2:
        return
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 2 and "GPU(s) are visible" in r.stderr and r.stdout.strip() == ""
    # inside a launcher whose world size disagrees with --gpus: refuse as well
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2"], env=dict(env, WORLD_SIZE="4", RANK="0", LOCAL_RANK="0"),
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "WORLD_SIZE=4" in (r.stderr + r.stdout)


def test_clock_watch_is_inert_without_gpu_sysfs():
    """bench.py's ClockWatch must never fail a run: with no GPU (here) or no hwmon files it samples nothing and reports None."""
    import importlib
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    bench = importlib.import_module("bench")
    cw = bench.ClockWatch(0)
    with cw:
        pass
    assert cw.summary() is None
