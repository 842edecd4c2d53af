"""Data-parallel synthesis over the GPUs of one node: one process per GPU, utterances sharded by cost, NO communication
during the ODE solve, and ONE all_gather of the generated mel at the end (RCCL over xGMI on GPUs; gloo in the CPU tests).

The reference shards the same way -- a static split of the prompt list per process and barriers only
(src/f5_tts/eval/eval_infer_batch.py:28,178-214; runtime/triton_trtllm/benchmark.py:199-212,340-341) -- and never
moves a tensor between ranks; the all_gather is what BASELINE.json's north_star adds on top.
"""
from __future__ import annotations

import torch
import torch.distributed as dist


def utterance_cost(n_frames: int) -> float:
    """Relative cost of one utterance of n frames: linear GEMM/conv terms + quadratic attention term
    (per block 16 N D^2 + 4 N^2 D with D = 1024)."""
    return 16.0 * n_frames * 1024 + 4.0 * n_frames * n_frames


# One backbone call works on at most 32,768 rows (csrc/engine.hip::chunk_utts: the activations of a chunk stay inside the
# Infinity Cache) = 16,384 frames with CFG: a batch larger than that is stepped chunk by chunk anyway, so a bigger batch
# buys nothing and only pads more rows to its longest member.
MAX_BATCH_FRAMES = int(__import__("os").environ.get("F5_MAX_BATCH_FRAMES", "16384"))   # (the override is for tools / A-B runs)


# Row granules of the four block GEMMs on 256 CUs with 256 x 256 output tiles (csrc/gemm3.h): a launch runs whole ROUNDS of 256
# tiles, so its time is that of its row count rounded up to 256 x 256 / (N / 256) rows -- QKV (N = 3072): 5,461 rows, FF1 (2048):
# 8,192, out-proj and FF2 (1024): 16,384.  Weights = each GEMM's share of the block's GEMM flops.  Measured (round 3, bench.py
# shard8): two batches of 11 utterances x 1,024 frames (22,528 rows with CFG: 1.4 rounds -> 2) cost as much as 16 + 16.
_GEMM_COL_TILES = ((3.0 / 8.0, 12), (1.0 / 8.0, 4), (2.0 / 8.0, 8), (2.0 / 8.0, 4))   # (share of the flops, N / 256) of QKV, out, FF1, FF2


def batch_cost(count: int, max_frames: int) -> float:
    """Cost of ONE batch as it runs: `count` utterances padded to `max_frames`, CFG doubling the rows; the GEMM term is quantised to
    whole rounds of tiles per GEMM, the attention term (per head and 128-query block: fine-grained) is not.  Same unit as
    utterance_cost (a batch of 8 x 1,024 frames, i.e. 16,384 rows, costs 8 x utterance_cost(1024))."""
    rows = 2 * count * max_frames
    row_tiles = -(-rows // 256)
    q = sum(w * (-(-row_tiles * c // 256)) * 65536.0 / c for w, c in _GEMM_COL_TILES)   # rows, rounded up to whole rounds of 256 tiles
    return 16.0 * 1024 * q / 2 + 4.0 * count * max_frames * max_frames


def _greedy_batches(s: list[int], durations: list[int], batch_size: int, max_batch_frames: int) -> list[list[int]]:
    out: list[list[int]] = []
    k = 0
    while k < len(s):
        cap = max(1, min(batch_size, max_batch_frames // max(durations[s[k]], 1)))
        out.append(s[k:k + cap])
        k += cap
    return out


def batches_of(shard: list[int], durations: list[int], batch_size: int, max_batch_frames: int = MAX_BATCH_FRAMES) -> list[list[int]]:
    """Length-sorted batches of a shard under a FRAME budget, the reference's batching rule (utils_eval.py:146-202:
    `infer_batch_size` counts mel frames, not utterances): a batch takes neighbours of the sorted list while
    count x (its longest member) <= max_batch_frames and count <= batch_size.  The last two batches are evened out when that is
    cheaper under batch_cost (it usually is not: 16 + 6 utterances of 1,024 frames run 3 rounds of GEMM tiles, 11 + 11 run 4)."""
    s = sorted(shard, key=lambda i: (-durations[i], i))
    out = _greedy_batches(s, durations, batch_size, max_batch_frames)
    if len(out) >= 2:
        both = out[-2] + out[-1]
        h = (len(both) + 1) // 2
        even = [both[:h], both[h:]]
        cost = lambda bs: sum(batch_cost(len(b), durations[b[0]]) for b in bs)
        if cost(even) < cost(out[-2:]):
            out[-2:] = even
    return out


def padded_cost(shard: list[int], durations: list[int], batch_size: int, max_batch_frames: int = MAX_BATCH_FRAMES) -> float:
    """Cost of a shard AS IT RUNS: every batch is padded to its longest member -- with the default attn_mask_enabled=False the
    pad rows are computed like any other (modules.py:499-508: no key mask) -- and its GEMMs run whole rounds of tiles."""
    return sum(batch_cost(len(b), durations[b[0]]) for b in batches_of(shard, durations, batch_size, max_batch_frames))


def partition(durations: list[int], world: int, batch_size: int = 32, max_batch_frames: int = MAX_BATCH_FRAMES) -> list[list[int]]:
    """Shards of utterance indices, one per rank: CONTIGUOUS slices of the length-sorted list (so that a rank's batches
    are length-homogeneous: the reference buckets by length for the same reason, utils_eval.py:146-202) cut so that the
    largest PADDED cost of a rank is minimal (binary search on the bound + greedy longest feasible slice; shard sizes
    differ: the rank with the longest utterances gets fewer of them).  Deterministic, so every rank derives the same
    assignment and no metadata is exchanged.  Ranks past the number of utterances get empty shards."""
    order = sorted(range(len(durations)), key=lambda i: (-durations[i], i))
    n = len(order)
    if n == 0:
        return [[] for _ in range(world)]
    valid = [utterance_cost(durations[i]) for i in order]

    def cut(bound: float) -> list[list[int]]:
        slices, i = [], 0
        while i < n:
            best, v = i + 1, 0.0
            for j in range(i, n):
                v += valid[j]                       # valid cost <= padded cost: nothing longer can fit once it is over
                if v > bound and j > i:
                    break
                if padded_cost(order[i:j + 1], durations, batch_size, max_batch_frames) <= bound:
                    best = j + 1
            slices.append(order[i:best])
            i = best
        return slices

    lo, hi = 0.0, padded_cost(order, durations, batch_size, max_batch_frames)   # hi: everything in one shard
    for _ in range(32):
        mid = 0.5 * (lo + hi)
        if len(cut(mid)) <= world:
            hi = mid
        else:
            lo = mid
    shards = cut(hi)
    return shards + [[] for _ in range(world - len(shards))]


@torch.no_grad()
def run_shard(model, conds: list[torch.Tensor], texts: list[torch.Tensor], durations: list[int], shard: list[int], *,
              per_rank: int, batch_size: int = 32, max_batch_frames: int = MAX_BATCH_FRAMES, device=None, **sample_kw) -> torch.Tensor:
    """One rank's part of the job: its shard (already in length-sorted order) batch by batch through model.sample().
    Returns f32[per_rank, N_max, mel]: slot k = the shard's k-th utterance, zero past its own length."""
    n_max = max(durations)
    mel_dim = conds[0].shape[-1]
    device = device if device is not None else getattr(model, "device", conds[0].device)
    local = torch.zeros(per_rank, n_max, mel_dim, device=device, dtype=torch.float32)
    k0 = 0
    for batch in batches_of(shard, durations, batch_size, max_batch_frames):
        assert batch == shard[k0:k0 + len(batch)], "shards are slices of the length-sorted list"
        ref_lens = [conds[u].shape[0] for u in batch]
        cond = torch.nn.utils.rnn.pad_sequence([conds[u] for u in batch], batch_first=True)
        text = torch.nn.utils.rnn.pad_sequence([texts[u] for u in batch], batch_first=True, padding_value=-1)
        dur = torch.tensor([durations[u] for u in batch], dtype=torch.long)
        out, _ = model.sample(cond, text, dur, lens=torch.tensor(ref_lens, dtype=torch.long), **sample_kw)
        # the batch's rows are consecutive slots: one masked store (frames past an utterance's own length are zeroed)
        nb = out.shape[1]
        valid = (torch.arange(nb)[None, :] < dur[:, None]).to(device=device, dtype=local.dtype, non_blocking=True)
        torch.mul(out.to(local.dtype), valid[..., None], out=local[k0:k0 + len(batch), :nb])
        k0 += len(batch)
    return local


@torch.no_grad()
def dp_sample(model, conds: list[torch.Tensor], texts: list[torch.Tensor], durations: list[int], *, batch_size: int = 32,
              max_batch_frames: int = MAX_BATCH_FRAMES, group=None, device=None, collective_on_host: bool = False, **sample_kw):
    """Synthesises every utterance of the job on its owner rank and returns ALL mels on every rank.

    conds[i]: f32[ref_i, mel] prompt mel; texts[i]: i64[nt_i]; durations[i]: total frames.
    Returns (mels, lens): mels f32[n_utt, N_max, mel] in the original order (zero padded), lens list[int].
    Exactly one collective: all_gather_into_tensor of a [per_rank_max, N_max, mel] buffer.
    collective_on_host: run that collective on host copies (rehearsals of N ranks on ONE GPU over gloo only).
    """
    import os

    from .utils import configure_host_threads
    configure_host_threads(int(os.environ.get("F5_HOST_THREADS", "1")))   # one rank per GPU shares the host: see utils.py
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    shards = partition(durations, world, batch_size, max_batch_frames)
    n_max = max(durations)
    per_rank = max(len(s) for s in shards)            # shard sizes differ (equal COST, not equal count): slots are padded
    mel_dim = conds[0].shape[-1]
    device = device if device is not None else getattr(model, "device", conds[0].device)
    local = run_shard(model, conds, texts, durations, shards[rank], per_rank=per_rank, batch_size=batch_size,
                      max_batch_frames=max_batch_frames, device=device, **sample_kw)
    if world == 1:
        gathered = local.unsqueeze(0)
    else:
        if collective_on_host:
            host = torch.empty(world * per_rank, n_max, mel_dim, dtype=torch.float32)
            dist.all_gather_into_tensor(host, local.cpu(), group=group)
            flat = host.to(device)
        else:
            flat = torch.empty(world * per_rank, n_max, mel_dim, device=device, dtype=torch.float32)
            dist.all_gather_into_tensor(flat, local, group=group)   # the ONE exchange of the job
        gathered = flat.view(world, per_rank, n_max, mel_dim)
    where = [0] * len(durations)                      # original order: one gather over the [world * per_rank] slots
    for r, shard in enumerate(shards):
        for k, u in enumerate(shard):
            where[u] = r * per_rank + k
    mels = gathered.reshape(world * per_rank, n_max, mel_dim).index_select(0, torch.tensor(where, dtype=torch.long).to(device))
    return mels, list(durations)
