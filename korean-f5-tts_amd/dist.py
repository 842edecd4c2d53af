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


def partition(durations: list[int], world: int) -> list[list[int]]:
    """Greedy longest-processing-time partition of utterance indices into `world` shards of balanced cost.
    Deterministic, so every rank derives the same assignment (no metadata exchange needed)."""
    order = sorted(range(len(durations)), key=lambda i: (-durations[i], i))
    loads = [0.0] * world
    shards: list[list[int]] = [[] for _ in range(world)]
    for i in order:
        r = min(range(world), key=lambda k: (loads[k], len(shards[k]), k))
        shards[r].append(i)
        loads[r] += utterance_cost(durations[i])
    return shards


def batches_of(shard: list[int], durations: list[int], batch_size: int) -> list[list[int]]:
    """Length-sorted batches of at most `batch_size` utterances (padding waste is smallest between neighbours)."""
    s = sorted(shard, key=lambda i: (-durations[i], i))
    return [s[k:k + batch_size] for k in range(0, len(s), batch_size)]


@torch.no_grad()
def dp_sample(model, conds: list[torch.Tensor], texts: list[torch.Tensor], durations: list[int], *, batch_size: int = 32,
              group=None, device=None, collective_on_host: bool = False, **sample_kw):
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
    shards = partition(durations, world)
    mine = shards[rank]
    n_max = max(durations)
    per_rank = max(len(s) for s in shards)
    mel_dim = conds[0].shape[-1]
    device = device if device is not None else getattr(model, "device", conds[0].device)
    local = torch.zeros(per_rank, n_max, mel_dim, device=device, dtype=torch.float32)
    slot = {u: k for k, u in enumerate(mine)}
    for batch in batches_of(mine, durations, batch_size):
        ref_lens = [conds[u].shape[0] for u in batch]
        cond = torch.nn.utils.rnn.pad_sequence([conds[u] for u in batch], batch_first=True)
        text = torch.nn.utils.rnn.pad_sequence([texts[u] for u in batch], batch_first=True, padding_value=-1)
        dur = torch.tensor([durations[u] for u in batch], dtype=torch.long)
        out, _ = model.sample(cond, text, dur, lens=torch.tensor(ref_lens, dtype=torch.long), **sample_kw)
        # the batch's rows go to their slots in ONE masked scatter (frames past an utterance's own length are zeroed)
        nb = out.shape[1]
        valid = (torch.arange(nb)[None, :] < dur[:, None]).to(device=device, dtype=local.dtype, non_blocking=True)
        idx = torch.tensor([slot[u] for u in batch], dtype=torch.long).to(device, non_blocking=True)
        local[:, :nb].index_copy_(0, idx, out.to(local.dtype) * valid[..., None])
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
