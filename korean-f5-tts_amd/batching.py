"""Length bucketing of an evaluation / batch-synthesis job into frame-budget batches, with the reference's behaviour
(src/f5_tts/eval/utils_eval.py:72-205, `get_inference_prompt`) -- SURVEY.md section 8(f) row 4.

The reference walks the prompt list once, drops each utterance into one of `num_buckets` length buckets, emits a bucket
as a batch as soon as its accumulated TOTAL mel frames reach `infer_batch_size` (a frame budget, not an utterance count),
flushes the leftovers bucket by bucket, and finally shuffles the batches with `random.seed(666)` so that the last ranks
of `split_between_processes` (eval_infer_batch.py:181) do not only get the short leftovers.  Only the index arithmetic
lives here: audio loading, resampling and the mel front-end are the callers' (infer.py / mel.py)."""
from __future__ import annotations

import math
import random


def prompt_text_and_frames(ref_mel_len: int, prompt_text: str, gt_text: str, speed: float = 1.0) -> tuple[str, int]:
    """utils_eval.py:117-119,140-143: a single-byte last character gets a trailing space; total frames = prompt frames +
    int(prompt frames / prompt bytes * target bytes / speed).  Returns (prompt_text + gt_text, total_mel_len)."""
    if len(prompt_text[-1].encode("utf-8")) == 1:
        prompt_text = prompt_text + " "
    ref_text_len = len(prompt_text.encode("utf-8"))
    gen_text_len = len(gt_text.encode("utf-8"))
    return prompt_text + gt_text, ref_mel_len + int(ref_mel_len / ref_text_len * gen_text_len / speed)


def bucket_prompts(total_mel_lens: list[int], infer_batch_size: int = 1, num_buckets: int = 200, min_secs: int = 3,
                   max_secs: int = 40, target_sample_rate: int = 24000, hop_length: int = 256,
                   shuffle_seed: int | None = 666) -> list[list[int]]:
    """Batches of utterance indices in the order `get_inference_prompt` returns them (utils_eval.py:91-96,146-202)."""
    assert infer_batch_size > 0, "infer_batch_size should be greater than 0."
    min_tokens = min_secs * target_sample_rate // hop_length
    max_tokens = max_secs * target_sample_rate // hop_length
    accum = [0] * num_buckets
    members: list[list[int]] = [[] for _ in range(num_buckets)]
    batches: list[list[int]] = []
    for i, n in enumerate(total_mel_lens):
        assert min_tokens <= n <= max_tokens, (
            f"utterance {i} has duration {n * hop_length // target_sample_rate}s out of range [{min_secs}, {max_secs}].")
        b = math.floor((n - min_tokens) / (max_tokens - min_tokens + 1) * num_buckets)
        members[b].append(i)
        accum[b] += n
        if accum[b] >= infer_batch_size:
            batches.append(members[b])
            accum[b] = 0
            members[b] = []
    for b, frames in enumerate(accum):
        if frames > 0:
            batches.append(members[b])
    if shuffle_seed is not None:
        random.Random(shuffle_seed).shuffle(batches)   # == random.seed(666); random.shuffle(prompts_all)
    return batches


def split_between_processes(items: list, world: int, rank: int) -> list:
    """accelerate's `split_between_processes` for a list (eval_infer_batch.py:181): contiguous chunks, the first
    len % world ranks get one extra item."""
    n = len(items)
    base, extra = divmod(n, world)
    start = rank * base + min(rank, extra)
    return items[start:start + base + (1 if rank < extra else 0)]
