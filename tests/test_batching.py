"""Frame-budget bucketing (korean-f5-tts_amd/batching.py) against what the reference's `get_inference_prompt`
(src/f5_tts/eval/utils_eval.py:72-205) returned for the same prompts: tests/golden/bucketing.json was produced by running
that function on length-only fakes (oracle/make_golden.py::bucketing_case)."""
import json
import os

import f5_tts_amd as P
from f5_tts_amd import batching

HERE = os.path.dirname(os.path.abspath(__file__))


def _fixture():
    with open(os.path.join(HERE, "golden", "bucketing.json")) as fh:
        return json.load(fh)


def test_bucketing_matches_reference_batches():
    fx = _fixture()
    prompts = fx["prompts"]
    for case in fx["cases"]:
        texts, totals, refs = [], [], []
        for p in prompts:
            ref_mel_len = p["nsamples"] // fx["hop_length"] + 1          # center=True framing of the mel front-end
            t, n = batching.prompt_text_and_frames(ref_mel_len, p["prompt_text"], p["gt_text"], case["speed"])
            texts.append(t)
            totals.append(n)
            refs.append(ref_mel_len)
        got = batching.bucket_prompts(totals, case["infer_batch_size"], min_secs=case["min_secs"], max_secs=case["max_secs"])
        want = case["batches"]
        assert len(got) == len(want)
        for g, w in zip(got, want):
            assert [prompts[i]["utt"] for i in g] == w["utts"]
            assert [totals[i] for i in g] == w["total_mel_lens"]
            assert [refs[i] for i in g] == w["ref_mel_lens"]
            assert [texts[i] for i in g] == w["texts"]
        assert sorted(i for b in got for i in b) == list(range(len(prompts))), "every utterance exactly once"


def test_frame_budget_and_residual_flush():
    # hand-computed: min_tokens = 3*24000//256 = 281, max_tokens = 3750, 10 buckets of width 347
    lens = [300, 310, 2000, 320, 3750, 330]
    got = batching.bucket_prompts(lens, infer_batch_size=900, num_buckets=10, shuffle_seed=None)
    # 2000 (bucket 4) alone meets the budget -> emitted at once; bucket 0 holds 300, 310, 320 -> emitted at 930 >= 900;
    # 3750 (bucket 9) at once; the 330 left in bucket 0 is flushed at the end
    assert got == [[2], [0, 1, 3], [4], [5]]


def test_split_between_processes_contiguous_chunks():
    items = list(range(10))
    parts = [batching.split_between_processes(items, 4, r) for r in range(4)]
    assert parts == [[0, 1, 2], [3, 4, 5], [6, 7], [8, 9]]
    assert batching.split_between_processes([], 3, 1) == []
