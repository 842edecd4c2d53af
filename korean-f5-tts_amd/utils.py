"""Host-side helpers of the sampler with the reference's names and behaviour (model/utils.py:32-37,53-58,92-106,
538-551).  Pure Python / torch-CPU bookkeeping; no model arithmetic happens here."""
from __future__ import annotations

import torch
from torch.nn.utils.rnn import pad_sequence


def exists(v):
    return v is not None


def default(v, d):
    return v if exists(v) else d


def lens_to_mask(t: torch.Tensor, length: int | None = None) -> torch.Tensor:
    if not exists(length):
        length = int(t.amax())
    seq = torch.arange(length, device=t.device)
    return seq[None, :] < t[:, None]


def list_str_to_tensor(text: list[str], padding_value=-1) -> torch.Tensor:
    tensors = [torch.tensor([*bytes(t, "UTF-8")]) for t in text]
    return pad_sequence(tensors, padding_value=padding_value, batch_first=True)


def list_str_to_idx(text, vocab_char_map: dict[str, int], padding_value=-1) -> torch.Tensor:
    tensors = [torch.tensor([vocab_char_map.get(c, 0) for c in t]) for t in text]
    return pad_sequence(tensors, padding_value=padding_value, batch_first=True)


_EPSS_TABLE = {
    5: [0, 2, 4, 8, 16, 32],
    6: [0, 2, 4, 6, 8, 16, 32],
    7: [0, 2, 4, 6, 8, 16, 24, 32],
    10: [0, 2, 4, 6, 8, 12, 16, 20, 24, 28, 32],
    12: [0, 2, 4, 6, 8, 10, 12, 14, 16, 20, 24, 28, 32],
    16: [0, 1, 2, 3, 4, 5, 6, 7, 8, 10, 12, 14, 16, 20, 24, 28, 32],
}


def get_epss_timesteps(n, device="cpu", dtype=torch.float32):
    t = _EPSS_TABLE.get(n, [])
    if not t:
        return torch.linspace(0, 1, n + 1, device=device, dtype=dtype)
    return (1 / 32) * torch.tensor(t, device=device, dtype=dtype)


def load_vocab(path: str) -> tuple[dict[str, int], int]:
    """vocab.txt -> ({char: idx}, vocab_size) exactly as get_tokenizer's custom/pinyin branch (utils.py:112-149)."""
    with open(path, "r", encoding="utf-8") as f:
        m = {}
        for i, ch in enumerate(f):
            m[ch[:-1]] = i
    return m, len(m)


def configure_host_threads(n: int = 1) -> None:
    """The engine's host side is a handful of tiny CPU tensor ops per call (seeded noise draw, pads, masks).  torch's default
    intra-op pool has one thread per logical CPU of the HOST, not of the job's share, so on a shared or multi-rank node
    those ops wake an oversubscribed OpenMP team and its spinning workers starve the HIP runtime's submission thread:
    occasional 100-200 ms utterances instead of 34 ms (tools/stall_probe.py).  Serving loops should call this once."""
    import torch

    torch.set_num_threads(n)

