"""Pins for the arithmetic-bearing third-party stand-ins (CPU).

x_transformers (rotary) and torchdiffeq (Euler) are absent from the container, so the harness that runs the reference
(oracle/ref_harness.py) and the oracle (oracle/f5_oracle.py) restate them.  Every golden fixture passes through those
restatements.  The reference itself holds an INDEPENDENT restatement of the same arithmetic in its TensorRT-LLM runtime;
this file transcribes those lines into numpy (float64 index / concat arithmetic only, no torch helpers shared with the
code under test) and checks that the stand-ins and the oracle agree with them on random tensors:

  rotary angles     runtime/triton_trtllm/model_repo_f5_tts/f5_tts/1/f5_tts_trtllm.py:230-237
                    (inv_freq = 1 / base^(arange(0, d, 2) / d); freqs = outer(arange(n), inv_freq).repeat_interleave(2, -1))
  rotate-every-two  runtime/triton_trtllm/patch/f5tts/modules.py:210-238 (x1 = x[..., 0::2], x2 = x[..., 1::2],
                    out = interleave(-x2, x1)) and :241-276 (x * cos + rotate(x) * sin on the first pe_attn_head heads only,
                    the remaining channels passed through)
  EPSS + sway grid  f5_tts_trtllm.py:240-250 (t = table / 32; time_step = 1 - cos(pi t / 2); delta_t = diff(time_step))
  Euler + CFG       f5_tts_trtllm.py:360-369 (guidance = cond + (cond - uncond) * cfg; noise += guidance * delta_t[i])
  time features     f5_tts_trtllm.py:252-260 (1000 * t * exp(-k ln(1e4) / 127), cat(sin, cos))
"""
import math

import numpy as np
import torch

from oracle import f5_oracle as O
from oracle import ref_harness as H

import f5_tts_amd as P


# ------------------------------------------------------------------------------- transcriptions (numpy)
def trt_freqs(n, head_dim=64):
    base = 10000.0 * 1.0 ** (head_dim / (head_dim - 2))                       # f5_tts_trtllm.py:232 (rescale factor 1)
    inv_freq = 1.0 / (base ** (np.arange(0, head_dim, 2, dtype=np.float32) / head_dim))          # :233
    freqs = np.outer(np.arange(n, dtype=np.float32), inv_freq.astype(np.float32)).astype(np.float32) / 1.0   # :234
    return np.repeat(freqs, 2, axis=-1)                                        # :235 repeat_interleave(2, dim=-1)


def trt_rotate_every_two(x):
    x1 = x[..., 0::2]                                                           # modules.py:228 slice stride 2 from 0
    x2 = x[..., 1::2]                                                           # :229 from 1
    out = np.stack([-x2, x1], axis=-1)                                          # :230-234 concat([0 - x2, x1], last)
    return out.reshape(x.shape)                                                 # :235 view


def trt_apply_rotary(x, cos, sin, pe_attn_head):
    """modules.py:241-276 for [B, N, D] input: heads are 64-wide slices of the last axis."""
    full = x.shape[-1]
    hd = cos.shape[-1]
    heads = full // hd if pe_attn_head is None else pe_attn_head
    parts = []
    for i in range(heads):
        xs = x[..., i * hd:(i + 1) * hd]
        parts.append(xs * cos + trt_rotate_every_two(xs) * sin)
    parts.append(x[..., heads * hd:])
    return np.concatenate(parts, axis=-1)


EPSS = {5: [0, 2, 4, 8, 16, 32], 6: [0, 2, 4, 6, 8, 16, 32], 7: [0, 2, 4, 6, 8, 16, 24, 32],
        10: [0, 2, 4, 6, 8, 12, 16, 20, 24, 28, 32], 12: [0, 2, 4, 6, 8, 10, 12, 14, 16, 20, 24, 28, 32],
        16: [0, 1, 2, 3, 4, 5, 6, 7, 8, 10, 12, 14, 16, 20, 24, 28, 32]}     # f5_tts_trtllm.py:240-247


def trt_time_steps(nfe):
    t = (1 / 32 * np.array(EPSS.get(nfe, list(range(nfe + 1))), dtype=np.float32)).astype(np.float32)   # :248
    return (1 - np.cos(np.float32(np.pi) * t / 2)).astype(np.float32)                                   # :249


# ----------------------------------------------------------------------------------------------- tests
def test_rotary_stand_ins_match_the_trt_restatement():
    rng = np.random.default_rng(0)
    for n, H_, pe in ((50, 4, 1), (129, 3, None), (7, 16, 1)):
        x = rng.standard_normal((2, n, H_ * 64)).astype(np.float32)
        fr = trt_freqs(n)
        want = trt_apply_rotary(x, np.cos(fr)[None], np.sin(fr)[None], pe)
        xt = torch.from_numpy(x).view(2, n, H_, 64).transpose(1, 2)             # [B, H, N, 64] as modules.py:476-478
        heads = H_ if pe is None else pe
        # (1) the x_transformers stand-in the reference runs on when the fixtures are generated
        freqs, scale = H._RotaryEmbedding(64).forward_from_seq_len(n)
        got = xt.clone()
        got[:, :heads] = H._apply_rotary_pos_emb(xt[:, :heads], freqs, scale)
        got = got.transpose(1, 2).reshape(2, n, H_ * 64).numpy()
        # (tolerance: the angles reach n * inv_freq_0 = n rad in f32, so library differences of one ulp in pow / cos / sin show
        #  up at ~1e-5; a structural difference -- pairing, sign, head slicing -- is O(1))
        assert np.abs(got - want).max() < 1e-4
        # (2) the oracle's own rotary
        fo = O.rotary_freqs(n, 64)
        got_o = xt.clone()
        got_o[:, :heads] = O.rotary_apply(xt[:, :heads], fo)
        got_o = got_o.transpose(1, 2).reshape(2, n, H_ * 64).numpy()
        assert np.abs(got_o - want).max() < 1e-4
    # (3) the table the product uploads to the engine (engine.py::aux_tables) holds cos / sin of the SAME angles, one per pair
    tabs = P.engine.aux_tables(64, 300, 512, 0)
    fr = trt_freqs(300)
    assert np.abs(tabs["aux.rope_cos"].numpy() - np.cos(fr)[:, 0::2]).max() < 1e-4
    assert np.abs(tabs["aux.rope_sin"].numpy() - np.sin(fr)[:, 0::2]).max() < 1e-4


def test_time_grid_matches_the_trt_restatement():
    # (the TRT fallback `list(range(nfe + 1)) / 32` is a uniform grid only for its hard-coded nfe = 32)
    for nfe in (5, 6, 7, 10, 12, 16, 32):
        want = trt_time_steps(nfe)
        got_o = O.time_grid(nfe, sway_sampling_coef=-1.0, use_epss=True).numpy()
        assert got_o.shape == want.shape and np.abs(got_o - want).max() < 2e-7, nfe
        # the product's grid: utils.get_epss_timesteps + the sway formula of cfm.py:216 as cfm.py evaluates it
        t = P.utils.get_epss_timesteps(nfe, device="cpu", dtype=torch.float32) if nfe in EPSS else torch.linspace(0, 1, nfe + 1)
        t = t + (-1.0) * (torch.cos(torch.pi / 2 * t) - 1 + t)
        assert np.abs(t.numpy() - want).max() < 2e-7, nfe


def test_euler_stand_ins_match_the_trt_update():
    """noise += (cond + (cond - uncond) * cfg) * delta_t[i]  (f5_tts_trtllm.py:360-369) against the torchdiffeq stand-in
    driven by the reference's closure shape (cfm.py:162-191) and against the oracle's solver."""
    rng = np.random.default_rng(1)
    y0 = rng.standard_normal((2, 9, 5)).astype(np.float32)
    A = rng.standard_normal((5, 5)).astype(np.float32) * 0.3
    Bm = rng.standard_normal((5, 5)).astype(np.float32) * 0.3
    cfg = 2.0
    ts = trt_time_steps(16)
    delta_t = np.diff(ts)                                                        # :250
    noise = y0.copy()
    want = [noise.copy()]
    for i in range(16):
        cond = np.tanh(noise @ A) * (1 + ts[i])
        uncond = np.tanh(noise @ Bm) - ts[i]
        guidance = cond + (cond - uncond) * cfg                                  # :365
        noise = noise + guidance * delta_t[i]                                    # :367
        want.append(noise.copy())
    want = np.stack(want, 0)

    def fn(t, x):
        c = torch.tanh(x @ torch.from_numpy(A)) * (1 + t)
        u = torch.tanh(x @ torch.from_numpy(Bm)) - t
        return c + (c - u) * cfg

    tt = torch.from_numpy(ts)
    got_h = H._odeint(fn, torch.from_numpy(y0), tt, method="euler").numpy()
    got_o = O.euler_odeint(fn, torch.from_numpy(y0), tt).numpy()
    assert np.abs(got_h - want).max() < 5e-6
    assert np.abs(got_o - want).max() < 5e-6


def test_time_features_match_the_trt_restatement():
    ts = trt_time_steps(16)
    half = 128
    emb_factor = math.log(10000) / (half - 1)                                    # :255
    emb_factor = 1000.0 * np.exp(np.arange(half, dtype=np.float32) * np.float32(-emb_factor))   # :256
    want = np.stack([np.concatenate([np.sin(ts[i] * emb_factor), np.cos(ts[i] * emb_factor)]) for i in range(16)])   # :257-259
    got = O.sinus_features(torch.from_numpy(ts[:16])).numpy()
    # arguments reach 1000 rad: sin / cos of an f32 argument move by ~ulp(1000) = 6e-5 when the product is rounded in a
    # different order (1000 * (t * e) vs (1000 * e) * t)
    assert np.abs(got - want).max() < 3e-4
    tabs = P.engine.aux_tables(64, 8, 512, 0)
    assert np.abs(tabs["aux.time_freqs"].numpy() * 1000.0 - emb_factor).max() < 1e-3 * 1e-3 * 1000
