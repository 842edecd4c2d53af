"""TEST INFRASTRUCTURE ONLY -- CPU (PyTorch fp32) restatement of the reference's F5-TTS inference hot path.

This file is the *checker*, never the product: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
may import it.  The product path (korean-f5-tts_amd/) must not and does not; it fails loudly when its HIP library is
missing.

What it restates (all citations relative to /root/reference/src/f5_tts/):
  sample()            model/cfm.py:83-229            (prologue, noise, time grid, CFG closure, Euler, epilogue)
  euler_odeint()      third-party torchdiffeq, fixed-grid Euler; pinned in-repo by
                      runtime/triton_trtllm/model_repo_f5_tts/f5_tts/1/f5_tts_trtllm.py:248-250,360-369
  dit_forward()       model/backbones/dit.py:278-329 (+ get_input_embed :234-273)
  text_embed()        model/backbones/dit.py:86-115, model/modules.py:202-213 (pos table)
  convnext_v2_block() model/modules.py:247-275, GRN :231-240
  input_embed()       model/backbones/dit.py:127-140
  conv_pos_embed()    model/modules.py:170-196
  time_embed()        model/modules.py:152-164,777-787
  rotary_*()          third-party x_transformers; pinned in-repo by f5_tts_trtllm.py:230-237 and
                      runtime/triton_trtllm/patch/f5tts/modules.py:210-276
  attention()         model/modules.py:459-544 (torch backend)
  dit_block()         model/modules.py:307-321,348-359,683-697
  final_layer()       model/modules.py:328-342, dit.py:326-327
  lens_to_mask()      model/utils.py:53-58;  epss_timesteps() model/utils.py:538-551
  unett_forward()     model/backbones/unett.py:36-280 (x_transformers.RMSNorm: parity unpinned)
  vocos_decode()      third-party `vocos` (not in the tree): ISTFT-head arithmetic pinned by
                      runtime/triton_trtllm/scripts/export_vocoder_to_onnx.py:45-59; iSTFT checked against
                      torch.istft; the ConvNeXt backbone is restated from the published Vocos architecture --
                      PARITY UNPINNED for the backbone (no source, weights or vectors in the container).

Pinning status: every function above except vocos backbone / x_transformers.RMSNorm is pinned against the reference
itself, imported in the build container by oracle/ref_harness.py (tests/test_oracle_vs_reference.py, live) and through
the committed vectors in tests/golden/ produced by oracle/make_golden.py from that same import.

Weights are a flat {name: tensor} dict that uses the reference's state-dict names without the leading
"transformer." (e.g. "transformer_blocks.3.attn.to_q.weight"); `cfg` is a plain dict with the reference's arch keys.
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F

# ------------------------------------------------------------------------------------------------ helpers


def lens_to_mask(lens: torch.Tensor, length: int | None = None) -> torch.Tensor:
    if length is None:
        length = int(lens.amax())
    return torch.arange(length)[None, :] < lens[:, None]


_EPSS = {
    5: [0, 2, 4, 8, 16, 32],
    6: [0, 2, 4, 6, 8, 16, 32],
    7: [0, 2, 4, 6, 8, 16, 24, 32],
    10: [0, 2, 4, 6, 8, 12, 16, 20, 24, 28, 32],
    12: [0, 2, 4, 6, 8, 10, 12, 14, 16, 20, 24, 28, 32],
    16: [0, 1, 2, 3, 4, 5, 6, 7, 8, 10, 12, 14, 16, 20, 24, 28, 32],
}


def epss_timesteps(n: int, dtype=torch.float32) -> torch.Tensor:
    tab = _EPSS.get(n)
    if not tab:
        return torch.linspace(0, 1, n + 1, dtype=dtype)
    return (1 / 32) * torch.tensor(tab, dtype=dtype)


def time_grid(steps: int, sway_sampling_coef=None, use_epss=True, t_start=0.0, dtype=torch.float32) -> torch.Tensor:
    """cfm.py:203-216."""
    if t_start == 0 and use_epss:
        t = epss_timesteps(steps, dtype)
    else:
        t = torch.linspace(t_start, 1, steps + 1, dtype=dtype)
    if sway_sampling_coef is not None:
        t = t + sway_sampling_coef * (torch.cos(torch.pi / 2 * t) - 1 + t)
    return t


def linear(x, W, name):
    return F.linear(x, W[name + ".weight"], W.get(name + ".bias"))


# ------------------------------------------------------------------------------------------- time embedding


def sinus_features(t: torch.Tensor, dim=256, scale=1000.0) -> torch.Tensor:
    half = dim // 2
    k = math.log(10000) / (half - 1)
    freqs = torch.exp(torch.arange(half).float() * -k)
    arg = scale * t.unsqueeze(1) * freqs.unsqueeze(0)
    return torch.cat((arg.sin(), arg.cos()), dim=-1)


def time_embed(W, t: torch.Tensor) -> torch.Tensor:
    """t: f32[b] -> f32[b, dim]."""
    h = sinus_features(t).to(t.dtype)
    h = linear(h, W, "time_embed.time_mlp.0")
    h = F.silu(h)
    return linear(h, W, "time_embed.time_mlp.2")


# -------------------------------------------------------------------------------------------- text encoder


def text_pos_table(dim: int, end: int, theta=10000.0) -> torch.Tensor:
    freqs = 1.0 / (theta ** (torch.arange(0, dim, 2)[: dim // 2].float() / dim))
    ang = torch.outer(torch.arange(end), freqs).float()
    return torch.cat([ang.cos(), ang.sin()], dim=-1)


def grn(x, gamma, beta):
    gx = torch.norm(x, p=2, dim=1, keepdim=True)  # over the SEQUENCE axis
    nx = gx / (gx.mean(dim=-1, keepdim=True) + 1e-6)
    return gamma * (x * nx) + beta + x


def convnext_v2_block(W, pfx, x):
    dim = x.shape[-1]
    h = F.conv1d(x.transpose(1, 2), W[pfx + ".dwconv.weight"], W[pfx + ".dwconv.bias"], padding=3, groups=dim)
    h = h.transpose(1, 2)
    h = F.layer_norm(h, (dim,), W[pfx + ".norm.weight"], W[pfx + ".norm.bias"], eps=1e-6)
    h = linear(h, W, pfx + ".pwconv1")
    h = F.gelu(h)  # exact (erf)
    h = grn(h, W[pfx + ".grn.gamma"], W[pfx + ".grn.beta"])
    h = linear(h, W, pfx + ".pwconv2")
    return x + h


def text_embed(W, cfg, text: torch.Tensor, seq_len: int, drop_text=False) -> torch.Tensor:
    """text: i64[b, nt] padded with -1 -> f32[b, seq_len, text_dim]   (dit.py:86-115)."""
    pfx = "text_embed"
    ids = text + 1
    ids = ids[:, :seq_len]
    ids = F.pad(ids, (0, seq_len - ids.shape[1]), value=0)
    mask_padding = cfg.get("text_mask_padding", True)
    pad = ids == 0
    if drop_text:
        ids = torch.zeros_like(ids)
    h = F.embedding(ids, W[pfx + ".text_embed.weight"])
    n_conv = cfg.get("conv_layers", 0)
    if n_conv > 0:
        h = h + text_pos_table(h.shape[-1], 8192)[:seq_len]
        if mask_padding:
            h = h.masked_fill(pad.unsqueeze(-1), 0.0)
        for i in range(n_conv):
            h = convnext_v2_block(W, f"{pfx}.text_blocks.{i}", h)
            if mask_padding:
                h = h.masked_fill(pad.unsqueeze(-1), 0.0)
    if cfg.get("text_embedding_average_upsampling", False):
        assert mask_padding, "text_embedding_average_upsampling requires text_mask_padding to be True"   # dit.py:41-42
        h = average_upsample_text_by_mask(h, ~pad)
    return h


def average_upsample_text_by_mask(text: torch.Tensor, text_mask: torch.Tensor) -> torch.Tensor:
    """dit.py:54-84 (zipvoice-style late upsampling): the valid tokens of every sample are repeated to fill the audio length,
    the LAST `remainder` tokens once more than the others; a sample without valid tokens stays zero."""
    b, audio_len, _ = text.shape
    out = torch.zeros_like(text)
    for i in range(b):
        valid = torch.where(text_mask[i])[0]
        tl = int(valid.numel())
        if tl == 0:
            continue
        base, rem = audio_len // tl, audio_len % tl
        idx = []
        for j in range(tl):
            idx.extend([j] * (base + (1 if j >= tl - rem else 0)))
        out[i, :audio_len] = text[i, valid][torch.tensor(idx[:audio_len], dtype=torch.long)]
    return out


def text_embed_batch(W, cfg, text, seq_len, mask, drop_text):
    """dit.py:244-258 -- with an audio mask every sample is embedded at its OWN length, then zero padded."""
    if mask is None:
        return text_embed(W, cfg, text, seq_len, drop_text)
    lens = mask.sum(dim=1)
    outs = []
    for i in range(text.shape[0]):
        e = text_embed(W, cfg, text[i : i + 1], int(lens[i]), drop_text)[0]
        outs.append(F.pad(e, (0, 0, 0, seq_len - e.shape[0])))
    return torch.stack(outs, 0)


# ------------------------------------------------------------------------------------------ input embedding


def conv_pos_embed(W, pfx, x, mask=None, groups=16):
    """modules.py:170-196: two grouped Conv1d(k=31) + Mish, masked after each conv when a mask is given."""
    h = x.permute(0, 2, 1)
    m = None
    if mask is not None:
        m = mask.unsqueeze(1)
        h = h.masked_fill(~m, 0.0)
    for idx in (0, 2):
        h = F.conv1d(h, W[f"{pfx}.conv1d.{idx}.weight"], W[f"{pfx}.conv1d.{idx}.bias"], padding=15, groups=groups)
        if m is not None:
            h = h.masked_fill(~m, 0.0)
        h = F.mish(h)
    return h.permute(0, 2, 1)


def input_embed(W, x, cond, text_emb, drop_audio_cond=False, mask=None):
    if drop_audio_cond:
        cond = torch.zeros_like(cond)
    h = linear(torch.cat((x, cond, text_emb), dim=-1), W, "input_embed.proj")
    return conv_pos_embed(W, "input_embed.conv_pos_embed", h, mask) + h


# ---------------------------------------------------------------------------------------------- attention


def rotary_freqs(n: int, dim_head: int, base=10000.0) -> torch.Tensor:
    inv = 1.0 / (base ** (torch.arange(0, dim_head, 2).float() / dim_head))
    f = torch.outer(torch.arange(n).float(), inv)
    return f.repeat_interleave(2, dim=-1)  # [n, dim_head], interleaved pairs share a frequency


def rotary_apply(x: torch.Tensor, freqs: torch.Tensor) -> torch.Tensor:
    """x: [b, h, n, d]; pair (a, b) at (2j, 2j+1) -> (a cos - b sin, b cos + a sin)."""
    xr = x.reshape(*x.shape[:-1], -1, 2)
    rot = torch.stack((-xr[..., 1], xr[..., 0]), dim=-1).reshape(x.shape)
    return x * freqs.cos() + rot * freqs.sin()


def attention(W, cfg, pfx, x, mask, freqs):
    b, n, _ = x.shape
    H = cfg["heads"]
    dh = cfg.get("dim_head", 64)
    q = linear(x, W, pfx + ".to_q").view(b, n, H, dh).transpose(1, 2)
    k = linear(x, W, pfx + ".to_k").view(b, n, H, dh).transpose(1, 2)
    v = linear(x, W, pfx + ".to_v").view(b, n, H, dh).transpose(1, 2)
    qk_norm = cfg.get("qk_norm")
    if qk_norm is not None:   # modules.py:397-404,481-484 (RMSNorm over dim_head, eps 1e-6, before the rotary embedding)
        assert qk_norm == "rms_norm", f"Unimplemented qk_norm: {qk_norm}"
        q = F.rms_norm(q, (dh,), weight=W[pfx + ".q_norm.weight"], eps=1e-6)
        k = F.rms_norm(k, (dh,), weight=W[pfx + ".k_norm.weight"], eps=1e-6)
    pn = cfg.get("pe_attn_head")
    if pn is None:
        q, k = rotary_apply(q, freqs), rotary_apply(k, freqs)
    else:
        q = torch.cat((rotary_apply(q[:, :pn], freqs), q[:, pn:]), dim=1)
        k = torch.cat((rotary_apply(k[:, :pn], freqs), k[:, pn:]), dim=1)
    s = torch.matmul(q, k.transpose(-1, -2)) * (dh**-0.5)
    if cfg.get("attn_mask_enabled", False) and mask is not None:
        s = s.masked_fill(~mask[:, None, None, :], float("-inf"))
    o = torch.matmul(torch.softmax(s, dim=-1), v)
    o = o.transpose(1, 2).reshape(b, n, H * dh)
    o = linear(o, W, pfx + ".to_out.0")
    if mask is not None:
        o = o.masked_fill(~mask.unsqueeze(-1), 0.0)
    return o


def dit_block(W, cfg, i, x, t, mask, freqs):
    pfx = f"transformer_blocks.{i}"
    dim = x.shape[-1]
    emb = linear(F.silu(t), W, pfx + ".attn_norm.linear")
    shift_msa, scale_msa, gate_msa, shift_mlp, scale_mlp, gate_mlp = torch.chunk(emb, 6, dim=1)
    h = F.layer_norm(x, (dim,), eps=1e-6) * (1 + scale_msa[:, None]) + shift_msa[:, None]
    x = x + gate_msa.unsqueeze(1) * attention(W, cfg, pfx + ".attn", h, mask, freqs)
    h = F.layer_norm(x, (dim,), eps=1e-6) * (1 + scale_mlp[:, None]) + shift_mlp[:, None]
    h = F.gelu(linear(h, W, pfx + ".ff.ff.0.0"), approximate="tanh")
    h = linear(h, W, pfx + ".ff.ff.2")
    return x + gate_mlp.unsqueeze(1) * h


def final_layer(W, x, t):
    dim = x.shape[-1]
    emb = linear(F.silu(t), W, "norm_out.linear")
    scale, shift = torch.chunk(emb, 2, dim=1)
    h = F.layer_norm(x, (dim,), eps=1e-6) * (1 + scale)[:, None, :] + shift[:, None, :]
    return linear(h, W, "proj_out")


# ------------------------------------------------------------------------------------------------- DiT


class TextCache:
    def __init__(self):
        self.cond = None
        self.uncond = None


def dit_forward(W, cfg, x, cond, text, time, mask=None, drop_audio_cond=False, drop_text=False, cfg_infer=False,
                cache: TextCache | None = None, taps: dict | None = None):
    """dit.py:278-329.  `taps`, when given, collects intermediates for kernel-level parity tests."""
    b, n = x.shape[:2]
    if time.ndim == 0:
        time = time.repeat(b)
    t = time_embed(W, time)

    def embed(drop_a, drop_t):
        te = None
        if cache is not None:
            te = cache.uncond if drop_t else cache.cond
        if te is None:
            te = text_embed_batch(W, cfg, text, n, mask, drop_t)
            if cache is not None:
                if drop_t:
                    cache.uncond = te
                else:
                    cache.cond = te
        if taps is not None:
            taps["text_uncond" if drop_t else "text_cond"] = te
        return input_embed(W, x, cond, te, drop_a, mask)

    if cfg_infer:
        h = torch.cat((embed(False, False), embed(True, True)), dim=0)
        t = torch.cat((t, t), dim=0)
        mask = torch.cat((mask, mask), dim=0) if mask is not None else None
    else:
        h = embed(drop_audio_cond, drop_text)
    if taps is not None:
        taps["time_embed"] = t
        taps["input_embed"] = h
    freqs = rotary_freqs(n, cfg.get("dim_head", 64))
    residual = h if cfg.get("long_skip_connection", False) else None      # dit.py:313-314
    for i in range(cfg["depth"]):
        h = dit_block(W, cfg, i, h, t, mask, freqs)
        if taps is not None:
            taps[f"block{i}"] = h
    if residual is not None:                                              # dit.py:323-324 (Linear(2 dim -> dim), no bias)
        h = F.linear(torch.cat((h, residual), dim=-1), W["long_skip_connection.weight"])
    return final_layer(W, h, t)


# ------------------------------------------------------------------------------------------------ UNetT


def x_rmsnorm(x, g):
    """x_transformers.RMSNorm (third party, not in tree; parity unpinned): normalize(x) * sqrt(dim) * g."""
    return F.normalize(x, dim=-1) * (x.shape[-1] ** 0.5) * g


def unett_text_embed(W, cfg, text, seq_len, drop_text=False):
    """unett.py:64-83."""
    ids = text + 1
    ids = ids[:, :seq_len]
    ids = F.pad(ids, (0, seq_len - ids.shape[1]), value=0)
    pad = ids == 0
    if drop_text:
        ids = torch.zeros_like(ids)
    h = F.embedding(ids, W["text_embed.text_embed.weight"])
    n_conv = cfg.get("conv_layers", 0)
    if n_conv > 0:
        h = h + text_pos_table(h.shape[-1], 4096)[torch.arange(seq_len).clamp(max=4095)]  # unett.py:66-70
        mp = cfg.get("text_mask_padding", True)
        if mp:
            h = h.masked_fill(pad.unsqueeze(-1), 0.0)
        for i in range(n_conv):
            h = convnext_v2_block(W, f"text_embed.text_blocks.{i}", h)
            if mp:
                h = h.masked_fill(pad.unsqueeze(-1), 0.0)
    return h


def unett_forward(W, cfg, x, cond, text, time, mask=None, drop_audio_cond=False, drop_text=False, cfg_infer=False,
                  cache: TextCache | None = None):
    """unett.py:217-280."""
    b, n = x.shape[:2]
    if time.ndim == 0:
        time = time.repeat(b)
    t = time_embed(W, time)

    def embed(drop_a, drop_t):
        te = None
        if cache is not None:
            te = cache.uncond if drop_t else cache.cond
        if te is None:
            te = unett_text_embed(W, cfg, text, n, drop_t)
            if cache is not None:
                if drop_t:
                    cache.uncond = te
                else:
                    cache.cond = te
        c = torch.zeros_like(cond) if drop_a else cond
        h = linear(torch.cat((x, c, te), dim=-1), W, "input_embed.proj")
        return conv_pos_embed(W, "input_embed.conv_pos_embed", h, None) + h  # unett.py:99 passes no mask

    if cfg_infer:
        h = torch.cat((embed(False, False), embed(True, True)), dim=0)
        t = torch.cat((t, t), dim=0)
        mask = torch.cat((mask, mask), dim=0) if mask is not None else None
    else:
        h = embed(drop_audio_cond, drop_text)
    bb = h.shape[0]
    h = torch.cat([t.unsqueeze(1), h], dim=1)
    if mask is not None:
        mask = F.pad(mask, (1, 0), value=True)
    freqs = rotary_freqs(n + 1, cfg.get("dim_head", 64))
    depth = cfg["depth"]
    skips = []
    skip_type = cfg.get("skip_connect_type", "concat")
    for i in range(depth):
        pfx = f"layers.{i}"  # ModuleList([skip_proj, attn_norm, attn, ff_norm, ff]) at unett.py:171-183
        if i < depth // 2:
            skips.append(h)
        else:
            s = skips.pop()
            if skip_type == "concat":
                h = F.linear(torch.cat((h, s), dim=-1), W[f"{pfx}.0.weight"])
            elif skip_type == "add":
                h = h + s
        h = attention(W, cfg, f"{pfx}.2", x_rmsnorm(h, W[f"{pfx}.1.g"]), mask, freqs) + h
        f = F.gelu(linear(x_rmsnorm(h, W[f"{pfx}.3.g"]), W, f"{pfx}.4.ff.0.0"), approximate="tanh")
        h = linear(f, W, f"{pfx}.4.ff.2") + h
    assert len(skips) == 0 and bb == h.shape[0]
    h = x_rmsnorm(h, W["norm_out.g"])[:, 1:, :]
    return linear(h, W, "proj_out")


# ----------------------------------------------------------------------------------------------- sampler


def euler_odeint(fn, y0, t):
    ys = [y0]
    y = y0
    for i in range(t.shape[0] - 1):
        y = y + (t[i + 1] - t[i]) * fn(t[i], y)
        ys.append(y)
    return torch.stack(ys, 0)


def sample(W, cfg, cond, text, duration, *, lens=None, steps=32, cfg_strength=1.0, sway_sampling_coef=None,
           seed=None, max_duration=65536, use_epss=True, no_ref_audio=False, edit_mask=None, backbone="DiT",
           y0=None, duplicate_test=False, t_inter=0.1):
    """cfm.py:83-229 for mel-in / mel-out (cond f32[b, n, mel]; text i64[b, nt] padded -1).

    `y0` overrides the noise draw (cfm.py:196-201) so identical noise can be fed to the HIP path and to this oracle.
    Returns (out, trajectory) exactly as the reference does."""
    fwd = dit_forward if backbone == "DiT" else unett_forward
    cond = cond.to(torch.float32)
    batch, cond_seq_len = cond.shape[:2]
    if lens is None:
        lens = torch.full((batch,), cond_seq_len, dtype=torch.long)
    cond_mask = lens_to_mask(lens)
    if edit_mask is not None:
        cond_mask = cond_mask & edit_mask
    if isinstance(duration, int):
        duration = torch.full((batch,), duration, dtype=torch.long)
    duration = torch.maximum(torch.maximum((text != -1).sum(dim=-1), lens) + 1, duration)
    duration = duration.clamp(max=max_duration)
    n = int(duration.amax())
    if duplicate_test:  # cfm.py:141-143
        test_cond = F.pad(cond, (0, 0, cond_seq_len, n - 2 * cond_seq_len), value=0.0)
    cond = F.pad(cond, (0, 0, 0, n - cond_seq_len), value=0.0)
    if no_ref_audio:
        cond = torch.zeros_like(cond)
    cond_mask = F.pad(cond_mask, (0, n - cond_mask.shape[-1]), value=False).unsqueeze(-1)
    step_cond = torch.where(cond_mask, cond, torch.zeros_like(cond))
    mask = lens_to_mask(duration) if batch > 1 else None
    cache = TextCache()

    def fn(t, x):
        if cfg_strength < 1e-5:
            return fwd(W, cfg, x, step_cond, text, t, mask, False, False, False, cache)
        p = fwd(W, cfg, x, step_cond, text, t, mask, cfg_infer=True, cache=cache)
        pred, null = torch.chunk(p, 2, dim=0)
        return pred + (pred - null) * cfg_strength

    if y0 is None:
        y0 = draw_noise(duration, cond.shape[-1], seed)
    t_start = 0.0
    if duplicate_test:  # cfm.py:203-208: start the solve at t_inter from a blend of noise and the prompt shifted by its length
        t_start = t_inter
        y0 = (1 - t_start) * y0 + t_start * test_cond
        steps = int(steps * (1 - t_start))
    t = time_grid(steps, sway_sampling_coef, use_epss, t_start)
    traj = euler_odeint(fn, y0, t)
    out = torch.where(cond_mask, cond, traj[-1])
    return out, traj


def draw_noise(duration: torch.Tensor, mel_dim: int, seed) -> torch.Tensor:
    """cfm.py:196-201 on the CPU generator: per sample `manual_seed(seed)` then randn(dur, mel); zero padded."""
    ys = []
    for d in duration.tolist():
        if seed is not None:
            torch.manual_seed(seed)
        ys.append(torch.randn(d, mel_dim))
    return torch.nn.utils.rnn.pad_sequence(ys, padding_value=0, batch_first=True)


# -------------------------------------------------------------------------------------------------- Vocos


def vocos_backbone(V, mel):
    """mel f32[b, 100, T] -> f32[b, T, 512].  Restated from the published Vocos architecture (parity unpinned)."""
    dim = V["backbone.embed.weight"].shape[0]
    h = F.conv1d(mel, V["backbone.embed.weight"], V["backbone.embed.bias"], padding=3)
    h = F.layer_norm(h.transpose(1, 2), (dim,), V["backbone.norm.weight"], V["backbone.norm.bias"], eps=1e-6)
    i = 0
    while f"backbone.convnext.{i}.dwconv.weight" in V:
        p = f"backbone.convnext.{i}"
        r = h
        g = F.conv1d(h.transpose(1, 2), V[p + ".dwconv.weight"], V[p + ".dwconv.bias"], padding=3, groups=dim)
        g = F.layer_norm(g.transpose(1, 2), (dim,), V[p + ".norm.weight"], V[p + ".norm.bias"], eps=1e-6)
        g = F.gelu(linear(g, V, p + ".pwconv1"))
        g = linear(g, V, p + ".pwconv2")
        h = r + V[p + ".gamma"] * g
        i += 1
    return F.layer_norm(h, (dim,), V["backbone.final_layer_norm.weight"], V["backbone.final_layer_norm.bias"],
                        eps=1e-6)


def istft_head_spec(V, h):
    """export_vocoder_to_onnx.py:51-59: Linear -> (mag, phase) -> exp, clip 1e2, cos/sin.  Returns (re, im) [b,513,T]."""
    x = linear(h, V, "head.out").transpose(1, 2)
    mag, p = x.chunk(2, dim=1)
    mag = torch.clip(torch.exp(mag), max=1e2)
    return mag * torch.cos(p), mag * torch.sin(p)


def vocos_decode(V, mel, n_fft=1024, hop=256):
    """mel f32[b, 100, T] -> wav f32[b, (T-1)*hop]  (Vocos ISTFTHead, padding="center" -> torch.istft center=True)."""
    re, im = istft_head_spec(V, vocos_backbone(V, mel))
    spec = torch.complex(re, im)
    return torch.istft(spec, n_fft, hop, n_fft, window=torch.hann_window(n_fft), center=True)


# ----------------------------------------------------------------------------------------- mel front-end


def htk_mel_filterbank(n_freqs, n_mels, sample_rate, f_min=0.0, f_max=None):
    """torchaudio.functional.melscale_fbanks(norm=None, mel_scale="htk") restated from its published definition
    (torchaudio is absent from the container: PARITY UNPINNED)."""
    f_max = float(sample_rate // 2) if f_max is None else f_max
    all_freqs = torch.linspace(0, sample_rate // 2, n_freqs)
    m_min = 2595.0 * math.log10(1.0 + f_min / 700.0)
    m_max = 2595.0 * math.log10(1.0 + f_max / 700.0)
    m_pts = torch.linspace(m_min, m_max, n_mels + 2)
    f_pts = 700.0 * (10 ** (m_pts / 2595.0) - 1.0)
    f_diff = f_pts[1:] - f_pts[:-1]
    slopes = f_pts.unsqueeze(0) - all_freqs.unsqueeze(1)
    down = (-1.0 * slopes[:, :-2]) / f_diff[:-1]
    up = slopes[:, 2:] / f_diff[1:]
    return torch.clamp(torch.min(down, up), min=0.0)  # [n_freqs, n_mels]


def mel_spectrogram_vocos(wav, n_fft=1024, hop=256, n_mels=100, sample_rate=24000):
    """modules.py:78-104: MelSpectrogram(power=1, center=True, norm=None) -> clamp(min=1e-5).log(); wav [b, nw] -> [b, n_mels, T]."""
    spec = torch.stft(wav, n_fft, hop_length=hop, win_length=n_fft, window=torch.hann_window(n_fft), center=True,
                      pad_mode="reflect", normalized=False, onesided=True, return_complex=True).abs()
    fb = htk_mel_filterbank(n_fft // 2 + 1, n_mels, sample_rate)
    mel = torch.matmul(spec.transpose(1, 2), fb).transpose(1, 2)
    return mel.clamp(min=1e-5).log()
