"""State-dict layouts (the reference's parameter names) and the deterministic synthetic-weight generator.

Names follow the reference's modules so real checkpoints load unchanged: src/f5_tts/model/backbones/dit.py:146-212,
model/modules.py (TimestepEmbedding :777-787, ConvNeXtV2Block :247-275, ConvPositionEmbedding :170-196, AdaLayerNorm
:307-321, Attention :366-435, FeedForward :348-359), backbones/unett.py:107-186; names are also listed at
runtime/triton_trtllm/scripts/convert_checkpoint.py:129-145.

The synthetic generator overwrites EVERY tensor, including the ones the reference zero-initialises
(dit.py:214-224: AdaLN linears and proj_out; modules.py:234-235: GRN gamma/beta), otherwise a fresh model predicts
exactly zero and every parity test would be vacuous (SURVEY.md section 7 "Zero-init trap").
"""
from __future__ import annotations

import zlib

import torch

from .config import normalize_arch


def dit_param_shapes(arch: dict, text_num_embeds: int, mel_dim: int = 100) -> dict[str, tuple]:
    a = normalize_arch(arch, mel_dim)
    D, Dt, F = a["dim"], a["text_dim"], int(a["dim"] * a["ff_mult"])
    inner = a["heads"] * a["dim_head"]
    s: dict[str, tuple] = {}
    s["time_embed.time_mlp.0.weight"] = (D, 256)
    s["time_embed.time_mlp.0.bias"] = (D,)
    s["time_embed.time_mlp.2.weight"] = (D, D)
    s["time_embed.time_mlp.2.bias"] = (D,)
    s["text_embed.text_embed.weight"] = (text_num_embeds + 1, Dt)
    for i in range(a["conv_layers"]):
        p = f"text_embed.text_blocks.{i}"
        s[p + ".dwconv.weight"] = (Dt, 1, 7)
        s[p + ".dwconv.bias"] = (Dt,)
        s[p + ".norm.weight"] = (Dt,)
        s[p + ".norm.bias"] = (Dt,)
        s[p + ".pwconv1.weight"] = (2 * Dt, Dt)
        s[p + ".pwconv1.bias"] = (2 * Dt,)
        s[p + ".grn.gamma"] = (1, 1, 2 * Dt)
        s[p + ".grn.beta"] = (1, 1, 2 * Dt)
        s[p + ".pwconv2.weight"] = (Dt, 2 * Dt)
        s[p + ".pwconv2.bias"] = (Dt,)
    s["input_embed.proj.weight"] = (D, 2 * mel_dim + Dt)
    s["input_embed.proj.bias"] = (D,)
    for j in (0, 2):
        s[f"input_embed.conv_pos_embed.conv1d.{j}.weight"] = (D, D // 16, 31)
        s[f"input_embed.conv_pos_embed.conv1d.{j}.bias"] = (D,)
    for i in range(a["depth"]):
        p = f"transformer_blocks.{i}"
        s[p + ".attn_norm.linear.weight"] = (6 * D, D)
        s[p + ".attn_norm.linear.bias"] = (6 * D,)
        for n in ("to_q", "to_k", "to_v"):
            s[f"{p}.attn.{n}.weight"] = (inner, D)
            s[f"{p}.attn.{n}.bias"] = (inner,)
        if a.get("qk_norm") is not None:                         # modules.py:397-404: RMSNorm(dim_head) on q and k
            s[p + ".attn.q_norm.weight"] = (a["dim_head"],)
            s[p + ".attn.k_norm.weight"] = (a["dim_head"],)
        s[p + ".attn.to_out.0.weight"] = (D, inner)
        s[p + ".attn.to_out.0.bias"] = (D,)
        s[p + ".ff.ff.0.0.weight"] = (F, D)
        s[p + ".ff.ff.0.0.bias"] = (F,)
        s[p + ".ff.ff.2.weight"] = (D, F)
        s[p + ".ff.ff.2.bias"] = (D,)
    if a.get("long_skip_connection"):                            # dit.py:205: nn.Linear(dim * 2, dim, bias=False)
        s["long_skip_connection.weight"] = (D, 2 * D)
    s["norm_out.linear.weight"] = (2 * D, D)
    s["norm_out.linear.bias"] = (2 * D,)
    s["proj_out.weight"] = (mel_dim, D)
    s["proj_out.bias"] = (mel_dim,)
    return s


def unett_param_shapes(arch: dict, text_num_embeds: int, mel_dim: int = 100) -> dict[str, tuple]:
    a = normalize_arch(arch, mel_dim)
    D, Dt, F = a["dim"], a["text_dim"], int(a["dim"] * a["ff_mult"])
    inner = a["heads"] * a["dim_head"]
    s: dict[str, tuple] = {}
    s["time_embed.time_mlp.0.weight"] = (D, 256)
    s["time_embed.time_mlp.0.bias"] = (D,)
    s["time_embed.time_mlp.2.weight"] = (D, D)
    s["time_embed.time_mlp.2.bias"] = (D,)
    s["text_embed.text_embed.weight"] = (text_num_embeds + 1, Dt)
    s["input_embed.proj.weight"] = (D, 2 * mel_dim + Dt)
    s["input_embed.proj.bias"] = (D,)
    for j in (0, 2):
        s[f"input_embed.conv_pos_embed.conv1d.{j}.weight"] = (D, D // 16, 31)
        s[f"input_embed.conv_pos_embed.conv1d.{j}.bias"] = (D,)
    for i in range(a["depth"]):
        p = f"layers.{i}"
        if i >= a["depth"] // 2:
            s[p + ".0.weight"] = (D, 2 * D)
        s[p + ".1.g"] = (D,)
        for n in ("to_q", "to_k", "to_v"):
            s[f"{p}.2.{n}.weight"] = (inner, D)
            s[f"{p}.2.{n}.bias"] = (inner,)
        s[p + ".2.to_out.0.weight"] = (D, inner)
        s[p + ".2.to_out.0.bias"] = (D,)
        s[p + ".3.g"] = (D,)
        s[p + ".4.ff.0.0.weight"] = (F, D)
        s[p + ".4.ff.0.0.bias"] = (F,)
        s[p + ".4.ff.2.weight"] = (D, F)
        s[p + ".4.ff.2.bias"] = (D,)
    s["norm_out.g"] = (D,)
    s["proj_out.weight"] = (mel_dim, D)
    s["proj_out.bias"] = (mel_dim,)
    return s


def vocos_param_shapes(v: dict) -> dict[str, tuple]:
    """Parameter names of charactr/vocos-mel-24khz's `pytorch_model.bin` (backbone.* / head.*)."""
    C, D, I, L = v["input_channels"], v["dim"], v["intermediate_dim"], v["num_layers"]
    s: dict[str, tuple] = {}
    s["backbone.embed.weight"] = (D, C, 7)
    s["backbone.embed.bias"] = (D,)
    s["backbone.norm.weight"] = (D,)
    s["backbone.norm.bias"] = (D,)
    for i in range(L):
        p = f"backbone.convnext.{i}"
        s[p + ".dwconv.weight"] = (D, 1, 7)
        s[p + ".dwconv.bias"] = (D,)
        s[p + ".norm.weight"] = (D,)
        s[p + ".norm.bias"] = (D,)
        s[p + ".pwconv1.weight"] = (I, D)
        s[p + ".pwconv1.bias"] = (I,)
        s[p + ".pwconv2.weight"] = (D, I)
        s[p + ".pwconv2.bias"] = (D,)
        s[p + ".gamma"] = (D,)
    s["backbone.final_layer_norm.weight"] = (D,)
    s["backbone.final_layer_norm.bias"] = (D,)
    s["head.out.weight"] = (v["n_fft"] + 2, D)
    s["head.out.bias"] = (v["n_fft"] + 2,)
    return s


def bigvgan_param_shapes(v: dict) -> dict[str, tuple]:
    """Parameter names of BigVGAN v2 (`bigvgan_generator.pt` of nvidia/bigvgan_v2_24khz_100band_256x) after
    `remove_weight_norm()` (utils_infer.py:151) -- restated from the published model, SURVEY a20."""
    s: dict[str, tuple] = {}
    ch = v["upsample_initial_channel"]
    s["conv_pre.weight"] = (ch, v["num_mels"], 7)
    s["conv_pre.bias"] = (ch,)
    nk, nd = len(v["resblock_kernel_sizes"]), len(v["resblock_dilation_sizes"])
    for i, (u, k) in enumerate(zip(v["upsample_rates"], v["upsample_kernel_sizes"])):
        s[f"ups.{i}.0.weight"] = (ch, ch // 2, k)      # ConvTranspose1d: [in, out, k]
        s[f"ups.{i}.0.bias"] = (ch // 2,)
        ch //= 2
        for j, rk in enumerate(v["resblock_kernel_sizes"]):
            p = f"resblocks.{i * nk + j}"
            for m in range(nd):
                for cv in ("convs1", "convs2"):
                    s[f"{p}.{cv}.{m}.weight"] = (ch, ch, rk)
                    s[f"{p}.{cv}.{m}.bias"] = (ch,)
            for a in range(2 * nd):
                s[f"{p}.activations.{a}.act.alpha"] = (ch,)
                s[f"{p}.activations.{a}.act.beta"] = (ch,)
    s["activation_post.act.alpha"] = (ch,)
    s["activation_post.act.beta"] = (ch,)
    s["conv_post.weight"] = (1, ch, 7)
    if v.get("use_bias_at_final"):
        s["conv_post.bias"] = (1,)
    return s


def _std_for(name: str, shape: tuple, std: float) -> tuple[float, float]:
    """(mean, std) per tensor class.  Norm weights ~ 1, layer-scale 1/8-ish, embeddings N(0,1), everything else N(0,std)."""
    if name.startswith(("conv_pre.", "ups.", "resblocks.", "conv_post.", "activation_post.")):   # BigVGAN: keep the signal O(1)
        if name.endswith((".alpha", ".beta")):
            return 0.0, 0.3                            # log-scale SnakeBeta parameters
        if name.endswith(".weight") and len(shape) == 3:
            fan = shape[0] * shape[2] / 2 if name.startswith("ups.") else shape[1] * shape[2]
            gain = 0.35 if name.startswith(("resblocks.", "conv_post.")) else 1.0   # residual branches / output stay small: few clipped samples
            return 0.0, gain * float(fan) ** -0.5
        return 0.0, std
    if name.endswith(".norm.weight") or name.endswith("final_layer_norm.weight") or name.endswith(".g") \
            or name == "backbone.norm.weight" or name.endswith(("q_norm.weight", "k_norm.weight")):
        return 1.0, 0.05
    if name.endswith(".gamma") and len(shape) == 1:  # vocos layer-scale
        return 0.125, 0.02
    if name == "text_embed.text_embed.weight":
        return 0.0, 1.0
    if "dwconv.weight" in name:
        return 0.0, 0.3
    if "conv_pos_embed" in name and name.endswith("weight"):
        return 0.0, std
    if name.startswith("head.out"):
        return 0.0, 0.05
    return 0.0, std


def synthetic_state_dict(shapes: dict[str, tuple], seed: int = 0, std: float = 0.02) -> dict[str, torch.Tensor]:
    """Deterministic fp32 CPU tensors keyed by (seed, tensor name); identical on every host with the same torch."""
    out = {}
    for name, shape in shapes.items():
        g = torch.Generator(device="cpu")
        g.manual_seed((zlib.crc32(name.encode()) ^ (seed * 0x9E3779B1)) & 0x7FFFFFFF)
        mean, s = _std_for(name, shape, std)
        out[name] = (torch.randn(shape, generator=g, dtype=torch.float32) * s + mean).contiguous()
    return out


def strip_prefixes(sd: dict[str, torch.Tensor]) -> dict[str, torch.Tensor]:
    """Accepts a reference checkpoint's state dict (utils_infer.py:242-286): strips `ema_model.` / `transformer.`
    prefixes and drops the non-backbone entries (mel_spec buffers, `initted`, `step`)."""
    out = {}
    for k, v in sd.items():
        if k in ("initted", "step", "ema_model.initted", "ema_model.step"):
            continue
        k = k.replace("ema_model.", "", 1) if k.startswith("ema_model.") else k
        if k.startswith("mel_spec."):
            continue
        if k.startswith("transformer."):
            k = k[len("transformer."):]
        out[k] = v
    return out
