"""Backbones with the reference's protocol -- `forward(x, cond, text, time, mask, drop_audio_cond, drop_text,
cfg_infer, cache)`, `.dim`, `.clear_cache()` (model/backbones/dit.py:146-329, unett.py:107-280) -- whose arithmetic
runs entirely in libf5hip (HIP kernels).  Parameters keep the reference's state-dict names so checkpoints load as is.
"""
from __future__ import annotations

import torch
from torch import nn

from . import weights as W
from .config import normalize_arch
from .engine import Engine


class _HipBackbone(nn.Module):
    backbone_name = "DiT"

    def __init__(self, *, mel_dim=100, text_num_embeds=256, precision="parity", device=None, max_pos=8192, **arch):
        super().__init__()
        # "parity" (the default): the fastest operand precision measured inside the 1e-3 mel bar of the fp32 CPU path for THIS backbone
        # (DESIGN.md section 3): DiT "f16p" (f16 blocks, split-f16 input / output layers); the E2-TTS UNetT has no AdaLN gates and
        # every one of its GEMM classes costs ~1e-3 in plain f16, so it gets "f16x3" (split-f16 GEMM products, f16 attention products)
        if precision == "parity":
            precision = "f16p" if self.backbone_name == "DiT" else "f16x3"
        arch.pop("dropout", None)
        arch.pop("attn_backend", None)          # one attention implementation: the gfx950 flash kernel
        arch.pop("checkpoint_activations", None)
        self.arch = normalize_arch(arch, mel_dim)
        self.dim = self.arch["dim"]
        self.depth = self.arch["depth"]
        self.mel_dim = mel_dim
        self.text_num_embeds = text_num_embeds
        self.precision = precision
        self.max_pos = max_pos
        self._device = torch.device(device) if device is not None else None
        self._sd: dict[str, torch.Tensor] = {}
        self._engine: Engine | None = None
        self._anchor = nn.Parameter(torch.zeros(1), requires_grad=False)  # lets `.to(device)` / `.device` work

    # ---- shapes / state dict -------------------------------------------------------------------------
    def param_shapes(self):
        fn = W.dit_param_shapes if self.backbone_name == "DiT" else W.unett_param_shapes
        return fn(self.arch, self.text_num_embeds, self.mel_dim)

    def init_synthetic(self, seed: int = 0):
        """Deterministic random weights (see weights.synthetic_state_dict); used by tests and bench."""
        self.load_state_dict(W.synthetic_state_dict(self.param_shapes(), seed=seed))
        return self

    def state_dict(self, *a, **k):  # noqa: D401 - reference names, fp32 host copies
        return dict(self._sd)

    def load_state_dict(self, sd, strict=True, assign=False):
        sd = W.strip_prefixes(sd)
        shapes = self.param_shapes()
        missing = [k for k in shapes if k not in sd]
        unexpected = [k for k in sd if k not in shapes]
        if strict and (missing or unexpected):
            raise RuntimeError(f"state dict mismatch: missing {missing[:4]}..., unexpected {unexpected[:4]}...")
        for k, shp in shapes.items():
            if k in sd and tuple(sd[k].shape) != tuple(shp):
                raise RuntimeError(f"{k}: shape {tuple(sd[k].shape)} != {tuple(shp)}")
        self._sd = {k: sd[k].detach().to("cpu", torch.float32) for k in shapes if k in sd}
        self._engine = None  # re-upload lazily on the current device
        return nn.modules.module._IncompatibleKeys(missing, unexpected)

    # ---- engine ----------------------------------------------------------------------------------------
    @property
    def device(self):
        return self._anchor.device

    def engine(self) -> Engine:
        dev = self._anchor.device
        if dev.type != "cuda":
            raise RuntimeError("the HIP backbone only runs on a GPU: call .to('cuda') first (there is no CPU path)")
        if self._engine is None or self._engine.device != dev:
            if not self._sd:
                raise RuntimeError("no weights loaded: call load_state_dict() or init_synthetic()")
            e = Engine(self.arch, self.text_num_embeds, self.mel_dim, backbone=self.backbone_name,
                       precision=self.precision, device=dev, max_pos=self.max_pos)
            e.load_state_dict(self._sd)
            self._engine = e
        return self._engine

    def clear_cache(self):
        """The reference clears its per-sample() text cache here (dit.py:275-276); the engine keeps no state
        between calls, so there is nothing to clear."""
        return None

    def forward(self, x, cond, text, time, mask=None, drop_audio_cond=False, drop_text=False, cfg_infer=False,
                cache=False):
        lens = None
        if mask is not None:
            lens = mask.sum(dim=1).tolist()
        t = time.reshape(-1).tolist() if isinstance(time, torch.Tensor) else [float(time)]
        out = self.engine().forward(x, cond, text, t, lens=lens, cfg_infer=cfg_infer, drop_audio_cond=drop_audio_cond,
                                    drop_text=drop_text)
        return out.to(x.dtype)


class DiT(_HipBackbone):
    backbone_name = "DiT"


class UNetT(_HipBackbone):
    backbone_name = "UNetT"

    def __init__(self, *, skip_connect_type="concat", **kw):
        if skip_connect_type != "concat":
            raise NotImplementedError("only skip_connect_type='concat' (the E2TTS configs) is built")
        super().__init__(**kw)
