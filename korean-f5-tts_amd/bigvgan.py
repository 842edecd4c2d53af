"""BigVGAN v2 vocoder with the surface the reference's harness uses for `mel_spec_type="bigvgan"`:
`vocoder(mel[B, 100, T]) -> wav[B, 1, 256 T]` (infer/utils_infer.py:138-152,705; eval/eval_infer_batch.py:208).
Arithmetic runs in libf5hip (csrc/bigvgan.hip).  The reference loads `nvidia/bigvgan_v2_24khz_100band_256x` through an
un-vendored submodule: parameter names are those of that published generator after `remove_weight_norm()`; the
architecture is restated from it (parity unpinned; the checker of tests/test_bigvgan.py is a CPU restatement of the same published design)."""
from __future__ import annotations

import ctypes as C
import math

import torch
from torch import nn

from . import _lib
from . import weights as W
from .config import BIGVGAN_V2_24K
from .engine import _dev_f32, _ptr, _stream_ptr


def kaiser_sinc_filter1d(cutoff: float, half_width: float, kernel_size: int) -> torch.Tensor:
    """The low-pass prototype of BigVGAN's alias-free activation (alias_free_activation/torch/filter.py), float32 [k]."""
    even = kernel_size % 2 == 0
    half_size = kernel_size // 2
    A = 2.285 * (half_size - 1) * math.pi * (4 * half_width) + 7.95
    beta = 0.1102 * (A - 8.7) if A > 50.0 else (0.5842 * (A - 21) ** 0.4 + 0.07886 * (A - 21.0) if A >= 21.0 else 0.0)
    window = torch.kaiser_window(kernel_size, beta=beta, periodic=False)
    time = (torch.arange(-half_size, half_size) + 0.5) if even else (torch.arange(kernel_size) - half_size)
    filt = 2 * cutoff * window * torch.sinc(2 * cutoff * time)
    return (filt / filt.sum()).to(torch.float32)


class BigVGAN(nn.Module):
    def __init__(self, cfg: dict = BIGVGAN_V2_24K, device=None, precision: str = "f32"):
        """precision: "f32" (f32 MFMA throughout) or "f16x3" (the wide stages' convolutions as split-f16 products, include/f5_hip.h
        F5_PREC_F16X3: f32-level results, ~2.5x the f32 GEMM rate)."""
        super().__init__()
        if precision not in ("f32", "f16x3"):
            raise ValueError("BigVGAN precision must be 'f32' or 'f16x3'")
        self.precision = precision
        self.cfg = dict(cfg)
        self._sd: dict[str, torch.Tensor] = {}
        self._h = None
        self._h_dev = None
        self._anchor = nn.Parameter(torch.zeros(1), requires_grad=False)
        self.total_up = 1
        for u in self.cfg["upsample_rates"]:
            self.total_up *= u
        if device is not None:
            self.to(device)

    def param_shapes(self):
        return W.bigvgan_param_shapes(self.cfg)

    def init_synthetic(self, seed: int = 0):
        self.load_state_dict(W.synthetic_state_dict(self.param_shapes(), seed=seed))
        return self

    def state_dict(self, *a, **k):
        return dict(self._sd)

    def remove_weight_norm(self):
        """The reference calls this after loading (utils_infer.py:151); checkpoints given to load_state_dict are expected
        with plain `weight` tensors (weight_g / weight_v pairs are folded here when present)."""
        return self

    def load_state_dict(self, sd, strict=True, assign=False):
        sd = dict(sd)
        for k in [k for k in sd if k.endswith(".weight_g")]:      # fold torch weight_norm parametrisations: w = g * v / ||v||
            base = k[: -len(".weight_g")]
            g, v = sd.pop(k), sd.pop(base + ".weight_v")
            sd[base + ".weight"] = v * (g / v.flatten(1).norm(dim=1).view(-1, *([1] * (v.dim() - 1))))
        sd = {k: v for k, v in sd.items() if not k.endswith((".filter", ".lowpass.filter"))}   # filter buffers are recomputed
        shapes = self.param_shapes()
        missing = [k for k in shapes if k not in sd]
        unexpected = [k for k in sd if k not in shapes]
        if strict and (missing or unexpected):
            raise RuntimeError(f"bigvgan state dict mismatch: missing {missing[:4]}, unexpected {unexpected[:4]}")
        self._sd = {k: sd[k].detach().to("cpu", torch.float32) for k in shapes if k in sd}
        self._drop_handle()
        return nn.modules.module._IncompatibleKeys(missing, unexpected)

    def _drop_handle(self):
        if self._h:
            _lib.load().f5_bigvgan_destroy(self._h)
        self._h = None
        self._h_dev = None

    def __del__(self):
        try:
            self._drop_handle()
        except Exception:
            pass

    def _handle(self):
        dev = self._anchor.device
        if dev.type != "cuda":
            raise RuntimeError("the HIP vocoder only runs on a GPU: call .to('cuda') first (there is no CPU path)")
        if self._h is not None and self._h_dev == dev:
            return self._h
        self._drop_handle()
        lib = _lib.load()
        if not self._sd:
            raise RuntimeError("no vocoder weights loaded")
        c = self.cfg
        cfg = _lib.f5_bigvgan_config()
        cfg.num_mels, cfg.upsample_initial_channel = c["num_mels"], c["upsample_initial_channel"]
        cfg.num_upsamples = len(c["upsample_rates"])
        for i, (u, k) in enumerate(zip(c["upsample_rates"], c["upsample_kernel_sizes"])):
            cfg.upsample_rates[i], cfg.upsample_kernel_sizes[i] = u, k
        cfg.num_kernels = len(c["resblock_kernel_sizes"])
        for j, k in enumerate(c["resblock_kernel_sizes"]):
            cfg.resblock_kernel_sizes[j] = k
        cfg.num_dilations = len(c["resblock_dilation_sizes"])
        for m, d in enumerate(c["resblock_dilation_sizes"]):
            cfg.resblock_dilations[m] = d
        cfg.use_tanh_at_final, cfg.use_bias_at_final = int(bool(c.get("use_tanh_at_final"))), int(bool(c.get("use_bias_at_final")))
        cfg.precision = _lib.PRECISIONS[self.precision]
        h = C.c_void_p()
        with torch.cuda.device(dev):
            _lib.check(lib.f5_bigvgan_create(C.byref(cfg), C.byref(h)), "f5_bigvgan_create")
            st = _stream_ptr(dev)
            f = kaiser_sinc_filter1d(0.25, 0.3, 12)     # Activation1d(up_ratio=2, down_ratio=2, kernel 12): cutoff 0.5/2, half width 0.6/2
            for name, t in list(self._sd.items()) + [("aux.up_filter", f), ("aux.down_filter", f)]:
                d = _dev_f32(t, dev)
                _lib.check(lib.f5_bigvgan_load_weight(h, name.encode(), _ptr(d), _lib.shape_array(d.shape), d.dim(), st),
                           f"f5_bigvgan_load_weight({name})")
            _lib.check(lib.f5_bigvgan_finalize(h, st), "f5_bigvgan_finalize")
        self._h, self._h_dev = h, dev
        return h

    @torch.no_grad()
    def forward(self, mel: torch.Tensor) -> torch.Tensor:
        """mel f32[B, num_mels, T] (any view) -> wav f32[B, 1, T * 256]."""
        h = self._handle()
        dev = self._anchor.device
        if mel.device != dev or mel.dtype != torch.float32:
            mel = mel.detach().to(device=dev, dtype=torch.float32)
        B, Cc, T = mel.shape
        assert Cc == self.cfg["num_mels"]
        wav = torch.empty(B, 1, T * self.total_up, device=dev, dtype=torch.float32)
        sb, sc, st = mel.stride()
        with torch.cuda.device(dev):
            _lib.check(_lib.load().f5_bigvgan_forward(h, _ptr(mel), B, T, sb, sc, st, _ptr(wav), _stream_ptr(dev)),
                       "f5_bigvgan_forward")
        return wav
