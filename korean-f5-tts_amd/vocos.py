"""Vocos vocoder with the surface the reference's harness uses: `vocoder.decode(mel[B, 100, T]) -> wav[B, T']`
(infer/utils_infer.py:114-137,702-703; eval/eval_infer_batch.py:206).  Arithmetic runs in libf5hip (csrc/vocos.hip).
Parameter names are those of charactr/vocos-mel-24khz's `pytorch_model.bin` (backbone.* / head.*)."""
from __future__ import annotations

import ctypes as C

import torch
from torch import nn

from . import _lib
from . import weights as W
from .config import VOCOS_24K
from .engine import _dev_f32, _ptr, _stream_ptr


def idft_basis(n_fft: int) -> tuple[torch.Tensor, torch.Tensor]:
    """(hann window [n_fft], Basis [n_fft, K2]) such that frames = [Re S | Im S | 0-pad] @ Basis^T equals
    irfft(S, n_fft) * hann  (what torch.istft computes per frame).  float64 on the host, rounded once to f32."""
    Fb = n_fft // 2 + 1
    K2 = (2 * Fb + 31) // 32 * 32  # whole 128-byte K-tiles of the f32 LDS-DMA GEMM
    j = torch.arange(n_fft, dtype=torch.float64)[:, None]
    f = torch.arange(Fb, dtype=torch.float64)[None, :]
    ang = 2 * torch.pi * f * j / n_fft
    c = torch.full((1, Fb), 2.0, dtype=torch.float64)
    c[0, 0] = 1.0
    c[0, -1] = 1.0
    win = torch.hann_window(n_fft, dtype=torch.float64)
    re = (c * torch.cos(ang)) * win[:, None] / n_fft
    im = (-c * torch.sin(ang)) * win[:, None] / n_fft
    im[:, 0] = 0.0   # c2r transforms ignore the imaginary part of the DC and Nyquist bins
    im[:, -1] = 0.0
    B = torch.zeros(n_fft, K2, dtype=torch.float64)
    B[:, :Fb] = re
    B[:, Fb:2 * Fb] = im
    return torch.hann_window(n_fft, dtype=torch.float32), B.to(torch.float32)


class Vocos(nn.Module):
    def __init__(self, cfg: dict = VOCOS_24K, device=None):
        super().__init__()
        self.cfg = dict(cfg)
        self._sd: dict[str, torch.Tensor] = {}
        self._h = None
        self._h_dev = None
        self._anchor = nn.Parameter(torch.zeros(1), requires_grad=False)
        if device is not None:
            self.to(device)

    def param_shapes(self):
        return W.vocos_param_shapes(self.cfg)

    def init_synthetic(self, seed: int = 0):
        self.load_state_dict(W.synthetic_state_dict(self.param_shapes(), seed=seed))
        return self

    def state_dict(self, *a, **k):
        return dict(self._sd)

    def load_state_dict(self, sd, strict=True, assign=False):
        shapes = self.param_shapes()
        sd = {k: v for k, v in sd.items() if not k.startswith("feature_extractor.")}
        missing = [k for k in shapes if k not in sd]
        unexpected = [k for k in sd if k not in shapes]
        if strict and (missing or unexpected):
            raise RuntimeError(f"vocos state dict mismatch: missing {missing[:4]}, unexpected {unexpected[:4]}")
        self._sd = {k: sd[k].detach().to("cpu", torch.float32) for k in shapes if k in sd}
        self._drop_handle()
        return nn.modules.module._IncompatibleKeys(missing, unexpected)

    def _drop_handle(self):
        if self._h:
            _lib.load().f5_vocos_destroy(self._h)
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
        cfg = _lib.f5_vocos_config()
        cfg.input_channels, cfg.dim, cfg.intermediate_dim = c["input_channels"], c["dim"], c["intermediate_dim"]
        cfg.num_layers, cfg.n_fft, cfg.hop_length = c["num_layers"], c["n_fft"], c["hop_length"]
        h = C.c_void_p()
        with torch.cuda.device(dev):
            _lib.check(lib.f5_vocos_create(C.byref(cfg), C.byref(h)), "f5_vocos_create")
            st = _stream_ptr(dev)
            win, basis = idft_basis(c["n_fft"])
            for name, t in list(self._sd.items()) + [("aux.hann", win), ("aux.idft_basis", basis)]:
                d = _dev_f32(t, dev)
                _lib.check(lib.f5_vocos_load_weight(h, name.encode(), _ptr(d), _lib.shape_array(d.shape), d.dim(), st),
                           f"f5_vocos_load_weight({name})")
            _lib.check(lib.f5_vocos_finalize(h, st), "f5_vocos_finalize")
        self._h, self._h_dev = h, dev
        return h

    @torch.no_grad()
    def decode(self, mel: torch.Tensor) -> torch.Tensor:
        """mel f32[B, C, T] -> wav f32[B, (T - 1) * hop]."""
        h = self._handle()
        dev = self._anchor.device
        if mel.device != dev or mel.dtype != torch.float32:
            mel = mel.detach().to(device=dev, dtype=torch.float32)
        B, Cc, T = mel.shape
        assert Cc == self.cfg["input_channels"]
        wav = torch.empty(B, (T - 1) * self.cfg["hop_length"], device=dev, dtype=torch.float32)
        sb, sc, st = mel.stride()   # any view: the callers pass sample()'s [B, T, C] output as .permute(0, 2, 1)
        with torch.cuda.device(dev):
            _lib.check(_lib.load().f5_vocos_decode_strided(h, _ptr(mel), B, T, sb, sc, st, _ptr(wav), _stream_ptr(dev)),
                       "f5_vocos_decode")
        return wav

    def forward(self, mel):
        return self.decode(mel)
