"""Prompt mel front-end with the reference's `MelSpec` surface (model/modules.py:107-146).  mel_spec_type="vocos":
torchaudio MelSpectrogram(n_fft=1024, win=1024, hop=256, n_mels=100, power=1, center=True, norm=None) then
clamp(min=1e-5).log() (modules.py:78-104); mel_spec_type="bigvgan": reflect pad (n_fft - hop) / 2, stft(center=False),
sqrt(re^2 + im^2 + 1e-9), librosa's slaney mel basis, log(clamp(., 1e-5)) (modules.py:33-75).  The arithmetic runs in libf5hip (csrc/mel.hip: strided-view STFT GEMM, magnitude, mel GEMM
with a log epilogue); this file only builds the constant tables on the host.

torchaudio is not installed in the build container, so its HTK filterbank (`melscale_fbanks`, norm=None) is restated
here from its published definition, and so is librosa's slaney filterbank (`librosa.filters.mel`, htk=False,
norm="slaney") for the bigvgan variant -- PARITY UNPINNED for those two tables (no fixture of them exists in the reference)."""
from __future__ import annotations

import ctypes as C
import math

import torch
from torch import nn

from . import _lib


def htk_mel_filterbank(n_freqs: int, n_mels: int, sample_rate: int, f_min: float = 0.0, f_max: float | None = None):
    """torchaudio.functional.melscale_fbanks(n_freqs, f_min, f_max, n_mels, sample_rate, norm=None, mel_scale="htk")
    transposed to [n_mels, n_freqs]."""
    f_max = float(sample_rate // 2) if f_max is None else f_max
    all_freqs = torch.linspace(0, sample_rate // 2, n_freqs)
    m_min = 2595.0 * math.log10(1.0 + f_min / 700.0)
    m_max = 2595.0 * math.log10(1.0 + f_max / 700.0)
    m_pts = torch.linspace(m_min, m_max, n_mels + 2)
    f_pts = 700.0 * (10 ** (m_pts / 2595.0) - 1.0)
    f_diff = f_pts[1:] - f_pts[:-1]
    slopes = f_pts.unsqueeze(0) - all_freqs.unsqueeze(1)          # [n_freqs, n_mels + 2]
    down = (-1.0 * slopes[:, :-2]) / f_diff[:-1]
    up = slopes[:, 2:] / f_diff[1:]
    fb = torch.clamp(torch.min(down, up), min=0.0)                # [n_freqs, n_mels]
    return fb.t().contiguous()


def slaney_mel_filterbank(n_freqs: int, n_mels: int, sample_rate: int, f_min: float = 0.0, f_max: float | None = None):
    """librosa.filters.mel(sr, n_fft, n_mels, fmin, fmax) with its defaults (slaney scale: linear below 1 kHz, log above;
    area-normalised triangles) -> [n_mels, n_freqs]."""
    f_max = sample_rate / 2.0 if f_max is None else f_max
    f_sp, min_log_hz = 200.0 / 3, 1000.0
    min_log_mel, logstep = min_log_hz / f_sp, math.log(6.4) / 27.0

    def hz_to_mel(f):
        return min_log_mel + math.log(f / min_log_hz) / logstep if f >= min_log_hz else f / f_sp

    m = torch.linspace(hz_to_mel(f_min), hz_to_mel(f_max), n_mels + 2, dtype=torch.float64)
    mel_f = torch.where(m >= min_log_mel, min_log_hz * torch.exp(logstep * (m - min_log_mel)), f_sp * m)
    fftfreqs = torch.linspace(0, sample_rate / 2.0, n_freqs, dtype=torch.float64)
    fdiff = mel_f[1:] - mel_f[:-1]
    ramps = mel_f[:, None] - fftfreqs[None, :]
    w = torch.clamp(torch.minimum(-ramps[:-2] / fdiff[:-1, None], ramps[2:] / fdiff[1:, None]), min=0.0)
    return (w * (2.0 / (mel_f[2:n_mels + 2] - mel_f[:n_mels]))[:, None]).to(torch.float32)


def stft_basis(n_fft: int) -> torch.Tensor:
    """[round_up(2F, 4), n_fft]: rows hann[j] cos(2 pi f j / n) (f < F) then hann[j] sin(2 pi f j / n); float64 -> f32."""
    Fb = n_fft // 2 + 1
    ns = (2 * Fb + 3) // 4 * 4
    j = torch.arange(n_fft, dtype=torch.float64)[None, :]
    f = torch.arange(Fb, dtype=torch.float64)[:, None]
    ang = 2 * torch.pi * f * j / n_fft
    win = torch.hann_window(n_fft, dtype=torch.float64)[None, :]
    B = torch.zeros(ns, n_fft, dtype=torch.float64)
    B[:Fb] = torch.cos(ang) * win
    B[Fb:2 * Fb] = torch.sin(ang) * win
    return B.to(torch.float32)


class MelSpec(nn.Module):
    def __init__(self, n_fft=1024, hop_length=256, win_length=1024, n_mel_channels=100, target_sample_rate=24_000,
                 mel_spec_type="vocos"):
        super().__init__()
        assert mel_spec_type in ["vocos", "bigvgan"]
        if win_length != n_fft:
            raise NotImplementedError("win_length != n_fft is unused by every shipped config")
        self.n_fft, self.hop_length, self.win_length = n_fft, hop_length, win_length
        self.n_mel_channels, self.target_sample_rate = n_mel_channels, target_sample_rate
        self.mel_spec_type = mel_spec_type
        self._h = None
        self._h_dev = None

    def _handle(self, dev):
        if self._h is not None and self._h_dev == dev:
            return self._h
        lib = _lib.load()
        h = C.c_void_p()
        with torch.cuda.device(dev):
            _lib.check(lib.f5_mel_create(self.n_fft, self.hop_length, self.n_mel_channels, C.byref(h)), "f5_mel_create")
            Fb = self.n_fft // 2 + 1
            kf = (Fb + 31) // 32 * 32
            fb = torch.zeros(self.n_mel_channels, kf)
            make_fb = htk_mel_filterbank if self.mel_spec_type == "vocos" else slaney_mel_filterbank
            fb[:, :Fb] = make_fb(Fb, self.n_mel_channels, self.target_sample_rate)
            st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
            for name, t in (("aux.dft_basis", stft_basis(self.n_fft)), ("aux.mel_fb", fb)):
                d = t.to(dev).contiguous()
                _lib.check(lib.f5_mel_load(h, name.encode(), C.c_void_p(d.data_ptr()), _lib.shape_array(d.shape), 2, st),
                           f"f5_mel_load({name})")
            torch.cuda.synchronize(dev)
        self._h, self._h_dev = h, dev
        return h

    @torch.no_grad()
    def forward(self, wav: torch.Tensor) -> torch.Tensor:
        """wav f32[b, nw] (or [b, 1, nw]) on a GPU -> log-mel f32[b, n_mels, T] (the reference's layout)."""
        if wav.dim() == 3:
            wav = wav.squeeze(1)
        if wav.device.type != "cuda":
            raise RuntimeError("the HIP mel front-end only runs on a GPU (there is no CPU path)")
        dev = wav.device
        wav = wav.to(torch.float32).contiguous()
        B, nw = wav.shape
        vocos = self.mel_spec_type == "vocos"
        pad, eps = (self.n_fft // 2, 0.0) if vocos else ((self.n_fft - self.hop_length) // 2, 1e-9)
        T = (nw + 2 * pad - self.n_fft) // self.hop_length + 1
        out = torch.empty(B, T, self.n_mel_channels, device=dev, dtype=torch.float32)
        h = self._handle(dev)
        with torch.cuda.device(dev):
            _lib.check(_lib.load().f5_mel_forward_ex(h, C.c_void_p(wav.data_ptr()), B, nw, pad, eps, C.c_void_p(out.data_ptr()),
                                                     C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)), "f5_mel_forward")
        return out.permute(0, 2, 1)
