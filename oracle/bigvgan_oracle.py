"""TEST INFRASTRUCTURE ONLY -- CPU restatement (PyTorch fp32) of the BigVGAN v2 generator the reference loads for
`mel_spec_type="bigvgan"` (src/f5_tts/infer/utils_infer.py:138-152: `bigvgan.BigVGAN.from_pretrained(
"nvidia/bigvgan_v2_24khz_100band_256x")`, `remove_weight_norm()`, called as `vocoder(mel)` at :705).

**PARITY UNPINNED.**  The reference takes BigVGAN from an un-vendored git submodule (`/root/reference/.gitmodules:1-3`,
directory empty) and its weights from the Hugging Face hub; neither source nor weights nor any test vector exists in the
container, and nothing may be fetched.  What follows restates the PUBLISHED architecture of NVIDIA's BigVGAN v2
(bigvgan.py / activations.py / alias_free_activation/torch/{act,filter,resample}.py of github.com/NVIDIA/BigVGAN, config
`bigvgan_v2_24khz_100band_256x`) from memory: it fixes the arithmetic the HIP kernels (csrc/bigvgan.hip) are tested
against, not the reference's.  Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s cpu_baseline may import this module.

  conv_pre Conv1d(mels, C0, 7, pad 3)
  for each upsample stage i (rates 4,4,2,2,2,2; kernels 8,8,4,4,4,4; C halves every stage from C0 = 1536):
      x = ConvTranspose1d(C_i, C_i / 2, k_u, stride u, padding (k_u - u) / 2)(x)
      x = mean_j AMPBlock1_j(x), j over resblock_kernel_sizes (3, 7, 11), dilations (1, 3, 5) each:
            for d in dilations:  xt = Act(x); xt = Conv1d(C, C, k, dilation d, same)(xt); xt = Act(xt); xt = Conv1d(C, C, k, same)(xt); x = x + xt
      Act = Activation1d(SnakeBeta(C, alpha_logscale=True)): 2x kaiser-sinc upsample -> x + sin^2(exp(alpha) x) / (exp(beta) + 1e-9)
            -> 2x kaiser-sinc low-pass downsample (the anti-aliased "alias-free" activation)
  activation_post (same Act, C_last) -> conv_post Conv1d(C_last, 1, 7, pad 3, bias = use_bias_at_final) -> clamp(-1, 1)
  (tanh instead of the clamp when use_tanh_at_final)."""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F

BIGVGAN_V2_24K_100B_256X = dict(num_mels=100, upsample_initial_channel=1536, upsample_rates=[4, 4, 2, 2, 2, 2],
                                upsample_kernel_sizes=[8, 8, 4, 4, 4, 4], resblock_kernel_sizes=[3, 7, 11],
                                resblock_dilation_sizes=[1, 3, 5], use_tanh_at_final=False, use_bias_at_final=False)


def kaiser_sinc_filter1d(cutoff: float, half_width: float, kernel_size: int) -> torch.Tensor:
    """alias_free_activation/torch/filter.py::kaiser_sinc_filter1d -> [kernel_size] (float32)."""
    even = kernel_size % 2 == 0
    half_size = kernel_size // 2
    delta_f = 4 * half_width
    A = 2.285 * (half_size - 1) * math.pi * delta_f + 7.95
    if A > 50.0:
        beta = 0.1102 * (A - 8.7)
    elif A >= 21.0:
        beta = 0.5842 * (A - 21) ** 0.4 + 0.07886 * (A - 21.0)
    else:
        beta = 0.0
    window = torch.kaiser_window(kernel_size, beta=beta, periodic=False)
    time = (torch.arange(-half_size, half_size) + 0.5) if even else (torch.arange(kernel_size) - half_size)
    filt = 2 * cutoff * window * torch.sinc(2 * cutoff * time)
    return (filt / filt.sum()).to(torch.float32)


def aa_filters(ratio: int = 2, kernel_size: int = 12):
    """(upsample filter, low-pass filter) of Activation1d(up_ratio=2, down_ratio=2, up/down_kernel_size=12): both are
    kaiser_sinc_filter1d(cutoff 0.5 / ratio, half_width 0.6 / ratio, 12)."""
    f = kaiser_sinc_filter1d(0.5 / ratio, 0.6 / ratio, kernel_size)
    return f.clone(), f.clone()


def upsample1d(x, filt, ratio=2):
    """resample.py::UpSample1d.forward; x [B, C, T]."""
    K = filt.numel()
    C = x.shape[1]
    pad = K // ratio - 1
    pad_left = pad * ratio + (K - ratio) // 2
    pad_right = pad * ratio + (K - ratio + 1) // 2
    x = F.pad(x, (pad, pad), mode="replicate")
    x = ratio * F.conv_transpose1d(x, filt.view(1, 1, K).expand(C, -1, -1), stride=ratio, groups=C)
    return x[..., pad_left:-pad_right]


def downsample1d(x, filt, ratio=2):
    """resample.py::DownSample1d -> filter.py::LowPassFilter1d.forward (stride = ratio, padding_mode replicate)."""
    K = filt.numel()
    C = x.shape[1]
    even = K % 2 == 0
    pad_left = K // 2 - int(even)
    pad_right = K // 2
    x = F.pad(x, (pad_left, pad_right), mode="replicate")
    return F.conv1d(x, filt.view(1, 1, K).expand(C, -1, -1), stride=ratio, groups=C)


def snake_beta(x, log_alpha, log_beta):
    """activations.py::SnakeBeta.forward with alpha_logscale=True; x [B, C, T]."""
    a = torch.exp(log_alpha)[None, :, None]
    b = torch.exp(log_beta)[None, :, None]
    return x + (1.0 / (b + 1e-9)) * torch.sin(x * a) ** 2


def activation1d(x, log_alpha, log_beta, fu, fd):
    return downsample1d(snake_beta(upsample1d(x, fu), log_alpha, log_beta), fd)


def amp_block1(V, pfx, x, k, dilations, fu, fd):
    for m, d in enumerate(dilations):
        xt = activation1d(x, V[f"{pfx}.activations.{2 * m}.act.alpha"], V[f"{pfx}.activations.{2 * m}.act.beta"], fu, fd)
        xt = F.conv1d(xt, V[f"{pfx}.convs1.{m}.weight"], V[f"{pfx}.convs1.{m}.bias"], dilation=d, padding=d * (k - 1) // 2)
        xt = activation1d(xt, V[f"{pfx}.activations.{2 * m + 1}.act.alpha"], V[f"{pfx}.activations.{2 * m + 1}.act.beta"], fu, fd)
        xt = F.conv1d(xt, V[f"{pfx}.convs2.{m}.weight"], V[f"{pfx}.convs2.{m}.bias"], padding=(k - 1) // 2)
        x = xt + x
    return x


@torch.no_grad()
def bigvgan_forward(V: dict, cfg: dict, mel: torch.Tensor) -> torch.Tensor:
    """mel f32[B, num_mels, T] -> wav f32[B, 1, T * prod(upsample_rates)]   (BigVGAN.forward)."""
    fu, fd = aa_filters()
    x = F.conv1d(mel.to(torch.float32), V["conv_pre.weight"], V["conv_pre.bias"], padding=3)
    nk = len(cfg["resblock_kernel_sizes"])
    for i, (u, k) in enumerate(zip(cfg["upsample_rates"], cfg["upsample_kernel_sizes"])):
        x = F.conv_transpose1d(x, V[f"ups.{i}.0.weight"], V[f"ups.{i}.0.bias"], stride=u, padding=(k - u) // 2)
        xs = None
        for j, rk in enumerate(cfg["resblock_kernel_sizes"]):
            r = amp_block1(V, f"resblocks.{i * nk + j}", x, rk, cfg["resblock_dilation_sizes"], fu, fd)
            xs = r if xs is None else xs + r
        x = xs / nk
    x = activation1d(x, V["activation_post.act.alpha"], V["activation_post.act.beta"], fu, fd)
    x = F.conv1d(x, V["conv_post.weight"], V.get("conv_post.bias") if cfg.get("use_bias_at_final") else None, padding=3)
    return torch.tanh(x) if cfg.get("use_tanh_at_final") else torch.clamp(x, min=-1.0, max=1.0)


def librosa_slaney_mel_fb(sr: int, n_fft: int, n_mels: int, fmin: float = 0.0, fmax: float | None = None) -> torch.Tensor:
    """librosa.filters.mel(sr, n_fft, n_mels, fmin, fmax) with its defaults htk=False, norm="slaney" (restated from
    memory: librosa is not installed) -> [n_mels, n_fft // 2 + 1] float32.  Used by the bigvgan mel variant
    (src/f5_tts/model/modules.py:33-75: `librosa_mel_fn(sr=24000, n_fft=1024, n_mels=100, fmin=0, fmax=None)`)."""
    fmax = sr / 2.0 if fmax is None else fmax

    def hz_to_mel(f):
        f = torch.as_tensor(f, dtype=torch.float64)
        f_sp = 200.0 / 3
        mels = f / f_sp
        min_log_hz = 1000.0
        min_log_mel = min_log_hz / f_sp
        logstep = math.log(6.4) / 27.0
        return torch.where(f >= min_log_hz, min_log_mel + torch.log(torch.clamp(f, min=1e-10) / min_log_hz) / logstep, mels)

    def mel_to_hz(m):
        f_sp = 200.0 / 3
        freqs = f_sp * m
        min_log_hz = 1000.0
        min_log_mel = min_log_hz / f_sp
        logstep = math.log(6.4) / 27.0
        return torch.where(m >= min_log_mel, min_log_hz * torch.exp(logstep * (m - min_log_mel)), freqs)

    fftfreqs = torch.linspace(0, sr / 2.0, n_fft // 2 + 1, dtype=torch.float64)
    mel_f = mel_to_hz(torch.linspace(float(hz_to_mel(fmin)), float(hz_to_mel(fmax)), n_mels + 2, dtype=torch.float64))
    fdiff = mel_f[1:] - mel_f[:-1]
    ramps = mel_f[:, None] - fftfreqs[None, :]
    lower = -ramps[:-2] / fdiff[:-1, None]
    upper = ramps[2:] / fdiff[1:, None]
    weights = torch.clamp(torch.minimum(lower, upper), min=0.0)
    enorm = 2.0 / (mel_f[2:n_mels + 2] - mel_f[:n_mels])
    return (weights * enorm[:, None]).to(torch.float32)


@torch.no_grad()
def mel_spectrogram_bigvgan(wav, n_fft=1024, hop=256, n_mels=100, sample_rate=24000, win=1024, fmin=0.0, fmax=None):
    """modules.py:33-75 `get_bigvgan_mel_spectrogram`: reflect pad (n_fft - hop) / 2 on both sides, stft(center=False, hann),
    sqrt(re^2 + im^2 + 1e-9), slaney mel basis, log(clamp(., 1e-5)).  wav [B, nw] -> [B, n_mels, T]."""
    fb = librosa_slaney_mel_fb(sample_rate, n_fft, n_mels, fmin, fmax)
    pad = (n_fft - hop) // 2
    w = F.pad(wav.unsqueeze(1), (pad, pad), mode="reflect").squeeze(1)
    spec = torch.stft(w, n_fft, hop_length=hop, win_length=win, window=torch.hann_window(win), center=False, pad_mode="reflect",
                      normalized=False, onesided=True, return_complex=True)
    mag = torch.sqrt(torch.view_as_real(spec).pow(2).sum(-1) + 1e-9)
    return torch.log(torch.clamp(torch.matmul(fb, mag), min=1e-5))
