"""Inference harness around the hot path, with the reference's function names and behaviour
(src/f5_tts/infer/utils_infer.py) -- SURVEY.md section 8(f) rows 1 and 3:

  chunk_text                         utils_infer.py:83-110      text -> chunks of <= max_chars utf-8 bytes
  convert_peft_state_dict_to_plain   utils_infer.py:198-239     PEFT/LoRA keys -> plain weights (W + B A * alpha / r)
  load_checkpoint                    utils_infer.py:242-286     .pt / .safetensors, EMA prefix, legacy mel buffers
  load_model                         utils_infer.py:292-345     CFM(model_cls(**cfg, text_num_embeds=vocab+1, mel_dim))
  load_vocoder                       utils_infer.py:114-137     local vocos checkpoint only (no network here)
  infer_batch_process / infer_process utils_infer.py:453-778    mono mix, RMS-to-0.1, duration formula, sample(),
                                                                 prompt slicing, fp32 vocoder input, rescale, cross-fade

Out of scope here: pydub silence clipping, Whisper ASR, resampling (torchaudio is absent:
prompts must already be at 24 kHz), pinyin / Korean G2P tokenisers (text is tokenised per character through
`vocab_char_map`, or as utf-8 bytes when the model has no vocabulary, exactly as CFM.sample does for list[str]).
Only checkpoints are loaded with loaders that execute nothing from the file (safetensors, torch.load(weights_only=True)).
"""
from __future__ import annotations

import re

import numpy as np
import torch

from .cfm import CFM
from .config import HOP_LENGTH, MEL_DIM, N_FFT, SAMPLE_RATE
from .utils import load_vocab
from .vocos import Vocos

target_sample_rate = SAMPLE_RATE
n_mel_channels = MEL_DIM
hop_length = HOP_LENGTH
win_length = N_FFT
n_fft = N_FFT
mel_spec_type = "vocos"
target_rms = 0.1
cross_fade_duration = 0.15
ode_method = "euler"
nfe_step = 32
cfg_strength = 2.0
sway_sampling_coef = -1.0
speed = 1.0
fix_duration = None


def chunk_text(text: str, max_chars: int = 135) -> list[str]:
    chunks, cur = [], ""
    for sent in re.split(r"(?<=[;:,.!?])\s+|(?<=[；：，。！？])", text):
        piece = sent + " " if sent and len(sent[-1].encode("utf-8")) == 1 else sent
        if len(cur.encode("utf-8")) + len(sent.encode("utf-8")) <= max_chars:
            cur += piece
        else:
            if cur:
                chunks.append(cur.strip())
            cur = piece
    if cur:
        chunks.append(cur.strip())
    return chunks


def convert_peft_state_dict_to_plain(state_dict: dict, lora_alpha: float = 32.0, lora_r: int = 16) -> dict:
    """PEFT checkpoints (`base_model.model.*`, `.base_layer.`, `.lora_A/B.default.`) -> plain names with the low-rank
    update merged: W <- W + (B @ A) * alpha / r."""
    prefix = "base_model.model."
    if not any(k.startswith(prefix) for k in state_dict):
        return state_dict
    scale = lora_alpha / lora_r
    plain = {k[len(prefix):]: v for k, v in state_dict.items() if k.startswith(prefix)}
    out = {}
    for k, v in plain.items():
        if ".lora_A." in k or ".lora_B." in k:
            continue
        if k.endswith(".base_layer.weight"):
            stem = k[: -len(".base_layer.weight")]
            a, b = plain.get(stem + ".lora_A.default.weight"), plain.get(stem + ".lora_B.default.weight")
            if a is not None and b is not None:
                out[stem + ".weight"] = (v + (b @ a).to(v.dtype) * scale).to(v.dtype)
            else:
                out[k] = v
        elif k.endswith(".base_layer.bias"):
            out.setdefault(k[: -len(".base_layer.bias")] + ".bias", v)
        else:
            out[k] = v
    return out


def load_checkpoint(model, ckpt_path: str, device: str, dtype=None, use_ema: bool = True):
    """Loads a reference checkpoint into a CFM whose backbone is a HIP backbone.  `dtype` is accepted for signature
    compatibility; the engine's operand precision is the backbone's `precision` (weights are kept in fp32 on the host
    and converted when they are uploaded)."""
    if ckpt_path.split(".")[-1] == "safetensors":
        from safetensors.torch import load_file

        ckpt = load_file(ckpt_path, device="cpu")
        ckpt = {"ema_model_state_dict": ckpt} if use_ema else {"model_state_dict": ckpt}
    else:
        ckpt = torch.load(ckpt_path, map_location="cpu", weights_only=True)
    if use_ema:
        sd = {k.replace("ema_model.", ""): v for k, v in ckpt["ema_model_state_dict"].items() if k not in ("initted", "step")}
        for legacy in ("mel_spec.mel_stft.mel_scale.fb", "mel_spec.mel_stft.spectrogram.window"):
            sd.pop(legacy, None)
    else:
        sd = ckpt["model_state_dict"]
    model.load_state_dict(convert_peft_state_dict_to_plain(sd))
    return model.to(device)


def load_model(model_cls, model_cfg: dict, ckpt_path: str | None, mel_spec_type=mel_spec_type, vocab_file: str = "",
               ode_method=ode_method, use_ema=True, device="cuda", precision="parity", **_ignored):
    """CFM(transformer=model_cls(**model_cfg, text_num_embeds=vocab_size + 1, mel_dim=100), ...) + load_checkpoint.
    `ckpt_path=None` keeps whatever weights the caller loads afterwards (e.g. init_synthetic())."""
    vocab_char_map, vocab_size = load_vocab(vocab_file) if vocab_file else (None, 256)
    tr = model_cls(**model_cfg, text_num_embeds=vocab_size + 1, mel_dim=n_mel_channels, precision=precision)
    model = CFM(transformer=tr,
                mel_spec_kwargs=dict(n_fft=n_fft, hop_length=hop_length, win_length=win_length,
                                     n_mel_channels=n_mel_channels, target_sample_rate=target_sample_rate,
                                     mel_spec_type=mel_spec_type),
                odeint_kwargs=dict(method=ode_method), vocab_char_map=vocab_char_map)
    if ckpt_path:
        model = load_checkpoint(model, ckpt_path, device, use_ema=use_ema)
    return model.to(device)


def load_vocoder(vocoder_name="vocos", is_local=True, local_path="", device="cuda", hf_cache_dir=None, precision="f32"):
    """utils_infer.py:114-153.  vocos: `pytorch_model.bin` of charactr/vocos-mel-24khz; bigvgan: `bigvgan_generator.pt` of
    nvidia/bigvgan_v2_24khz_100band_256x ({"generator": state_dict}, weight-norm pairs folded on load).
    precision (bigvgan only; not a reference argument): "f32" or "f16x3" (split-f16 products, f32-level results, ~2x faster)."""
    if not is_local:
        raise RuntimeError("no network in this environment: pass is_local=True and a directory with the vocoder weights")
    if vocoder_name == "bigvgan":
        from .bigvgan import BigVGAN
        voc = BigVGAN(precision=precision)
        ck = torch.load(f"{local_path}/bigvgan_generator.pt", map_location="cpu", weights_only=True)
        voc.load_state_dict(ck.get("generator", ck))
        return voc.eval().to(device)
    if vocoder_name != "vocos":
        raise ValueError(f"unknown vocoder {vocoder_name!r}")
    voc = Vocos()
    sd = torch.load(f"{local_path}/pytorch_model.bin", map_location="cpu", weights_only=True)
    voc.load_state_dict(sd)
    return voc.eval().to(device)


def sinc_resample(waveform: torch.Tensor, orig_freq: int, new_freq: int, lowpass_filter_width: int = 6,
                  rolloff: float = 0.99) -> torch.Tensor:
    """torchaudio.transforms.Resample(orig_freq, new_freq) with its defaults (resampling_method="sinc_interp_hann",
    lowpass_filter_width=6, rolloff=0.99), which the reference applies to prompts that are not at 24 kHz
    (utils_infer.py:530-532, on the host, before the audio goes to the device).  torchaudio is not installed: this restates
    its published polyphase windowed-sinc algorithm (functional._get_sinc_resample_kernel / _apply_sinc_resample_kernel)
    -- PARITY UNPINNED for this step.  waveform f32[c, n] on the host -> f32[c, ceil(n * new / orig)]."""
    import math
    if orig_freq == new_freq:
        return waveform
    g = math.gcd(int(orig_freq), int(new_freq))
    orig, new = int(orig_freq) // g, int(new_freq) // g
    base_freq = min(orig, new) * rolloff
    width = math.ceil(lowpass_filter_width * orig / base_freq)
    idx = torch.arange(-width, width + orig, dtype=torch.float64)[None, None] / orig
    t = torch.arange(0, -new, -1, dtype=torch.float64)[:, None, None] / new + idx
    t = (t * base_freq).clamp_(-lowpass_filter_width, lowpass_filter_width)
    window = torch.cos(t * math.pi / lowpass_filter_width / 2) ** 2
    t = t * math.pi
    kernels = torch.where(t == 0, torch.ones_like(t), t.sin() / t) * window * (base_freq / orig)
    kernels = kernels.to(waveform.dtype)                                       # [new, 1, 2 width + orig]
    n = waveform.shape[-1]
    x = torch.nn.functional.pad(waveform.reshape(-1, n), (width, width + orig))
    y = torch.nn.functional.conv1d(x[:, None], kernels, stride=orig)           # [c, new, frames]
    y = y.transpose(1, 2).reshape(x.shape[0], -1)[..., : math.ceil(new * n / orig)]
    return y.reshape(*waveform.shape[:-1], -1)


def prompt_numerics(audio: torch.Tensor, sr: int, ref_text: str, gen_text: str, speed_: float = speed,
                    fix_duration_=None, target_rms_: float = target_rms):
    """The host arithmetic of process_batch (utils_infer.py:523-533,541-544,678-685) factored out so that it can be
    pinned by hand-computed cases: returns (audio mono RMS-normalised [1, nw], rms, ref_audio_len, duration)."""
    if audio.shape[0] > 1:
        audio = torch.mean(audio, dim=0, keepdim=True)
    rms = torch.sqrt(torch.mean(torch.square(audio)))
    if rms < target_rms_:
        audio = audio * target_rms_ / rms
    if sr != target_sample_rate:
        audio = sinc_resample(audio, sr, target_sample_rate)                   # utils_infer.py:530-532
    if len(ref_text[-1].encode("utf-8")) == 1:
        ref_text = ref_text + " "
    local_speed = 0.3 if len(gen_text.encode("utf-8")) < 10 else speed_
    ref_audio_len = audio.shape[-1] // hop_length
    if fix_duration_ is not None:
        duration = int(fix_duration_ * target_sample_rate / hop_length)
    else:
        ref_text_len = len(ref_text.encode("utf-8"))
        gen_text_len = len(gen_text.encode("utf-8"))
        duration = ref_audio_len + int(ref_audio_len / ref_text_len * gen_text_len / local_speed)
    return audio, float(rms), ref_text, ref_audio_len, duration


def cross_fade_concat(waves: list[np.ndarray], cross_fade_duration_: float = cross_fade_duration) -> np.ndarray:
    """utils_infer.py:734-775."""
    if cross_fade_duration_ <= 0:
        return np.concatenate(waves)
    final = waves[0]
    for nxt in waves[1:]:
        n = min(int(cross_fade_duration_ * target_sample_rate), len(final), len(nxt))
        if n <= 0:
            final = np.concatenate([final, nxt])
            continue
        mix = final[-n:] * np.linspace(1, 0, n) + nxt[:n] * np.linspace(0, 1, n)
        final = np.concatenate([final[:-n], mix, nxt[n:]])
    return final


def infer_batch_process(ref_audio, ref_text, gen_text_batches, model_obj, vocoder, mel_spec_type=mel_spec_type,
                        progress=None, target_rms=target_rms, cross_fade_duration=cross_fade_duration,
                        nfe_step=nfe_step, cfg_strength=cfg_strength, sway_sampling_coef=sway_sampling_coef,
                        speed=speed, fix_duration=fix_duration, device=None, streaming=False, chunk_size=2048, seed=None,
                        text_tokenizer=None):
    """A GENERATOR, as in the reference (utils_infer.py:504-522,711-778):
      streaming=False  yields ONE item (final_wave f32 numpy, sample_rate, combined mel [100, T_total]) -- the cross-faded
                       concatenation over the text batches; (None, sample_rate, None) when there is no batch
                       (`infer_process` returns `next(...)` of it);
      streaming=True   yields (wave[j : j + chunk_size], sample_rate) per chunk of every batch's waveform in turn, no cross-fade
                       (the socket server's mode, socket_server.py:138-177).
    `progress`: None or an object with `.tqdm(iterable)` (the reference passes the tqdm module).

    Text front-end: the reference turns `ref_text + gen_text` into tokens per `model_obj._tokenizer_type` -- for the kor_*
    types through Korean G2P / jamo decomposition / allophone rules (utils_infer.py:549-660: g2pk and the repo's own rule
    tables, CPU string work outside this engine's scope).  Here the string is handed to `model_obj.sample`, which maps it
    per character through `vocab_char_map` (the reference's "custom"/char path); a kor_* model therefore needs
    `text_tokenizer`: a callable str -> list[str] producing exactly the reference's tokens.  Without one this raises rather
    than synthesise from ids that are almost all 0 (raw Hangul is not in a jamo / allophone vocabulary)."""
    audio, sr = ref_audio
    device = device if device is not None else model_obj.device
    tok_type = getattr(model_obj, "_tokenizer_type", "custom")
    if text_tokenizer is None and isinstance(tok_type, str) and tok_type.startswith("kor_"):
        raise NotImplementedError(f"model tokenizer type {tok_type!r}: pass text_tokenizer= (str -> list[str], the reference's "
                                  "utils_infer.py:549-660 conversion) -- the Korean G2P / allophone front-end is not part of this engine")

    def process_batch(gen_text):
        """One text batch -> (wave f32 numpy [nw], generated mel numpy [100, T])   (utils_infer.py:541-709)."""
        a, rms, rtext, ref_len, duration = prompt_numerics(audio, sr, ref_text, gen_text, speed, fix_duration, target_rms)
        a = a.to(device)
        text_list = [text_tokenizer(rtext + gen_text)] if text_tokenizer is not None else [rtext + gen_text]
        vocab = getattr(model_obj, "vocab_char_map", None)
        if vocab is not None:
            toks = text_list[0]
            missing = sum(1 for c in toks if c not in vocab)
            if len(toks) >= 8 and missing > 0.3 * len(toks):
                import warnings
                warnings.warn(f"{missing} of {len(toks)} text tokens are not in the model's vocabulary (they all map to id 0): "
                              "this checkpoint expects a tokenised input (text_tokenizer=)", RuntimeWarning, stacklevel=2)
        with torch.inference_mode():
            generated, _ = model_obj.sample(cond=a, text=text_list, duration=duration, steps=nfe_step,
                                            cfg_strength=cfg_strength, sway_sampling_coef=sway_sampling_coef, seed=seed)
            generated = generated.to(torch.float32)[:, ref_len:, :].permute(0, 2, 1)
            wave = vocoder.decode(generated) if mel_spec_type == "vocos" else vocoder(generated)   # utils_infer.py:702-705
            if rms < target_rms:
                wave = wave * rms / target_rms
            return wave.squeeze().cpu().numpy(), generated[0].cpu().numpy()

    batches = progress.tqdm(gen_text_batches) if progress is not None and hasattr(progress, "tqdm") else gen_text_batches
    if streaming:
        for gen_text in batches:
            wave, _spec = process_batch(gen_text)
            for j in range(0, len(wave), chunk_size):
                yield wave[j:j + chunk_size], target_sample_rate
        return
    waves, specs = [], []
    for gen_text in batches:
        wave, spec = process_batch(gen_text)
        waves.append(wave)
        specs.append(spec)
    if not waves:
        yield None, target_sample_rate, None
        return
    yield cross_fade_concat(waves, cross_fade_duration), target_sample_rate, np.concatenate(specs, axis=1)


def infer_process(ref_audio, ref_text, gen_text, model_obj, vocoder, mel_spec_type=mel_spec_type, show_info=print, progress=None,
                  target_rms=target_rms, cross_fade_duration=cross_fade_duration, nfe_step=nfe_step, cfg_strength=cfg_strength,
                  sway_sampling_coef=sway_sampling_coef, speed=speed, fix_duration=fix_duration, device=None, seed=None,
                  text_tokenizer=None):
    """ref_audio = (tensor [channels, nw], sample_rate) instead of a path (no torchaudio.load here); otherwise
    utils_infer.py:453-498: max_chars from the prompt's bytes-per-second, chunk, infer_batch_process."""
    audio, sr = ref_audio
    max_chars = int(len(ref_text.encode("utf-8")) / (audio.shape[-1] / sr) * (22 - audio.shape[-1] / sr) * speed)
    batches = chunk_text(gen_text, max_chars=max_chars)
    if show_info is not None:
        show_info(f"Generating audio in {len(batches)} batches...")
    return next(infer_batch_process((audio, sr), ref_text, batches, model_obj, vocoder, mel_spec_type=mel_spec_type,
                                    progress=progress, target_rms=target_rms, cross_fade_duration=cross_fade_duration,
                                    nfe_step=nfe_step, cfg_strength=cfg_strength, sway_sampling_coef=sway_sampling_coef,
                                    speed=speed, fix_duration=fix_duration, device=device, seed=seed,
                                    text_tokenizer=text_tokenizer))
