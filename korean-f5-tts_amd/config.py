"""Architecture constants of the reference's backbones and vocoder.

Values mirror the reference's hydra configs: src/f5_tts/configs/F5TTS_Base.yaml:24-36 (DiT Base),
F5TTS_v1_Base.yaml:31-34, E2TTS_Base.yaml:24-31 (UNetT) and the charactr/vocos-mel-24khz hyper-parameters the
reference's `load_vocoder` instantiates (src/f5_tts/infer/utils_infer.py:114-137).
"""
from __future__ import annotations

MEL_DIM = 100
HOP_LENGTH = 256
SAMPLE_RATE = 24000
N_FFT = 1024
VOCAB_SIZE = 2545  # src/f5_tts/infer/examples/vocab.txt line count

F5TTS_BASE = dict(dim=1024, depth=22, heads=16, dim_head=64, ff_mult=2, text_dim=512, text_mask_padding=False,
                  conv_layers=4, pe_attn_head=1, attn_mask_enabled=False, qk_norm=None)
F5TTS_V1_BASE = dict(F5TTS_BASE, text_mask_padding=True, pe_attn_head=None)
E2TTS_BASE = dict(dim=1024, depth=24, heads=16, dim_head=64, ff_mult=4, text_mask_padding=False, pe_attn_head=1,
                  text_dim=None, conv_layers=0, attn_mask_enabled=False, qk_norm=None)
# small arch used by golden fixtures and CPU-speed parity tests (same code paths, every dimension scaled down)
F5TTS_TINY = dict(dim=256, depth=2, heads=4, dim_head=64, ff_mult=2, text_dim=64, text_mask_padding=False,
                  conv_layers=2, pe_attn_head=1, attn_mask_enabled=False, qk_norm=None)
VOCOS_24K = dict(input_channels=100, dim=512, intermediate_dim=1536, num_layers=8, n_fft=1024, hop_length=256)
# nvidia/bigvgan_v2_24khz_100band_256x (the reference's bigvgan vocoder, infer/utils_infer.py:147-149; config.json of that
# model as published -- restated from memory)
BIGVGAN_V2_24K = dict(num_mels=100, upsample_initial_channel=1536, upsample_rates=[4, 4, 2, 2, 2, 2],
                      upsample_kernel_sizes=[8, 8, 4, 4, 4, 4], resblock_kernel_sizes=[3, 7, 11],
                      resblock_dilation_sizes=[1, 3, 5], use_tanh_at_final=False, use_bias_at_final=False)
BIGVGAN_TINY = dict(num_mels=100, upsample_initial_channel=64, upsample_rates=[4, 2], upsample_kernel_sizes=[8, 4],
                    resblock_kernel_sizes=[3, 7, 11], resblock_dilation_sizes=[1, 3, 5], use_tanh_at_final=False,
                    use_bias_at_final=False)
VOCOS_TINY = dict(input_channels=100, dim=64, intermediate_dim=192, num_layers=2, n_fft=1024, hop_length=256)


def normalize_arch(arch: dict, mel_dim: int = MEL_DIM) -> dict:
    """Fill the reference constructors' defaults (dit.py:147-168, unett.py:107-128)."""
    a = dict(arch)
    a.setdefault("dim_head", 64)
    a.setdefault("heads", 8)
    a.setdefault("depth", 8)
    a.setdefault("ff_mult", 4)
    if a.get("text_dim") is None:
        a["text_dim"] = mel_dim
    a.setdefault("text_mask_padding", True)
    a.setdefault("conv_layers", 0)
    a.setdefault("pe_attn_head", None)
    a.setdefault("attn_mask_enabled", False)
    a.setdefault("qk_norm", None)
    a.setdefault("long_skip_connection", False)                  # dit.py:166,205 (DiT only)
    a.setdefault("text_embedding_average_upsampling", False)     # dit.py:160,39-42 (DiT only)
    a["mel_dim"] = mel_dim
    return a
