"""Prompt mel front-end placeholder with the reference's attribute surface (model/modules.py:107-146).

The hot path takes mel in and gives mel (or wav) out; the wav -> log-mel front-end is SURVEY.md section 8(f) row 2
("next") and is not built yet.  Calling it raises instead of silently computing on a PyTorch fallback."""
from __future__ import annotations

from torch import nn


class MelSpec(nn.Module):
    def __init__(self, n_fft=1024, hop_length=256, win_length=1024, n_mel_channels=100, target_sample_rate=24_000,
                 mel_spec_type="vocos"):
        super().__init__()
        assert mel_spec_type in ["vocos", "bigvgan"]
        self.n_fft, self.hop_length, self.win_length = n_fft, hop_length, win_length
        self.n_mel_channels, self.target_sample_rate = n_mel_channels, target_sample_rate
        self.mel_spec_type = mel_spec_type

    def forward(self, wav):
        raise NotImplementedError("wav -> mel front-end is not built yet: pass the prompt as a mel [b, n, 100]")
