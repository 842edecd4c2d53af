"""Drop-in for the reference's `CFM` inference surface (model/cfm.py:34-229): same constructor keywords, same
`sample()` signature and return value, same attributes read by callers (`.transformer`, `.mel_spec`,
`.vocab_char_map`, `.device`, `.dim`, `.num_channels`).  Argument handling (cfm.py:103-158, 196-216, 219-229) is
host-side Python; the ODE solve itself (cfm.py:160-191, 218) is ONE call into the HIP engine (f5_sample).
Training (`forward`, cfm.py:231-302) is out of scope.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F
from torch import nn
from torch.nn.utils.rnn import pad_sequence

from .utils import default, exists, get_epss_timesteps, lens_to_mask, list_str_to_idx, list_str_to_tensor


class CFM(nn.Module):
    def __init__(self, transformer: nn.Module, sigma=0.0, odeint_kwargs: dict = dict(method="euler"),
                 audio_drop_prob=0.3, cond_drop_prob=0.2, num_channels=None, mel_spec_module: nn.Module | None = None,
                 mel_spec_kwargs: dict = dict(), frac_lengths_mask=(0.7, 1.0), vocab_char_map=None):
        super().__init__()
        if odeint_kwargs.get("method", "euler") != "euler":
            raise NotImplementedError("only the fixed-grid Euler solver (every shipped config) is built")
        self.frac_lengths_mask = frac_lengths_mask
        if mel_spec_module is None:
            from .mel import MelSpec
            mel_spec_module = MelSpec(**mel_spec_kwargs)
        self.mel_spec = mel_spec_module
        self.num_channels = default(num_channels, getattr(self.mel_spec, "n_mel_channels", 100))
        self.audio_drop_prob, self.cond_drop_prob = audio_drop_prob, cond_drop_prob
        self.transformer = transformer
        self.dim = transformer.dim
        self.sigma = sigma
        self.odeint_kwargs = odeint_kwargs
        self.vocab_char_map = vocab_char_map
        # The reference draws y0 with the generator of the model's device (cfm.py:196-201).  "cpu" reproduces the
        # reference's CPU path bit for bit (the parity target); "cuda" draws on the GPU generator instead.
        self.noise_device = "cpu"

    @property
    def device(self):
        return self.transformer.device

    def forward(self, *a, **k):
        raise NotImplementedError("training (cfm.py:231-302) is outside the inference hot path")

    def load_state_dict(self, sd, strict=True, assign=False):
        """Accepts the reference's CFM state dict (`transformer.*` keys, optional `mel_spec.*` buffers): the backbone
        keeps the weights; everything else in a CFM is parameter-free."""
        return self.transformer.load_state_dict(sd, strict=strict)

    def state_dict(self, *a, **k):
        return {"transformer." + n: t for n, t in self.transformer.state_dict().items()}

    @torch.no_grad()
    def sample(self, cond, text, duration, *, lens=None, steps=32, cfg_strength=1.0, sway_sampling_coef=None,
               seed: int | None = None, max_duration=65536, vocoder=None, use_epss=True, no_ref_audio=False,
               duplicate_test=False, t_inter=0.1, edit_mask=None):
        self.eval()
        device = self.device
        if cond.ndim == 2:  # raw wave -> mel (cfm.py:106-109)
            cond = self.mel_spec(cond.to(device))
            cond = cond.permute(0, 2, 1)
            assert cond.shape[-1] == self.num_channels
        cond = cond.to(device=device, dtype=torch.float32)
        batch, cond_seq_len = cond.shape[:2]
        if not exists(lens):
            lens = torch.full((batch,), cond_seq_len, dtype=torch.long)
        lens = lens.to("cpu", torch.long)

        if isinstance(text, list):
            text = list_str_to_idx(text, self.vocab_char_map) if exists(self.vocab_char_map) else list_str_to_tensor(text)
            assert text.shape[0] == batch
        text_cpu = text.to("cpu", torch.long)

        cond_mask = lens_to_mask(lens)
        if edit_mask is not None:
            cond_mask = cond_mask & edit_mask.to("cpu")
        if isinstance(duration, int):
            duration = torch.full((batch,), duration, dtype=torch.long)
        duration = duration.to("cpu", torch.long)
        duration = torch.maximum(torch.maximum((text_cpu != -1).sum(dim=-1), lens) + 1, duration)
        duration = duration.clamp(max=max_duration)
        N = int(duration.amax())

        if duplicate_test:
            test_cond = F.pad(cond, (0, 0, cond_seq_len, N - 2 * cond_seq_len), value=0.0)
        # cond = F.pad(cond, (0, 0, 0, N - cond_seq_len)) (cfm.py:145) happens inside the engine: the prompt goes down
        # unpadded with its frame count; no_ref_audio (cfm.py:146-147: an all-zero cond) is "zero prompt frames"
        cond_mask = F.pad(cond_mask, (0, N - cond_mask.shape[-1]), value=False)   # (host tensor)

        # noise (cfm.py:196-201): same seed for every sample, drawn per sample at its own length, zero padded
        y0 = []
        for dur in duration.tolist():
            if exists(seed):
                torch.manual_seed(seed)
            y0.append(torch.randn(dur, self.num_channels, device=self.noise_device if self.noise_device == "cpu" else device,
                                  dtype=torch.float32))
        y0 = pad_sequence(y0, padding_value=0, batch_first=True)   # stays where it was drawn; Engine.sample copies it asynchronously

        t_start = 0
        if duplicate_test:
            t_start = t_inter
            y0 = (1 - t_start) * y0.to(device) + t_start * test_cond
            steps = int(steps * (1 - t_start))
        if t_start == 0 and use_epss:
            t = get_epss_timesteps(steps, device="cpu", dtype=torch.float32)
        else:
            t = torch.linspace(t_start, 1, steps + 1, dtype=torch.float32)
        if sway_sampling_coef is not None:
            t = t + sway_sampling_coef * (torch.cos(torch.pi / 2 * t) - 1 + t)

        eng = self.transformer.engine()
        out, trajectory = eng.sample(None if no_ref_audio else cond, cond_mask, y0, text_cpu, t.tolist(), cfg_strength,
                                     lens=duration.tolist() if batch > 1 else None, want_traj=True)
        self.transformer.clear_cache()
        if exists(vocoder):
            out = vocoder(out.permute(0, 2, 1))
        return out, trajectory
