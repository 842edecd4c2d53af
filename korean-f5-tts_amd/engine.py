"""Thin Python owner of one libf5hip engine handle: uploads weights, passes raw device pointers + the current HIP stream.

PyTorch is used only for device memory (torch.Tensor.data_ptr), the current stream and host-side constant tables.
"""
from __future__ import annotations

import ctypes as C
import math

import torch

from . import _lib
from .config import normalize_arch


def _stream_ptr(device) -> C.c_void_p:
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def _ptr(t: torch.Tensor | None) -> C.c_void_p:
    return C.c_void_p(0 if t is None else t.data_ptr())


def _dev_f32(t: torch.Tensor, device) -> torch.Tensor:
    return t.detach().to(device=device, dtype=torch.float32).contiguous()


def _h2d_async(t: torch.Tensor, device, dtype) -> torch.Tensor:
    """Host tensor -> device without blocking the host: a pageable `.to(device)` waits for everything already queued on
    the stream (the previous utterance's whole ODE solve), which serialises the host's per-call work with the GPU.
    Staged through torch's caching pinned allocator (it keeps the block alive until the copy has executed)."""
    t = t.detach().to(dtype).contiguous()
    if t.device.type != "cpu":
        return t.to(device)
    return t.pin_memory().to(device, non_blocking=True)


def aux_tables(dim_head: int, max_pos: int, text_dim: int, text_pos_rows: int) -> dict[str, torch.Tensor]:
    """Host-computed constant tables, evaluated with the same torch fp32 ops the reference uses:
    rotary angles (x_transformers RotaryEmbedding as called at dit.py:184,311; restated in-repo at
    runtime/triton_trtllm/model_repo_f5_tts/f5_tts/1/f5_tts_trtllm.py:230-237), the sinusoidal time frequencies
    (modules.py:159-161) and the text position table precompute_freqs_cis (modules.py:202-213)."""
    inv_freq = 1.0 / (10000.0 ** (torch.arange(0, dim_head, 2).float() / dim_head))
    ang = torch.einsum("i,j->ij", torch.arange(max_pos).type_as(inv_freq), inv_freq)  # [max_pos, dh/2]
    half = 128
    tf = torch.exp(torch.arange(half).float() * -(math.log(10000) / (half - 1)))
    out = {"aux.rope_cos": ang.cos(), "aux.rope_sin": ang.sin(), "aux.time_freqs": tf}
    if text_pos_rows > 0:
        freqs = 1.0 / (10000.0 ** (torch.arange(0, text_dim, 2)[: text_dim // 2].float() / text_dim))
        a = torch.outer(torch.arange(text_pos_rows), freqs).float()
        out["aux.text_pos"] = torch.cat([a.cos(), a.sin()], dim=-1)
    return out


class Engine:
    """One f5_engine handle bound to one device (one process per GPU; handles are not shared across streams)."""

    def __init__(self, arch: dict, text_num_embeds: int, mel_dim: int = 100, *, backbone: str = "DiT",
                 precision: str = "f16p", device="cuda", max_pos: int = 8192):
        self.lib = _lib.load()  # raises if the HIP library is not built
        if not torch.cuda.is_available():
            raise _lib.F5Error("no GPU visible: the F5-TTS engine has no CPU path")
        self.device = torch.device(device if device != "cuda" else f"cuda:{torch.cuda.current_device()}")
        a = normalize_arch(arch, mel_dim)
        self.arch = a
        self.backbone = backbone
        self.precision = precision
        self.mel_dim = mel_dim
        self.text_num_embeds = text_num_embeds
        self.max_pos = max_pos
        cfg = _lib.f5_config()
        cfg.backbone = _lib.F5_BACKBONE_DIT if backbone == "DiT" else _lib.F5_BACKBONE_UNETT
        cfg.precision = _lib.PRECISIONS[precision]
        cfg.dim, cfg.depth, cfg.heads, cfg.dim_head = a["dim"], a["depth"], a["heads"], a["dim_head"]
        cfg.ff_dim = int(a["dim"] * a["ff_mult"])
        cfg.text_dim, cfg.conv_layers = a["text_dim"], a["conv_layers"]
        cfg.pe_attn_head = -1 if a["pe_attn_head"] is None else int(a["pe_attn_head"])
        cfg.text_mask_padding = int(bool(a["text_mask_padding"]))
        cfg.attn_mask_enabled = int(bool(a["attn_mask_enabled"]))
        cfg.text_num_embeds, cfg.mel_dim, cfg.max_pos = text_num_embeds, mel_dim, max_pos
        if a.get("qk_norm") not in (None, "rms_norm"):
            raise ValueError(f"Unimplemented qk_norm: {a['qk_norm']}")                       # modules.py:404
        if (a.get("qk_norm") or a.get("long_skip_connection") or a.get("text_embedding_average_upsampling")) and backbone != "DiT":
            raise _lib.F5Error("qk_norm / long_skip_connection / text_embedding_average_upsampling are DiT options")
        if a.get("text_embedding_average_upsampling") and not a["text_mask_padding"]:
            raise AssertionError("text_embedding_average_upsampling requires text_mask_padding to be True")   # dit.py:41-42
        cfg.options = ((_lib.F5_OPT_QK_RMSNORM if a.get("qk_norm") else 0) | (_lib.F5_OPT_LONG_SKIP if a.get("long_skip_connection") else 0) |
                       (_lib.F5_OPT_TEXT_AVG_UPSAMPLE if a.get("text_embedding_average_upsampling") else 0))
        self._h = C.c_void_p()
        with torch.cuda.device(self.device):
            _lib.check(self.lib.f5_create(C.byref(cfg), C.byref(self._h)), "f5_create")
        self.ready = False

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            try:
                self.lib.f5_destroy(h)
            except Exception:
                pass
            self._h = None

    # ------------------------------------------------------------------ weights
    def load_state_dict(self, sd: dict[str, torch.Tensor]):
        a = self.arch
        text_rows = 0
        if a["conv_layers"] > 0:
            text_rows = 8192 if self.backbone == "DiT" else 4096  # dit.py:46 / unett.py:46
        tables = aux_tables(a["dim_head"], self.max_pos, a["text_dim"], text_rows)
        with torch.cuda.device(self.device):
            st = _stream_ptr(self.device)
            for name, t in list(sd.items()) + list(tables.items()):
                d = _dev_f32(t, self.device)
                _lib.check(self.lib.f5_load_weight(self._h, name.encode(), _ptr(d), _lib.shape_array(d.shape), d.dim(),
                                                   st), f"f5_load_weight({name})")
            _lib.check(self.lib.f5_finalize(self._h, st), "f5_finalize")
        self.ready = True

    def reserve(self, max_batch: int, max_frames: int, max_steps: int = 32):
        with torch.cuda.device(self.device):
            _lib.check(self.lib.f5_reserve(self._h, max_batch, max_frames, max_steps), "f5_reserve")

    # ------------------------------------------------------------------ compute
    def text_embed(self, text: torch.Tensor, N: int, lens=None, drop_text=False) -> torch.Tensor:
        B, nt = text.shape
        text = text.to(device=self.device, dtype=torch.long).contiguous()
        out = torch.empty(B, N, self.arch["text_dim"], device=self.device, dtype=torch.float32)
        with torch.cuda.device(self.device):
            _lib.check(self.lib.f5_text_embed(self._h, _ptr(text), B, nt, _lib.int_array(lens), N, int(drop_text),
                                              _ptr(out), _stream_ptr(self.device)), "f5_text_embed")
        return out

    def forward(self, x, cond, text, time, lens=None, cfg_infer=False, drop_audio_cond=False, drop_text=False):
        B, N, _ = x.shape
        x = _dev_f32(x, self.device)
        cond = _dev_f32(cond, self.device)
        text = text.to(device=self.device, dtype=torch.long).contiguous()
        tl = [float(v) for v in time] if not isinstance(time, float) else [time] * B
        if len(tl) == 1 and B > 1:
            tl = tl * B
        out = torch.empty((2 * B if cfg_infer else B), N, self.mel_dim, device=self.device, dtype=torch.float32)
        with torch.cuda.device(self.device):
            _lib.check(self.lib.f5_dit_forward(self._h, _ptr(x), _ptr(cond), _ptr(text), text.shape[1],
                                               _lib.float_array(tl), _lib.int_array(lens), B, N, int(cfg_infer),
                                               int(drop_audio_cond), int(drop_text), _ptr(out),
                                               _stream_ptr(self.device)), "f5_dit_forward")
        return out

    def sample(self, cond, cond_mask, y0, text, t_grid, cfg_strength, lens=None, want_traj=True):
        """cond f32[B,Nc,mel] (Nc <= N: the engine zero-pads, cfm.py:145; None = no_ref_audio), cond_mask bool[B,N],
        y0 f32[B,N,mel], text i64[B,nt], t_grid list[float]."""
        B, N, mel = y0.shape
        cond_frames = 0 if cond is None else cond.shape[1]
        steps = len(t_grid) - 1
        if cond is not None:
            cond = _h2d_async(cond, self.device, torch.float32)
        y0 = _h2d_async(y0, self.device, torch.float32)
        cm = _h2d_async(cond_mask, self.device, torch.uint8)
        text = _h2d_async(text, self.device, torch.long)
        out = torch.empty(B, N, mel, device=self.device, dtype=torch.float32)
        traj = torch.empty(steps + 1, B, N, mel, device=self.device, dtype=torch.float32) if want_traj else None
        with torch.cuda.device(self.device):
            _lib.check(self.lib.f5_sample(self._h, _ptr(cond), cond_frames, _ptr(cm), _ptr(y0), _ptr(text), text.shape[1],
                                          _lib.float_array(t_grid), steps, float(cfg_strength), _lib.int_array(lens),
                                          B, N, _ptr(out), _ptr(traj), _stream_ptr(self.device)), "f5_sample")
        return out, traj

    # ------------------------------------------------------------------ profiling
    def profile(self, on: bool):
        _lib.check(self.lib.f5_profile_enable(self._h, int(on)), "f5_profile_enable")

    def profile_read(self) -> dict:
        n = len(_lib.PROFILE_CLASSES)
        ms = (C.c_float * n)()
        cnt = (C.c_int32 * n)()
        fl = (C.c_double * n)()
        _lib.check(self.lib.f5_profile_read(self._h, ms, cnt, fl, n), "f5_profile_read")
        return {name: dict(ms=ms[i], launches=cnt[i], flops=fl[i]) for i, name in enumerate(_lib.PROFILE_CLASSES)}
