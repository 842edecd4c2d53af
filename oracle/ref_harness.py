"""TEST INFRASTRUCTURE ONLY -- loads the *reference* hot path from /root/reference for oracle pinning.

Imports `f5_tts.model.cfm`, `f5_tts.model.backbones.dit`, `f5_tts.model.backbones.unett`, `f5_tts.model.modules`
and `f5_tts.model.utils` unmodified from the reference tree, in this container only, so that the CPU oracle
(`oracle/f5_oracle.py`) can be checked against the real code and golden vectors can be generated
(`oracle/make_golden.py`).  Nothing here is imported by the product path, and nothing from the reference is
copied: the reference files are executed where they lie (sys.dont_write_bytecode, the tree is read-only).

Six third-party modules the reference imports are absent from this image.  They are replaced by
stand-ins that live only in this harness:

  arithmetic-bearing (formulas pinned by the reference's own independent TRT-LLM restatement):
    x_transformers.x_transformers.RotaryEmbedding / apply_rotary_pos_emb
        interleaved-pair rotary, inv_freq = 10000^(-2j/d); pinned by
        runtime/triton_trtllm/model_repo_f5_tts/f5_tts/1/f5_tts_trtllm.py:230-237 (repeat_interleave(2) freqs) and
        runtime/triton_trtllm/patch/f5tts/modules.py:210-276 (rotate-every-two, x*cos + rot(x)*sin).
    x_transformers.RMSNorm (UNetT only) -- F.normalize(x, dim=-1) * sqrt(dim) * g; NO in-repo restatement:
        parity unpinned for that one op.
    torchdiffeq.odeint(method="euler") -- fixed-grid Euler y += dt * f(t, y), returns the stacked trajectory;
        pinned by f5_tts_trtllm.py:248-250,360-369.
  inert (imported at module top, never executed on the mel-in / mel-out path):
    torchaudio, librosa.filters.mel, rjieba, pypinyin.
The `f5_tts`, `f5_tts.model`, `f5_tts.model.backbones` package objects are pre-registered empty (with __path__) so
that f5_tts/model/__init__.py (which pulls the trainer -> wandb / ema_pytorch) is not executed.
"""
from __future__ import annotations

import importlib
import math
import os
import sys
import types

import torch

REF_ROOT = os.environ.get("F5_REFERENCE_ROOT", "/root/reference")
REF_SRC = os.path.join(REF_ROOT, "src")


def available() -> bool:
    return os.path.isfile(os.path.join(REF_SRC, "f5_tts", "model", "cfm.py"))


# ----------------------------------------------------------------------------------------------- stand-ins


class _RotaryEmbedding(torch.nn.Module):
    """x_transformers.RotaryEmbedding(dim) as used at dit.py:184,311 (no xpos, interpolation 1)."""

    def __init__(self, dim, base=10000.0):
        super().__init__()
        inv_freq = 1.0 / (base ** (torch.arange(0, dim, 2).float() / dim))
        self.register_buffer("inv_freq", inv_freq, persistent=False)

    def forward_from_seq_len(self, seq_len):
        t = torch.arange(seq_len, device=self.inv_freq.device)
        return self.forward(t)

    def forward(self, t):
        freqs = torch.einsum("i,j->ij", t.type_as(self.inv_freq), self.inv_freq)
        freqs = torch.stack((freqs, freqs), dim=-1).reshape(*freqs.shape[:-1], -1)  # interleave pairs
        return freqs[None], 1.0  # [1, n, d], xpos scale = 1


def _rotate_half(x):
    x = x.reshape(*x.shape[:-1], -1, 2)
    x1, x2 = x.unbind(dim=-1)
    return torch.stack((-x2, x1), dim=-1).reshape(*x.shape[:-2], -1)


def _apply_rotary_pos_emb(t, freqs, scale=1):
    rot_dim, seq_len, orig_dtype = freqs.shape[-1], t.shape[-2], t.dtype
    freqs = freqs[:, -seq_len:, :]
    if t.ndim == 4 and freqs.ndim == 3:
        freqs = freqs[:, None]
    t, t_unrotated = t[..., :rot_dim], t[..., rot_dim:]
    t = (t * freqs.cos() * scale) + (_rotate_half(t) * freqs.sin() * scale)
    return torch.cat((t, t_unrotated), dim=-1).type(orig_dtype)


class _XRMSNorm(torch.nn.Module):
    """x_transformers.RMSNorm [from memory; parity unpinned]: normalize * sqrt(dim) * g, g init 1."""

    def __init__(self, dim):
        super().__init__()
        self.scale = dim**0.5
        self.g = torch.nn.Parameter(torch.ones(dim))

    def forward(self, x):
        return torch.nn.functional.normalize(x, dim=-1) * self.scale * self.g


def _odeint(fn, y0, t, method="euler", **kw):
    assert method == "euler", "stand-in implements the fixed-grid Euler solver only"
    ys = [y0]
    y = y0
    for i in range(t.shape[0] - 1):
        dt = t[i + 1] - t[i]
        y = y + dt * fn(t[i], y)
        ys.append(y)
    return torch.stack(ys, dim=0)


def _mod(name, **attrs):
    import importlib.machinery

    m = types.ModuleType(name)
    m.__spec__ = importlib.machinery.ModuleSpec(name, loader=None)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


_loaded = None


def load():
    """Returns a namespace with the reference's CFM, DiT, UNetT classes and the modules/utils modules."""
    global _loaded
    if _loaded is not None:
        return _loaded
    if not available():
        raise RuntimeError(f"reference tree not found under {REF_ROOT}")
    sys.dont_write_bytecode = True

    xt = _mod("x_transformers", RMSNorm=_XRMSNorm)
    xtx = _mod("x_transformers.x_transformers", RotaryEmbedding=_RotaryEmbedding,
               apply_rotary_pos_emb=_apply_rotary_pos_emb)
    xt.x_transformers = xtx
    _mod("torchdiffeq", odeint=_odeint)
    ta = _mod("torchaudio")
    ta.transforms = _mod("torchaudio.transforms")
    lb = _mod("librosa")
    lbf = _mod("librosa.filters", mel=lambda *a, **k: (_ for _ in ()).throw(RuntimeError("inert stand-in")))
    lb.filters = lbf
    _mod("rjieba")
    _mod("pypinyin", lazy_pinyin=None, Style=None)

    for pkg, rel in (("f5_tts", "f5_tts"), ("f5_tts.model", "f5_tts/model"),
                     ("f5_tts.model.backbones", "f5_tts/model/backbones")):
        if pkg not in sys.modules:
            m = types.ModuleType(pkg)
            m.__path__ = [os.path.join(REF_SRC, rel)]
            sys.modules[pkg] = m

    utils = importlib.import_module("f5_tts.model.utils")
    modules = importlib.import_module("f5_tts.model.modules")
    dit = importlib.import_module("f5_tts.model.backbones.dit")
    unett = importlib.import_module("f5_tts.model.backbones.unett")
    cfm = importlib.import_module("f5_tts.model.cfm")
    _loaded = types.SimpleNamespace(CFM=cfm.CFM, DiT=dit.DiT, UNetT=unett.UNetT, modules=modules, utils=utils,
                                    cfm=cfm, dit=dit, unett=unett)
    return _loaded


class InertMelSpec(torch.nn.Module):
    """Passed as CFM(mel_spec_module=...) so the reference never builds torchaudio's MelSpectrogram."""

    n_mel_channels = 100

    def forward(self, wav):  # pragma: no cover
        raise RuntimeError("mel front-end is not part of the mel-in oracle harness")


def build_reference_cfm(arch: dict, text_num_embeds: int, mel_dim: int = 100, backbone: str = "DiT"):
    ref = load()
    cls = ref.DiT if backbone == "DiT" else ref.UNetT
    tr = cls(**arch, text_num_embeds=text_num_embeds, mel_dim=mel_dim)
    model = ref.CFM(transformer=tr, mel_spec_module=InertMelSpec())
    return model.eval()


def load_infer():
    """Imports the reference's inference harness (src/f5_tts/infer/utils_infer.py) for pinning its host arithmetic
    (chunk_text, _convert_peft_state_dict_to_plain, the duration formula / slicing / rescale / cross-fade of
    infer_batch_process).  Additional inert stand-ins: matplotlib, pydub, vocos, and a pass-through `rjieba.cut`
    (the default pinyin tokeniser then leaves ASCII text as a list of characters; tokenisers are out of scope and the
    fake model used by the tests ignores token identity)."""
    ref = load()
    if getattr(ref, "infer", None) is not None:
        return ref.infer
    mpl = _mod("matplotlib", use=lambda *a, **k: None)
    mpl.pylab = _mod("matplotlib.pylab")
    _mod("pydub", AudioSegment=object, silence=types.SimpleNamespace())
    _mod("vocos", Vocos=object)
    if "transformers" not in sys.modules:  # only `pipeline` (Whisper ASR, out of scope) is imported from it
        _mod("transformers", pipeline=None)
    sys.modules["rjieba"].cut = lambda t: [t]
    sys.modules["pypinyin"].lazy_pinyin = lambda seg, **k: list(seg)
    sys.modules["pypinyin"].Style = types.SimpleNamespace(TONE3=0)
    model_pkg = sys.modules["f5_tts.model"]
    model_pkg.CFM = ref.CFM
    for pkg, rel in (("f5_tts.infer", "f5_tts/infer"), ("f5_tts.train", "f5_tts/train"),
                     ("f5_tts.train.datasets", "f5_tts/train/datasets")):
        if pkg not in sys.modules:
            m = types.ModuleType(pkg)
            m.__path__ = [os.path.join(REF_SRC, rel)]
            sys.modules[pkg] = m
    ref.infer = importlib.import_module("f5_tts.infer.utils_infer")
    return ref.infer


def load_eval():
    """Imports the reference's evaluation helpers (src/f5_tts/eval/utils_eval.py) for pinning the batch bucketing of
    `get_inference_prompt`.  Its file / DSP dependencies are the inert torchaudio stand-in plus whatever the caller
    monkeypatches onto the returned module (`torchaudio.load`, `MelSpec`)."""
    ref = load_infer() and load()
    if getattr(ref, "eval", None) is not None:
        return ref.eval
    for pkg, rel in (("f5_tts.eval", "f5_tts/eval"),):
        if pkg not in sys.modules:
            m = types.ModuleType(pkg)
            m.__path__ = [os.path.join(REF_SRC, rel)]
            sys.modules[pkg] = m
    ref.eval = importlib.import_module("f5_tts.eval.utils_eval")
    return ref.eval
