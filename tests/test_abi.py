"""CPU: the C-ABI library loads, exports every symbol include/f5_hip.h declares, and validates arguments without a
GPU (no compute calls here)."""
import ctypes as C
import os
import re

import pytest

from conftest import ROOT

import f5_tts_amd as P
from f5_tts_amd import _lib


def declared_functions():
    src = open(os.path.join(ROOT, "include", "f5_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    names = re.findall(r"\b(f5k?_[a-z0-9_]+)\s*\(", src)
    return sorted(set(names))


def test_library_is_built_in_tree():
    assert os.path.exists(_lib.LIB_PATH), "run `python korean-f5-tts_amd/build.py` (or __graft_entry__.build())"


def test_every_declared_symbol_is_exported_and_bound():
    lib = _lib.load()
    decl = declared_functions()
    assert len(decl) >= 20
    for name in decl:
        assert hasattr(lib, name), f"{name} declared in include/f5_hip.h but not exported by libf5hip.so"
        assert name in _lib.SIGNATURES, f"{name} has no ctypes signature in _lib.py"
    assert sorted(_lib.SIGNATURES) == decl


def test_version_and_argument_validation_without_gpu():
    lib = _lib.load()
    assert b"gfx950" in lib.f5_version()
    cfg = _lib.f5_config()
    h = C.c_void_p()
    cfg.dim, cfg.depth, cfg.heads, cfg.dim_head, cfg.ff_dim = 1024, 22, 16, 32, 2048  # dim_head must be 64
    cfg.text_dim, cfg.mel_dim = 512, 100
    rc = lib.f5_create(C.byref(cfg), C.byref(h))
    assert rc == -1 and b"dim_head" in lib.f5_last_error()
    cfg.dim_head = 64
    cfg.dim = 1000
    assert lib.f5_create(C.byref(cfg), C.byref(h)) == -1
    with pytest.raises(_lib.F5Error):
        _lib.check(lib.f5_create(None, C.byref(h)), "f5_create")


def test_product_has_no_cpu_path():
    """Backbone / vocoder objects refuse to run off-GPU instead of falling back."""
    import torch
    m = P.DiT(**P.config.F5TTS_TINY, text_num_embeds=40, mel_dim=100).init_synthetic()
    with pytest.raises(RuntimeError):
        m.engine()
    v = P.Vocos(P.config.VOCOS_TINY).init_synthetic()
    with pytest.raises(RuntimeError):
        v.decode(torch.zeros(1, 100, 8))


def test_product_does_not_import_the_oracle():
    pkg = os.path.join(ROOT, "korean-f5-tts_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(dp, f), encoding="utf-8", errors="ignore").read()
                assert "oracle" not in txt.replace("the oracle", "").replace("CPU oracle", ""), f"{f} mentions oracle/"
