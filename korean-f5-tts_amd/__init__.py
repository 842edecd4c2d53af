"""MI355X-native F5-TTS inference engine (CFM.sample -> DiT x NFE -> Vocos) behind the reference's Python API.

Host side = Python on PyTorch-ROCm (device memory, streams, torch.distributed); compute = hand-written HIP kernels
for gfx950 in csrc/, reached through the C-ABI declared in include/f5_hip.h (ctypes, no torch types cross it).
"""
from . import config, mel, utils, weights  # noqa: F401

__all__ = ["config", "weights", "mel", "utils", "CFM", "DiT", "UNetT", "Vocos", "BigVGAN", "lib", "infer", "dist", "batching"]


def __getattr__(name):  # lazy: importing the package must not require the built library (CPU-only tooling)
    if name in ("CFM",):
        from .cfm import CFM
        return CFM
    if name in ("DiT", "UNetT"):
        from . import backbones
        return getattr(backbones, name)
    if name == "Vocos":
        from .vocos import Vocos
        return Vocos
    if name == "BigVGAN":
        from .bigvgan import BigVGAN
        return BigVGAN
    if name == "lib":
        from . import _lib
        return _lib
    if name in ("infer", "dist", "batching", "engine", "bigvgan", "vocos"):
        import importlib
        return importlib.import_module("." + name, __name__)
    raise AttributeError(name)
