"""Builds libf5hip.so (the C-ABI engine, include/f5_hip.h) for gfx950 with hipcc, in-tree.

    python korean-f5-tts_amd/build.py [--force]

hipcc cross-compiles without a GPU.  Objects go to build/ (git-ignored); the .so lands next to this file so that it
travels with the repository snapshot to the GPU box.
"""
from __future__ import annotations

import hashlib
import os
import subprocess
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
ROOT = os.path.dirname(HERE)
LIB = os.path.join(HERE, "libf5hip.so")
OBJDIR = os.path.join(ROOT, "build", "f5hip")
SOURCES = ["engine.hip", "engine_bf16.hip", "engine_f16.hip", "engine_f32.hip", "vocos.hip", "bigvgan.hip", "kapi.hip", "kapi_diag.hip", "mel.hip"]
HEADERS = sorted(f for f in os.listdir(CSRC) if f.endswith(".h")) + ["../../include/f5_hip.h"]
# -ffp-contract=off: a*b + c is evaluated as written in every kernel.  With the default (fast) hipcc fuses multiply-adds
# where its instruction selection happens to see them, which differs between instantiations of one epilogue in different
# GEMM tile shapes: the f32 results then differ in the last bit, and after 22 layers of 16-bit re-rounding by ~1e-2 --
# a result must not depend on which tile (i.e. on how many utterances share a batch) computed it.
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wall", "-Wno-unused-function",
         "-Wno-unused-variable"]


def _hipcc() -> str:
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


def _digest(sources) -> str:
    h = hashlib.sha256()
    for f in list(sources) + HEADERS:
        with open(os.path.join(CSRC, f), "rb") as fh:
            h.update(fh.read())
    h.update(" ".join(FLAGS).encode())
    return h.hexdigest()


def _read(path: str) -> str:
    try:
        with open(path) as fh:
            return fh.read()
    except OSError:
        return ""


def build(force: bool = False, verbose: bool = True) -> str:
    """Compiles the sources whose text (or any header, or the flags) changed since their object was built, then links."""
    os.makedirs(OBJDIR, exist_ok=True)
    stamp = os.path.join(OBJDIR, "digest.txt")
    dig = _digest(SOURCES)
    if not force and os.path.exists(LIB) and _read(stamp) == dig:
        return LIB
    hipcc = _hipcc()
    t0 = time.time()
    procs = []
    objs = []
    for src in SOURCES:
        obj = os.path.join(OBJDIR, src.replace(".hip", ".o"))
        objs.append(obj)
        sdig = _digest([src])
        if not force and os.path.exists(obj) and _read(obj + ".digest") == sdig:
            continue
        cmd = [hipcc, *FLAGS, "-c", os.path.join(CSRC, src), "-o", obj]
        procs.append((src, obj, sdig, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    for src, obj, sdig, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{out}")
        with open(obj + ".digest", "w") as fh:
            fh.write(sdig)
        if verbose and out.strip():
            print(out)
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stdout}")
    with open(stamp, "w") as fh:
        fh.write(dig)
    if verbose:
        print(f"built {LIB} in {time.time() - t0:.1f}s ({len(procs)} of {len(SOURCES)} sources compiled)")
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
