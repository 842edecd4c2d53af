"""Correctness + timing of the v3 ping-pong GEMM (cfg 20) against the v2 configs at the many-row shapes of C3 / C4.
usage: python tools/gemm3_check.py [M=16384] [cfgs=13,20]"""
import ctypes as C
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from f5_tts_amd import _lib

lib = _lib.load()
fn = lib.f5x_gemm2
fn.restype = C.c_int32
fn.argtypes = [C.c_int32] + [C.c_void_p] * 3 + [C.c_int32, C.c_void_p] + [C.c_int32] * 5 + [C.POINTER(C.c_float), C.c_void_p]
dev = "cuda:0"
s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
M = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
cfgs = [int(c) for c in sys.argv[2].split(",")] if len(sys.argv) > 2 else [13, 20]
shapes = [("qkv", M, 3072, 1024), ("out", M, 1024, 1024), ("ff1", M, 2048, 1024), ("ff2", M, 1024, 2048),
          ("edge", 700, 520, 1024), ("tiny", 300, 100, 768)]
for prec, pname, tol in ((1, "bf16", 1.5e-2), (2, "f16", 2e-3), (0, "f32", 2e-5)):
    for name, m, n, k in shapes:
        if prec == 0 and m > 4096:
            m = 4096
        g = torch.Generator().manual_seed(m + n)
        A = torch.randn(m, k, generator=g).to(dev)
        W = (torch.randn(n, k, generator=g) / k ** 0.5).to(dev)
        b = torch.randn(n, generator=g).to(dev)
        ref = F.linear(A, W, b)
        row = []
        for cfg in cfgs:
            out = torch.zeros(m, n, device=dev)
            us = C.c_float(0)
            rc = fn(prec, A.data_ptr(), W.data_ptr(), b.data_ptr(), 0, out.data_ptr(), m, n, k, cfg, 30 if m >= 2048 else 0, C.byref(us), s)
            if rc != 0:
                row.append(f"[{cfg}] ERR {lib.f5_last_error().decode()[:60]}")
                continue
            err = ((out - ref).abs().max() / ref.abs().max()).item()
            ok = "ok " if err < tol else f"BAD({err:.1e})"
            tf = 2.0 * m * n * k / us.value / 1e6 if us.value > 0 else 0
            # run-to-run determinism (race detector)
            out2 = torch.zeros(m, n, device=dev)
            fn(prec, A.data_ptr(), W.data_ptr(), b.data_ptr(), 0, out2.data_ptr(), m, n, k, cfg, 0, C.byref(us), s)
            det = "" if torch.equal(out, out2) else " NONDET"
            row.append(f"[{cfg}] {ok}{det} {us.value:7.1f}us {tf:7.1f}TF")
        print(pname, name, (m, n, k), " | ".join(row), flush=True)
