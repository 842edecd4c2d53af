"""Correctness + timing sweep of the v2 (glds ring) GEMM configs through the experimental f5x_gemm2 entry."""
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
M = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
cfgs = [int(c) for c in sys.argv[2].split(",")] if len(sys.argv) > 2 else list(range(10))
if len(sys.argv) > 3:
    lib.f5x_set_xcd_mode(int(sys.argv[3]))
if len(sys.argv) > 4:
    lib.f5x_set_cold_weights(int(sys.argv[4]))
names = {0: "128x128 4w ns3", 1: "128x128 4w ns4", 2: "128x128 8w(2x4) ns4", 3: "128x64 4w ns4", 4: "128x64 4w ns3",
         5: "64x64 4w ns4", 6: "256x128 8w ns3", 7: "128x128 8w(4x2) ns3", 8: "64x64 4w ns3", 9: "128x64 8w ns4"}
shapes = [("qkv", M, 3072, 1024), ("out", M, 1024, 1024), ("ff1", M, 2048, 1024), ("ff2", M, 1024, 2048), ("odd", 300, 100, 768)]
for prec, pname, tol in ((1, "bf16", 1.5e-2), (0, "f32", 2e-5)):
    for name, m, n, k in shapes:
        g = torch.Generator().manual_seed(m + n)
        A = torch.randn(m, k, generator=g).to(dev)
        W = (torch.randn(n, k, generator=g) / k ** 0.5).to(dev)
        b = torch.randn(n, generator=g).to(dev)
        ref = F.linear(A.double(), W.double(), b.double()).float()
        row = []
        for cfg in cfgs:
            out = torch.zeros(m, n, device=dev)
            us = C.c_float(0)
            rc = fn(prec, A.data_ptr(), W.data_ptr(), b.data_ptr(), 0, out.data_ptr(), m, n, k, cfg, 64 if name != "odd" else 0, C.byref(us), s)
            if rc != 0:
                row.append(f"[{cfg}] ERR {lib.f5_last_error().decode()[:60]}")
                continue
            err = ((out - ref).abs().max() / ref.abs().max()).item()
            ok = "ok " if err < tol else f"BAD({err:.1e})"
            tf = 2.0 * m * n * k / us.value / 1e6 if us.value > 0 else 0
            row.append(f"[{cfg}] {ok} {us.value:6.1f}us {tf:6.1f}TF")
        print(pname, name, (m, n, k), " | ".join(row), flush=True)
print({k: v for k, v in names.items() if k in cfgs})
