"""The four GEMMs of a DiT block (bf16, real epilogues, block order, cold weights) at the row counts of the benchmark configs:
2,048 (C2), 16,384 and 32,768 (one chunk of C3 / C4).  Per GEMM: us per launch and TFLOP/s; plus the plain-store ping-pong kernel
with and without its epilogue (cfg 20 vs 420) = what the stores cost by omission.
usage: python tools/block_gemm_time.py [cfg ...]      (cfg -1 = the product's tile choice)"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from f5_tts_amd import _lib
lib = _lib.load()
s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
torch.zeros(1, device="cuda:0")
fn = lib.f5x_block_gemm_time
fn.restype = C.c_int32
fn.argtypes = [C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_float), C.c_void_p]
cfgs = [int(a) for a in sys.argv[1:]] or [-1]
FL = lambda m: [2.0 * m * 3072 * 1024, 2.0 * m * 1024 * 1024, 2.0 * m * 2048 * 1024, 2.0 * m * 1024 * 2048]
for m in (2048, 16384, 32768):
    for cfg in cfgs:
        us = (C.c_float * 4)()
        rc = fn(m, cfg, 10, us, s)
        if rc != 0:
            print("rows", m, "cfg", cfg, "error", lib.f5_last_error()); continue
        tot = sum(us)
        print(f"rows {m:6d} cfg {cfg:3d}: " + " | ".join(f"{n} {u:7.1f} us {f / u / 1e6:5.0f} TF" for n, u, f in zip(("qkv", "out", "ff1", "ff2"), us, FL(m)))
              + f" | sum {tot:7.1f} us {sum(FL(m)) / tot / 1e6:5.0f} TF", flush=True)
g2 = lib.f5x_gemm2
g2.restype = C.c_int32
g2.argtypes = [C.c_int32] + [C.c_void_p] * 3 + [C.c_int32, C.c_void_p] + [C.c_int32] * 5 + [C.POINTER(C.c_float), C.c_void_p]
lib.f5x_set_out_bf16(1)
for (m, n, k) in ((32768, 1024, 1024), (32768, 2048, 1024), (32768, 3072, 1024), (32768, 1024, 2048), (16384, 1024, 1024)):
    A = torch.randn(m, k, device="cuda:0"); W = torch.randn(n, k, device="cuda:0") / k ** 0.5; out = torch.zeros(m, n, device="cuda:0")
    row = []
    for cfg in (20, 420, 13, 413, 2, 16, 18, 17):
        us = C.c_float(0)
        rc = g2(1, A.data_ptr(), W.data_ptr(), None, 0, out.data_ptr(), m, n, k, cfg, 20, C.byref(us), s)
        row.append(f"cfg {cfg}: {us.value:7.1f} us {2.0 * m * n * k / us.value / 1e6:5.0f} TF" if rc == 0 else f"cfg {cfg}: rc {rc}")
    print(f"{m}x{n}x{k} bf16 store: " + " | ".join(row), flush=True)
