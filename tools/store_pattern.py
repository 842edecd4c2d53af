"""What the store pattern of a GEMM epilogue costs (csrc/kapi_diag.hip::store_pattern_kernel): MFMA-fragment order (32 / 64 contiguous
bytes per row per instruction) against row-major full lines, bf16 and f32 (with and without reading the same addresses first)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from f5_tts_amd import _lib
lib = _lib.load()
s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
torch.zeros(1, device="cuda:0")
fn = lib.f5x_store_pattern_probe
fn.restype = C.c_int32
fn.argtypes = [C.c_int32] * 5 + [C.POINTER(C.c_float), C.c_void_p]
names = {0: "bf16 fragment order (16 rows x 32 B)", 1: "bf16 row-major (8 rows x 128 B)", 2: "f32 fragment order (16 rows x 64 B)", 3: "f32 row-major (4 rows x 256 B)"}
for (m, n) in ((32768, 1024), (32768, 2048), (32768, 3072), (16384, 1024)):
    for mode in (0, 1, 2, 3):
        for rd in ((0,) if mode < 2 else (0, 1)):
            us = C.c_float(0)
            rc = fn(m, n, mode, rd, 20, C.byref(us), s)
            b = m * n * (2 if mode < 2 else 4) * (2 if rd else 1)
            print(f"{m}x{n} {names[mode]}{' read+write' if rd else ' write'}: {us.value:7.1f} us  {b / us.value / 1e6:6.2f} TB/s" if rc == 0 else f"rc {rc}", flush=True)
