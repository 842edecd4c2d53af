"""What the epilogue costs at the C2 row count (M = 2,048; cold weights): each v2 tile with a plain bf16 / f32 store against the same
kernel without its epilogue (gemm2.h MODE 4)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from f5_tts_amd import _lib
lib = _lib.load()
fn = lib.f5x_gemm2
fn.restype = C.c_int32
fn.argtypes = [C.c_int32] + [C.c_void_p] * 3 + [C.c_int32, C.c_void_p] + [C.c_int32] * 5 + [C.POINTER(C.c_float), C.c_void_p]
s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
lib.f5x_set_cold_weights(1)
for name, m, n, k, cfg, cfg0 in (("out 128x64", 2048, 1024, 1024, 9, 409), ("ff2 128x64", 2048, 1024, 2048, 9, 409), ("ff1 128x128", 2048, 2048, 1024, 2, 402),
                                 ("qkv 128x128", 2048, 3072, 1024, 2, 402)):
    A = torch.randn(m, k, device="cuda:0"); W = torch.randn(n, k, device="cuda:0") / k ** 0.5; b = torch.randn(n, device="cuda:0")
    out = torch.zeros(m, n, device="cuda:0")
    row = []
    for obf in (1, 0):
        lib.f5x_set_out_bf16(obf)
        for c in (cfg, cfg0):
            us = C.c_float(0)
            fn(1, A.data_ptr(), W.data_ptr(), b.data_ptr(), 0, out.data_ptr(), m, n, k, c, 64, C.byref(us), s)
            row.append(f"{'bf16' if obf else 'f32'} out cfg {c}: {us.value:6.2f} us")
    print(name, (m, n, k), " | ".join(row), flush=True)
