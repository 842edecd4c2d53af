"""What rotary on every head (pe_attn_head = None: F5TTS_v1_Base, E2-TTS UNetT) costs the QKV projection's epilogue: the block's QKV
GEMM (bf16, real epilogue, block order, cold weights: f5x_block_gemm_time) with rotary on 1 head and on all 16."""
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
for m in (2048, 16384, 32768):
    row = []
    for pe in (1, 16, 1, 16):
        os.environ["F5X_QKV_PE"] = str(pe)
        us = (C.c_float * 4)()
        assert fn(m, -1, 20, us, s) == 0
        row.append(f"pe {pe:2d}: qkv {us[0]:7.1f} us")
    print(f"rows {m:6d}: " + " | ".join(row), flush=True)
