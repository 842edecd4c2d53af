"""Bare v_mfma_f32_16x16x32_bf16 issue rate and the shader clock under that load (s_memtime vs the 100 MHz s_memrealtime):
what "dense bf16 peak" this part sustains in practice, to read the roofline fractions against."""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from f5_tts_amd import _lib
lib = _lib.load()
torch.cuda.init()
s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
for blocks, threads, iters in ((256, 256, 20000), (256, 512, 20000), (256, 512, 200000), (1024, 512, 50000)):
    cyc, mhz, tf = C.c_double(0), C.c_double(0), C.c_double(0)
    rc = lib.f5x_mfma_rate_probe(blocks, threads, iters, C.byref(cyc), C.byref(mhz), C.byref(tf), s)
    print(f"blocks {blocks} x {threads // 64} waves, {iters * 8} MFMAs/wave: {cyc.value:.2f} shader clocks per MFMA per SIMD, "
          f"shader clock {mhz.value:.0f} MHz, {tf.value:.0f} TFLOP/s (rc {rc})", flush=True)
