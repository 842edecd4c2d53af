"""Cost of a software grid barrier (atomic arrive + bounded spin + agent-scope fences) between phases of a persistent kernel,
with and without a per-phase payload written by every block and read by a block on another XCD."""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from f5_tts_amd import _lib
lib = _lib.load()
torch.cuda.init()
s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
for blocks in (64, 256):
    for payload in (0, 4, 32):   # floats per thread per phase: 0, 8 KB, 64 KB per block
        us, bad = C.c_float(0), C.c_int32(0)
        rc = lib.f5x_grid_barrier_probe(blocks, 200, payload, C.byref(us), C.byref(bad), s)
        print(f"blocks {blocks} payload {payload * 2048} B/block: {us.value:.2f} us per phase+barrier, errors {bad.value}, rc {rc}", flush=True)
