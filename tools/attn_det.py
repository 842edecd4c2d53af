import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from f5_tts_amd import _lib
from gpu_util import k_attention
lib = _lib.load()
g = torch.Generator().manual_seed(0)
q, k, v = (torch.randn(2, 16, 1024, 64, generator=g).to("cuda:0") for _ in range(3))
for var in (0, 1, 2, 3):
    lib.f5x_set_attn_variant(var)
    ref = k_attention("bf16", q, k, v)
    bad = sum(0 if torch.equal(k_attention("bf16", q, k, v), ref) else 1 for _ in range(20))
    print("variant", var, "nondeterministic runs:", bad, "/ 20", flush=True)
