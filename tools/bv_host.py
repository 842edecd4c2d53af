import sys, time, importlib, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
P = importlib.import_module("korean-f5-tts_amd")
voc = P.BigVGAN(P.config.BIGVGAN_V2_24K, precision="f16x3").init_synthetic(seed=1).to("cuda:0")
mel = torch.randn(1, 100, 768, device="cuda:0")
for _ in range(2): voc(mel)
torch.cuda.synchronize()
for _ in range(3):
    t0 = time.perf_counter(); w = voc(mel); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"host return {1e3*(t1-t0):.2f} ms, device done {1e3*(t2-t0):.2f} ms")
mel8 = torch.randn(8, 100, 768, device="cuda:0")
voc(mel8); torch.cuda.synchronize()
t0 = time.perf_counter(); w = voc(mel8); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print(f"B=8: host return {1e3*(t1-t0):.2f} ms, device done {1e3*(t2-t0):.2f} ms")
