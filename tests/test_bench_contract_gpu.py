"""The driver's contract for bench.py, checked on the GPU box: ONE JSON line on stdout with the required keys, a `roofline` object
(bound / achieved / peak / unit / frac / traffic) and -- in the default run -- `cpu_baseline`; here a short run without the CPU leg."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_line_contract():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--no-cpu-baseline",
                        "--no-precisions", "--no-c3", "--c4-utts", "24", "--setup-runs", "1"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, "exactly one line on stdout"
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["higher_is_better"] is True and d["vs_baseline"] is None
    assert d["scaling"] == "weak" and d["data"] == "synthetic" and d["dtype"] == "f16" and d["precision_mode"] == "f16p"
    assert "workload" in d["config"] and "model" not in d["config"]
    assert abs(d["value"] - 8.192 / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]          # 768 generated frames per step
    rf = d["roofline"]
    assert rf["bound"] == "mfma" and rf["unit"] == "TFLOP/s" and rf["peak"] == 2500.0
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-9 and 0.02 < rf["frac"] < 1.0
    assert rf["traffic"] is None or rf["traffic"] > 0
    assert "clocks" in d and (d["clocks"] is None or 500 < d["clocks"]["sclk_mhz_avg"] <= 2600)
    c4 = d["c4"]
    assert c4["scaling"] == "strong" and c4["n_gpus"] == 1 and len(c4["shard8_wall_sec"]) == 8 and c4["predicted_scaling_8"] > 1.0
