import json
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    meta = json.loads(bytes(z["meta_json"]).decode())
    arrs = {k: torch.from_numpy(z[k]) for k in z.files if k != "meta_json"}
    return meta, arrs


@pytest.fixture(scope="session")
def golden():
    return load_golden


def synthetic_weights(meta):
    import f5_tts_amd as P
    fn = P.weights.unett_param_shapes if meta.get("backbone", "DiT") == "UNetT" else P.weights.dit_param_shapes
    sd = P.weights.synthetic_state_dict(fn(meta["arch"], meta["nvocab"]), seed=meta.get("wseed", 0))
    chk = float(sum(v.double().abs().sum().item() for v in sd.values()))
    assert abs(chk - meta["weights_checksum"]) <= 1e-6 * abs(chk), "synthetic weight generator drifted from the fixtures"
    return sd
