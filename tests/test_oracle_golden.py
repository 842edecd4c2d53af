"""CPU: the oracle (oracle/f5_oracle.py) against the golden vectors produced by RUNNING THE REFERENCE
(oracle/make_golden.py, reference imported from /root/reference in the build container)."""
import pytest
import torch

from conftest import load_golden, synthetic_weights
from oracle import f5_oracle as O

TOL = 2e-5  # fp32 CPU vs fp32 CPU, different op order only

SAMPLE_CASES = ["sample_b1_nfe16", "sample_b3_masked", "sample_b3_attnmask", "sample_b1_editmask",
                "sample_b1_norefaudio", "sample_b2_v1arch", "sample_b2_options", "sample_b1_options_pe1", "sample_b1_nocfg_linspace", "sample_b1_textclamp",
                "sample_b1_duplicate", "sample_unett_b2"]


def run_oracle_sample(meta, a, sd, **over):
    dur = meta["duration"]
    dur = dur if isinstance(dur, int) else torch.tensor(dur)
    kw = dict(steps=meta["steps"], cfg_strength=meta["cfg_strength"], sway_sampling_coef=meta["sway"],
              seed=meta["seed"], use_epss=meta["use_epss"], no_ref_audio=meta["no_ref_audio"],
              backbone=meta["backbone"])
    if meta["lens"] is not None:
        kw["lens"] = torch.tensor(meta["lens"])
    if "edit_mask" in a:
        kw["edit_mask"] = a["edit_mask"]
    kw.update(over)
    if meta.get("duplicate_test"):
        kw.update(duplicate_test=True, t_inter=meta["t_inter"])
    return O.sample(sd, meta["arch"], a["cond"], a["text"], dur, **kw)


@pytest.mark.parametrize("name", SAMPLE_CASES)
def test_sample_matches_reference_vectors(name):
    meta, a = load_golden(name)
    sd = synthetic_weights(meta)
    out, traj = run_oracle_sample(meta, a, sd)
    assert out.shape == a["out"].shape and traj.shape == a["traj"].shape
    assert (traj - a["traj"]).abs().max() < TOL
    assert (out - a["out"]).abs().max() < TOL
    # the ODE must actually move (guards against the zero-init trap, dit.py:214-224)
    assert (a["traj"][-1] - a["traj"][0]).abs().max() > 0.1


def test_time_grids():
    _, g = load_golden("time_grids")
    assert torch.equal(O.time_grid(16, -1.0, True), g["grid_16_-1.0_1"])
    assert torch.equal(O.time_grid(32, -1.0, True), g["grid_32_-1.0_1"])
    assert torch.equal(O.time_grid(7, None, True), g["grid_7_None_1"])
    assert torch.equal(O.time_grid(8, 0.5, False), g["grid_8_0.5_0"])


@pytest.mark.parametrize("name", ["dit_forward_taps", "dit_forward_taps_masked"])
def test_dit_forward_taps(name):
    meta, a = load_golden(name)
    sd = synthetic_weights(meta)
    taps = {}
    mask = a.get("mask")
    out = O.dit_forward(sd, meta["arch"], a["x"], a["cond"], a["text"], a["time"], mask=mask, cfg_infer=True,
                        cache=O.TextCache(), taps=taps)
    B = a["x"].shape[0]
    assert (out - a["out"]).abs().max() < TOL
    assert (taps["text_cond"] - a["text_cond"]).abs().max() < TOL
    assert (taps["text_uncond"] - a["text_uncond"]).abs().max() < TOL
    assert (taps["time_embed"][:B] - a["time_embed"]).abs().max() < TOL
    assert (taps["input_embed"][B:] - a["input_embed_last"]).abs().max() < TOL  # hook saw the uncond call last
    for i in range(meta["arch"]["depth"]):
        assert (taps[f"block{i}"] - a[f"block{i}"]).abs().max() < TOL


def test_vocos_istft_matches_torch_and_head_formula():
    """iSTFT-head arithmetic (export_vocoder_to_onnx.py:51-59) + torch.istft; backbone itself is parity-unpinned."""
    import f5_tts_amd as P
    v = P.config.VOCOS_TINY
    V = P.weights.synthetic_state_dict(P.weights.vocos_param_shapes(v), seed=3)
    mel = torch.randn(2, 100, 37)
    wav = O.vocos_decode(V, mel)
    assert wav.shape == (2, 36 * 256)
    assert torch.isfinite(wav).all() and wav.abs().max() > 1e-3
    # explicit overlap-add restatement of torch.istft(center=True) used by the HIP path's design
    re, im = O.istft_head_spec(V, O.vocos_backbone(V, mel))
    frames = torch.fft.irfft(torch.complex(re, im).transpose(1, 2), n=1024) * torch.hann_window(1024)
    T = frames.shape[1]
    y = torch.zeros(2, (T - 1) * 256 + 1024)
    env = torch.zeros((T - 1) * 256 + 1024)
    for t in range(T):
        y[:, t * 256:t * 256 + 1024] += frames[:, t]
        env[t * 256:t * 256 + 1024] += torch.hann_window(1024) ** 2
    y = (y / env)[:, 512:-512]
    assert (y - wav).abs().max() < 1e-4 * max(1.0, wav.abs().max().item())
