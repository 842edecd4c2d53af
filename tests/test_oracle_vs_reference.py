"""CPU, build container only: the oracle against the reference itself, imported live from /root/reference
(skipped on the GPU box, where the reference does not exist)."""
import pytest
import torch

import f5_tts_amd as P
from oracle import f5_oracle as O
from oracle import ref_harness as rh

pytestmark = pytest.mark.skipif(not rh.available(), reason="reference tree not present (GPU box)")


def test_base_arch_cfg_forward_matches():
    """F5-TTS Base (configs/F5TTS_Base.yaml:24-36), one packed CFG forward at N=192."""
    arch = P.config.F5TTS_BASE
    nv = P.config.VOCAB_SIZE + 1  # load_model passes vocab_size + 1 (utils_infer.py:313-317)
    sd = P.weights.synthetic_state_dict(P.weights.dit_param_shapes(arch, nv))
    ref = rh.build_reference_cfm(dict(arch), nv)
    ref.transformer.load_state_dict(sd, strict=True)
    g = torch.Generator().manual_seed(5)
    x = torch.randn(1, 192, 100, generator=g)
    cond = torch.randn(1, 192, 100, generator=g)
    cond[:, 60:] = 0
    text = torch.randint(0, nv - 1, (1, 40), generator=g)
    t = torch.tensor(0.25)
    with torch.no_grad():
        r = ref.transformer(x=x, cond=cond, text=text, time=t, cfg_infer=True, cache=False)
    o = O.dit_forward(sd, arch, x, cond, text, t, cfg_infer=True)
    assert (r - o).abs().max() < 5e-5
    assert r.abs().max() > 0.05


def test_tiny_sample_variants_match():
    arch = dict(P.config.F5TTS_TINY, attn_mask_enabled=True)
    nv = 33
    sd = P.weights.synthetic_state_dict(P.weights.dit_param_shapes(arch, nv), seed=4)
    ref = rh.build_reference_cfm(dict(arch), nv)
    ref.transformer.load_state_dict(sd, strict=True)
    g = torch.Generator().manual_seed(9)
    cond = torch.randn(2, 28, 100, generator=g)
    text = torch.randint(0, nv, (2, 15), generator=g)
    text[1, 9:] = -1
    kw = dict(lens=torch.tensor([28, 19]), steps=10, cfg_strength=2.0, sway_sampling_coef=-1.0, seed=11)
    r_out, r_traj = ref.sample(cond, text, torch.tensor([60, 47]), **kw)
    o_out, o_traj = O.sample(sd, arch, cond, text, torch.tensor([60, 47]), **kw)
    assert (r_traj - o_traj).abs().max() < 2e-5 and (r_out - o_out).abs().max() < 2e-5


def test_noise_draw_matches_reference_rule():
    """cfm.py:196-201: same seed for every sample -> identical noise prefixes across the batch."""
    y0 = O.draw_noise(torch.tensor([7, 4]), 100, seed=3)
    assert torch.equal(y0[0, :4], y0[1, :4]) and torch.all(y0[1, 4:] == 0)
