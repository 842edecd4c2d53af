"""TEST INFRASTRUCTURE ONLY -- generates tests/golden/*.npz by RUNNING THE REFERENCE (imported via ref_harness).

Run in the build container only (the reference never travels to the GPU box):
    python -m oracle.make_golden
Every output tensor in the fixtures comes from the reference's own code (cfm.py / dit.py / unett.py / modules.py);
inputs and synthetic-weight seeds are recorded beside them so that the oracle and the HIP path can be fed the same
problem.  Weights are NOT stored: they are regenerated from (seed, tensor name) by
korean-f5-tts_amd/weights.py::synthetic_state_dict and a checksum is stored to detect generator drift.
"""
from __future__ import annotations

import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import f5_tts_amd as P  # noqa: E402
from oracle import ref_harness as rh  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
NVOCAB = 40


def weights_checksum(sd):
    return float(sum(v.double().abs().sum().item() for v in sd.values()))


ONLY = set(sys.argv[1:])   # optional fixture names: regenerate just these


def save(name, meta, **arrays):
    if ONLY and name not in ONLY:
        return
    arrs = {k: (v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else np.asarray(v)) for k, v in arrays.items()}
    arrs["meta_json"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **arrs)
    print("wrote", name, {k: v.shape for k, v in arrs.items() if k != "meta_json"})


def build(arch, backbone="DiT", seed=0):
    shapes = (P.weights.dit_param_shapes if backbone == "DiT" else P.weights.unett_param_shapes)(arch, NVOCAB)
    sd = P.weights.synthetic_state_dict(shapes, seed=seed)
    model = rh.build_reference_cfm({k: v for k, v in arch.items()}, NVOCAB, backbone=backbone)
    model.transformer.load_state_dict(sd, strict=True)
    return model, sd


def sample_case(name, arch, *, B, cond_len, nt, duration, lens=None, steps, cfg_strength=2.0, sway=-1.0, seed=7,
                use_epss=True, no_ref_audio=False, edit_mask=None, backbone="DiT", text_pad=None, wseed=0,
                duplicate_test=False, t_inter=0.1):
    model, sd = build(arch, backbone, wseed)
    g = torch.Generator().manual_seed(1000 + len(name))
    cond = torch.randn(B, cond_len, 100, generator=g)
    text = torch.randint(0, NVOCAB, (B, nt), generator=g)
    if text_pad:
        for b, n_valid in enumerate(text_pad):
            text[b, n_valid:] = -1
    kw = dict(steps=steps, cfg_strength=cfg_strength, sway_sampling_coef=sway, seed=seed, use_epss=use_epss,
              no_ref_audio=no_ref_audio)
    if lens is not None:
        kw["lens"] = torch.tensor(lens)
    if edit_mask is not None:
        kw["edit_mask"] = edit_mask
    if duplicate_test:
        kw.update(duplicate_test=True, t_inter=t_inter)
    dur = duration if isinstance(duration, int) else torch.tensor(duration)
    out, traj = model.sample(cond, text, dur, **kw)
    meta = dict(arch=arch, backbone=backbone, nvocab=NVOCAB, wseed=wseed, steps=steps, cfg_strength=cfg_strength,
                sway=sway, seed=seed, use_epss=use_epss, no_ref_audio=no_ref_audio,
                duration=duration, lens=lens, weights_checksum=weights_checksum(sd),
                duplicate_test=duplicate_test, t_inter=t_inter)
    arrays = dict(cond=cond, text=text, out=out, traj=traj)
    if edit_mask is not None:
        arrays["edit_mask"] = edit_mask
    save(name, meta, **arrays)


def forward_taps_case(name, arch, B, N, nt, masked):
    """One DiT forward (cfg_infer packed) with intermediates captured by forward hooks on the reference modules."""
    model, sd = build(arch)
    tr = model.transformer
    g = torch.Generator().manual_seed(77)
    x = torch.randn(B, N, 100, generator=g)
    cond = torch.randn(B, N, 100, generator=g)
    cond[:, N // 3:] = 0
    text = torch.randint(0, NVOCAB, (B, nt), generator=g)
    time = torch.tensor(0.3183)
    mask = None
    if masked:
        lens = torch.tensor([N] + [N - 9 * (i + 1) for i in range(B - 1)])
        mask = torch.arange(N)[None] < lens[:, None]
    taps = {}

    def hook(key):
        def f(mod, inp, out):
            taps[key] = (out[0] if isinstance(out, tuple) else out).detach().clone()
        return f

    hs = [tr.time_embed.register_forward_hook(hook("time_embed")),
          tr.input_embed.register_forward_hook(hook("input_embed_last")),
          tr.input_embed.proj.register_forward_hook(hook("input_proj_last")),
          tr.transformer_blocks[0].attn_norm.register_forward_hook(hook("b0_attn_norm")),
          tr.transformer_blocks[0].attn.register_forward_hook(hook("b0_attn")),
          tr.transformer_blocks[0].ff.register_forward_hook(hook("b0_ff")),
          tr.norm_out.register_forward_hook(hook("norm_out"))]
    for i, blk in enumerate(tr.transformer_blocks):
        hs.append(blk.register_forward_hook(hook(f"block{i}")))
    with torch.no_grad():
        out = tr(x=x, cond=cond, text=text, time=time, mask=mask, cfg_infer=True, cache=True)
        taps["text_cond"] = tr.text_cond.clone()
        taps["text_uncond"] = tr.text_uncond.clone()
        tr.clear_cache()
    for h in hs:
        h.remove()
    meta = dict(arch=arch, nvocab=NVOCAB, wseed=0, weights_checksum=weights_checksum(sd), masked=masked)
    arrays = dict(x=x, cond=cond, text=text, time=time, out=out, **taps)
    if mask is not None:
        arrays["mask"] = mask
    save(name, meta, **arrays)



def bucketing_case(name="bucketing"):
    """Runs the reference's `get_inference_prompt` (eval/utils_eval.py:72-205) on synthetic "files": `torchaudio.load` and
    `MelSpec` are replaced by fakes that only carry lengths, so the fixture pins the duration formula, the bucket index,
    the frame-budget batching, the residual flush and the seeded shuffle."""
    if ONLY and name not in ONLY:
        return
    import random

    ev = rh.load_eval()
    rng = random.Random(12)
    words = ["hello", "speech", "flow", "matching", "korean", "안녕하세요", "텍스트", "voice", "a", "synthesis."]
    meta, nsamp = [], {}
    for i in range(57):
        secs = rng.uniform(1.2, 9.0)
        path = f"/fake/prompt_{i}.wav"
        nsamp[path] = int(secs * 24000)
        ptext = " ".join(rng.choice(words) for _ in range(rng.randint(3, 9)))
        gtext = " ".join(rng.choice(words) for _ in range(rng.randint(3, 14)))
        meta.append((f"utt{i}", ptext, path, gtext, "/fake/none.wav"))

    class FakeMel:
        def __init__(self, **kw):
            self.hop = kw["hop_length"]

        def __call__(self, audio):
            return torch.zeros(1, 100, audio.shape[-1] // self.hop + 1)   # center=True framing

    ev.torchaudio.load = lambda path: (torch.full((1, nsamp[path]), 0.2), 24000)
    ev.MelSpec = FakeMel
    cases = []
    for bs, speed in ((1, 1.0), (3000, 1.0), (6000, 0.8)):
        out = ev.get_inference_prompt(meta, speed=speed, tokenizer="char", infer_batch_size=bs, min_secs=1, max_secs=120)
        cases.append(dict(infer_batch_size=bs, speed=speed, min_secs=1, max_secs=120,
                          batches=[dict(utts=list(b[0]), ref_mel_lens=[int(v) for v in b[3]],
                                        total_mel_lens=[int(v) for v in b[4]], texts=list(b[5])) for b in out]))
    fixture = dict(note="eval/utils_eval.py:72-205 run on length-only fakes", hop_length=256, target_sample_rate=24000,
                   prompts=[dict(utt=m[0], prompt_text=m[1], nsamples=nsamp[m[2]], gt_text=m[3]) for m in meta],
                   cases=cases)
    with open(os.path.join(OUT, name + ".json"), "w") as fh:
        json.dump(fixture, fh, ensure_ascii=False, indent=0)
    print("wrote", name, [len(c["batches"]) for c in cases])


def main():
    os.makedirs(OUT, exist_ok=True)
    ref = rh.load()
    tiny = dict(P.config.F5TTS_TINY)
    tiny_v1 = dict(tiny, text_mask_padding=True, pe_attn_head=None)

    grids = {}
    for steps, sway, epss in ((16, -1.0, True), (32, -1.0, True), (7, None, True), (8, 0.5, False)):
        t = ref.utils.get_epss_timesteps(steps, device="cpu", dtype=torch.float32) if epss else \
            torch.linspace(0, 1, steps + 1)
        if sway is not None:
            t = t + sway * (torch.cos(torch.pi / 2 * t) - 1 + t)
        grids[f"grid_{steps}_{sway}_{int(epss)}"] = t
    save("time_grids", dict(note="cfm.py:211-216 evaluated by the reference's get_epss_timesteps / torch"), **grids)

    sample_case("sample_b1_nfe16", tiny, B=1, cond_len=24, nt=14, duration=64, steps=16)
    sample_case("sample_b3_masked", tiny, B=3, cond_len=30, nt=16, duration=[72, 41, 57], lens=[30, 17, 22],
                steps=6, text_pad=[16, 9, 12])
    sample_case("sample_b3_attnmask", dict(tiny, attn_mask_enabled=True), B=3, cond_len=30, nt=16,
                duration=[72, 41, 57], lens=[30, 17, 22], steps=6, text_pad=[16, 9, 12])
    em = torch.ones(1, 24, dtype=torch.bool)
    em[:, 10:16] = False
    sample_case("sample_b1_editmask", tiny, B=1, cond_len=24, nt=10, duration=24, steps=5, edit_mask=em)
    sample_case("sample_b1_norefaudio", tiny, B=1, cond_len=20, nt=12, duration=48, steps=5, no_ref_audio=True)
    sample_case("sample_b2_v1arch", tiny_v1, B=2, cond_len=20, nt=12, duration=[50, 33], lens=[20, 11], steps=7,
                text_pad=[12, 7])
    # the three DiT options no shipped config switches on (dit.py:160-161,166): RMSNorm on q / k, the long skip connection and the
    # zipvoice-style average upsampling of the text embedding (which needs text_mask_padding)
    tiny_opts = dict(tiny_v1, qk_norm="rms_norm", long_skip_connection=True, text_embedding_average_upsampling=True)
    sample_case("sample_b2_options", tiny_opts, B=2, cond_len=20, nt=12, duration=[50, 33], lens=[20, 11], steps=6,
                text_pad=[12, 7])
    sample_case("sample_b1_options_pe1", dict(tiny_opts, pe_attn_head=1), B=1, cond_len=16, nt=9, duration=41, steps=5)
    sample_case("sample_b1_nocfg_linspace", tiny, B=1, cond_len=16, nt=8, duration=40, steps=8, cfg_strength=0.0,
                sway=None, use_epss=False)
    sample_case("sample_b1_textclamp", tiny, B=1, cond_len=10, nt=30, duration=12, steps=5)  # duration raised to nt+1
    # cfm.py:141-143,203-208: solve started at t_inter from noise blended with the prompt; steps shrink to int(8 * 0.8)
    sample_case("sample_b1_duplicate", tiny, B=1, cond_len=18, nt=9, duration=52, steps=8, duplicate_test=True, t_inter=0.2)
    forward_taps_case("dit_forward_taps", tiny, B=1, N=48, nt=20, masked=False)
    forward_taps_case("dit_forward_taps_masked", tiny, B=2, N=40, nt=20, masked=True)
    e2_tiny = dict(dim=256, depth=4, heads=4, dim_head=64, ff_mult=2, text_mask_padding=False, pe_attn_head=1,
                   text_dim=None, conv_layers=0, attn_mask_enabled=False, qk_norm=None)
    bucketing_case()
    sample_case("sample_unett_b2", e2_tiny, B=2, cond_len=20, nt=12, duration=[44, 31], lens=[20, 12], steps=5,
                text_pad=[12, 8], backbone="UNetT")


if __name__ == "__main__":
    main()
