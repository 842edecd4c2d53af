"""CPU: the harness numerics around the hot path (korean-f5-tts_amd/infer.py) against the REFERENCE's own
src/f5_tts/infer/utils_infer.py, imported live in the build container (skipped where the reference is absent), plus
hand-computed cases that also run on the GPU box.  The model and vocoder are deterministic fakes: what is pinned here
is the host arithmetic -- chunking, LoRA merge, RMS normalisation, the duration formula, prompt slicing, the inverse
rescale and the cross-fade."""
import numpy as np
import pytest
import torch

import f5_tts_amd as P
from f5_tts_amd import infer as I
from oracle import ref_harness as rh


class FakeModel:
    """Records what sample() receives; returns a mel that depends on (duration, first samples of the prompt)."""
    device = "cpu"
    vocab_char_map = {c: i for i, c in enumerate(" abcdefghijklmnopqrstuvwxyz,.!?")}

    def __init__(self):
        self.calls = []

    def sample(self, cond, text, duration, steps, cfg_strength, sway_sampling_coef, **kw):
        self.calls.append(dict(nw=cond.shape[-1], rms=float(cond.pow(2).mean().sqrt()), text=text, duration=duration,
                               steps=steps, cfg=cfg_strength, sway=sway_sampling_coef))
        t = torch.arange(duration, dtype=torch.float32)[None, :, None]
        mel = torch.sin(t * 0.1 + torch.arange(100)[None, None, :] * 0.01) + cond[0, :1].sum()
        return mel, None


class FakeVocoder:
    def decode(self, mel):  # [1, 100, T] -> [1, T*4]
        return mel.mean(dim=1).repeat_interleave(4, dim=-1)


TEXTS = ["Short one.", "This is a much longer sentence, with several clauses; it should be split: maybe twice? Yes! Indeed."]


def test_hand_computed_prompt_numerics():
    audio = torch.full((2, 2560), 0.01)            # stereo, rms 0.01 < 0.1 -> scaled to 0.1; 10 frames
    a, rms, rtext, ref_len, dur = I.prompt_numerics(audio, 24000, "abcd", "0123456789ab", speed_=1.0)
    assert a.shape == (1, 2560) and abs(float(a.pow(2).mean().sqrt()) - 0.1) < 1e-6 and abs(rms - 0.01) < 1e-8
    assert rtext == "abcd " and ref_len == 10
    assert dur == 10 + int(10 / 5 * 12 / 1.0)      # ref_text bytes 5 (space appended), gen bytes 12
    _, _, _, _, dur_short = I.prompt_numerics(audio, 24000, "abcd", "hi", speed_=1.0)
    assert dur_short == 10 + int(10 / 5 * 2 / 0.3)  # < 10 bytes -> local_speed 0.3 (utils_infer.py:543-544)
    _, _, _, _, dur_fix = I.prompt_numerics(audio, 24000, "abcd", "hello there", fix_duration_=2.0)
    assert dur_fix == int(2.0 * 24000 / 256)


def test_cross_fade_hand_case():
    a, b = np.ones(5000, dtype=np.float32), np.zeros(4000, dtype=np.float32)
    out = I.cross_fade_concat([a, b], 0.15)
    n = int(0.15 * 24000)
    assert len(out) == 5000 + 4000 - n
    assert np.allclose(out[5000 - n:5000], np.linspace(1, 0, n)) and out[0] == 1 and out[-1] == 0
    assert np.array_equal(I.cross_fade_concat([a, b], 0.0), np.concatenate([a, b]))


needs_ref = pytest.mark.skipif(not rh.available(), reason="reference tree not present (GPU box)")


@needs_ref
def test_chunk_text_matches_reference():
    ui = rh.load_infer()
    for t in TEXTS + ["", "한국어 문장입니다。 두번째 문장！ 세번째？ 끝.", "no punctuation at all " * 20]:
        for mc in (10, 40, 135):
            assert I.chunk_text(t, mc) == ui.chunk_text(t, mc)


@needs_ref
def test_peft_merge_matches_reference():
    ui = rh.load_infer()
    g = torch.Generator().manual_seed(0)
    sd = {}
    for name, (o, i) in {"transformer.transformer_blocks.0.attn.to_q": (8, 6), "transformer.input_embed.proj": (5, 7)}.items():
        sd[f"base_model.model.{name}.base_layer.weight"] = torch.randn(o, i, generator=g)
        sd[f"base_model.model.{name}.base_layer.bias"] = torch.randn(o, generator=g)
        sd[f"base_model.model.{name}.lora_A.default.weight"] = torch.randn(4, i, generator=g)
        sd[f"base_model.model.{name}.lora_B.default.weight"] = torch.randn(o, 4, generator=g)
    sd["base_model.model.transformer.proj_out.weight"] = torch.randn(3, 3, generator=g)
    sd["unrelated.key"] = torch.zeros(1)
    mine, ref = I.convert_peft_state_dict_to_plain(dict(sd)), ui._convert_peft_state_dict_to_plain(dict(sd))
    assert sorted(mine) == sorted(ref)
    for k in ref:
        assert torch.equal(mine[k], ref[k]), k
    plain = {"a": torch.ones(1)}
    assert I.convert_peft_state_dict_to_plain(plain) is plain


@needs_ref
def test_infer_batch_process_matches_reference_harness():
    ui = rh.load_infer()
    g = torch.Generator().manual_seed(3)
    audio = torch.randn(2, 24000, generator=g) * 0.02     # stereo prompt, 1 s, below target rms
    for speed, fix in ((1.0, None), (0.8, None), (1.0, 3.0)):
        kw = dict(nfe_step=7, cfg_strength=1.5, sway_sampling_coef=-0.5, speed=speed, fix_duration=fix,
                  cross_fade_duration=0.01)
        m_ref, m_mine = FakeModel(), FakeModel()
        m_ref._tokenizer_type = "custom"
        wave_r, sr_r, spec_r = next(ui.infer_batch_process((audio, 24000), "ref text here.", TEXTS, m_ref, FakeVocoder(),
                                                           progress=None, device="cpu", **kw))
        wave_m, sr_m, spec_m = next(I.infer_batch_process((audio, 24000), "ref text here.", TEXTS, m_mine, FakeVocoder(),
                                                           device="cpu", **kw))
        assert sr_r == sr_m == 24000
        assert len(m_ref.calls) == len(m_mine.calls) == 2
        for cr, cm in zip(m_ref.calls, m_mine.calls):
            assert cr["nw"] == cm["nw"] and cr["duration"] == cm["duration"] and abs(cr["rms"] - cm["rms"]) < 1e-7
            assert (cr["steps"], cr["cfg"], cr["sway"]) == (cm["steps"], cm["cfg"], cm["sway"])
            # the reference's default tokeniser (convert_char_to_pinyin, out of scope) maps ';' to ','; otherwise the
            # same characters reach the model
            assert "".join(cr["text"][0]) == "".join(cm["text"][0]).replace(";", ",")
        assert np.allclose(wave_r, wave_m, atol=1e-7) and np.array_equal(spec_r, spec_m)
        # the streaming branch (utils_infer.py:711-714,725-728: chunks of every batch's waveform in turn, no cross-fade)
        for chunk_size in (64, 37):
            m_ref, m_mine = FakeModel(), FakeModel()
            m_ref._tokenizer_type = "custom"
            ch_r = list(ui.infer_batch_process((audio, 24000), "ref text here.", TEXTS, m_ref, FakeVocoder(), progress=None,
                                               device="cpu", streaming=True, chunk_size=chunk_size, **kw))
            ch_m = list(I.infer_batch_process((audio, 24000), "ref text here.", TEXTS, m_mine, FakeVocoder(), device="cpu",
                                              streaming=True, chunk_size=chunk_size, **kw))
            assert len(ch_r) == len(ch_m) > 2
            for (wr, sr_r), (wm, sr_m) in zip(ch_r, ch_m):
                assert sr_r == sr_m == 24000 and wr.shape == wm.shape and len(wm) <= chunk_size
                assert np.allclose(wr, wm, atol=1e-7)
    # no batches: one (None, sr, None) item in both
    assert next(ui.infer_batch_process((audio, 24000), "ref text here.", [], FakeModel(), FakeVocoder(), progress=None,
                                       device="cpu")) == (None, 24000, None)
    assert next(I.infer_batch_process((audio, 24000), "ref text here.", [], FakeModel(), FakeVocoder(), device="cpu")) == (None, 24000, None)
    import tqdm
    w_t, _, _ = next(I.infer_batch_process((audio, 24000), "ref text here.", TEXTS, FakeModel(), FakeVocoder(), device="cpu",
                                           progress=tqdm, **kw))
    assert np.allclose(w_t, wave_m, atol=0)


def test_korean_tokenizer_types_need_an_explicit_tokenizer():
    """kor_* checkpoints expect jamo / allophone tokens (utils_infer.py:549-660, out of scope): raw text must not be silently
    mapped to id 0; a caller-supplied tokenizer is passed through to sample() as list[list[str]]."""
    audio = torch.randn(1, 24000, generator=torch.Generator().manual_seed(0)) * 0.1
    m = FakeModel()
    m._tokenizer_type = "kor_allophone"
    with pytest.raises(NotImplementedError, match="text_tokenizer"):
        next(I.infer_batch_process((audio, 24000), "ref text.", ["some text to say."], m, FakeVocoder(), device="cpu"))
    next(I.infer_batch_process((audio, 24000), "ref text.", ["some text to say."], m, FakeVocoder(), device="cpu",
                               text_tokenizer=lambda s: ["<" + c + ">" for c in s]))
    assert m.calls and m.calls[0]["text"][0][0] == "<r>" and isinstance(m.calls[0]["text"][0], list)


def test_cfm_state_dict_roundtrip_and_load_model_surface(tmp_path):
    """load_checkpoint semantics (utils_infer.py:242-286) on files written here: .pt with ema_model_state_dict,
    .safetensors, PEFT keys; loads need no GPU (weights are uploaded lazily)."""
    from safetensors.torch import save_file

    arch = P.config.F5TTS_TINY
    shapes = P.weights.dit_param_shapes(arch, 257)
    sd = P.weights.synthetic_state_dict(shapes, seed=9)
    ema = {"ema_model.transformer." + k: v for k, v in sd.items()}
    ema.update({"initted": torch.tensor(1), "step": torch.tensor(5),
                "ema_model.mel_spec.mel_stft.mel_scale.fb": torch.zeros(3)})
    pt = tmp_path / "model_last.pt"
    torch.save({"ema_model_state_dict": ema, "model_state_dict": {"transformer." + k: v for k, v in sd.items()}}, pt)
    st = tmp_path / "model.safetensors"
    save_file({"ema_model.transformer." + k: v.contiguous() for k, v in sd.items()}, str(st))
    for path, use_ema in ((pt, True), (pt, False), (st, True)):
        model = I.load_model(P.DiT, dict(arch), str(path), vocab_file="", use_ema=use_ema, device="cpu")
        got = model.transformer.state_dict()
        assert all(torch.equal(got[k], sd[k]) for k in sd)
    assert sorted(model.state_dict()) == sorted("transformer." + k for k in sd)


def test_sinc_resample_properties():
    """torchaudio.transforms.Resample restated (parity unpinned: torchaudio is absent).  Properties of the published
    algorithm: output length ceil(n * new / orig); a sine well below both Nyquists keeps its frequency and amplitude; the
    identity rate is a no-op; 48 kHz -> 24 kHz removes content above 12 kHz."""
    import math

    from f5_tts_amd import infer as I
    n, sr = 16000, 16000
    t = torch.arange(n, dtype=torch.float32) / sr
    x = torch.sin(2 * math.pi * 440.0 * t)[None]
    y = I.sinc_resample(x, sr, 24000)
    assert y.shape == (1, math.ceil(n * 3 / 2))
    ref = torch.sin(2 * math.pi * 440.0 * torch.arange(y.shape[1], dtype=torch.float32) / 24000)
    assert (y[0, 200:-200] - ref[200:-200]).abs().max() < 2e-3
    assert I.sinc_resample(x, 24000, 24000) is x
    t48 = torch.arange(48000, dtype=torch.float32) / 48000
    hi = torch.sin(2 * math.pi * 18000.0 * t48)[None]
    lo = torch.sin(2 * math.pi * 3000.0 * t48)[None]
    assert I.sinc_resample(hi, 48000, 24000)[0, 200:-200].abs().max() < 2e-2
    assert (I.sinc_resample(lo, 48000, 24000)[0, 200:-200].abs().max() - 1).abs() < 1e-2
    # the harness takes a prompt at another rate: duration bookkeeping happens on the resampled signal
    a, rms, rtext, ref_len, dur = I.prompt_numerics(x * 0.5, 16000, "hello there", "general kenobi, you are bold")
    assert a.shape[-1] == 24000 and ref_len == 24000 // 256
