"""Does a 16-bit result depend on the GEMM tile configuration?  (it must not)"""
import os, sys, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import torch
    import f5_tts_amd as P
    NV = P.config.VOCAB_SIZE + 1
    arch = dict(P.config.F5TTS_BASE); arch["depth"] = int(os.environ.get("DEPTH", "22"))
    sd = P.weights.synthetic_state_dict(P.weights.dit_param_shapes(arch, NV))
    g = torch.Generator().manual_seed(5)
    durs = [1024, 900, 700, 500]; refs = [d // 4 for d in durs]; nts = [round(0.15 * d) for d in durs]
    cond = torch.zeros(4, max(refs), 100); text = torch.full((4, max(nts)), -1, dtype=torch.long)
    for i, (r, n) in enumerate(zip(refs, nts)):
        cond[i, :r] = torch.randn(r, 100, generator=g); text[i, :n] = torch.randint(1, NV - 2, (n,), generator=g)
    kw = dict(steps=1, cfg_strength=2.0, sway_sampling_coef=-1.0, seed=0, lens=torch.tensor(refs), use_epss=False)
    tr = P.DiT(**arch, text_num_embeds=NV, mel_dim=100, precision=sys.argv[2]); tr.load_state_dict(sd)
    m = P.CFM(transformer=tr, mel_spec_module=P.mel.MelSpec()).to("cuda:0")
    _, t = m.sample(cond, text, torch.tensor(durs), **kw)
    torch.save(t.cpu(), sys.argv[3])
else:
    import torch
    prec = sys.argv[1] if len(sys.argv) > 1 else "bf16"
    ref = None
    only_n = os.environ.get("ONLY_N", "")
    for cfg in ("2", "13") if only_n else ("2", "13", "10", "9", "20", ""):
        env = dict(os.environ, F5_PACK_ROWS="0")
        if cfg: env["F5_GEMM_CFG"] = cfg
        if only_n and cfg != "2": env["F5_GEMM_CFG_N"] = only_n
        out = f"/tmp/cfgdep_{cfg or 'auto'}.pt"
        subprocess.run([sys.executable, __file__, "child", prec, out], env=env, check=True, stderr=subprocess.DEVNULL)
        t = torch.load(out)
        if ref is None: ref = t
        print(prec, "depth", os.environ.get("DEPTH", "22"), "cfg", cfg or "auto", "max diff vs cfg 2:", (t - ref).abs().max().item(), flush=True)
