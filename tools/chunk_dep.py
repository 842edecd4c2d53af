"""Does a bf16 result depend on how a batch is chunked?  (it must not: rows are independent)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import f5_tts_amd as P
NV = P.config.VOCAB_SIZE + 1
prec = sys.argv[1] if len(sys.argv) > 1 else "bf16"
mask = len(sys.argv) > 2 and sys.argv[2] == "mask"
arch = dict(P.config.F5TTS_BASE, attn_mask_enabled=mask)
sd = P.weights.synthetic_state_dict(P.weights.dit_param_shapes(arch, NV))
gl = torch.Generator().manual_seed(99)
durs = [1024] + [int(x) for x in torch.randint(384, 1025, (15,), generator=gl)]
g = torch.Generator().manual_seed(5)
refs = [d // 4 for d in durs]; nts = [round(0.15 * d) for d in durs]
cond = torch.zeros(16, max(refs), 100); text = torch.full((16, max(nts)), -1, dtype=torch.long)
for i, (r, n) in enumerate(zip(refs, nts)):
    cond[i, :r] = torch.randn(r, 100, generator=g); text[i, :n] = torch.randint(1, NV - 2, (n,), generator=g)
kw = dict(steps=1, cfg_strength=2.0, sway_sampling_coef=-1.0, seed=0, lens=torch.tensor(refs), use_epss=False)
valid = (torch.arange(1024)[None, :] < torch.tensor(durs)[:, None])[..., None].cuda()
res = {}
for rows in ("20000", "12000", "100000", "5000"):
    os.environ["F5_CHUNK_ROWS"] = rows
    tr = P.DiT(**arch, text_num_embeds=NV, mel_dim=100, precision=prec); tr.load_state_dict(sd)
    m = P.CFM(transformer=tr, mel_spec_module=P.mel.MelSpec()).to("cuda:0")
    _, t = m.sample(cond, text, torch.tensor(durs), **kw)
    res[rows] = t.clone()
    d = ((t - res["20000"]) * valid).abs()
    print(prec, "mask" if mask else "nomask", "chunk rows", rows, "diff vs 20000:", d.max().item(), "per utt:", [round(x, 4) for x in d.amax(dim=(0, 2, 3)).tolist()])
