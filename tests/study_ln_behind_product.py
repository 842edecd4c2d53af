"""Numerical study (CPU, not collected by pytest): would "LayerNorm behind the product"  keep the fp16
blocks inside the parity bar?  The CPU oracle's DiT block is re-run at the C2 size with the engine's fp16 rounding points emulated,

  A (today):    QKV / FF1 read   f16( LN(x) (1 + s) + t )
  B (proposed): QKV / FF1 read   f16( x (1 + s) )   and the epilogue applies   rstd_r acc - rstd_r mu_r c + d,  c = W16 (1+s), d = W16 t + b

(both with f16 weights, f16 q / k / v / P / attention-out / FFN-hidden, f32 accumulation, f32 residual stream, f32 I/O layers as in
F5_PREC_F16P), and the ODE trajectory is compared with the unmodified f32 oracle.     python tests/study_ln_behind_product.py [nfe]
"""
import os
import sys
import time

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import f5_tts_amd as P  # noqa: E402
from oracle import f5_oracle as O  # noqa: E402

NV = P.config.VOCAB_SIZE + 1
h16 = lambda a: a.half().float()   # noqa: E731


def lin16(a16, W, name):            # f16 operands, f32 accumulate (the MFMA)
    return F.linear(a16, h16(W[name + ".weight"]), None)


def attention16(W, cfg, pfx, q, k, v, mask, freqs, b, n):
    H, dh = cfg["heads"], cfg.get("dim_head", 64)
    q, k, v = (z.view(b, n, H, dh).transpose(1, 2) for z in (q, k, v))
    pn = cfg.get("pe_attn_head")
    if pn is None:
        q, k = O.rotary_apply(q, freqs), O.rotary_apply(k, freqs)
    else:
        q = torch.cat((O.rotary_apply(q[:, :pn], freqs), q[:, pn:]), dim=1)
        k = torch.cat((O.rotary_apply(k[:, :pn], freqs), k[:, pn:]), dim=1)
    q, k, v = h16(q * dh ** -0.5), h16(k), h16(v)
    p = torch.softmax(torch.matmul(q, k.transpose(-1, -2)), dim=-1)
    o = torch.matmul(h16(p), v).transpose(1, 2).reshape(b, n, H * dh)
    o = lin16(h16(o), W, pfx + ".to_out.0") + W[pfx + ".to_out.0.bias"]
    if mask is not None:
        o = o.masked_fill(~mask.unsqueeze(-1), 0.0)
    return o


def make_block(scheme):
    def normed_linear(x, scale, shift, W, names):
        """[LN(x)(1+scale)+shift] W^T + b for the weights `names` (concatenated), fp16 operands."""
        Wc = torch.cat([h16(W[n + ".weight"]) for n in names], 0)
        bc = torch.cat([W[n + ".bias"] for n in names], 0)
        if scheme == "A":
            hN = F.layer_norm(x, (x.shape[-1],), eps=1e-6) * (1 + scale[:, None]) + shift[:, None]
            return F.linear(h16(hN), Wc) + bc
        mu = x.mean(-1, keepdim=True)
        var = (x * x).mean(-1, keepdim=True) - mu * mu          # what partial sums of x and x^2 give
        rstd = torch.rsqrt(var + 1e-6)
        acc = F.linear(h16(x * (1 + scale[:, None])), Wc)        # the MFMA product on the un-normalised operand
        c = F.linear(1 + scale, Wc)[:, None]                     # [B, 1, N]
        d = (F.linear(shift, Wc) + bc)[:, None]
        return rstd * acc - (rstd * mu) * c + d

    def block(W, cfg, i, x, t, mask, freqs):
        pfx = f"transformer_blocks.{i}"
        b, n, _ = x.shape
        emb = O.linear(F.silu(t), W, pfx + ".attn_norm.linear")
        shift_msa, scale_msa, gate_msa, shift_mlp, scale_mlp, gate_mlp = torch.chunk(emb, 6, dim=1)
        a = pfx + ".attn"
        qkv = normed_linear(x, scale_msa, shift_msa, W, [a + ".to_q", a + ".to_k", a + ".to_v"])
        q, k, v = torch.chunk(qkv, 3, dim=-1)
        x = x + gate_msa.unsqueeze(1) * attention16(W, cfg, a, q, k, v, mask, freqs, b, n)
        hh = normed_linear(x, scale_mlp, shift_mlp, W, [pfx + ".ff.ff.0.0"])
        hh = h16(F.gelu(hh, approximate="tanh"))
        hh = lin16(hh, W, pfx + ".ff.ff.2") + W[pfx + ".ff.ff.2.bias"]
        return x + gate_mlp.unsqueeze(1) * hh
    return block


def main():
    nfe = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    torch.set_num_threads(os.cpu_count() or 8)
    arch = P.config.F5TTS_BASE
    sd = P.weights.synthetic_state_dict(P.weights.dit_param_shapes(arch, NV))
    g = torch.Generator().manual_seed(1)
    cond = torch.randn(1, 256, 100, generator=g)
    text = torch.randint(1, NV - 2, (1, round(0.15 * 1024)), generator=g)
    kw = dict(steps=nfe, cfg_strength=2.0, sway_sampling_coef=-1.0, seed=0)
    orig = O.dit_block
    res = {}
    with torch.no_grad():
        for name, blk in (("f32", orig), ("A", make_block("A")), ("B", make_block("B"))):
            O.dit_block = blk
            t0 = time.time()
            res[name] = O.sample(sd, arch, cond, text, 1024, **kw)[1]
            print(f"{name}: {time.time() - t0:.0f} s", flush=True)
    O.dit_block = orig
    ref = res["f32"]
    print(f"state magnitude {ref.abs().max():.2f}; trajectory Linf vs the f32 oracle (NFE={nfe}): "
          f"A (round the normalised operand, today) {(res['A'] - ref).abs().max():.3e}, "
          f"B (round x(1+s), normalise behind the product) {(res['B'] - ref).abs().max():.3e}; A vs B {(res['A'] - res['B']).abs().max():.3e}")
    # how far from "mean << sigma" the residual stream is on this input
    print("done")


if __name__ == "__main__":
    main()
