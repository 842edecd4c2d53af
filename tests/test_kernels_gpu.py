"""GPU: every hand-written kernel against a torch fp32 restatement of the same op (through the C ABI's f5k_* entry
points).  f32 mode must agree to fp32 rounding.  The 16-bit-operand modes (bf16, f16) are checked twice: against the
plain fp32 op at the operand type's rounding, and -- tightly -- against the same op evaluated in float64 on operands
ROUNDED to that type, which leaves only the kernel's own arithmetic (f32 accumulation, P rounded to 16 bit inside the
attention kernel): a 2x accuracy regression in a speed path fails these (tolerances written per test)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from gpu_util import DEV, k_attention, k_convpos, k_gemm, k_layernorm_mod  # noqa: E402


def rel_err(a, b):
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-12)).item()


TDTYPE = {"bf16": torch.bfloat16, "f16": torch.float16}


def rnd(prec, x):
    """x as the kernel's MFMA operand type sees it (f32: unchanged; f16x3: hi + lo f16 halves = 22 bits, treated as unchanged)."""
    return x if prec in ("f32", "f16x3") else x.to(TDTYPE[prec]).float()


GEMM_SHAPES = [(2048, 1024, 1024), (2048, 3072, 1024), (2048, 1024, 2048), (300, 100, 1024), (77, 64, 712),
               (16, 6144, 1024), (1, 4, 8), (130, 260, 40), (513, 1028, 512)]


@pytest.mark.parametrize("M,N,K", GEMM_SHAPES)
@pytest.mark.parametrize("prec,tol", [("f32", 2e-5), ("f16x3", 2e-5), ("bf16", 1.5e-2), ("f16", 2e-3)])
def test_gemm_bias_matches_torch(M, N, K, prec, tol):
    g = torch.Generator().manual_seed(M * 7 + N)
    A = torch.randn(M, K, generator=g).to(DEV)
    W = (torch.randn(N, K, generator=g) / K ** 0.5).to(DEV)
    b = torch.randn(N, generator=g).to(DEV)
    ref = F.linear(A.double(), W.double(), b.double()).float()
    out = k_gemm(prec, A, W, b)
    assert torch.isfinite(out).all()
    assert rel_err(out, ref) < tol
    # same operands as the kernel multiplies: only the f32 accumulation order is left
    ref_r = F.linear(rnd(prec, A).double(), rnd(prec, W).double(), b.double()).float()
    assert rel_err(out, ref_r) < 2e-5


@pytest.mark.parametrize("M,N,K", [(2048, 1024, 1024), (300, 100, 1024), (513, 1028, 512), (77, 64, 712)])
def test_gemm_f16x3_presplit_a_operand_is_bit_identical(M, N, K):
    """F5_PREC_F16X3: the block GEMMs read an A operand that its producer stored already split (store4_planar -> gemm2.h MODE 5).
    Splitting in memory or in registers yields the same hi / lo halves, so both kernels must agree bit for bit (every tile)."""
    g = torch.Generator().manual_seed(M + K)
    A = torch.randn(M, K, generator=g).to(DEV)
    W = (torch.randn(N, K, generator=g) / K ** 0.5).to(DEV)
    b = torch.randn(N, generator=g).to(DEV)
    for cfg in (0, -8, -9, -2, -10, -13):
        in_regs = k_gemm("f16x3", A, W, b, tile=(cfg, 0))
        in_mem = k_gemm("f16x3", A, W, b, tile=(cfg, 5))
        assert torch.equal(in_regs, in_mem), cfg


@pytest.mark.parametrize("prec", ["bf16", "f16"])
def test_gemm_many_rows_with_a_small_remainder_is_split_and_bit_identical(prec):
    """M = 48 x 256 + 16 (UNetT batches: B x 1025 rows): launch_gemm sends the 256-row multiple to the ping-pong kernel and the
    last 16 rows to 64x64 tiles (gemm_dispatch.h); same K order per element, so equal bit for bit to one forced-tile launch."""
    g = torch.Generator().manual_seed(5)
    M, N, K = 48 * 256 + 16, 1024, 512
    A = torch.randn(M, K, generator=g).to(DEV)
    W = (torch.randn(N, K, generator=g) / K ** 0.5).to(DEV)
    b = torch.randn(N, generator=g).to(DEV)
    out = k_gemm(prec, A, W, b)
    one = k_gemm(prec, A, W, b, tile=(-13, 0))
    assert torch.equal(out, one)
    ref = F.linear(rnd(prec, A).double(), rnd(prec, W).double(), b.double()).float()
    assert rel_err(out, ref) < 2e-5


@pytest.mark.parametrize("tile", [(128, 128), (128, 64), (64, 64), (-2, 0), (-8, 0), (-9, 0)])  # v1 tiles, v2 config ids
@pytest.mark.parametrize("prec", ["f32", "bf16", "f16"])
def test_gemm_every_tile_shape_and_identity(tile, prec):
    """A = I with an ASYMMETRIC W catches a transposed accumulator map (guide section 3)."""
    M = N = K = 256
    A = torch.eye(M).to(DEV)
    W = (torch.arange(N * K, dtype=torch.float32).reshape(N, K) % 251 - 125).to(DEV) / 16  # exact in bf16
    out = k_gemm(prec, A, W, None, tile=tile)
    assert torch.equal(out, W.t().contiguous())


@pytest.mark.parametrize("act,fn", [(1, lambda x: F.gelu(x, approximate="tanh")), (2, F.gelu), (3, F.silu), (4, F.mish)])
def test_gemm_activation_epilogues(act, fn):
    g = torch.Generator().manual_seed(act)
    A = torch.randn(200, 96, generator=g).to(DEV)
    W = (torch.randn(128, 96, generator=g) * 0.2).to(DEV)
    b = torch.randn(128, generator=g).to(DEV)
    ref = fn(F.linear(A, W, b))
    out = k_gemm("f32", A, W, b, act=act)
    assert (out - ref).abs().max() < 2e-5


def sdpa_ref(q, k, v, lens=None):
    s = torch.matmul(q.double(), k.double().transpose(-1, -2)) / 8.0
    if lens is not None:
        N = q.shape[2]
        m = torch.arange(N, device=q.device)[None, :] < torch.tensor(lens, device=q.device)[:, None]
        s = s.masked_fill(~m[:, None, None, :], float("-inf"))
    o = torch.matmul(torch.softmax(s, dim=-1), v.double())
    return o.transpose(1, 2).reshape(q.shape[0], q.shape[2], -1).float()


# tol: against SDPA of the unrounded operands; tol_r: against SDPA of the operands as the kernel rounds them (what is
# left is the 16-bit rounding of P inside the kernel and f32 accumulation)
ATTN_TOLS = [("f32", 3e-5, 3e-5), ("f16x3", 3e-5, 3e-5), ("bf16", 2e-2, 6e-3), ("f16", 3e-3, 8e-4)]


@pytest.mark.parametrize("Bp,H,N", [(2, 16, 1024), (1, 4, 64), (2, 4, 48), (3, 2, 200), (1, 2, 129), (2, 3, 777)])
@pytest.mark.parametrize("prec,tol,tol_r", ATTN_TOLS)
def test_attention_matches_sdpa(Bp, H, N, prec, tol, tol_r):
    g = torch.Generator().manual_seed(N)
    q, k, v = (torch.randn(Bp, H, N, 64, generator=g).to(DEV) for _ in range(3))
    out = k_attention(prec, q, k, v)
    ref = sdpa_ref(q, k, v)
    assert torch.isfinite(out).all()
    assert (out - ref).abs().max() < tol
    e_r = (out - sdpa_ref(*_as_operands(prec, q, k, v))).abs().max().item()
    print(f"[attention {prec}] Bp={Bp} H={H} N={N}: Linf vs SDPA of the rounded operands {e_r:.2e}")
    assert e_r < tol_r


@pytest.mark.parametrize("prec,tol", [("f32", 3e-5), ("f16x3", 3e-5), ("bf16", 2e-2), ("f16", 3e-3)])
def test_attention_key_padding_mask_and_peaked_softmax(prec, tol):
    """attn_mask_enabled path (modules.py:501-506) + a forced running-max jump (one key dominates late)."""
    g = torch.Generator().manual_seed(3)
    Bp, H, N = 3, 2, 300
    q, k, v = (torch.randn(Bp, H, N, 64, generator=g).to(DEV) for _ in range(3))
    k[:, :, 250] = q[:, :, 7] * 3.0  # key 250 spikes against query 7: max jumps in the last tile
    lens = [300, 131, 257]
    out = k_attention(prec, q, k, v, lens)
    ref = sdpa_ref(q, k, v, lens)
    assert (out - ref).abs().max() < tol


def _as_operands(prec, q, k, v):
    """What the kernel multiplies: the 16-bit paths round k, v and q * attention_q_scale (dim_head^-0.5 * log2 e, attn2.h)
    to the operand type.  With scores of magnitude 100+ that operand rounding moves the softmax far more than any
    kernel-internal arithmetic, so the extreme-score cases compare against SDPA of the ROUNDED operands."""
    if prec == "f32":
        return q, k, v
    qs = 0.125 * 1.4426950408889634
    return rnd(prec, q * qs) / qs, rnd(prec, k), rnd(prec, v)


@pytest.mark.parametrize("prec,tol", [("f32", 3e-5), ("f16x3", 1e-4), ("bf16", 2.5e-2), ("f16", 4e-3)])
def test_attention_reference_tracking_extremes(prec, tol):
    """The 16-bit kernel keeps a lazily updated softmax reference (attn2.h): scores that keep growing tile after tile
    (reference moves many times), scores that are all very negative (the first-tile reference must follow DOWN or the
    row underflows to 0/0), and rows whose valid keys end inside the first tile / first key half."""
    g = torch.Generator().manual_seed(11)
    Bp, H, N = 2, 2, 520
    q, k, v = (torch.randn(Bp, H, N, 64, generator=g).to(DEV) for _ in range(3))
    ramp = torch.linspace(0.2, 6.0, N, device=DEV)
    k_grow = k * ramp[None, None, :, None]                  # later keys score higher and higher
    out = k_attention(prec, q, k_grow, v)
    assert torch.isfinite(out).all()
    assert (out - sdpa_ref(*_as_operands(prec, q, k_grow, v))).abs().max() < tol
    k_neg = k.clone()
    k_neg[..., 0] = 40.0
    q_neg = q.clone()
    q_neg[..., 0] = -q_neg[..., 0].abs() * 8.0 - 8.0         # every score is about -40 .. -400 after the 1/8
    out = k_attention(prec, q_neg, k_neg, v)
    assert torch.isfinite(out).all()
    assert (out - sdpa_ref(*_as_operands(prec, q_neg, k_neg, v))).abs().max() < tol
    lens = [5, 40]                                           # 5 < 32: the second key half of tile 0 is fully masked
    out = k_attention(prec, q, k, v, lens)
    assert torch.isfinite(out).all()
    assert (out - sdpa_ref(q, k, v, lens)).abs().max() < tol


@pytest.mark.parametrize("D,N,Bp", [(256, 48, 2), (1024, 300, 2), (512, 129, 1), (1024, 1024, 2), (768, 200, 2)])
@pytest.mark.parametrize("prec,tol", [("f32", 3e-5), ("f16x3", 3e-5), ("bf16", 2e-2), ("f16", 3e-3)])
def test_convpos_matches_conv1d_mish(D, N, Bp, prec, tol):
    g = torch.Generator().manual_seed(D + N)
    x = torch.randn(Bp, N, D, generator=g).to(DEV)
    w = (torch.randn(D, D // 16, 31, generator=g) * (1.0 / (31 * D / 16) ** 0.5)).to(DEV)
    b = torch.randn(D, generator=g).to(DEV)
    res = torch.randn(Bp, N, D, generator=g).to(DEV)
    ref = F.mish(F.conv1d(x.permute(0, 2, 1), w, b, padding=15, groups=16)).permute(0, 2, 1) + res
    out = k_convpos(prec, x, w, b, res)
    assert (out - ref).abs().max() < tol * max(1.0, ref.abs().max().item())
    ref_r = F.mish(F.conv1d(rnd(prec, x).permute(0, 2, 1), rnd(prec, w), b, padding=15, groups=16)).permute(0, 2, 1) + res
    assert (out - ref_r).abs().max() < 5e-5 * max(1.0, ref.abs().max().item())


def test_convpos_masked_rows():
    g = torch.Generator().manual_seed(9)
    Bp, N, D = 2, 150, 256
    x = torch.randn(Bp, N, D, generator=g).to(DEV)
    w = (torch.randn(D, 16, 31, generator=g) * 0.05).to(DEV)
    b = torch.randn(D, generator=g).to(DEV)
    lens = [150, 97]
    m = (torch.arange(N, device=DEV)[None] < torch.tensor(lens, device=DEV)[:, None])[:, None]  # [B,1,N]
    h = x.permute(0, 2, 1).masked_fill(~m, 0.0)
    h = F.conv1d(h, w, b, padding=15, groups=16).masked_fill(~m, 0.0)
    ref = F.mish(h).permute(0, 2, 1)
    out = k_convpos("f32", x, w, b, None, lens)
    assert (out - ref).abs().max() < 3e-5


@pytest.mark.parametrize("D", [64, 256, 512, 1024])
def test_layernorm_modulate(D):
    g = torch.Generator().manual_seed(D)
    R, rpb = 96, 32
    x = (torch.randn(R, D, generator=g) * 3 + 1).to(DEV)
    sc = torch.randn(R // rpb, D, generator=g).to(DEV)
    sh = torch.randn(R // rpb, D, generator=g).to(DEV)
    ref = F.layer_norm(x, (D,), eps=1e-6) * (1 + sc.repeat_interleave(rpb, 0)) + sh.repeat_interleave(rpb, 0)
    out = k_layernorm_mod(x, sc, sh, rpb)
    assert (out - ref).abs().max() < 2e-5


@pytest.mark.parametrize("prec", ["bf16", "f16", "f32"])
def test_kernels_are_bitwise_deterministic(prec):
    """Race detector: a stale LDS / register read shows up as run-to-run differences long before it breaks a tolerance
    (an inline-asm VALU consumer of MFMA accumulators did exactly that: nothing pads that hazard)."""
    g = torch.Generator().manual_seed(0)
    q, k, v = (torch.randn(2, 8, 1024, 64, generator=g).to(DEV) for _ in range(3))
    ref = k_attention(prec, q, k, v)
    for _ in range(6):
        assert torch.equal(k_attention(prec, q, k, v), ref)
    A = torch.randn(2048, 1024, generator=g).to(DEV)
    W = (torch.randn(1024, 1024, generator=g) / 32).to(DEV)
    for tile in ((0, 0), (-2, 0), (-8, 0), (-9, 0)):
        ref = k_gemm(prec, A, W, None, tile=tile)
        for _ in range(4):
            assert torch.equal(k_gemm(prec, A, W, None, tile=tile), ref)
    x = torch.randn(2, 300, 1024, generator=g).to(DEV)
    w = (torch.randn(1024, 64, 31, generator=g) * 0.02).to(DEV)
    b = torch.randn(1024, generator=g).to(DEV)
    ref = k_convpos(prec, x, w, b, x)
    for _ in range(4):
        assert torch.equal(k_convpos(prec, x, w, b, x), ref)
