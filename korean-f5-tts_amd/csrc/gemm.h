// MFMA "TN" GEMM for gfx950:  C[M,N] = A[M,K] * W[N,K]^T  with fused epilogues.
//
// Both operands are K-contiguous (activations [rows, features]; torch nn.Linear weights [out, in]), so a 16-byte
// chunk of either is directly one MFMA operand fragment:
//   bf16 / f16: v_mfma_f32_16x16x32_{bf16,f16} -- lane l holds row (l&15), k = 8*(l>>4)..+7 (one fragment = 1 MFMA)
//   f32 : v_mfma_f32_16x16x4_f32   -- lane l holds row (l&15), 4 consecutive k of chunk (l>>4); element j feeds the
//         j-th of 4 MFMAs (the k order inside a 16-wide group is permuted identically for both operands, which a sum
//         over k does not care about).  Exact f32: bit-for-bit an fmaf chain (MI355X_MICROARCH "Matrix cores").
// The A- and B-operand fragment formats are identical, so swapping the two MFMA arguments transposes the 16x16
// accumulator for free.  Default ("row-packed") orientation: W fragment in the A slot -> each lane owns 4 CONSECUTIVE
// output columns of one output row (8/16-byte stores, float4 bias/gate loads, rotary pairs lane-local).  An epilogue
// may ask for the other orientation per column tile (4 consecutive ROWS per lane) -- used to emit V transposed.
//
// Block = 256 threads = 4 waves (2x2); tile BM x BN x 128 bytes of K; LDS rows are 128 B + 16 B pad (144-B stride:
// the 16 rows of a fragment read land on 16 distinct 16-B slots -> conflict-free ds_read_b128); register-staged
// double buffering (global loads for tile t+1 are issued before the MFMAs of tile t and written to the other LDS
// buffer after them; one barrier per K tile).
#pragma once
#include <type_traits>
#include "f5_common.h"

namespace f5 {

constexpr int GEMM_ROW_BYTES = 128;
constexpr int GEMM_ROW_STRIDE = 144;

template <typename T> struct Mma;
template <> struct Mma<bf16_t> {
    static __device__ __forceinline__ f32x4 run(const u32x4& a, const u32x4& b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c,
                                                       0, 0, 0);
    }
};
template <> struct Mma<f16_t> {   // v_mfma_f32_16x16x32_f16: same fragment layout and rate as the bf16 form, 10-bit mantissa operands
    static __device__ __forceinline__ f32x4 run(const u32x4& a, const u32x4& b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    }
};
template <> struct Mma<float> {
    static __device__ __forceinline__ f32x4 run(const u32x4& a, const u32x4& b, f32x4 c) {
        const f32x4 fa = __builtin_bit_cast(f32x4, a), fb = __builtin_bit_cast(f32x4, b);
#pragma unroll
        for (int j = 0; j < 4; ++j) c = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[j], fb[j], c, 0, 0, 0);
        return c;
    }
};

// ---------------------------------------------------------------------------------------------- epilogues
// An epilogue splits its per-element work into a ROW context (computed once per output row a lane touches: batch
// index, position, masks -- this is where the integer divisions live), a COLUMN context (once per 4-column group:
// bias, head, ...) and a store that combines them:
//   RowCtx row(m) / ColCtx col(n)                      row-packed orientation: v[r] = C[m][n + r]      (n % 4 == 0)
//   Pre    preload(RowCtx, ColCtx)                    operands the store needs from memory (residual, gate): requested
//                                                     BEFORE the K loop by the v2 kernel so the epilogue is not a round trip deep
//   void   store(RowCtx, ColCtx, v, Pre)
//   TRowCtx trow(m, M) / TColCtx tcol(n)               transposed orientation: v[r] = C[m + r][n]      (m % 4 == 0)
//   void   tstore(TRowCtx, TColCtx, v)                 -- only reached if tile_transposed() can be true
// STAGED stores (gemm3.h, many-row tiles).  A lane of an MFMA accumulator owns 4 columns of 16 different rows, so a fragment-order
// store instruction writes 16 pieces of 32 (16-bit output) or 64 (f32) contiguous bytes: measured 2.4-3.7 TB/s against 5.8-6.7 for
// whole rows (tools/store_pattern.py).  An epilogue with  kStaged = true  turns a wave's 128 x 64 tile through 16 KiB of LDS itself
// (fragment order in, rows out: every store instruction writes whole 128-byte lines) and is written so that every operand load
// of the tile is in flight at once and nothing branches around a store -- hipcc puts `s_waitcnt vmcnt(0)` in front of every store
// that follows a conditional load, which made the first generic form of this one round trip per row:
//   bool staged_ok(bool transposed)                      once per tile (wave-uniform)
//   void staged<MI, NJ>(slice, lane, mw, nw, acc, transposed)
// Results are bit-identical to store() / tstore().  16-byte chunk c of staging row r lives in slot c ^ (r & 7) (128-byte rows) or
// c ^ (r & 15) (256-byte rows); a wave only reads what it wrote itself, so no barrier is involved.
// (The same idea for the v2 kernel at M = 2,048 -- whole block tile through LDS behind two barriers, residual operands requested
// before the K loop -- was built and measured in round 3: 77.9 us per DiT block's four GEMMs against 77.5; at that size the epilogue
// is a latency chain, not a store-bandwidth problem.  Removed.)
struct NoCtx {};

// ACT >= 0 fixes the activation at compile time.  This matters more than it looks: with a run-time `act` every one of
// the 32 stores of a lane inlines all six activations and the kernel grows to 63 KB (128x128) -- the whole 64 KB
// instruction cache a CU pair shares -- so every launch that follows a different kernel starts instruction-cold
// (+7 us measured per launch, tools/pair_time.py).  Host code builds EpiStore<TO> (ACT = -1, run-time field) and the
// launchers rewrite it to the static form through with_static_act().
template <typename TO, int ACT = -1> struct EpiStore {  // out = act(acc + bias)
    TO* out; int ldo; const float* bias; int act;
    int planar = 0;      // TO = float only: store pre-split for the next GEMM's A operand (f5_common.h store4_planar)
    struct RowCtx { TO* p; };
    struct ColCtx { float4 b; int n; };
    typedef NoCtx TRowCtx;
    typedef NoCtx TColCtx;
    static constexpr bool kTransposes = false;
    __device__ __forceinline__ bool tile_transposed(int) const { return false; }
    __device__ __forceinline__ RowCtx row(int m) const { return {out + (size_t)m * ldo}; }
    __device__ __forceinline__ ColCtx col(int n) const {
        return {bias ? *reinterpret_cast<const float4*>(bias + n) : make_float4(0, 0, 0, 0), n};
    }
    typedef NoCtx Pre;
    __device__ __forceinline__ NoCtx preload(const RowCtx&, const ColCtx&) const { return {}; }
    __device__ __forceinline__ void store(const RowCtx& r, const ColCtx& c, f32x4 v, const NoCtx&) const {
        const int a = ACT >= 0 ? ACT : act;
        store4_at(r.p, c.n, planar, apply_act(v[0] + c.b.x, a), apply_act(v[1] + c.b.y, a), apply_act(v[2] + c.b.z, a),
                  apply_act(v[3] + c.b.w, a));
    }
    static constexpr bool kStaged = sizeof(TO) == 2;
    __device__ __forceinline__ bool staged_ok(bool transposed) const { return !transposed; }
    template <int MI, int NJ>
    __device__ __forceinline__ void staged(char* slice, int lane, int mw, int nw, const f32x4 (&acc)[MI][NJ], bool) const {
        if constexpr (sizeof(TO) == 2) {
            static_assert(MI == 8 && NJ == 4, "128 x 64 wave tile");
            const int l15 = lane & 15, g = lane >> 4, a = ACT >= 0 ? ACT : act;
            float4 bj[NJ];
#pragma unroll
            for (int j = 0; j < NJ; ++j) bj[j] = bias ? *reinterpret_cast<const float4*>(bias + nw + j * 16 + g * 4) : make_float4(0, 0, 0, 0);
#pragma unroll
            for (int j = 0; j < NJ; ++j)
#pragma unroll
                for (int i = 0; i < MI; ++i) {
                    const f32x4 v = acc[i][j];
                    const int r = i * 16 + l15;
                    *reinterpret_cast<u32x2*>(slice + r * 128 + (((j * 2 + (g >> 1)) ^ (r & 7)) << 4) + ((g & 1) << 3)) =
                        pack4<TO>(apply_act(v[0] + bj[j].x, a), apply_act(v[1] + bj[j].y, a), apply_act(v[2] + bj[j].z, a), apply_act(v[3] + bj[j].w, a));
                }
            const int rr = lane >> 3, ch = lane & 7;
            TO* dst = out + (size_t)(mw + rr) * ldo + nw + ch * 8;
#pragma unroll
            for (int p = 0; p < 16; ++p) {
                const int r = p * 8 + rr;
                *reinterpret_cast<u32x4*>(dst + (size_t)p * 8 * ldo) = *reinterpret_cast<const u32x4*>(slice + r * 128 + ((ch ^ (r & 7)) << 4));
            }
        }
    }
    __device__ __forceinline__ NoCtx trow(int, int) const { return {}; }
    __device__ __forceinline__ NoCtx tcol(int) const { return {}; }
    __device__ __forceinline__ void tstore(const NoCtx&, const NoCtx&, f32x4) const {}
};

// f(epilogue-with-static-activation); any other epilogue type passes through unchanged
template <typename Epi, typename F> inline hipError_t with_static_act(const Epi& e, F&& f) { return f(e); }
template <typename TO, typename F> inline hipError_t with_static_act(const EpiStore<TO, -1>& e, F&& f) {
    switch (e.act) {
        case F5_ACT_NONE: return f(EpiStore<TO, F5_ACT_NONE>{e.out, e.ldo, e.bias, e.act, e.planar});
        case F5_ACT_GELU_TANH: return f(EpiStore<TO, F5_ACT_GELU_TANH>{e.out, e.ldo, e.bias, e.act, e.planar});
        case F5_ACT_GELU_ERF: return f(EpiStore<TO, F5_ACT_GELU_ERF>{e.out, e.ldo, e.bias, e.act, e.planar});
        case F5_ACT_SILU: return f(EpiStore<TO, F5_ACT_SILU>{e.out, e.ldo, e.bias, e.act, e.planar});
        case F5_ACT_MISH: return f(EpiStore<TO, F5_ACT_MISH>{e.out, e.ldo, e.bias, e.act, e.planar});
        case F5_ACT_LOGCLAMP: return f(EpiStore<TO, F5_ACT_LOGCLAMP>{e.out, e.ldo, e.bias, e.act, e.planar});
        default: return hipErrorInvalidValue;
    }
}

// x[m][n] = res[m][n] + gate[b(m)][n] * (acc + bias)    (res may alias x; gate == nullptr -> 1; rows m with
// (m % rows_per_batch) >= lens[m / rows_per_batch] are left as res: the reference's masked_fill(~mask, 0) on the
// attention output, modules.py:540-542)
struct EpiGateRes {
    float* x; const float* res; int ld; const float* bias; const float* gate; int gate_stride; int rows_per_batch;
    const int* lens;
    struct RowCtx { size_t off; const float* g; bool masked; };
    struct ColCtx { float4 b; int n; };
    typedef NoCtx TRowCtx;
    typedef NoCtx TColCtx;
    static constexpr bool kTransposes = false;
    __device__ __forceinline__ bool tile_transposed(int) const { return false; }
    __device__ __forceinline__ RowCtx row(int m) const {
        const int b = m / rows_per_batch;
        return {(size_t)m * ld, gate ? gate + (size_t)b * gate_stride : nullptr,
                lens ? (m - b * rows_per_batch) >= lens[b] : false};
    }
    __device__ __forceinline__ ColCtx col(int n) const {
        return {bias ? *reinterpret_cast<const float4*>(bias + n) : make_float4(0, 0, 0, 0), n};
    }
    struct Pre { float4 r, g; };
    __device__ __forceinline__ Pre preload(const RowCtx& rc, const ColCtx& c) const {
        return {*reinterpret_cast<const float4*>(res + rc.off + c.n),
                rc.g ? *reinterpret_cast<const float4*>(rc.g + c.n) : make_float4(1, 1, 1, 1)};
    }
    __device__ __forceinline__ void store(const RowCtx& rc, const ColCtx& c, f32x4 v, const Pre& p) const {
        float4 r = p.r;
        if (!rc.masked) {
            const float4 g = p.g;
            r.x += g.x * (v[0] + c.b.x); r.y += g.y * (v[1] + c.b.y);
            r.z += g.z * (v[2] + c.b.z); r.w += g.w * (v[3] + c.b.w);
        }
        *reinterpret_cast<float4*>(x + rc.off + c.n) = r;
    }
    __device__ __forceinline__ NoCtx trow(int, int) const { return {}; }
    __device__ __forceinline__ NoCtx tcol(int) const { return {}; }
    __device__ __forceinline__ void tstore(const NoCtx&, const NoCtx&, f32x4) const {}
    // Staged form: the RAW accumulators are turned (two halves of 64 rows x 256 B); a lane then owns 4 columns of one row per step:
    // 16 lanes x 16 B = one 256-byte piece of a row of x, read and written whole.  The 16 residual loads of a half are requested
    // before the half is staged; the (at most two) batch rows a half touches have their gate vectors and lengths loaded once.
    static constexpr bool kStaged = true;
    __device__ __forceinline__ bool staged_ok(bool transposed) const { return !transposed && rows_per_batch >= 64; }
    template <int MI, int NJ>
    __device__ __forceinline__ void staged(char* slice, int lane, int mw, int nw, const f32x4 (&acc)[MI][NJ], bool) const {
        static_assert(MI == 8 && NJ == 4, "128 x 64 wave tile");
        const int l15 = lane & 15, g = lane >> 4, rr = lane >> 4, ch = lane & 15;
        const int n = nw + ch * 4;
        const float4 bv = bias ? *reinterpret_cast<const float4*>(bias + n) : make_float4(0, 0, 0, 0);
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            const int mb = mw + half * 64;
            const int b0 = mb / rows_per_batch, b1 = (mb + 63) / rows_per_batch;
            const int m0 = b0 * rows_per_batch, m1 = b1 * rows_per_batch;   // first rows of the two batch rows (m1 == m0 when there is one)
            float4 g0 = make_float4(1, 1, 1, 1), g1 = g0;
            if (gate) {
                g0 = *reinterpret_cast<const float4*>(gate + (size_t)b0 * gate_stride + n);
                g1 = *reinterpret_cast<const float4*>(gate + (size_t)b1 * gate_stride + n);
            }
            int len0 = 0x7fffffff, len1 = 0x7fffffff;
            if (lens) { len0 = lens[b0]; len1 = lens[b1]; }
            float4 rv[16];
            const float* rp = res + (size_t)(mb + rr) * ld + n;
#pragma unroll
            for (int p = 0; p < 16; ++p) rv[p] = *reinterpret_cast<const float4*>(rp + (size_t)p * 4 * ld);
#pragma unroll
            for (int j = 0; j < NJ; ++j)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int r = i * 16 + l15;
                    *reinterpret_cast<f32x4*>(slice + r * 256 + (((j * 4 + g) ^ (r & 15)) << 4)) = acc[half * 4 + i][j];
                }
            float* xp = x + (size_t)(mb + rr) * ld + n;
#pragma unroll
            for (int p = 0; p < 16; ++p) {
                const int r = p * 4 + rr, m = mb + r;
                const f32x4 v = *reinterpret_cast<const f32x4*>(slice + r * 256 + ((ch ^ (r & 15)) << 4));
                const bool second = m >= m1 && b1 != b0;
                const float4 gg = second ? g1 : g0;
                const bool masked = (m - (second ? m1 : m0)) >= (second ? len1 : len0);
                float4 o = rv[p];
                const float tx = gg.x * (v[0] + bv.x), ty = gg.y * (v[1] + bv.y), tz = gg.z * (v[2] + bv.z), tw = gg.w * (v[3] + bv.w);
                o.x = masked ? o.x : o.x + tx; o.y = masked ? o.y : o.y + ty; o.z = masked ? o.z : o.z + tz; o.w = masked ? o.w : o.w + tw;
                *reinterpret_cast<float4*>(xp + (size_t)p * 4 * ld) = o;
            }
        }
    }
};

// Fused QKV projection epilogue (modules.py:469-497): bias, interleaved-pair rotary on the first `pe_heads` heads of
// q and k, q pre-scaled by dim_head^-0.5, head split.  q,k -> [B', H, Nseq, 64]; v -> TRANSPOSED [B', H, 64, Npad]
// (so that both attention B-operands are K-contiguous).  16-column sub-tiles of the V third use the transposed
// orientation.
// rowmap (may be null): GEMM row m is (batch row, position) = rowmap[m] (packed variable-length batches, RowPack) instead
// of (m / Nseq, m % Nseq); every utterance then starts at a multiple of 4 rows, and positions >= Nseq (alignment rows
// of the longest utterance) are dropped.
template <typename TO> struct EpiQKV {
    // rope: the rotary table in FRAGMENT order (rope_frag_kernel, elementwise.h): [maxpos][g = 0..3][j = 0..3]{cos, cos, sin, sin} of the
    // pairs (j*8 + g*2, +1) -- the 4 columns j*16 + g*4 .. +3 of a head that one accumulator lane owns need ONE 16-byte load, and a lane's
    // four sub-tiles j are 64 contiguous bytes (two tables of [maxpos][32] cost four 8-byte loads per sub-tile pair: all-heads rotary,
    // pe_attn_head = None, was +38 us per 32,768-row QKV launch)
    TO* q; TO* k; TO* vt; const float* bias; const float* rope;
    int Nseq, Npad, H, pe_heads; float q_scale;
    const int2* rowmap = nullptr;
    struct RowCtx { size_t base; const float* cs; };      // base = (b*H*Nseq + pos) * 64; cs = the position's 64 table floats; null: drop
    struct ColCtx { float4 b; TO* dst; size_t hoff; int d; bool rot; float scale; };
    struct TRowCtx { size_t base; int pos; int b; bool fast; int m; int M; };
    struct TColCtx { float b; size_t hoff; };
    static constexpr bool kTransposes = true;
    __device__ __forceinline__ bool tile_transposed(int n0) const { return n0 >= 2 * H * 64; }
    __device__ __forceinline__ RowCtx row(int m) const {
        int b = m / Nseq, pos = m - b * Nseq;
        if (rowmap) { const int2 bp = rowmap[m]; b = bp.x; pos = bp.y; }
        if (pos >= Nseq) return {0, nullptr};
        return {((size_t)b * H * Nseq + pos) * 64, rope + pos * 64};
    }
    __device__ __forceinline__ ColCtx col(int n) const {
        const int inner = H * 64;
        const int which = n >= inner ? 1 : 0, c = n - which * inner, h = c >> 6, d = c & 63;
        return {*reinterpret_cast<const float4*>(bias + n), which ? k : q, (size_t)h * Nseq * 64 + d, d, h < pe_heads,
                which ? 1.0f : q_scale};
    }
    struct Pre { float2 cs, sn; };
    __device__ __forceinline__ Pre preload(const RowCtx& r, const ColCtx& c) const {
        if (!c.rot || !r.cs) return {make_float2(1, 1), make_float2(0, 0)};
        const float4 t = *reinterpret_cast<const float4*>(r.cs + ((c.d >> 2) & 3) * 16 + (c.d >> 4) * 4);   // d = j*16 + g*4
        return {make_float2(t.x, t.y), make_float2(t.z, t.w)};
    }
    __device__ __forceinline__ void store(const RowCtx& r, const ColCtx& c, f32x4 v, const Pre& p) const {
        float a0 = v[0] + c.b.x, a1 = v[1] + c.b.y, a2 = v[2] + c.b.z, a3 = v[3] + c.b.w;
        if (c.rot) {
            const float2 cs = p.cs;
            const float2 sn = p.sn;
            const float r0 = a0 * cs.x - a1 * sn.x, r1 = a1 * cs.x + a0 * sn.x;
            const float r2 = a2 * cs.y - a3 * sn.y, r3 = a3 * cs.y + a2 * sn.y;
            a0 = r0; a1 = r1; a2 = r2; a3 = r3;
        }
        if (r.cs) store4(c.dst + r.base + c.hoff, a0 * c.scale, a1 * c.scale, a2 * c.scale, a3 * c.scale);
    }
    // Staged form.  q / k: the wave tile is ONE head (64 columns = the 128 contiguous bytes of a (batch row, head, position) row), so
    // whether rotary applies is wave-uniform and the no-rotary heads (15 of 16 with pe_attn_head = 1) load nothing but the bias;
    // rotary heads request the cos / sin pairs of 2 fragment rows at a time (4: the kernel spills).  V^T: staging row = head dim (64 of them, 128 positions
    // x 2 B), out: 16 lanes x 16 B = 256 contiguous bytes of one V^T row; needs 8 aligned rows to be 8 consecutive positions of one
    // utterance (no packing, Nseq % 8 == 0).
    static constexpr bool kStaged = sizeof(TO) == 2;
    __device__ __forceinline__ bool staged_ok(bool transposed) const {
        return Nseq >= 8 && (!transposed || (rowmap == nullptr && (Nseq & 7) == 0));
    }
    template <int MI, int NJ>
    __device__ __forceinline__ void staged(char* slice, int lane, int mw, int nw, const f32x4 (&acc)[MI][NJ], bool transposed) const {
        if constexpr (sizeof(TO) == 2) {
            static_assert(MI == 8 && NJ == 4, "128 x 64 wave tile");
            const int l15 = lane & 15, g = lane >> 4;
            if (transposed) {
                float bn[NJ];
#pragma unroll
                for (int j = 0; j < NJ; ++j) bn[j] = bias[nw + j * 16 + l15];
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    const int r = j * 16 + l15;
#pragma unroll
                    for (int i = 0; i < MI; ++i) {
                        const f32x4 v = acc[i][j];
                        *reinterpret_cast<u32x2*>(slice + r * 256 + (((i * 2 + (g >> 1)) ^ (r & 15)) << 4) + ((g & 1) << 3)) =
                            pack4<TO>(v[0] + bn[j], v[1] + bn[j], v[2] + bn[j], v[3] + bn[j]);
                    }
                }
                const int rr = lane >> 4, ch = lane & 15;
                const int m = mw + ch * 8, b = m / Nseq, pos = m - b * Nseq;
                TO* dst = vt + (size_t)b * H * 64 * Npad + (size_t)(nw - 2 * H * 64 + rr) * Npad + pos;
#pragma unroll
                for (int p = 0; p < 16; ++p) {
                    const int r = p * 4 + rr;
                    *reinterpret_cast<u32x4*>(dst + (size_t)p * 4 * Npad) = *reinterpret_cast<const u32x4*>(slice + r * 256 + ((ch ^ (r & 15)) << 4));
                }
                return;
            }
            const int inner = H * 64;
            const int which = nw >= inner ? 1 : 0, h = (nw - which * inner) >> 6;
            const bool rot = h < pe_heads;
            const float scale = which ? 1.0f : q_scale;
            float4 bj[NJ];
#pragma unroll
            for (int j = 0; j < NJ; ++j) bj[j] = *reinterpret_cast<const float4*>(bias + nw + j * 16 + g * 4);
            auto put = [&](int i, int j, float a0, float a1, float a2, float a3) {
                const int r = i * 16 + l15;
                *reinterpret_cast<u32x2*>(slice + r * 128 + (((j * 2 + (g >> 1)) ^ (r & 7)) << 4) + ((g & 1) << 3)) =
                    pack4<TO>(a0 * scale, a1 * scale, a2 * scale, a3 * scale);
            };
            if (rot) {
#pragma unroll
                for (int ih = 0; ih < 4; ++ih) {
                    float4 t[2][NJ];
#pragma unroll
                    for (int i4 = 0; i4 < 2; ++i4) {
                        const int m = mw + (ih * 2 + i4) * 16 + l15;
                        int pos = m - (m / Nseq) * Nseq;
                        if (rowmap) pos = min(rowmap[m].y, Nseq - 1);   // (alignment rows of a packed utterance: any valid table row, dropped below)
                        const float4* tp = reinterpret_cast<const float4*>(rope + pos * 64 + g * 16);
#pragma unroll
                        for (int j = 0; j < NJ; ++j) t[i4][j] = tp[j];
                    }
#pragma unroll
                    for (int i4 = 0; i4 < 2; ++i4)
#pragma unroll
                        for (int j = 0; j < NJ; ++j) {
                            const f32x4 v = acc[ih * 2 + i4][j];
                            const float a0 = v[0] + bj[j].x, a1 = v[1] + bj[j].y, a2 = v[2] + bj[j].z, a3 = v[3] + bj[j].w;
                            const float2 c = make_float2(t[i4][j].x, t[i4][j].y), sv = make_float2(t[i4][j].z, t[i4][j].w);
                            put(ih * 2 + i4, j, a0 * c.x - a1 * sv.x, a1 * c.x + a0 * sv.x, a2 * c.y - a3 * sv.y, a3 * c.y + a2 * sv.y);
                        }
                }
            } else {
#pragma unroll
                for (int j = 0; j < NJ; ++j)
#pragma unroll
                    for (int i = 0; i < MI; ++i) {
                        const f32x4 v = acc[i][j];
                        put(i, j, v[0] + bj[j].x, v[1] + bj[j].y, v[2] + bj[j].z, v[3] + bj[j].w);
                    }
            }
            const int rr = lane >> 3, ch = lane & 7;
            TO* dsth = (which ? k : q) + (size_t)h * Nseq * 64 + ch * 8;
            if (rowmap) {
                int2 bp[16];
#pragma unroll
                for (int p = 0; p < 16; ++p) bp[p] = rowmap[mw + p * 8 + rr];
                __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0), visible: the guarded stores below wait for nothing
#pragma unroll
                for (int p = 0; p < 16; ++p) {
                    const int r = p * 8 + rr;
                    const u32x4 d = *reinterpret_cast<const u32x4*>(slice + r * 128 + ((ch ^ (r & 7)) << 4));
                    if (bp[p].y < Nseq) *reinterpret_cast<u32x4*>(dsth + ((size_t)bp[p].x * H * Nseq + bp[p].y) * 64) = d;
                }
            } else {
                const int m = mw + rr;
                int b = m / Nseq, pos = m - b * Nseq;
#pragma unroll
                for (int p = 0; p < 16; ++p) {
                    const int r = p * 8 + rr;
                    *reinterpret_cast<u32x4*>(dsth + ((size_t)b * H * Nseq + pos) * 64) = *reinterpret_cast<const u32x4*>(slice + r * 128 + ((ch ^ (r & 7)) << 4));
                    pos += 8;
                    if (pos >= Nseq) { pos -= Nseq; ++b; }
                }
            }
        }
    }
    __device__ __forceinline__ TRowCtx trow(int m, int M) const {
        int b = m / Nseq, pos = m - b * Nseq;
        if (rowmap) {   // rows m .. m+3 belong to one utterance (its rows start at a multiple of 4): V^T columns pos .. pos+3 < Npad
            const int2 bp = rowmap[m];
            return {(size_t)bp.x * H * 64 * Npad + bp.y, bp.y, bp.x, m + 3 < M, m, M};
        }
        return {(size_t)b * H * 64 * Npad + pos, pos, b, (pos & 3) == 0 && pos + 3 < Nseq, m, M};
    }
    __device__ __forceinline__ TColCtx tcol(int n) const {
        const int c = n - 2 * H * 64;  // = h*64 + d
        return {bias[n], (size_t)c * Npad};
    }
    __device__ __forceinline__ void tstore(const TRowCtx& r, const TColCtx& c, f32x4 v) const {
        if (r.fast) {
            store4(vt + r.base + c.hoff, v[0] + c.b, v[1] + c.b, v[2] + c.b, v[3] + c.b);
        } else {
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                const int mm = r.m + rr;
                if (mm < r.M) {
                    int bb = mm / Nseq, pp = mm - bb * Nseq;
                    if (rowmap) { const int2 bp = rowmap[mm]; bb = bp.x; pp = bp.y; }
                    vt[(size_t)bb * H * 64 * Npad + c.hoff + pp] = from_f32<TO>(v[rr] + c.b);
                }
            }
        }
    }
};

// ------------------------------------------------------------------------------------------------ kernel

template <typename T, int BM, int BN, typename Epi>
__global__ __launch_bounds__(256) void gemm_tn_kernel(const T* __restrict__ A, int lda, const T* __restrict__ W, int ldw,
                                                      int M, int N, int K, Epi epi) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int EPC = 16 / sizeof(T);                 // elements per 16-byte chunk
    constexpr int KT = GEMM_ROW_BYTES / sizeof(T);      // k elements per tile
    constexpr int MI = BM / 32, NJ = BN / 32;           // 16x16 sub-tiles per wave (wave tile = BM/2 x BN/2)
    constexpr int A_CH = BM * 8 / 256, W_CH = BN * 8 / 256;
    constexpr int BUF = (BM + BN) * GEMM_ROW_STRIDE;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    const int nkt = (K + KT - 1) / KT;
    const bool transposed = epi.tile_transposed(n0);

    u32x4 ra[A_CH], rw[W_CH];
    auto gload = [&](int kt) {
#pragma unroll
        for (int i = 0; i < A_CH; ++i) {
            const int c = tid + i * 256, row = c >> 3, cc = c & 7;
            const int ke = kt * KT + cc * EPC, m = m0 + row;
            ra[i] = (m < M && ke < K) ? *reinterpret_cast<const u32x4*>(A + (size_t)m * lda + ke) : u32x4{0u, 0u, 0u, 0u};
        }
#pragma unroll
        for (int i = 0; i < W_CH; ++i) {
            const int c = tid + i * 256, row = c >> 3, cc = c & 7;
            const int ke = kt * KT + cc * EPC, n = n0 + row;
            rw[i] = (n < N && ke < K) ? *reinterpret_cast<const u32x4*>(W + (size_t)n * ldw + ke) : u32x4{0u, 0u, 0u, 0u};
        }
    };
    auto sstore = [&](int buf) {
        char* base = smem + buf * BUF;
#pragma unroll
        for (int i = 0; i < A_CH; ++i) {
            const int c = tid + i * 256, row = c >> 3, cc = c & 7;
            *reinterpret_cast<u32x4*>(base + row * GEMM_ROW_STRIDE + cc * 16) = ra[i];
        }
#pragma unroll
        for (int i = 0; i < W_CH; ++i) {
            const int c = tid + i * 256, row = c >> 3, cc = c & 7;
            *reinterpret_cast<u32x4*>(base + (BM + row) * GEMM_ROW_STRIDE + cc * 16) = rw[i];
        }
    };

    f32x4 acc[MI][NJ];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    gload(0);
    sstore(0);
    __syncthreads();
    const int frag_off = (lane & 15) * GEMM_ROW_STRIDE + (lane >> 4) * 16;
    for (int kt = 0; kt < nkt; ++kt) {
        if (kt + 1 < nkt) gload(kt + 1);
        const char* As = smem + (kt & 1) * BUF + (wr * (BM / 2)) * GEMM_ROW_STRIDE + frag_off;
        const char* Ws = smem + (kt & 1) * BUF + (BM + wc * (BN / 2)) * GEMM_ROW_STRIDE + frag_off;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            u32x4 af[MI], wf[NJ];
#pragma unroll
            for (int i = 0; i < MI; ++i) af[i] = *reinterpret_cast<const u32x4*>(As + i * 16 * GEMM_ROW_STRIDE + kk * 64);
#pragma unroll
            for (int j = 0; j < NJ; ++j) wf[j] = *reinterpret_cast<const u32x4*>(Ws + j * 16 * GEMM_ROW_STRIDE + kk * 64);
            if (!transposed) {
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int j = 0; j < NJ; ++j) acc[i][j] = Mma<T>::run(wf[j], af[i], acc[i][j]);
            } else {
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int j = 0; j < NJ; ++j) acc[i][j] = Mma<T>::run(af[i], wf[j], acc[i][j]);
            }
        }
        if (kt + 1 < nkt) sstore((kt + 1) & 1);
        __syncthreads();
    }

    const int mw = m0 + wr * (BM / 2), nw = n0 + wc * (BN / 2);
    if (!transposed) {
        typename Epi::ColCtx cc[NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j) cc[j] = epi.col(min(nw + j * 16 + (lane >> 4) * 4, N - 4));
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            const int m = mw + i * 16 + (lane & 15);
            if (m >= M) continue;
            const typename Epi::RowCtx rc = epi.row(m);
#pragma unroll
            for (int j = 0; j < NJ; ++j)
                if (nw + j * 16 + (lane >> 4) * 4 < N) epi.store(rc, cc[j], acc[i][j], epi.preload(rc, cc[j]));
        }
    } else {
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            const int m = mw + i * 16 + (lane >> 4) * 4;
            if (m >= M) continue;
            const typename Epi::TRowCtx rc = epi.trow(m, M);
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const int n = nw + j * 16 + (lane & 15);
                if (n < N) epi.tstore(rc, epi.tcol(n), acc[i][j]);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------- launcher

struct GemmTile { int bm, bn; };

inline GemmTile pick_tile(int M, int N) {
    auto blocks = [&](int bm, int bn) { return (long)((M + bm - 1) / bm) * ((N + bn - 1) / bn); };
    if (M > 64 && N > 64 && blocks(128, 128) >= 200) return {128, 128};
    if (M > 64 && blocks(128, 64) >= 160) return {128, 64};
    return {64, 64};
}

template <typename T, int BM, int BN, typename Epi>
inline hipError_t launch_gemm_tile_raw(hipStream_t s, const T* A, int lda, const T* W, int ldw, int M, int N, int K,
                                       const Epi& epi) {
    constexpr int smem = 2 * (BM + BN) * GEMM_ROW_STRIDE;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_tn_kernel<T, BM, BN, Epi>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, smem);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    dim3 grid((N + BN - 1) / BN, (M + BM - 1) / BM);
    hipLaunchKernelGGL((gemm_tn_kernel<T, BM, BN, Epi>), grid, dim3(256), smem, s, A, lda, W, ldw, M, N, K, epi);
    return hipGetLastError();
}

template <typename T, int BM, int BN, typename Epi>
inline hipError_t launch_gemm_tile(hipStream_t s, const T* A, int lda, const T* W, int ldw, int M, int N, int K,
                                   const Epi& epi) {
    return with_static_act(epi, [&](const auto& e) {
        return launch_gemm_tile_raw<T, BM, BN, std::decay_t<decltype(e)>>(s, A, lda, W, ldw, M, N, K, e);
    });
}

// A: [M, K] (lda elements), W: [N, K] (ldw elements); K, lda, ldw multiples of 16/sizeof(T); N multiple of 4.
template <typename T, typename Epi>
inline hipError_t launch_gemm_v1(hipStream_t s, const T* A, int lda, const T* W, int ldw, int M, int N, int K,
                                 const Epi& epi, int force_bm = 0, int force_bn = 0) {
    if (M <= 0 || N <= 0) return hipSuccess;
    GemmTile t = pick_tile(M, N);
    if (force_bm) t = {force_bm, force_bn};
    if (t.bm == 128 && t.bn == 128) return launch_gemm_tile<T, 128, 128, Epi>(s, A, lda, W, ldw, M, N, K, epi);
    if (t.bm == 128 && t.bn == 64) return launch_gemm_tile<T, 128, 64, Epi>(s, A, lda, W, ldw, M, N, K, epi);
    return launch_gemm_tile<T, 64, 64, Epi>(s, A, lda, W, ldw, M, N, K, epi);
}

}  // namespace f5
