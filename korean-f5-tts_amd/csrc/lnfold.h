// AdaLayerNorm folded into its neighbouring GEMMs (bf16 precision, one modulation vector per launch):
//
//   reference (modules.py:314-320,335-341 + the nn.Linear that follows):
//       y[m][n] = sum_k ( (x[m][k] - mu_m) r_m (1 + s_k) + t_k ) W[n][k] + b[n]            mu, r = LayerNorm statistics of row m
//   folded:
//       A'[m][k] = x[m][k] (1 + s_k)                       written (as the MFMA operand type) by whoever produces x
//       y[m][n]  = r_m ( sum_k A'[m][k] W[n][k]  -  mu_m c1[n] )  +  c2[n]
//       c1[n] = sum_k (1 + s_k) W[n][k],   c2[n] = sum_k t_k W[n][k] + b[n]                  (s, t depend on the ODE step only:
//                                                                                              computed for all steps at once)
// so the two LayerNorm launches of a DiT block disappear: the producer's GEMM epilogue (EpiGateResLN) also writes A' and
// per-row partial sums (sum x, sum x^2 over the columns its wave owns; plain stores, no atomics -> deterministic), and the
// consumer's GEMM (EpiFold<EpiQKV> / EpiFold<EpiStore>) reduces the partials of its rows once per workgroup into LDS
// (block_prologue) and applies the affine correction before its own epilogue.  A launch boundary costs ~5 us on this
// part and a software grid barrier far more (tools/grid_barrier_probe.py), which is why the
// fusion is algebraic rather than a persistent kernel.
//
// Statistics buffer: stat[row][LNFOLD_SLOTS][2] floats; a producer wave whose first column is nw writes slot nw / 16;
// the consumer sums the slots {0, stride, 2 stride, ...} (stride = producer wave-tile width / 16).
#pragma once
#include "gemm2.h"

namespace f5 {

constexpr int LNFOLD_SLOTS = 64;   // dim / 16 for dim = 1024

// ---- producer: EpiGateRes + A' + row partial sums.  next_scale == nullptr: plain EpiGateRes behaviour.
template <typename TA> struct EpiGateResLN {
    float* x; const float* res; int ld; const float* bias; const float* gate; int gate_stride; int rows_per_batch;
    const int* lens;
    TA* xs; int ldxs; const float* next_scale; float* stat;
    struct RowCtx { size_t off; const float* g; bool masked; int m; float s1, s2; };
    struct ColCtx { float4 b; float4 ns; int n; };
    typedef NoCtx TRowCtx;
    typedef NoCtx TColCtx;
    static constexpr bool kTransposes = false;
    static constexpr int kRowLdsFloats = 0;
    static constexpr bool kRowDone = true;
    __device__ __forceinline__ bool tile_transposed(int) const { return false; }
    __device__ __forceinline__ RowCtx row(int m) const {
        const int b = m / rows_per_batch;
        return {(size_t)m * ld, gate ? gate + (size_t)b * gate_stride : nullptr,
                lens ? (m - b * rows_per_batch) >= lens[b] : false, m, 0.f, 0.f};
    }
    __device__ __forceinline__ ColCtx col(int n) const {
        float4 ns = next_scale ? *reinterpret_cast<const float4*>(next_scale + n) : make_float4(0, 0, 0, 0);
        ns.x += 1.f; ns.y += 1.f; ns.z += 1.f; ns.w += 1.f;
        return {bias ? *reinterpret_cast<const float4*>(bias + n) : make_float4(0, 0, 0, 0), ns, n};
    }
    struct Pre { float4 r, g; };
    __device__ __forceinline__ Pre preload(const RowCtx& rc, const ColCtx& c) const {
        return {*reinterpret_cast<const float4*>(res + rc.off + c.n),
                rc.g ? *reinterpret_cast<const float4*>(rc.g + c.n) : make_float4(1, 1, 1, 1)};
    }
    __device__ __forceinline__ void store(RowCtx& rc, const ColCtx& c, f32x4 v, const Pre& p) const {
        float4 r = p.r;
        if (!rc.masked) {
            const float4 g = p.g;
            r.x += g.x * (v[0] + c.b.x); r.y += g.y * (v[1] + c.b.y);
            r.z += g.z * (v[2] + c.b.z); r.w += g.w * (v[3] + c.b.w);
        }
        *reinterpret_cast<float4*>(x + rc.off + c.n) = r;
        if (next_scale) {
            store4(xs + (size_t)rc.m * ldxs + c.n, r.x * c.ns.x, r.y * c.ns.y, r.z * c.ns.z, r.w * c.ns.w);
            rc.s1 += (r.x + r.y) + (r.z + r.w);
            rc.s2 += (r.x * r.x + r.y * r.y) + (r.z * r.z + r.w * r.w);
        }
    }
    // all 64 lanes call this (wave-uniform control flow); `valid` = this lane's row exists
    __device__ __forceinline__ void row_done(const RowCtx& rc, int slot, bool valid) const {
        if (!next_scale) return;
        float s1 = rc.s1, s2 = rc.s2;   // the four lane groups of a row hold different columns: sum them
        s1 += __shfl_xor(s1, 16, 64); s2 += __shfl_xor(s2, 16, 64);
        s1 += __shfl_xor(s1, 32, 64); s2 += __shfl_xor(s2, 32, 64);
        if (valid && (threadIdx.x & 48) == 0)
            *reinterpret_cast<float2*>(stat + ((size_t)rc.m * LNFOLD_SLOTS + slot) * 2) = make_float2(s1, s2);
    }
    __device__ __forceinline__ NoCtx trow(int, int) const { return {}; }
    __device__ __forceinline__ NoCtx tcol(int) const { return {}; }
    __device__ __forceinline__ void tstore(const NoCtx&, const NoCtx&, f32x4) const {}
};

// ---- consumer: y = r_m (acc - mu_m c1[n]) + (inner epilogue with bias := c2)
template <typename Inner> struct EpiFold {
    Inner inner;                // its bias pointer must be c2
    const float* c1;            // [N]
    const float* stat;          // [rows][LNFOLD_SLOTS][2]
    int nparts, part_stride;    // partial slots to sum: 0, part_stride, ..., (nparts-1) part_stride
    float inv_d, eps;
    struct RowCtx { typename Inner::RowCtx in; float mu, r; };
    struct ColCtx { typename Inner::ColCtx in; float4 c1; };
    struct TRowCtx { typename Inner::TRowCtx in; float mu[4], r[4]; };
    struct TColCtx { typename Inner::TColCtx in; float c1; };
    typedef typename Inner::Pre Pre;
    static constexpr bool kTransposes = Inner::kTransposes;
    static constexpr int kRowLdsFloats = 2;
    static constexpr bool kRowDone = false;
    __device__ __forceinline__ bool tile_transposed(int n0) const { return inner.tile_transposed(n0); }
    // G = nthreads / rows adjacent threads per row of the workgroup's stripe reduce the producer's partial sums ->
    // (mu, rstd) in LDS.  The loads of a thread are independent and issued together (a serial chain of 32 L2 round trips
    // per GEMM launch cost more than the LayerNorm launch this replaces).
    __device__ __forceinline__ void block_prologue(float* lds, int m0, int rows, int M, int tid, int nthreads) const {
        int G = nthreads / rows;             // 2, 4 or 8 for the tile shapes in use (power of two)
        G = G < 1 ? 1 : (G > 8 ? 8 : G);
        for (int base = 0; base < rows * G; base += nthreads) {
            const int t = base + tid;
            const int row = t / G, sub = t - row * G;
            const int m = min(m0 + min(row, rows - 1), M - 1);
            const float2* p = reinterpret_cast<const float2*>(stat) + (size_t)m * LNFOLD_SLOTS;
            float s1 = 0.f, s2 = 0.f;
            for (int i0 = 0; i0 < nparts; i0 += 8 * G) {
                float2 v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int i = i0 + sub + u * G;
                    v[u] = i < nparts ? p[i * part_stride] : make_float2(0.f, 0.f);
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) { s1 += v[u].x; s2 += v[u].y; }
            }
            for (int o = 1; o < G; o <<= 1) {   // the G threads of a row are adjacent lanes
                s1 += __shfl_xor(s1, o, 64);
                s2 += __shfl_xor(s2, o, 64);
            }
            if (sub == 0 && row < rows) {
                const float mu = s1 * inv_d;
                const float var = fmaxf(s2 * inv_d - mu * mu, 0.f);
                lds[2 * row] = mu;
                lds[2 * row + 1] = rsqrtf(var + eps);
            }
        }
    }
    __device__ __forceinline__ RowCtx row_lds(int m, const float* l) const { return {inner.row(m), l[0], l[1]}; }
    __device__ __forceinline__ ColCtx col(int n) const { return {inner.col(n), *reinterpret_cast<const float4*>(c1 + n)}; }
    __device__ __forceinline__ Pre preload(const RowCtx& r, const ColCtx& c) const { return inner.preload(r.in, c.in); }
    __device__ __forceinline__ void store(const RowCtx& r, const ColCtx& c, f32x4 v, const Pre& p) const {
        const float a = r.r, b = -r.r * r.mu;
        f32x4 y = {a * v[0] + b * c.c1.x, a * v[1] + b * c.c1.y, a * v[2] + b * c.c1.z, a * v[3] + b * c.c1.w};
        inner.store(r.in, c.in, y, p);
    }
    // transposed orientation: v[rr] = C[m + rr][n]; `l` points at the LDS pair of row m, lrows = rows staged past it
    __device__ __forceinline__ TRowCtx trow_lds(int m, int M, const float* l, int lrows) const {
        TRowCtx t;
        t.in = inner.trow(m, M);
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            const int o = rr < lrows ? rr : lrows - 1;
            t.mu[rr] = l[2 * o];
            t.r[rr] = l[2 * o + 1];
        }
        return t;
    }
    __device__ __forceinline__ TColCtx tcol(int n) const { return {inner.tcol(n), c1[n]}; }
    __device__ __forceinline__ void tstore(const TRowCtx& r, const TColCtx& c, f32x4 v) const {
        f32x4 y;
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) y[rr] = r.r[rr] * (v[rr] - r.mu[rr] * c.c1);
        inner.tstore(r.in, c.in, y);
    }
};

template <typename TO, typename F> inline hipError_t with_static_act(const EpiFold<EpiStore<TO, -1>>& e, F&& f) {
    return with_static_act(e.inner, [&](const auto& in) {
        return f(EpiFold<std::decay_t<decltype(in)>>{in, e.c1, e.stat, e.nparts, e.part_stride, e.inv_d, e.eps});
    });
}

// ---- c1 / c2 for every ODE step.  One wave owns 8 output features (their W rows stay in registers for all steps); the
// step's (1 + scale) and shift vectors are staged once per workgroup in LDS.  K <= 1024, K % 64 == 0.
//   c1[s][n] = sum_k (1 + scale[s][k]) W[n][k],  c2[s][n] = sum_k shift[s][k] W[n][k] + bias[n]
template <typename T>
__global__ __launch_bounds__(256) void fold_vectors_kernel(const T* __restrict__ W, int ldw, const float* __restrict__ bias,
                                                           int N, int K, const float* __restrict__ scale,
                                                           const float* __restrict__ shift, long mod_step_stride, int steps,
                                                           float* __restrict__ c1, float* __restrict__ c2, long out_step_stride) {
    __shared__ float sv[2][1024];
    constexpr int RPW = 8;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n0 = (blockIdx.x * 4 + wave) * RPW;
    const int per = K / 64;              // values per lane, k = i * 64 + lane (coalesced across the wave)
    float w[RPW][16];
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
        const int n = min(n0 + r, N - 1);
#pragma unroll
        for (int i = 0; i < 16; ++i) w[r][i] = i < per ? (float)W[(size_t)n * ldw + i * 64 + lane] : 0.f;
    }
    for (int s = 0; s < steps; ++s) {
        __syncthreads();
        for (int k = threadIdx.x; k < K; k += 256) {
            sv[0][k] = 1.f + scale[(size_t)s * mod_step_stride + k];
            sv[1][k] = shift[(size_t)s * mod_step_stride + k];
        }
        __syncthreads();
        float a[16], b[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            a[i] = i < per ? sv[0][i * 64 + lane] : 0.f;
            b[i] = i < per ? sv[1][i * 64 + lane] : 0.f;
        }
#pragma unroll
        for (int r = 0; r < RPW; ++r) {
            float a1 = 0.f, a2 = 0.f;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                a1 += a[i] * w[r][i];
                a2 += b[i] * w[r][i];
            }
            a1 = wave_sum(a1);
            a2 = wave_sum(a2);
            if (lane == 0 && n0 + r < N) {
                c1[(size_t)s * out_step_stride + n0 + r] = a1;
                c2[(size_t)s * out_step_stride + n0 + r] = a2 + (bias ? bias[n0 + r] : 0.f);
            }
        }
    }
}

// ---- first LayerNorm of a forward: x comes from the input embedding, not from an EpiGateResLN.
// One wave per row: A' = x (1 + scale), statistics into slot 0 (the consumer is told nparts = 1).
template <typename TA>
__global__ __launch_bounds__(256) void fold_prep_kernel(const float* __restrict__ x, int ldx, TA* __restrict__ xs, int ldxs,
                                                        int R, int D, const float* __restrict__ scale,
                                                        float* __restrict__ stat) {
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= R) return;
    float s1 = 0.f, s2 = 0.f;
    for (int c = lane * 4; c < D; c += 256) {
        const float4 v = *reinterpret_cast<const float4*>(x + (size_t)r * ldx + c);
        const float4 sc = *reinterpret_cast<const float4*>(scale + c);
        store4(xs + (size_t)r * ldxs + c, v.x * (1.f + sc.x), v.y * (1.f + sc.y), v.z * (1.f + sc.z), v.w * (1.f + sc.w));
        s1 += (v.x + v.y) + (v.z + v.w);
        s2 += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
    }
    s1 = wave_sum(s1);
    s2 = wave_sum(s2);
    if (lane == 0) *reinterpret_cast<float2*>(stat + (size_t)r * LNFOLD_SLOTS * 2) = make_float2(s1, s2);
}

}  // namespace f5
