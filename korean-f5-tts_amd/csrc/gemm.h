// MFMA "TN" GEMM for gfx950:  C[M,N] = A[M,K] * W[N,K]^T  with fused epilogues.
//
// Both operands are K-contiguous (activations [rows, features]; torch nn.Linear weights [out, in]), so a 16-byte
// chunk of either is directly one MFMA operand fragment:
//   bf16: v_mfma_f32_16x16x32_bf16 -- lane l holds row (l&15), k = 8*(l>>4)..+7            (one fragment = 1 MFMA)
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
template <> struct Mma<float> {
    static __device__ __forceinline__ f32x4 run(const u32x4& a, const u32x4& b, f32x4 c) {
        const f32x4 fa = __builtin_bit_cast(f32x4, a), fb = __builtin_bit_cast(f32x4, b);
#pragma unroll
        for (int j = 0; j < 4; ++j) c = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[j], fb[j], c, 0, 0, 0);
        return c;
    }
};

// ---------------------------------------------------------------------------------------------- epilogues
// row4(m, n, v): v[r] = C[m][n + r]           (row-packed orientation; n % 4 == 0)
// col4(m, n, v): v[r] = C[m + r][n]           (transposed orientation; m % 4 == 0) -- only if tile_transposed()

template <typename TO> struct EpiStore {  // out = act(acc + bias)
    TO* out; int ldo; const float* bias; int act;
    __device__ __forceinline__ bool tile_transposed(int) const { return false; }
    __device__ __forceinline__ void row4(int m, int n, f32x4 v, int, int) const {
        float4 b = bias ? *reinterpret_cast<const float4*>(bias + n) : make_float4(0, 0, 0, 0);
        store4(out + (size_t)m * ldo + n, apply_act(v[0] + b.x, act), apply_act(v[1] + b.y, act),
               apply_act(v[2] + b.z, act), apply_act(v[3] + b.w, act));
    }
    __device__ __forceinline__ void col4(int, int, f32x4, int, int) const {}
};

// x[m][n] = res[m][n] + gate[b(m)][n] * (acc + bias)    (res may alias x; gate == nullptr -> 1; rows m with
// (m % rows_per_batch) >= lens[m / rows_per_batch] are left as res: the reference's masked_fill(~mask, 0) on the
// attention output, modules.py:540-542)
struct EpiGateRes {
    float* x; const float* res; int ld; const float* bias; const float* gate; int gate_stride; int rows_per_batch;
    const int* lens;
    __device__ __forceinline__ bool tile_transposed(int) const { return false; }
    __device__ __forceinline__ void row4(int m, int n, f32x4 v, int, int) const {
        const int b = m / rows_per_batch;
        float4 r = *reinterpret_cast<const float4*>(res + (size_t)m * ld + n);
        if (!(lens && (m - b * rows_per_batch) >= lens[b])) {
            float4 bi = bias ? *reinterpret_cast<const float4*>(bias + n) : make_float4(0, 0, 0, 0);
            float4 g = gate ? *reinterpret_cast<const float4*>(gate + (size_t)b * gate_stride + n)
                            : make_float4(1, 1, 1, 1);
            r.x += g.x * (v[0] + bi.x); r.y += g.y * (v[1] + bi.y);
            r.z += g.z * (v[2] + bi.z); r.w += g.w * (v[3] + bi.w);
        }
        *reinterpret_cast<float4*>(x + (size_t)m * ld + n) = r;
    }
    __device__ __forceinline__ void col4(int, int, f32x4, int, int) const {}
};

// Fused QKV projection epilogue (modules.py:469-497): bias, interleaved-pair rotary on the first `pe_heads` heads of
// q and k, q pre-scaled by dim_head^-0.5, head split.  q,k -> [B', H, Nseq, 64]; v -> TRANSPOSED [B', H, 64, Npad]
// (so that both attention B-operands are K-contiguous).  Column tiles of the V third use the transposed orientation.
template <typename TO> struct EpiQKV {
    TO* q; TO* k; TO* vt; const float* bias; const float* rope_cos; const float* rope_sin;  // [maxpos][32]
    int Nseq, Npad, H, pe_heads; float q_scale;
    __device__ __forceinline__ bool tile_transposed(int n0) const { return n0 >= 2 * H * 64; }
    __device__ __forceinline__ void row4(int m, int n, f32x4 v, int, int) const {
        const int inner = H * 64;
        const int which = n / inner, c = n - which * inner, h = c >> 6, d = c & 63;
        const int b = m / Nseq, pos = m - b * Nseq;
        const float4 bi = *reinterpret_cast<const float4*>(bias + n);
        float a0 = v[0] + bi.x, a1 = v[1] + bi.y, a2 = v[2] + bi.z, a3 = v[3] + bi.w;
        if (h < pe_heads) {
            const float2 cs = *reinterpret_cast<const float2*>(rope_cos + pos * 32 + (d >> 1));
            const float2 sn = *reinterpret_cast<const float2*>(rope_sin + pos * 32 + (d >> 1));
            const float r0 = a0 * cs.x - a1 * sn.x, r1 = a1 * cs.x + a0 * sn.x;
            const float r2 = a2 * cs.y - a3 * sn.y, r3 = a3 * cs.y + a2 * sn.y;
            a0 = r0; a1 = r1; a2 = r2; a3 = r3;
        }
        TO* dst = which == 0 ? q : k;
        if (which == 0) { a0 *= q_scale; a1 *= q_scale; a2 *= q_scale; a3 *= q_scale; }
        store4(dst + (((size_t)b * H + h) * Nseq + pos) * 64 + d, a0, a1, a2, a3);
    }
    __device__ __forceinline__ void col4(int m, int n, f32x4 v, int M, int) const {
        const int c = n - 2 * H * 64, h = c >> 6, d = c & 63;
        const float bi = bias[n];
        const int b = m / Nseq, pos = m - b * Nseq;
        if ((pos & 3) == 0 && pos + 3 < Nseq) {
            store4(vt + (((size_t)b * H + h) * 64 + d) * Npad + pos, v[0] + bi, v[1] + bi, v[2] + bi, v[3] + bi);
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int mm = m + r;
                if (mm < M) {
                    const int bb = mm / Nseq, pp = mm - bb * Nseq;
                    vt[(((size_t)bb * H + h) * 64 + d) * Npad + pp] = from_f32<TO>(v[r] + bi);
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
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            const int m = mw + i * 16 + (lane & 15);
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const int n = nw + j * 16 + (lane >> 4) * 4;
                if (m < M && n < N) epi.row4(m, n, acc[i][j], M, N);
            }
        }
    } else {
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            const int m = mw + i * 16 + (lane >> 4) * 4;
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const int n = nw + j * 16 + (lane & 15);
                if (m < M && n < N) epi.col4(m, n, acc[i][j], M, N);
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
inline hipError_t launch_gemm_tile(hipStream_t s, const T* A, int lda, const T* W, int ldw, int M, int N, int K,
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
