// Grouped Conv1d(k=31, groups=16, same padding) + Mish as an implicit GEMM on MFMA -- the reference's
// ConvPositionEmbedding (modules.py:170-196) in token-major layout, no [B,D,N] permute.
//
//   Y[b, n, co] = mish( bias[co] + sum_{tap<31} sum_{ci<CPG} Wp[co][tap*CPG + ci] * X[b, n + tap - 15, g*CPG + ci] )  (+ res)
//
// One block = one (batch row, group, 128-token tile).  The block's input window (128 + 30 halo tokens x CPG channels)
// is converted to the MFMA operand type and kept in LDS for all 31 taps; the packed weights Wp[co][31*CPG] (engine
// repacks torch's [co][ci][tap] at load time) stream through an 8-stage LDS-DMA ring exactly like gemm2.h's W operand
// (global_load_lds, XOR-swizzled unpadded [rows][128 B] image, counted vmcnt, one raw barrier per K-tile).  The K loop
// is 31 short steps (16 MFMAs per wave each), so it is bound by how far ahead the weight tiles are requested: with a
// register-staged double buffer every step waited a full L2 round trip (33 us per launch at D = 1024, N = 1024).
// The im2col row of a token is never materialised: the fragment for k-chunk (tap, ci0) of token t is simply the
// 16 bytes at LDS row (t + tap), column ci0.
// Masking (batched inference, modules.py:187-192): tokens >= lens[b] read as zero and produce zero.
// row_start (RowPack, may be null): batch row b' owns rows row_start[b'] .. row_start[b'+1]-1 of X / res / Y (packed
// variable-length batch) instead of b' N .. b' N + N - 1; token tiles past that span retire at once.
#pragma once
#include "gemm2.h"

namespace f5 {

// SPLIT (T = float, CPG % 32 == 0; F5_PREC_F16X3): products on the f16 pipe with split operands as in gemm2.h MODE 3 -- the weights
// arrive pre-split (split_planar_kernel: per 32-element K block, which lies inside one tap, chunk g = hi of k = 4g..4g+3,
// 16+4g..16+4g+3, chunk 4+g = lo), the input window is split once while it is staged (per row: CPG f16 hi, then CPG f16 lo).
template <typename T, int CPG, int NS, bool SPLIT = false>
__global__ __launch_bounds__(256) void convpos_kernel(const float* __restrict__ X, const T* __restrict__ Wp, int Kp,
                                                      const float* __restrict__ bias, const float* __restrict__ res,
                                                      float* __restrict__ Y, int N, int D, const int* __restrict__ lens,
                                                      int nbatch_lens, const int* __restrict__ row_start) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const size_t rbase = row_start ? (size_t)row_start[blockIdx.z] : (size_t)blockIdx.z * N;   // first row of this batch row
    const int nrows = row_start ? row_start[blockIdx.z + 1] - row_start[blockIdx.z] : N;        // rows it owns
    if ((int)blockIdx.x * 128 >= nrows) return;         // (block-uniform, before any barrier)
    constexpr int TM = 128;
    constexpr int XR = TM + 32;                       // rows of the input window kept in LDS
    constexpr int XRB = CPG * sizeof(T);              // bytes per window row
    constexpr int XRS = XRB + 16;
    constexpr int KT = GEMM_ROW_BYTES / sizeof(T);    // k elements per weight tile (128 bytes)
    constexpr int EPC = 16 / sizeof(T);
    constexpr int NJ = CPG / 16;
    // NS = weight ring stages: 8 when the grid is one workgroup per CU (latency-bound, deep prefetch), 4 for large grids
    // (55 KB of LDS: two workgroups per CU cover each other's prologue / epilogue)
    constexpr int WTILE = CPG * GEMM_ROW_BYTES;       // bytes of one weight K-tile (CPG rows x 128 B)
    constexpr int PIECES = CPG / 8;                   // 1 KiB LDS-DMA pieces per tile
    constexpr int L = (PIECES + 3) / 4;               // pieces per wave per tile (pieces % 4 != 0: some are staged twice, same bytes)
    char* ws = smem;                                  // ring first: 1 KiB-aligned DMA destinations
    char* xs = smem + NS * WTILE;

    const int tid = threadIdx.x, lane = tid & 63, l15 = lane & 15, g = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tok0 = blockIdx.x * TM, grp = blockIdx.y, b = blockIdx.z;
    const int len = lens ? min(N, lens[b % nbatch_lens]) : N;
    const int nkt = Kp / KT;                          // the engine pads Kp to whole K-tiles with zero weights

    // ---- weight ring: wave w stages pieces (w + 4 i) % PIECES of every tile; lane -> row lane>>3 of the piece, source
    // chunk (lane & 7) ^ (row & 7) so that chunk c of row r lands in slot c ^ (r & 7)
    const int lr = lane >> 3;
    const T* wsrc[L];
    int wdst[L];
#pragma unroll
    for (int i = 0; i < L; ++i) {
        const int piece = (wave + 4 * i) % PIECES;
        const int row = piece * 8 + lr;
        wsrc[i] = Wp + (size_t)(grp * CPG + row) * Kp + ((lane & 7) ^ (row & 7)) * EPC;
        wdst[i] = piece * 1024;
    }
    auto issue = [&](int kt, int stage) {
        if (kt >= nkt) return;
#pragma unroll
        for (int i = 0; i < L; ++i) glds16(wsrc[i] + (size_t)kt * KT, ws + stage * WTILE + wdst[i]);
    };
    auto wait_tiles = [&](int tiles) {  // at most `tiles` of this wave's requested tiles may still be in flight
        switch (tiles < NS - 2 ? tiles : NS - 2) {
            case 0: wait_vmcnt<0>(); break;
            case 1: wait_vmcnt<1 * L>(); break;
            case 2: wait_vmcnt<2 * L>(); break;
            case 3: wait_vmcnt<(NS > 4 ? 3 : 2) * L>(); break;
            case 4: wait_vmcnt<(NS > 4 ? 4 : 2) * L>(); break;
            case 5: wait_vmcnt<(NS > 4 ? 5 : 2) * L>(); break;
            default: wait_vmcnt<(NS > 4 ? 6 : 2) * L>(); break;
        }
    };
#pragma unroll
    for (int t = 0; t < NS - 1; ++t) issue(t, t);

    // ---- stage the input window (fp32 -> T) while the first weight tiles are in flight.  ALL of a thread's loads are issued
    // before the first store, from clamped (always valid) addresses: as a load-if-in-range / store loop hipcc emitted
    // `global_load; s_waitcnt vmcnt(0); ds_write` per iteration -- ten serialised memory round trips (and a wait for the
    // weight ring's prologue) before the first MFMA of a 24 us kernel.
    constexpr int XCH = XR * (CPG / 4), XIT = (XCH + 255) / 256;
    float4 xv[XIT];
#pragma unroll
    for (int i = 0; i < XIT; ++i) {
        const int c = min(tid + i * 256, XCH - 1), row = c / (CPG / 4), c4 = c % (CPG / 4);
        const int tok = tok0 - 15 + row, tokc = max(min(tok, len - 1), 0);
        xv[i] = *reinterpret_cast<const float4*>(X + (rbase + tokc) * D + grp * CPG + c4 * 4);
    }
#pragma unroll
    for (int i = 0; i < XIT; ++i) {
        const int c = tid + i * 256, row = c / (CPG / 4), c4 = c % (CPG / 4);
        const int tok = tok0 - 15 + row;
        const bool in = tok >= 0 && tok < len;      // (rows outside the sequence are zero: conv padding / masked rows)
        if (c < XCH) {
            const float4 v = make_float4(in ? xv[i].x : 0.f, in ? xv[i].y : 0.f, in ? xv[i].z : 0.f, in ? xv[i].w : 0.f);
            if constexpr (SPLIT) {
                u32x2 hi, lo;
                split4_f16(__builtin_bit_cast(u32x4, f32x4{v.x, v.y, v.z, v.w}), hi, lo);
                *reinterpret_cast<u32x2*>(xs + row * XRS + c4 * 8) = hi;
                *reinterpret_cast<u32x2*>(xs + row * XRS + CPG * 2 + c4 * 8) = lo;
            } else {
                store4(reinterpret_cast<T*>(xs + row * XRS) + c4 * 4, v.x, v.y, v.z, v.w);
            }
        }
    }
    __syncthreads();   // (also retires the window's global loads, which are older than nothing the ring counts below:
                       //  the compiler waits vmcnt(0) for them, i.e. for the first NS-1 weight tiles too -- once)

    f32x4 acc[2][NJ];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int wsw = l15 & 7;
    int stage = 0;
    for (int kt = 0; kt < nkt; ++kt) {
        wait_tiles(nkt - 1 - kt);          // this wave's pieces of tile kt have landed ...
        __builtin_amdgcn_s_barrier();      // ... everyone's have, and stage (kt-1) % NS is free again
        int pf = stage + NS - 1;
        if (pf >= NS) pf -= NS;
        issue(kt + NS - 1, pf);
        const char* Ws = ws + stage * WTILE + l15 * GEMM_ROW_BYTES;
        if constexpr (SPLIT) {
            // one 32-deep block of tap `tap`, channels cib .. cib+31: slot order of lane group g = {cib + 4g ..+3, cib + 16 + 4g ..+3}
            const int kidx = kt * KT, tap = kidx / CPG, cib = kidx - tap * CPG;
            u32x4 xh[2], xl[2], wh[NJ], wl[NJ];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const char* xr = xs + (wave * 32 + i * 16 + l15 + tap) * XRS + (cib + 4 * g) * 2;
                const u32x2 a0 = *reinterpret_cast<const u32x2*>(xr), a1 = *reinterpret_cast<const u32x2*>(xr + 32);
                const u32x2 b0 = *reinterpret_cast<const u32x2*>(xr + CPG * 2), b1 = *reinterpret_cast<const u32x2*>(xr + CPG * 2 + 32);
                xh[i] = u32x4{a0.x, a0.y, a1.x, a1.y};
                xl[i] = u32x4{b0.x, b0.y, b1.x, b1.y};
            }
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                wh[j] = *reinterpret_cast<const u32x4*>(Ws + j * 16 * GEMM_ROW_BYTES + ((g ^ wsw) * 16));
                wl[j] = *reinterpret_cast<const u32x4*>(Ws + j * 16 * GEMM_ROW_BYTES + (((4 + g) ^ wsw) * 16));
            }
#pragma unroll
            for (int term = 0; term < 3; ++term)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < NJ; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, term == 0 ? wl[j] : wh[j]),
                                                                           __builtin_bit_cast(f16x8, term == 1 ? xl[i] : xh[i]), acc[i][j], 0, 0, 0);
        } else {
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            // im2col k index of this lane's 16-byte chunk -> (tap, ci0)
            const int kidx = kt * KT + kk * (KT / 2) + g * EPC;
            const int tap = kidx / CPG, ci0 = kidx - tap * CPG;
            u32x4 xf[2], wf[NJ];
#pragma unroll
            for (int i = 0; i < 2; ++i)
                xf[i] = *reinterpret_cast<const u32x4*>(xs + (wave * 32 + i * 16 + l15 + tap) * XRS + ci0 * sizeof(T));
#pragma unroll
            for (int j = 0; j < NJ; ++j)
                wf[j] = *reinterpret_cast<const u32x4*>(Ws + j * 16 * GEMM_ROW_BYTES + (((kk * 4 + g) ^ wsw) * 16));
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j) acc[i][j] = Mma<T>::run(wf[j], xf[i], acc[i][j]);
        }
        }
        stage = stage + 1 == NS ? 0 : stage + 1;
    }

    // epilogue operands first (clamped rows, no bounds branches around the loads), ONE visible wait, then the guarded
    // stores: a store that is the first use of a pending load inside an exec-masked block makes hipcc wait vmcnt(0) -- for
    // the previous store's round trip -- in front of every store (gemm2.h)
    float4 bi[NJ], rv[2][NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) bi[j] = *reinterpret_cast<const float4*>(bias + grp * CPG + j * 16 + g * 4);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int tokc = min(tok0 + wave * 32 + i * 16 + l15, nrows - 1);
#pragma unroll
        for (int j = 0; j < NJ; ++j)
            rv[i][j] = res ? *reinterpret_cast<const float4*>(res + (rbase + tokc) * D + grp * CPG + j * 16 + g * 4)
                           : make_float4(0, 0, 0, 0);
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int tok = tok0 + wave * 32 + i * 16 + l15;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int co = grp * CPG + j * 16 + g * 4;
            float4 v = make_float4(acc[i][j][0] + bi[j].x, acc[i][j][1] + bi[j].y, acc[i][j][2] + bi[j].z, acc[i][j][3] + bi[j].w);
            if (tok >= len) v = make_float4(0, 0, 0, 0);
            v.x = mish(v.x) + rv[i][j].x; v.y = mish(v.y) + rv[i][j].y; v.z = mish(v.z) + rv[i][j].z; v.w = mish(v.w) + rv[i][j].w;
            if (tok < nrows) *reinterpret_cast<float4*>(Y + (rbase + tok) * D + co) = v;
        }
    }
}

template <typename T, int CPG, int NS, bool SPLIT = false>
inline hipError_t launch_convpos_ns(hipStream_t s, const float* X, const T* Wp, int Kp, const float* bias,
                                    const float* res, float* Y, int Bp, int N, int D, const int* lens, int nbl,
                                    const int* row_start) {
    constexpr int smem = NS * CPG * GEMM_ROW_BYTES + (128 + 32) * (CPG * (int)sizeof(T) + 16);
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&convpos_kernel<T, CPG, NS, SPLIT>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, smem);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    dim3 grid((N + 127) / 128, D / CPG, Bp);
    hipLaunchKernelGGL((convpos_kernel<T, CPG, NS, SPLIT>), grid, dim3(256), smem, s, X, Wp, Kp, bias, res, Y, N, D, lens, nbl, row_start);
    return hipGetLastError();
}

template <typename T, int CPG>
inline hipError_t launch_convpos_cpg(hipStream_t s, const float* X, const T* Wp, int Kp, const float* bias,
                                     const float* res, float* Y, int Bp, int N, int D, const int* lens, int nbl,
                                     const int* row_start, bool split = false) {
    constexpr int KT = GEMM_ROW_BYTES / (int)sizeof(T);
    if (Kp % KT != 0) return hipErrorInvalidValue;   // weights must be padded to whole K-tiles
    const long blocks = (long)((N + 127) / 128) * (D / CPG) * Bp;
    if constexpr (std::is_same_v<T, float> && CPG % 32 == 0) {
        if (split) {
            if (blocks > 384) return launch_convpos_ns<T, CPG, 4, true>(s, X, Wp, Kp, bias, res, Y, Bp, N, D, lens, nbl, row_start);
            return launch_convpos_ns<T, CPG, 8, true>(s, X, Wp, Kp, bias, res, Y, Bp, N, D, lens, nbl, row_start);
        }
    }
    if (split) return hipErrorInvalidValue;          // (split weights need the split kernel: float, 32 | channels per group)
    if (blocks > 384) return launch_convpos_ns<T, CPG, 4>(s, X, Wp, Kp, bias, res, Y, Bp, N, D, lens, nbl, row_start);
    return launch_convpos_ns<T, CPG, 8>(s, X, Wp, Kp, bias, res, Y, Bp, N, D, lens, nbl, row_start);
}

// D/16 channels per group must be 16, 32, 48 or 64 (dim 256 / 512 / 768 / 1024).  split: the weights are pre-split
// (convpos_can_split(D) only) and the products run on the f16 pipe.
inline bool convpos_can_split(int D) { return (D / 16) % 32 == 0; }
template <typename T>
inline hipError_t launch_convpos(hipStream_t s, const float* X, const T* Wp, int Kp, const float* bias, const float* res,
                                 float* Y, int Bp, int N, int D, const int* lens, int nbl, const int* row_start = nullptr,
                                 bool split = false) {
    switch (D / 16) {
        case 16: return launch_convpos_cpg<T, 16>(s, X, Wp, Kp, bias, res, Y, Bp, N, D, lens, nbl, row_start, split);
        case 32: return launch_convpos_cpg<T, 32>(s, X, Wp, Kp, bias, res, Y, Bp, N, D, lens, nbl, row_start, split);
        case 48: return launch_convpos_cpg<T, 48>(s, X, Wp, Kp, bias, res, Y, Bp, N, D, lens, nbl, row_start, split);
        case 64: return launch_convpos_cpg<T, 64>(s, X, Wp, Kp, bias, res, Y, Bp, N, D, lens, nbl, row_start, split);
        default: return hipErrorInvalidValue;
    }
}

}  // namespace f5
