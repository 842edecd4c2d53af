// launch_gemm(): picks the kernel and tile for C[M,N] = A[M,K] W[N,K]^T + epilogue.
//   v2 (gemm2.h, LDS-DMA ring) whenever K is a whole number of 128-byte K-tiles (the engine pads its operands so
//   that this always holds on the hot path); v3 (gemm3.h, 256x256 ping-pong) for many-row problems that fill the chip
//   with such tiles (the utterance batches of C3 / C4); v1 (gemm.h, register-staged, any K % 16 bytes == 0) otherwise.
// Tile choice (measured on MI355X at M = 2048, tools/gemm2_sweep.py): the B=1 shapes are latency / L2-bandwidth
// bound, so the tile is the largest one that still yields >= ~1 workgroup per CU.
#pragma once
#include "gemm3.h"

namespace f5 {

enum GemmCfg { G2_128x128_8W = 2, G2_128x64_8W = 9, G2_64x64_4W = 8, G2_128x192_8W = 10, G2_256x128_8W = 13, G3_256x256_PP = 20 };
// (Tried for the many-row block GEMMs and dropped, tools/block_gemm_time.py: 128x128 tiles with a 2-stage ring = 64 KB of LDS, two
//  workgroups per CU so that one's epilogue runs under the other's K loop -- 8 waves: 939 us per block of 32,768 rows against 794
//  for the tiles chosen below, QKV 420 against 284 us; 4 waves: 1,086 us.)

// cost = rounds of workgroups over the 256 CUs x the time of one tile of that shape (us at K = 1024, measured with
// tools/gemm2_sweep.py on a full chip: the per-K-step time grows much more slowly than the tile area, so the largest
// tile that does not add a round wins; e.g. the QKV projection 2048 x 3072: 128x128 = 384 tiles = 2 rounds (25 us),
// 128x192 = 256 tiles = 1 round (20 us)).
// g3_penalty: extra us per round for the ping-pong tile when the epilogue is memory-heavy (see launch_gemm)
inline int pick_cfg_v2(int M, int N, bool allow_v3 = false, float g3_penalty = 0.f) {
    static const int forced = getenv("F5_GEMM_CFG") ? atoi(getenv("F5_GEMM_CFG")) : -1;   // diagnostic: one tile for every GEMM
    static const int forced_n = getenv("F5_GEMM_CFG_N") ? atoi(getenv("F5_GEMM_CFG_N")) : 0;   // ... only for this N
    if (forced >= 0 && (forced != G3_256x256_PP || allow_v3) && (forced_n == 0 || forced_n == N)) return forced;
    if (M <= 64) return G2_64x64_4W;  // skinny (time MLP, AdaLN stack over the NFE steps): weight-streaming, no row reuse to gain
    struct Cand { int id, bm, bn; float t; };
    // (256x128: 865 TFLOP/s at M = 16384, N = 2048 against 722 for 128x128: the many-utterance batches C3 / C4)
    // (256x256 ping-pong, gemm3.h: 16384 x {1024, 2048, 3072} x 1024 in 39 / 74 / 110 us = 37 us per round of 256 tiles against
    //  42 us for two rounds of 256x128 tiles; 1.22 PFLOP/s in the K loop against 1.02)
    static const Cand cands[] = {{G3_256x256_PP, 256, 256, 37.0f}, {G2_256x128_8W, 256, 128, 21.0f}, {G2_128x192_8W, 128, 192, 19.3f},
                                 {G2_128x128_8W, 128, 128, 14.5f}, {G2_128x64_8W, 128, 64, 7.3f}, {G2_64x64_4W, 64, 64, 4.0f}};
    int best = G2_64x64_4W;
    float best_cost = 3.0e38f;
    for (const Cand& c : cands) {
        if (c.id == G3_256x256_PP && !allow_v3) continue;
        const long tiles = (long)((M + c.bm - 1) / c.bm) * ((N + c.bn - 1) / c.bn);
        const float cost = (float)((tiles + 255) / 256) * (c.t + (c.id == G3_256x256_PP ? g3_penalty : 0.f));
        if (cost < best_cost) { best_cost = cost; best = c.id; }  // ties keep the larger tile (listed first)
    }
    return best;
}

template <typename Epi> inline bool epilogue_streams_residual(const Epi&) { return false; }
inline bool epilogue_streams_residual(const EpiGateRes&) { return true; }

template <typename T, typename Epi>
inline hipError_t launch_gemm_v2(hipStream_t s, const T* A, int lda, const T* W, int ldw, int M, int N, int K,
                                 const Epi& epi, int cfg, const int* ml = nullptr, const GemmConv& cv = GemmConv{}, int split = 0) {
    if constexpr (std::is_same_v<T, float>) {
        if (split == 2) {   // both operands pre-split (gemm2.h MODE 5; the ping-pong kernel's DIAG bit 8)
            switch (cfg) {
                case G3_256x256_PP:
                    if (cv.tpt == 0 && cv.m_base == 0) return launch_gemm3<T, Epi, 8>(s, A, lda, W, ldw, M, N, K, epi, ml);
                    [[fallthrough]];
                case G2_256x128_8W: return launch_gemm2_cfg<T, 256, 128, 4, 2, 3, Epi, 5>(s, A, lda, W, ldw, M, N, K, epi, ml, cv);
                case G2_128x192_8W: return launch_gemm2_cfg<T, 128, 192, 2, 4, 3, Epi, 5>(s, A, lda, W, ldw, M, N, K, epi, ml, cv);
                case G2_128x128_8W: return launch_gemm2_cfg<T, 128, 128, 2, 4, 4, Epi, 5>(s, A, lda, W, ldw, M, N, K, epi, ml, cv);
                case G2_128x64_8W: return launch_gemm2_cfg<T, 128, 64, 4, 2, 4, Epi, 5>(s, A, lda, W, ldw, M, N, K, epi, ml, cv);
                default: return launch_gemm2_cfg<T, 64, 64, 2, 2, 3, Epi, 5>(s, A, lda, W, ldw, M, N, K, epi, ml, cv);
            }
        }
        if (split) {   // W in the split_planar layout, A f32 split in registers, products on the f16 pipe (gemm2.h MODE 3)
            switch (cfg) {
                case G3_256x256_PP:
                case G2_256x128_8W: return launch_gemm2_cfg<T, 256, 128, 4, 2, 3, Epi, 3>(s, A, lda, W, ldw, M, N, K, epi, ml, cv);
                case G2_128x192_8W: return launch_gemm2_cfg<T, 128, 192, 2, 4, 3, Epi, 3>(s, A, lda, W, ldw, M, N, K, epi, ml, cv);
                case G2_128x128_8W: return launch_gemm2_cfg<T, 128, 128, 2, 4, 4, Epi, 3>(s, A, lda, W, ldw, M, N, K, epi, ml, cv);
                case G2_128x64_8W: return launch_gemm2_cfg<T, 128, 64, 4, 2, 4, Epi, 3>(s, A, lda, W, ldw, M, N, K, epi, ml, cv);
                default: return launch_gemm2_cfg<T, 64, 64, 2, 2, 3, Epi, 3>(s, A, lda, W, ldw, M, N, K, epi, ml, cv);
            }
        }
    } else if (split) return hipErrorInvalidValue;
    switch (cfg) {
        case G3_256x256_PP:
            if (cv.tpt == 0 && cv.m_base == 0) return launch_gemm3<T, Epi>(s, A, lda, W, ldw, M, N, K, epi, ml);
            [[fallthrough]];   // (the ping-pong kernel has no implicit-conv / row-offset mode)
        case G2_256x128_8W: return launch_gemm2_cfg<T, 256, 128, 4, 2, 3, Epi>(s, A, lda, W, ldw, M, N, K, epi, ml, cv);
        case G2_128x192_8W: return launch_gemm2_cfg<T, 128, 192, 2, 4, 3, Epi>(s, A, lda, W, ldw, M, N, K, epi, ml, cv);
        case G2_128x128_8W: return launch_gemm2_cfg<T, 128, 128, 2, 4, 4, Epi>(s, A, lda, W, ldw, M, N, K, epi, ml, cv);
        case G2_128x64_8W: return launch_gemm2_cfg<T, 128, 64, 4, 2, 4, Epi>(s, A, lda, W, ldw, M, N, K, epi, ml, cv);
        default: return launch_gemm2_cfg<T, 64, 64, 2, 2, 3, Epi>(s, A, lda, W, ldw, M, N, K, epi, ml, cv);
    }
}

// m_limit (device int, may be null): rows actually present, <= M (the v2 / v3 kernels only: the engine's operands always qualify)
template <typename T, typename Epi>
inline hipError_t launch_gemm(hipStream_t s, const T* A, int lda, const T* W, int ldw, int M, int N, int K,
                              const Epi& epi, int force_cfg = -1, const int* m_limit = nullptr, int m_hint = 0,
                              const GemmConv& cv = GemmConv{}, int split = 0) {   // split: 0 plain, 1 W pre-split, 2 A and W pre-split
    if (M <= 0 || N <= 0) return hipSuccess;
    if (split && (K % (GEMM_ROW_BYTES / (int)sizeof(T)) != 0 || force_cfg == -2)) return hipErrorInvalidValue;   // split operands: v2 kernels only
    constexpr int KT = GEMM_ROW_BYTES / (int)sizeof(T);
    if (cv.tpt > 0 && (K % KT != 0 || force_cfg == -2)) return hipErrorInvalidValue;   // implicit conv: v2 kernels only
    // (m_hint: the row count the caller expects behind m_limit -- the tile is chosen for it, the grid covers M)
    // A many-row problem whose row count is a few rows past a multiple of 256 (UNetT: 16 x 1025 = 16,400 rows) would pay
    // a whole extra round of 256-row tiles for the last 16 rows: the 256-row multiple goes to the ping-pong kernel and the
    // remainder to one row of 64x64 tiles (same K order per element: bit-identical to a single launch).
    // (Tried for the remainder launch and dropped: running it BESIDE the main launch on a second stream -- fork / join events, in the
    // captured graph a two-node parallel branch per GEMM -- C5 350.5 -> 356.0 ms: the cross-stream dependencies of 1,920 branches cost
    // more than the ~20 ms of remainder launches they hide.  Deeper LDS-DMA rings for its 64x64 tiles (6, 8 stages): 6.3 -> 6.5-6.8 us.)
    // (Round 2 sent the residual epilogue -- EpiGateRes reads and writes the f32 stream, 268 MB per launch at 32,768 rows -- to two
    // rounds of 256x128 tiles because the ping-pong kernel's fragment-order epilogue overlapped with nothing.  With the staged
    // row-major epilogue (gemm.h) the ping-pong tile wins there too: tools/block_gemm_time.py, 32,768 rows, out-proj / FF2.)
    const float g3_pen = 0.0f;
    const bool v3_ok = ((sizeof(T) == 2 && !split) || split == 2) && gemm3_epilogue_ok(epi);   // 16-bit operands, or f32 rows of pre-split planes
    if (K % KT == 0 && force_cfg == -1 && !m_limit && cv.tpt == 0 && v3_ok) {
        const int rem = M % 256, main = M - rem;
        const int cfg_main = (rem > 0 && rem <= 64 && main >= 4096) ? pick_cfg_v2(main, N, true, g3_pen) : -1;
        if (cfg_main == G3_256x256_PP || cfg_main == G2_256x128_8W) {
            hipError_t e = launch_gemm_v2<T, Epi>(s, A, lda, W, ldw, main, N, K, epi, cfg_main, nullptr, GemmConv{}, split);
            if (e != hipSuccess) return e;
            GemmConv tail{};
            tail.m_base = main;
            return launch_gemm_v2<T, Epi>(s, A + (size_t)main * lda, lda, W, ldw, rem, N, K, epi, G2_64x64_4W, nullptr, tail, split);
        }
    }
    if (K % KT == 0 && force_cfg != -2) {
        int cfg = force_cfg >= 0 ? force_cfg : pick_cfg_v2(m_hint > 0 ? m_hint : M, N, v3_ok && cv.tpt == 0, g3_pen);
        // pre-split operands (F5_PREC_F16X3 block GEMMs): three MFMAs per fragment pair shift the balance towards the small tile
        // where both fit in two rounds (2048 x 1024 x {1024, 2048}: 64x64 16.2 / 28.7 us, 128x64 18.0 / 30.8 -- tools/probe/gemm_split_probe.hip)
        if (split == 2 && force_cfg < 0 && cfg == G2_128x64_8W && (long)((M + 63) / 64) * ((N + 63) / 64) <= 512) cfg = G2_64x64_4W;
        return launch_gemm_v2<T, Epi>(s, A, lda, W, ldw, M, N, K, epi, cfg, m_limit, cv, split);
    }
    if (m_limit) return hipErrorInvalidValue;
    return launch_gemm_v1<T, Epi>(s, A, lda, W, ldw, M, N, K, epi);
}

}  // namespace f5
