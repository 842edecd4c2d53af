// launch_gemm(): picks the kernel and tile for C[M,N] = A[M,K] W[N,K]^T + epilogue.
//   v2 (gemm2.h, LDS-DMA ring) whenever K is a whole number of 128-byte K-tiles (the engine pads its operands so
//   that this always holds on the hot path); v1 (gemm.h, register-staged, any K % 16 bytes == 0) otherwise.
// Tile choice (measured on MI355X at M = 2048, tools/gemm2_sweep.py): the B=1 shapes are latency / L2-bandwidth
// bound, so the tile is the largest one that still yields >= ~1 workgroup per CU.
#pragma once
#include "gemm2.h"

namespace f5 {

enum GemmCfg { G2_128x128_8W = 2, G2_128x64_8W = 9, G2_64x64_4W = 8 };

inline int pick_cfg_v2(int M, int N) {
    auto tiles = [&](int bm, int bn) { return (long)((M + bm - 1) / bm) * ((N + bn - 1) / bn); };
    if (M <= 64) return G2_64x64_4W;  // skinny (time MLP, AdaLN stack over the NFE steps): weight-streaming, no row reuse to gain
    if (tiles(128, 128) >= 240) return G2_128x128_8W;
    if (tiles(128, 64) >= 200) return G2_128x64_8W;
    return G2_64x64_4W;
}

template <typename T, typename Epi>
inline hipError_t launch_gemm_v2(hipStream_t s, const T* A, int lda, const T* W, int ldw, int M, int N, int K,
                                 const Epi& epi, int cfg) {
    switch (cfg) {
        case G2_128x128_8W: return launch_gemm2_cfg<T, 128, 128, 2, 4, 4, Epi>(s, A, lda, W, ldw, M, N, K, epi);
        case G2_128x64_8W: return launch_gemm2_cfg<T, 128, 64, 4, 2, 4, Epi>(s, A, lda, W, ldw, M, N, K, epi);
        default: return launch_gemm2_cfg<T, 64, 64, 2, 2, 3, Epi>(s, A, lda, W, ldw, M, N, K, epi);
    }
}

template <typename T, typename Epi>
inline hipError_t launch_gemm(hipStream_t s, const T* A, int lda, const T* W, int ldw, int M, int N, int K,
                              const Epi& epi, int force_cfg = -1) {
    if (M <= 0 || N <= 0) return hipSuccess;
    constexpr int KT = GEMM_ROW_BYTES / (int)sizeof(T);
    if (K % KT == 0 && force_cfg != -2) return launch_gemm_v2<T, Epi>(s, A, lda, W, ldw, M, N, K, epi, force_cfg >= 0 ? force_cfg : pick_cfg_v2(M, N));
    return launch_gemm_v1<T, Epi>(s, A, lda, W, ldw, M, N, K, epi);
}

}  // namespace f5
