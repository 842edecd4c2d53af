// Ping-pong MFMA "TN" GEMM for many rows (v3): 256 x 256 block tile, 8 waves, same contract and epilogues as gemm.h / gemm2.h.
//
// gemm2.h's loop runs every wave through the same program with one barrier per K-tile: the two waves that share a SIMD
// issue their LDS fragment reads together and their MFMAs together, so the matrix pipe idles while both read (measured
// there: 0.51 us per 128x128x64 step against 0.216 us of bare MFMA issue).  Here the K-tile is cut into four PHASES of
// 16 MFMAs per wave (one 64 x 32 quadrant of the wave's 128 x 64 output, both 32-deep halves of the K-tile), each phase
// is a LOAD part (LDS fragment reads for that quadrant + the LDS-DMA requests of one quarter of the NEXT K-tile) and a
// COMPUTE part (the 16 MFMAs) separated by raw s_barriers, and waves 4-7 (the SIMD partners of waves 0-3) run ONE
// barrier behind waves 0-3: in every slot between two barriers one wave of each SIMD computes while its partner loads.
//
//   wave = (wr, wc), wr = wave >> 2 (the group: 0 leads, 1 follows), wc = wave & 3; wave tile = rows wr*128 .. +127,
//   columns wc*64 .. +63 = 8 x 4 MFMA tiles; quadrant walk Q00 -> Q01 -> Q11 -> Q10 so that each phase needs at most one
//   new operand half:   phase 1: A rows 0-63 (8 reads) + B cols 0-31 (4 reads)   phase 2: B cols 32-63 (4 reads)
//                       phase 3: A rows 64-127 (8 reads, same registers)         phase 4: nothing (B cols 0-31 are kept)
//   LDS: 2 buffers x (256 A rows + 256 B rows) x 128 B = 128 KiB, image and XOR swizzle of gemm2.h.  The next K-tile is
//   requested in four UNITS of 128 rows, one per phase, in the order its phases need them:
//        U_A1 = A rows {0-63, 128-191}        (phase 1 of both wave rows)     requested in phase 1 of the tile before
//        U_B1 = B cols {wc*64 + 0..31}        (phase 1)                        ... phase 2
//        U_B2 = B cols {wc*64 + 32..63}       (phase 2)                        ... phase 3
//        U_A2 = A rows {64-127, 192-255}      (phase 3)                        ... phase 4
//   Every wave requests two 1-KiB pieces of each unit.  A unit is read one slot after a barrier that every wave passed
//   behind a counted `s_waitcnt vmcnt(4)` (everything but the two youngest units has landed): the leading group waits
//   at the end of its COMPUTE parts 4, 1, 2, the following group at the end of its LOAD parts 4, 1, 2 -- the same
//   barriers -- so each unit has 5-7 slots (~1,300+ cycles) to arrive.  The buffer a tile is requested into was last read
//   two slots or more before the first request (write-after-read), and nothing is requested past the last K-tile.
// Requirements as gemm2.h: K % (128 / sizeof(T)) == 0, lda / ldw multiples of 16 bytes.
#pragma once
#include "gemm2.h"

namespace f5 {

// does the epilogue offer the staged (row-major through LDS) store interface of gemm.h, and in which form
template <typename E, typename = void> struct epi_staged : std::false_type {};
template <typename E> struct epi_staged<E, std::void_t<decltype(E::kStaged)>> : std::integral_constant<bool, E::kStaged> {};

template <int OFF> __device__ __forceinline__ void lds_read_b128_off(u32x4& dst, unsigned addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF));
}
__device__ __forceinline__ void reg_fence(u32x4& a) { asm volatile("" : "+v"(a)); }

// DIAG (tools only; 0 in the product): 4 = no epilogue (the K loop + launch alone: what the epilogue of a shape costs by omission)
// DIAG bit 8 (product, T = float): both operands PRE-SPLIT into f16 hi / lo planes (gemm2.h MODE 5, F5_PREC_F16X3: a 128-byte row is 32
// elements, chunks 0-3 hi, 4-7 lo, so the two fragment reads of a row ARE hi and lo); a phase is then 24 f16 MFMAs -- per accumulator
// and K-tile lo x hi, hi x lo, hi x hi, the order of gemm2's MODE 5 (bit-identical results).
template <typename T, typename Epi, int DIAG = 0>
__global__ __launch_bounds__(512) void gemm_pp_kernel(const T* __restrict__ A, int lda, const T* __restrict__ W, int ldw, int M,
                                                      int N, int K, Epi epi, int xa, int xb, const int* __restrict__ m_limit) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int BM = 256, BN = 256;
    constexpr int KT = GEMM_ROW_BYTES / sizeof(T);
    constexpr int EPC = 16 / sizeof(T);
    constexpr int MI = 8, NJ = 4;
    constexpr int RB = GEMM_ROW_BYTES;
    constexpr int BUF = (BM + BN) * RB;                 // 64 KiB per K-tile

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;            // wr is also the group (0 leads, 1 follows by one barrier)
    int tile_m = blockIdx.y, tile_n = blockIdx.x;
    if (m_limit) {                                      // device-side row count: row tiles dealt over the XCDs (gemm2.h)
        const int tiles_n = gridDim.x;
        const int bid = blockIdx.y * tiles_n + blockIdx.x;
        const int idx = bid >> 3;
        tile_m = (idx / tiles_n) * 8 + (bid & 7);
        tile_n = idx % tiles_n;
    } else if (xa > 0) {                                // XCD-aware tile order (gemm2.h)
        const int tiles_n = gridDim.x;
        const int bid = blockIdx.y * tiles_n + blockIdx.x;
        const int xcd = bid & 7, idx = bid >> 3;
        const int rects_n = tiles_n / xb;
        const int per_rect = xa * xb;
        const int rect = xcd + 8 * (idx / per_rect), in = idx % per_rect;
        tile_m = (rect / rects_n) * xa + in / xb;
        tile_n = (rect % rects_n) * xb + in % xb;
    }
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    if (m_limit) {                                      // device-side row count (gemm2.h): tiles past it retire at once
        M = min(M, __builtin_amdgcn_readfirstlane(*m_limit));
        if (m0 >= M) return;
    }
    const int nkt = K / KT;
    const bool transposed = Epi::kTransposes && epi.tile_transposed(n0);   // block tiles never straddle the boundary (host check)

    // ---- LDS-DMA sources: two pieces (8 rows x 128 B) of each unit per wave; lane -> row lr of the piece, swizzled chunk
    const int lr = lane >> 3, lc = (lane & 7) ^ lr;
    const T* asrc[2][2];     // [unit A1 / A2][piece]
    const T* bsrc[2][2];     // [unit B1 / B2][piece]
    int adst[2][2], bdst[2][2];
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int ar = h * 128 + u * 64 + wave * 8;                       // A row of the piece inside the block tile
            asrc[u][h] = A + (size_t)min(m0 + ar + lr, M - 1) * lda + lc * EPC;
            adst[u][h] = ar * RB;
            const int q = wave + 8 * h;                                       // B piece index inside the unit (0..15)
            const int br = (q >> 2) * 64 + u * 32 + (q & 3) * 8;
            bsrc[u][h] = W + (size_t)min(n0 + br + lr, N - 1) * ldw + lc * EPC;
            bdst[u][h] = (BM + br) * RB;
        }
    auto issue_a = [&](int u, int koff, char* base) {
        glds16(asrc[u][0] + koff, base + adst[u][0]);
        glds16(asrc[u][1] + koff, base + adst[u][1]);
    };
    auto issue_b = [&](int u, int koff, char* base) {
        glds16(bsrc[u][0] + koff, base + bdst[u][0]);
        glds16(bsrc[u][1] + koff, base + bdst[u][1]);
    };

    f32x4 acc[MI][NJ];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // ---- fragment addressing (gemm2.h): row R, 16-byte chunk c = kk*4 + g  ->  R*128 + ((c ^ (R & 7)) * 16)
    const int l15 = lane & 15, g = lane >> 4, sw = l15 & 7;
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
    const unsigned a_row = (wr * 128 + l15) * RB, b_row = (BM + wc * 64 + l15) * RB;
    const unsigned c0 = ((0 + g) ^ sw) * 16, c1 = ((4 + g) ^ sw) * 16;

    // ---- prologue: the whole first K-tile, then the stagger
    issue_a(0, 0, smem);
    issue_b(0, 0, smem);
    issue_b(1, 0, smem);
    issue_a(1, 0, smem);
    __builtin_amdgcn_s_waitcnt(0xC07F);   // lgkmcnt(0): only the loop's own LDS reads on that counter from here on
    wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    if (wr == 1) __builtin_amdgcn_s_barrier();          // the following group starts one slot late

    u32x4 af[4][2], b01[2][2], b23[2][2];               // A rows (one half of the wave tile at a time), B cols 0-31 / 32-63
    auto kloop = [&](auto trc) {
        constexpr bool TR = decltype(trc)::value;
        auto mma_quadrant = [&](auto i0c, u32x4 (&bf)[2][2], auto j0c) {   // (compile-time tile origin: acc stays in registers)
            constexpr int i0 = decltype(i0c)::value, j0 = decltype(j0c)::value;
            __builtin_amdgcn_s_setprio(1);
            if constexpr ((DIAG & 8) != 0) {
                static_assert(std::is_same_v<T, float>, "pre-split operands are f32 rows of f16 hi / lo planes");
#pragma unroll
                for (int term = 0; term < 3; ++term)      // term-major, as gemm2.h MODE 5: (w lo, a hi), (w hi, a lo), (w hi, a hi)
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int j = 0; j < 2; ++j) {
                            const u32x4& wv = bf[j][term == 0 ? 1 : 0];
                            const u32x4& av = af[i][term == 1 ? 1 : 0];
                            if (!TR) acc[i0 + i][j0 + j] = Mma<f16_t>::run(wv, av, acc[i0 + i][j0 + j]);
                            else acc[i0 + i][j0 + j] = Mma<f16_t>::run(av, wv, acc[i0 + i][j0 + j]);
                        }
            } else {
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        if (!TR) acc[i0 + i][j0 + j] = Mma<T>::run(bf[j][kk], af[i][kk], acc[i0 + i][j0 + j]);
                        else acc[i0 + i][j0 + j] = Mma<T>::run(af[i][kk], bf[j][kk], acc[i0 + i][j0 + j]);
                    }
            }
            __builtin_amdgcn_s_setprio(0);
        };
        auto read_a = [&](unsigned ab0, unsigned ab1, auto half) {   // half = 0: rows 0-63 of the wave tile, 1: rows 64-127
            constexpr int H = decltype(half)::value;
            lds_read_b128_off<(H * 4 + 0) * 16 * RB>(af[0][0], ab0); lds_read_b128_off<(H * 4 + 1) * 16 * RB>(af[1][0], ab0);
            lds_read_b128_off<(H * 4 + 2) * 16 * RB>(af[2][0], ab0); lds_read_b128_off<(H * 4 + 3) * 16 * RB>(af[3][0], ab0);
            lds_read_b128_off<(H * 4 + 0) * 16 * RB>(af[0][1], ab1); lds_read_b128_off<(H * 4 + 1) * 16 * RB>(af[1][1], ab1);
            lds_read_b128_off<(H * 4 + 2) * 16 * RB>(af[2][1], ab1); lds_read_b128_off<(H * 4 + 3) * 16 * RB>(af[3][1], ab1);
        };
        auto read_b = [&](u32x4 (&bf)[2][2], unsigned bb0, unsigned bb1, auto half) {
            constexpr int H = decltype(half)::value;
            lds_read_b128_off<(H * 2 + 0) * 16 * RB>(bf[0][0], bb0); lds_read_b128_off<(H * 2 + 1) * 16 * RB>(bf[1][0], bb0);
            lds_read_b128_off<(H * 2 + 0) * 16 * RB>(bf[0][1], bb1); lds_read_b128_off<(H * 2 + 1) * 16 * RB>(bf[1][1], bb1);
        };
        auto fence_a = [&]() {
#pragma unroll
            for (int i = 0; i < 4; ++i) { reg_fence(af[i][0]); reg_fence(af[i][1]); }
        };
        auto fence_b = [&](u32x4 (&bf)[2][2]) { reg_fence(bf[0][0]); reg_fence(bf[0][1]); reg_fence(bf[1][0]); reg_fence(bf[1][1]); };
        auto lgkm0 = [&]() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); };
        auto bar = [&]() {
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
        };
        // counted wait of the group whose turn it is: `last` = no K-tile is being requested any more
        auto unit_wait = [&](bool mine, bool more, auto allow_last) {
            if (mine) {
                if (more) wait_vmcnt<4>();
                else wait_vmcnt<decltype(allow_last)::value>();
            }
        };
        using I0 = std::integral_constant<int, 0>;
        using I1 = std::integral_constant<int, 1>;
        using I2 = std::integral_constant<int, 2>;
        using I4 = std::integral_constant<int, 4>;
        const bool lead = wr == 0;
        for (int kt = 0; kt < nkt; ++kt) {
            const unsigned sb = lds_base + (unsigned)((kt & 1) * BUF);
            char* nbase = smem + ((kt + 1) & 1) * BUF;
            const bool more = kt + 1 < nkt;
            const int koff = (kt + 1) * KT;
            const unsigned ab0 = sb + a_row + c0, ab1 = sb + a_row + c1, bb0 = sb + b_row + c0, bb1 = sb + b_row + c1;
            // ---------------- phase 1: Q00
            read_a(ab0, ab1, I0{});
            read_b(b01, bb0, bb1, I0{});
            if (more) issue_a(0, koff, nbase);
            unit_wait(!lead, more, I2{});              // (follower, LOAD part 1: U_B2 of this tile must have landed)
            bar();
            lgkm0(); fence_a(); fence_b(b01);
            __builtin_amdgcn_sched_barrier(0);
            mma_quadrant(I0{}, b01, I0{});
            unit_wait(lead, more, I2{});
            bar();
            // ---------------- phase 2: Q01
            read_b(b23, bb0, bb1, I1{});
            if (more) issue_b(0, koff, nbase);
            unit_wait(!lead, more, I0{});              // (U_A2 of this tile)
            bar();
            lgkm0(); fence_b(b23);
            __builtin_amdgcn_sched_barrier(0);
            mma_quadrant(I0{}, b23, I2{});
            unit_wait(lead, more, I0{});
            bar();
            // ---------------- phase 3: Q11
            read_a(ab0, ab1, I1{});
            if (more) issue_b(1, koff, nbase);
            bar();
            lgkm0(); fence_a();
            __builtin_amdgcn_sched_barrier(0);
            mma_quadrant(I4{}, b23, I2{});
            bar();
            // ---------------- phase 4: Q10
            if (more) issue_a(1, koff, nbase);
            unit_wait(!lead, more, I4{});              // (U_A1, U_B1 of the next tile; nothing pending after the last)
            bar();
            fence_b(b01);
            mma_quadrant(I4{}, b01, I0{});
            unit_wait(lead, more, I4{});
            bar();
        }
    };
    if constexpr (Epi::kTransposes) {
        if (transposed) kloop(std::true_type{});
        else kloop(std::false_type{});
    } else {
        kloop(std::false_type{});
    }
    if (wr == 0) __builtin_amdgcn_s_barrier();          // balances the follower's extra barrier
    if constexpr ((DIAG & 7) == 4) {
        float sink = 0.f;
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j) sink += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
        if (sink == 1234.5678f && lda < 0) *reinterpret_cast<volatile float*>(smem) = sink;
        return;
    }

    // ---- epilogue (contexts are set up after the loop: 128 accumulator + 64 fragment registers were live in it)
    // Interior block tiles (a block-uniform test) run without per-lane bounds checks, so that the compiler streams the operand
    // loads and the stores of the 32 sub-tiles with counted waits.  Edge tiles load through clamped indices, retire those
    // loads with one visible wait per column group and then store under the bounds checks: a store inside an exec-masked
    // block must not be the first use of a pending load, or hipcc waits vmcnt(0) -- i.e. for the previous store -- in
    // front of every store (gemm2.h).
    const int mw = m0 + wr * 128, nw = n0 + wc * 64;
    const bool interior = m0 + BM <= M && n0 + BN <= N;
    // Staged stores (gemm.h): the ring is free now (every wave is past its last fragment read and no LDS-DMA is in flight), so each
    // wave turns its 128 x 64 tile through its own 16 KiB slice -- fragment order in, rows out.  No barrier: a wave only reads what
    // it wrote itself.
    if constexpr (epi_staged<Epi>::value) {
        if (interior && epi.staged_ok(transposed)) {
            epi.template staged<MI, NJ>(smem + wave * (16 * 1024), lane, mw, nw, acc, transposed);
            return;
        }
    }
    if (!transposed) {
        if (interior) {
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const typename Epi::ColCtx cc = epi.col(nw + j * 16 + g * 4);
#pragma unroll
                for (int i = 0; i < MI; ++i) {
                    const typename Epi::RowCtx rc = epi.row(mw + i * 16 + l15);
                    epi.store(rc, cc, acc[i][j], epi.preload(rc, cc));
                }
            }
        } else {
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const int n = nw + j * 16 + g * 4;
                const typename Epi::ColCtx cc = epi.col(min(n, N - 4));
                typename Epi::RowCtx rc[MI];
                typename Epi::Pre pre[MI];
#pragma unroll
                for (int i = 0; i < MI; ++i) {
                    rc[i] = epi.row(min(mw + i * 16 + l15, M - 1));
                    pre[i] = epi.preload(rc[i], cc);
                }
                __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): the guarded stores below wait for nothing
#pragma unroll
                for (int i = 0; i < MI; ++i)
                    if (n < N && mw + i * 16 + l15 < M) epi.store(rc[i], cc, acc[i][j], pre[i]);
            }
        }
    } else {
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int n = nw + j * 16 + l15;
            const typename Epi::TColCtx tc = epi.tcol(min(n, N - 1));
            __builtin_amdgcn_s_waitcnt(0x0F70);
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                const int m = mw + i * 16 + g * 4;
                if (n < N && m < M) epi.tstore(epi.trow(m, M), tc, acc[i][j]);
            }
        }
    }
}

// can this epilogue's transposed region only start at a multiple of the 256-column block tile?
template <typename Epi> inline bool gemm3_epilogue_ok(const Epi&) { return !Epi::kTransposes; }
template <typename TO> inline bool gemm3_epilogue_ok(const EpiQKV<TO>& e) { return (2 * e.H * 64) % 256 == 0; }

template <typename T, typename Epi, int DIAG = 0>
inline hipError_t launch_gemm3_raw(hipStream_t s, const T* A, int lda, const T* W, int ldw, int M, int N, int K, const Epi& epi,
                                   const int* m_limit) {
    constexpr int smem = 2 * 512 * GEMM_ROW_BYTES;   // 128 KiB
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_pp_kernel<T, Epi, DIAG>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, smem);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    dim3 grid((N + 255) / 256, (M + 255) / 256);
    if (m_limit) grid.y = (grid.y + 7) / 8 * 8;   // (the device-side tile order deals row tiles in groups of 8)
    int xa = 0, xb = 0;
    if (!m_limit) pick_xcd_rect((int)grid.y, (int)grid.x, &xa, &xb);
    hipLaunchKernelGGL((gemm_pp_kernel<T, Epi, DIAG>), grid, dim3(512), smem, s, A, lda, W, ldw, M, N, K, epi, xa, xb, m_limit);
    return hipGetLastError();
}

template <typename T, typename Epi, int DIAG = 0>
inline hipError_t launch_gemm3(hipStream_t s, const T* A, int lda, const T* W, int ldw, int M, int N, int K, const Epi& epi,
                               const int* m_limit = nullptr) {
    return with_static_act(epi, [&](const auto& e) {
        return launch_gemm3_raw<T, std::decay_t<decltype(e)>, DIAG>(s, A, lda, W, ldw, M, N, K, e, m_limit);
    });
}

}  // namespace f5
