// Pipelined MFMA "TN" GEMM (v2): same contract and epilogues as gemm.h, different data movement.
//
//   * operands go HBM/L2 -> LDS directly with global_load_lds (16 B per lane, 1 KiB per wave-instruction, no VGPR
//     staging, no ds_write); the LDS image of a tile is [rows][128 B] with NO padding (LDS-DMA writes lane-linear),
//     bank conflicts are removed by an XOR swizzle applied to the per-lane SOURCE address and to the fragment reads
//     (16-byte chunk c of row r lives in slot c ^ (r & 7));
//   * NS-stage ring (NS-1 K-tiles issued ahead), ONE raw s_barrier per K-tile, counted `s_waitcnt vmcnt(N)` that only
//     retires the tile about to be consumed -- the loads of the next NS-2 tiles stay in flight across the barrier;
//   * tail: nothing is requested past the last K-tile (a dummy load would cost one more L2 round trip before the
//     workgroup may retire); the vmcnt allowance shrinks with the tiles left (one scalar branch per K-tile);
//     rows beyond M / N are clamped (their results are never stored).
// Requirements: K % (128 / sizeof(T)) == 0 (the engine pads K), lda/ldw multiples of 16 bytes.
#pragma once
#include "gemm.h"


namespace f5 {

template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

__device__ __forceinline__ void glds16(const void* gsrc, char* lds_dst) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_dst, 16, 0, 0);
}

// MODE (diagnostic builds only): 0 = normal, 1 = LDS-DMA only (no fragment reads / MFMA), 2 = compute only (no DMA
// in the K loop): the two floors of the pipeline.
// Fragment reads issued as inline asm so that hipcc does not wait lgkmcnt(0) before the first MFMA (it does for every
// compiler-visible ds_read_b128 on this toolchain): the waits below are placed by hand with counted lgkmcnt(N); the
// fragments a wait retires are threaded through it as "+v" operands, so no MFMA can be scheduled above its wait.
__device__ __forceinline__ void lds_read_b128_asm(u32x4& dst, unsigned addr) {
    asm volatile("ds_read_b128 %0, %1" : "=v"(dst) : "v"(addr));
}
// (the lgkmcnt field is 4 bits: a larger allowance is clamped to 15, which only waits a little longer)
template <int N> __device__ __forceinline__ void wait_lgkm(u32x4& a) { asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(a) : "n"(N > 15 ? 15 : N)); }
template <int N> __device__ __forceinline__ void wait_lgkm(u32x4& a, u32x4& b) {
    asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(a), "+v"(b) : "n"(N > 15 ? 15 : N));
}
template <int N> __device__ __forceinline__ void wait_lgkm(u32x4& a, u32x4& b, u32x4& c) {
    asm volatile("s_waitcnt lgkmcnt(%3)" : "+v"(a), "+v"(b), "+v"(c) : "n"(N > 15 ? 15 : N));
}
template <int N> __device__ __forceinline__ void wait_lgkm(u32x4& a, u32x4& b, u32x4& c, u32x4& d) {
    asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "n"(N > 15 ? 15 : N));
}
template <int N> __device__ __forceinline__ void wait_lgkm(u32x4& a, u32x4& b, u32x4& c, u32x4& d, u32x4& e) {
    asm volatile("s_waitcnt lgkmcnt(%5)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e) : "n"(N > 15 ? 15 : N));
}

// wait for the fragments MFMA row i of half kk needs (issue order per kk: a[0], w[0..NJ-1], a[1..MI-1])
template <int R, int MI, int NJ, int KK, int I>
__device__ __forceinline__ void wait_row_ct(u32x4 (&af)[2][MI], u32x4 (&wf)[2][NJ]) {
    constexpr int idx = KK * (MI + NJ) + NJ + I;
    constexpr int allow = R - 1 - idx;
    if constexpr (I == 0) {
        if constexpr (NJ == 1) wait_lgkm<allow>(af[KK][0], wf[KK][0]);
        else if constexpr (NJ == 2) wait_lgkm<allow>(af[KK][0], wf[KK][0], wf[KK][1]);
        else if constexpr (NJ == 3) wait_lgkm<allow>(af[KK][0], wf[KK][0], wf[KK][1], wf[KK][2]);
        else wait_lgkm<allow>(af[KK][0], wf[KK][0], wf[KK][1], wf[KK][2], wf[KK][3]);
    } else {
        wait_lgkm<allow>(af[KK][I]);
    }
}
template <int R, int MI, int NJ>
__device__ __forceinline__ void wait_row(int kk, int i, u32x4 (&af)[2][MI], u32x4 (&wf)[2][NJ]) {
    // kk and i are compile-time constants after unrolling; the switch folds away
#define F5_WR(KK, I) if (kk == KK && i == I) { if constexpr (I < MI) wait_row_ct<R, MI, NJ, KK, I>(af, wf); }
    F5_WR(0, 0) F5_WR(0, 1) F5_WR(0, 2) F5_WR(0, 3) F5_WR(0, 4) F5_WR(0, 5) F5_WR(0, 6) F5_WR(0, 7)
    F5_WR(1, 0) F5_WR(1, 1) F5_WR(1, 2) F5_WR(1, 3) F5_WR(1, 4) F5_WR(1, 5) F5_WR(1, 6) F5_WR(1, 7)
#undef F5_WR
}

// ---- MODE 3: f32 operands, products on the f16 matrix pipe ("f16x3", F5_PREC_F16X3) -------------------------------------------
// a = a_hi + a_lo with a_hi = f16(a), a_lo = f16(a - a_hi): 22 significant bits (v_mfma_f32_16x16x32_f16 keeps f16 subnormal
// inputs -- tools/f16_denorm.py -- so a_lo is good down to 2^-25 absolute);  a w ~= a_hi w_hi + a_lo w_hi + a_hi w_lo, every
// product exact in the f32 accumulator, the dropped a_lo w_lo term ~2^-22 relative.  Three 16-cycle f16 MFMAs replace eight
// 32-cycle f32 ones per 32-deep K-tile.
// The A operand stays f32 in memory and in LDS (same LDS-DMA ring as the f32 kernel) and is split in registers after the
// fragment read; the W operand is split ONCE (split_planar_kernel, elementwise.h) into the same 128 bytes per 32 elements:
//   16-byte chunk g (g = 0..3) = f16 hi of k = 4g..4g+3, 16+4g..16+4g+3;   chunk 4 + g = the f16 lo of the same k
// which is exactly what lane group g of the f32 fragment reads (chunk g, chunk 4 + g) returns for A as 8 floats -- so both
// operands feed slot s of lane group g with the same k and the swizzled image, the DMA and the read addresses are unchanged.
// |a| must stay below 65504 (as in the f16 precision).
typedef __attribute__((ext_vector_type(2))) float f32x2_t;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2_t;
__device__ __forceinline__ void split8_f16(const u32x4& c0, const u32x4& c1, f16x8& hi, f16x8& lo) {
    const f32x4 x0 = __builtin_bit_cast(f32x4, c0), x1 = __builtin_bit_cast(f32x4, c1);
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const f32x2_t v = p < 2 ? f32x2_t{x0[2 * p], x0[2 * p + 1]} : f32x2_t{x1[2 * p - 4], x1[2 * p - 3]};
        const f16x2_t h = __builtin_convertvector(v, f16x2_t);
        const f32x2_t r = v - __builtin_convertvector(h, f32x2_t);
        const f16x2_t l = __builtin_convertvector(r, f16x2_t);
        hi[2 * p] = h[0]; hi[2 * p + 1] = h[1];
        lo[2 * p] = l[0]; lo[2 * p + 1] = l[1];
    }
}
// orders a fragment register behind the hand-placed lgkmcnt wait that precedes it (no instruction is emitted)
__device__ __forceinline__ void tie(u32x4& a) { asm volatile("" : "+v"(a)); }

// Implicit-GEMM Conv1d over a time-major [rows, C] activation (BigVGAN's dilated convolutions, bigvgan.hip): K index
// tap * C + ci of the tap-major weight operand multiplies A[row + (tap - half) * dil][ci], so K-tile kt of the A operand
// starts (kt / tpt - half) * dil ROWS away and at column (kt % tpt) * KT -- no im2col operand is materialised.  The caller
// pads the activation with half * dil zero rows on both sides.  tpt = K-tiles per tap (C / KT); tpt == 0: plain GEMM.
// m_base: the epilogue addresses row (m_base + m) for the kernel's row m (a launch over the LAST rows of a larger problem:
// launch_gemm's remainder launch; the caller passes A already offset by m_base rows).
struct GemmConv { int tpt = 0, dil = 0, half = 0, m_base = 0; };

template <typename T, int BM, int BN, int WM, int WN, int NS, typename Epi, int MODE = 0>
__device__ __forceinline__ void gemm_tn_glds_body(char* smem, const T* __restrict__ A, int lda, const T* __restrict__ W,
                                                  int ldw, int M, int N, int K, const Epi& epi, int xa, int xb,
                                                  const int* __restrict__ m_limit, const GemmConv cv = GemmConv{}) {
    constexpr int NW = WM * WN;
    constexpr int KT = GEMM_ROW_BYTES / sizeof(T);
    constexpr int EPC = 16 / sizeof(T);
    constexpr int TM = BM / WM, TN = BN / WN;           // wave tile
    constexpr int MI = TM / 16, NJ = TN / 16;
    constexpr int STAGE = (BM + BN) * GEMM_ROW_BYTES;   // bytes per ring stage
    constexpr int RG_A = BM / 8, RG_W = BN / 8;         // 8-row groups (= 1 KiB glds pieces)
    constexpr int LA = RG_A / NW, LW = RG_W / NW;       // pieces per wave per K-tile
    static_assert(RG_A % NW == 0 && RG_W % NW == 0, "tile rows must split evenly over the waves");
    constexpr int L = LA + LW;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave / WN, wc = wave % WN;
    // XCD-aware tile order: workgroup ids are dealt round-robin over the 8 XCDs (id % 8 labels the XCD's share); give
    // each share a compact xa x xb rectangle of output tiles so that the A rows / W rows it touches fit its private L2.
    int tile_m = blockIdx.y, tile_n = blockIdx.x;
    if (m_limit) {
        // device-side row count: which row tiles carry work is only known here, so the tile order is the one that spreads
        // ANY prefix of the row tiles evenly over the XCDs: share x of the round-robin deal owns row tiles x, x + 8, x + 16, ...
        // with all their column tiles (the launcher rounds the grid up to a multiple of 8 row tiles).  A rectangle
        // order fixed for the padded grid left the XCDs that own the trailing rectangles idle (-30 % at C3's pad fraction).
        const int tiles_n = gridDim.x;
        const int bid = blockIdx.y * tiles_n + blockIdx.x;
        const int idx = bid >> 3;
        tile_m = (idx / tiles_n) * 8 + (bid & 7);
        tile_n = idx % tiles_n;
    } else if (xa > 0) {
        const int tiles_n = gridDim.x;
        const int bid = blockIdx.y * tiles_n + blockIdx.x;
        const int xcd = bid & 7, idx = bid >> 3;          // idx-th tile of this XCD's share
        const int rects_n = tiles_n / xb;                 // rectangles per row of rectangles
        const int per_rect = xa * xb;
        const int rect = xcd + 8 * (idx / per_rect), in = idx % per_rect;
        tile_m = (rect / rects_n) * xa + in / xb;
        tile_n = (rect % rects_n) * xb + in % xb;
    }
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    // Row count known only on the device (packed variable-length batches, engine_impl.h RowPack): the grid covers the
    // padded row count so that a captured graph does not depend on the lengths; tiles past the limit retire at once.
    if (m_limit) {
        M = min(M, __builtin_amdgcn_readfirstlane(*m_limit));
        if (m0 >= M) return;
    }
    const int nkt = K / KT;
    // Orientation per 16-column sub-tile of this wave (wave-uniform).  The transposed columns are a suffix of the output
    // (EpiQKV: the V third), so within a wave they are the LAST nt of its NJ sub-tiles: nt is 0 or NJ except in the one
    // wave of a block tile that straddles the boundary (only tiles whose width does not divide 64 * heads can: BN = 192).
    bool trj[NJ];
    int nt = 0;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        trj[j] = Epi::kTransposes && epi.tile_transposed(n0 + (wave % WN) * TN + j * 16);
        nt += trj[j] ? 1 : 0;
    }
    nt = __builtin_amdgcn_readfirstlane(nt);

    // per-lane source pointers for this wave's pieces (row inside the 8-row group = lane >> 3, swizzled chunk)
    const int lr = lane >> 3, lc = (lane & 7) ^ lr;
    const T* asrc[LA];
    const T* wsrc[LW];
#pragma unroll
    for (int i = 0; i < LA; ++i) {
        const int row = min(m0 + (wave + i * NW) * 8 + lr, M - 1);
        asrc[i] = A + (size_t)row * lda + lc * EPC;
    }
#pragma unroll
    for (int i = 0; i < LW; ++i) {
        const int row = min(n0 + (wave + i * NW) * 8 + lr, N - 1);
        wsrc[i] = W + (size_t)row * ldw + lc * EPC;
    }
    auto issue_piece = [&](int p, long akoff, int koff, char* base) {  // p is a compile-time constant after unrolling
        if (p < LA) glds16(asrc[p] + akoff, base + (wave + p * NW) * 1024);
        else glds16(wsrc[p - LA] + koff, base + BM * GEMM_ROW_BYTES + (wave + (p - LA) * NW) * 1024);
    };
    // (tiles past the end are never requested: a dummy tail load would have to be waited for before the workgroup may
    // retire its LDS -- one more L2 round trip at the end of every launch)
    auto issue = [&](int kt, int stage) {
        if (kt >= nkt) return;
        char* base = smem + stage * STAGE;
        const int koff = kt * KT;
        long akoff = koff;
        if (cv.tpt > 0) {                               // implicit conv: this K-tile's tap shifts the A rows
            const int tap = kt / cv.tpt;
            akoff = (long)(tap - cv.half) * cv.dil * lda + (long)(kt - tap * cv.tpt) * KT;
        }
#pragma unroll
        for (int p = 0; p < L; ++p) issue_piece(p, akoff, koff, base);
    };
    // wait until at most `tiles` (<= NS-2) of this wave's requested tiles are still in flight
    auto wait_tiles = [&](int tiles) {
        if (tiles >= NS - 2) wait_vmcnt<(NS - 2) * L>();
        else if (NS >= 4 && tiles == NS - 3) wait_vmcnt<(NS >= 4 ? NS - 3 : 0) * L>();
        else if (NS >= 5 && tiles == NS - 4) wait_vmcnt<(NS >= 5 ? NS - 4 : 0) * L>();
        else wait_vmcnt<0>();
    };

    f32x4 acc[MI][NJ];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // fragment addressing: row R = base + l15 (base multiple of 16), chunk c = kk*4 + g  ->  R*128 + ((c ^ (R&7)) * 16)
    const int l15 = lane & 15, g = lane >> 4;
    const int sw = l15 & 7;
    const int a_row_off = (wr * TM + l15) * GEMM_ROW_BYTES;
    const int w_row_off = BM * GEMM_ROW_BYTES + (wc * TN + l15) * GEMM_ROW_BYTES;
    const int c0 = ((0 + g) ^ sw) * 16, c1 = ((4 + g) ^ sw) * 16;
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;  // LDS byte address of smem

    // Epilogue contexts and the memory operands of the store (bias, residual, gate, rotary table) are set up and REQUESTED
    // here, before the first tile: their latency hides under the K loop instead of adding a round trip after it.  They
    // are older than every LDS-DMA piece, so the counted vmcnt waits of the loop retire them first.  Large wave tiles
    // cannot afford the registers (the 256x128 tile spilled): they set the epilogue up after the loop.
    constexpr int PRE_WORDS = sizeof(typename Epi::Pre) >= 4 ? (int)sizeof(typename Epi::Pre) / 4 : 0;
    // (measured, normalised by the attention kernel of the same run: early setup helps the residual GEMMs ~2 %, is
    // neutral for FF1 and costs the 128x192 QKV tile ~3 % -- 230 VGPRs live across the loop -- so big transposing tiles
    // set up late)
    constexpr bool EARLY = MI * NJ * (1 + PRE_WORDS) <= 64 && !(Epi::kTransposes && MI * NJ >= 12);
    const int mw = m0 + wr * TM, nw = n0 + wc * TN;
    bool any_row = false, any_tr = false;
#pragma unroll
    for (int j = 0; j < NJ; ++j) { any_row |= !trj[j]; any_tr |= trj[j]; }
    typename Epi::RowCtx rc[MI];
    typename Epi::ColCtx cc[NJ];
    typename Epi::Pre pre[MI][NJ];
    typename Epi::TRowCtx trc[MI];
    typename Epi::TColCtx tcc[NJ];
    auto epilogue_setup = [&]() {
        if (any_row) {
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                const int m = min(mw + i * 16 + l15, M - 1);
                rc[i] = epi.row(m + cv.m_base);
            }
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                cc[j] = epi.col(min(nw + j * 16 + g * 4, N - 4));
#pragma unroll
                for (int i = 0; i < MI; ++i) pre[i][j] = epi.preload(rc[i], cc[j]);
            }
        }
        if (any_tr) {
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                const int m = min(mw + i * 16 + g * 4, M - 1);
                trc[i] = epi.trow(m + cv.m_base, M + cv.m_base);
            }
#pragma unroll
            for (int j = 0; j < NJ; ++j) tcc[j] = epi.tcol(min(nw + j * 16 + l15, N - 1));
        }
    };
#pragma unroll
    for (int t = 0; t < NS - 1; ++t) issue(t, t);
    if (EARLY) epilogue_setup();

    // retire every scalar/LDS operation of the prologue: with nothing of another kind pending on the lgkm counter the
    // compiler can use counted lgkmcnt(N) waits (in-order LDS returns) inside the loop instead of lgkmcnt(0)
    __builtin_amdgcn_s_waitcnt(0xC07F);
    // The K loop exists once per orientation pattern: a run-time orientation test inside it compiles to a pair of taken
    // branches around every MFMA (measured +11 us on the 2048x3072x1024 QKV projection).
    // Tried on top of this loop and dropped (tools/gemm2_sweep.py, M = 2048, cold weights): (a) staggering waves 4-7
    // half a block behind waves 0-3 so that SIMD partners alternate LDS reads and MFMAs: 13.8 -> 15.5 us on FF1;
    // (b) a second fragment set so that block kt reads tile kt+1 while it multiplies tile kt: +-2 % at M = 2048 and
    //     -9 % at M = 16384 (664 -> 605 TFLOP/s on 128x128; the 256x128 tile does not fit the registers).
    // (c) issuing the LDS-DMA pieces between MFMA rows instead of at the head of the block: out 9.1 -> 9.5 us;
    // (d) two K-tiles per barrier (ring of 6) for the small 128x64 / 64x64 wave tiles: out 10.0 -> 10.3, ff2 17.4 -> 17.1 us.
    // Floors of this loop for the 128x128 / 8-wave tile, per 64-deep K-step on a full chip (tools/gemm_scale.py with the
    // diagnostic MODEs, K = 1024 -> 8192): bare MFMA issue 0.216 us (tools/mfma_rate_probe.py: 2.3-2.4 PFLOP/s at
    // 2.1-2.37 GHz); + one barrier per step 0.28; + the 12 fragment reads per wave (96 KB of LDS reads per step), however
    // they are scheduled (up front, register-pipelined a step ahead, or interleaved two per MFMA) 0.385-0.43; LDS-DMA
    // only 0.32; everything 0.51.  The fragment reads cost ~0.1 us of MFMA issue even when nothing waits on them, so
    // only a wave tile with fewer reads per MFMA (64x64: the 256x128 configuration) moves this, and M = 2048 cannot
    // fill the chip with such tiles.
    auto kloop = [&](auto ntc) {
        constexpr int NT = decltype(ntc)::value;  // number of trailing transposed sub-tiles (compile-time per loop copy)
        constexpr int R = 2 * (MI + NJ);
        // one K-tile: all fragment reads are issued first, in the order the MFMAs consume them:
        //   per kk: a[0], w[0..NJ-1], a[1..MI-1];  the MFMA row i of kk may start once read (kk*(MI+NJ) + NJ + i) is back
        auto tile = [&](int stage) {
            u32x4 af[2][MI], wf[2][NJ];
            const unsigned sbu = (unsigned)(stage * STAGE) + lds_base;
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                const unsigned co = kk ? c1 : c0;
                lds_read_b128_asm(af[kk][0], sbu + a_row_off + co);
#pragma unroll
                for (int j = 0; j < NJ; ++j) lds_read_b128_asm(wf[kk][j], sbu + w_row_off + j * 16 * GEMM_ROW_BYTES + co);
#pragma unroll
                for (int i = 1; i < MI; ++i) lds_read_b128_asm(af[kk][i], sbu + a_row_off + i * 16 * GEMM_ROW_BYTES + co);
            }
            // MODE 5: the A operand is ALREADY split in memory (store4_planar by its producer): its two fragment reads are hi and lo.
            // (6, 7: tools/probe/gemm_split_probe.hip -- 2 of 3 / 1 of 3 MFMAs)
            if constexpr (MODE == 3 || MODE >= 5) {
                // row i needs both halves of a[i] and (i == 0) every w fragment: the last of them is read (MI + NJ) + NJ + i
#pragma unroll
                for (int i = 0; i < MI; ++i) {
                    __builtin_amdgcn_sched_barrier(0);
                    wait_row<R, MI, NJ>(1, i, af, wf);
                    tie(af[0][i]);
                    if (i == 0) {
#pragma unroll
                        for (int j = 0; j < NJ; ++j) tie(wf[0][j]);
                    }
                    f16x8 ah, al;
                    if constexpr (MODE == 5) { ah = __builtin_bit_cast(f16x8, af[0][i]); al = __builtin_bit_cast(f16x8, af[1][i]); }
                    else split8_f16(af[0][i], af[1][i], ah, al);
                    // term-major: consecutive MFMAs write different accumulators (a dependent MFMA waits out the whole
                    // pipeline of its predecessor)
#pragma unroll
                    for (int term = (MODE == 6 ? 1 : MODE == 7 ? 2 : 0); term < 3; ++term) {
#pragma unroll
                        for (int j = 0; j < NJ; ++j) {
                            const f16x8 w = __builtin_bit_cast(f16x8, wf[term == 0 ? 1 : 0][j]);   // lo hi hi
                            const f16x8 a = term == 1 ? al : ah;                                    // hi lo hi
                            if (j < NJ - NT) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w, a, acc[i][j], 0, 0, 0);
                            else acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, w, acc[i][j], 0, 0, 0);
                        }
                    }
                }
                return;
            }
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
                for (int i = 0; i < MI; ++i) {
                    __builtin_amdgcn_sched_barrier(0);  // keep the previous row's MFMAs above this wait
                    // outstanding reads allowed when row i of kk starts = R - 1 - (index of the last read it needs)
                    wait_row<R, MI, NJ>(kk, i, af, wf);
#pragma unroll
                    for (int j = 0; j < NJ; ++j) {
                        if (j < NJ - NT) acc[i][j] = Mma<T>::run(wf[kk][j], af[kk][i], acc[i][j]);
                        else acc[i][j] = Mma<T>::run(af[kk][i], wf[kk][j], acc[i][j]);
                    }
                }
            }
        };
        int stage = 0;
        {
            for (int kt = 0; kt < nkt; ++kt) {
                if (MODE != 2) wait_tiles(nkt - 1 - kt);  // this wave's pieces of tile kt have landed (later tiles may be in flight) ...
                __builtin_amdgcn_s_barrier();     // ... and so have everyone else's; stage (kt-1)%NS is free again
                int pf = stage + NS - 1;
                if (pf >= NS) pf -= NS;
                if (MODE != 2) issue(kt + NS - 1, pf);
                if (MODE != 1) tile(stage);
                stage = stage + 1 == NS ? 0 : stage + 1;
            }
        }
    };
    if constexpr (Epi::kTransposes) {
        constexpr bool MIXED = (BN % 64) != 0 || BN == 192;  // block tiles of 64/128/256 columns never straddle the boundary
        if (nt == 0) kloop(std::integral_constant<int, 0>{});
        else if (nt == NJ) kloop(std::integral_constant<int, NJ>{});
        else if constexpr (MIXED && NJ >= 2) {
            if (nt == 1) kloop(std::integral_constant<int, 1>{});
            else if constexpr (NJ >= 3) {
                if (nt == 2) kloop(std::integral_constant<int, 2>{});
                else if constexpr (NJ >= 4) kloop(std::integral_constant<int, 3>{});
            }
        }
    } else {
        kloop(std::integral_constant<int, 0>{});
    }

    if (MODE == 4 && K > 0) {  // diagnostic: no epilogue (K <= 0 never happens; keeps the accumulators alive)
        float sink = 0.f;
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j) sink += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
        if (sink == 1234.5678f && lda < 0) *reinterpret_cast<volatile float*>(smem) = sink;
        return;
    }
    if (!EARLY) epilogue_setup();
    // Every memory operand of the epilogue (bias, residual, gate, rotary table) has been requested by now; retire them with
    // ONE wait the compiler can see.  Without it hipcc's wait-count pass -- which cannot see the loop's asm waits and has to
    // merge the "pending load" state across the exec-masked bounds checks around each store -- puts `s_waitcnt vmcnt(0)` in
    // front of EVERY store: each store then waits for the previous one's round trip (4-12 serialized stores per lane).
    __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0) only
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int n = nw + j * 16 + g * 4;
        if (trj[j] || n >= N) continue;
#pragma unroll
        for (int i = 0; i < MI; ++i)
            if (mw + i * 16 + l15 < M) epi.store(rc[i], cc[j], acc[i][j], pre[i][j]);
    }
    if (any_tr) {
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int n = nw + j * 16 + l15;
            if (!trj[j] || n >= N) continue;
#pragma unroll
            for (int i = 0; i < MI; ++i)
                if (mw + i * 16 + g * 4 < M) epi.tstore(trc[i], tcc[j], acc[i][j]);
        }
    }
}

template <typename T, int BM, int BN, int WM, int WN, int NS, typename Epi, int MODE = 0>
__global__ __launch_bounds__(WM* WN * 64) void gemm_tn_glds_kernel(const T* __restrict__ A, int lda,
                                                                   const T* __restrict__ W, int ldw, int M, int N, int K,
                                                                   Epi epi, int xa, int xb, const int* __restrict__ m_limit,
                                                                   const GemmConv cv) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    gemm_tn_glds_body<T, BM, BN, WM, WN, NS, Epi, MODE>(smem, A, lda, W, ldw, M, N, K, epi, xa, xb, m_limit, cv);
}

// chooses (xa, xb): tiles_m % xa == 0, tiles_n % xb == 0 and the number of rectangles a multiple of 8; prefers the most
// square rectangle with xa * xb close to one XCD's share of a full wave of workgroups (32 CUs).  (0, 0) = keep order.
inline int& xcd_mode() { static int m = 1; return m; }
inline void pick_xcd_rect(int tiles_m, int tiles_n, int* xa, int* xb) {
    *xa = *xb = 0;
    if (!xcd_mode()) return;
    long best = -1;
    for (int a = 1; a <= tiles_m && a <= 16; ++a) {
        if (tiles_m % a) continue;
        for (int b = 1; b <= tiles_n && b <= 32; ++b) {
            if (tiles_n % b) continue;
            const long rects = (long)(tiles_m / a) * (tiles_n / b);
            if (rects % 8) continue;
            const int area = a * b;
            if (area > 64) continue;
            // score: prefer area near 32, then small perimeter (a + b)
            const long score = 1000L * (64 - (area > 32 ? area - 32 : 32 - area)) - 10L * (a + b);
            if (score > best) { best = score; *xa = a; *xb = b; }
        }
    }
    if (*xa * *xb <= 1) *xa = *xb = 0;
}

template <typename T, int BM, int BN, int WM, int WN, int NS, typename Epi, int MODE = 0>
inline hipError_t launch_gemm2_raw(hipStream_t s, const T* A, int lda, const T* W, int ldw, int M, int N, int K,
                                   const Epi& epi, const int* m_limit = nullptr, const GemmConv& cv = GemmConv{}) {
    constexpr int smem = NS * (BM + BN) * GEMM_ROW_BYTES;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_tn_glds_kernel<T, BM, BN, WM, WN, NS, Epi, MODE>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, smem);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    dim3 grid((N + BN - 1) / BN, (M + BM - 1) / BM);
    if (m_limit) grid.y = (grid.y + 7) / 8 * 8;   // (the device-side tile order deals row tiles in groups of 8)
    // rectangle of tiles per XCD share: valid only when the grid splits into whole rectangles, 8 at a time
    int xa = 0, xb = 0;
    if (!m_limit) pick_xcd_rect((int)grid.y, (int)grid.x, &xa, &xb);
    hipLaunchKernelGGL((gemm_tn_glds_kernel<T, BM, BN, WM, WN, NS, Epi, MODE>), grid, dim3(WM * WN * 64), smem, s, A, lda, W,
                       ldw, M, N, K, epi, xa, xb, m_limit, cv);
    return hipGetLastError();
}

template <typename T, int BM, int BN, int WM, int WN, int NS, typename Epi, int MODE = 0>
inline hipError_t launch_gemm2_cfg(hipStream_t s, const T* A, int lda, const T* W, int ldw, int M, int N, int K,
                                   const Epi& epi, const int* m_limit = nullptr, const GemmConv& cv = GemmConv{}) {
    return with_static_act(epi, [&](const auto& e) {
        return launch_gemm2_raw<T, BM, BN, WM, WN, NS, std::decay_t<decltype(e)>, MODE>(s, A, lda, W, ldw, M, N, K, e, m_limit, cv);
    });
}

}  // namespace f5
