// Flash attention v2 for dim_head = 64 (same contract as attn.h): 8 waves per 128-query block, K/V^T tiles streamed by
// LDS-DMA through a 3-stage ring, key-split wave pairs.
//
//   wave = (qg, kh): query group qg (32 query rows, as in attn.h) x key half kh of every 64-key tile.  Each wave runs
//   its OWN online softmax over its key subset (running max m, partial sum l, O^T accumulator); the two halves of a
//   pair are merged once at the end through LDS (flash-decoding style):  m = max(m0, m1), O = O0 e^(m0-m) + O1 e^(m1-m).
//   Two waves per SIMD: one's softmax VALU work overlaps the other's MFMAs (dh = 64 attention is VALU-heavy:
//   ~0.34 VALU cycles per score vs 0.25 MFMA cycles).
//   Softmax reference (VAR bit 2, the default): q arrives pre-scaled by log2 e (attention_q_scale), the score
//   accumulators start at -reference, and the reference only moves when a tile exceeds it by 2^8 (or on a wave's first
//   tile), decided by an integer max3 tree on the score bit patterns; the move (rescale of l and O) is an out-of-line
//   slow path.  Steady state per score: exp2 + add.  (VAR bit 2 clear: exact running maximum, rescale skipped while
//   no lane sees its maximum grow.)
//   Key order inside a 32-key half: MFMA row 4a + b of S^T sub-tile ks is key 8a + 4ks + b, so that lane group g ends
//   up owning the 8 CONSECUTIVE keys 8g..8g+7 -- its P^T fragment pairs with ONE 16-byte V^T fragment read.
//   LDS image: K tile [64 keys][128 B], 16-byte chunk c of row r in slot c ^ kswz(r); V^T tile [64 dh][128 B], slot
//   c ^ (r & 7); both make every ds_read_b128 fragment read conflict-free.  The swizzle is applied to the per-lane
//   SOURCE address of the LDS-DMA (the LDS side of a global_load_lds is lane-linear).
// 16-bit operands (T = bf16_t or f16_t); the exact-f32 precision keeps attn.h's kernel.
#pragma once
#include "attn.h"
#include "gemm2.h"

namespace f5 {

// One KV tile for one wave.  STAGE and EDGE are compile-time so that LDS addresses fold into instruction offsets and the
// key-padding mask costs nothing on interior tiles.
// max over the 4 lane groups (lanes l, l^16, l^32, l^48) with the gfx950 half/row swaps instead of LDS permutes
__device__ __forceinline__ float group_max(float v) {
    const unsigned u = __builtin_bit_cast(unsigned, v);
    auto a = __builtin_amdgcn_permlane32_swap(u, u, false, false);   // {own | partner l^32} in some order per half
    float m = fmaxf(__builtin_bit_cast(float, (unsigned)a[0]), __builtin_bit_cast(float, (unsigned)a[1]));
    const unsigned w = __builtin_bit_cast(unsigned, m);
    auto b = __builtin_amdgcn_permlane16_swap(w, w, false, false);   // {own | partner l^16}
    return fmaxf(__builtin_bit_cast(float, (unsigned)b[0]), __builtin_bit_cast(float, (unsigned)b[1]));
}

// One KV tile for one wave.  STAGE and EDGE are compile-time so that LDS addresses fold into instruction offsets and the
// key-padding mask costs nothing on interior tiles.  All eight fragment reads of the tile (4 K, 4 V^T) are issued
// up front as asm ds_reads with hand-counted lgkmcnt waits (gemm2.h explains why).
template <typename T, int STAGE_IDX, bool EDGE, int VAR>
__device__ __forceinline__ void attn2_tile(const char* smem_ptr, unsigned lds_base, int k_off, int kc0, int kc1, int v_off, int vc,
                                           const u32x4 (&qf)[2][2], f32x4 (&o)[4][2], float (&mrun)[2], float (&lrun)[2],
                                           int key_base, int kv_len, bool first) {
    constexpr bool LAZY = (VAR & 4) != 0;
    constexpr int TILE = 64 * 128;
    constexpr float L2E = 1.4426950408889634f;
    const unsigned sb = lds_base + STAGE_IDX * (2 * TILE);
    u32x4 kf[2][2], vf[4];   // kf[f][ks]
    const char* sbp = smem_ptr + STAGE_IDX * (2 * TILE);
#pragma unroll
    for (int f = 0; f < 2; ++f)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            if (VAR & 1) lds_read_b128_asm(kf[f][ks], sb + k_off + ks * 4 * 128 + (f ? kc1 : kc0));
            else kf[f][ks] = *reinterpret_cast<const u32x4*>(sbp + k_off + ks * 4 * 128 + (f ? kc1 : kc0));
        }
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
        if (VAR & 1) lds_read_b128_asm(vf[dt], sb + v_off + dt * 16 * 128 + vc);
        else vf[dt] = *reinterpret_cast<const u32x4*>(sbp + v_off + dt * 16 * 128 + vc);
    }
    // ---- S^T (this wave's 32 keys x 32 queries); MFMA row i = 4a + b of sub-tile ks holds key 8a + 4ks + b
    f32x4 s[2][2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int qs = 0; qs < 2; ++qs) {
            // LAZY: q arrives pre-scaled by log2(e) and the accumulators start at -reference, so the MFMAs deliver
            // log2-unit scores already relative to the running reference: no multiply, no subtract per score
            const float c0 = LAZY ? -mrun[qs] : 0.f;
            s[ks][qs] = f32x4{c0, c0, c0, c0};
        }
    if (VAR & 1) {
        __builtin_amdgcn_sched_barrier(0);
        wait_lgkm<6>(kf[0][0], kf[0][1]);
    }
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int qs = 0; qs < 2; ++qs) s[ks][qs] = Mma<T>::run(kf[0][ks], qf[qs][0], s[ks][qs]);
    if (VAR & 1) {
        __builtin_amdgcn_sched_barrier(0);
        wait_lgkm<4>(kf[1][0], kf[1][1]);
    }
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int qs = 0; qs < 2; ++qs) s[ks][qs] = Mma<T>::run(kf[1][ks], qf[qs][1], s[ks][qs]);
    // ---- online softmax over this wave's keys (lane group g owns keys key_base + 4 ks + r)
    if (LAZY) {
        // Softmax is shift-invariant: any per-query reference c works as long as 2^(s - c) stays in range.  c follows the
        // running maximum only when a tile exceeds it by more than 2^LAZY_THR (and on the first tile, where it is SET to
        // the tile maximum), so after the first tiles the per-score work is max -> exp2 -> add: p <= 2^LAZY_THR = 256.
        constexpr float LAZY_THR = 8.0f;
        // Fast test "does any score of this tile exceed the reference by more than LAZY_THR": for a positive threshold,
        // s > THR  <=>  (int)bits(s) > (int)bits(THR) (positive floats order like their bit patterns, negative floats are
        // negative integers, NaNs of either sign land in the slow path or are ignored like fmaxf would), so the tile
        // maximum is an INTEGER max3 tree: no canonicalising v_max on the MFMA results.  The float maximum itself is
        // only needed on the slow path.
        int imax[2];
#pragma unroll
        for (int qs = 0; qs < 2; ++qs) {
            if (EDGE) {
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (key_base + ks * 4 + r >= kv_len) s[ks][qs][r] = -1e30f;
            }
            auto bits = [](float x) { return __builtin_bit_cast(int, x); };
            int m = max(max(max(bits(s[0][qs][0]), bits(s[0][qs][1])), max(bits(s[0][qs][2]), bits(s[0][qs][3]))),
                        max(max(bits(s[1][qs][0]), bits(s[1][qs][1])), max(bits(s[1][qs][2]), bits(s[1][qs][3]))));
            imax[qs] = m;
        }
        const int thr_bits = __builtin_bit_cast(int, LAZY_THR);
        const bool upd = first || imax[0] > thr_bits || imax[1] > thr_bits;   // per lane; any lane of the wave -> slow path
        if (__builtin_expect(__any(upd), 0)) {  // wave-uniform, rare after the first tiles: move the reference, rescale l and O
#pragma unroll
            for (int qs = 0; qs < 2; ++qs) {
                float mloc = fmaxf(fmaxf(fmaxf(s[0][qs][0], s[0][qs][1]), fmaxf(s[0][qs][2], s[0][qs][3])),
                                   fmaxf(fmaxf(s[1][qs][0], s[1][qs][1]), fmaxf(s[1][qs][2], s[1][qs][3])));
                mloc = group_max(mloc);           // the 4 lane groups of a query column agree on its maximum ...
                const float dq = (first || mloc > LAZY_THR) ? mloc : 0.f;   // ... and therefore on the new reference
                // (first tile: l and O are still zero and dq may be the -1e30 of a fully masked tile: keep alpha finite)
                const float alpha = __builtin_amdgcn_exp2f(-fmaxf(dq, 0.f));
                mrun[qs] += dq;
                lrun[qs] *= alpha;
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                    for (int r = 0; r < 4; ++r) s[ks][qs][r] -= dq;
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) {
                    o[dt][qs][0] *= alpha; o[dt][qs][1] *= alpha; o[dt][qs][2] *= alpha; o[dt][qs][3] *= alpha;
                }
            }
        }
#pragma unroll
        for (int qs = 0; qs < 2; ++qs) {
            float psum = 0.f;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float p = __builtin_amdgcn_exp2f(s[ks][qs][r]);
                    if (EDGE && key_base + ks * 4 + r >= kv_len) p = 0.f;
                    s[ks][qs][r] = p;
                    psum += p;
                }
            lrun[qs] += psum;
        }
    } else {
    float alpha[2];
    bool grew = false;
#pragma unroll
    for (int qs = 0; qs < 2; ++qs) {
        if (EDGE) {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (key_base + ks * 4 + r >= kv_len) s[ks][qs][r] = -1e30f;
        }
        // scores in log2 units first: the products are compiler-visible VALU results (hipcc pads the MFMA -> VALU
        // wait states itself and knows they are canonical, so fmaxf() needs no extra canonicalising v_max and fuses
        // into v_max3).  NEVER feed MFMA accumulators to inline-asm VALU: nothing pads that hazard (stale reads).
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int r = 0; r < 4; ++r) s[ks][qs][r] *= L2E;
        float mloc = fmaxf(fmaxf(fmaxf(s[0][qs][0], s[0][qs][1]), fmaxf(s[0][qs][2], s[0][qs][3])),
                           fmaxf(fmaxf(s[1][qs][0], s[1][qs][1]), fmaxf(s[1][qs][2], s[1][qs][3])));
        if (VAR & 2) {
            mloc = group_max(mloc);
        } else {
            mloc = fmaxf(mloc, __shfl_xor(mloc, 16, 64));
            mloc = fmaxf(mloc, __shfl_xor(mloc, 32, 64));
        }
        const float mnew = fmaxf(mrun[qs], mloc);
        grew = grew || (mnew > mrun[qs]);
        alpha[qs] = __builtin_amdgcn_exp2f(mrun[qs] - mnew);   // running max kept in log2 units
        mrun[qs] = mnew;
        float psum = 0.f;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float p = __builtin_amdgcn_exp2f(s[ks][qs][r] - mnew);
                if (EDGE && key_base + ks * 4 + r >= kv_len) p = 0.f;
                s[ks][qs][r] = p;
                psum += p;
            }
        lrun[qs] = lrun[qs] * alpha[qs] + psum;
    }
    if (__any(grew)) {  // wave-uniform: skip the O-wide rescale when every factor is exactly 1
#pragma unroll
        for (int qs = 0; qs < 2; ++qs)
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                o[dt][qs][0] *= alpha[qs]; o[dt][qs][1] *= alpha[qs]; o[dt][qs][2] *= alpha[qs]; o[dt][qs][3] *= alpha[qs];
            }
    }
    }
    // ---- O^T += V^T P^T: one 32-key MFMA step; lane group g supplies keys 8g .. 8g+7 of this wave's half on both sides
    u32x4 pf2[2];
#pragma unroll
    for (int qs = 0; qs < 2; ++qs) {
        pf2[qs] = pack8<T>(s[0][qs][0], s[0][qs][1], s[0][qs][2], s[0][qs][3], s[1][qs][0], s[1][qs][1], s[1][qs][2], s[1][qs][3]);
    }
    if (VAR & 1) {
        __builtin_amdgcn_sched_barrier(0);
        wait_lgkm<0>(vf[0], vf[1], vf[2], vf[3]);
    }
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
#pragma unroll
        for (int qs = 0; qs < 2; ++qs) o[dt][qs] = Mma<T>::run(vf[dt], pf2[qs], o[dt][qs]);
    }
}

// K-tile swizzle: 16-byte chunk c of key row r lives in slot c ^ kswz(r).  The S^T fragment of sub-tile ks reads the 16
// rows {8a + 4ks + b}; kswz makes those land on 16 distinct 16-byte slots of the 256-byte bank row.
__device__ __forceinline__ int kswz(int r) { return (((r >> 3) & 3) << 1) | ((r >> 1) & 1); }

template <typename T, int VAR>
__global__ __launch_bounds__(512) void attn2_fwd_kernel(const T* __restrict__ Q, const T* __restrict__ K,
                                                               const T* __restrict__ Vt, T* __restrict__ O, int H,
                                                               int N, int Npad, const int* __restrict__ kv_lens,
                                                               int nbatch_lens, const int* __restrict__ q_lens,
                                                               const int* __restrict__ o_row_start, float* __restrict__ Of = nullptr) {
    // Of (F5_PREC_F16X3, or null): the output as f32 rows PRE-SPLIT into f16 hi / lo planes (store4_planar) -- the A operand of the
    // out-projection's pre-split GEMM -- instead of T rows in O
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // query blocks wholly past the sample's own length (padded batches): their output rows are zeroed by the reference
    // (modules.py:540-542; here: never read, the out-projection epilogue masks those rows) -- exit before any barrier
    int bhid, qblk;
    attn_block_map(bhid, qblk);   // (attn.h: the query blocks of a head run back to back on one XCD)
    if (q_lens && qblk * 128 >= q_lens[(bhid / H) % nbatch_lens]) return;
    constexpr int NS = 3;
    constexpr int TILE = 64 * 128;      // bytes of one K or V^T tile
    constexpr int STAGE = 2 * TILE;
    constexpr int L = 2;                // LDS-DMA pieces per wave per tile (16 pieces of 1 KiB over 8 waves)
    constexpr float L2E = 1.4426950408889634f;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int qg = wave >> 1, kh = wave & 1;
    const int l15 = lane & 15, g = lane >> 4;
    const int b = bhid / H, h = bhid - b * H;
    const size_t bh = (size_t)bhid;
    const int q0 = qblk * 128 + qg * 32;
    int kv_len = N;
    if (kv_lens) kv_len = min(N, kv_lens[b % nbatch_lens]);
    const int nkt = (kv_len + 63) / 64;

    // Q fragments (B operand of S^T = K Q^T): row q, 16-byte chunk f*4 + g
    u32x4 qf[2][2];
#pragma unroll
    for (int qs = 0; qs < 2; ++qs) {
        const int q = min(q0 + qs * 16 + l15, N - 1);
#pragma unroll
        for (int f = 0; f < 2; ++f)
            qf[qs][f] = *reinterpret_cast<const u32x4*>(reinterpret_cast<const char*>(Q + (bh * N + q) * 64) + f * 64 + g * 16);
    }

    // LDS-DMA sources: wave w stages piece w of the K tile (keys 8w..8w+7) and piece w of the V^T tile (dh rows 8w..8w+7)
    const int lr = lane >> 3, ls = lane & 7;
    const int krow = wave * 8 + lr;                       // key row inside the tile
    const int kchunk = ls ^ kswz(krow);
    const int vrow = wave * 8 + lr;                       // dh row
    const int vchunk = ls ^ (vrow & 7);
    const T* ksrc = K + bh * (size_t)N * 64 + kchunk * 8;
    const T* vsrc = Vt + (bh * 64 + vrow) * (size_t)Npad + vchunk * 8;
    auto issue = [&](int kt, int stage) {
        if (kt >= nkt) return;   // nothing past the last tile: no dummy load to wait for at the end
        const int t = kt;
        char* base = smem + stage * STAGE;
        const int key = min(t * 64 + krow, N - 1);        // rows past N are clamped; their scores are masked
        glds16(ksrc + (size_t)key * 64, base + wave * 1024);
        glds16(vsrc + t * 64, base + TILE + wave * 1024);
    };

    f32x4 o[4][2];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
#pragma unroll
        for (int qs = 0; qs < 2; ++qs) o[dt][qs] = f32x4{0.f, 0.f, 0.f, 0.f};
    // running reference (log2 units): LAZY starts at 0 and SETS it on the first tile; the exact-max path starts at -inf
    float mrun[2] = {(VAR & 4) ? 0.f : -1e30f, (VAR & 4) ? 0.f : -1e30f}, lrun[2] = {0.f, 0.f};

    // fragment read offsets (stage base and sub-tile strides are compile-time immediates in attn2_tile)
    const int ka = l15 >> 2, kb = l15 & 3;
    const int krd = kh * 32 + 8 * ka + kb;                // + 4 ks
    const int ksw = kswz(krd);                            // independent of ks (bit 2 of the row is not used)
    const int k_off = krd * 128;
    const int kc0 = ((0 + g) ^ ksw) * 16, kc1 = ((4 + g) ^ ksw) * 16;
    const int v_off = TILE + l15 * 128;
    const int vc = ((4 * kh + g) ^ (l15 & 7)) * 16;       // keys 32 kh + 8 g .. + 7 of dh row dt*16 + l15
    const int key_lane = kh * 32 + 8 * g;                 // first key (inside the tile) owned by this lane group
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;

    issue(0, 0);
    issue(1, 1);
    if (VAR & 1) __builtin_amdgcn_s_waitcnt(0xC07F);      // nothing but the tile loop's own LDS reads on the lgkm counter
    const int nfull = kv_len / 64;                        // tiles without masked keys
    int kt = 0;
#define F5_ATTN_STEP(SI, EDGE_)                                                                                        \
    {                                                                                                                  \
        if (kt + 1 < nkt) wait_vmcnt<(NS - 2) * L>(); else wait_vmcnt<0>();                                            \
        __builtin_amdgcn_s_barrier();                                                                                  \
        issue(kt + NS - 1, (SI + NS - 1) % NS);                                                                        \
        attn2_tile<T, SI, EDGE_, VAR>(smem, lds_base, k_off, kc0, kc1, v_off, vc, qf, o, mrun, lrun, kt * 64 + key_lane, kv_len, kt == 0);        \
        ++kt;                                                                                                          \
    }
    while (kt + 3 <= nfull) {
        F5_ATTN_STEP(0, false)
        F5_ATTN_STEP(1, false)
        F5_ATTN_STEP(2, false)
    }
    // remainder: up to 2 full tiles + at most one edge tile, continuing the stage rotation (kt % 3 == 0 here)
    if (kt < nkt) { if (kt < nfull) F5_ATTN_STEP(0, false) else F5_ATTN_STEP(0, true) }
    if (kt < nkt) { if (kt < nfull) F5_ATTN_STEP(1, false) else F5_ATTN_STEP(1, true) }
    if (kt < nkt) { F5_ATTN_STEP(2, true) }
#undef F5_ATTN_STEP
    __builtin_amdgcn_s_barrier();  // every wave is done with the ring (and no load is in flight): reuse it for the pair merge

    // ---- merge the two key halves of each query group (kh = 1 publishes, kh = 0 combines and stores)
#pragma unroll
    for (int qs = 0; qs < 2; ++qs) {
        float l = lrun[qs];
        l += __shfl_xor(l, 16, 64);
        l += __shfl_xor(l, 32, 64);
        lrun[qs] = l;
    }
    float* mb = reinterpret_cast<float*>(smem) + (size_t)qg * (36 * 64);   // [36][64 lanes] per pair
    if (kh == 1) {
#pragma unroll
        for (int dt = 0; dt < 4; ++dt)
#pragma unroll
            for (int qs = 0; qs < 2; ++qs)
#pragma unroll
                for (int r = 0; r < 4; ++r) mb[((dt * 2 + qs) * 4 + r) * 64 + lane] = o[dt][qs][r];
        mb[32 * 64 + lane] = mrun[0];
        mb[33 * 64 + lane] = mrun[1];
        mb[34 * 64 + lane] = lrun[0];
        mb[35 * 64 + lane] = lrun[1];
    }
    __syncthreads();
    if (kh == 0) {
#pragma unroll
        for (int qs = 0; qs < 2; ++qs) {
            const float m1 = mb[(32 + qs) * 64 + lane], l1 = mb[(34 + qs) * 64 + lane];
            const float m = fmaxf(mrun[qs], m1);
            const float a0 = exp2f(mrun[qs] - m), a1 = exp2f(m1 - m);   // maxima are kept in log2 units
            const float inv = 1.0f / (lrun[qs] * a0 + l1 * a1);
            const int q = q0 + qs * 16 + l15;
            // (o_row_start: output rows of a packed variable-length batch, RowPack: batch row b owns rows o_row_start[b] ..)
            const size_t orow = o_row_start ? (size_t)o_row_start[b] + q : (size_t)b * N + q;
            if (q < (o_row_start ? min(N, o_row_start[b + 1] - o_row_start[b]) : N)) {
                T* dst = O + orow * (H * 64) + h * 64 + g * 4;
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) {
                    float v[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        v[r] = (o[dt][qs][r] * a0 + mb[((dt * 2 + qs) * 4 + r) * 64 + lane] * a1) * inv;
                    if (Of) store4_planar(Of + orow * (H * 64), h * 64 + g * 4 + dt * 16, v[0], v[1], v[2], v[3]);
                    else store4(dst + dt * 16, v[0], v[1], v[2], v[3]);
                }
            }
        }
    }
}

// diagnostic switch (bit 0: asm LDS reads + counted waits, bit 1: permlane swaps, bit 2: lazy softmax reference with q
// pre-scaled by log2 e).  Producers of q must use attention_q_scale<T>().
inline int& attn2_variant() { static int v = 7; return v; }
template <typename T> inline float attention_q_scale();   // dim_head^-0.5 (x log2 e where the kernel works in log2 units)
template <> inline float attention_q_scale<float>() { return 0.125f; }
template <> inline float attention_q_scale<bf16_t>() { return (attn2_variant() & 4) ? 0.125f * 1.4426950408889634f : 0.125f; }
template <> inline float attention_q_scale<f16_t>() { return attention_q_scale<bf16_t>(); }
template <typename T>
inline hipError_t launch_attention_v2(hipStream_t s, const T* Q, const T* K, const T* Vt, T* O, int Bp, int H,
                                      int N, int Npad, const int* kv_lens, int nbatch_lens, const int* q_lens, const int* o_row_start,
                                      float* o_planar_f32 = nullptr) {
    constexpr int smem = 3 * 2 * 64 * 128;  // 48 KiB ring (>= 4 * 36 * 64 * 4 = 36 KiB merge scratch)
    dim3 grid(Bp * H, (N + 127) / 128);
    const int var = attn2_variant();
#define F5_ATTN2_LAUNCH(V)                                                                                             \
    {                                                                                                                  \
        static bool attr_set = false;                                                                                  \
        if (!attr_set) {                                                                                               \
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&attn2_fwd_kernel<T, V>),                    \
                                               hipFuncAttributeMaxDynamicSharedMemorySize, smem);                      \
            if (e != hipSuccess) return e;                                                                             \
            attr_set = true;                                                                                           \
        }                                                                                                              \
        hipLaunchKernelGGL((attn2_fwd_kernel<T, V>), grid, dim3(512), smem, s, Q, K, Vt, O, H, N, Npad, kv_lens, nbatch_lens, q_lens, o_row_start, o_planar_f32); \
    }
    if (var == 0) F5_ATTN2_LAUNCH(0)
    else if (var == 3) F5_ATTN2_LAUNCH(3)
    else F5_ATTN2_LAUNCH(7)
#undef F5_ATTN2_LAUNCH
    return hipGetLastError();
}

// precision dispatch: bf16 / f16 -> v2 (this file), f32 -> attn.h
// (kv_lens: key-padding mask, the reference's attn_mask_enabled; q_lens: rows whose output nobody reads; both index
//  batch row b as lens[b % nbl] and may be null)
inline hipError_t launch_attention_any(hipStream_t s, const bf16_t* Q, const bf16_t* K, const bf16_t* Vt, bf16_t* O, int Bp,
                                       int H, int N, int Npad, const int* kv_lens, int nbl, const int* q_lens = nullptr,
                                       const int* o_row_start = nullptr, bool /*split16*/ = false, bool /*o_planar*/ = false, int /*hi_only*/ = 0) {
    return launch_attention_v2<bf16_t>(s, Q, K, Vt, O, Bp, H, N, Npad, kv_lens, nbl, q_lens, o_row_start);
}
inline hipError_t launch_attention_any(hipStream_t s, const f16_t* Q, const f16_t* K, const f16_t* Vt, f16_t* O, int Bp,
                                       int H, int N, int Npad, const int* kv_lens, int nbl, const int* q_lens = nullptr,
                                       const int* o_row_start = nullptr, bool /*split16*/ = false, bool /*o_planar*/ = false, int /*hi_only*/ = 0) {
    return launch_attention_v2<f16_t>(s, Q, K, Vt, O, Bp, H, N, Npad, kv_lens, nbl, q_lens, o_row_start);
}
inline hipError_t launch_attention_any(hipStream_t s, const float* Q, const float* K, const float* Vt, float* O, int Bp, int H,
                                       int N, int Npad, const int* kv_lens, int nbl, const int* q_lens = nullptr,
                                       const int* o_row_start = nullptr, bool split16 = false, bool o_planar = false, int hi_only = 0) {
    if (split16) return launch_attention_split(s, Q, K, Vt, O, Bp, H, N, Npad, kv_lens, nbl, q_lens, o_row_start, o_planar, hi_only);   // F5_PREC_F16X3
    return launch_attention<float>(s, Q, K, Vt, O, Bp, H, N, Npad, kv_lens, nbl, q_lens, o_row_start);
}

}  // namespace f5
