// Non-causal flash attention for dim_head = 64 on gfx950 MFMA (reference: modules.py:499-508, torch SDPA backend).
//
// Inputs come from the fused QKV epilogue (gemm.h EpiQKV): Q (pre-scaled by 64^-0.5, rotary applied) and K as
// [B', H, N, 64], V TRANSPOSED as [B', H, 64, Npad].  Output O as [B', N, H*64].
//
// Everything is computed transposed so that no LDS round trip or cross-lane shuffle is needed for P:
//   S^T[key, q] = K . Q^T        MFMA(A = K fragment, B = Q fragment)   -> lane owns ONE q column (l & 15) and, per
//                                16-key sub-tile, keys 4*(l>>4)..+3 in its 4 accumulator registers
//   O^T[dh, q]  = V^T . P^T      MFMA(A = V^T fragment, B = P^T fragment): the B operand wants, per lane, k-values of
//                                column q -- exactly what the S^T accumulators already are.  bf16: the 8 k-slots of a
//                                32-key step are filled with keys {4g..4g+3} of two adjacent 16-key sub-tiles, and
//                                the V^T fragment is read with the same key permutation (two 8-byte LDS reads).
//                                f32 (16x16x4): step r of a 16-key sub-tile takes accumulator register r directly.
// Softmax statistics are per q column = per lane: the running max needs two shuffles (xor 16, 32) per tile, the
// rescale factor and the running sum are lane-local (partial sums are combined once at the end).
//
// Block = 4 waves x 32 q rows = 128 q rows of one (b', h); KV tiles of 64 keys, double-buffered in LDS with register
// prefetch (one barrier per tile).  Keys >= kv_len (the sample's own length when the reference's attn_mask_enabled is
// set, else N) are masked; K/V tiles beyond kv_len are never loaded.  Query blocks that lie wholly past q_lens[b] (the
// sample's own length in a padded batch) exit at once: the reference zeroes those rows of the attention output
// (modules.py:540-542, here the row mask of the out-projection epilogue), so nothing reads them.
#pragma once
#include "gemm.h"

namespace f5 {

// Workgroup -> (batch row x head, query block).  The grid is (Bp * H, query blocks) and workgroup ids (x fastest) are dealt
// round-robin over the 8 XCDs, so with the plain mapping all query blocks of one head already meet in ONE XCD's L2 -- but a layer
// of 512 different heads is dispatched between two query blocks of the same head: at C3's 32 x 16 heads every XCD streams 16 MB
// of K / V per layer through its 4 MB L2 and each query block fetches its head's K / V again (rocprofv3 FETCH_SIZE at 32,768
// rows: 1,141 MB per launch against 201 MB of q + k + v, 5.2 TB/s on the fabric).  Here the ids of one XCD walk the query blocks
// of a head before the next head (8 heads x 8 blocks x 256 KB of K / V = 2 MB live per XCD).
__device__ __forceinline__ void attn_block_map(int& bhid, int& qblk) {
    const int nbh = gridDim.x, nqb = gridDim.y;
    bhid = blockIdx.x;
    qblk = blockIdx.y;
    if ((nbh & 7) == 0) {
        const int lin = blockIdx.y * nbh + blockIdx.x;
        const int xcd = lin & 7, j = lin >> 3;
        qblk = j % nqb;
        bhid = (j / nqb) * 8 + xcd;
    }
}


template <typename T>
__global__ __launch_bounds__(256) void attn_fwd_kernel(const T* __restrict__ Q, const T* __restrict__ K,
                                                       const T* __restrict__ Vt, T* __restrict__ O, int H, int N,
                                                       int Npad, const int* __restrict__ kv_lens, int nbatch_lens,
                                                       const int* __restrict__ q_lens, const int* __restrict__ o_row_start) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // grid: x = (batch row, head), y = query block: consecutive workgroup ids are dealt round-robin over the 8 XCDs, so with
    // heads on x the query blocks that share one head's K / V meet in ONE XCD's L2 (with query blocks on x every XCD
    // streamed every head's K / V: tools/probe/attn_probe.hip)
    int bhid, qblk;
    attn_block_map(bhid, qblk);
    if (q_lens && qblk * 128 >= q_lens[(bhid / H) % nbatch_lens]) return;   // (block-uniform, before any barrier)
    constexpr int RB = 64 * sizeof(T);          // bytes per 64-element row (128 / 256)
    constexpr int RS = RB + 16;                 // padded LDS row stride
    constexpr int NF = RB / 64;                 // 16-byte fragments per lane per 64-element row (2 / 4)
    constexpr int EPC = 16 / sizeof(T);
    constexpr int CH = 64 * (RB / 16) / 256;    // 16-byte chunks per thread per tile (2 / 4)
    constexpr int TILE = 64 * RS;
    constexpr int BUF = 2 * TILE;
    constexpr float L2E = 1.4426950408889634f;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, g = lane >> 4;
    const int h = bhid % H, b = bhid / H;
    const size_t bh = (size_t)b * H + h;
    const int q0 = qblk * 128 + wave * 32;
    int kv_len = N;
    if (kv_lens) kv_len = min(N, kv_lens[b % nbatch_lens]);
    const int nkt = (kv_len + 63) / 64;

    // Q fragments stay in registers for the whole kernel
    u32x4 qf[2][NF];
#pragma unroll
    for (int qs = 0; qs < 2; ++qs) {
        const int q = q0 + qs * 16 + l15;
#pragma unroll
        for (int f = 0; f < NF; ++f)
            qf[qs][f] = q < N ? *reinterpret_cast<const u32x4*>(reinterpret_cast<const char*>(Q + (bh * N + q) * 64) +
                                                                f * 64 + g * 16)
                              : u32x4{0u, 0u, 0u, 0u};
    }

    u32x4 rk[CH], rv[CH];
    auto gload = [&](int kt) {
#pragma unroll
        for (int i = 0; i < CH; ++i) {
            const int c = tid + i * 256, row = c / (RB / 16), cc = c % (RB / 16);
            // rows past the sequence read the last row (masked below: kv_len <= N).  A conditional load here makes hipcc's
            // wait-count pass put vmcnt waits BETWEEN the loads of a tile (pessimistic merge at the exec-masked branches):
            // the wave then sits out a memory round trip per tile before its first MFMA (tools/probe/attn_probe.hip: 12 of 47 us).
            const int key = min(kt * 64 + row, N - 1);
            rk[i] = *reinterpret_cast<const u32x4*>(K + (bh * N + key) * 64 + cc * EPC);
            rv[i] = *reinterpret_cast<const u32x4*>(Vt + (bh * 64 + row) * Npad + kt * 64 + cc * EPC);
        }
    };
    auto sstore = [&](int buf) {
        char* base = smem + buf * BUF;
#pragma unroll
        for (int i = 0; i < CH; ++i) {
            const int c = tid + i * 256, row = c / (RB / 16), cc = c % (RB / 16);
            *reinterpret_cast<u32x4*>(base + row * RS + cc * 16) = rk[i];
            *reinterpret_cast<u32x4*>(base + TILE + row * RS + cc * 16) = rv[i];
        }
    };

    f32x4 o[4][2];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
#pragma unroll
        for (int qs = 0; qs < 2; ++qs) o[dt][qs] = f32x4{0.f, 0.f, 0.f, 0.f};
    float mrun[2] = {-1e30f, -1e30f}, lrun[2] = {0.f, 0.f};

    gload(0);
    sstore(0);
    __syncthreads();
    for (int kt = 0; kt < nkt; ++kt) {
        if (kt + 1 < nkt) gload(kt + 1);
        const char* Ks = smem + (kt & 1) * BUF + l15 * RS + g * 16;
        const char* Vs = smem + (kt & 1) * BUF + TILE + l15 * RS;

        // ---- S^T = K Q^T
        f32x4 s[4][2];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int qs = 0; qs < 2; ++qs) s[ks][qs] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int f = 0; f < NF; ++f) {
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const u32x4 kf = *reinterpret_cast<const u32x4*>(Ks + ks * 16 * RS + f * 64);
#pragma unroll
                for (int qs = 0; qs < 2; ++qs) s[ks][qs] = Mma<T>::run(kf, qf[qs][f], s[ks][qs]);
            }
        }

        // ---- online softmax (per q column = per lane)
        const int key_base = kt * 64 + g * 4;
        const bool edge = (kt * 64 + 64 > kv_len);
#pragma unroll
        for (int qs = 0; qs < 2; ++qs) {
            float mloc = -1e30f;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (edge && key_base + ks * 16 + r >= kv_len) s[ks][qs][r] = -1e30f;
                    mloc = fmaxf(mloc, s[ks][qs][r]);
                }
            mloc = fmaxf(mloc, __shfl_xor(mloc, 16, 64));
            mloc = fmaxf(mloc, __shfl_xor(mloc, 32, 64));
            const float mnew = fmaxf(mrun[qs], mloc);
            const float alpha = exp2f((mrun[qs] - mnew) * L2E);
            mrun[qs] = mnew;
            const float mb = mnew * L2E;
            float psum = 0.f;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float p = exp2f(s[ks][qs][r] * L2E - mb);
                    if (edge && key_base + ks * 16 + r >= kv_len) p = 0.f;
                    s[ks][qs][r] = p;
                    psum += p;
                }
            lrun[qs] = lrun[qs] * alpha + psum;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                o[dt][qs][0] *= alpha; o[dt][qs][1] *= alpha; o[dt][qs][2] *= alpha; o[dt][qs][3] *= alpha;
            }
        }

        // ---- O^T += V^T P^T
        if constexpr (sizeof(T) == 2) {
#pragma unroll
            for (int kp = 0; kp < 2; ++kp) {
                u32x4 pf[2];
#pragma unroll
                for (int qs = 0; qs < 2; ++qs) {
                    pf[qs] = pack8<T>(s[2 * kp][qs][0], s[2 * kp][qs][1], s[2 * kp][qs][2], s[2 * kp][qs][3],
                                      s[2 * kp + 1][qs][0], s[2 * kp + 1][qs][1], s[2 * kp + 1][qs][2], s[2 * kp + 1][qs][3]);
                }
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) {
                    const char* vrow = Vs + dt * 16 * RS + (32 * kp + 4 * g) * 2;
                    const u32x2 lo = *reinterpret_cast<const u32x2*>(vrow);
                    const u32x2 hi = *reinterpret_cast<const u32x2*>(vrow + 32);
                    const u32x4 vf = u32x4{lo.x, lo.y, hi.x, hi.y};
#pragma unroll
                    for (int qs = 0; qs < 2; ++qs) o[dt][qs] = Mma<T>::run(vf, pf[qs], o[dt][qs]);
                }
            }
        } else {
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) {
                    const u32x4 vf = *reinterpret_cast<const u32x4*>(Vs + dt * 16 * RS + (16 * ks + 4 * g) * 4);
#pragma unroll
                    for (int qs = 0; qs < 2; ++qs)
                        o[dt][qs] = Mma<T>::run(vf, __builtin_bit_cast(u32x4, s[ks][qs]), o[dt][qs]);
                }
            }
        }

        if (kt + 1 < nkt) sstore((kt + 1) & 1);
        __syncthreads();
    }

    // ---- normalise and store O[b, q, h*64 + dh]
#pragma unroll
    for (int qs = 0; qs < 2; ++qs) {
        float l = lrun[qs];
        l += __shfl_xor(l, 16, 64);
        l += __shfl_xor(l, 32, 64);
        const float inv = 1.0f / l;
        const int q = q0 + qs * 16 + l15;
        // (o_row_start: output rows of a packed variable-length batch, RowPack: batch row b owns rows o_row_start[b] ..)
        const size_t orow = o_row_start ? (size_t)o_row_start[b] + q : (size_t)b * N + q;
        if (q < (o_row_start ? min(N, o_row_start[b + 1] - o_row_start[b]) : N)) {
            T* dst = O + orow * (H * 64) + h * 64 + g * 4;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt)
                store4(dst + dt * 16, o[dt][qs][0] * inv, o[dt][qs][1] * inv, o[dt][qs][2] * inv, o[dt][qs][3] * inv);
        }
    }
}

template <typename T>
inline hipError_t launch_attention(hipStream_t s, const T* Q, const T* K, const T* Vt, T* O, int Bp, int H, int N,
                                   int Npad, const int* kv_lens, int nbatch_lens, const int* q_lens = nullptr,
                                   const int* o_row_start = nullptr) {
    constexpr int smem = 2 * 2 * 64 * (64 * (int)sizeof(T) + 16);
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_fwd_kernel<T>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, smem);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    dim3 grid(H * Bp, (N + 127) / 128);
    hipLaunchKernelGGL((attn_fwd_kernel<T>), grid, dim3(256), smem, s, Q, K, Vt, O, H, N, Npad, kv_lens, nbatch_lens, q_lens, o_row_start);
    return hipGetLastError();
}


// ---- F5_PREC_F16X3: f32 in / f32 out, both products on the f16 matrix pipe with split operands ----------------------------
// Same transposed formulation as above with the 16-bit fragment layouts (32-deep f16 MFMA steps).  Every operand x is used as
// x_hi + x_lo (x_hi = f16(x), x_lo = f16(x - x_hi); the matrix pipe keeps f16 subnormals) and every product as
// hi*hi + lo*hi + hi*lo accumulated in f32:
//   Q   split once per block into registers (the same 32 registers the f32 fragments took);
//   K,V split by the thread that stages them: global f32 chunk -> two 8-byte LDS writes into an f16 hi plane and an f16 lo plane
//       (the four planes of a tile take the bytes of the two f32 images they replace);
//   P   (the softmax numerators, f32 accumulators) split in registers into the B operand of V^T P^T.
// 6 + 6 sixteen-cycle MFMAs per (16 keys x 16 queries x 64) instead of 16 + 16 thirty-two-cycle f32 ones.
__device__ __forceinline__ f32x4 mma_h(const u32x4& a, const u32x4& b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}

// DIAG (tools/attn_probe.hip only; 0 in the product): bit 0 = stage K / V without splitting, 1 = no softmax arithmetic,
// 2 = no V^T P^T MFMAs, 3 = no K Q^T MFMAs, 4 = no global loads after the first tile -- the cost of each part by omission.
// HI (diagnostic F5_X3_ABLATE, tools/x3_ablate.py): bit 0 = K Q^T with the hi x hi product only (plain f16), bit 1 = V^T P^T likewise.
template <int QS, int NW = 4, int DIAG = 0, int HI = 0>
static __global__ __launch_bounds__(NW * 64) void attn_split_fwd_kernel(const float* __restrict__ Q, const float* __restrict__ K,
                                                             const float* __restrict__ Vt, float* __restrict__ O, int H, int N,
                                                             int Npad, const int* __restrict__ kv_lens, int nbatch_lens,
                                                             const int* __restrict__ q_lens, const int* __restrict__ o_row_start,
                                                             int o_planar) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int bhid, qblk;
    attn_block_map(bhid, qblk);
    if (q_lens && qblk * (NW * 16 * QS) >= q_lens[(bhid / H) % nbatch_lens]) return;   // (grid as attn_fwd_kernel)
    constexpr int RS = 128 + 16;                // f16 plane row: 64 elements + pad
    constexpr int PLANE = 64 * RS;
    constexpr int BUF = 4 * PLANE;              // K hi, K lo, V hi, V lo
    constexpr float L2E = 1.4426950408889634f;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, g = lane >> 4;
    const int h = bhid % H, b = bhid / H;
    const size_t bh = (size_t)b * H + h;
    const int q0 = qblk * (NW * 16 * QS) + wave * (16 * QS);
    int kv_len = N;
    if (kv_lens) kv_len = min(N, kv_lens[b % nbatch_lens]);
    const int nkt = (kv_len + 63) / 64;

    // Q fragments: step f (32 dims) wants dims f*32 + g*8 .. +7 of row q
    u32x4 qh[QS][2], ql[QS][2];
#pragma unroll
    for (int qs = 0; qs < QS; ++qs) {
        const int q = q0 + qs * 16 + l15;
#pragma unroll
        for (int f = 0; f < 2; ++f) {
            u32x4 c0{0u, 0u, 0u, 0u}, c1{0u, 0u, 0u, 0u};
            if (q < N) {
                const float* src = Q + (bh * N + q) * 64 + f * 32 + g * 8;
                c0 = *reinterpret_cast<const u32x4*>(src);
                c1 = *reinterpret_cast<const u32x4*>(src + 4);
            }
            u32x2 h0, l0, h1, l1;
            split4_f16(c0, h0, l0);
            split4_f16(c1, h1, l1);
            qh[qs][f] = u32x4{h0.x, h0.y, h1.x, h1.y};
            ql[qs][f] = u32x4{l0.x, l0.y, l1.x, l1.y};
        }
    }

    // staging: 64 rows x 16 four-float chunks per operand per tile = CH chunks per thread
    constexpr int CH = 1024 / (NW * 64);
    u32x4 rk[CH], rv[CH];
    auto gload = [&](int kt) {
#pragma unroll
        for (int i = 0; i < CH; ++i) {
            const int c = tid + i * (NW * 64), row = c >> 4, cc = c & 15;
            const int key = min(kt * 64 + row, N - 1);      // (unconditional: see attn_fwd_kernel)
            rk[i] = *reinterpret_cast<const u32x4*>(K + (bh * N + key) * 64 + cc * 4);
            rv[i] = *reinterpret_cast<const u32x4*>(Vt + (bh * 64 + row) * Npad + kt * 64 + cc * 4);
        }
    };
    auto sstore = [&](int buf) {
        char* base = smem + buf * BUF;
#pragma unroll
        for (int i = 0; i < CH; ++i) {
            const int c = tid + i * (NW * 64), row = c >> 4, cc = c & 15;
            u32x2 hi, lo;
            if constexpr (DIAG & 1) {
                *reinterpret_cast<u32x2*>(base + row * RS + cc * 8) = u32x2{rk[i].x, rk[i].y};
                *reinterpret_cast<u32x2*>(base + PLANE + row * RS + cc * 8) = u32x2{rk[i].z, rk[i].w};
                *reinterpret_cast<u32x2*>(base + 2 * PLANE + row * RS + cc * 8) = u32x2{rv[i].x, rv[i].y};
                *reinterpret_cast<u32x2*>(base + 3 * PLANE + row * RS + cc * 8) = u32x2{rv[i].z, rv[i].w};
                continue;
            }
            split4_f16(rk[i], hi, lo);
            *reinterpret_cast<u32x2*>(base + row * RS + cc * 8) = hi;
            *reinterpret_cast<u32x2*>(base + PLANE + row * RS + cc * 8) = lo;
            split4_f16(rv[i], hi, lo);
            *reinterpret_cast<u32x2*>(base + 2 * PLANE + row * RS + cc * 8) = hi;
            *reinterpret_cast<u32x2*>(base + 3 * PLANE + row * RS + cc * 8) = lo;
        }
    };

    f32x4 o[4][QS];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
#pragma unroll
        for (int qs = 0; qs < QS; ++qs) o[dt][qs] = f32x4{0.f, 0.f, 0.f, 0.f};
    float mrun[QS], lrun[QS];
#pragma unroll
    for (int qs = 0; qs < QS; ++qs) { mrun[qs] = -1e30f; lrun[qs] = 0.f; }

    gload(0);
    sstore(0);
    __syncthreads();
    for (int kt = 0; kt < nkt; ++kt) {
        if (kt + 1 < nkt && !(DIAG & 16)) gload(kt + 1);
        const char* Ks = smem + (kt & 1) * BUF + l15 * RS + g * 16;
        const char* Vs = smem + (kt & 1) * BUF + 2 * PLANE + l15 * RS;

        // ---- S^T = K Q^T
        f32x4 s[4][QS];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int qs = 0; qs < QS; ++qs) s[ks][qs] = f32x4{0.f, 0.f, 0.f, 0.f};
        // (term-major over the 8 accumulators of a step: a dependent MFMA would wait out its predecessor's whole pipeline)
#pragma unroll
        for (int f = 0; f < 2; ++f) {
            u32x4 kh[4], kl[4];
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                kh[ks] = *reinterpret_cast<const u32x4*>(Ks + ks * 16 * RS + f * 64);
                kl[ks] = *reinterpret_cast<const u32x4*>(Ks + PLANE + ks * 16 * RS + f * 64);
            }
#pragma unroll
            for (int term = (HI & 1) ? 2 : 0; term < ((DIAG & 8) ? 0 : 3); ++term)
#pragma unroll
                for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                    for (int qs = 0; qs < QS; ++qs)
                        s[ks][qs] = mma_h(term == 0 ? kl[ks] : kh[ks], term == 1 ? ql[qs][f] : qh[qs][f], s[ks][qs]);
        }

        // ---- online softmax (per q column = per lane)
        // (the tile that straddles kv_len -- the last one -- masks its tail in a branch of its own: masked scores sit 1e30 below
        //  the row maximum, which a valid key of the same tile sets, so their numerators come out as exact zeros;
        //  v_exp_f32 directly: every argument is <= 0, a flushed denormal numerator is a zero that does not matter)
        const int key_base = kt * 64 + g * 4;
        if (kt * 64 + 64 > kv_len) {
#pragma unroll
            for (int qs = 0; qs < QS; ++qs)
#pragma unroll
                for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (key_base + ks * 16 + r >= kv_len) s[ks][qs][r] = -1e30f;
        }
#pragma unroll
        for (int qs = 0; qs < ((DIAG & 2) ? 0 : QS); ++qs) {
            float mloc = -1e30f;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                for (int r = 0; r < 4; ++r) mloc = fmaxf(mloc, s[ks][qs][r]);
            mloc = fmaxf(mloc, __shfl_xor(mloc, 16, 64));
            mloc = fmaxf(mloc, __shfl_xor(mloc, 32, 64));
            const float mnew = fmaxf(mrun[qs], mloc);
            const float alpha = __builtin_amdgcn_exp2f((mrun[qs] - mnew) * L2E);
            mrun[qs] = mnew;
            const float mb = mnew * L2E;
            float psum = 0.f;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float p = __builtin_amdgcn_exp2f(s[ks][qs][r] * L2E - mb);
                    s[ks][qs][r] = p;
                    psum += p;
                }
            lrun[qs] = lrun[qs] * alpha + psum;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                o[dt][qs][0] *= alpha; o[dt][qs][1] *= alpha; o[dt][qs][2] *= alpha; o[dt][qs][3] *= alpha;
            }
        }

        // ---- O^T += V^T P^T : the 8 k-slots of a 32-key step = keys {4g..4g+3} of two adjacent 16-key sub-tiles
#pragma unroll
        for (int kp = 0; kp < 2; ++kp) {
            u32x4 ph[QS], pl[QS];
#pragma unroll
            for (int qs = 0; qs < QS; ++qs) {
                u32x2 h0, l0, h1, l1;
                split4_f16(__builtin_bit_cast(u32x4, s[2 * kp][qs]), h0, l0);
                split4_f16(__builtin_bit_cast(u32x4, s[2 * kp + 1][qs]), h1, l1);
                ph[qs] = u32x4{h0.x, h0.y, h1.x, h1.y};
                pl[qs] = u32x4{l0.x, l0.y, l1.x, l1.y};
            }
            u32x4 vh[4], vl[4];
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                const char* vrow = Vs + dt * 16 * RS + (32 * kp + 4 * g) * 2;
                const u32x2 a0 = *reinterpret_cast<const u32x2*>(vrow), a1 = *reinterpret_cast<const u32x2*>(vrow + 32);
                const u32x2 b0 = *reinterpret_cast<const u32x2*>(vrow + PLANE), b1 = *reinterpret_cast<const u32x2*>(vrow + PLANE + 32);
                vh[dt] = u32x4{a0.x, a0.y, a1.x, a1.y};
                vl[dt] = u32x4{b0.x, b0.y, b1.x, b1.y};
            }
#pragma unroll
            for (int term = (HI & 2) ? 2 : 0; term < ((DIAG & 4) ? 0 : 3); ++term)
#pragma unroll
                for (int dt = 0; dt < 4; ++dt)
#pragma unroll
                    for (int qs = 0; qs < QS; ++qs)
                        o[dt][qs] = mma_h(term == 0 ? vl[dt] : vh[dt], term == 1 ? pl[qs] : ph[qs], o[dt][qs]);
        }

        if (kt + 1 < nkt) sstore((kt + 1) & 1);
        __syncthreads();
    }

#pragma unroll
    for (int qs = 0; qs < QS; ++qs) {
        float l = lrun[qs];
        l += __shfl_xor(l, 16, 64);
        l += __shfl_xor(l, 32, 64);
        const float inv = 1.0f / l;
        const int q = q0 + qs * 16 + l15;
        const size_t orow = o_row_start ? (size_t)o_row_start[b] + q : (size_t)b * N + q;
        if (q < (o_row_start ? min(N, o_row_start[b + 1] - o_row_start[b]) : N)) {
            // (o_planar: the out-projection reads this buffer as a pre-split A operand, f5_common.h store4_planar)
            float* dst = O + orow * (H * 64);
#pragma unroll
            for (int dt = 0; dt < 4; ++dt)
                store4_at(dst, h * 64 + g * 4 + dt * 16, o_planar, o[dt][qs][0] * inv, o[dt][qs][1] * inv, o[dt][qs][2] * inv,
                          o[dt][qs][3] * inv);
        }
    }
}

template <int HI>
inline hipError_t launch_attention_split_hi(hipStream_t s, const float* Q, const float* K, const float* Vt, float* O, int Bp, int H, int N,
                                            int Npad, const int* kv_lens, int nbatch_lens, const int* q_lens, const int* o_row_start,
                                            int o_planar) {
    // 128 query rows per workgroup as 8 waves x 16 rows: two waves per SIMD (one's softmax / split arithmetic overlaps the other's
    // MFMAs) sharing one staged K / V tile.  (4 waves x 32 rows: one wave per SIMD, every phase serial: 44 us at C2 against 39;
    // 4 waves x 16 rows: twice the staging per query, 47 us -- tools/probe/attn_probe.hip)
    constexpr int smem = 2 * 4 * 64 * (128 + 16);
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_split_fwd_kernel<1, 8, 0, HI>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, smem);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    dim3 grid(H * Bp, (N + 127) / 128);
    hipLaunchKernelGGL((attn_split_fwd_kernel<1, 8, 0, HI>), grid, dim3(512), smem, s, Q, K, Vt, O, H, N, Npad, kv_lens, nbatch_lens, q_lens,
                       o_row_start, o_planar);
    return hipGetLastError();
}
inline hipError_t launch_attention_split(hipStream_t s, const float* Q, const float* K, const float* Vt, float* O, int Bp, int H, int N,
                                         int Npad, const int* kv_lens, int nbatch_lens, const int* q_lens = nullptr,
                                         const int* o_row_start = nullptr, int o_planar = 0, int hi_only = 0) {
    switch (hi_only & 3) {   // (1..3: diagnostic F5_X3_ABLATE)
        case 1: return launch_attention_split_hi<1>(s, Q, K, Vt, O, Bp, H, N, Npad, kv_lens, nbatch_lens, q_lens, o_row_start, o_planar);
        case 2: return launch_attention_split_hi<2>(s, Q, K, Vt, O, Bp, H, N, Npad, kv_lens, nbatch_lens, q_lens, o_row_start, o_planar);
        case 3: return launch_attention_split_hi<3>(s, Q, K, Vt, O, Bp, H, N, Npad, kv_lens, nbatch_lens, q_lens, o_row_start, o_planar);
        default: return launch_attention_split_hi<0>(s, Q, K, Vt, O, Bp, H, N, Npad, kv_lens, nbatch_lens, q_lens, o_row_start, o_planar);
    }
}

}  // namespace f5
