// BigVGAN v2 generator (mel -> waveform) on gfx950 -- the vocoder of BASELINE config 5 / SURVEY a20, which the reference
// loads from an un-vendored submodule (infer/utils_infer.py:138-152, called as `vocoder(mel)` at :705; forced to fp32,
// :332).  PARITY UNPINNED: neither BigVGAN's source nor its weights are in the container; this restates the published
// architecture of nvidia/bigvgan_v2_24khz_100band_256x (tests/test_bigvgan.py checks the kernels against a CPU
// restatement of the same published design).  All f32, time-major [L, C] activations (no [B, C, T] permutes), every contraction an exact-f32 MFMA GEMM:
//   conv_pre            Conv1d(mels, C0, 7)            = im2col (strided mel view) x W[C0, 7 mels]
//   ups[i]              ConvTranspose1d(C, C/2, k, u)  = GEMM x[L, C] x W'[(tap, co), C]^T -> Z[L, k C/2], then each output
//                                                        sample gathers its k / u taps from Z (+ bias)
//   AMPBlock1 convs     Conv1d(C, C, k, dilation d)    = dilated im2col x W[C, k C] with bias (+ residual) epilogues
//   Activation1d        2x kaiser-sinc upsample -> SnakeBeta -> 2x low-pass downsample, ONE kernel: an output sample needs the
//                       12 upsampled samples around it, each a 6-tap sum over the input rows t-5 .. t+5 held in registers
//   conv_post           Conv1d(C_last, 1, 7) + clamp / tanh, direct (168 MACs per sample)
#include <cmath>
#include <map>
#include <string>
#include <vector>

#include "elementwise.h"
#include "gemm_dispatch.h"
#include "internal.h"

using namespace f5;
#define fail f5_fail

// ------------------------------------------------------------------------------------------------ kernels
// A[t][tap * C + c] = x[t + (tap - (k-1)/2) * dil][c]  (zero outside [0, L); columns >= k C zero: K padding of the GEMM)
static __global__ void bv_im2col_kernel(const float* __restrict__ x, float* __restrict__ A, long L, int C, int k, int dil, int ld) {
    const int l4 = ld / 4;
    const long total = L * l4;
    const int half = (k - 1) / 2;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int col = (int)(i % l4) * 4;
        const long t = i / l4;
        float4 v = make_float4(0, 0, 0, 0);
        if (col < k * C) {
            const int tap = col / C, c = col - tap * C;
            const long tt = t + (long)(tap - half) * dil;
            if (tt >= 0 && tt < L) v = *reinterpret_cast<const float4*>(x + tt * C + c);
        }
        reinterpret_cast<float4*>(A)[i] = v;
    }
}

// sin for the snake activation: three-constant Cody-Waite reduction by pi/2 (fma: |x| up to ~1e6 keeps the absolute error
// at the f32 level) + degree-7 / degree-8 minimax polynomials on [-pi/4, pi/4].  Measured against float64 sin over
// |x| <= 200: max |error| 9.2e-8 (libm's f32 sinf: 6.9e-8).  The library sinf inlines a Payne-Hanek path at each of the
// activation kernel's call sites (17,000 instructions, 284 registers, one wave per SIMD); this is ~20 instructions.
__device__ __forceinline__ float bv_sin(float x) {
    const float k = rintf(x * 0.63661977236758134f);
    float r = fmaf(-k, 1.57079637050628662109375f, x);
    r = fmaf(-k, -4.371138828673792886547744274139404296875e-8f, r);
    r = fmaf(-k, -1.7151245100058818728e-15f, r);
    const int q = (int)k;
    const float r2 = r * r;
    const float ps = fmaf(fmaf(fmaf(-1.9515295891e-4f, r2, 8.3321608736e-3f), r2, -1.6666654611e-1f) * r2, r, r);
    const float pc = fmaf(fmaf(fmaf(2.443315711809948e-5f, r2, -1.388731625493765e-3f), r2, 4.166664568298827e-2f), r2 * r2,
                          fmaf(-0.5f, r2, 1.0f));
    const float res = (q & 1) ? pc : ps;
    return (q & 2) ? -res : res;
}

// precision "f16x3": the same reduction to [-pi, pi] by one multiple of 2 pi, then the hardware v_sin_f32 (argument in revolutions):
// 9 issue slots instead of ~20; max |error| 3.5e-7 over the reduced range (tools/probe/sin_probe.hip) against 9.2e-8.
__device__ __forceinline__ float bv_sin_hw(float x) {
    const float k = rintf(x * 0.15915494309189535f);
    float r = fmaf(-k, 6.283185482025146484375f, x);
    r = fmaf(-k, -1.7484555314695172e-7f, r);
    return __builtin_amdgcn_sinf(r * 0.15915494309189535f);
}
template <bool HW> __device__ __forceinline__ float bv_snake(float v, float a, float invb) {
    const float s = HW ? bv_sin_hw(v * a) : bv_sin(v * a);
    return v + invb * (s * s);
}

// Activation1d(SnakeBeta) (alias-free activation): y[t] = sum_m fd[m] a[clamp(2t + m - 5, 0, 2L-1)],
// a[u] = snake(2 sum_{j = (u+1) mod 2 + 2q} fu[j] x[clamp((u + 15 - j) / 2 - 5, 0, L-1)])
// One thread = BV_TT consecutive time steps x 4 channels, streamed: the 12 upsampled samples a[2t-5 .. 2t+6] that output t
// reads live in registers and slide by two per step; the two new ones (u = 2t+5, 2t+6) both read input rows t .. t+5, a
// six-row register window that slides by one.  Every upsampled sample (one sinf) is so evaluated once per thread instead of
// once per output that reads it: (10 + 2 BV_TT) / BV_TT evaluations per output instead of 12.
constexpr int BV_TT = 8;

struct BvSnake { float a0, a1, a2, a3, b0, b1, b2, b3; };

template <bool HW> __device__ __forceinline__ float4 bv_snake4(const float4 u, const BvSnake& p) {
    return make_float4(bv_snake<HW>(2.0f * u.x, p.a0, p.b0), bv_snake<HW>(2.0f * u.y, p.a1, p.b1), bv_snake<HW>(2.0f * u.z, p.a2, p.b2),
                       bv_snake<HW>(2.0f * u.w, p.a3, p.b3));
}

// a[clamp(u)] from global rows (tile start and sequence edges)
template <bool HW>
__device__ __forceinline__ float4 bv_up_generic(const float* __restrict__ x, long u, long L, int C, int c, const float* sfu,
                                                const BvSnake& p) {
    u = u < 0 ? 0 : (u > 2 * L - 1 ? 2 * L - 1 : u);
    float4 s = make_float4(0, 0, 0, 0);
    for (int j = (int)((u + 1) & 1); j < 12; j += 2) {
        long tt = (u + 15 - j) / 2 - 5;
        tt = tt < 0 ? 0 : (tt > L - 1 ? L - 1 : tt);
        const float f = sfu[j];
        const float4 v = *reinterpret_cast<const float4*>(x + tt * C + c);
        s.x += f * v.x; s.y += f * v.y; s.z += f * v.z; s.w += f * v.w;
    }
    return bv_snake4<HW>(s, p);
}

template <bool HW>
static __global__ __launch_bounds__(256) void bv_act_kernel(const float* __restrict__ x, float* __restrict__ y, long L, int C,
                                                             const float* __restrict__ log_alpha, const float* __restrict__ log_beta,
                                                             const float* __restrict__ fu, const float* __restrict__ fd, int planar) {
    // planar: rows written pre-split for the split-operand convolution GEMM that reads them (f5_common.h store4_planar)
    __shared__ float sfu[12], sfd[12];
    if (threadIdx.x < 12) { sfu[threadIdx.x] = fu[threadIdx.x]; sfd[threadIdx.x] = fd[threadIdx.x]; }
    __syncthreads();
    const int c4n = C / 4;
    const long tiles = (L + BV_TT - 1) / BV_TT;
    const long total = tiles * c4n;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % c4n) * 4;
        const long t0 = (i / c4n) * BV_TT;
        const float4 la = *reinterpret_cast<const float4*>(log_alpha + c), lb = *reinterpret_cast<const float4*>(log_beta + c);
        const BvSnake p{expf(la.x), expf(la.y), expf(la.z), expf(la.w), 1.0f / (expf(lb.x) + 1e-9f), 1.0f / (expf(lb.y) + 1e-9f),
                        1.0f / (expf(lb.z) + 1e-9f), 1.0f / (expf(lb.w) + 1e-9f)};
        auto row = [&](long tt) {
            tt = tt < 0 ? 0 : (tt > L - 1 ? L - 1 : tt);
            return *reinterpret_cast<const float4*>(x + tt * C + c);
        };
        float4 a[12], r[6];
#pragma unroll
        for (int m = 0; m < 10; ++m) a[m] = bv_up_generic<HW>(x, 2 * t0 + m - 5, L, C, c, sfu, p);
#pragma unroll
        for (int q = 0; q < 6; ++q) r[q] = row(t0 + q);
#pragma unroll 1
        for (int k = 0; k < BV_TT; ++k) {
            const long t = t0 + k;
            if (t < L) {
                const float4 nxt = row(t + 6);                     // (issued early: consumed at the end of the step)
                // u = 2t+5 (odd): taps j = 2q on rows t+5-q;  u = 2t+6 (even): taps j = 2q+1 on the same rows
                float4 s0 = make_float4(0, 0, 0, 0), s1 = make_float4(0, 0, 0, 0);
#pragma unroll
                for (int q = 0; q < 6; ++q) {
                    const float f0 = sfu[2 * q], f1 = sfu[2 * q + 1];
                    const float4 v = r[5 - q];
                    s0.x += f0 * v.x; s0.y += f0 * v.y; s0.z += f0 * v.z; s0.w += f0 * v.w;
                    s1.x += f1 * v.x; s1.y += f1 * v.y; s1.z += f1 * v.z; s1.w += f1 * v.w;
                }
                // past the end a[u] repeats a[2L-1] (replicate padding of the low-pass filter)
                a[10] = 2 * t + 5 <= 2 * L - 1 ? bv_snake4<HW>(s0, p) : a[9];
                a[11] = 2 * t + 6 <= 2 * L - 1 ? bv_snake4<HW>(s1, p) : a[10];
                float4 acc = make_float4(0, 0, 0, 0);
#pragma unroll
                for (int m = 0; m < 12; ++m) {
                    const float g = sfd[m];
                    acc.x += g * a[m].x; acc.y += g * a[m].y; acc.z += g * a[m].z; acc.w += g * a[m].w;
                }
                store4_at(y + t * C, c, planar, acc.x, acc.y, acc.z, acc.w);
#pragma unroll
                for (int m = 0; m < 10; ++m) a[m] = a[m + 2];
#pragma unroll
                for (int q = 0; q < 5; ++q) r[q] = r[q + 1];
                r[5] = nxt;
            }
        }
    }
}

// Conv1d(C, C, k, dilation) for the NARROW last stages (C < 64: 48 and 24 channels in the 24 kHz config, where a GEMM tile
// would be mostly padding and the im2col operand ~200 MB).  One workgroup = 4 waves = 64 MT time steps x all channels:
//   - the input rows [t0 - halo, t0 + TB + halo) are staged once in LDS (row stride C + 1 words: the 16 rows x 2 columns a
//     half-wave reads at once fall in distinct banks), zero outside [0, L): every tap re-reads them shifted, no im2col;
//   - v_mfma_f32_16x16x4_f32 with the WEIGHT as first operand: lane (lr = lane % 16, lk = lane / 16) feeds W[n = 16 nt + lr]
//     [K = 4 kk + lk] and x[t = m0 + lr + shift(tap)][ci = 4 cs + lk], and ends with out[t = m0 + lr][n = 16 nt + 4 lk .. + 3]
//     -- a float4 of channels per lane for the bias / residual / store;
//   - the weights are pre-packed [K / 4][NT][64 lanes] (bv_pack_narrow_kernel) so a fragment is one coalesced 256-byte
//     load, requested one k-step ahead of its use (all workgroups read the same <= 100 KB: L1/L2 hits).
typedef float bv_f32x4 __attribute__((ext_vector_type(4)));

static __global__ void bv_pack_narrow_kernel(const float* __restrict__ w /*[Co][ld] tap-major*/, float* __restrict__ wn, int Co, int ld,
                                             int K, int NT) {
    const long total = (long)(K / 4) * NT * 64;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int lane = (int)(i & 63), lr = lane & 15, lk = lane >> 4;
        const long f = i >> 6;
        const int nt = (int)(f % NT), kk = (int)(f / NT);
        const int n = nt * 16 + lr;
        wn[i] = n < Co ? w[(long)n * ld + kk * 4 + lk] : 0.0f;
    }
}

template <int NT, int MT>
static __global__ __launch_bounds__(256) void bv_conv_narrow_kernel(const float* __restrict__ x, const float* __restrict__ wn,
                                                                     const float* __restrict__ bias, const float* res, float* out,
                                                                     long L, int C, int k, int dil) {
    extern __shared__ float bv_xs[];
    constexpr int TB = 4 * MT * 16;
    const int half = (k - 1) / 2, halo = half * dil, S = C + 1, rows = TB + 2 * halo, c4n = C / 4;
    const long t0 = (long)blockIdx.x * TB;
    // (four loads in flight per thread, from clamped addresses, before their stores: a load-if-in-range / store loop compiles to
    //  one serialised memory round trip per iteration -- convpos.h)
    for (int base = threadIdx.x; base < rows * c4n; base += 4 * 256) {
        float4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = min(base + u * 256, rows * c4n - 1), row = i / c4n, col = (i - row * c4n) * 4;
            const long g = min(max(t0 - halo + row, 0L), L - 1);
            v[u] = *reinterpret_cast<const float4*>(x + g * C + col);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = base + u * 256, row = i / c4n, col = (i - row * c4n) * 4;
            const long g = t0 - halo + row;
            const bool in = g >= 0 && g < L;
            if (i < rows * c4n) {
                float* d = bv_xs + row * S + col;
                d[0] = in ? v[u].x : 0.f; d[1] = in ? v[u].y : 0.f; d[2] = in ? v[u].z : 0.f; d[3] = in ? v[u].w : 0.f;
            }
        }
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, lr = lane & 15, lk = lane >> 4;
    bv_f32x4 acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = bv_f32x4{0, 0, 0, 0};
    const float* wl = wn + lane;
    const float* xl = bv_xs + (wave * MT * 16 + lr + halo) * S + lk;
    const int total = k * c4n;
    float b[NT], bn[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) b[nt] = wl[nt * 64];
    int kk = 0;
    for (int tap = 0; tap < k; ++tap) {
        const float* xt = xl + (tap - half) * dil * S;
        for (int cs = 0; cs < c4n; ++cs, ++kk) {
            const int kn = min(kk + 1, total - 1);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) bn[nt] = wl[(long)(kn * NT + nt) * 64];
            float a[MT];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) a[mt] = xt[mt * 16 * S + cs * 4];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(b[nt], a[mt], acc[mt][nt], 0, 0, 0);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) b[nt] = bn[nt];
        }
    }
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const long t = t0 + (wave * MT + mt) * 16 + lr;
        if (t >= L) continue;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int n = nt * 16 + 4 * lk;
            if (n >= C) continue;
            const float4 bv = *reinterpret_cast<const float4*>(bias + n);
            float4 o = make_float4(acc[mt][nt][0] + bv.x, acc[mt][nt][1] + bv.y, acc[mt][nt][2] + bv.z, acc[mt][nt][3] + bv.w);
            if (res) {
                const float4 rv = *reinterpret_cast<const float4*>(res + t * C + n);
                o.x += rv.x; o.y += rv.y; o.z += rv.z; o.w += rv.w;
            }
            *reinterpret_cast<float4*>(out + t * C + n) = o;
        }
    }
}

constexpr int BV_NARROW_MT = 2;
static hipError_t launch_conv_narrow(hipStream_t s, const float* x, const float* wn, const float* bias, const float* res, float* out,
                                     long L, int C, int k, int dil) {
    constexpr int MT = BV_NARROW_MT, TB = 4 * MT * 16;
    const int NT = (C + 15) / 16;
    const size_t smem = (size_t)(TB + (k - 1) * dil) * (C + 1) * 4;
    const dim3 grid((unsigned)((L + TB - 1) / TB)), block(256);
    switch (NT) {
        case 1: hipLaunchKernelGGL((bv_conv_narrow_kernel<1, MT>), grid, block, smem, s, x, wn, bias, res, out, L, C, k, dil); break;
        case 2: hipLaunchKernelGGL((bv_conv_narrow_kernel<2, MT>), grid, block, smem, s, x, wn, bias, res, out, L, C, k, dil); break;
        case 3: hipLaunchKernelGGL((bv_conv_narrow_kernel<3, MT>), grid, block, smem, s, x, wn, bias, res, out, L, C, k, dil); break;
        case 4: hipLaunchKernelGGL((bv_conv_narrow_kernel<4, MT>), grid, block, smem, s, x, wn, bias, res, out, L, C, k, dil); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}
// the narrow kernel's LDS tile must fit the default 64 KB dynamic limit
static bool conv_narrow_ok(int C, int k, int dil) {
    return C % 4 == 0 && C < 64 && (size_t)(4 * BV_NARROW_MT * 16 + (k - 1) * dil) * (C + 1) * 4 <= 64 * 1024;
}

// ConvTranspose1d tail: y[n][co] = bias[co] + sum_q Z[(n + p - j) / u][j * Co + co], j = (n + p) % u + q u < k, 0 <= (n + p - j) / u < Li
static __global__ void bv_upsample_gather_kernel(const float* __restrict__ Z, const float* __restrict__ bias, float* __restrict__ y,
                                                 long Li, int Co, int k, int u, int p) {
    const int c4n = Co / 4;
    const long Lo = Li * u;
    const long total = Lo * c4n;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % c4n) * 4;
        const long n = i / c4n;
        float4 acc = *reinterpret_cast<const float4*>(bias + c);
        for (int j = (int)((n + p) % u); j < k; j += u) {
            const long ii = (n + p - j) / u;
            if (n + p - j >= 0 && ii < Li) {
                const float4 v = *reinterpret_cast<const float4*>(Z + ii * (long)k * Co + (long)j * Co + c);
                acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
            }
        }
        reinterpret_cast<float4*>(y)[i] = acc;
    }
}
// x = (r0 + r1 + ... ) / n over n buffers spaced `stride` floats apart  (xs / num_kernels)
static __global__ void bv_mean_kernel(const float* __restrict__ r, long stride, int n, float* __restrict__ x, long count4) {
    const float inv = 1.0f / (float)n;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < count4; i += (long)gridDim.x * blockDim.x) {
        float4 a = reinterpret_cast<const float4*>(r)[i];
        for (int j = 1; j < n; ++j) {
            const float4 b = reinterpret_cast<const float4*>(r + j * stride)[i];
            a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
        }
        reinterpret_cast<float4*>(x)[i] = make_float4(a.x * inv, a.y * inv, a.z * inv, a.w * inv);
    }
}
// conv_post: wav[t] = clamp / tanh( bias + sum_{tap, c} w[tap * C + c] a[t + tap - 3][c] )
static __global__ void bv_post_kernel(const float* __restrict__ a, const float* __restrict__ w, const float* __restrict__ bias,
                                      float* __restrict__ wav, long L, int C, int use_tanh) {
    for (long t = blockIdx.x * (long)blockDim.x + threadIdx.x; t < L; t += (long)gridDim.x * blockDim.x) {
        float s = bias ? bias[0] : 0.f;
        for (int tap = 0; tap < 7; ++tap) {
            const long tt = t + tap - 3;
            if (tt < 0 || tt >= L) continue;
            for (int c = 0; c < C; c += 4) {
                const float4 v = *reinterpret_cast<const float4*>(a + tt * C + c);
                const float4 ww = *reinterpret_cast<const float4*>(w + tap * C + c);
                s += v.x * ww.x; s += v.y * ww.y; s += v.z * ww.z; s += v.w * ww.w;
            }
        }
        wav[t] = use_tanh ? tanhf(s) : fminf(fmaxf(s, -1.0f), 1.0f);
    }
}
// ConvTranspose1d weight [Ci][Co][k] (torch) -> GEMM operand W'[(j * Co + co)][ci]
static __global__ void bv_pack_convT_kernel(const float* __restrict__ in, float* __restrict__ out, int Ci, int Co, int k, int ld) {
    const long total = (long)k * Co * ld;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int ci = (int)(i % ld);
        const long r = i / ld;
        const int co = (int)(r % Co), j = (int)(r / Co);
        out[i] = ci < Ci ? in[((long)ci * Co + co) * k + j] : 0.f;
    }
}

// ------------------------------------------------------------------------------------------------ handle
namespace {
struct BT {
    float* p = nullptr;
    std::vector<int64_t> shape;
};
struct BConv { float *w = nullptr, *b = nullptr, *wn = nullptr; int ld = 0; bool split = false; };   // split: w in the split_planar layout (F5_PREC_F16X3)   // wn: bv_pack_narrow_kernel layout (C < 64 only)          // [Co, ld] tap-major, ld = round_up(k C, 32)
struct BAct { float *alpha = nullptr, *beta = nullptr; };
struct BRes { std::vector<BConv> c1, c2; std::vector<BAct> act; int k = 0; };
struct BUp { float *w = nullptr, *b = nullptr; int ci = 0, co = 0, k = 0, u = 0, ld = 0; bool split = false; };
}  // namespace

struct f5_bigvgan {
    f5_bigvgan_config cfg{};
    std::map<std::string, BT> raw;
    std::vector<void*> owned;
    bool finalized = false;
    BConv pre;
    int kpre = 0;
    std::vector<BUp> ups;
    std::vector<BRes> res;
    BAct post_act;
    float *post_w = nullptr, *post_b = nullptr, *fu = nullptr, *fd = nullptr;
    int c_last = 0, total_up = 1;
    Arena arena;
    ~f5_bigvgan() {
        for (auto& kv : raw)
            if (kv.second.p) (void)hipFree(kv.second.p);
        for (void* p : owned) (void)hipFree(p);
    }
};

extern "C" int f5_bigvgan_create(const f5_bigvgan_config* c, f5_bigvgan** out) {
    if (!c || !out) return fail(F5_EINVAL, "f5_bigvgan_create: null argument");
    if (c->num_upsamples <= 0 || c->num_upsamples > 8 || c->num_kernels <= 0 || c->num_kernels > 4 || c->num_dilations <= 0 ||
        c->num_dilations > 4 || c->num_mels % 4 || c->upsample_initial_channel % (4 << c->num_upsamples))
        return fail(F5_EINVAL, "f5_bigvgan_create: unsupported dimensions (channels must stay a multiple of 4 after every halving)");
    for (int i = 0; i < c->num_upsamples; ++i)
        if (c->upsample_rates[i] <= 0 || c->upsample_kernel_sizes[i] % c->upsample_rates[i] || (c->upsample_kernel_sizes[i] - c->upsample_rates[i]) % 2)
            return fail(F5_EINVAL, "f5_bigvgan_create: upsample kernel must be a multiple of its rate and k - u even");
    for (int j = 0; j < c->num_kernels; ++j)
        if (c->resblock_kernel_sizes[j] % 2 == 0) return fail(F5_EINVAL, "f5_bigvgan_create: resblock kernels must be odd");
    if (c->precision != F5_PREC_F32 && c->precision != F5_PREC_F16X3)
        return fail(F5_EINVAL, "f5_bigvgan_create: precision must be F5_PREC_F32 or F5_PREC_F16X3");
    f5_bigvgan* v = new f5_bigvgan();
    v->cfg = *c;
    v->kpre = round_up(7 * c->num_mels, 32);
    v->c_last = c->upsample_initial_channel >> c->num_upsamples;
    for (int i = 0; i < c->num_upsamples; ++i) v->total_up *= c->upsample_rates[i];
    *out = v;
    return F5_OK;
}
extern "C" int f5_bigvgan_destroy(f5_bigvgan* v) {
    if (v) {
        (void)hipDeviceSynchronize();
        delete v;
    }
    return F5_OK;
}
extern "C" int f5_bigvgan_load_weight(f5_bigvgan* v, const char* name, const void* dev, const int64_t* shape, int32_t ndim,
                                      f5_stream stream) {
    if (!v || !name || !dev || ndim < 0 || ndim > 4) return fail(F5_EINVAL, "f5_bigvgan_load_weight: bad arguments");
    BT t;
    t.shape.assign(shape, shape + ndim);
    size_t n = 1;
    for (auto d : t.shape) n *= (size_t)d;
    auto it = v->raw.find(name);
    if (it != v->raw.end()) {
        (void)hipFree(it->second.p);
        v->raw.erase(it);
    }
    HIPCHK(hipMalloc((void**)&t.p, std::max<size_t>(n * 4, 16)));
    HIPCHK(hipMemcpyAsync(t.p, dev, n * 4, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    v->raw[name] = t;
    v->finalized = false;
    return F5_OK;
}

static int bneed(f5_bigvgan* v, const std::string& n, std::vector<int64_t> shape, const BT** out) {
    auto it = v->raw.find(n);
    if (it == v->raw.end()) return fail(F5_ESTATE, "missing bigvgan weight '%s'", n.c_str());
    if (it->second.shape != shape) return fail(F5_EINVAL, "bigvgan weight '%s' has the wrong shape", n.c_str());
    *out = &it->second;
    return F5_OK;
}
static int balloc(f5_bigvgan* v, size_t n, float** out) {
    void* p = nullptr;
    HIPCHK(hipMalloc(&p, std::max<size_t>(n * 4, 16)));
    v->owned.push_back(p);
    *out = (float*)p;
    return F5_OK;
}
static int bcopy(f5_bigvgan* v, hipStream_t s, const std::string& n, std::vector<int64_t> shape, float** out) {
    const BT* t = nullptr;
    CHK(bneed(v, n, shape, &t));
    size_t cnt = 1;
    for (auto d : shape) cnt *= (size_t)d;
    CHK(balloc(v, cnt, out));
    HIPCHK(hipMemcpyAsync(*out, t->p, cnt * 4, hipMemcpyDeviceToDevice, s));
    return F5_OK;
}
// Conv1d weight [Co, Ci, k] (+ bias) -> tap-major GEMM operand [Co, round_up(k Ci, 32)]
static int bconv(f5_bigvgan* v, hipStream_t s, const std::string& pfx, int Co, int Ci, int k, bool bias, BConv* c) {
    const BT* t = nullptr;
    CHK(bneed(v, pfx + ".weight", {Co, Ci, k}, &t));
    c->ld = round_up(k * Ci, 32);
    CHK(balloc(v, (size_t)Co * c->ld, &c->w));
    hipLaunchKernelGGL((conv_pack_kernel<float>), dim3(ew_blocks((long)Co * c->ld)), dim3(256), 0, s, t->p, c->w, (long)Co, Ci, k, c->ld);
    KCHK();
    c->b = nullptr;
    if (bias) CHK(bcopy(v, s, pfx + ".bias", {Co}, &c->b));
    c->wn = nullptr;
    if (Co == Ci && Ci % 4 == 0 && Ci < 64 && bias) {
        const int NT = (Co + 15) / 16, K = k * Ci;
        CHK(balloc(v, (size_t)(K / 4) * NT * 64, &c->wn));
        hipLaunchKernelGGL(bv_pack_narrow_kernel, dim3(ew_blocks((long)(K / 4) * NT * 64)), dim3(256), 0, s, c->w, c->wn, Co, c->ld, K, NT);
        KCHK();
    }
    c->split = v->cfg.precision == F5_PREC_F16X3 && !c->wn;     // (the narrow stages keep f32 MFMA: 7 % of the generator's flops)
    if (c->split) {
        hipLaunchKernelGGL(split_planar_kernel, dim3(ew_blocks((long)Co * c->ld / 32)), dim3(256), 0, s, c->w, (long)Co * c->ld / 32);
        KCHK();
    }
    return F5_OK;
}

static const float* epi_res(const EpiStore<float>&) { return nullptr; }
static float* epi_out(const EpiStore<float>& e) { return e.out; }
static const float* epi_res(const EpiGateRes& e) { return e.res; }
static float* epi_out(const EpiGateRes& e) { return e.x; }

extern "C" int f5_bigvgan_finalize(f5_bigvgan* v, f5_stream stream) {
    if (!v) return fail(F5_EINVAL, "null bigvgan");
    hipStream_t s = (hipStream_t)stream;
    const f5_bigvgan_config& c = v->cfg;
    for (void* p : v->owned) (void)hipFree(p);
    v->owned.clear();
    CHK(bconv(v, s, "conv_pre", c.upsample_initial_channel, c.num_mels, 7, true, &v->pre));
    v->ups.assign(c.num_upsamples, BUp{});
    v->res.assign((size_t)c.num_upsamples * c.num_kernels, BRes{});
    int ch = c.upsample_initial_channel;
    for (int i = 0; i < c.num_upsamples; ++i) {
        BUp& u = v->ups[i];
        u.ci = ch; u.co = ch / 2; u.k = c.upsample_kernel_sizes[i]; u.u = c.upsample_rates[i]; u.ld = round_up(ch, 32);
        const std::string p = "ups." + std::to_string(i) + ".0";
        const BT* t = nullptr;
        CHK(bneed(v, p + ".weight", {u.ci, u.co, u.k}, &t));
        CHK(balloc(v, (size_t)u.k * u.co * u.ld, &u.w));
        hipLaunchKernelGGL(bv_pack_convT_kernel, dim3(ew_blocks((long)u.k * u.co * u.ld)), dim3(256), 0, s, t->p, u.w, u.ci, u.co, u.k, u.ld);
        KCHK();
        u.split = c.precision == F5_PREC_F16X3 && u.ci % 32 == 0;      // (K = ci must be whole K-tiles for the LDS-DMA kernels)
        if (u.split) {
            hipLaunchKernelGGL(split_planar_kernel, dim3(ew_blocks((long)u.k * u.co * u.ld / 32)), dim3(256), 0, s, u.w, (long)u.k * u.co * u.ld / 32);
            KCHK();
        }
        CHK(bcopy(v, s, p + ".bias", {u.co}, &u.b));
        ch /= 2;
        for (int j = 0; j < c.num_kernels; ++j) {
            BRes& r = v->res[(size_t)i * c.num_kernels + j];
            r.k = c.resblock_kernel_sizes[j];
            const std::string rp = "resblocks." + std::to_string(i * c.num_kernels + j);
            r.c1.assign(c.num_dilations, BConv{});
            r.c2.assign(c.num_dilations, BConv{});
            r.act.assign(2 * c.num_dilations, BAct{});
            for (int m = 0; m < c.num_dilations; ++m) {
                CHK(bconv(v, s, rp + ".convs1." + std::to_string(m), ch, ch, r.k, true, &r.c1[m]));
                CHK(bconv(v, s, rp + ".convs2." + std::to_string(m), ch, ch, r.k, true, &r.c2[m]));
            }
            for (int a = 0; a < 2 * c.num_dilations; ++a) {
                CHK(bcopy(v, s, rp + ".activations." + std::to_string(a) + ".act.alpha", {ch}, &r.act[a].alpha));
                CHK(bcopy(v, s, rp + ".activations." + std::to_string(a) + ".act.beta", {ch}, &r.act[a].beta));
            }
        }
    }
    CHK(bcopy(v, s, "activation_post.act.alpha", {v->c_last}, &v->post_act.alpha));
    CHK(bcopy(v, s, "activation_post.act.beta", {v->c_last}, &v->post_act.beta));
    {   // conv_post [1, C, 7] -> [7][C]
        const BT* t = nullptr;
        CHK(bneed(v, "conv_post.weight", {1, v->c_last, 7}, &t));
        CHK(balloc(v, (size_t)7 * v->c_last, &v->post_w));
        hipLaunchKernelGGL((permute_last2_kernel<float>), dim3(ew_blocks(7L * v->c_last)), dim3(256), 0, s, t->p, v->post_w, 1L, v->c_last, 7);
        KCHK();
        v->post_b = nullptr;
        if (c.use_bias_at_final) CHK(bcopy(v, s, "conv_post.bias", {1}, &v->post_b));
    }
    CHK(bcopy(v, s, "aux.up_filter", {12}, &v->fu));
    CHK(bcopy(v, s, "aux.down_filter", {12}, &v->fd));
    HIPCHK(hipStreamSynchronize(s));
    for (auto& kv : v->raw)
        if (kv.second.p) (void)hipFree(kv.second.p);
    v->raw.clear();
    v->finalized = true;
    return F5_OK;
}

// mel f32 addressed as mel[b * sb + c * sc + t * st] (element strides) -> wav f32[B, T * prod(rates)]
// (precision f16x3 takes the hardware-sin activation)
#define BV_ACT_LAUNCH(...)                                                                          \
    do {                                                                                            \
        if (v->cfg.precision == F5_PREC_F16X3) hipLaunchKernelGGL(bv_act_kernel<true>, __VA_ARGS__);  \
        else hipLaunchKernelGGL(bv_act_kernel<false>, __VA_ARGS__);                                  \
    } while (0)

extern "C" int f5_bigvgan_forward(f5_bigvgan* v, const float* mel, int32_t B, int32_t T, int64_t sb, int64_t sc, int64_t st,
                                  float* wav, f5_stream stream) {
    if (!v || !mel || !wav) return fail(F5_EINVAL, "f5_bigvgan_forward: null argument");
    if (!v->finalized) return fail(F5_ESTATE, "f5_bigvgan_finalize has not been called");
    if (B <= 0 || T < 1) return fail(F5_EINVAL, "f5_bigvgan_forward: need B >= 1 and T >= 1 frames");
    hipStream_t s = (hipStream_t)stream;
    const f5_bigvgan_config& c = v->cfg;
    // workspace for one utterance (utterances are decoded one after another: a stage's im2col operand is ~200 MB at 8 s)
    size_t lc_max = (size_t)T * c.upsample_initial_channel, col_max = (size_t)T * v->kpre, z_max = 0;
    {
        long L = T;
        int ch = c.upsample_initial_channel;
        for (int i = 0; i < c.num_upsamples; ++i) {
            z_max = std::max(z_max, (size_t)L * c.upsample_kernel_sizes[i] * (ch / 2));
            L *= c.upsample_rates[i];
            ch /= 2;
            lc_max = std::max(lc_max, (size_t)L * ch);
            for (int j = 0; j < c.num_kernels; ++j) col_max = std::max(col_max, (size_t)L * round_up(c.resblock_kernel_sizes[j] * ch, 32));
        }
    }
    // Stages whose channel count is a multiple of the f32 K-tile (32) run their dilated convolutions as implicit GEMMs
    // (GemmConv, gemm2.h) straight off the activation buffer, which then carries `halo` zero rows on both sides; the narrow
    // last stages (48 / 24 channels in the 24 kHz config) keep the materialised operand.
    int halo = 0;
    for (int j = 0; j < c.num_kernels; ++j)
        for (int m = 0; m < c.num_dilations; ++m) halo = std::max(halo, (c.resblock_kernel_sizes[j] - 1) / 2 * c.resblock_dilations[m]);
    const bool narrow_ok = !(getenv("F5_BIGVGAN_NARROW") && getenv("F5_BIGVGAN_NARROW")[0] == '0');         // diagnostic: 0 = GEMM path for C < 64
    const bool implicit_ok = !(getenv("F5_BIGVGAN_IMPLICIT") && getenv("F5_BIGVGAN_IMPLICIT")[0] == '0');   // diagnostic: 0 = im2col everywhere
    auto plan = [&](Arena& a, float** x, float** r, float** act, float** t1, float** col, float** Z) {
        a.reset();
        *x = a.take<float>(lc_max);
        *r = a.take<float>(lc_max * c.num_kernels);
        *act = a.take<float>(lc_max + 2 * (size_t)halo * (c.upsample_initial_channel / 2));
        *t1 = a.take<float>(lc_max);
        *col = a.take<float>(col_max);
        *Z = a.take<float>(z_max);
        return align_up(a.off, 256) + 256;
    };
    float *x, *r, *act, *t1, *col, *Z;
    Arena dry;
    const size_t need_b = plan(dry, &x, &r, &act, &t1, &col, &Z);
    if (need_b > v->arena.cap) {
        HIPCHK(hipDeviceSynchronize());
        if (v->arena.base) (void)hipFree(v->arena.base);
        v->arena.base = nullptr;
        v->arena.cap = 0;
        HIPCHK(hipMalloc((void**)&v->arena.base, need_b));
        v->arena.cap = need_b;
    }
    (void)plan(v->arena, &x, &r, &act, &t1, &col, &Z);
    const long Lout = (long)T * v->total_up;
    for (int b = 0; b < B; ++b) {
        long L = T;
        int ch = c.upsample_initial_channel;
        // conv_pre
        hipLaunchKernelGGL(im2col7_kernel, dim3(ew_blocks((long)T * v->kpre)), dim3(256), 0, s, mel + (size_t)b * sb, 0L, (long)sc, (long)st,
                           col, 1, c.num_mels, T, v->kpre);
        KCHK();
        HIPCHK(launch_gemm<float>(s, col, v->kpre, v->pre.w, v->pre.ld, T, ch, v->kpre, EpiStore<float>{x, ch, v->pre.b, F5_ACT_NONE}, -1, nullptr, 0,
                                  GemmConv{}, v->pre.split));
        for (int i = 0; i < c.num_upsamples; ++i) {
            const BUp& u = v->ups[i];
            // ConvTranspose1d: Z[L, k Co] = x[L, Ci] W'^T, then gather
            HIPCHK(launch_gemm<float>(s, x, u.ci, u.w, u.ld, (int)L, u.k * u.co, u.ci, EpiStore<float>{Z, u.k * u.co, nullptr, F5_ACT_NONE}, -1, nullptr, 0,
                                      GemmConv{}, u.split));
            hipLaunchKernelGGL(bv_upsample_gather_kernel, dim3(ew_blocks(L * u.u * (u.co / 4))), dim3(256), 0, s, Z, u.b, x, L, u.co, u.k, u.u,
                               (u.k - u.u) / 2);
            KCHK();
            L *= u.u;
            ch = u.co;
            const long cnt = L * ch;
            const bool implicit = implicit_ok && ch % 32 == 0;
            float* acti = act;                      // first activation row
            if (implicit) {
                acti = act + (size_t)halo * ch;
                HIPCHK(hipMemsetAsync(act, 0, (size_t)halo * ch * 4, s));
                HIPCHK(hipMemsetAsync(acti + cnt, 0, (size_t)halo * ch * 4, s));
            }
            auto conv = [&](const BConv& cw, int k, int d, const auto& epi) -> hipError_t {
                if (narrow_ok && cw.wn && conv_narrow_ok(ch, k, d))
                    return launch_conv_narrow(s, acti, cw.wn, cw.b, epi_res(epi), epi_out(epi), L, ch, k, d);
                // (split 2: the activation kernel wrote these rows pre-split -- planar1 / planar2 below)
                if (implicit) return launch_gemm<float>(s, acti, ch, cw.w, cw.ld, (int)L, ch, cw.ld, epi, -1, nullptr, 0,
                                                        GemmConv{ch / 32, d, (k - 1) / 2}, cw.split ? 2 : 0);
                hipLaunchKernelGGL(bv_im2col_kernel, dim3(ew_blocks(L * (cw.ld / 4))), dim3(256), 0, s, acti, col, L, ch, k, d, cw.ld);
                return launch_gemm<float>(s, col, cw.ld, cw.w, cw.ld, (int)L, ch, cw.ld, epi, -1, nullptr, 0, GemmConv{}, cw.split);
            };
            for (int j = 0; j < c.num_kernels; ++j) {
                const BRes& rb = v->res[(size_t)i * c.num_kernels + j];
                float* rj = r + (size_t)j * lc_max;
                for (int m = 0; m < c.num_dilations; ++m) {
                    const float* rin = m == 0 ? x : rj;     // the block's input: the first pair reads x itself (no copy into rj)
                    const int d = c.resblock_dilations[m];
                    const bool use_narrow1 = narrow_ok && rb.c1[m].wn && conv_narrow_ok(ch, rb.k, d);
                    const bool use_narrow2 = narrow_ok && rb.c2[m].wn && conv_narrow_ok(ch, rb.k, 1);
                    const int planar1 = implicit && !use_narrow1 && rb.c1[m].split, planar2 = implicit && !use_narrow2 && rb.c2[m].split;
                    BV_ACT_LAUNCH(dim3(ew_blocks((L + BV_TT - 1) / BV_TT * (ch / 4))), dim3(256), 0, s, rin, acti, L, ch, rb.act[2 * m].alpha,
                                       rb.act[2 * m].beta, v->fu, v->fd, planar1);
                    KCHK();
                    HIPCHK(conv(rb.c1[m], rb.k, d, EpiStore<float>{t1, ch, rb.c1[m].b, F5_ACT_NONE}));
                    BV_ACT_LAUNCH(dim3(ew_blocks((L + BV_TT - 1) / BV_TT * (ch / 4))), dim3(256), 0, s, t1, acti, L, ch, rb.act[2 * m + 1].alpha,
                                       rb.act[2 * m + 1].beta, v->fu, v->fd, planar2);
                    KCHK();
                    // x_j = x_j + (conv2(.) + bias): residual epilogue (in place from the second pair on)
                    HIPCHK(conv(rb.c2[m], rb.k, 1, EpiGateRes{rj, rin, ch, rb.c2[m].b, nullptr, 0, (int)L + 1, nullptr}));
                }
            }
            hipLaunchKernelGGL(bv_mean_kernel, dim3(ew_blocks(cnt / 4)), dim3(256), 0, s, r, (long)lc_max, c.num_kernels, x, cnt / 4);
            KCHK();
        }
        BV_ACT_LAUNCH(dim3(ew_blocks((L + BV_TT - 1) / BV_TT * (ch / 4))), dim3(256), 0, s, x, act, L, ch, v->post_act.alpha, v->post_act.beta,
                           v->fu, v->fd, 0);
        hipLaunchKernelGGL(bv_post_kernel, dim3(ew_blocks(L)), dim3(256), 0, s, act, v->post_w, v->post_b, wav + (size_t)b * Lout, L, ch,
                           c.use_tanh_at_final);
        KCHK();
    }
    return F5_OK;
}

