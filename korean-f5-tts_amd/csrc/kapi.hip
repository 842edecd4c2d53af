// Kernel-level C entry points (f5k_*): each runs ONE kernel of the engine on fp32 inputs so that tests/ can compare it
// with a torch fp32 restatement, and the micro-benchmarks can time it.  Scratch is allocated per call; every function
// synchronises the stream before returning.
#include <vector>

#include "attn2.h"
#include "convpos.h"
#include "elementwise.h"
#include "gemm_dispatch.h"
#include "internal.h"

using namespace f5;
#define fail f5_fail
// precision dispatch of a function template call FN<T>(args...)
#define F5K_BY_PREC(prec, FN, ...) \
    ((prec) == F5_PREC_BF16 ? FN<bf16_t>(__VA_ARGS__) : (prec) == F5_PREC_F16 ? FN<f16_t>(__VA_ARGS__) : FN<float>(__VA_ARGS__))
// F5_PREC_F16X3 runs the f32 instantiation with the W operand split (f5k_gemm / f5k_gemm_time only)
static bool g_split16 = false;
template <typename T> static hipError_t maybe_split(hipStream_t s, T* w, size_t elems) {
    if constexpr (std::is_same_v<T, float>) {
        if (g_split16) hipLaunchKernelGGL(split_planar_kernel, dim3(ew_blocks((long)(elems / 32))), dim3(256), 0, s, w, (long)(elems / 32));
    }
    return hipGetLastError();
}

template <typename T>
static int gemm_impl(const float* A, const float* W, const float* bias, int act, float* out, int M, int N, int K, int tm,
                     int tn, hipStream_t s) {
    const int Kp = tm > 0 ? round_up(K, 8) : round_up(K, 128 / (int)sizeof(T));  // tm > 0 forces the v1 kernel
    Scratch<T> a, w;
    HIPCHK(a.alloc((size_t)M * Kp));
    HIPCHK(w.alloc((size_t)N * Kp));
    hipLaunchKernelGGL((cast_pad_kernel<T>), dim3(ew_blocks((long)M * Kp)), dim3(256), 0, s, A, K, M, K, a.p, Kp, M);
    hipLaunchKernelGGL((cast_pad_kernel<T>), dim3(ew_blocks((long)N * Kp)), dim3(256), 0, s, W, K, N, K, w.p, Kp, N);
    KCHK();
    const bool split = g_split16 && std::is_same_v<T, float>;
    if (split && tm > 0) return fail(F5_EINVAL, "f5k_gemm: the split-operand mode has no v1 kernel");
    HIPCHK(maybe_split<T>(s, w.p, (size_t)N * Kp));
    // F5_PREC_F16X3 with tn == 5: the A operand pre-split as well (what store4_planar producers hand the block GEMMs: MODE 5)
    const bool a_planar = split && tm <= 0 && tn == 5;
    if (a_planar) HIPCHK(maybe_split<T>(s, a.p, (size_t)M * Kp));
    if (tm > 0) HIPCHK(launch_gemm_v1<T>(s, a.p, Kp, w.p, Kp, M, N, Kp, EpiStore<float>{out, N, bias, act}, tm, tn));
    else HIPCHK(launch_gemm<T>(s, a.p, Kp, w.p, Kp, M, N, Kp, EpiStore<float>{out, N, bias, act}, tm < 0 ? -tm : -1, nullptr, 0, GemmConv{},
                               a_planar ? 2 : (split ? 1 : 0)));
    HIPCHK(hipStreamSynchronize(s));
    return F5_OK;
}

extern "C" int f5k_gemm(int32_t prec, const float* A, const float* W, const float* bias, int32_t act, float* out, int32_t M,
                        int32_t N, int32_t K, int32_t tm, int32_t tn, f5_stream stream) {
    if (!A || !W || !out || M <= 0 || N <= 0 || K <= 0 || (N % 4)) return fail(F5_EINVAL, "f5k_gemm: bad arguments (N %% 4 == 0)");
    if (tm > 0 && !((tm == 128 && (tn == 128 || tn == 64)) || (tm == 64 && tn == 64))) return fail(F5_EINVAL, "f5k_gemm: bad tile");
    if (tm < 0 && tm != -2 && tm != -8 && tm != -9 && tm != -10 && tm != -13 && tm != -20) return fail(F5_EINVAL, "f5k_gemm: bad v2 / v3 config id");
    hipStream_t s = (hipStream_t)stream;
    g_split16 = prec == F5_PREC_F16X3;
    const int rc = F5K_BY_PREC(prec, gemm_impl, A, W, bias, act, out, M, N, K, tm, tn, s);
    g_split16 = false;
    return rc;
}

template <typename T>
static int gemm_time_impl(int M, int N, int K, int tm, int tn, int iters, float* avg_us, hipStream_t s) {
    const int Kp = tm > 0 ? round_up(K, 8) : round_up(K, 128 / (int)sizeof(T));
    Scratch<T> a, w, o;
    HIPCHK(a.alloc((size_t)M * Kp));
    HIPCHK(w.alloc((size_t)N * Kp));
    HIPCHK(o.alloc((size_t)M * N));
    // random-ish operand bits (bench on non-zero data: MI355X_MICROARCH "DVFS give-back")
    Scratch<float> tmp;
    const size_t nmax = std::max((size_t)M, (size_t)N) * Kp;
    HIPCHK(tmp.alloc(nmax));
    std::vector<float> h(nmax);
    unsigned x = 12345u;
    for (size_t i = 0; i < nmax; ++i) {
        x = x * 1664525u + 1013904223u;
        h[i] = ((x >> 8) & 0xFFFF) / 32768.0f - 1.0f;
    }
    HIPCHK(hipMemcpy(tmp.p, h.data(), nmax * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL((cast_pad_kernel<T>), dim3(ew_blocks((long)M * Kp)), dim3(256), 0, s, tmp.p, Kp, M, Kp, a.p, Kp, M);
    hipLaunchKernelGGL((cast_pad_kernel<T>), dim3(ew_blocks((long)N * Kp)), dim3(256), 0, s, tmp.p, Kp, N, Kp, w.p, Kp, N);
    KCHK();
    hipEvent_t e0, e1;
    HIPCHK(hipEventCreate(&e0));
    HIPCHK(hipEventCreate(&e1));
    auto go = [&]() -> hipError_t {
        if (tm > 0) return launch_gemm_v1<T>(s, a.p, Kp, w.p, Kp, M, N, Kp, EpiStore<T>{o.p, N, nullptr, 0}, tm, tn);
        return launch_gemm<T>(s, a.p, Kp, w.p, Kp, M, N, Kp, EpiStore<T>{o.p, N, nullptr, 0}, tm < 0 ? -tm : -1, nullptr, 0, GemmConv{},
                              g_split16 && std::is_same_v<T, float>);
    };
    for (int i = 0; i < 3; ++i) HIPCHK(go());
    HIPCHK(hipEventRecord(e0, s));
    for (int i = 0; i < iters; ++i) HIPCHK(go());
    HIPCHK(hipEventRecord(e1, s));
    HIPCHK(hipEventSynchronize(e1));
    float ms = 0.f;
    HIPCHK(hipEventElapsedTime(&ms, e0, e1));
    *avg_us = ms * 1000.0f / iters;
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    return F5_OK;
}

extern "C" int f5k_gemm_time(int32_t prec, int32_t M, int32_t N, int32_t K, int32_t tm, int32_t tn, int32_t iters,
                             float* avg_us, f5_stream stream) {
    if (!avg_us || M <= 0 || N <= 0 || K <= 0 || iters <= 0 || (N % 4)) return fail(F5_EINVAL, "f5k_gemm_time: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    g_split16 = prec == F5_PREC_F16X3;   // (timing only: the random W bits are used as they are)
    const int rc = F5K_BY_PREC(prec, gemm_time_impl, M, N, K, tm, tn, iters, avg_us, s);
    g_split16 = false;
    return rc;
}

// packs fp32 [Bp,H,N,64] q/k/v into the engine layouts (q scaled, v transposed) -- test-side glue only
template <typename T>
__global__ void pack_qkv_test_kernel(const float* q, const float* k, const float* v, T* qo, T* ko, T* vto, long nrows, int N,
                                     int Npad, float qscale) {
    const long total = nrows * 64;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int d = (int)(i % 64);
        const long row = i / 64;  // (b*H + h)*N + n
        const long bh = row / N;
        const int n = (int)(row % N);
        qo[i] = from_f32<T>(q[i] * qscale);
        ko[i] = from_f32<T>(k[i]);
        vto[(bh * 64 + d) * Npad + n] = from_f32<T>(v[i]);
    }
}
template <typename T> __global__ void to_f32_kernel(const T* in, float* out, long n) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) out[i] = (float)in[i];
}

template <typename T>
static int attn_impl(const float* q, const float* k, const float* v, const int32_t* lens_host, float* out, int Bp, int H, int N,
                     hipStream_t s) {
    const int Npad = round_up(N, 64);
    const long rows = (long)Bp * H * N;
    Scratch<T> qd, kd, vd, od;
    Scratch<int> ld;
    HIPCHK(qd.alloc(rows * 64));
    HIPCHK(kd.alloc(rows * 64));
    HIPCHK(vd.alloc((size_t)Bp * H * 64 * Npad));
    HIPCHK(od.alloc(rows * 64));
    HIPCHK(hipMemsetAsync(vd.p, 0, (size_t)Bp * H * 64 * Npad * sizeof(T), s));
    if (lens_host) {
        HIPCHK(ld.alloc(Bp));
        HIPCHK(hipMemcpy(ld.p, lens_host, (size_t)Bp * 4, hipMemcpyHostToDevice));
    }
    hipLaunchKernelGGL((pack_qkv_test_kernel<T>), dim3(ew_blocks(rows * 64)), dim3(256), 0, s, q, k, v, qd.p, kd.p, vd.p, rows,
                       N, Npad, attention_q_scale<T>());
    KCHK();
    HIPCHK(launch_attention_any(s, qd.p, kd.p, vd.p, od.p, Bp, H, N, Npad, lens_host ? ld.p : nullptr, Bp, nullptr, nullptr, g_split16));
    hipLaunchKernelGGL((to_f32_kernel<T>), dim3(ew_blocks(rows * 64)), dim3(256), 0, s, od.p, out, rows * 64);
    KCHK();
    HIPCHK(hipStreamSynchronize(s));
    return F5_OK;
}

extern "C" int f5k_attention(int32_t prec, const float* q, const float* k, const float* v, const int32_t* kv_lens_host,
                             float* out, int32_t Bp, int32_t H, int32_t N, f5_stream stream) {
    if (!q || !k || !v || !out || Bp <= 0 || H <= 0 || N <= 0) return fail(F5_EINVAL, "f5k_attention: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    g_split16 = prec == F5_PREC_F16X3;
    const int rc = F5K_BY_PREC(prec, attn_impl, q, k, v, kv_lens_host, out, Bp, H, N, s);
    g_split16 = false;
    return rc;
}

template <typename T>
static int convpos_impl(const float* x, const float* w, const float* bias, const float* res, const int32_t* lens_host, float* y,
                        int Bp, int N, int D, hipStream_t s) {
    const int cpg = D / 16, Kp = round_up(31 * cpg, GEMM_ROW_BYTES / (int)sizeof(T));
    Scratch<T> wp;
    Scratch<int> ld;
    HIPCHK(wp.alloc((size_t)D * Kp));
    if (lens_host) {
        HIPCHK(ld.alloc(Bp));
        HIPCHK(hipMemcpy(ld.p, lens_host, (size_t)Bp * 4, hipMemcpyHostToDevice));
    }
    hipLaunchKernelGGL((conv_pack_kernel<T>), dim3(ew_blocks((long)D * Kp)), dim3(256), 0, s, w, wp.p, (long)D, cpg, 31, Kp);
    KCHK();
    const bool split = g_split16 && std::is_same_v<T, float> && convpos_can_split(D);
    if (split) HIPCHK(maybe_split<T>(s, wp.p, (size_t)D * Kp));
    HIPCHK(launch_convpos<T>(s, x, wp.p, Kp, bias, res, y, Bp, N, D, lens_host ? ld.p : nullptr, Bp, nullptr, split));
    HIPCHK(hipStreamSynchronize(s));
    return F5_OK;
}

extern "C" int f5k_convpos(int32_t prec, const float* x, const float* w, const float* bias, const float* res,
                           const int32_t* lens_host, float* y, int32_t Bp, int32_t N, int32_t D, f5_stream stream) {
    if (!x || !w || !bias || !y || Bp <= 0 || N <= 0) return fail(F5_EINVAL, "f5k_convpos: bad arguments");
    if (D != 256 && D != 512 && D != 768 && D != 1024) return fail(F5_EINVAL, "f5k_convpos: D must be 256, 512, 768 or 1024");
    hipStream_t s = (hipStream_t)stream;
    g_split16 = prec == F5_PREC_F16X3;
    const int rc = F5K_BY_PREC(prec, convpos_impl, x, w, bias, res, lens_host, y, Bp, N, D, s);
    g_split16 = false;
    return rc;
}

extern "C" int f5k_layernorm_mod(const float* x, const float* scale, const float* shift, float* out, int32_t R, int32_t D,
                                 int32_t rows_per_batch, float eps, f5_stream stream) {
    if (!x || !out || R <= 0 || D <= 0 || D % 4 || D > 2048 || rows_per_batch <= 0) return fail(F5_EINVAL, "f5k_layernorm_mod: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL((layernorm_kernel<float>), dim3((R + 3) / 4), dim3(256), 0, s, x, D, out, D, R, D, eps, scale, shift, D,
                       rows_per_batch, 1, Prefetch{});
    KCHK();
    HIPCHK(hipStreamSynchronize(s));
    return F5_OK;
}

// ------------------------------------------------------------------------------------ v2 (glds ring) GEMM
#include "gemm2.h"
static int g_cold_weights = 0;
static int g_out_bf16 = 0;

template <typename T, typename Epi>
static hipError_t gemm2_dispatch(int cfg, hipStream_t s, const T* A, int lda, const T* W, int ldw, int M, int N, int K,
                                 const Epi& epi) {
    switch (cfg) {
        case 0: return launch_gemm2_cfg<T, 128, 128, 2, 2, 3, Epi>(s, A, lda, W, ldw, M, N, K, epi);
        case 1: return launch_gemm2_cfg<T, 128, 128, 2, 2, 4, Epi>(s, A, lda, W, ldw, M, N, K, epi);
        case 13: return launch_gemm2_cfg<T, 256, 128, 4, 2, 3, Epi>(s, A, lda, W, ldw, M, N, K, epi);
        case 20: return launch_gemm3<T, Epi>(s, A, lda, W, ldw, M, N, K, epi);
        case 2: return launch_gemm2_cfg<T, 128, 128, 2, 4, 4, Epi>(s, A, lda, W, ldw, M, N, K, epi);
        case 3: return launch_gemm2_cfg<T, 128, 64, 2, 2, 4, Epi>(s, A, lda, W, ldw, M, N, K, epi);
        case 5: return launch_gemm2_cfg<T, 64, 64, 2, 2, 4, Epi>(s, A, lda, W, ldw, M, N, K, epi);
        case 7: return launch_gemm2_cfg<T, 128, 128, 4, 2, 3, Epi>(s, A, lda, W, ldw, M, N, K, epi);
        case 8: return launch_gemm2_cfg<T, 64, 64, 2, 2, 3, Epi>(s, A, lda, W, ldw, M, N, K, epi);
        case 9: return launch_gemm2_cfg<T, 128, 64, 4, 2, 4, Epi>(s, A, lda, W, ldw, M, N, K, epi);
        case 10: return launch_gemm2_cfg<T, 128, 192, 2, 4, 3, Epi>(s, A, lda, W, ldw, M, N, K, epi);
        case 11: return launch_gemm2_cfg<T, 128, 192, 2, 4, 4, Epi>(s, A, lda, W, ldw, M, N, K, epi);
        // diagnostic floors of config 2 (outputs are garbage): 1xx = DMA only, 2xx = compute only
        case 102: return launch_gemm2_cfg<T, 128, 128, 2, 4, 4, Epi, 1>(s, A, lda, W, ldw, M, N, K, epi);
        case 202: return launch_gemm2_cfg<T, 128, 128, 2, 4, 4, Epi, 2>(s, A, lda, W, ldw, M, N, K, epi);
        case 16: return launch_gemm2_cfg<T, 128, 128, 2, 4, 2, Epi>(s, A, lda, W, ldw, M, N, K, epi);   // 64 KB LDS: 2 workgroups per CU
        case 17: return launch_gemm2_cfg<T, 256, 128, 4, 2, 2, Epi>(s, A, lda, W, ldw, M, N, K, epi);   // 96 KB
        case 18: return launch_gemm2_cfg<T, 128, 128, 2, 2, 2, Epi>(s, A, lda, W, ldw, M, N, K, epi);   // 4 waves, 64 KB: 2 per CU
        case 113: return launch_gemm2_cfg<T, 256, 128, 4, 2, 3, Epi, 1>(s, A, lda, W, ldw, M, N, K, epi);
        case 213: return launch_gemm2_cfg<T, 256, 128, 4, 2, 3, Epi, 2>(s, A, lda, W, ldw, M, N, K, epi);
        case 413: return launch_gemm2_cfg<T, 256, 128, 4, 2, 3, Epi, 4>(s, A, lda, W, ldw, M, N, K, epi);
        case 402: return launch_gemm2_cfg<T, 128, 128, 2, 4, 4, Epi, 4>(s, A, lda, W, ldw, M, N, K, epi);  // no epilogue
        case 409: return launch_gemm2_cfg<T, 128, 64, 4, 2, 4, Epi, 4>(s, A, lda, W, ldw, M, N, K, epi);
        default: return hipErrorInvalidValue;
    }
}

template <typename T>
static int gemm2_impl(const float* A, const float* W, const float* bias, int act, float* out, int M, int N, int K, int cfg,
                      int iters, float* avg_us, hipStream_t s) {
    const int Kp = round_up(K, 128 / (int)sizeof(T));
    Scratch<T> a, w;
    HIPCHK(a.alloc((size_t)M * Kp));
    HIPCHK(w.alloc((size_t)N * Kp));
    hipLaunchKernelGGL((cast_pad_kernel<T>), dim3(ew_blocks((long)M * Kp)), dim3(256), 0, s, A, K, M, K, a.p, Kp, M);
    hipLaunchKernelGGL((cast_pad_kernel<T>), dim3(ew_blocks((long)N * Kp)), dim3(256), 0, s, W, K, N, K, w.p, Kp, N);
    KCHK();
    HIPCHK(gemm2_dispatch<T>(cfg, s, a.p, Kp, w.p, Kp, M, N, Kp, EpiStore<float>{out, N, bias, act}));
    if (iters > 0 && avg_us) {
        // iters < 0 is not used; a NEGATIVE act selects "cold weights": the launches cycle through enough copies of W
        // to exceed the 256 MiB Infinity Cache, as the 22 layers x 4 projections of a DiT step do
        const bool cold = g_cold_weights != 0;
        const size_t wbytes = (size_t)N * Kp * sizeof(T);
        const int ncopy = cold ? (int)std::min<size_t>(64, (320u << 20) / wbytes + 1) : 1;
        Scratch<T> wc;
        if (cold) {
            HIPCHK(wc.alloc((size_t)ncopy * N * Kp));
            for (int c = 0; c < ncopy; ++c)
                HIPCHK(hipMemcpyAsync(wc.p + (size_t)c * N * Kp, w.p, wbytes, hipMemcpyDeviceToDevice, s));
        }
        hipEvent_t e0, e1;
        HIPCHK(hipEventCreate(&e0));
        HIPCHK(hipEventCreate(&e1));
        for (int i = 0; cold && i < ncopy; ++i)
            HIPCHK(gemm2_dispatch<T>(cfg, s, a.p, Kp, wc.p + (size_t)(i % ncopy) * N * Kp, Kp, M, N, Kp, EpiStore<float>{out, N, bias, act}));
        HIPCHK(hipEventRecord(e0, s));
        for (int i = 0; i < iters; ++i) {
            const T* wp = cold ? wc.p + (size_t)(i % ncopy) * N * Kp : w.p;
            if (g_out_bf16)  // timing-only: reinterpret the fp32 output buffer as bf16 (half of it is written)
                HIPCHK(gemm2_dispatch<T>(cfg, s, a.p, Kp, wp, Kp, M, N, Kp, EpiStore<bf16_t>{reinterpret_cast<bf16_t*>(out), N, bias, act}));
            else
                HIPCHK(gemm2_dispatch<T>(cfg, s, a.p, Kp, wp, Kp, M, N, Kp, EpiStore<float>{out, N, bias, act}));
        }
        HIPCHK(hipEventRecord(e1, s));
        HIPCHK(hipEventSynchronize(e1));
        float ms = 0.f;
        HIPCHK(hipEventElapsedTime(&ms, e0, e1));
        *avg_us = ms * 1000.f / iters;
        (void)hipEventDestroy(e0);
        (void)hipEventDestroy(e1);
    }
    HIPCHK(hipStreamSynchronize(s));
    return F5_OK;
}

// experimental entry (not in the public header): v2 GEMM with config id; optional timing over `iters` launches
extern "C" int f5x_set_xcd_mode(int32_t on) { xcd_mode() = on; return 0; }
extern "C" int f5x_set_attn_variant(int32_t v) { attn2_variant() = v; return 0; }
extern "C" int f5x_set_cold_weights(int32_t on) { g_cold_weights = on; return 0; }
extern "C" int f5x_set_out_bf16(int32_t on) { g_out_bf16 = on; return 0; }
extern "C" int f5x_gemm2(int32_t prec, const float* A, const float* W, const float* bias, int32_t act, float* out, int32_t M,
                         int32_t N, int32_t K, int32_t cfg, int32_t iters, float* avg_us, f5_stream stream) {
    if (!A || !W || !out || M <= 0 || N <= 0 || K <= 0 || (N % 4)) return fail(F5_EINVAL, "f5x_gemm2: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    return F5K_BY_PREC(prec, gemm2_impl, A, W, bias, act, out, M, N, K, cfg, iters, avg_us, s);
}


// diagnostic: what a boundary between DIFFERENT kernels costs.  Repeating patterns, cold (rotating) weights:
//   LN alone, GEMM alone, [LN -> GEMM] (real producer/consumer edge), [LN' -> GEMM] (no data edge),
//   [GEMM a -> GEMM b] (two instantiations), [GEMM a -> GEMM a]
extern "C" int f5x_pair_time(int32_t M, int32_t N, int32_t K, int32_t cfg, int32_t iters, float* res6, f5_stream stream) {
    hipStream_t s = (hipStream_t)stream;
    typedef bf16_t T;
    Scratch<float> x, sc;
    Scratch<T> xn, xn2, w, o;
    const int ncopy = 48;
    HIPCHK(x.alloc((size_t)M * K));
    HIPCHK(sc.alloc((size_t)2 * K));
    HIPCHK(xn.alloc((size_t)M * K));
    HIPCHK(xn2.alloc((size_t)M * K));
    HIPCHK(w.alloc((size_t)ncopy * N * K));
    HIPCHK(o.alloc((size_t)M * N));
    HIPCHK(hipMemsetAsync(x.p, 0x3c, (size_t)M * K * 4, s));
    HIPCHK(hipMemsetAsync(sc.p, 0, (size_t)2 * K * 4, s));
    HIPCHK(hipMemsetAsync(w.p, 0x3c, (size_t)ncopy * N * K * 2, s));
    auto ln = [&](T* dst) {
        hipLaunchKernelGGL((layernorm_kernel<T>), dim3((M + 3) / 4), dim3(256), 0, s, x.p, K, dst, K, M, K, 1e-6f, sc.p, sc.p + K, 0, M, 1, Prefetch{});
    };
    auto gm = [&](int i, int c) -> hipError_t {
        return gemm2_dispatch<T>(c, s, xn.p, K, w.p + (size_t)(i % ncopy) * N * K, K, M, N, K, EpiStore<T>{o.p, N, nullptr, F5_ACT_GELU_TANH});
    };
    const int cfg_b = cfg == 2 ? 7 : 2;
    hipEvent_t e0, e1;
    HIPCHK(hipEventCreate(&e0));
    HIPCHK(hipEventCreate(&e1));
    for (int i = 0; i < 8; ++i) { ln(xn.p); HIPCHK(gm(i, cfg)); HIPCHK(gm(i, cfg_b)); }
    // modes 6, 7 (N == 3 * 16 * 64 only): the same GEMM with the fused QKV epilogue / with EpiStore + bias, both alone
    Scratch<T> q3;
    Scratch<float> tab;
    const bool qkv_ok = N == 3072 && M % 1024 == 0;
    HIPCHK(q3.alloc((size_t)3 * M * 1024 + 4096));
    HIPCHK(tab.alloc((size_t)2 * 4096 * 32 + N));
    HIPCHK(hipMemsetAsync(tab.p, 0, ((size_t)2 * 4096 * 32 + N) * 4, s));
    static int variant = getenv("F5X_QKV_VARIANT") ? atoi(getenv("F5X_QKV_VARIANT")) : 0;
    auto gq = [&](int i) -> hipError_t {
        EpiQKV<T> e{q3.p, q3.p + (size_t)M * 1024, q3.p + (size_t)2 * M * 1024, tab.p + 2 * 4096 * 32, tab.p, tab.p + 4096 * 32,
                    1024, 1024, 16, 1, 0.125f};
        if (variant == 1) { e.H = 24; e.k = q3.p + (size_t)M * 1536; }  // no transposed third: q | k of 24 heads each
        if (variant == 2) e.pe_heads = 0;                                 // no rotary
        if (variant == 3) { e.H = 24; e.k = q3.p + (size_t)M * 1536; e.Nseq = 1; e.Npad = 1; }  // same code, row-major q | k
        return gemm2_dispatch<T>(cfg, s, xn.p, K, w.p + (size_t)(i % ncopy) * N * K, K, M, N, K, e);
    };
    auto gb = [&](int i) -> hipError_t {
        return gemm2_dispatch<T>(cfg, s, xn.p, K, w.p + (size_t)(i % ncopy) * N * K, K, M, N, K,
                                 EpiStore<T>{q3.p, N, tab.p + 2 * 4096 * 32, F5_ACT_NONE});
    };
    for (int mode = 0; mode < 8; ++mode) {
        if (mode >= 6 && !qkv_ok) { res6[mode] = 0.f; continue; }
        HIPCHK(hipEventRecord(e0, s));
        for (int i = 0; i < iters; ++i) {
            switch (mode) {
                case 6: HIPCHK(gq(i)); break;
                case 7: HIPCHK(gb(i)); break;
                case 0: ln(xn.p); break;
                case 1: HIPCHK(gm(i, cfg)); break;
                case 2: ln(xn.p); HIPCHK(gm(i, cfg)); break;
                case 3: ln(xn2.p); HIPCHK(gm(i, cfg)); break;
                case 4: HIPCHK(gm(i, cfg)); HIPCHK(gm(i + 7, cfg_b)); break;
                default: HIPCHK(gm(i, cfg)); HIPCHK(gm(i + 7, cfg)); break;
            }
        }
        HIPCHK(hipEventRecord(e1, s));
        HIPCHK(hipEventSynchronize(e1));
        float ms = 0.f;
        HIPCHK(hipEventElapsedTime(&ms, e0, e1));
        res6[mode] = ms * 1000.f / iters;
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    return F5_OK;
}


// diagnostic: what a software grid barrier between the phases of a persistent kernel would cost on this part (cross-XCD
// visibility included: agent-scope release before arriving, acquire after leaving, neighbour's data verified).  Every
// spin is bounded, so the kernel always terminates; *bad counts timeouts and stale reads.
__global__ __launch_bounds__(512) void grid_barrier_probe_kernel(unsigned* ctr, unsigned* data, int iters, int payload_floats,
                                                                 float* payload, unsigned* bad) {
    const unsigned nb = gridDim.x;
    unsigned errors = 0;
    for (int it = 0; it < iters; ++it) {
        // phase work: each block writes a slice (payload_floats per thread) and its tag
        for (int p = 0; p < payload_floats; ++p)
            payload[((size_t)blockIdx.x * payload_floats + p) * blockDim.x + threadIdx.x] = (float)it;
        if (threadIdx.x == 0) data[blockIdx.x] = (unsigned)it + 1u;
        __threadfence();                       // release: make this block's writes visible device-wide
        __syncthreads();
        if (threadIdx.x == 0) {
            __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned target = (unsigned)(it + 1) * nb;
            int spins = 0;
            while (__hip_atomic_load(ctr, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) {
                __builtin_amdgcn_s_sleep(2);
                if (++spins > 2000000) { errors += 1000000u; break; }
            }
        }
        __syncthreads();
        __threadfence();                       // acquire on behalf of the whole block
        const unsigned nbr = (blockIdx.x + 37u) % nb;   // a block on another XCD (ids are dealt round-robin over 8 XCDs)
        if (threadIdx.x == 0 && __hip_atomic_load(&data[nbr], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != (unsigned)it + 1u) errors++;
        if (payload_floats > 0) {
            const float v = payload[((size_t)nbr * payload_floats) * blockDim.x + threadIdx.x];
            if (v != (float)it) errors++;
        }
    }
    if (errors) atomicAdd(bad, errors);
}

extern "C" int f5x_grid_barrier_probe(int32_t blocks, int32_t iters, int32_t payload_floats, float* us_per_barrier, int32_t* bad_out,
                                      f5_stream stream) {
    hipStream_t s = (hipStream_t)stream;
    Scratch<unsigned> ctr, data, bad;
    Scratch<float> payload;
    HIPCHK(ctr.alloc(1));
    HIPCHK(bad.alloc(1));
    HIPCHK(data.alloc(blocks));
    HIPCHK(payload.alloc((size_t)blocks * std::max(payload_floats, 1) * 512));
    hipEvent_t e0, e1;
    HIPCHK(hipEventCreate(&e0));
    HIPCHK(hipEventCreate(&e1));
    float ms = 0.f;
    for (int rep = 0; rep < 2; ++rep) {   // first launch warms the code
        HIPCHK(hipMemsetAsync(ctr.p, 0, 4, s));
        HIPCHK(hipMemsetAsync(bad.p, 0, 4, s));
        HIPCHK(hipMemsetAsync(data.p, 0, (size_t)blocks * 4, s));
        HIPCHK(hipEventRecord(e0, s));
        hipLaunchKernelGGL(grid_barrier_probe_kernel, dim3(blocks), dim3(512), 0, s, ctr.p, data.p, iters, payload_floats, payload.p, bad.p);
        HIPCHK(hipEventRecord(e1, s));
        HIPCHK(hipEventSynchronize(e1));
        HIPCHK(hipEventElapsedTime(&ms, e0, e1));
    }
    unsigned b = 0;
    HIPCHK(hipMemcpy(&b, bad.p, 4, hipMemcpyDeviceToHost));
    *us_per_barrier = ms * 1000.f / iters;
    *bad_out = (int32_t)b;
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    return F5_OK;
}


// diagnostic: bare MFMA issue rate and the shader clock it runs at.  Every wave issues `iters` x 8 independent
// v_mfma_f32_16x16x32_bf16 (operands in registers); s_memtime counts shader clocks, s_memrealtime a constant 100 MHz.
__global__ __launch_bounds__(512) void mfma_rate_probe_kernel(int iters, unsigned long long* out) {
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (bf16_t)(float)(threadIdx.x & 7); b[i] = (bf16_t)1.0f; }
    f32x4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    const unsigned long long c0 = __builtin_readcyclecounter(), r0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
    }
    const unsigned long long c1 = __builtin_readcyclecounter(), r1 = wall_clock64();
    float sink = 0.f;
    for (int i = 0; i < 8; ++i) sink += acc[i][0] + acc[i][3];
    if (threadIdx.x == 0) {
        out[blockIdx.x * 3 + 0] = c1 - c0;
        out[blockIdx.x * 3 + 1] = r1 - r0;
        out[blockIdx.x * 3 + 2] = (unsigned long long)(sink != 12345.678f);
    }
}

extern "C" int f5x_mfma_rate_probe(int32_t blocks, int32_t threads, int32_t iters, double* cyc_per_mfma_per_simd, double* mhz,
                                   double* tflops, f5_stream stream) {
    hipStream_t s = (hipStream_t)stream;
    Scratch<unsigned long long> out;
    HIPCHK(out.alloc((size_t)blocks * 3));
    hipEvent_t e0, e1;
    HIPCHK(hipEventCreate(&e0));
    HIPCHK(hipEventCreate(&e1));
    float ms = 0.f;
    for (int rep = 0; rep < 3; ++rep) {
        HIPCHK(hipEventRecord(e0, s));
        hipLaunchKernelGGL(mfma_rate_probe_kernel, dim3(blocks), dim3(threads), 0, s, iters, out.p);
        HIPCHK(hipEventRecord(e1, s));
        HIPCHK(hipEventSynchronize(e1));
        HIPCHK(hipEventElapsedTime(&ms, e0, e1));
    }
    std::vector<unsigned long long> h((size_t)blocks * 3);
    HIPCHK(hipMemcpy(h.data(), out.p, h.size() * 8, hipMemcpyDeviceToHost));
    double cyc = 0, real = 0;
    for (int b = 0; b < blocks; ++b) { cyc += (double)h[b * 3]; real += (double)h[b * 3 + 1]; }
    cyc /= blocks; real /= blocks;
    const int waves_per_simd = (threads / 64 + 3) / 4;
    *cyc_per_mfma_per_simd = cyc / ((double)iters * 8 * waves_per_simd);
    *mhz = cyc / (real / 100.0);   // real counts 100 MHz ticks -> microseconds = real / 100
    *tflops = (double)blocks * (threads / 64) * iters * 8 * 16384.0 / (ms * 1e-3) / 1e12;
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    return F5_OK;
}
