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
