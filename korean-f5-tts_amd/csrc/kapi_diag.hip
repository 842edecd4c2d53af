// Diagnostic / micro-benchmark entry points (f5x_*, not in the public header): GEMM tile sweeps and pipeline floors, the DiT-block
// GEMM sequence with its real epilogues, kernel-boundary costs, the grid-barrier and MFMA-rate probes.  Split from kapi.hip so
// that the many diagnostic GEMM instantiations compile beside it, not after it.
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
        case 420: return launch_gemm3<T, Epi, 4>(s, A, lda, W, ldw, M, N, K, epi);   // no epilogue
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


// diagnostic: the four GEMMs of a DiT block (bf16) with their REAL epilogues, in block order over the block's real buffers
// (so that the residual stream and the weights are as cold as in the engine: q/k/v^T + ffh between two uses of x), each launch
// between its own event pair.  cfg: -1 = the product's tile choice (launch_gemm), else a GemmCfg id forced on all four.
// us4 = average us of {QKV, out-proj, FF1, FF2}.  tools/block_gemm_time.py
extern "C" int f5x_block_gemm_time(int32_t M, int32_t cfg, int32_t iters, float* us4, f5_stream stream) {
    if (M <= 0 || M % 1024 || iters <= 0 || !us4) return fail(F5_EINVAL, "f5x_block_gemm_time: M must be a multiple of 1024");
    hipStream_t s = (hipStream_t)stream;
    typedef bf16_t T;
    const int D = 1024, F = 2048, N = 1024, ncopy = 8;
    // diagnostics: F5X_QKV_H = 24 makes the epilogue treat all 3,072 columns as q / k heads (no transposed V^T region: q / k buffers
    // are allocated 1.5x), F5X_QKV_PE = heads that get rotary
    const int H = getenv("F5X_QKV_H") ? atoi(getenv("F5X_QKV_H")) : 16;
    const int PE = getenv("F5X_QKV_PE") ? atoi(getenv("F5X_QKV_PE")) : 1;
    Scratch<T> xn, ao, ffh, q, k, vt, wqkv, wout, wff1, wff2;
    Scratch<float> x, bias, gate, rope, rnd;
    HIPCHK(xn.alloc((size_t)M * D)); HIPCHK(ao.alloc((size_t)M * D)); HIPCHK(ffh.alloc((size_t)M * F));
    HIPCHK(q.alloc((size_t)M * D * 3 / 2)); HIPCHK(k.alloc((size_t)M * D * 3 / 2)); HIPCHK(vt.alloc((size_t)M * D));
    HIPCHK(wqkv.alloc((size_t)ncopy * 3 * D * D)); HIPCHK(wout.alloc((size_t)ncopy * D * D));
    HIPCHK(wff1.alloc((size_t)ncopy * F * D)); HIPCHK(wff2.alloc((size_t)ncopy * D * F));
    HIPCHK(x.alloc((size_t)M * D)); HIPCHK(bias.alloc(3 * D)); HIPCHK(gate.alloc((size_t)(M / N) * D)); HIPCHK(rope.alloc((size_t)2 * N * 32));
    const size_t nr = (size_t)4 << 20;
    HIPCHK(rnd.alloc(nr));
    std::vector<float> h(nr);
    unsigned r = 777u;
    for (size_t i = 0; i < nr; ++i) { r = r * 1664525u + 1013904223u; h[i] = (((r >> 8) & 0xFFFF) / 32768.0f - 1.0f) * 0.05f; }
    HIPCHK(hipMemcpy(rnd.p, h.data(), nr * 4, hipMemcpyHostToDevice));
    auto fill_t = [&](T* dst, size_t n) {   // bf16 operands from the random table (repeated)
        for (size_t o = 0; o < n; o += nr) {
            const size_t c = std::min(nr, n - o);
            hipLaunchKernelGGL((cast_pad_kernel<T>), dim3(ew_blocks((long)c)), dim3(256), 0, s, rnd.p, (int)c, 1, (int)c, dst + o, (int)c, 1);
        }
    };
    fill_t(xn.p, (size_t)M * D); fill_t(ao.p, (size_t)M * D); fill_t(ffh.p, (size_t)M * F);
    fill_t(wqkv.p, (size_t)ncopy * 3 * D * D); fill_t(wout.p, (size_t)ncopy * D * D); fill_t(wff1.p, (size_t)ncopy * F * D); fill_t(wff2.p, (size_t)ncopy * D * F);
    HIPCHK(hipMemcpyAsync(bias.p, rnd.p, 3 * D * 4, hipMemcpyDeviceToDevice, s));
    HIPCHK(hipMemcpyAsync(gate.p, rnd.p + 4096, (size_t)(M / N) * D * 4, hipMemcpyDeviceToDevice, s));
    HIPCHK(hipMemcpyAsync(rope.p, rnd.p + 65536, (size_t)2 * N * 32 * 4, hipMemcpyDeviceToDevice, s));
    HIPCHK(hipMemsetAsync(x.p, 0, (size_t)M * D * 4, s));
    KCHK();
    hipEvent_t ev[5];
    for (auto& e : ev) HIPCHK(hipEventCreate(&e));
    double acc[4] = {0, 0, 0, 0};
    for (int it = -2; it < iters; ++it) {
        const int c = (it + 2) % ncopy;
        HIPCHK(hipEventRecord(ev[0], s));
        HIPCHK(launch_gemm<T>(s, xn.p, D, wqkv.p + (size_t)c * 3 * D * D, D, M, 3 * D, D,
                              EpiQKV<T>{q.p, k.p, vt.p, bias.p, rope.p, N, N, H, PE, 0.18f, nullptr}, cfg));
        HIPCHK(hipEventRecord(ev[1], s));
        HIPCHK(launch_gemm<T>(s, ao.p, D, wout.p + (size_t)c * D * D, D, M, D, D, EpiGateRes{x.p, x.p, D, bias.p, gate.p, D, N, nullptr}, cfg));
        HIPCHK(hipEventRecord(ev[2], s));
        HIPCHK(launch_gemm<T>(s, xn.p, D, wff1.p + (size_t)c * F * D, D, M, F, D, EpiStore<T>{ffh.p, F, bias.p, F5_ACT_GELU_TANH}, cfg));
        HIPCHK(hipEventRecord(ev[3], s));
        HIPCHK(launch_gemm<T>(s, ffh.p, F, wff2.p + (size_t)c * D * F, F, M, D, F, EpiGateRes{x.p, x.p, D, bias.p, gate.p, D, N, nullptr}, cfg));
        HIPCHK(hipEventRecord(ev[4], s));
        HIPCHK(hipEventSynchronize(ev[4]));
        if (it >= 0)
            for (int g = 0; g < 4; ++g) {
                float ms = 0.f;
                HIPCHK(hipEventElapsedTime(&ms, ev[g], ev[g + 1]));
                acc[g] += ms * 1000.0;
            }
    }
    for (int g = 0; g < 4; ++g) us4[g] = (float)(acc[g] / iters);
    for (auto& e : ev) (void)hipEventDestroy(e);
    return F5_OK;
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
        EpiQKV<T> e{q3.p, q3.p + (size_t)M * 1024, q3.p + (size_t)2 * M * 1024, tab.p + 2 * 4096 * 32, tab.p,
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

// diagnostic: what the STORE PATTERN of a GEMM epilogue costs.  256 x 256 output tiles, 8 waves as gemm3.h lays them out (wave tile
// 128 rows x 64 columns), every lane writes constants:
//   mode 0: bf16, MFMA-fragment order -- per instruction 16 rows x 32 contiguous bytes (what EpiStore<bf16> / EpiQKV do today)
//   mode 1: bf16, row-major -- per instruction 8 rows x 128 contiguous bytes (what an LDS-transposed epilogue would do)
//   mode 2: f32, fragment order -- 16 rows x 64 bytes (EpiGateRes: also READS the same pattern when `rd`)
//   mode 3: f32, row-major -- 4 rows x 256 bytes
__global__ __launch_bounds__(512) void store_pattern_kernel(void* out, int ld, int mode, int rd) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wr = wave >> 2, wc = wave & 3;
    const int l15 = lane & 15, g = lane >> 4;
    const size_t m0 = (size_t)blockIdx.y * 256 + wr * 128, n0 = (size_t)blockIdx.x * 256 + wc * 64;
    if (mode == 0) {
        unsigned short* o = reinterpret_cast<unsigned short*>(out);
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int i = 0; i < 8; ++i)
                *reinterpret_cast<uint2*>(o + (m0 + i * 16 + l15) * ld + n0 + j * 16 + g * 4) = make_uint2(0x3c003c00u + i, 0x3c003c00u + j);
    } else if (mode == 1) {
        unsigned short* o = reinterpret_cast<unsigned short*>(out);
#pragma unroll
        for (int p = 0; p < 16; ++p)
            *reinterpret_cast<uint4*>(o + (m0 + p * 8 + (lane >> 3)) * ld + n0 + (lane & 7) * 8) = make_uint4(0x3c003c00u + p, 1u, 2u, 3u);
    } else if (mode == 2) {
        float* o = reinterpret_cast<float*>(out);
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                float4* p = reinterpret_cast<float4*>(o + (m0 + i * 16 + l15) * ld + n0 + j * 16 + g * 4);
                float4 v = rd ? *p : make_float4(1.f, 2.f, 3.f, 4.f);
                v.x += 1.f; v.y += (float)i; v.z += (float)j; v.w += 1.f;
                *p = v;
            }
    } else {
        float* o = reinterpret_cast<float*>(out);
#pragma unroll
        for (int p = 0; p < 32; ++p) {
            float4* q = reinterpret_cast<float4*>(o + (m0 + p * 4 + (lane >> 4)) * ld + n0 + (lane & 15) * 4);
            float4 v = rd ? *q : make_float4(1.f, 2.f, 3.f, 4.f);
            v.x += 1.f; v.y += (float)p; v.z += 2.f; v.w += 1.f;
            *q = v;
        }
    }
}
extern "C" int f5x_store_pattern_probe(int32_t M, int32_t N, int32_t mode, int32_t rd, int32_t iters, float* avg_us, f5_stream stream) {
    if (M % 256 || N % 256 || mode < 0 || mode > 3 || iters <= 0 || !avg_us) return fail(F5_EINVAL, "f5x_store_pattern_probe: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    Scratch<float> buf;
    HIPCHK(buf.alloc((size_t)M * N));
    HIPCHK(hipMemsetAsync(buf.p, 0, (size_t)M * N * 4, s));
    hipEvent_t e0, e1;
    HIPCHK(hipEventCreate(&e0));
    HIPCHK(hipEventCreate(&e1));
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(store_pattern_kernel, dim3(N / 256, M / 256), dim3(512), 0, s, buf.p, N, mode, rd);
    HIPCHK(hipEventRecord(e0, s));
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(store_pattern_kernel, dim3(N / 256, M / 256), dim3(512), 0, s, buf.p, N, mode, rd);
    HIPCHK(hipEventRecord(e1, s));
    HIPCHK(hipEventSynchronize(e1));
    KCHK();
    float ms = 0.f;
    HIPCHK(hipEventElapsedTime(&ms, e0, e1));
    *avg_us = ms * 1000.f / iters;
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    return F5_OK;
}
