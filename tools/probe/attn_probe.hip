// Cost of each part of attn_split_fwd_kernel by omission (DIAG bits, csrc/attn.h).  Standalone: hipcc -> build/attn_probe.
// usage: build/attn_probe [Bp=2] [H=16] [N=1024]
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "attn.h"
using namespace f5;

template <int QS, int NW, int DIAG>
static float run(const float* q, const float* k, const float* v, float* o, int Bp, int H, int N, int iters) {
    constexpr int smem = 2 * 4 * 64 * (128 + 16);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_split_fwd_kernel<QS, NW, DIAG>), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    dim3 grid(H * Bp, (N + NW * 16 * QS - 1) / (NW * 16 * QS));
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((attn_split_fwd_kernel<QS, NW, DIAG>), grid, dim3(NW * 64), smem, 0, q, k, v, o, H, N, N, nullptr, 1, nullptr, nullptr, 0);
    hipEventRecord(e0, 0);
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL((attn_split_fwd_kernel<QS, NW, DIAG>), grid, dim3(NW * 64), smem, 0, q, k, v, o, H, N, N, nullptr, 1, nullptr, nullptr, 0);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    return ms * 1000.0f / iters;
}

int main(int argc, char** argv) {
    const int Bp = argc > 1 ? atoi(argv[1]) : 2, H = argc > 2 ? atoi(argv[2]) : 16, N = argc > 3 ? atoi(argv[3]) : 1024;
    const size_t n = (size_t)Bp * H * N * 64;
    std::vector<float> h(n);
    unsigned x = 1;
    for (size_t i = 0; i < n; ++i) { x = x * 1664525u + 1013904223u; h[i] = ((x >> 8) & 0xFFFF) / 32768.0f - 1.0f; }
    float *q, *k, *v, *o;
    hipMalloc(&q, n * 4); hipMalloc(&k, n * 4); hipMalloc(&v, n * 4); hipMalloc(&o, n * 4);
    hipMemcpy(q, h.data(), n * 4, hipMemcpyHostToDevice); hipMemcpy(k, h.data(), n * 4, hipMemcpyHostToDevice);
    hipMemcpy(v, h.data(), n * 4, hipMemcpyHostToDevice);
    printf("Bp=%d H=%d N=%d  (us per launch)\n", Bp, H, N);
#define ROW(QS, NW, D, what) printf("QS=%d NW=%d %-44s %7.1f\n", QS, NW, what, run<QS, NW, D>(q, k, v, o, Bp, H, N, 50));
    ROW(2, 4, 0, "full")
    ROW(1, 4, 0, "full")
    ROW(1, 8, 0, "full")
    ROW(1, 8, 1, "- K/V split")
    ROW(1, 8, 2, "- softmax")
    ROW(1, 8, 4, "- PV mfma")
    ROW(1, 8, 8, "- S mfma")
    ROW(1, 8, 12, "- all mfma")
    ROW(1, 8, 16, "- global loads")
    ROW(1, 8, 31, "nothing but LDS reads / P split / barriers")
    return 0;
}
