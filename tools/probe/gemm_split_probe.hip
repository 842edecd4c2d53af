// Where the time of the split-operand GEMM (gemm2.h MODE 3) goes, by omission.  Standalone: hipcc -> build/gemm_split_probe.
// usage: build/gemm_split_probe [M=16384] [N=1024] [K=1024]
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "gemm_dispatch.h"
using namespace f5;

template <int BM, int BN, int WM, int WN, int NS, int MODE>
static float run(const float* A, const float* W, float* O, int M, int N, int K, int iters) {
    typedef EpiStore<float, 0> E;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    auto go = [&]() { return launch_gemm2_raw<float, BM, BN, WM, WN, NS, E, MODE>(0, A, K, W, K, M, N, K, E{O, N, nullptr, 0}); };
    for (int i = 0; i < 3; ++i) go();
    hipEventRecord(e0, 0);
    for (int i = 0; i < iters; ++i) go();
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    return ms * 1000.0f / iters;
}

int main(int argc, char** argv) {
    const int M = argc > 1 ? atoi(argv[1]) : 16384, N = argc > 2 ? atoi(argv[2]) : 1024, K = argc > 3 ? atoi(argv[3]) : 1024;
    std::vector<float> h((size_t)std::max(M, N) * K);
    unsigned x = 1;
    for (auto& v : h) { x = x * 1664525u + 1013904223u; v = ((x >> 8) & 0xFFFF) / 32768.0f - 1.0f; }
    float *A, *W, *O;
    hipMalloc(&A, (size_t)M * K * 4); hipMalloc(&W, (size_t)N * K * 4); hipMalloc(&O, (size_t)M * N * 4);
    hipMemcpy(A, h.data(), (size_t)M * K * 4, hipMemcpyHostToDevice); hipMemcpy(W, h.data(), (size_t)N * K * 4, hipMemcpyHostToDevice);
    const double gf = 2.0 * M * N * K * 1e-6;
    printf("M=%d N=%d K=%d (us, TFLOP/s as if all three terms ran)\n", M, N, K);
#define ROW(BM, BN, WM, WN, NS, MODE, what) { float us = run<BM, BN, WM, WN, NS, MODE>(A, W, O, M, N, K, 30); printf("%dx%d %-38s %8.1f  %7.1f\n", BM, BN, what, us, gf / us); }
    if (argc > 4) {   // tile sweep with both operands pre-split (MODE 5), as the engine runs the block GEMMs
        ROW(256, 128, 4, 2, 3, 5, "[13] pre-split")
        ROW(128, 192, 2, 4, 3, 5, "[10] pre-split")
        ROW(128, 128, 2, 4, 4, 5, "[2] pre-split")
        ROW(128, 64, 4, 2, 4, 5, "[9] pre-split")
        ROW(64, 64, 2, 2, 3, 5, "[8] pre-split")
        return 0;
    }
    ROW(256, 128, 4, 2, 3, 0, "f32 MFMA")
    ROW(256, 128, 4, 2, 3, 3, "split, full")
    ROW(256, 128, 4, 2, 3, 5, "split, A not converted")
    ROW(256, 128, 4, 2, 3, 6, "split, 2 of 3 MFMAs")
    ROW(256, 128, 4, 2, 3, 7, "split, 1 of 3 MFMAs")
    ROW(256, 128, 4, 2, 3, 1, "LDS-DMA only")
    ROW(128, 128, 2, 4, 4, 3, "split, full")
    ROW(128, 128, 2, 4, 4, 5, "split, A not converted")
    ROW(128, 128, 2, 4, 4, 7, "split, 1 of 3 MFMAs")
    ROW(128, 192, 2, 4, 3, 3, "split, full")
    ROW(128, 64, 4, 2, 4, 3, "split, full")
    ROW(128, 64, 4, 2, 4, 5, "split, A not converted")
    return 0;
}
