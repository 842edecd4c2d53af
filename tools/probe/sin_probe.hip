// Accuracy of the hardware v_sin_f32 (input in revolutions) against double sin over |x| <= R radians.  hipcc -> build/sin_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>
__global__ void k(const float* x, float* y, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) y[i] = __builtin_amdgcn_sinf(x[i] * 0.15915494309189535f);
}
int main() {
    const int n = 1 << 22;
    for (float R : {3.2f, 20.f, 200.f, 1500.f}) {
        std::vector<float> h(n), o(n);
        unsigned s = 7;
        for (int i = 0; i < n; ++i) { s = s * 1664525u + 1013904223u; h[i] = ((s >> 8) / 8388608.0f - 1.0f) * R; }
        float *dx, *dy;
        hipMalloc(&dx, n * 4); hipMalloc(&dy, n * 4);
        hipMemcpy(dx, h.data(), n * 4, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dx, dy, n);
        hipMemcpy(o.data(), dy, n * 4, hipMemcpyDeviceToHost);
        double e = 0;
        for (int i = 0; i < n; ++i) e = std::fmax(e, std::fabs((double)o[i] - std::sin((double)h[i])));
        printf("|x| <= %6.1f: max abs error %.3e\n", R, e);
        hipFree(dx); hipFree(dy);
    }
    return 0;
}
