// launch-rate floor on this box: back-to-back launches of (a) an empty kernel, (b) a 128x256-thread kernel that touches memory
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
__global__ void k_empty() {}
__global__ void k_touch(float *p) { p[blockIdx.x * blockDim.x + threadIdx.x] += 1.0f; }
int main() {
    float *d; hipMalloc(&d, 1 << 20); hipMemset(d, 0, 1 << 20);
    hipStream_t s; hipStreamCreate(&s);
    for (int mode = 0; mode < 2; ++mode) {
        for (int w = 0; w < 200; ++w) { if (mode) hipLaunchKernelGGL(k_touch, dim3(128), dim3(256), 0, s, d); else hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, s); }
        hipStreamSynchronize(s);
        auto t0 = std::chrono::steady_clock::now();
        const int n = 20000;
        for (int w = 0; w < n; ++w) { if (mode) hipLaunchKernelGGL(k_touch, dim3(128), dim3(256), 0, s, d); else hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, s); }
        hipStreamSynchronize(s);
        double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / n;
        printf("%s: %.2f us per launch\n", mode ? "touch 128x256" : "empty", us);
    }
    return 0;
}
