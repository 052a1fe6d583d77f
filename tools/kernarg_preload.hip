// What the kernel-argument fetch costs at the head of a small kernel on this box, and what kernarg preloading (scalar leading
// arguments delivered in SGPRs at wave launch, -mllvm -amdgpu-kernarg-preload-count=N) buys: back-to-back launches of a
// 128 x 256-thread kernel whose first action is a load through a pointer argument, (a) pointer inside a by-value struct behind
// 700 bytes of other arguments (the shape of the stepper's Consts + Ptrs arguments), (b) pointer as the first scalar argument.
// Build twice: with and without the -mllvm flag (tools/kernarg_preload.sh).
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
struct Big { float pad[176]; float *p; unsigned more[30]; };
__global__ void k_struct(Big b) { b.p[blockIdx.x * blockDim.x + threadIdx.x] += 1.0f; }
__global__ void k_scalar(float *p, Big b) { p[blockIdx.x * blockDim.x + threadIdx.x] += 1.0f + b.pad[0] * 0.0f; }
int main() {
    float *d; hipMalloc(&d, 1 << 20); hipMemset(d, 0, 1 << 20);
    hipStream_t s; hipStreamCreate(&s);
    Big b = {}; b.p = d;
    for (int mode = 0; mode < 2; ++mode) {
        auto launch = [&]() { if (mode) hipLaunchKernelGGL(k_scalar, dim3(128), dim3(256), 0, s, d, b); else hipLaunchKernelGGL(k_struct, dim3(128), dim3(256), 0, s, b); };
        for (int w = 0; w < 500; ++w) launch();
        hipStreamSynchronize(s);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        const int n = 20000;
        hipEventRecord(e0, s);
        for (int w = 0; w < n; ++w) launch();
        hipEventRecord(e1, s);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%s: %.3f us per launch\n", mode ? "pointer as first scalar argument" : "pointer inside a by-value struct", ms * 1e3 / n);
    }
    return 0;
}
