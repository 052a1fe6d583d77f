// How long does the dispatcher take to START all workgroups of a launch, as a function of the launch's shape?  Every workgroup stamps the
// constant 100 MHz clock at entry, then spins ~6 us (so that none finishes before the last starts); the kernel is launched back to back.
// hipcc --offload-arch=gfx950 -O3 -o dispatch_spread dispatch_spread.hip && ./dispatch_spread
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <algorithm>
#include <vector>
template <int V>
__global__ void k(unsigned long long *starts, float *sink) {
    extern __shared__ float lds[];
    const unsigned long long t0 = wall_clock64();
    if (threadIdx.x == 0) starts[blockIdx.x] = t0;
    if (V == 1) asm volatile("v_mov_b32 v120, 0" ::: "v120");
    if (V == 2) asm volatile("v_mov_b32 v250, 0" ::: "v250");
    float x = threadIdx.x;
    while (wall_clock64() - t0 < 600) x = x * 1.0001f + 1.0f;
    if (x == 12345.0f) sink[0] = x + lds[threadIdx.x];
}
template <int V>
static void run(int grid, int block, int ldsb, unsigned long long *d, float *sink) {
    hipFuncSetAttribute((const void *)k<V>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    std::vector<unsigned long long> h(grid);
    std::vector<double> spreads, p50;
    for (int it = 0; it < 40; ++it) {
        hipLaunchKernelGGL(k<V>, dim3(grid), dim3(block), ldsb, 0, d, sink);
        hipLaunchKernelGGL(k<V>, dim3(grid), dim3(block), ldsb, 0, d, sink);
        hipDeviceSynchronize();
        hipMemcpy(h.data(), d, grid * 8, hipMemcpyDeviceToHost);
        std::sort(h.begin(), h.end());
        spreads.push_back((h.back() - h.front()) * 0.01);
        p50.push_back((h[grid / 2] - h.front()) * 0.01);
    }
    std::sort(spreads.begin(), spreads.end()); std::sort(p50.begin(), p50.end());
    printf("grid %4d x %4d threads, LDS %6d B, VGPRs %s: first -> last start median %.2f us (min %.2f), first -> median workgroup %.2f us\n", grid, block, ldsb,
           V == 0 ? "few" : (V == 1 ? ">=121" : ">=251"), spreads[20], spreads[0], p50[20]);
}
int main() {
    unsigned long long *d; float *sink;
    hipMalloc(&d, 8 * 4096); hipMalloc(&sink, 4096);
    const int grids[] = {128, 256, 1024}, blocks[] = {256, 512}, ldss[] = {0, 32768, 65536};
    for (int g : grids) for (int b : blocks) for (int l : ldss) {
        if (g == 1024 && (l > 32768 || b > 256)) continue;
        run<0>(g, b, l, d, sink);
        if (l == 32768) { run<1>(g, b, l, d, sink); if (g <= 256 && b == 256) run<2>(g, b, l, d, sink); }
    }
    return 0;
}
