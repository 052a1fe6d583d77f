// HBM ceilings for the stepper's access mix on MI355X: pure writes / copies / 20:80 read:write, wide (16 B per lane) and the
// stepper's own shapes (4 B per lane into ~60 component-major arrays; the same bytes as one contiguous block per wave).
// build: hipcc --offload-arch=gfx950 -O3 -o tools/ubench_hbm tools/ubench_hbm.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void k_write16(float4 *dst, size_t n4) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
    float4 v = make_float4(1.f, 2.f, 3.f, (float)i);
    for (; i < n4; i += stride) dst[i] = v;
}
__global__ void k_copy16(const float4 *src, float4 *dst, size_t n4) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n4; i += stride) dst[i] = src[i];
}
// reads n4/4 float4, writes n4 float4 (20 % read : 80 % write)
__global__ void k_mix16(const float4 *src, float4 *dst, size_t n4) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n4; i += stride) {
        float4 v = src[i];
        size_t o = i * 4;
        dst[o] = v; v.x += 1.f; dst[o + 1] = v; v.y += 1.f; dst[o + 2] = v; v.z += 1.f; dst[o + 3] = v;
    }
}
// one wave per block of 64 "drones": RC components read, WC components written, component-major (SoA: component c of drone g
// at base[c*T + g], 4 B per lane) or wave-blocked (AoSoA: the wave's RC/WC rows contiguous)
template <int RC, int WC, bool BLOCKED>
__global__ void __launch_bounds__(64) k_soa(const float *src, float *dst, size_t T) {
    const size_t w = blockIdx.x, lane = threadIdx.x, g = w * 64 + lane;
    float acc = 0.f;
#pragma unroll
    for (int c = 0; c < RC; ++c) acc += BLOCKED ? src[(w * RC + c) * 64 + lane] : src[(size_t)c * T + g];
#pragma unroll
    for (int c = 0; c < WC; ++c) { if (BLOCKED) dst[(w * WC + c) * 64 + lane] = acc + (float)c; else dst[(size_t)c * T + g] = acc + (float)c; }
}
// the stepper's shape: 34 component-major reads, 40 component-major writes (4 B per lane) + one contiguous 54*64*4-byte row block
template <bool BLOCKED>
__global__ void __launch_bounds__(64) k_stepper_like(const float *src, float *dst, float4 *obs, size_t T) {
    const size_t w = blockIdx.x, lane = threadIdx.x, g = w * 64 + lane;
    float acc = 0.f;
#pragma unroll
    for (int c = 0; c < 34; ++c) acc += BLOCKED ? src[(w * 34 + c) * 64 + lane] : src[(size_t)c * T + g];
#pragma unroll
    for (int c = 0; c < 40; ++c) { if (BLOCKED) dst[(w * 40 + c) * 64 + lane] = acc + (float)c; else dst[(size_t)c * T + g] = acc + (float)c; }
    float4 v = make_float4(acc, 1.f, 2.f, 3.f);
    for (int k = (int)lane; k < 54 * 64 / 4; k += 64) obs[w * (54 * 64 / 4) + k] = v;
}

int main() {
    const size_t bytes = 1ull << 30;   // 1 GiB buffers: past the 256 MiB Infinity Cache
    float4 *a, *b;
    CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, 4 * bytes / 4 + bytes));
    CK(hipMemset(a, 1, bytes)); CK(hipMemset(b, 0, bytes));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int reps = 20;
    auto report = [&](const char *name, double moved_bytes, float ms) { printf("%-64s %8.1f GB/s  (%.1f us per pass)\n", name, moved_bytes * reps / (ms * 1e-3) / 1e9, ms * 1e3 / reps); };
    float ms;
    const size_t n4 = bytes / 16;
    for (int blocks : {2048, 8192}) {
        k_write16<<<blocks, 256>>>(b, n4); CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0)); for (int r = 0; r < reps; ++r) k_write16<<<blocks, 256>>>(b, n4); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1)); char nm[96]; snprintf(nm, sizeof nm, "write-only, 16 B/lane, %d blocks", blocks); report(nm, (double)bytes, ms);
        CK(hipEventRecord(e0)); for (int r = 0; r < reps; ++r) k_copy16<<<blocks, 256>>>(a, b, n4); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1)); snprintf(nm, sizeof nm, "copy (50:50), 16 B/lane, %d blocks", blocks); report(nm, 2.0 * bytes, ms);
        CK(hipEventRecord(e0)); for (int r = 0; r < reps; ++r) k_mix16<<<blocks, 256>>>(a, b, n4 / 4); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1)); snprintf(nm, sizeof nm, "20:80 read:write, 16 B/lane, %d blocks", blocks); report(nm, 1.25 * bytes, ms);
    }
    const size_t T = 1u << 20;   // 2^20 drones = the 131072-env C2 batch
    const float *src = (const float *)a; float *dst = (float *)b;
    float4 *obs = (float4 *)((char *)b + 40 * T * 4);
#define RUN(NAME, KERNEL, MOVED) do { KERNEL; CK(hipDeviceSynchronize()); CK(hipEventRecord(e0)); for (int r = 0; r < reps; ++r) { KERNEL; } CK(hipEventRecord(e1)); \
        CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1)); report(NAME, (double)(MOVED), ms); } while (0)
    RUN("SoA 4 B/lane: 0 read + 40 written components", (k_soa<0, 40, false><<<T / 64, 64>>>(src, dst, T)), 40.0 * T * 4);
    RUN("wave-blocked 4 B/lane: 0 read + 40 written components", (k_soa<0, 40, true><<<T / 64, 64>>>(src, dst, T)), 40.0 * T * 4);
    RUN("SoA 4 B/lane: 34 read + 40 written components", (k_soa<34, 40, false><<<T / 64, 64>>>(src, dst, T)), 74.0 * T * 4);
    RUN("wave-blocked 4 B/lane: 34 read + 40 written components", (k_soa<34, 40, true><<<T / 64, 64>>>(src, dst, T)), 74.0 * T * 4);
    RUN("stepper-like: SoA 34 r + 40 w + 216 B/drone row block (512 B/drone)", (k_stepper_like<false><<<T / 64, 64>>>(src, dst, obs, T)), 128.0 * T * 4);
    RUN("stepper-like, wave-blocked state", (k_stepper_like<true><<<T / 64, 64>>>(src, dst, obs, T)), 128.0 * T * 4);
    return 0;
}
