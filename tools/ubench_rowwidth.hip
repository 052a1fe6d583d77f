// What one wave pays to fetch and to put back its 11 KB state block as 45 dword rows (the wave-blocked layout: 256 contiguous bytes per
// instruction) against 11-12 dwordx4 pieces (16 bytes per lane: 1024 contiguous bytes per instruction), in the latency regime of the headline:
// 128 workgroups of 512 threads, wave 0 alone touches the block, launches back to back so that every launch starts from written-back lines.
// Stamps (s_memtime, shader clocks): entry -> last load issued -> data there -> (200 dependent FMAs) -> last store issued; plus the launch
// duration from HIP events.
// build: hipcc --offload-arch=gfx950 -O3 -o tools/ubench_rowwidth tools/ubench_rowwidth.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

constexpr int ROWS = 44;              // 44 dword rows = 11 x 16 bytes per lane
constexpr int BLOCK_DW = ROWS * 64;   // dwords of a block

__device__ __forceinline__ unsigned long long now() { unsigned long long t; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory"); return t; }

template <int WIDE>
__global__ void __launch_bounds__(512) k_block(float *state, unsigned long long *stamps, float a) {
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    __shared__ float sh[64];
    if (wv == 0) {
        float *blk = state + (size_t)blockIdx.x * BLOCK_DW;
        const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void *)blk, 0, BLOCK_DW * 4, 0x00020000);
        float v[ROWS];
        const unsigned long long t0 = now();
        if (WIDE) {
#pragma unroll
            for (int g = 0; g < ROWS / 4; ++g) {
                typedef unsigned int u4 __attribute__((ext_vector_type(4)));
                const u4 x = __builtin_amdgcn_raw_buffer_load_b128(r, lane * 16, g * 1024, 0);
                v[4 * g] = __uint_as_float(x.x); v[4 * g + 1] = __uint_as_float(x.y); v[4 * g + 2] = __uint_as_float(x.z); v[4 * g + 3] = __uint_as_float(x.w);
            }
        } else {
#pragma unroll
            for (int q = 0; q < ROWS; ++q) v[q] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, lane * 4, q * 256, 0));
        }
        asm volatile("" ::: "memory");
        const unsigned long long t1 = now();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned long long t2 = now();
        // a dependent chain that uses every value (the sub-steps' stand-in), so that the stores cannot be folded into the loads
        float acc = a;
#pragma unroll
        for (int q = 0; q < ROWS; ++q) { acc = fmaf(acc, 0.999f, v[q]); v[q] = fmaf(v[q], 0.5f, acc * 1e-3f); }
        asm volatile("" : "+v"(acc));
        const unsigned long long t3 = now();
        if (WIDE) {
#pragma unroll
            for (int g = 0; g < ROWS / 4; ++g) {
                typedef unsigned int u4 __attribute__((ext_vector_type(4)));
                u4 x = {__float_as_uint(v[4 * g]), __float_as_uint(v[4 * g + 1]), __float_as_uint(v[4 * g + 2]), __float_as_uint(v[4 * g + 3])};
                __builtin_amdgcn_raw_buffer_store_b128(x, r, lane * 16, g * 1024, 0);
            }
        } else {
#pragma unroll
            for (int q = 0; q < ROWS; ++q) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v[q]), r, lane * 4, q * 256, 0);
        }
        asm volatile("" ::: "memory");
        const unsigned long long t4 = now();
        sh[lane] = acc;
        if (lane == 0 && blockIdx.x < 128) {
            unsigned long long *s = stamps + blockIdx.x * 8;
            s[0] = t0; s[1] = t1; s[2] = t2; s[3] = t3; s[4] = t4;
        }
    }
    __syncthreads();
    if (sh[lane] == 12345.678f) state[0] = 0;   // (keeps the other waves honest; never true)
}

template <int WIDE> int run(const char *name, float *state, unsigned long long *stamps, int grid) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int K = 2000;
    for (int i = 0; i < 200; ++i) k_block<WIDE><<<grid, 512>>>(state, stamps, 0.5f);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int i = 0; i < K; ++i) k_block<WIDE><<<grid, 512>>>(state, stamps, 0.5f);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    // stamps of single launches (each synchronised: the launch before it has written the block back)
    std::vector<double> d(4, 0.0);
    const int R = 50;
    std::vector<unsigned long long> h(128 * 8);
    for (int i = 0; i < R; ++i) {
        k_block<WIDE><<<grid, 512>>>(state, stamps, 0.5f);
        k_block<WIDE><<<grid, 512>>>(state, stamps, 0.5f);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(h.data(), stamps, h.size() * 8, hipMemcpyDeviceToHost));
        for (int w = 0; w < std::min(grid, 128); ++w) for (int k = 0; k < 4; ++k) d[k] += (double)(h[w * 8 + k + 1] - h[w * 8 + k]) / (R * std::min(grid, 128));
    }
    printf("%-8s grid %4d: %.3f us per launch; clocks: issue loads %.0f, wait data %.0f, chain %.0f, issue stores %.0f\n", name, grid, ms * 1e3 / K, d[0], d[1], d[2], d[3]);
    return 0;
}

// ---- the throughput regime: 2^20 drones = 16384 blocks, one wave per block, every wave moves its block in and out.
// MODE 0: 44 dword rows; 1: the lane-major mix of the state arrays (12-byte pieces at a 12-byte lane pitch + 16-byte pieces); 2: 16-byte pieces only
typedef unsigned int u3 __attribute__((ext_vector_type(3)));
typedef unsigned int u4 __attribute__((ext_vector_type(4)));
template <int MODE>
__global__ void __launch_bounds__(64) k_stream(float *state, float a) {
    const int lane = threadIdx.x & 63;
    float *blk = state + (size_t)blockIdx.x * BLOCK_DW;
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void *)blk, 0, BLOCK_DW * 4, 0x00020000);
    unsigned v[ROWS];
    if (MODE == 0) {
#pragma unroll
        for (int q = 0; q < ROWS; ++q) v[q] = __builtin_amdgcn_raw_buffer_load_b32(r, lane * 4, q * 256, 0);
    } else if (MODE == 1) {   // 4 x (3 dwords) then 8 x (4 dwords)
#pragma unroll
        for (int g = 0; g < 4; ++g) { const u3 x = __builtin_amdgcn_raw_buffer_load_b96(r, lane * 12, g * 768, 0); v[3 * g] = x.x; v[3 * g + 1] = x.y; v[3 * g + 2] = x.z; }
#pragma unroll
        for (int g = 0; g < 8; ++g) { const u4 x = __builtin_amdgcn_raw_buffer_load_b128(r, lane * 16, 3072 + g * 1024, 0); v[12 + 4 * g] = x.x; v[13 + 4 * g] = x.y; v[14 + 4 * g] = x.z; v[15 + 4 * g] = x.w; }
    } else {
#pragma unroll
        for (int g = 0; g < ROWS / 4; ++g) { const u4 x = __builtin_amdgcn_raw_buffer_load_b128(r, lane * 16, g * 1024, 0); v[4 * g] = x.x; v[4 * g + 1] = x.y; v[4 * g + 2] = x.z; v[4 * g + 3] = x.w; }
    }
    float acc = a;
#pragma unroll
    for (int q = 0; q < ROWS; ++q) { acc = fmaf(acc, 0.999f, __uint_as_float(v[q])); v[q] = __float_as_uint(fmaf(__uint_as_float(v[q]), 0.5f, acc * 1e-3f)); }
    if (MODE == 0) {
#pragma unroll
        for (int q = 0; q < ROWS; ++q) __builtin_amdgcn_raw_buffer_store_b32(v[q], r, lane * 4, q * 256, 0);
    } else if (MODE == 1) {
#pragma unroll
        for (int g = 0; g < 4; ++g) { const u3 x = {v[3 * g], v[3 * g + 1], v[3 * g + 2]}; __builtin_amdgcn_raw_buffer_store_b96(x, r, lane * 12, g * 768, 0); }
#pragma unroll
        for (int g = 0; g < 8; ++g) { const u4 x = {v[12 + 4 * g], v[13 + 4 * g], v[14 + 4 * g], v[15 + 4 * g]}; __builtin_amdgcn_raw_buffer_store_b128(x, r, lane * 16, 3072 + g * 1024, 0); }
    } else {
#pragma unroll
        for (int g = 0; g < ROWS / 4; ++g) { const u4 x = {v[4 * g], v[4 * g + 1], v[4 * g + 2], v[4 * g + 3]}; __builtin_amdgcn_raw_buffer_store_b128(x, r, lane * 16, g * 1024, 0); }
    }
}
template <int MODE> int stream(const char *name, float *state, int grid) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int K = 300;
    for (int i = 0; i < 50; ++i) k_stream<MODE><<<grid, 64>>>(state, 0.5f);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int i = 0; i < K; ++i) k_stream<MODE><<<grid, 64>>>(state, 0.5f);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1e3 / K, gb = 2.0 * grid * BLOCK_DW * 4 / 1e9;
    printf("stream %-22s grid %6d: %.2f us per launch, %.2f TB/s (read + write)\n", name, grid, us, gb / us * 1e6 / 1e3);
    return 0;
}

int main() {
    {
        float *big; const int G = 16384;
        CK(hipMalloc(&big, (size_t)G * BLOCK_DW * 4)); CK(hipMemset(big, 0, (size_t)G * BLOCK_DW * 4));
        for (int rep = 0; rep < 3; ++rep) {
            if (stream<0>("44 x dword rows", big, G)) return 1;
            if (stream<1>("4 x 12 B + 8 x 16 B", big, G)) return 1;
            if (stream<2>("11 x 16 B", big, G)) return 1;
        }
        CK(hipFree(big));
    }
    float *state; unsigned long long *stamps;
    const int G = 1024;
    CK(hipMalloc(&state, (size_t)G * BLOCK_DW * 4)); CK(hipMemset(state, 0, (size_t)G * BLOCK_DW * 4));
    CK(hipMalloc(&stamps, 128 * 8 * 8)); CK(hipMemset(stamps, 0, 128 * 8 * 8));
    for (int rep = 0; rep < 2; ++rep)
        for (int grid : {128, 512}) {
            if (run<0>("dword", state, stamps, grid)) return 1;
            if (run<1>("dwordx4", state, stamps, grid)) return 1;
        }
    return 0;
}
