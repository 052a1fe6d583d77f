// micro-benchmark: cycles per wave64 instruction for a lone wave on a SIMD (gfx950)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define REP 256
template <int MODE> __global__ void k(float *out, unsigned long long *cyc, float a0, uint32_t u0) {
    float a = a0 + threadIdx.x, b = a0 * 2 + threadIdx.x, c = a0 * 3, d = a0 * 5, e = a0 * 7, f = a0 * 11, g = a0 * 13, h = a0 * 17;
    uint32_t x = u0 + threadIdx.x, y = u0 * 3 + threadIdx.x, z = u0 * 5, w = u0 * 7;
    double da = a0, db = a0 * 3;
    unsigned long long t0 = clock64();
#pragma unroll 1
    for (int it = 0; it < 16; ++it) {
#pragma unroll
        for (int r = 0; r < REP; ++r) {
            if (MODE == 0) { a = __builtin_fmaf(a, b, c); }                                      // dependent fma chain
            if (MODE == 1) { a = __builtin_fmaf(a, b, c); d = __builtin_fmaf(d, b, c); e = __builtin_fmaf(e, b, c); f = __builtin_fmaf(f, b, c); }  // 4 indep
            if (MODE == 2) { uint64_t p = (uint64_t)x * 0xD2511F53u; x = (uint32_t)(p >> 32) ^ (uint32_t)p ^ y; }   // mad_u64 dependent
            if (MODE == 3) { x = x * 0xCD9E8D57u + y; }                                         // mul_lo
            if (MODE == 4) { a = __builtin_amdgcn_sqrtf(a) + b; }                                // trans dependent
            if (MODE == 5) { a = (a > b) ? c : a + d; }                                          // cmp+cndmask+add
            if (MODE == 6) { da = __builtin_fma(da, db, da); }                                   // f64 fma dependent
            if (MODE == 7) { x ^= y; y += x; z ^= w; w += z; }                                   // int alu 2 chains
            if (MODE == 8) { a = __builtin_fmaf(a, b, c); x = x * 0xCD9E8D57u + y; }             // mix
            if (MODE == 9) { a = a / b; }                                                        // precise div
        }
    }
    unsigned long long t1 = clock64();
    out[threadIdx.x + blockIdx.x * 64] = a + b + c + d + e + f + g + h + (float)(x + y + z + w) + (float)da;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
template <int MODE> void run(const char *name, int blocks, float ops_per_rep) {
    float *out; unsigned long long *cyc, hc;
    hipMalloc(&out, 64 * blocks * 4); hipMalloc(&cyc, 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64), 0, 0, out, cyc, 1.0001f, 12345u);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64), 0, 0, out, cyc, 1.0001f, 12345u);
    hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(&hc, cyc, 8, hipMemcpyDeviceToHost);
    printf("%-28s blocks=%5d  memtime ticks/inst = %6.2f   kernel %.1f us  (ns per inst per wave: %.2f)\n", name, blocks,
           (double)hc / (16.0 * REP * ops_per_rep), ms * 1e3, ms * 1e6 / (16.0 * REP * ops_per_rep));
    hipFree(out); hipFree(cyc);
}
int main() {
    for (int blocks : {128, 1024, 2048, 4096}) {
        run<0>("fma dependent", blocks, 1); run<1>("fma 4 independent", blocks, 4); run<2>("mad_u64+2xor dependent", blocks, 3);
        run<3>("mul_lo+add (mad_u32)", blocks, 1); run<4>("sqrt+add dependent", blocks, 2); run<5>("cmp+cndmask+add", blocks, 3);
        run<6>("fma f64 dependent", blocks, 1); run<7>("xor/add 2 chains", blocks, 4); run<8>("fma + mad_u32", blocks, 2); run<9>("f32 precise div", blocks, 1);
        printf("\n");
    }
    return 0;
}
