// VALU issue cost of the tanh epilogue's instructions on gfx950 (one wave alone on a SIMD): cycles per wave64 instruction, 8 independent chains
// hipcc --offload-arch=gfx950 -O3 -o ubench_valu ubench_valu.hip && ./ubench_valu
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(2))) float f32x2;
#define REP 64
#define BODY(name, asmline)                                                                                       \
    __global__ void k_##name(float *out, unsigned long long *cyc) {                                               \
        float v0 = threadIdx.x * 0.001f + 0.1f, v1 = v0 + 1, v2 = v0 + 2, v3 = v0 + 3, v4 = v0 + 4, v5 = v0 + 5, v6 = v0 + 6, v7 = v0 + 7; \
        f32x2 p0 = {v0, v1}, p1 = {v2, v3}, p2 = {v4, v5}, p3 = {v6, v7};                                          \
        unsigned long long t0 = __builtin_readcyclecounter();                                                    \
        for (int i = 0; i < REP; ++i) { asmline }                                                                 \
        unsigned long long t1 = __builtin_readcyclecounter();                                                    \
        out[threadIdx.x] = v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y;  \
        if (threadIdx.x == 0) cyc[0] = t1 - t0;                                                                   \
    }
#define A8(op) asm volatile(op " %0, %0\n" op " %1, %1\n" op " %2, %2\n" op " %3, %3\n" op " %4, %4\n" op " %5, %5\n" op " %6, %6\n" op " %7, %7" \
                            : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7));
#define F8(op) asm volatile(op " %0, %0, %0, %0\n" op " %1, %1, %1, %1\n" op " %2, %2, %2, %2\n" op " %3, %3, %3, %3\n" op " %4, %4, %4, %4\n" op " %5, %5, %5, %5\n" op " %6, %6, %6, %6\n" op " %7, %7, %7, %7" \
                            : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7));
#define P4(op) asm volatile(op " %0, %0, %0, %0\n" op " %1, %1, %1, %1\n" op " %2, %2, %2, %2\n" op " %3, %3, %3, %3\n" op " %0, %0, %0, %0\n" op " %1, %1, %1, %1\n" op " %2, %2, %2, %2\n" op " %3, %3, %3, %3" \
                            : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3));
#define H8(op) asm volatile(op " %0, %0, %0, %0\n" op " %1, %1, %1, %1\n" op " %2, %2, %2, %2\n" op " %3, %3, %3, %3\n" op " %4, %4, %4, %4\n" op " %5, %5, %5, %5\n" op " %6, %6, %6, %6\n" op " %7, %7, %7, %7" \
                            : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7));
#define C8(op) asm volatile(op " %0, %0, %1\n" op " %1, %1, %2\n" op " %2, %2, %3\n" op " %3, %3, %4\n" op " %4, %4, %5\n" op " %5, %5, %6\n" op " %6, %6, %7\n" op " %7, %7, %0" \
                            : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7));
BODY(exp_f32, A8("v_exp_f32"))
BODY(rcp_f32, A8("v_rcp_f32"))
BODY(exp_f16, A8("v_exp_f16"))
BODY(rcp_f16, A8("v_rcp_f16"))
BODY(fma_f32, F8("v_fma_f32"))
BODY(pk_fma_f32, P4("v_pk_fma_f32"))
BODY(pk_fma_f16, H8("v_pk_fma_f16"))
BODY(cvt_pk_bf16, C8("v_cvt_pk_bf16_f32"))
BODY(mov, A8("v_mov_b32"))
#define RUN(name) do { hipLaunchKernelGGL(k_##name, dim3(1), dim3(64 * waves), 0, 0, out, cyc); hipDeviceSynchronize(); unsigned long long c; \
        hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost); printf("%-14s %d wave(s)/SIMD: %.2f cycles per instruction (s_memtime ticks / %d)\n", #name, (waves + 3) / 4, (double)c / (REP * 8), REP * 8); } while (0)
int main() {
    float *out; unsigned long long *cyc;
    hipMalloc(&out, 4096 * 4); hipMalloc(&cyc, 8);
    for (int waves = 1; waves <= 8; waves += 7) {   // 1 wave; 8 waves = 2 per SIMD (wave 0's count: its share)
        RUN(mov); RUN(fma_f32); RUN(pk_fma_f32); RUN(pk_fma_f16); RUN(exp_f32); RUN(rcp_f32); RUN(exp_f16); RUN(rcp_f16); RUN(cvt_pk_bf16);
    }
    return 0;
}
