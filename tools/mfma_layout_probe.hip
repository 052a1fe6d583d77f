// layout probe for v_mfma_f32_16x16x32_bf16: D[m][n] = sum_k A[m][k] B[k][n]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <cmath>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
__device__ __forceinline__ __bf16 tobf(float x) { return (__bf16)x; }
__global__ void probe(const float *A, const float *B, float *D) {   // A[16][32], B[32][16] row-major, D[16][16]
    const int l = threadIdx.x;
    bf16x8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = tobf(A[(l & 15) * 32 + 8 * (l >> 4) + j]); b[j] = tobf(B[(8 * (l >> 4) + j) * 16 + (l & 15)]); }
    f32x4 c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) D[((l >> 4) * 4 + r) * 16 + (l & 15)] = c[r];
}
int main() {
    std::vector<float> A(16 * 32), B(32 * 16), D(256), R(256, 0.f);
    for (int i = 0; i < 16; ++i) for (int k = 0; k < 32; ++k) A[i * 32 + k] = (float)((i * 7 + k * 3) % 11 - 5);
    for (int k = 0; k < 32; ++k) for (int j = 0; j < 16; ++j) B[k * 16 + j] = (float)((k * 5 + j * 13) % 7 - 3);
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) for (int k = 0; k < 32; ++k) R[i * 16 + j] += A[i * 32 + k] * B[k * 16 + j];
    float *dA, *dB, *dD;
    hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dD, 256 * 4);
    hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, dA, dB, dD);
    hipMemcpy(D.data(), dD, 256 * 4, hipMemcpyDeviceToHost);
    double err = 0; for (int i = 0; i < 256; ++i) err = fmax(err, fabs(D[i] - R[i]));
    printf("mfma_f32_16x16x32_bf16 layout probe: max err %g (%s)\n", err, err == 0 ? "layout OK" : "LAYOUT WRONG");
    return err == 0 ? 0 : 1;
}
