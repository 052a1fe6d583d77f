// HBM ceilings for the stepper's access mix on MI355X: pure writes / copies / 20:80 read:write, wide (16 B per lane) and the
// stepper's own shapes (4 B per lane into ~60 component-major arrays; the same bytes as one contiguous block per wave).
// build: hipcc --offload-arch=gfx950 -O3 -o tools/ubench_hbm tools/ubench_hbm.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
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
// ---- the copy ceiling, swept: size x grid x unroll x cache policy.  /opt/skills/guides/MI355X_MICROARCH.md quotes 6.29 TB/s for a float4
// copy; the naive loop above reaches 4.6-4.8 TB/s at 1 GiB.  What separates the two is looked for here, not assumed. ----
typedef float vf4 __attribute__((ext_vector_type(4)));   // (the nontemporal builtins take clang vectors, not HIP's float4 struct)
template <int NT> __device__ __forceinline__ vf4 ld4(const vf4 *p) { if (NT & 1) return __builtin_nontemporal_load(p); return *p; }
template <int NT> __device__ __forceinline__ void st4(vf4 *p, vf4 v) { if (NT & 2) __builtin_nontemporal_store(v, p); else *p = v; }
template <int U, int NT>
__global__ void __launch_bounds__(256) k_copy_sweep(const float4 *__restrict__ src_, float4 *__restrict__ dst_, size_t n4) {
    const vf4 *__restrict__ src = (const vf4 *)src_; vf4 *__restrict__ dst = (vf4 *)dst_;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    for (; i + (U - 1) * stride < n4; i += U * stride) {
        vf4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = ld4<NT>(&src[i + u * stride]);
#pragma unroll
        for (int u = 0; u < U; ++u) st4<NT>(&dst[i + u * stride], v[u]);
    }
    for (; i < n4; i += stride) dst[i] = src[i];
}
template <int U, int NT>
__global__ void __launch_bounds__(256) k_write_sweep(float4 *__restrict__ dst_, size_t n4) {
    vf4 *__restrict__ dst = (vf4 *)dst_;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    const vf4 v = {1.f, 2.f, 3.f, (float)i};
    for (; i + (U - 1) * stride < n4; i += U * stride) {
#pragma unroll
        for (int u = 0; u < U; ++u) st4<NT>(&dst[i + u * stride], v);
    }
    for (; i < n4; i += stride) dst[i] = v;
}
template <int U, int NT>
__global__ void __launch_bounds__(256) k_read_sweep(const float4 *__restrict__ src_, float4 *__restrict__ sink, size_t n4) {
    const vf4 *__restrict__ src = (const vf4 *)src_;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    vf4 acc = {0.f, 0.f, 0.f, 0.f};
    for (; i + (U - 1) * stride < n4; i += U * stride) {
#pragma unroll
        for (int u = 0; u < U; ++u) acc += ld4<NT>(&src[i + u * stride]);
    }
    if (acc.x == 123.456f) *(vf4 *)sink = acc;   // never true: keeps the loads
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

// ---- counter calibration (`ubench_hbm calib`, run under rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE): kernels that move a KNOWN byte count
// with the stepper's own access pattern - one wave per block, every instruction one 256-byte row (4 B per lane) of the wave's contiguous
// block through a buffer resource, exactly what BufRow::ld / st emit - next to 16-byte-per-lane streams (the guide's calibrated case:
// FETCH_SIZE = 1/2 of the bytes).  Working sets > 512 MB, so nothing is served from the L2 / Infinity Cache of a previous launch. ----
template <int RC>
__global__ void __launch_bounds__(64) k_calib_read4(const float *src, float *sink, uint32_t bytes) {
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void *)src, 0, bytes, 0x00020000);
    const uint32_t blk = blockIdx.x * (uint32_t)(RC * 256), lane = threadIdx.x * 4;
    float acc = 0.f;
#pragma unroll
    for (int c = 0; c < RC; ++c) acc += __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, lane, blk + (uint32_t)c * 256, 0));
    if (acc == 123.456f) *sink = acc;
}
template <int WC>
__global__ void __launch_bounds__(64) k_calib_write4(float *dst, uint32_t bytes) {
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void *)dst, 0, bytes, 0x00020000);
    const uint32_t blk = blockIdx.x * (uint32_t)(WC * 256), lane = threadIdx.x * 4;
#pragma unroll
    for (int c = 0; c < WC; ++c) __builtin_amdgcn_raw_buffer_store_b32((uint32_t)(blockIdx.x + c), r, lane, blk + (uint32_t)c * 256, 0);
}
// the observation rows' store: 16 B per lane, non-temporal, one contiguous 54 x 64 x 4-byte block per wave
__global__ void __launch_bounds__(64) k_calib_write16nt(float4 *dst) {
    vf4 v = {1.f, 2.f, 3.f, (float)blockIdx.x};
    for (int k = threadIdx.x; k < 54 * 64 / 4; k += 64) __builtin_nontemporal_store(v, (vf4 *)dst + (size_t)blockIdx.x * (54 * 64 / 4) + k);
}
__global__ void __launch_bounds__(256) k_calib_read16(const float4 *src_, float4 *sink, size_t n4) {
    const vf4 *src = (const vf4 *)src_;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    vf4 acc = {0.f, 0.f, 0.f, 0.f};
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += stride) acc += src[i];
    if (acc.x == 123.456f) *(vf4 *)sink = acc;
}
__global__ void __launch_bounds__(256) k_calib_write16(float4 *dst_, size_t n4) {
    vf4 *dst = (vf4 *)dst_;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    const vf4 v = {1.f, 2.f, 3.f, 4.f};
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += stride) dst[i] = v;
}
static int calib() {
    const size_t T = 1u << 22, waves = T / 64;                       // 2^22 "drones" = 65536 waves
    const size_t rbytes = (size_t)47 * T * 4, wbytes = (size_t)54 * T * 4;   // 47 state rows read / 54 rows written (and the 54-column obs block)
    float *a, *b; float4 *o;
    CK(hipMalloc(&a, rbytes)); CK(hipMalloc(&b, wbytes)); CK(hipMalloc(&o, wbytes));
    CK(hipMemset(a, 1, rbytes)); CK(hipMemset(b, 0, wbytes)); CK(hipMemset(o, 0, wbytes));
    printf("# known bytes per launch: k_calib_read4<47> %zu | k_calib_read4<8> %zu | k_calib_write4<54> %zu | k_calib_write4<8> %zu | k_calib_write16nt %zu | k_calib_read16 %zu | k_calib_write16 %zu\n",
           rbytes, (size_t)8 * T * 4, wbytes, (size_t)8 * T * 4, wbytes, rbytes, wbytes);
    for (int rep = 0; rep < 4; ++rep) {
        k_calib_read4<47><<<waves, 64>>>(a, b, (uint32_t)rbytes);
        k_calib_read4<8><<<waves, 64>>>(a, b, (uint32_t)rbytes);    // a short row run per wave (8 rows = 2 KB): partial use of the prefetched lines?
        k_calib_write4<54><<<waves, 64>>>(b, (uint32_t)wbytes);
        k_calib_write4<8><<<waves, 64>>>(b, (uint32_t)wbytes);
        k_calib_write16nt<<<waves, 64>>>(o);
        k_calib_read16<<<2048, 256>>>((const float4 *)a, o, rbytes / 16);
        k_calib_write16<<<2048, 256>>>(o, wbytes / 16);
        CK(hipDeviceSynchronize());
    }
    return 0;
}

int main(int argc, char **argv) {
    if (argc > 1 && !strcmp(argv[1], "calib")) return calib();
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
    {   // ---- sweep: bytes per buffer x blocks x unroll x policy (bit 0 = nt loads, bit 1 = nt stores); best line per size printed last ----
        const size_t cap = 4ull << 30;
        float4 *s2 = nullptr, *d2 = nullptr;
        if (hipMalloc(&s2, cap) == hipSuccess && hipMalloc(&d2, cap) == hipSuccess) {
            CK(hipMemset(s2, 1, cap)); CK(hipMemset(d2, 0, cap));
            printf("# copy / write / read sweep: GB/s of bytes moved (copy counts read + written); 256 threads per block\n");
            for (size_t mb : {64, 256, 1024, 4096}) {
                const size_t nb = mb << 20, m4 = nb / 16;
                double best_copy = 0, best_write = 0, best_read = 0; char bc[96] = "", bw[96] = "", br[96] = "";
                for (int blocks : {1024, 2048, 4096, 8192, 16384}) {
#define SWEEP(U, NT) do { \
                    float t; char nm[96]; \
                    k_copy_sweep<U, NT><<<blocks, 256>>>(s2, d2, m4); CK(hipDeviceSynchronize()); \
                    CK(hipEventRecord(e0)); for (int r = 0; r < reps; ++r) k_copy_sweep<U, NT><<<blocks, 256>>>(s2, d2, m4); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); \
                    CK(hipEventElapsedTime(&t, e0, e1)); { double g = 2.0 * nb * reps / (t * 1e-3) / 1e9; snprintf(nm, sizeof nm, "%d blocks, unroll %d, policy %d", blocks, U, NT); \
                        printf("  copy  %5zu MiB %-40s %8.1f GB/s\n", mb, nm, g); if (g > best_copy) { best_copy = g; snprintf(bc, sizeof bc, "%s", nm); } } \
                    if (((NT) & 1) == 0) { CK(hipEventRecord(e0)); for (int r = 0; r < reps; ++r) k_write_sweep<U, NT><<<blocks, 256>>>(d2, m4); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); \
                        CK(hipEventElapsedTime(&t, e0, e1)); double g = 1.0 * nb * reps / (t * 1e-3) / 1e9; if (g > best_write) { best_write = g; snprintf(bw, sizeof bw, "%s", nm); } } \
                    if (((NT) & 2) == 0) { CK(hipEventRecord(e0)); for (int r = 0; r < reps; ++r) k_read_sweep<U, NT><<<blocks, 256>>>(s2, d2, m4); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); \
                        CK(hipEventElapsedTime(&t, e0, e1)); double g = 1.0 * nb * reps / (t * 1e-3) / 1e9; if (g > best_read) { best_read = g; snprintf(br, sizeof br, "%s", nm); } } } while (0)
                    SWEEP(1, 0); SWEEP(4, 0); SWEEP(8, 0); SWEEP(4, 1); SWEEP(4, 2); SWEEP(4, 3); SWEEP(8, 3);
#undef SWEEP
                }
                printf("BEST %5zu MiB per buffer: copy %8.1f GB/s [%s] | write-only %8.1f GB/s [%s] | read-only %8.1f GB/s [%s]\n", mb, best_copy, bc, best_write, bw, best_read, br);
            }
            CK(hipFree(s2)); CK(hipFree(d2));
        } else { (void)hipGetLastError(); printf("# sweep skipped: cannot allocate 2 x 4 GiB\n"); }
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
