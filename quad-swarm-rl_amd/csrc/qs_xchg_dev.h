// qs_xchg_dev.h - device-side types and helpers of the observation exchange (include/quadswarm_exchange.h), shared by the exchange's
// own kernels (qs_exchange.hip) and by the step kernels' fused epilogue (qs_step_team.inc).
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/quadswarm_exchange.h"

namespace qsx {

struct FlagWin {
    unsigned long long arrive[2][QS_XCHG_MAX_RANKS];   // written by rank s: arrive[slot][s] = seq of the rows now in slot
    unsigned long long ack[QS_XCHG_MAX_RANKS];         // written by rank c: ack[c] = last seq rank c has finished reading
};
struct Local {   // device memory of the owning rank only
    unsigned long long push_seq, wait_seq, release_seq;
    unsigned int ticket[QS_XCHG_MAX_RANKS], ticket_all, status;
};
// The fused form: what a step kernel needs to store its observation rows into every rank's window itself (qs_set_obs_exchange).
// Lives in device memory, owned by the endpoint.
struct XchgDev {
    char *data_win[QS_XCHG_MAX_RANKS];
    FlagWin *flag_win[QS_XCHG_MAX_RANKS];
    FlagWin *mine;
    Local *loc;
    long long n;                // elements per rank (rows * cols)
    long long slot_bytes;
    int world, rank, wire;
    int auto_ack;               // 1: the launch that pushed sequence number s also waits for s from every rank and releases it (no in-place reader)
    unsigned int blocks;        // workgroups of one step launch (set by qs_set_obs_exchange)
    unsigned long long timeout_ticks;
};

struct PushArgs {
    const float *src, *staging[2];
    char *data_win[QS_XCHG_MAX_RANKS];
    FlagWin *flag_win[QS_XCHG_MAX_RANKS];
    FlagWin *mine;
    Local *loc;
    long long n;                // elements per rank (rows * cols)
    long long slot_bytes;       // bytes of one slot of a data window = world * n * wire size
    int world, rank, wire;
    unsigned long long timeout_ticks;
};

__device__ __forceinline__ unsigned int f32_to_bf16_rne(float f) {
    unsigned int u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return 0x7fc0u;   // NaN (what torch's conversion produces)
    u += 0x7fffu + ((u >> 16) & 1u);
    return u >> 16;
}
// Flags: relaxed system-scope accesses (they bypass the caches; no fence per access - an acquire / release at system scope invalidates
// / writes back the whole L2 on this part).  What orders them against the rows: the rows are written through and drained
// (s_waitcnt vmcnt(0)) before a workgroup takes its ticket, the flag is stored by whoever takes the last ticket; readers of the rows
// are separate launches behind the wait (kernel-boundary acquire).
__device__ __forceinline__ unsigned long long ld_sys(const unsigned long long *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
__device__ __forceinline__ void st_sys(unsigned long long *p, unsigned long long v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }

// Write-through stores for the rows that go into a window (own or a peer's): `sc0 sc1` = system scope, the line leaves the L2 at once.
// With them a workgroup only has to wait for its own stores (s_waitcnt vmcnt(0)) before it takes its ticket - no system-scope release
// fence per workgroup, which on this part writes the whole L2 back and cost ~20 us per step when 128-256 workgroups each issued one
// (profiles/r03d_bench_lines.txt: 27.7 us per C2 step with the fences, 7.7 us without any exchange).
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void st16_wt(void *p, u32x4_t v) { asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory"); }
__device__ __forceinline__ void st4_wt(void *p, unsigned int v) { asm volatile("global_store_dword %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory"); }
__device__ __forceinline__ void st2_wt(void *p, unsigned int v) { asm volatile("global_store_short %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory"); }
__device__ __forceinline__ void wt_drain() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// bounded poll: true when *p >= want before the deadline
__device__ __forceinline__ bool poll_ge(const unsigned long long *p, unsigned long long want, unsigned long long timeout_ticks) {
    if (ld_sys(p) >= want) return true;
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < timeout_ticks) {
        if (ld_sys(p) >= want) return true;
        __builtin_amdgcn_s_sleep(8);
    }
    return ld_sys(p) >= want;
}


}   // namespace qsx
