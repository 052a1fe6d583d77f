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
__device__ __forceinline__ unsigned long long ld_sys(const unsigned long long *p) { return __hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM); }
__device__ __forceinline__ void st_sys(unsigned long long *p, unsigned long long v) { __hip_atomic_store(p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM); }

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
