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
// QS_WIRE_Q8 row layout on the device (include/quadswarm_exchange.h): [bf16 x c16p][int8 x n8p]
struct Q8Dev {
    int D, q0, q1;        // columns; the int8 block
    int w16, row_words;   // 4-byte words of the bf16 section / of the whole row
    float s0, s1, s2, s3, s4, s5;   // 127 / clip of the six column kinds - named fields, not an array: a select over array elements is
                                    // turned into an indexed read of a private (scratch) copy by the compiler, ~10 us per C4 step
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
    // 1: the launch that pushed sequence number s also waits for s from every rank and releases it (no in-place reader)
    int auto_ack;
    unsigned int blocks;        // workgroups of one step launch (set by qs_set_obs_exchange)
    unsigned long long timeout_ticks;
    Q8Dev q8;                   // wire == QS_WIRE_Q8
    // QS_XCHG_FENCED=1: a system-scope release fence in front of every flag store, an acquire fence behind every flag wait
    int fenced;
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
    Q8Dev q8;
    long long rows;             // rows per rank (wire == QS_WIRE_Q8: n = rows * q8.D)
    int fenced;
};

__device__ __forceinline__ unsigned int f32_to_bf16_rne(float f) {
    unsigned int u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return 0x7fc0u;   // NaN (what torch's conversion produces)
    u += 0x7fffu + ((u >> 16) & 1u);
    return u >> 16;
}
// (every layout value by VALUE: with the struct passed by reference the compiler turns the select over its adjacent scale fields into an
// indexed load, which pins the whole struct in scratch memory - 28-48 bytes of private segment and ~10 us per C4 step in the fused
// epilogue)
__device__ __forceinline__ float q8_scale_of(int a, float s0, float s1, float s2, float s3, float s4, float s5) {
    float sc = s0;
    sc = a == 1 ? s1 : sc; sc = a == 2 ? s2 : sc; sc = a == 3 ? s3 : sc; sc = a == 4 ? s4 : sc; sc = a == 5 ? s5 : sc;
    return sc;
}
__device__ __forceinline__ float q8_scale(const Q8Dev &q, int a) { return q8_scale_of(a, q.s0, q.s1, q.s2, q.s3, q.s4, q.s5); }
// QS_WIRE_Q8: 4-byte word j of the wire row of the float32 row `row` (LDS or global memory)
__device__ __forceinline__ unsigned int q8_byte(float x, float scale) {
    const float v = __builtin_rintf(x * scale);                       // round half to even (v_rndne_f32)
    // (NaN -> -127 by the min / max order: not a value the env produces)
    const int q = (int)__builtin_fminf(__builtin_fmaxf(v, -127.0f), 127.0f);
    return (unsigned int)q & 0xffu;
}
__device__ __forceinline__ unsigned int q8_row_word_v(const float *row, int D, int q0, int q1, int w16, float s0, float s1, float s2,
    float s3, float s4, float s5, int j) {
    const int nq = q1 - q0;
    if (j < w16) {
        const int b0 = 2 * j, b1 = b0 + 1;
        const int c0 = b0 < q0 ? b0 : b0 + nq, c1 = b1 < q0 ? b1 : b1 + nq;
        const unsigned int lo = c0 < D ? f32_to_bf16_rne(row[c0]) : 0u, hi = c1 < D ? f32_to_bf16_rne(row[c1]) : 0u;
        return lo | (hi << 16);
    }
    const int m = 4 * (j - w16);
    // = 0, 4 or 2: the column kind (3 relative-position, 3 relative-velocity columns per neighbour) of the word's first byte
    const int a0 = m % 6;
    unsigned int o = 0;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int mm = m + u;
        const int a = (a0 + u >= 6) ? a0 + u - 6 : a0 + u;
        const float sc = q8_scale_of(a, s0, s1, s2, s3, s4, s5);
        const float x = row[q0 + (mm < nq ? mm : 0)];
        o |= (mm < nq ? q8_byte(x, sc) : 0u) << (8 * u);
    }
    return o;
}
__device__ __forceinline__ unsigned int q8_row_word(const float *row, const Q8Dev &q, int j) {
    return q8_row_word_v(row, q.D, q.q0, q.q1, q.w16, q.s0, q.s1, q.s2, q.s3, q.s4, q.s5, j);
}

// Flags: relaxed system-scope accesses (they bypass the caches; no fence per access - an acquire / release at system scope invalidates
// / writes back the whole L2 on this part).  What orders them against the rows: the rows are written through and drained
// (s_waitcnt vmcnt(0)) before a workgroup takes its ticket, the flag is stored by whoever takes the last ticket; readers of the rows
// are separate launches behind the wait (kernel-boundary acquire).
__device__ __forceinline__ unsigned long long ld_sys(const unsigned long long *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED,
    __HIP_MEMORY_SCOPE_SYSTEM); }
__device__ __forceinline__ void st_sys(unsigned long long *p,
    unsigned long long v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }

// The FENCED variant of the protocol (qs_xchg_set_fenced, or endpoints created under QS_XCHG_FENCED=1; the fallback if ObsExchange.verify()
// ever fails on a real xGMI node, before giving the transport up for RCCL).  Producer: EVERY workgroup executes a system-scope RELEASE
// fence (buffer_wbl2 sc0 sc1 + wait) behind its drained rows and in front of its ticket - a release writes back the L2 of the XCD it runs
// on, so the fence of the last workgroup alone (round 5) would have covered only the rows produced on that one XCD; ~20 us per
// step when it was measured in round 3, acceptable on a fallback.  Consumer: whoever has seen a flag executes a system-scope ACQUIRE fence
// (buffer_inv sc0 sc1), once per launch, on the wave that polls.
__device__ __forceinline__ void fence_release_sys(int fenced) { if (fenced) __builtin_amdgcn_fence(__ATOMIC_RELEASE, ""); }
__device__ __forceinline__ void fence_acquire_sys(int fenced) { if (fenced) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, ""); }

// Write-through stores for the rows that go into a window (own or a peer's): `sc0 sc1` = system scope, the line leaves the L2 at once.
// With them a workgroup only has to wait for its own stores (s_waitcnt vmcnt(0)) before it takes its ticket - no system-scope release
// fence per workgroup, which on this part writes the whole L2 back and cost ~20 us per step when 128-256 workgroups each issued one
// (profiles/r03d_bench_lines.txt: 27.7 us per C2 step with the fences, 7.7 us without any exchange).
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void st16_wt(void *p,
    u32x4_t v) { asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory"); }
__device__ __forceinline__ void st4_wt(void *p,
    unsigned int v) { asm volatile("global_store_dword %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory"); }
__device__ __forceinline__ void st2_wt(void *p,
    unsigned int v) { asm volatile("global_store_short %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory"); }
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


// ------------------------------------------------------------------------------------------------
// Resident-state stepping (include/quadswarm.h: qs_gate_*, qs_step_gated): the multi-step team kernels keep the drone state in registers
// across control steps and, per step, WAIT for the step's actions and PUBLISH that its outputs are in HBM - through per-workgroup
// sequence words in device memory, so that whoever produces the actions (a policy's kernels on another stream, the trivial producer
// of the benchmark) runs concurrently, with no kernel boundary and no state round trip between two control steps.  Everything is on
// ONE device: agent-scope accesses (`sc1`: the L2s of the 8 XCDs are not coherent with each other, so data handed between concurrently
// running kernels is written through / read past them), relaxed flags behind `s_waitcnt vmcnt(0)` of the data, bounded polls.
// ------------------------------------------------------------------------------------------------
struct Gate {   // (sequence numbers: control steps since the gate was created, 1-based; a launch is told its base as a kernel argument)
    unsigned long long timeout_ticks;
    char *act_ring;                      // [ring_len][T][4] real: the action batch of sequence number s is slot (s - 1) % ring_len
    unsigned long long act_stride;       // bytes of one batch
    unsigned int ring_len, groups, wg_per_group, blocks;
    // status bits: 1 = an action wait timed out (the launch then stopped waiting), 2 = a producer wait timed out
    unsigned int status, pad;
    // [groups]  producer -> stepper: the actions of sequence number <= act_flag[g] for the workgroups of group g are in the ring
    unsigned long long *act_flag;
    // [blocks]  stepper -> consumers: obs / reward / done of sequence number done_flag[w] of workgroup w's environments are in HBM
    unsigned long long *done_flag;
};
// (the gate's ring and sequence words live in fine-grained, uncached device memory - qs_gate_create - like the exchange's flag windows;
// relaxed system-scope accesses go straight to it)
__device__ __forceinline__ unsigned long long ld_agent(const unsigned long long *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED,
    __HIP_MEMORY_SCOPE_SYSTEM); }
__device__ __forceinline__ void st_agent(unsigned long long *p,
    unsigned long long v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
__device__ __forceinline__ bool poll_ge_agent(const unsigned long long *p, unsigned long long want, unsigned long long timeout_ticks) {
    if (ld_agent(p) >= want) return true;
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < timeout_ticks) {
        if (ld_agent(p) >= want) return true;
        __builtin_amdgcn_s_sleep(2);
    }
    return ld_agent(p) >= want;
}
// Data handed between concurrently running kernels.  Stepper -> consumer (observation rows, reward, done, masks; ordinary coarse-grained
// device memory): SYSTEM-scope write-through stores (`sc0 sc1`, like the producer's ring writes: with `sc1` alone a flag was seen to
// overtake its batch), drained (s_waitcnt vmcnt(0)) before done_flag is raised.  A consumer that has seen done_flag >= s executes an
// agent-scope acquire (`buffer_inv sc1` = __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent")) before it reads the rows: its own XCD's L2 may
// still hold the rows of step s - 1 (include/quadswarm.h; tests/test_gated_gpu.py runs such a consumer against a resident
// launch). Producer -> stepper (the action ring, uncached memory): `sc0 sc1` loads.
typedef float f32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ u32x4_t ld16_sc1(const void *p) { u32x4_t v;
    asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory"); return v; }
__device__ __forceinline__ unsigned int ld4_sc1(const void *p) { unsigned int v;
    asm volatile("global_load_dword %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory"); return v; }
__device__ __forceinline__ void st16_sc1(void *p,
    u32x4_t v) { asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory"); }
__device__ __forceinline__ void st8_sc1(void *p,
    unsigned long long v) { asm volatile("global_store_dwordx2 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory"); }
__device__ __forceinline__ void st4_sc1(void *p,
    unsigned int v) { asm volatile("global_store_dword %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory"); }
__device__ __forceinline__ void st1_sc1(void *p,
    unsigned int v) { asm volatile("global_store_byte %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory"); }

}   // namespace qsx
