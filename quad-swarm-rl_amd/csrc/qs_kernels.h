// qs_kernels.h - device side of the MI355X (gfx950) QuadSwarm stepper: buffers, LDS layout, kernels.
// Included by quadswarm_hip.hip (generic library) and by qs_spec_kernels.hip (config-specialised code objects).
#pragma once
#include <hip/hip_runtime.h>

#include <cstring>

#include "qs_device.h"
#include "qs_scenarios.h"
#include "qs_xchg_dev.h"

using namespace qs;

#define QS_WAVE 64
// "team" kernels (qs_step_team.inc): 4 waves per workgroup.  Their exchanges go through LDS only, so the workgroup
// barrier waits for LDS traffic but leaves global loads / stores in flight; inside one wave LDS operations are
// processed in issue order, so a wave-local hand-over needs a compiler barrier only.
#define QS_TEAM_WAVES 4
#define QS_TEAM_THREADS (QS_WAVE * QS_TEAM_WAVES)
#define QS_TEAM_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
#define QS_WAVE_SYNC() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")

// ------------------------------------------------------------------------------------------------
// device buffers
// ------------------------------------------------------------------------------------------------
// The per-drone arrays the step kernels touch every launch live in ONE allocation, so that the kernels address them through a
// single buffer resource: `buffer_load/store vdata, voffset(lane), srsrc, soffset(array + component)` needs one SALU add per
// access where a flat 64-bit address needs three VALU instructions (about 90 accesses per drone-step; the throughput
// regime of the single-wave kernels is VALU-bound).  The typed pointers below still point into the block.
// The per-drone state lives in ONE allocation, wave-blocked: block b holds the drones of the `epb` environments one wave steps
// (epb = 64 / N, lane = local env * N + drone), so that everything a wave loads and stores each step is one contiguous `block_bytes`
// chunk (11 KB in float32) instead of 42 rows scattered over 42 component-major arrays (tools/ubench_hbm.hip: 113.7 -> 97.6 us for
// 2^20 drones).  Inside a block the arrays follow each other - pos | vel | rot | omega | ... | flags | pair mask - and an
// array of `comps` components per drone has one of TWO element orders, fixed per handle (StateBlk::lane_major):
//   rows (0):       component-major, element (q, lane) at q * 64 + lane: every access is one 4-byte element per lane, 256 contiguous
//                   bytes per wave - 45 loads and 45 stores per block and step.  The single-wave throughput kernels and the 4-wave team
//                   kernels (N > 8) run on this order: in the bandwidth regime the narrow rows stream 1 % faster and leave the register
//                   allocator free of tuple constraints (tools/ubench_rowwidth.hip, profiles/r05l_ab_layout.txt).
//   lane-major (1): the components of one drone adjacent, element (q, lane) at lane * comps + q: a lane moves an array with ONE 12- or
//                   16-byte access (`buffer_load_dwordx3/x4`), 14 loads and 14 stores per block and step.  The specialised 8-wave team
//                   kernels (N <= 8: the latency regime of the headline) run on this order: wave 0 issues its loads in 240 instead of 600
//                   clocks and its stores in 250 instead of 400 (same microbenchmark); C2 7.82 -> 7.64 us per step.
// pos .. pair are byte offsets of an array INSIDE a block (the same for both orders); the per-step outputs (newpair, reward, done, ohit)
// stay flat component-major arrays behind the blocks (absolute offsets) - their consumers read them as plain vectors.
// bytes of one state block: 40 x 64 reals (pos 3, vel 3, rot 9, omega 3, rot_damp 4, cmds_damp 4, ou 4, goal 3, ring 4, sums 3), 64 flag
// words (u32), 64 pair masks (u64) - create_typed (quadswarm_hip.hip) lays the arrays out and checks this figure
static inline int qs_block_bytes(int real_size) { return 40 * 64 * real_size + 64 * 4 + 64 * 8; }
// byte offsets of the arrays inside a state block, in the order create_typed lays them out (it checks them)
template <typename real> struct BlkOff {
    static constexpr uint32_t row = 64 * sizeof(real);
    static constexpr uint32_t pos = 0, vel = 3 * row, rot = 6 * row, omega = 15 * row, rot_damp = 18 * row, cmds_damp = 22 * row,
        ou = 26 * row, goal = 30 * row,
                              ring = 33 * row, sums = 37 * row, flags = 40 * row, pair = 40 * row + 256, bytes = 40 * row + 768;
};
// components per drone of the blocked arrays (the lane pitch inside an array, in elements)
namespace blkc { constexpr int pos = 3, vel = 3, rot = 9, omega = 3, rot_damp = 4, cmds_damp = 4, ou = 4, goal = 3, ring = 4, sums = 3,
    flags = 1, pair = 1; }
struct StateBlk { char *base; uint32_t bytes, block_bytes, epb, pos, vel, rot, omega, rot_damp, cmds_damp, ou, goal, ring, sums, flags,
    pair, newpair, reward, done, ohit, lane_major; };
// element (component q of drone i of env e) of a blocked array of `comps` components, for the code outside the step kernels'
// buffer-resource views
template <typename TT> __device__ __forceinline__ TT &blk_at(const StateBlk &b, uint32_t arr, int comps, int q, int e, int i, int N) {
    const int blk = e / (int)b.epb, lane = (e - blk * (int)b.epb) * N + i;
    return *(TT *)(b.base + (size_t)blk * b.block_bytes + arr + (b.lane_major
        ? (size_t)lane * comps + q : (size_t)q * 64 + lane) * sizeof(TT));
}
#define QS_BLK_AT(TT, b, name, q, e, i, N) blk_at<TT>((b), (b).name, blkc::name, (q), (e), (i), (N))

template <typename real> struct Ptrs {
    StateBlk blk;
    real *pos, *vel, *rot, *omega, *rot_damp, *cmds_damp, *ou, *goal;
    uint32_t *flags;
    uint64_t *pair_mask, *new_pair_mask;
    real *obs, *reward, *rew_info;
    uint8_t *done;
    int32_t *obst_hit_idx;
    uint64_t *unique_col, *obst_new, *room_new;
    int32_t *counters, *tick;
    uint32_t *step_ctr;
    real *obst_pos;
    real *dist_ring, *dist_sums;
    real *ep_stats;
    int32_t *ep_counters;
    real *scen_real;
    int32_t *scen_int;
    uint32_t *error_flag;
    uint64_t *scen_omap;   // [4, E] obstacle map bitsets (scenarios that sample free cells during an episode)
    int32_t *scenario_id;  // [E] active scenario (the sub-scenario under `mix`)
    int32_t *ep_scenario;  // [E] scenario of the last finished episode
    real *run_sums, *ep_sums;   // [QS_SUM_COUNT, T] per-episode sums (running / last finished episode)
    int32_t *obst_count;   // [E] obstacles of the running episode (domain randomisation; the other slots are parked far away)
    real *obst_size_env, *obst_density_env;   // [E] obstacle size / density of the running episode
    const int32_t *dr_count;                  // [QS_MAX_DR_CHOICES] --quads_domain_random choice tables: obstacle count,
    const real *dr_density, *dr_size;         //                     density, size
    uint8_t *reset_mask;   // [E] nonzero => reset kernel re-initialises this env
    unsigned long long *timing;   // [32] phase time stamps of workgroup 0 (only written by -DQS_TIMING builds)
    const real *rew_rt;   // [QS_REW_COUNT + 1] run-time reward coefficients + proximity slope (qs_set_reward_coeffs)
    const qsx::XchgDev *xchg;   // qs_set_obs_exchange: the team step kernels also store their observation rows into every rank's window
    // noise tape (qs_set_noise_tape; consumed by the QS_TAPE kernels only): [E][tape_len] reference draws, per-env cursor
    const double *tape;
    int32_t *tape_pos;
    int64_t tape_len;
};

// One array of the state allocation seen from one lane: element type T; `off` = scalar offset of the array (for a blocked array: of this
// wave's block too), `lane_off` = this lane's first byte inside it, `comp_bytes` = bytes between the components of one lane (a blocked
// array: sizeof(T) in the lane-major order, a row of 64 elements otherwise; a flat component-major array: its row pitch), `wide` = the
// lane's components are adjacent and ldv / stv may move them as 12- / 16-byte pieces.
typedef unsigned int qs_u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int qs_u32x3 __attribute__((ext_vector_type(3)));
typedef unsigned int qs_u32x4 __attribute__((ext_vector_type(4)));
template <typename T> struct BufRow {
    __amdgpu_buffer_rsrc_t r;
    uint32_t off, comp_bytes, lane_off;
    bool wide;
    __device__ __forceinline__ T ld(int q = 0) const {
        if constexpr (sizeof(T) == 8) return __builtin_bit_cast(T,
            __builtin_amdgcn_raw_buffer_load_b64(r, lane_off, off + (uint32_t)q * comp_bytes, 0));
        else return __builtin_bit_cast(T, __builtin_amdgcn_raw_buffer_load_b32(r, lane_off, off + (uint32_t)q * comp_bytes, 0));
    }
    __device__ __forceinline__ void st(T v, int q = 0) const {
        if constexpr (sizeof(T) == 8) __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(qs_u32x2, v), r, lane_off,
            off + (uint32_t)q * comp_bytes, 0);
        else if constexpr (sizeof(T) == 4) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, v), r, lane_off,
            off + (uint32_t)q * comp_bytes, 0);
        else __builtin_amdgcn_raw_buffer_store_b8((uint8_t)v, r, lane_off, off + (uint32_t)q * comp_bytes, 0);
    }
    // the D dwords from byte `at` of this lane's piece of a lane-major array, as the widest accesses that cover them (16, 12, 8, 4 bytes)
    template <int D> __device__ __forceinline__ void ld_dwords(uint32_t *w, uint32_t at) const {
        if constexpr (D >= 4) { const qs_u32x4 x = __builtin_amdgcn_raw_buffer_load_b128(r, lane_off, off + at, 0); w[0] = x.x;
            w[1] = x.y; w[2] = x.z; w[3] = x.w; if constexpr (D > 4) ld_dwords<D - 4>(w + 4, at + 16); }
        else if constexpr (D == 3) { const qs_u32x3 x = __builtin_amdgcn_raw_buffer_load_b96(r, lane_off, off + at, 0); w[0] = x.x;
            w[1] = x.y; w[2] = x.z; }
        else if constexpr (D == 2) { const qs_u32x2 x = __builtin_amdgcn_raw_buffer_load_b64(r, lane_off, off + at, 0); w[0] = x.x;
            w[1] = x.y; }
        else w[0] = __builtin_amdgcn_raw_buffer_load_b32(r, lane_off, off + at, 0);
    }
    template <int D> __device__ __forceinline__ void st_dwords(const uint32_t *w, uint32_t at) const {
        if constexpr (D >= 4) { const qs_u32x4 x = {w[0], w[1], w[2], w[3]};
            __builtin_amdgcn_raw_buffer_store_b128(x, r, lane_off, off + at, 0); if constexpr (D > 4) st_dwords<D - 4>(w + 4, at + 16); }
        else if constexpr (D == 3) { const qs_u32x3 x = {w[0], w[1], w[2]};
            __builtin_amdgcn_raw_buffer_store_b96(x, r, lane_off, off + at, 0); }
        else if constexpr (D == 2) { const qs_u32x2 x = {w[0], w[1]}; __builtin_amdgcn_raw_buffer_store_b64(x, r, lane_off, off + at, 0); }
        else __builtin_amdgcn_raw_buffer_store_b32(w[0], r, lane_off, off + at, 0);
    }
    // all K components of this lane's drone (blocked arrays: K = the array's component count)
    template <int K> __device__ __forceinline__ void ldv(T *dst) const {
        if (!wide) {
#pragma unroll
            for (int k = 0; k < K; ++k) dst[k] = ld(k);
            return;
        }
        constexpr int D = K * (int)sizeof(T) / 4;
        uint32_t w[D];
        ld_dwords<D>(w, 0);
#pragma unroll
        for (int k = 0; k < K; ++k) {
            if constexpr (sizeof(T) == 8) dst[k] = __builtin_bit_cast(T, (uint64_t)w[2 * k] | ((uint64_t)w[2 * k + 1] << 32));
            else dst[k] = __builtin_bit_cast(T, w[k]);
        }
    }
    template <int K> __device__ __forceinline__ void stv(const T *src) const {
        if (!wide) {
#pragma unroll
            for (int k = 0; k < K; ++k) st(src[k], k);
            return;
        }
        constexpr int D = K * (int)sizeof(T) / 4;
        uint32_t w[D];
#pragma unroll
        for (int k = 0; k < K; ++k) {
            if constexpr (sizeof(T) == 8) { const uint64_t u = __builtin_bit_cast(uint64_t, src[k]); w[2 * k] = (uint32_t)u;
                w[2 * k + 1] = (uint32_t)(u >> 32); }
            else w[k] = __builtin_bit_cast(uint32_t, src[k]);
        }
        st_dwords<D>(w, 0);
    }
};
#define QS_BUF_RSRC(p) __builtin_amdgcn_make_buffer_rsrc((void *)(p).blk.base, 0, (p).blk.bytes, 0x00020000)
// blocked state array: the workgroup's block (blockIdx.x: one block = the environments of one workgroup), lane = position inside the wave;
// component offsets are compile-time constants next to the scalar block offset.  QS_LM: the element order of the handle's blocks - a
// literal in a config-specialised object (lane-major <=> the 8-wave team kernels), the handle's flag in the generic kernels
#define QS_BLK blockIdx.x
#if defined(QS_SPEC_TEAM)
#define QS_LM(p) (QS_SPEC_TEAM == 8)
#else
#define QS_LM(p) ((p).blk.lane_major != 0)
#endif
#define QS_ROW(TYPE, name, rs, p, T, g) const BufRow<TYPE> b_##name = {rs, (uint32_t)(QS_BLK * (p).blk.block_bytes + (p).blk.name), (uint32_t)((QS_LM(p) ? 1 : 64) * sizeof(TYPE)), \
                                                                       (uint32_t)((threadIdx.x & 63) * ((QS_LM(p) ? blkc::name : 1) * sizeof(TYPE))), QS_LM(p)}
// flat component-major array (per-step outputs): row pitch T elements, lane offset = global drone index
#define QS_ROWF(TYPE, name, rs, p, T, g) const BufRow<TYPE> b_##name = {rs, (p).blk.name, (uint32_t)((T) * sizeof(TYPE)), (uint32_t)((g) * sizeof(TYPE)), false}

// (neighbour metric, drone index) as one unsigned key whose order is "smaller metric first, lower index first": the IEEE bit
// pattern of a float is monotone after flipping the sign bit of non-negatives and all bits of negatives
__device__ __forceinline__ unsigned long long nbr_key(float m, int idx) {
    const uint32_t b = __float_as_uint(m), o = b ^ ((b >> 31) ? 0xffffffffu : 0x80000000u);
    return ((unsigned long long)o << 32) | (uint32_t)idx;
}

// neighbour metric of neighborhood_indices (quadrotor_multi.py:247-274) for the relative position rp / velocity rv of a partner:
// max(|rp|, 0.01) + rp.rv / that.  Symmetric in the pair - (-rp).(-rv) = rp.rv and |-rp| = |rp| exactly in IEEE arithmetic - which is
// what lets the team kernels evaluate every unordered pair once.  ONE definition for every site that ranks neighbours.
template <typename real> __device__ __forceinline__ real nbr_metric(const real rp[3], const real rv[3]) {
    const real rd = M<real>::fmax(norm3<real>(rp), (real)0.01);
    return rd + (rp[0] * rv[0] + rp[1] * rv[1] + rp[2] * rv[2]) * M<real>::rcp(rd);
}
// fp32: the multiply-add chain written out (the one the compiler forms from the expression above), so that the two-partners-at-once form
// below (v_pk_mul / v_pk_fma_f32: both halves round like the scalar instruction) gives the same bits as this one
template <> __device__ __forceinline__ float nbr_metric<float>(const float rp[3], const float rv[3]) {
    const float d2 = __builtin_fmaf(rp[2], rp[2], __builtin_fmaf(rp[1], rp[1], rp[0] * rp[0]));
    const float dot = __builtin_fmaf(rp[2], rv[2], __builtin_fmaf(rp[1], rv[1], rp[0] * rv[0]));
    const float rd = fmaxf(__builtin_amdgcn_sqrtf(d2), 0.01f);
    return __builtin_fmaf(dot, __builtin_amdgcn_rcpf(rd), rd);
}
typedef float qs_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ qs_f32x2 nbr_metric_x2(const qs_f32x2 rp[3], const qs_f32x2 rv[3]) {
    const qs_f32x2 d2 = __builtin_elementwise_fma(rp[2], rp[2], __builtin_elementwise_fma(rp[1], rp[1], rp[0] * rp[0]));
    const qs_f32x2 dot = __builtin_elementwise_fma(rp[2], rv[2], __builtin_elementwise_fma(rp[1], rv[1], rp[0] * rv[0]));
    const qs_f32x2 rd = {fmaxf(__builtin_amdgcn_sqrtf(d2.x), 0.01f), fmaxf(__builtin_amdgcn_sqrtf(d2.y), 0.01f)};
    const qs_f32x2 rc = {__builtin_amdgcn_rcpf(rd.x), __builtin_amdgcn_rcpf(rd.y)};
    return __builtin_elementwise_fma(dot, rc, rd);
}
// (metric, index) with the order "smaller metric first, lower index first" = position in the stable argsort
template <typename real> struct NbrKey;
template <> struct NbrKey<float> {
    unsigned long long k;
    static __device__ __forceinline__ NbrKey make(float m, int j) { return {nbr_key(m, j)}; }
    __device__ __forceinline__ bool less(const NbrKey &o) const { return k < o.k; }
    __device__ __forceinline__ int idx() const { return (int)(uint32_t)k; }
    // LDS list entry `slot` of lane tid: keys [slots][B]
    static __device__ __forceinline__ void st(unsigned char *base, int slots, int slot, int B, int tid,
        const NbrKey &v) { ((unsigned long long *)base)[slot * B + tid] = v.k; }
    static __device__ __forceinline__ NbrKey ld(const unsigned char *base, int slots, int slot, int B,
        int tid) { return {((const unsigned long long *)base)[slot * B + tid]}; }
};
template <> struct NbrKey<double> {
    double m; int j;
    static __device__ __forceinline__ NbrKey make(double m, int j) { return {m, j}; }
    __device__ __forceinline__ bool less(const NbrKey &o) const { return (m < o.m) | ((m == o.m) & (j < o.j)); }
    __device__ __forceinline__ int idx() const { return j; }
    // metrics [slots][B] doubles, then indices [slots][B] ints
    static __device__ __forceinline__ void st(unsigned char *base, int slots, int slot, int B, int tid, const NbrKey &v) {
        ((double *)base)[slot * B + tid] = v.m; ((int *)(base + (size_t)slots * B * 8))[slot * B + tid] = v.j; }
    static __device__ __forceinline__ NbrKey ld(const unsigned char *base, int slots, int slot, int B, int tid) {
        return {((const double *)base)[slot * B + tid], ((const int *)(base + (size_t)slots * B * 8))[slot * B + tid]}; }
};

#ifdef QS_TIMING
#define QS_STAMP(k) do { if (blockIdx.x == 0 && threadIdx.x == 0) p.timing[k] = clock64(); } while (0)
// helper waves of the team kernels: stamp k of wave w (1..3) lands in timing[32*w + k]
#define QS_STAMPW(k) do { if (blockIdx.x == 0 && (threadIdx.x & 63) == 0 && threadIdx.x >= 64) p.timing[32 * (threadIdx.x >> 6) + (k)] = clock64(); } while (0)
// start / end of EVERY workgroup (wave 0; s_memtime and the constant 100 MHz wall clock) and where it ran: HW_ID (gfx9: wave 3:0, simd 5:4,
// pipe 7:6, cu 11:8, sh 12, se 15:13) and XCC_ID
#define QS_WG_STRIDE 16
#define QS_STAMP_WG(k) do { if (threadIdx.x == 0) { p.timing[128 + QS_WG_STRIDE * blockIdx.x + (k)] = clock64(); p.timing[128 + QS_WG_STRIDE * blockIdx.x + 4 + (k)] = wall_clock64(); \
        if ((k) == 0) { p.timing[128 + QS_WG_STRIDE * blockIdx.x + 2] = __builtin_amdgcn_s_getreg((4) | (0 << 6) | (31 << 11)); \
                        p.timing[128 + QS_WG_STRIDE * blockIdx.x + 3] = __builtin_amdgcn_s_getreg((20) | (0 << 6) | (31 << 11)); } } } while (0)
// phase boundaries of EVERY workgroup's wave 0 (slots 6..15): which phase makes the slowest workgroup of a launch slow
#define QS_STAMP_WGK(k) do { if (threadIdx.x == 0) p.timing[128 + QS_WG_STRIDE * blockIdx.x + (k)] = clock64(); } while (0)
#else
#define QS_STAMP(k) do { } while (0)
#define QS_STAMPW(k) do { } while (0)
#define QS_STAMP_WG(k) do { } while (0)
#define QS_STAMP_WGK(k) do { } while (0)
#endif

// Debug builds (QS_SPEC_EXTRA_FLAGS="-DQS_POISON_IDLE=1|2 [-DQS_POISON_PARTS=mask]"; tests/test_object_identity_gpu.py): the lanes of a
// wave that own no drone - 64 % N of them, and every lane past the batch's last environment - start from garbage instead of a copy of drone
// 0 (1: NaN, 2: 3e30 in every float; all ones in flags and masks), and the dynamic LDS is filled with the same pattern before its first
// use.  Results must be bit-identical to the plain build's: nothing an active lane computes may depend on an idle lane's registers or on an
// LDS word that nobody wrote in this launch.  QS_POISON_PARTS (default: everything) picks what is poisoned, for bisecting: 1 drone state, 2
// flags / pair mask, 4 LDS, 8 actions, 16 goal / distance sums / ring.
#ifdef QS_POISON_IDLE
#ifndef QS_POISON_PARTS
#define QS_POISON_PARTS 31
#endif
// (opaque: fast-math would fold a literal NaN away)
__device__ __forceinline__ uint32_t qs_poison_bits() { uint32_t u = QS_POISON_IDLE == 1 ? 0x7fc0deadu : 0x7217e7d5u;
    asm volatile("" : "+v"(u)); return u; }
template <typename real> __device__ __forceinline__ real qs_poison() { return (real)__builtin_bit_cast(float, qs_poison_bits()); }
#define QS_POISON_LDS(total_bytes, nthreads) do { if (QS_POISON_PARTS & 4) { for (int w_ = threadIdx.x; w_ < (total_bytes) / 4; w_ += (nthreads)) ((uint32_t *)smem)[w_] = qs_poison_bits(); __syncthreads(); } } while (0)
#define QS_POISON_ARR_(a, n) do { if (!active) { _Pragma("unroll") for (int q_ = 0; q_ < (n); ++q_) (a)[q_] = qs_poison<real>(); } } while (0)
#define QS_POISON_ARR(a, n, part) do { if (QS_POISON_PARTS & (part)) QS_POISON_ARR_(a, n); } while (0)
#define QS_POISON_DRONE(d) do { if (QS_POISON_PARTS & 1) { QS_POISON_ARR_((d).pos, 3); QS_POISON_ARR_((d).vel, 3); QS_POISON_ARR_((d).rot, 9); QS_POISON_ARR_((d).omega, 3); \
                                QS_POISON_ARR_((d).rot_damp, 4); QS_POISON_ARR_((d).cmds_damp, 4); QS_POISON_ARR_((d).ou, 4); } if ((QS_POISON_PARTS & 2) && !active) (d).flags = ~0u; } while (0)
#define QS_POISON_U64(x) do { if ((QS_POISON_PARTS & 2) && !active) (x) = ~0ull; } while (0)
#else
#define QS_POISON_LDS(total_bytes, nthreads) do { } while (0)
#define QS_POISON_ARR(a, n, part) do { } while (0)
#define QS_POISON_DRONE(d) do { } while (0)
#define QS_POISON_U64(x) do { } while (0)
#endif

struct LdsLayout { int off_mask, off_omap, off_si, off_sr, off_envflag, off_scratch, off_pos, off_vel, off_zax, off_om, off_goal,
    off_obst, off_metric, off_obs, goal_rows, total;
                   int off_t_rot, off_t_goal, off_t_prox, off_t_col, off_t_dw, off_t_ohit;
                   // team kernels: the step's 17 reward terms + the done flag, for the wave that keeps the episode sums
                   int off_t_ri;
                   // team kernels, N > 8 (pair-once scan): proximity masks, per-wave top-8 lists, per-env "velocities changed" flags
                   int off_t_near, off_topk, off_t_redo;
                   int off_self, off_rows, rows_per_pass, scr_cap;   // single-wave kernels: observation output (see obs_copy_rows)
                   int off_cur; };   // QS_TAPE kernels: per-env tape cursor
// neighbour + SDF columns a lane can keep in registers between the row passes (K <= 8)   // single-wave kernels: observation columns staged
// per pass (see obs_flush)
#define QS_NV_MAX 57
#define QS_RESET_SCRATCH_INTS 160   // per env, full scenario set: virtual-pool index/value lists (2x64) + two DP rows (2x16)

// Observation rows leave through LDS so that they reach HBM as whole, aligned cache lines (a first version wrote column
// groups straight from a small stage: 72-byte runs at a 216-byte stride cost more than the rest of the step - see
// profiles/r02_exp_colgroup_flush.txt).  Team kernels (small batches, occupancy irrelevant) stage the workgroup's complete
// rows [64][D].  Single-wave (throughput) kernels keep the self columns in a dense block [64][S] (written early, read by
// the bookkeeping) and pass the other D-S columns through a stage of `rows_per_pass` rows: each lane holds its neighbour / SDF
// values in registers, the lanes of one row group store them, the whole wave copies that group's complete rows
// (rows_per_pass * D * 4 contiguous bytes) out, next group.  The stage shares its LDS with the buffers of the rare paths
// (collision responses, goal scratch rows, reset scratch), none of which is live while rows are being copied: a workgroup
// needs ~9.5 KB instead of ~21 KB, and 16 of them fit a CU.
static LdsLayout lds_layout(int real_size, int B, int N, int epb, int obs_dim, int num_obst, int K, int team /* waves per workgroup of the team kernels, 0 = single-wave */,
                            bool full /* kernels of the full scenario set: per-env scenario state in LDS */, int scenario = -1, int rows_per_pass = 64) {
    LdsLayout L;
    memset(&L, 0, sizeof L);
    const int self_dim = obs_dim - 6 * K - (num_obst > 0 ? 9 : 0), extra = obs_dim - self_dim;
    // goal scratch rows per env: two formations (+ sphere padding) for swarm_vs_swarm and the full scenario set, one otherwise
    L.goal_rows = (full || scenario < 0 || scenario == QS_SCENARIO_SWARM_VS_SWARM) ? 2 * N + 8 : (N < 3 ? 3 : N);
    // reset scratch per env: virtual-pool index / value lists (2 x cap: at most one entry per obstacle / per drone) + two DP rows
    // (2 x 16); the full scenario set keeps the fixed 2 x 64 split its scenario code addresses
    L.scr_cap = full ? 64 : (num_obst > N ? num_obst : N);
    const int goal_bytes = real_size * 3 * L.goal_rows * epb, scratch_bytes = num_obst > 0 ? 4 * (2 * L.scr_cap + 32) * epb : 0;
    int o = 0;
    L.off_omap = o; o += full ? 8 * 4 * epb : 0;      // full-scenario kernels: obstacle map bitset, scenario ints / reals per env
    L.off_si = o; o += full ? 4 * SI_COUNT * epb : 0;
    L.off_sr = o; o += full ? real_size * SR_COUNT * epb : 0;
    o = (o + 15) & ~15;
    // [epb] have-spawn-points flags + [epb] swarm_vs_swarm periods + [epb] obstacle-size choice
    L.off_envflag = o; o += 4 * ((3 * epb + 3) & ~3);
#ifdef QS_TAPE
    L.off_cur = o; o += 4 * ((epb + 3) & ~3);
#endif
    o = (o + 15) & ~15;
    L.off_pos = o; o += real_size * 3 * B;
    L.off_vel = o; o += real_size * 3 * B;
    L.off_zax = o; o += real_size * 3 * B;            // body z axes (downwash); spawn points in the reset tail
    L.off_obst = o; o += real_size * 2 * (num_obst > 0 ? num_obst : 1) * epb;   // obstacle xy of the block's envs
    // neighbour metric rows [N][B]; team kernels with N > 8: one sorted top-8 list per wave, metrics [8W][B] + indices [8W][B]
    L.off_metric = o; o += ((team ? K > 0 : K > 8) && K < N - 1)
        ? ((team && N > 8) ? ((real_size + 4) * 8 * team > real_size * N
        ? (real_size + 4) * 8 * team : real_size * N) * B : real_size * N * B) : 0;
    o = (o + 15) & ~15;
    if (team) {
        L.off_mask = o; o += 8 * B;                   // u64 per lane: new-pair masks for the serial response path
        L.off_om = o; o += real_size * 3 * B;
        L.off_goal = o; o += goal_bytes;
        L.off_scratch = o; o += scratch_bytes;
        o = (o + 15) & ~15;
        L.off_t_col = o; o += 8 * B;                  // exchange rows of the team kernels
        L.off_t_dw = o; o += 8 * B;
        L.off_t_rot = o; o += real_size * 6 * B;
        L.off_t_goal = o; o += real_size * 3 * B;
        L.off_t_prox = o; o += real_size * (team - 1) * B;
        L.off_t_ohit = o; o += 4 * B;
        L.off_t_near = o; o += 8 * B;
        L.off_t_ri = o; o += real_size * (QS_RI_COUNT + 1) * B;
        L.off_t_redo = o; o += 4 * ((epb + 3) & ~3);
        o = (o + 15) & ~15;
        // N > 8 with K <= 8: every ranking wave (all but wave 0) publishes the top-8 of its candidates, compacted by local rank:
        // float32: one 64-bit (metric, index) key per entry; float64: metric + index
        L.off_topk = o; o += (N > 8 && K > 0 && K < N - 1 && K <= 8) ? (team - 1) * 8 * (real_size + 4) * B : 0;
        o = (o + 15) & ~15;
        L.rows_per_pass = B;
        L.off_obs = o; L.off_self = o; L.off_rows = o + real_size * self_dim * B;   // complete rows [B][D]; the reset kernel sees them as
        o += real_size * obs_dim * B;                                                // the self block followed by the rows stage
    } else {
        L.rows_per_pass = rows_per_pass < 1 ? 1 : (rows_per_pass > B ? B : rows_per_pass);
        L.off_self = o; L.off_obs = o; o += real_size * self_dim * B;
        o = (o + 15) & ~15;
        // the rows stage and, in the same bytes, what only the rare paths touch: (a) collision responses: pair masks + angular
        // velocities, (b) scenario goal switch / reset: goal scratch rows + reset scratch
        const int stage_bytes = real_size * extra * L.rows_per_pass, rare_a = 8 * B + real_size * 3 * B,
            rare_b = goal_bytes + scratch_bytes;
        int u = stage_bytes > rare_a ? stage_bytes : rare_a;
        u = u > rare_b ? u : rare_b;
        L.off_rows = o; L.off_mask = o; L.off_om = o + 8 * B; L.off_goal = o; L.off_scratch = o + goal_bytes;
        o += u;
    }
    L.total = (o + 15) & ~15;
    return L;
}

// ------------------------------------------------------------------------------------------------
// K-nearest neighbour observation for one drone (neighborhood_indices quadrotor_multi.py:247-274,
// extend_obs_space :233-245).  pos/vel of the env's drones are in LDS (component-major, stride B).
// The N-1 metrics are evaluated once into an LDS row, then K rounds of arg-min (lowest index wins ties,
// like the stable ordering of argsort on distinct keys).
// ------------------------------------------------------------------------------------------------
// neighborhood_indices quadrotor_multi.py:247-274 + extend_obs_space :233-245.
// pos/vel of the env's drones are in LDS (component-major, stride B).  Measured on MI355X (lone wave per SIMD, every
// LDS round trip exposed): N <= 8 is fastest with all candidates in registers and rank-by-counting (independent
// compares); larger N with a streaming sorted top-K list (insertion by compare-exchange).  Both give the first K
// entries of the stable ascending order of the metric = argsort.
template <typename real>
__device__ __forceinline__ void neighbor_obs(const Consts<real> &c, int N, int i, int base, int B, int tid, const real *s_pos,
    const real *s_vel,
                                             real *s_metric, const real mypos[3], const real myvel[3], real *o) {
    const int K = c.num_neighbors;
    if (K <= 0) return;
    if (K == N - 1 || N <= 8) {
        const bool all_others = (K == N - 1);   // all other drones in index order (:253-254)
        for (int j0 = 0; j0 < N; j0 += 8) {      // (a single chunk unless all_others with N > 8)
            real rp[8][3], rv[8][3];
            int rank[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {        // 48 LDS reads issued before the first use: one round trip
                const int j = (j0 + u < N) ? j0 + u : N - 1;
#pragma unroll
                for (int a = 0; a < 3; ++a) { rp[u][a] = s_pos[a * B + base + j] - mypos[a];
                    rv[u][a] = s_vel[a * B + base + j] - myvel[a]; }
            }
            if (all_others) {
#pragma unroll
                for (int u = 0; u < 8; ++u) rank[u] = (j0 + u < i) ? j0 + u : j0 + u - 1;
            } else {
                real mj[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    real rd = M<real>::fmax(norm3<real>(rp[u]), (real)0.01);
                    real m = rd + (rp[u][0] * rv[u][0] + rp[u][1] * rv[u][1] + rp[u][2] * rv[u][2]) * M<real>::rcp(rd);
                    mj[u] = (u < N && u != i) ? m : (real)3.4e38;
                    rank[u] = 0;
                }
#pragma unroll
                for (int k = 0; k < 8; ++k)
#pragma unroll
                    for (int u = 0; u < 8; ++u) rank[u] += (int)((mj[k] < mj[u]) | ((mj[k] == mj[u]) & (k < u)));
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int j = j0 + u;
                if (j < N && j != i && rank[u] < K) {
                    real *oo = o + rank[u] * 6;
#pragma unroll
                    for (int a = 0; a < 3; ++a) {
                        oo[a] = clipr<real>(rp[u][a], -c.nbr_clip_pos[a], c.nbr_clip_pos[a]);
                        oo[3 + a] = clipr<real>(rv[u][a], -c.nbr_clip_vel[a], c.nbr_clip_vel[a]);
                    }
                }
            }
        }
        return;
    }
    if (K <= 8) {
        // streaming pass over the candidates, 4 per LDS round trip; sorted top-8 (metric, index) list in registers.  A new
        // candidate is inserted behind entries with an equal metric, i.e. lower index first.
        real bm[8];
        int bi[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) { bm[k] = (real)3.4e38; bi[k] = 0; }
        for (int j0 = 0; j0 < N; j0 += 4) {
            real rp[4][3], rv[4][3];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int j = (j0 + u < N) ? j0 + u : N - 1;
#pragma unroll
                for (int a = 0; a < 3; ++a) { rp[u][a] = s_pos[a * B + base + j] - mypos[a];
                    rv[u][a] = s_vel[a * B + base + j] - myvel[a]; }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int j = j0 + u;
                real rd = M<real>::fmax(norm3<real>(rp[u]), (real)0.01);
                real m = rd + (rp[u][0] * rv[u][0] + rp[u][1] * rv[u][1] + rp[u][2] * rv[u][2]) * M<real>::rcp(rd);
                m = (j < N && j != i) ? m : (real)3.4e38;
                int mi = j;
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const bool lt = m < bm[k];
                    const real tm = lt ? bm[k] : m;
                    const int ti = lt ? bi[k] : mi;
                    bm[k] = lt ? m : bm[k];
                    bi[k] = lt ? mi : bi[k];
                    m = tm; mi = ti;
                }
            }
        }
        real vals[8][6];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            if (k < K) {
#pragma unroll
                for (int a = 0; a < 3; ++a) { vals[k][a] = s_pos[a * B + base + bi[k]] - mypos[a];
                    vals[k][3 + a] = s_vel[a * B + base + bi[k]] - myvel[a]; }
            }
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            if (k < K) {
#pragma unroll
                for (int a = 0; a < 3; ++a) {
                    o[k * 6 + a] = clipr<real>(vals[k][a], -c.nbr_clip_pos[a], c.nbr_clip_pos[a]);
                    o[k * 6 + 3 + a] = clipr<real>(vals[k][3 + a], -c.nbr_clip_vel[a], c.nbr_clip_vel[a]);
                }
            }
        }
        return;
    }
    // 8 < K < N-1 (unusual): metrics into this lane's LDS column, then K rounds of arg-min (lowest index wins ties)
    for (int j = 0; j < N; ++j) {
        real rp[3] = {s_pos[0 * B + base + j] - mypos[0], s_pos[1 * B + base + j] - mypos[1], s_pos[2 * B + base + j] - mypos[2]};
        real rv[3] = {s_vel[0 * B + base + j] - myvel[0], s_vel[1 * B + base + j] - myvel[1], s_vel[2 * B + base + j] - myvel[2]};
        real rd = M<real>::fmax(norm3<real>(rp), (real)0.01);
        real mm = rd + (rp[0] * rv[0] + rp[1] * rv[1] + rp[2] * rv[2]) * M<real>::rcp(rd);
        s_metric[j * B + tid] = (j == i) ? (real)3.4e38 : mm;
    }
    uint64_t taken = 1ull << i;
    for (int k = 0; k < K; ++k) {
        int best = -1;
        real bmin = (real)3.4e38;
        for (int j = 0; j < N; ++j) {
            real mm = s_metric[j * B + tid];
            bool better = !(taken >> j & 1) && (best < 0 || mm < bmin);
            best = better ? j : best;
            bmin = better ? mm : bmin;
        }
        taken |= 1ull << best;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            o[k * 6 + a] = clipr<real>(s_pos[a * B + base + best] - mypos[a], -c.nbr_clip_pos[a], c.nbr_clip_pos[a]);
            o[k * 6 + 3 + a] = clipr<real>(s_vel[a * B + base + best] - myvel[a], -c.nbr_clip_vel[a], c.nbr_clip_vel[a]);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Observation output of the single-wave kernels: rows [r0, r0 + nr) of the workgroup's block, complete, to the row-major
// observation matrix - self columns from the dense self block, the others from the rows stage.  Consecutive lanes write
// consecutive words: every store instruction is one contiguous run.  rowmask (bit = row within the block) selects the rows to
// write (an auto-reset rewrites only the rows of the finished environments).  Whole wave, between barriers.
// ------------------------------------------------------------------------------------------------
// QS_NT_OBS: cache policy of the observation rows' stores.  They are write-once per step and not read again by the stepper (42 % of a
// step's bytes): 1 = non-temporal (`nt`), so that they do not displace the state rows - which the next step re-reads - from the 256 MiB
// Infinity Cache.  Measured on MI355X, C2 shape at 131072 envs (539 MB per step): 132.2 -> 107.0 us per step, 0.50 -> 0.61 of 8 TB/s, same
// FETCH_SIZE / WRITE_SIZE (profiles/r03b_nt_obs_E131072.txt).
#ifndef QS_NT_OBS
#define QS_NT_OBS 1
#endif
typedef float qs_f32x2 __attribute__((ext_vector_type(2)));
typedef float qs_f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void obs_st2(float *dst, qs_f32x2 v) { if (QS_NT_OBS) __builtin_nontemporal_store(v, (qs_f32x2 *)dst);
    else *(qs_f32x2 *)dst = v; }
__device__ __forceinline__ void obs_st4(float *dst, qs_f32x4 v) { if (QS_NT_OBS) __builtin_nontemporal_store(v, (qs_f32x4 *)dst);
    else *(qs_f32x4 *)dst = v; }
template <typename real> __device__ __forceinline__ void obs_st1(real *dst, real v) { if (QS_NT_OBS) __builtin_nontemporal_store(v, dst);
    else *dst = v; }

template <typename real>
__device__ __forceinline__ void obs_copy_rows(real *__restrict__ dst_block, const real *s_self, const real *s_rows, int S, int D, int r0,
    int nr, uint64_t rowmask, int tid) {
#ifdef QS_EXP_NOFLUSH   // experiment: how much of the step is the observation output path
    if (rowmask != 0x1234567ull) return;
#endif
    const int X = D - S;
    real *dst = dst_block + (size_t)r0 * D;
    if (sizeof(real) == 4 && ((D | S) & 1) == 0) {   // 8-byte elements: rows (D*4 bytes) and the self / other boundary stay 8-byte aligned
        const int half = D >> 1, total = nr * half;
        for (int idx = tid; idx < total; idx += QS_WAVE) {
            const int row = idx / half, c2 = 2 * (idx - row * half);
            const float *src = (c2 < S) ? (const float *)s_self + (r0 + row) * S + c2 : (const float *)s_rows + row * X + (c2 - S);
            if ((rowmask >> (r0 + row)) & 1) obs_st2((float *)dst + 2 * idx, *(const qs_f32x2 *)src);
        }
    } else {
        const int total = nr * D;
        for (int idx = tid; idx < total; idx += QS_WAVE) {
            const int row = idx / D, cc = idx - row * D;
            const real v = (cc < S) ? s_self[(r0 + row) * S + cc] : s_rows[row * X + (cc - S)];
            if ((rowmask >> (r0 + row)) & 1) obs_st1<real>(dst + idx, v);
        }
    }
}

// K-nearest neighbour selection of one drone, split into "select once" + "emit the neighbours of ranks [k0, k1)" so that the
// rows can be staged a few neighbours at a time (same four cases and the same results as neighbor_obs above).
template <typename real> struct NbrSel {
    int bi[8];                 // K <= 8: indices of the (up to 8) nearest in order
    uint64_t taken;            // 8 < K < N-1: drones already emitted (arg-min rounds over the LDS metric column)
};
// fp32, K <= 8: the sorted top-(K + 1) list as ONE 32-bit integer key per entry - the metric's order-preserving bit pattern with its low
// IB bits replaced by the drone index - so that the insertion is one v_med3_i32 per slot (new[k] = med3(cand, old[k-1], old[k]) of a
// sorted list) instead of a compare and four selects.  The truncation is monotone, so two entries with DIFFERENT truncated metrics are
// in the exact order; wherever the exact (metric, index) order could differ from the key order - inside the first K, or across the
// K / K+1 boundary - two neighbouring entries of the sorted first K + 1 share a truncated metric, which is what the return value
// reports (false = within 2^IB ulps of a tie: the caller takes the exact path; ~1e-4 of the drones).  IB = bits of a drone index.
__device__ __forceinline__ int qs_med3_i32(int a, int b, int c) {
    int r; asm("v_med3_i32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r;
}
// the key list of one drone: K + 1 sorted entries (9 slots for K <= 8), the index / truncation masks
struct NbrKeys {
    int key[9];
    int IM, TM;
    __device__ __forceinline__ void init(int N) {
        const int IB = N <= 8 ? 3 : (N <= 16 ? 4 : (N <= 32 ? 5 : 6));
        IM = (1 << IB) - 1;
#ifdef QS_NBR_TRUNC_BITS   // (tests: a wider truncation than the index needs - the exact path is taken often, inside the same waves)
        TM = (1 << QS_NBR_TRUNC_BITS) - 1;
#else
        TM = IM;
#endif
#pragma unroll
        for (int k = 0; k < 9; ++k) key[k] = 0x7fffffff;
    }
    // candidate j with metric m (valid: a partner, i.e. j < N and j != i)
    __device__ __forceinline__ void insert(float m, int j, bool valid, int K) {
        const int b = __float_as_int(m);
        int cand = ((b ^ ((b >> 31) & 0x7fffffff)) & ~TM) | (j & IM);   // signed order = float order; -inf-wards truncation
        cand = valid ? cand : ((0x7fffffff & ~TM) | (j & IM));
        int below = key[0];
        key[0] = cand < key[0] ? cand : key[0];
#pragma unroll
        for (int k = 1; k < 9; ++k)   // (K is a literal in the specialised objects: K + 1 slots)
            if (k <= K) { const int nk = qs_med3_i32(cand, below, key[k]); below = key[k]; key[k] = nk; }
    }
    // indices of the K nearest in order; false: two neighbouring entries of the first K + 1 share a truncated metric
    __device__ __forceinline__ bool finish(int K, int bi[8]) const {
        uint32_t closest = 0xffffffffu;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const uint32_t x = (uint32_t)(key[k] ^ key[k + 1]);
            closest = (k < K && x < closest) ? x : closest;
            bi[k] = key[k] & IM;
        }
        return closest > (uint32_t)TM;
    }
};
template <typename real>
__device__ __forceinline__ bool nbr_select_keys(const Consts<real> &c, int N, int i, int base, int B, const real *s_pos, const real *s_vel,
                                                const real mypos[3], const real myvel[3], int bi[8]) {
    const int K = c.num_neighbors;
    NbrKeys L;
    L.init(N);
    for (int j0 = 0; j0 < N; j0 += 4) {
        float m4[4];
#pragma unroll
        for (int h = 0; h < 2; ++h) {   // two partners per packed instruction (neighbouring LDS words = an aligned register pair)
            const int ja = (j0 + 2 * h < N) ? j0 + 2 * h : N - 1, jb = (j0 + 2 * h + 1 < N) ? j0 + 2 * h + 1 : N - 1;
            qs_f32x2 rp[3], rv[3];
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                const qs_f32x2 pj = {(float)s_pos[a * B + base + ja], (float)s_pos[a * B + base + jb]};
                const qs_f32x2 vj = {(float)s_vel[a * B + base + ja], (float)s_vel[a * B + base + jb]};
                rp[a] = pj - (float)mypos[a]; rv[a] = vj - (float)myvel[a];
            }
            const qs_f32x2 m = nbr_metric_x2(rp, rv);
            m4[2 * h] = m.x; m4[2 * h + 1] = m.y;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) L.insert(m4[u], j0 + u, (j0 + u < N) & (j0 + u != i), K);
    }
    return L.finish(K, bi);
}

template <typename real>
__device__ __forceinline__ void nbr_select(const Consts<real> &c, int N, int i, int base, int B, int tid, const real *s_pos,
    const real *s_vel,
                                           real *s_metric, const real mypos[3], const real myvel[3], NbrSel<real> &S) {
    const int K = c.num_neighbors;
    S.taken = 1ull << i;
    if (K <= 0 || K == N - 1) return;
#ifndef QS_EXACT_NBR_SELECT   // (-DQS_EXACT_NBR_SELECT: every drone on the exact path below; tests/test_object_identity_gpu.py)
    if constexpr (sizeof(real) == 4) {
        // (N <= 8: rank-by-counting below stays - measured at 8 x 131072, profiles/r06s2_ab_keys_packed_scan.txt: 97.8 us against 99.2 with keys)
        if (K <= 8 && N > 8 && nbr_select_keys<real>(c, N, i, base, B, s_pos, s_vel, mypos, myvel, S.bi)) return;
    }
#endif
    if (N <= 8) {   // all candidates at once, rank-by-counting (independent compares), then the inverse permutation: only the 8 indices
        real mj[8];  // stay live (the emit re-reads the chosen drones from LDS: registers decide the occupancy of this kernel)
        int rank[8];
        {
            real rp[8][3], rv[8][3];
#pragma unroll
            for (int u = 0; u < 8; ++u) {        // 48 LDS reads issued before the first use: one round trip
                const int j = (u < N) ? u : N - 1;
#pragma unroll
                for (int a = 0; a < 3; ++a) { rp[u][a] = s_pos[a * B + base + j] - mypos[a];
                    rv[u][a] = s_vel[a * B + base + j] - myvel[a]; }
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                real rd = M<real>::fmax(norm3<real>(rp[u]), (real)0.01);
                real m = rd + (rp[u][0] * rv[u][0] + rp[u][1] * rv[u][1] + rp[u][2] * rv[u][2]) * M<real>::rcp(rd);
                mj[u] = (u < N && u != i) ? m : (real)3.4e38;
                rank[u] = 0;
            }
        }
#pragma unroll
        for (int k = 0; k < 8; ++k)
#pragma unroll
            for (int u = 0; u < 8; ++u) rank[u] += (int)((mj[k] < mj[u]) | ((mj[k] == mj[u]) & (k < u)));
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            int b = 0;
#pragma unroll
            for (int u = 1; u < 8; ++u) b = (rank[u] == k) ? u : b;   // the ranks are a permutation of 0..7
            S.bi[k] = b;
        }
        return;
    }
    if (K <= 8) {   // streaming sorted top-8 (metric, index) list; a candidate goes behind entries with an equal metric (lower index first)
        real bm[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) { bm[k] = (real)3.4e38; S.bi[k] = 0; }
        for (int j0 = 0; j0 < N; j0 += 4) {
            real rp[4][3], rv[4][3];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int j = (j0 + u < N) ? j0 + u : N - 1;
#pragma unroll
                for (int a = 0; a < 3; ++a) { rp[u][a] = s_pos[a * B + base + j] - mypos[a];
                    rv[u][a] = s_vel[a * B + base + j] - myvel[a]; }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int j = j0 + u;
                real rd = M<real>::fmax(norm3<real>(rp[u]), (real)0.01);
                real m = rd + (rp[u][0] * rv[u][0] + rp[u][1] * rv[u][1] + rp[u][2] * rv[u][2]) * M<real>::rcp(rd);
                m = (j < N && j != i) ? m : (real)3.4e38;
                int mi = j;
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const bool lt = m < bm[k];
                    const real tm = lt ? bm[k] : m;
                    const int ti = lt ? S.bi[k] : mi;
                    bm[k] = lt ? m : bm[k];
                    S.bi[k] = lt ? mi : S.bi[k];
                    m = tm; mi = ti;
                }
            }
        }
        return;
    }
    for (int j = 0; j < N; ++j) {   // 8 < K < N-1: metrics into this lane's LDS column
        real rp[3] = {s_pos[0 * B + base + j] - mypos[0], s_pos[1 * B + base + j] - mypos[1], s_pos[2 * B + base + j] - mypos[2]};
        real rv[3] = {s_vel[0 * B + base + j] - myvel[0], s_vel[1 * B + base + j] - myvel[1], s_vel[2 * B + base + j] - myvel[2]};
        real rd = M<real>::fmax(norm3<real>(rp), (real)0.01);
        real mm = rd + (rp[0] * rv[0] + rp[1] * rv[1] + rp[2] * rv[2]) * M<real>::rcp(rd);
        s_metric[j * B + tid] = (j == i) ? (real)3.4e38 : mm;
    }
}
// neighbours of ranks [k0, k1) -> o[(rank - k0) * 6 ..]   (o = this lane's dense stage row of (k1 - k0) * 6 columns)
template <typename real>
__device__ __forceinline__ void nbr_emit(const Consts<real> &c, int N, int i, int base, int B, int tid, const real *s_pos,
    const real *s_vel,
                                         const real *s_metric, const real mypos[3], const real myvel[3], NbrSel<real> &S, int k0, int k1,
                                             real *o) {
    const int K = c.num_neighbors;
    for (int k = k0; k < k1; ++k) {
        int j;
        if (K == N - 1) j = (k < i) ? k : k + 1;   // all other drones in index order (:253-254)
        else if (K <= 8) {
            j = S.bi[0];
#pragma unroll
            for (int q = 1; q < 8; ++q) j = (q == k) ? S.bi[q] : j;   // register array: select, no dynamic indexing
        } else {   // arg-min round over the LDS metric column (lowest index wins ties)
            int best = -1;
            real bmin = (real)3.4e38;
            for (int jj = 0; jj < N; ++jj) {
                real mm = s_metric[jj * B + tid];
                bool better = !(S.taken >> jj & 1) && (best < 0 || mm < bmin);
                best = better ? jj : best;
                bmin = better ? mm : bmin;
            }
            S.taken |= 1ull << best;
            j = best;
        }
        real vals[6];
#pragma unroll
        for (int a = 0; a < 3; ++a) { vals[a] = s_pos[a * B + base + j] - mypos[a]; vals[3 + a] = s_vel[a * B + base + j] - myvel[a]; }
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            o[(k - k0) * 6 + a] = clipr<real>(vals[a], -c.nbr_clip_pos[a], c.nbr_clip_pos[a]);
            o[(k - k0) * 6 + 3 + a] = clipr<real>(vals[3 + a], -c.nbr_clip_vel[a], c.nbr_clip_vel[a]);
        }
    }
}

// get_surround_sdfs obstacles/utils.py:5-27 (obstacle xy of the env in LDS)
template <typename real>
__device__ __forceinline__ void sdf_obs(const Consts<real> &c, const real *ox, const real *oy, int M_, real px, real py, real *o,
    real radius) {
    const real res = (real)0.1;
    real gx[3] = {px - res, px, px + res}, gy[3] = {py - res, py, py + res};
    // min over the obstacles of the SQUARED distance, one square root per cell at the end: the square root is monotone, so
    // min_k sqrt(d2_k) == sqrt(min_k d2_k) - 9 instead of 9 * M quarter-rate v_sqrt_f32 per drone and step (the reference's start value
    // 100.0, utils.py:17, is the cap 100^2 on the squared side)
    real mind2[9];
#pragma unroll
    for (int q = 0; q < 9; ++q) mind2[q] = (real)10000;
    for (int k = 0; k < M_; ++k) {
        real x = ox[k], y = oy[k];
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int b = 0; b < 3; ++b) {
                real dx = gx[a] - x, dy = gy[b] - y, d2 = dx * dx + dy * dy;
                mind2[a * 3 + b] = d2 < mind2[a * 3 + b] ? d2 : mind2[a * 3 + b];
            }
    }
#pragma unroll
    for (int q = 0; q < 9; ++q) o[q] = M<real>::sqrt(mind2[q]) - radius;
}

// The neighbour and SDF columns of the wave's rows -> rows stage -> HBM together with the self columns (obs_copy_rows), one row
// group at a time.  Whole wave; `live` lanes compute (their rows are the ones in rowmask).  The stage shares LDS with the
// rare-path buffers: nothing of those may be live here; ends with the stage free again (RP < 64) or still being read (RP = 64:
// the caller's next barrier).
template <typename real>
__device__ __forceinline__ void stream_rows(const Consts<real> &c, const LdsLayout &L, real *__restrict__ dst_block, const real *s_self,
    real *s_rows, int N, int i, int le, int base, int tid,
                                            const real *s_pos, const real *s_vel, real *s_metric, const real *s_obst,
                                                const real mypos[3], const real myvel[3],
                                            bool live, int nrows, uint64_t rowmask, real obst_radius) {
    const int B = QS_WAVE, K = c.num_neighbors, D = c.obs_dim, S = c.self_dim, X = D - S, M_ = c.num_obstacles, RP = L.rows_per_pass;
    NbrSel<real> sel;
    if (live) nbr_select<real>(c, N, i, base, B, tid, s_pos, s_vel, s_metric, mypos, myvel, sel);
#ifdef QS_SPEC
    if (RP < B) {   // (config-specialised kernels only: X is a literal <= QS_NV_MAX, the register array below has static indices)
        real nv[QS_NV_MAX];
        if (live) {
            nbr_emit<real>(c, N, i, base, B, tid, s_pos, s_vel, s_metric, mypos, myvel, sel, 0, K, nv);
            if (c.use_obstacles) sdf_obs<real>(c, s_obst + (le * 2 + 0) * M_, s_obst + (le * 2 + 1) * M_, M_, mypos[0], mypos[1],
                nv + 6 * K, obst_radius);
        }
        for (int r0 = 0; r0 < nrows; r0 += RP) {
            if (live && tid >= r0 && tid < r0 + RP) {
                real *o = s_rows + (tid - r0) * X;
#pragma unroll
                for (int q = 0; q < QS_NV_MAX; ++q) if (q < X) o[q] = nv[q];
            }
            __syncthreads();
            obs_copy_rows<real>(dst_block, s_self, s_rows, S, D, r0, (nrows - r0) < RP ? (nrows - r0) : RP, rowmask, tid);
            __syncthreads();   // the group has left the stage
        }
        return;
    }
#endif
    if (live) {   // complete rows at once: straight into the stage
        real *o = s_rows + tid * X;
        nbr_emit<real>(c, N, i, base, B, tid, s_pos, s_vel, s_metric, mypos, myvel, sel, 0, K, o);
        if (c.use_obstacles) sdf_obs<real>(c, s_obst + (le * 2 + 0) * M_, s_obst + (le * 2 + 1) * M_, M_, mypos[0], mypos[1], o + 6 * K,
            obst_radius);
    }
    __syncthreads();
    obs_copy_rows<real>(dst_block, s_self, s_rows, S, D, 0, nrows, rowmask, tid);
}

// The same response in two parts (pair_responses, parallel form).  None of the response's random draws depends on the state - Philox is keyed
// by (env, step, site, slot, i, j) - and they are nine tenths of its cost (11 Philox blocks, 27 Box-Muller normals): one LANE per Philox
// block draws them into a table, tbl = [3 attempts][cons, n1, n2][3] | decay[2] | omega words[4]; what has to follow the reference's list order
// (a later pair reads the velocities an earlier one left) is the arithmetic below, a tenth of the work.
#define QS_DD_CALLS 11    // Philox blocks behind one pair: 9 normal triples (slot = attempt * 3 + {cons, n1, n2}), the decay pair, the omega words
#define QS_DD_DRAWS 33
#define QS_DD_CHUNK 5     // pairs drawn at once by one wave: 55 lanes
template <typename real>
__device__ __forceinline__ void collide_drones_draw(const RngKey &key, int call, int i, int j, real *tbl) {
    uint32_t w[4];
    rng_words(key, call < 9 ? QS_SITE_DD_N : (call == 9 ? QS_SITE_DD_U : QS_SITE_DD_W), call < 9 ? call : 0, i, j, w);
    if (call < 9) {   // rng_normal_s<real, 3>
        real z[4];
        box_muller4<real>(w, z);
        const real scale = (call % 3 == 0) ? (real)0.8 : (real)0.15;
#pragma unroll
        for (int q = 0; q < 3; ++q) tbl[call * 3 + q] = scale * z[q];
    } else if (call == 9) {   // rng_uniform<real, 2>(0.2, 0.8)
        tbl[27] = (real)0.2 + ((real)0.8 - (real)0.2) * u01<real>(w[0]);
        tbl[28] = (real)0.2 + ((real)0.8 - (real)0.2) * u01<real>(w[1]);
    } else {   // uniform(-1, 1, 3), uniform(10 pi, 20 pi)
#pragma unroll
        for (int q = 0; q < 3; ++q) tbl[29 + q] = (real)-1 + (real)2 * u01<real>(w[q]);
        tbl[32] = (real)(10.0 * QS_PI_D) + (real)(20.0 * QS_PI_D - 10.0 * QS_PI_D) * u01<real>(w[3]);
    }
}
template <typename real>
__device__ __forceinline__ void collide_drones_apply(const real *tbl, int i, int j, int base, int B, const real *s_pos, real *s_vel, real *s_om) {
    real p1[3], p2[3], v1[3], v2[3];
    for (int q = 0; q < 3; ++q) { p1[q] = s_pos[q * B + base + i]; p2[q] = s_pos[q * B + base + j]; v1[q] = s_vel[q * B + base + i];
        v2[q] = s_vel[q * B + base + j]; }
    real n[3] = {p1[0] - p2[0], p1[1] - p2[1], p1[2] - p2[2]};
    real mag = norm3<real>(n), den = (mag == (real)0) ? mag + (real)1e-5 : mag;
    for (int q = 0; q < 3; ++q) n[q] /= den;
    real v1n = dot3<real>(v1, n), v2n = dot3<real>(v2, n);
    real vc[3] = {(v2n - v1n) * n[0], (v2n - v1n) * n[1], (v2n - v1n) * n[2]};
    real s1[3] = {vc[0], vc[1], vc[2]}, s2[3] = {-vc[0], -vc[1], -vc[2]};
    for (int t = 0; t < 3; ++t) {
        real t1[3], t2[3];
        for (int q = 0; q < 3; ++q) {
            real a = tbl[t * 9 + q] + tbl[t * 9 + 3 + q], b = -tbl[t * 9 + q] + tbl[t * 9 + 6 + q];
            s1[q] = vc[q] + a; s2[q] = -vc[q] + b;
            t1[q] = v1[q] + s1[q]; t2[q] = v2[q] + s2[q];
        }
        if (dot3<real>(t1, n) > (real)0 && (real)0 > dot3<real>(t2, n)) break;
    }
    real maxv = M<real>::fmax(norm3<real>(v1), norm3<real>(v2));
    compute_new_vel<real>(maxv, v1, s1, tbl[27]);
    compute_new_vel<real>(maxv, v2, s2, tbl[28]);
    const real u[4] = {tbl[29], tbl[30], tbl[31], tbl[32]};
    real dw[3]; compute_new_omega<real>(u, dw);
    for (int q = 0; q < 3; ++q) {
        s_vel[q * B + base + i] = v1[q]; s_vel[q * B + base + j] = v2[q];
        s_om[q * B + base + i] += dw[q]; s_om[q * B + base + j] -= dw[q];
    }
}

// perform_collision_between_drones collisions/quadrotors.py:24-59 on LDS-resident vel/omega (serial per env)
template <typename real>
__device__ __forceinline__ void collide_drones_lds(const RngKey &key, int i, int j, int base, int B, const real *s_pos, real *s_vel,
    real *s_om) {
    real p1[3], p2[3], v1[3], v2[3];
    for (int q = 0; q < 3; ++q) { p1[q] = s_pos[q * B + base + i]; p2[q] = s_pos[q * B + base + j]; v1[q] = s_vel[q * B + base + i];
        v2[q] = s_vel[q * B + base + j]; }
    real n[3] = {p1[0] - p2[0], p1[1] - p2[1], p1[2] - p2[2]};
    real mag = norm3<real>(n), den = (mag == (real)0) ? mag + (real)1e-5 : mag;
    for (int q = 0; q < 3; ++q) n[q] /= den;
    real v1n = dot3<real>(v1, n), v2n = dot3<real>(v2, n);
    real vc[3] = {(v2n - v1n) * n[0], (v2n - v1n) * n[1], (v2n - v1n) * n[2]};
    real s1[3] = {vc[0], vc[1], vc[2]}, s2[3] = {-vc[0], -vc[1], -vc[2]};
    for (int t = 0; t < 3; ++t) {
        real cons[3], n1[3], n2[3], t1[3], t2[3];
        rng_normal_s<real, 3>(key, QS_SITE_DD_N, t * 3 + 0, i, j, (real)0.8, cons);
        rng_normal_s<real, 3>(key, QS_SITE_DD_N, t * 3 + 1, i, j, (real)0.15, n1);
        rng_normal_s<real, 3>(key, QS_SITE_DD_N, t * 3 + 2, i, j, (real)0.15, n2);
        for (int q = 0; q < 3; ++q) {
            real a = cons[q] + n1[q], b = -cons[q] + n2[q];
            s1[q] = vc[q] + a; s2[q] = -vc[q] + b;
            t1[q] = v1[q] + s1[q]; t2[q] = v2[q] + s2[q];
        }
        if (dot3<real>(t1, n) > (real)0 && (real)0 > dot3<real>(t2, n)) break;
    }
    real maxv = M<real>::fmax(norm3<real>(v1), norm3<real>(v2));
    real dec[2]; rng_uniform<real, 2>(key, QS_SITE_DD_U, 0, i, j, (real)0.2, (real)0.8, dec);
    compute_new_vel<real>(maxv, v1, s1, dec[0]);
    compute_new_vel<real>(maxv, v2, s2, dec[1]);
    real u[4];
    if (QS_ON_TAPE(key)) { for (int q = 0; q < 4; ++q) u[q] = (real)tape_pop(key); }   // uniform(-1,1,3), uniform(10 pi, 20 pi)
    else {
        uint32_t w[4]; rng_words(key, QS_SITE_DD_W, 0, i, j, w);
        u[0] = (real)-1 + (real)2 * u01<real>(w[0]); u[1] = (real)-1 + (real)2 * u01<real>(w[1]);
        u[2] = (real)-1 + (real)2 * u01<real>(w[2]);
        u[3] = (real)(10.0 * QS_PI_D) + (real)(20.0 * QS_PI_D - 10.0 * QS_PI_D) * u01<real>(w[3]);
    }
    real dw[3]; compute_new_omega<real>(u, dw);
    for (int q = 0; q < 3; ++q) {
        s_vel[q * B + base + i] = v1[q]; s_vel[q * B + base + j] = v2[q];
        s_om[q * B + base + i] += dw[q]; s_om[q * B + base + j] -= dw[q];
    }
}

// rare per-drone responses kept out of line so the hot path stays compact
template <typename real>
__device__ __forceinline__ void room_obst_responses(const Consts<real> *cp, const RngKey &key, int i, uint32_t bits, real ox, real oy,
    real pos[3], real vel[3], real omega[3],
                                                    real obst_size) {
    const Consts<real> &c = *cp;
    Drone<real> d;
#pragma unroll
    for (int q = 0; q < 3; ++q) { d.pos[q] = pos[q]; d.vel[q] = vel[q]; d.omega[q] = omega[q]; }
    if (bits & B_OBST_NEW) collide_obstacle<real>(c, key, i, d, ox, oy, obst_size);      // collisions/obstacles.py:23-50
    if (bits & B_WALL_NEW) collide_room<real>(c, key, i, d, true);            // collisions/room.py:6-44
    if (bits & B_CEIL_NEW) collide_room<real>(c, key, i, d, false);           // collisions/room.py:91-113
#pragma unroll
    for (int q = 0; q < 3; ++q) { vel[q] = d.vel[q]; omega[q] = d.omega[q]; }
}

// Team kernels, more than 8 drones, K <= 8 (qs_step_team.inc, pair-once): wave wv >= 1 ranks ITS candidates j = (wv-1), (wv-1) + (W-1), ...
// of drone i's row of the metric matrix by counting inside the stripe (all keys in registers; every compare feeds two ranks), and
// publishes the 8 nearest of them compacted by local rank: s_topk slot (wv-1)*8 + rank.  recompute: the metrics are evaluated from the
// current positions / velocities instead of read from the matrix (environments whose velocities changed after the pair scan).
// (the candidate count is a literal in a config-specialised object: register arrays sized for 64 drones there keep the compiler from
// promoting the kernel's constant block out of scratch memory)
#ifdef QS_SPEC_N
#define QS_TEAM_NMAX QS_SPEC_N
#else
#define QS_TEAM_NMAX QS_MAX_AGENTS
#endif
template <typename real, int W>
__device__ __forceinline__ void team_rank_local(int N, int i, int base, int B, int tid, int wv, const real *s_pos, const real *s_vel,
                                                const real *s_metric, unsigned char *s_topk, bool recompute) {
    constexpr int RW = W - 1, CMAX = (QS_TEAM_NMAX + RW - 1) / RW, SL = RW * 8;
    const int C = (N + RW - 1) / RW;
    NbrKey<real> key[CMAX];
    int lr[CMAX];
    real mp[3] = {0, 0, 0}, mv[3] = {0, 0, 0};
    if (recompute) {
#pragma unroll
        for (int q = 0; q < 3; ++q) { mp[q] = s_pos[q * B + tid]; mv[q] = s_vel[q * B + tid]; }
    }
#pragma unroll
    for (int u = 0; u < CMAX; ++u) {
        lr[u] = 0;
        if (u < C) {
            const int j = (wv - 1) + RW * u, jc = j < N ? j : N - 1;
            const bool valid = (j < N) & (j != i);
            real m;
            if (recompute) {
                const real rp[3] = {s_pos[0 * B + base + jc] - mp[0], s_pos[1 * B + base + jc] - mp[1], s_pos[2 * B + base + jc] - mp[2]};
                const real rv[3] = {s_vel[0 * B + base + jc] - mv[0], s_vel[1 * B + base + jc] - mv[1], s_vel[2 * B + base + jc] - mv[2]};
                m = nbr_metric<real>(rp, rv);
            } else {
                m = s_metric[jc * B + tid];
            }
            key[u] = NbrKey<real>::make(valid ? m : (real)3.4e38, valid ? j : 0x7fffff00 + u);
        }
    }
#pragma unroll
    for (int u = 0; u < CMAX; ++u) {
#pragma unroll
        for (int k = u + 1; k < CMAX; ++k) {
            if (k < C) {   // keys are distinct (the index breaks ties): exactly one of the two is the smaller
                const int lt = (int)key[u].less(key[k]);
                lr[k] += lt; lr[u] += 1 - lt;
            }
        }
    }
#pragma unroll
    for (int u = 0; u < CMAX; ++u) {
        if (u < C && lr[u] < 8) NbrKey<real>::st(s_topk, SL, (wv - 1) * 8 + lr[u], B, tid, key[u]);
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        // fewer candidates than slots
        if (q >= C) NbrKey<real>::st(s_topk, SL, (wv - 1) * 8 + q, B, tid, NbrKey<real>::make((real)3.4e38, 0x7fffff00 + q));
    }
}

// full-scenario kernels: per-env scenario state HBM <-> LDS (each drone of the env moves a strided part)
template <typename real>
__device__ __forceinline__ void scen_lds_load(const Ptrs<real> &p, const LdsLayout &L, unsigned char *smem, int E, int e, int le, int i,
    int N) {
    real *sr = (real *)(smem + L.off_sr) + le * SR_COUNT;
    int *si = (int *)(smem + L.off_si) + le * SI_COUNT;
    uint64_t *om = (uint64_t *)(smem + L.off_omap) + le * 4;
    for (int k = i; k < SR_COUNT; k += N) sr[k] = p.scen_real[k * E + e];
    for (int k = i; k < SI_COUNT; k += N) si[k] = p.scen_int[k * E + e];
    for (int k = i; k < 4; k += N) om[k] = p.scen_omap[k * E + e];
}
#ifdef QS_SPEC_N
// The same load in two halves for the latency-bound team kernels: the global loads are issued with the state loads, the LDS writes wait
// until the sub-steps are done (a write right behind its load makes the wave wait for EVERY load in flight before it has drawn its noise).
template <typename real> struct ScenRegs {
    real r[(SR_COUNT + QS_SPEC_N - 1) / QS_SPEC_N];
    int n[(SI_COUNT + QS_SPEC_N - 1) / QS_SPEC_N];
    uint64_t o[(4 + QS_SPEC_N - 1) / QS_SPEC_N];
};
template <typename real>
__device__ __forceinline__ void scen_regs_load(const Ptrs<real> &p, int E, int e, int i, ScenRegs<real> &v) {
    constexpr int N = QS_SPEC_N;
#pragma unroll
    for (int j = 0; j < (SR_COUNT + N - 1) / N; ++j) { const int k = i + j * N; v.r[j] = (k < SR_COUNT)
        ? p.scen_real[k * E + e] : (real)0; }
#pragma unroll
    for (int j = 0; j < (SI_COUNT + N - 1) / N; ++j) { const int k = i + j * N; v.n[j] = (k < SI_COUNT) ? p.scen_int[k * E + e] : 0; }
#pragma unroll
    for (int j = 0; j < (4 + N - 1) / N; ++j) { const int k = i + j * N; v.o[j] = (k < 4) ? p.scen_omap[k * E + e] : 0ull; }
}
template <typename real>
__device__ __forceinline__ void scen_regs_to_lds(const LdsLayout &L, unsigned char *smem, int le, int i, const ScenRegs<real> &v) {
    constexpr int N = QS_SPEC_N;
    real *sr = (real *)(smem + L.off_sr) + le * SR_COUNT;
    int *si = (int *)(smem + L.off_si) + le * SI_COUNT;
    uint64_t *om = (uint64_t *)(smem + L.off_omap) + le * 4;
#pragma unroll
    for (int j = 0; j < (SR_COUNT + N - 1) / N; ++j) { const int k = i + j * N; if (k < SR_COUNT) sr[k] = v.r[j]; }
#pragma unroll
    for (int j = 0; j < (SI_COUNT + N - 1) / N; ++j) { const int k = i + j * N; if (k < SI_COUNT) si[k] = v.n[j]; }
#pragma unroll
    for (int j = 0; j < (4 + N - 1) / N; ++j) { const int k = i + j * N; if (k < 4) om[k] = v.o[j]; }
}
#endif
template <typename real>
__device__ __forceinline__ void scen_lds_store(const Ptrs<real> &p, const LdsLayout &L, unsigned char *smem, int E, int e, int le, int i,
    int N) {
    const real *sr = (const real *)(smem + L.off_sr) + le * SR_COUNT;
    const int *si = (const int *)(smem + L.off_si) + le * SI_COUNT;
    const uint64_t *om = (const uint64_t *)(smem + L.off_omap) + le * 4;
    for (int k = i; k < SR_COUNT; k += N) p.scen_real[k * E + e] = sr[k];
    for (int k = i; k < SI_COUNT; k += N) p.scen_int[k * E + e] = si[k];
    for (int k = i; k < 4; k += N) p.scen_omap[k * E + e] = om[k];
}

// ------------------------------------------------------------------------------------------------
// Episode reset of the envs whose lanes have `do_reset` set: QuadrotorEnvMulti.reset quadrotor_multi.py:339-411
// (+ QuadrotorSingle._reset quadrotor_single.py:387-447, obst_generation_given_density quadrotor_multi.py:304-325,
// scenario.reset()).  Shared by the reset kernel and the tail of the step kernel (auto-reset inside step, :720).
// Must be entered by the whole workgroup (contains barriers).  Outputs: d / goal (registers), the obs row in LDS,
// per-env global scratch (obstacle positions, scenario state).  `stale_vel` are the previous episode's final
// velocities: the first neighbour obs of an episode is computed from them (SURVEY App. A reset quirk).
// ------------------------------------------------------------------------------------------------
template <typename real, bool FULL, bool TEAM = false, bool STREAM = false>
__device__ __forceinline__ void reset_body(const Consts<real> *cp, const Ptrs<real> *pp, const LdsLayout *Lp, unsigned char *smem,
    int epb, const RngKey &key,
                                        bool do_reset, Drone<real> *dp, real goal[3], const real stale_vel[3], int blk = -1) {
    const int bidx = blk < 0 ? (int)blockIdx.x : blk;
    const Consts<real> &c = *cp;
    const Ptrs<real> &p = *pp;
    const LdsLayout &L = *Lp;
    Drone<real> &d = *dp;
    const int B = QS_WAVE, N = c.num_agents, E = c.num_envs;
    real *s_pos = (real *)(smem + L.off_pos), *s_vel = (real *)(smem + L.off_vel), *s_spawn = (real *)(smem + L.off_zax);
    real *s_goal = (real *)(smem + L.off_goal), *s_obs = (real *)(smem + L.off_obs), *s_obst = (real *)(smem + L.off_obst);
    real *s_metric = (real *)(smem + L.off_metric);
    uint32_t *s_envflag = (uint32_t *)(smem + L.off_envflag);
    const int tid = threadIdx.x, le = tid / N, i = tid - le * N, e = bidx * epb + le, base = le * N;
    const int M_ = c.num_obstacles;
    // STREAM: dense self block + rows stage
    real *myobs = STREAM ? (real *)(smem + L.off_self) + tid * c.self_dim : s_obs + tid * c.obs_dim;
    int *tidx = (int *)(smem + L.off_scratch) + le * (2 * L.scr_cap + 32), *tval = tidx + L.scr_cap, *prev_row = tidx + 2 * L.scr_cap,
        *cur_row = prev_row + 16;

#ifdef QS_TAPE
    int *s_cur = (int *)(smem + L.off_cur);
#endif
    // ---- per-env part (one lane): obstacle map + scenario.reset() ----
    if (do_reset && i == 0) {
#ifdef QS_TAPE
        if (QS_ON_TAPE(key)) *key.cur = s_cur[le];   // this lane consumes the env's tape sequentially
#endif
        real *goals = s_goal + le * L.goal_rows * 3;
        uint32_t have_spawn = 0;
        uint64_t omap[4] = {0, 0, 0, 0};   // obstacle map bitset, cell id = rid*W + cid
        const int Lr = c.obst_area[0], W = c.obst_area[1], cells = Lr * W;
        if (FULL) for (int q = 0; q < 4; ++q) ((uint64_t *)(smem + L.off_omap))[le * 4 + q] = 0;
        int Me = M_;   // obstacles of this episode
        if (c.use_obstacles) {
            // --quads_domain_random: this episode's density / size (quad_experience_replay.py:106-118,:191-206 -> reset(obst_density,
            // obst_size))
            if (c.dr_on) {
                if (c.dr_num_density > 0) {
                    const int k = rng_index<real>(key, QS_SITE_REPLAY, 2, c.dr_num_density);
                    Me = p.dr_count[k];
                    p.obst_density_env[e] = p.dr_density[k];
                }
                if (c.dr_num_size > 0) {
                    const int k = rng_index<real>(key, QS_SITE_REPLAY, 3, c.dr_num_size);
                    p.obst_size_env[e] = p.dr_size[k];
                    s_envflag[2 * epb + le] = (uint32_t)k;   // for the first observation of the episode, later in this launch
                }
                p.obst_count[e] = Me;
            }
            // np.random.choice(cells, M, replace=False): partial Fisher-Yates on a virtual pool
            int nt = 0;
            for (int k = 0; k < M_; ++k) {
                if (k >= Me) {   // unused slot: parked where no first-hit test and no SDF minimum can see it
                    p.obst_pos[(size_t)e * M_ + k] = (real)1e6; p.obst_pos[(size_t)E * M_ + (size_t)e * M_ + k] = (real)1e6;
                    s_obst[(le * 2 + 0) * M_ + k] = (real)1e6; s_obst[(le * 2 + 1) * M_ + k] = (real)1e6;
                    continue;
                }
                int vj;
                if (QS_ON_TAPE(key)) vj = (int)tape_pop(key);   // the tape holds the chosen cell ids (quadrotor_multi.py:313)
                else {
                    int j = k + rng_index<real>(key, QS_SITE_OBST_MAP, k, cells - k);
                    int vk = k, pj = -1;
                    vj = j;
                    for (int q = 0; q < nt; ++q) { if (tidx[q] == k) vk = tval[q]; if (tidx[q] == j) { vj = tval[q]; pj = q; } }
                    if (pj >= 0) tval[pj] = vk; else { tidx[nt] = j; tval[nt] = vk; ++nt; }   // pool[j] = pool[k]; the pick is old pool[j]
                }
                int id = vj, rid = id / W, cid = id - rid * W;
                omap[id >> 6] |= 1ull << (id & 63);
                if (FULL) ((uint64_t *)(smem + L.off_omap))[le * 4 + (id >> 6)] |= 1ull << (id & 63);
                // cell centre index rid + L*cid (quadrotor_multi.py:321); centres per obstacles/utils.py:47-58
                int ci = rid + Lr * cid, ii = ci / W, jj = (W - 1) - (ci - ii * W);
                real ox = (real)ii + (real)0.5 - (real)(Lr / 2), oy = (real)jj + (real)0.5 - (real)(W / 2);
                p.obst_pos[(size_t)e * M_ + k] = ox;
                p.obst_pos[(size_t)E * M_ + (size_t)e * M_ + k] = oy;
                s_obst[(le * 2 + 0) * M_ + k] = ox;
                s_obst[(le * 2 + 1) * M_ + k] = oy;
            }
        }
        Formation<real> F;
        if (FULL) {
            ScenCtx<real> x = {(real *)(smem + L.off_sr) + le * SR_COUNT, (int *)(smem + L.off_si) + le * SI_COUNT,
                               (uint64_t *)(smem + L.off_omap) + le * 4, goals, s_spawn, B, base, N};
            scenario_reset_full<real>(c, key, x, tidx);
            have_spawn = (uint32_t)x.si[SI_HAVE_SPAWN];
            s_envflag[epb + le] = (uint32_t)x.si[SI_PERIOD];
            p.scenario_id[e] = x.si[SI_SCEN];
        } else if (c.scenario == QS_SCENARIO_STATIC_SAME_GOAL) {
            update_formation<real>(c.scenario, key, 0, N, F);
            real center[3] = {0, 0, 2};
            const int rows = generate_goals<real>(F, N, 1, center, goals, 3);
            if (QS_ON_TAPE(key)) tape_skip(key, rows);   // np.random.shuffle(goals) of base.py:151: the rows are all the same point
            (void)rows;
        } else if (c.scenario == QS_SCENARIO_O_STATIC_SAME_GOAL) {
            // obstacles/o_static_same_goal.py:27-48 + o_base.py:69-81,:124-153
            int nfree = cells - Me;
            int nt = 0;
            const bool on_tape = QS_ON_TAPE(key);
            if (on_tape) tape_skip(key, 1);   // the duration draw of o_static_same_goal.py:29 (unused by a static goal)
            for (int k = 0; k < N; ++k) {
                int vj;
                if (on_tape) vj = (int)tape_pop(key);   // np.random.choice(free cells, N, replace=False): the ids themselves
                else {
                    int j = k + rng_index<real>(key, QS_SITE_SCEN, 16 + k, nfree - k);
                    int vk = k, pj = -1;
                    vj = j;
                    for (int q = 0; q < nt; ++q) { if (tidx[q] == k) vk = tval[q]; if (tidx[q] == j) { vj = tval[q]; pj = q; } }
                    if (pj >= 0) tval[pj] = vk; else { tidx[nt] = j; tval[nt] = vk; ++nt; }
                }
                int seen = 0, cell = 0;   // vj-th free cell in row-major order (np.where(obst_map == 0))
                for (int id = 0; id < cells; ++id) if (!(omap[id >> 6] >> (id & 63) & 1)) { if (seen == vj) { cell = id; break; } ++seen; }
                int x = cell / W, y = cell - x * W, index = x + Lr * y, ii = index / W, jj = (W - 1) - (index - ii * W);
                s_spawn[0 * B + base + k] = (real)ii + (real)0.5 - (real)(Lr / 2);
                s_spawn[1 * B + base + k] = (real)jj + (real)0.5 - (real)(W / 2);
                if (!on_tape) s_spawn[2 * B + base + k] = rng_uniform1<real>(key, QS_SITE_SCEN, 96 + k, 0, 0, (real)1, (real)3);
            }
            // the N heights follow the N ids (o_base.py:69-81)
            if (on_tape) for (int k = 0; k < N; ++k) s_spawn[2 * B + base + k] = (real)tape_pop(key);
            have_spawn = 1;
            // max_square_area_center o_base.py:124-153 (two-row dynamic programme)
            int max_size = 0, cx = 0, cy = 0;
            for (int q = 0; q < W; ++q) prev_row[q] = (int)(omap[q >> 6] >> (q & 63) & 1);
            for (int r = 1; r < Lr; ++r) {
                int id0 = r * W;
                cur_row[0] = (int)(omap[id0 >> 6] >> (id0 & 63) & 1);
                for (int q = 1; q < W; ++q) {
                    int id = r * W + q;
                    cur_row[q] = 0;
                    if (!(omap[id >> 6] >> (id & 63) & 1)) {
                        int m = prev_row[q] < cur_row[q - 1] ? prev_row[q] : cur_row[q - 1];
                        if (prev_row[q - 1] < m) m = prev_row[q - 1];
                        cur_row[q] = m + 1;
                        if (cur_row[q] > max_size) { max_size = cur_row[q]; cx = r - (max_size - 1) / 2; cy = q - (max_size - 1) / 2; }
                    }
                }
                for (int q = 0; q < W; ++q) prev_row[q] = cur_row[q];
            }
            int index = cx + W * cy, ii = index / W, jj = (W - 1) - (index - ii * W);
            real end[3] = {(real)ii + (real)0.5 - (real)(Lr / 2), (real)jj + (real)0.5 - (real)(W / 2), 0};
            end[2] = rng_uniform1<real>(key, QS_SITE_SCEN, 9, 0, 0, (real)1.5, (real)3);
            // update_formation_and_relate_param (:41): formation index, size, layer distance - unused here
            if (on_tape) tape_skip(key, 3);
            for (int k = 0; k < N; ++k) for (int q = 0; q < 3; ++q) goals[k * 3 + q] = end[q];
        } else {
            // swarm_vs_swarm.py:80-94 (reset) + :17-50 (formation_centers) + scenarios/utils.py:170-181 (get_z_value)
            const int period = draw_period<real>(key, 8, 4.0, 6.0, c.control_freq);
            p.scen_int[e] = period;
            s_envflag[epb + le] = (uint32_t)period;
            update_formation<real>(c.scenario, key, 0, N, F);
            real box = c.spawn_box, xy[2];
            rng_uniform<real, 2>(key, QS_SITE_SCEN, 9, 0, 0, -box, box, xy);
            real z = rng_uniform1<real>(key, QS_SITE_SCEN, 10, 0, 0, (real)-0.5 * box, (real)0.5 * box) + (real)2, zlb = (real)0.25;
            const int f = F.f;
            if (f == 3 || f == 1 || f == 2) zlb = F.size + (real)0.25;
            else if (f == 5 || f == 6) { int rn = N < F.per_layer ? N : F.per_layer, d1, d2; grid_dim(rn, &d1, &d2);
                zlb = (real)d1 * F.size + (real)0.25; }
            z = M<real>::fmax(zlb, z);
            real c1[3] = {xy[0], xy[1], z}, c2[3];
            real dist = rng_uniform1<real>(key, QS_SITE_SCEN, 11, 0, 0, box / (real)4, box);
            real phi = rng_uniform1<real>(key, QS_SITE_SCEN, 12, 0, 0, (real)-QS_PI_D, (real)QS_PI_D);
            real theta = rng_uniform1<real>(key, QS_SITE_SCEN, 13, 0, 0, (real)(-0.5 * QS_PI_D), (real)(0.5 * QS_PI_D));
            real st, ct, sp, cph; M<real>::sincos(theta, &st, &ct); M<real>::sincos(phi, &sp, &cph);
            c2[0] = c1[0] + dist * (st * cph); c2[1] = c1[1] + dist * (st * sp); c2[2] = c1[2] + dist * ct;
            int s = f_suffix(f), ax = (s == 0) ? 2 : ((s == 1) ? 1 : ((s == 2) ? 0 : -1));
            if (ax >= 0) {
                real df = c2[ax] - c1[ax];
                if (M<real>::fabs(df) < F.lo) { real sg = (real)((df > 0) - (df < 0)); c2[ax] = sg * F.lo + c1[ax]; }
            }
            for (int q = 0; q < 3; ++q) { p.scen_real[q * E + e] = c1[q]; p.scen_real[(3 + q) * E + e] = c2[q]; }
            svs_create_formations<real>(key, F, N, QS_CUBE_FD(c), c1, c2, false, goals);
        }
        s_envflag[le] = have_spawn;
#ifdef QS_TAPE
        if (QS_ON_TAPE(key)) s_cur[le] = *key.cur;
#endif
    }
    if (TEAM) QS_WAVE_SYNC(); else __syncthreads();   // team kernels: only wave 0 is here

    // ---- per-drone part: QuadrotorSingle._reset quadrotor_single.py:387-447 ----
    auto per_drone = [&]() {
        real spawn[3];
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            goal[q] = s_goal[(le * L.goal_rows + i) * 3 + q];
            spawn[q] = s_envflag[le] ? s_spawn[q * B + tid] : goal[q];
        }
        real u[3];
        rng_uniform<real, 3>(key, QS_SITE_SPAWN, 0, i, 0, -c.spawn_box, c.spawn_box, u);
#pragma unroll
        for (int q = 0; q < 3; ++q) d.pos[q] = u[q] + spawn[q];
        if (d.pos[2] < (real)0.75) d.pos[2] = (real)0.75;
        real xy[3] = {-d.pos[0], -d.pos[1], 0}, n = norm3<real>(xy);
        if (n >= (real)0.00001) { xy[0] /= n; xy[1] /= n; }
        for (int t = 0; t < 256; ++t) {   // yaw rejection (:431-434)
            real th = rng_uniform1<real>(key, QS_SITE_SPAWN_YAW, t, i, 0, (real)-QS_PI_D, (real)QS_PI_D);
            yaw_rot<real>(th, d.rot);
            if (!(d.rot[0] * xy[0] + d.rot[3] * xy[1] < (real)0.5)) break;
        }
#pragma unroll
        for (int q = 0; q < 3; ++q) { s_vel[q * B + tid] = stale_vel[q]; s_pos[q * B + tid] = d.pos[q]; d.vel[q] = 0; d.omega[q] = 0; }
#pragma unroll
        for (int q = 0; q < 4; ++q) { d.rot_damp[q] = 0; d.cmds_damp[q] = 0; }
        d.flags = F_COL_AGENT_OK | F_COL_OBST_OK | (d.flags & F_SVD_MASK);   // since_last_svd persists (App. A)
        SensNoise<real> sn;
        if (c.sense_noise) sensor_noise_draw<real>(c, key, i, 0, sn);
        self_obs<real>(c, sn, d, goal, myobs);
    };
#ifdef QS_TAPE
    // the reference resets the drones one after the other (spawn draw, yaw rejection loop, sensor noise): take turns
    if (QS_ON_TAPE(key)) {
        for (int turn = 0; turn < N; ++turn) {
            if (do_reset && i == turn) { *key.cur = s_cur[le]; per_drone(); s_cur[le] = *key.cur; }
            __syncthreads();
        }
    } else
#endif
    if (do_reset) per_drone();
    if (TEAM) QS_WAVE_SYNC(); else __syncthreads();
    if (STREAM) {   // single-wave kernels: the rows of the re-initialised envs go out through the rows stage
        const uint64_t rowmask = __ballot(do_reset);
        const int first_env = bidx * epb;
        int nenv = E - first_env; nenv = nenv < epb ? nenv : epb;
        stream_rows<real>(c, L, p.obs + (size_t)first_env * N * c.obs_dim, (const real *)(smem + L.off_self),
            (real *)(smem + L.off_rows), N, i, le, base, tid,
                          s_pos, s_vel, s_metric, s_obst, d.pos, stale_vel, do_reset, nenv * N, rowmask,
                          c.dr_num_size > 0 ? (real)0.5 * p.dr_size[s_envflag[2 * epb + (le < epb ? le : 0)] & 7] : c.obst_radius);
    } else if (do_reset) {
        neighbor_obs<real>(c, N, i, base, B, tid, s_pos, s_vel, s_metric, d.pos, stale_vel, myobs + c.self_dim);
        if (c.use_obstacles)
            sdf_obs<real>(c, s_obst + (le * 2 + 0) * M_, s_obst + (le * 2 + 1) * M_, M_, d.pos[0], d.pos[1],
                myobs + c.self_dim + 6 * c.num_neighbors,
                          c.dr_num_size > 0 ? (real)0.5 * p.dr_size[s_envflag[2 * epb + le] & 7] : c.obst_radius);
    }
}

// ------------------------------------------------------------------------------------------------
// Kernel instantiation.  Generic library: 16 template kernels (single-wave / team) x (fast / full scenario set) x
// (step / rollout) x (float / double), every configuration value a run-time kernel argument.
// Config-specialised code object (qs_spec_kernels.hip, -DQS_SPEC_FILE="<header from qs_spec_header()>"): the three kernels
// one configuration needs, with every constant of the configuration, the LDS layout and the envs-per-workgroup count as
// literals - loops over drones / neighbours / obstacles unroll, LDS addresses fold into instruction offsets and the
// scalar bookkeeping of the generic kernels disappears (a lone wave per SIMD issues one instruction every ~4.5 cycles
// whatever its type, so instruction count is what the per-step latency is made of).  What stays a run-time argument:
// seed, env_id_offset, num_envs; the reward coefficients (annealed by the reward-shaping wrapper) are read from device memory.
// ------------------------------------------------------------------------------------------------
template <typename real, bool FULL>
__device__ __forceinline__ void qs_reset_impl(const Consts<real> &c, Ptrs<real> &p, const LdsLayout &L, int epb) {
    extern __shared__ __align__(16) unsigned char smem[];
    const Consts<real> *cp = &c;
    const int N = c.num_agents, E = c.num_envs, T = E * N;
    const int tid = threadIdx.x, le = tid / N, i = tid - le * N, e = blockIdx.x * epb + le;
    const bool in_range = (le < epb) && (e < E);
    const bool do_reset = in_range && p.reset_mask[e] != 0;
    const int g = in_range ? e * N + i : 0;
    // an explicit reset takes the env's next draw counter, like a step does: consecutive resets (or a reset right after the
    // auto-reset of a finished episode) start different episodes
    const uint32_t ctr = in_range ? p.step_ctr[e] + 1u : 0u;
#ifdef QS_TAPE
    int *s_cur = (int *)(smem + L.off_cur);
    int cursor = 0;
    const double *tape_env = p.tape ? p.tape + (size_t)(in_range ? e : 0) * (size_t)p.tape_len : nullptr;
    if (tape_env && in_range && i == 0) s_cur[le] = p.tape_pos[e];
    __syncthreads();
    RngKey key = {c.seed_lo, c.seed_hi, (uint32_t)(c.env_id_offset + (in_range ? e : 0)), ctr, tape_env, &cursor};
#else
    RngKey key = {c.seed_lo, c.seed_hi, (uint32_t)(c.env_id_offset + (in_range ? e : 0)), ctr};
#endif
    Drone<real> d;
    real goal[3] = {0, 0, 0}, stale_vel[3];
    const int es = in_range ? e : 0;   // (inactive lanes read drone 0's slot of a valid env, as before)
#pragma unroll
    for (int q = 0; q < 3; ++q) stale_vel[q] = QS_BLK_AT(real, p.blk, vel, q, es, in_range ? i : 0, N);
    d.flags = QS_BLK_AT(uint32_t, p.blk, flags, 0, es, in_range ? i : 0, N);
    if (FULL && in_range) scen_lds_load<real>(p, L, smem, E, e, le, i, N);
    if (FULL) __syncthreads();
    reset_body<real, FULL, false, true>(cp, &p, &L, smem, epb, key, do_reset, &d, goal, stale_vel);   // writes the obs rows itself
    if (FULL) { __syncthreads(); if (do_reset) scen_lds_store<real>(p, L, smem, E, e, le, i, N); }
    if (do_reset) {
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            QS_BLK_AT(real, p.blk, pos, q, e, i, N) = d.pos[q]; QS_BLK_AT(real, p.blk, vel, q, e, i, N) = 0;
            QS_BLK_AT(real, p.blk, omega, q, e, i, N) = 0; QS_BLK_AT(real, p.blk, goal, q, e, i, N) = goal[q];
        }
#pragma unroll
        for (int q = 0; q < 9; ++q) QS_BLK_AT(real, p.blk, rot, q, e, i, N) = d.rot[q];
#pragma unroll
        for (int q = 0; q < 4; ++q) { QS_BLK_AT(real, p.blk, rot_damp, q, e, i, N) = 0;
            QS_BLK_AT(real, p.blk, cmds_damp, q, e, i, N) = 0; QS_BLK_AT(real, p.blk, ring, q, e, i, N) = 0; }
#pragma unroll
        // the step kernels leave the sums alone until the episode's 5-s window opens (qs_step_sem.h)
        for (int q = 0; q < 3; ++q) QS_BLK_AT(real, p.blk, sums, q, e, i, N) = 0;
        QS_BLK_AT(uint32_t, p.blk, flags, 0, e, i, N) = d.flags;
        QS_BLK_AT(uint64_t, p.blk, pair, 0, e, i, N) = 0;
        p.new_pair_mask[g] = 0;
        p.obst_hit_idx[g] = -1;
        if (c.episode_sums) {
            for (int q = 0; q < QS_SUM_COUNT; ++q) p.run_sums[q * T + g] = 0;
        }
        if (i == 0) {
            for (int q = 0; q < QS_CNT_COUNT; ++q) p.counters[q * E + e] = 0;
            p.tick[e] = 0;
            p.unique_col[e] = 0; p.obst_new[e] = 0; p.room_new[e] = 0;
            p.reset_mask[e] = 0;
            p.step_ctr[e] = ctr;
#ifdef QS_TAPE
            if (tape_env) p.tape_pos[e] = s_cur[le];
#endif
        }
    }
}

#include "qs_step_sem.h"   // the step semantics shared by the two step bodies below

#ifdef QS_SPEC
#if QS_SPEC_PRECISION == 8
typedef double real;
#else
typedef float real;
#endif
__device__ __forceinline__ Consts<real> qs_spec_consts(const Consts<real> &rt) {
    Consts<real> c = __builtin_bit_cast(Consts<real>, qs_spec_cw);
    c.seed_lo = rt.seed_lo; c.seed_hi = rt.seed_hi; c.env_id_offset = rt.env_id_offset; c.num_envs = rt.num_envs;
    return c;
}
#define QS_KARGS const Consts<real> c_rt, Ptrs<real> p, const real *__restrict__ actions, LdsLayout L_rt
#define QS_EPB_ARG epb_rt
#define QS_SPEC_PROLOGUE const Consts<real> c = qs_spec_consts(c_rt); const LdsLayout L = __builtin_bit_cast(LdsLayout, qs_spec_lw); const int epb = QS_SPEC_EPB; (void)L_rt; (void)epb_rt;
#define QS_SCEN_FULL QS_SPEC_FULL
#if QS_SPEC_TEAM
#define QS_TW QS_SPEC_TEAM
#define QS_MULTI 0
#include "qs_step_team.inc"
#undef QS_MULTI
#define QS_MULTI 1
#include "qs_step_team.inc"
#undef QS_GATED
#define QS_GATED 1
#include "qs_step_team.inc"
#undef QS_GATED
#undef QS_MULTI
#else
#define QS_MULTI 0
#include "qs_step_kernel.inc"
#undef QS_MULTI
#define QS_MULTI 1
#include "qs_step_kernel.inc"
#undef QS_MULTI
#endif
extern "C" __global__ void __launch_bounds__(QS_WAVE) qs_spec_reset(const Consts<real> c_rt, Ptrs<real> p, LdsLayout L_rt, int epb_rt) {
    QS_SPEC_PROLOGUE
    qs_reset_impl<real, (QS_SPEC_FULL != 0)>(c, p, L, epb);
}
#undef QS_SCEN_FULL

#elif defined(QS_TAPE)   // ---- noise-tape flavour (qs_tape_kernels.hip): single-wave single-step kernels + reset, own names ----
#define QS_KARGS const Consts<real> c, Ptrs<real> p, const real *__restrict__ actions, LdsLayout L
#define QS_EPB_ARG epb
#define QS_SPEC_PROLOGUE
#define qs_step_kernel qs_tape_step_kernel
#define qs_step_kernel_full qs_tape_step_kernel_full
#define QS_MULTI 0
#define QS_SCEN_FULL 0
#include "qs_step_kernel.inc"
#undef QS_SCEN_FULL
#define QS_SCEN_FULL 1
#include "qs_step_kernel.inc"
#undef QS_SCEN_FULL
#undef QS_MULTI
#undef qs_step_kernel
#undef qs_step_kernel_full
template <typename real, bool FULL>
__global__ void __launch_bounds__(QS_WAVE) qs_tape_reset_kernel(const Consts<real> c, Ptrs<real> p, LdsLayout L, int epb) {
    qs_reset_impl<real, FULL>(c, p, L, epb);
}

#else   // ---- generic kernels ----
#define QS_KARGS const Consts<real> c, Ptrs<real> p, const real *__restrict__ actions, LdsLayout L
#define QS_EPB_ARG epb
#define QS_SPEC_PROLOGUE

#define QS_SCEN_FULL 0
#define QS_MULTI 0
#include "qs_step_kernel.inc"
#undef QS_MULTI
#define QS_MULTI 1
#include "qs_step_kernel.inc"
#undef QS_MULTI
#undef QS_SCEN_FULL
#define QS_SCEN_FULL 1      // all scenarios incl. `mix` (qs_scenarios.h); scenario state in LDS
#define QS_MULTI 0
#include "qs_step_kernel.inc"
#undef QS_MULTI
#define QS_MULTI 1
#include "qs_step_kernel.inc"
#undef QS_MULTI
#undef QS_SCEN_FULL
// the same four kernels as a team of 4 waves per workgroup (latency variant for small batches)
#define QS_TW QS_TEAM_WAVES
#define QS_SCEN_FULL 0
#define QS_MULTI 0
#include "qs_step_team.inc"
#undef QS_MULTI
#define QS_MULTI 1
#include "qs_step_team.inc"
#undef QS_GATED
#define QS_GATED 1
#include "qs_step_team.inc"
#undef QS_GATED
#undef QS_MULTI
#undef QS_SCEN_FULL
#define QS_SCEN_FULL 1
#define QS_MULTI 0
#include "qs_step_team.inc"
#undef QS_MULTI
#define QS_MULTI 1
#include "qs_step_team.inc"
#undef QS_GATED
#define QS_GATED 1
#include "qs_step_team.inc"
#undef QS_GATED
#undef QS_MULTI
#undef QS_SCEN_FULL

// reset kernel (qs_reset): resets the envs flagged in reset_mask
template <typename real, bool FULL>
__global__ void __launch_bounds__(QS_WAVE) qs_reset_kernel(const Consts<real> c, Ptrs<real> p, LdsLayout L, int epb) {
    qs_reset_impl<real, FULL>(c, p, L, epb);
}
#endif

#if !defined(QS_TAPE) && !defined(QS_SPEC)
// ------------------------------------------------------------------------------------------------
// Batched experience replay (include/quadswarm.h: qs_replay_enable): one wave per environment, launched after every step.
// Lane 0 runs the wrapper's decision logic on the env's scalars; the whole wave then moves the snapshot it decided on.
// A snapshot = every per-drone / per-env array of one environment (the list of qs_snapshot_*), packed array after array.
// ------------------------------------------------------------------------------------------------
#define QS_REPLAY_MAX_ARR 40
// strides / counts in elements, off in bytes; group > 0: wave-blocked array (envs per block, bytes between blocks)
struct ReplayArr { char *base; uint32_t elem, comps, per_env, off; uint64_t comp_stride; uint32_t group, group_stride; };
struct ReplayParams {
    ReplayArr arr[QS_REPLAY_MAX_ARR];
    // obs_arr / tick_arr: index of that array in arr[]
    int32_t narr, N, E, use_obstacles, cp_every, grace_ticks, min_gap, obs_arr, tick_arr, ep_len;
    uint32_t snap_bytes, seed_lo, seed_hi;
    int32_t env_id_offset;
    float sample_prob;
    char *pool;                                  // [E][QS_REPLAY_RING + QS_REPLAY_EVENTS][snap_bytes]
    const uint8_t *done; const int32_t *tick; const uint32_t *step_ctr; const uint64_t *unique_col, *obst_new;
    int32_t *counters;                           // [QS_CNT_COUNT][E]
    const void *ep_sums; int32_t real_size, T;   // per-episode crash reward of drone 0: ep_sums[QS_RI_REW_CRASH][e*N]
    const void *run_sums;                        // the running episode's sums (an explicit reset records the crash reward so far)
    // per-env wrapper state
    uint8_t *active, *saved, *ep_saved;          // ep_saved: was the episode that just ended a replayed one (quadrotor_multi.py:629-633)
    float *crash_hist; int32_t *crash_n, *crash_pos;    // deque(maxlen=100) of crashes_last_episode
    int32_t *ck_count, *ck_head, *last_added;    // checkpoint ring: entries, slot of the oldest; tick of the last filed event
    int32_t *ev_len, *ev_idx, *ev_replayed;      // event buffer: entries, buffer_idx, num_replayed [QS_REPLAY_EVENTS][E]
    int32_t *ev_slot;                            // [QS_REPLAY_EVENTS][E]: pool slot of the k-th event (the deque order of the reference)
    int32_t *episodes, *replayed, *errors;
    // tick the running episode started at (a replayed one: its checkpoint's); control steps of the last finished one
    int32_t *start_tick, *last_steps;
};

__device__ __forceinline__ void replay_copy(const ReplayParams &P, int e, char *snap, bool save, int lane) {
    for (int a = 0; a < P.narr; ++a) {
        const ReplayArr A = P.arr[a];
        const int total = (int)(A.comps * A.per_env);
        char *sp = snap + A.off;
        for (int idx = lane; idx < total; idx += QS_WAVE) {
            const int cpt = idx / (int)A.per_env, k = idx - cpt * (int)A.per_env;
            const int gb = A.group ? e / (int)A.group : 0, ge = A.group ? e - gb * (int)A.group : e;
            char *live = A.base + (size_t)gb * A.group_stride + ((size_t)cpt * A.comp_stride + (size_t)ge * A.per_env + k) * A.elem;
            char *sn = sp + (size_t)idx * A.elem;
            if (A.elem == 8) { if (save) *(uint64_t *)sn = *(const uint64_t *)live; else *(uint64_t *)live = *(const uint64_t *)sn; }
            else if (A.elem == 4) { if (save) *(uint32_t *)sn = *(const uint32_t *)live; else *(uint32_t *)live = *(const uint32_t *)sn; }
            else { if (save) *sn = *live; else *live = *sn; }
        }
    }
}

// QuadrotorEnvMulti.reset's replay-buffer lines (quadrotor_multi.py:356-359, can_drones_fly :284-287): until the buffer is active, every
// reset files crashes_last_episode in a 100-entry history and activates the buffer once >= 10 episodes average above -1
__device__ __forceinline__ void replay_record_reset(const ReplayParams &P, int e, float crashes) {
    if (P.active[e]) return;
    const int E = P.E;
    int n = P.crash_n[e], pos = P.crash_pos[e];
    P.crash_hist[(size_t)pos * E + e] = crashes;
    pos = (pos + 1) % 100; n = n < 100 ? n + 1 : 100;
    P.crash_n[e] = n; P.crash_pos[e] = pos;
    float sum = 0;
    for (int q = 0; q < n; ++q) sum += P.crash_hist[(size_t)q * E + e];
    if (fabsf(sum / (float)n) < 1.0f && n >= 10) P.active[e] = 1;
}

// An EXPLICIT reset (qs_reset) of environments whose replay wrapper is running - what ExperienceReplayWrapper.reset() ->
// QuadrotorEnvMulti.reset() does to the wrapper's bookkeeping (quad_experience_replay.py:106-118): the crash reward the running episode
// has collected so far goes into the history (the reference appends crashes_last_episode on every reset), and the new episode starts
// at tick 0 (start_tick feeds the per-episode step count of the action statistics).  Like the reference, the checkpoint deque and the
// tick of the last filed event are NOT cleared by a reset - only by an episode end (new_episode, :167-209).  Launched by qs_reset
// before the reset kernel, one lane per environment.
__global__ void __launch_bounds__(QS_WAVE) qs_replay_reset_kernel(const ReplayParams P, const uint8_t *reset_mask) {
    const int e = blockIdx.x * QS_WAVE + threadIdx.x;
    if (e >= P.E || !reset_mask[e]) return;
    const size_t at = (size_t)QS_RI_REW_CRASH * P.T + (size_t)e * P.N;
    replay_record_reset(P, e, P.real_size == 8 ? (float)((const double *)P.run_sums)[at] : ((const float *)P.run_sums)[at]);
    P.start_tick[e] = 0;
}

__global__ void __launch_bounds__(QS_WAVE) qs_replay_kernel(const ReplayParams P) {
    __shared__ int s_act[4];   // action, source slot, destination slot, flags
    const int e = blockIdx.x, lane = threadIdx.x, E = P.E, SLOTS = QS_REPLAY_RING + QS_REPLAY_EVENTS;
    enum { ACT_NONE = 0, ACT_SAVE = 1, ACT_FILE = 2, ACT_RESTORE = 3 };
    if (lane == 0) {
        int act = ACT_NONE, src = 0, dst = 0;
        const bool done = P.done[(size_t)e * P.N] != 0;
        const int tick = P.tick[e];
        RngKey key = {P.seed_lo, P.seed_hi, (uint32_t)(P.env_id_offset + e), P.step_ctr[e]};
        auto record_reset = [&](float crashes) { replay_record_reset(P, e, crashes); };
        if (done) {   // ExperienceReplayWrapper.new_episode, quad_experience_replay.py:167-209
            const float crashes = P.real_size == 8 ? (float)((const double *)P.ep_sums)[(size_t)QS_RI_REW_CRASH * P.T + (size_t)e * P.N]
                                                   : ((const float *)P.ep_sums)[(size_t)QS_RI_REW_CRASH * P.T + (size_t)e * P.N];
            record_reset(crashes);                 // the auto-reset inside QuadrotorEnvMulti.step
            P.ep_saved[e] = P.saved[e];
            P.last_steps[e] = P.ep_len + 1 - P.start_tick[e];
            P.start_tick[e] = 0;
            P.episodes[e] += 1;
            P.last_added[e] = -1000000000;
            P.ck_count[e] = 0; P.ck_head[e] = 0;
            const float u = rng_uniform1<float>(key, QS_SITE_REPLAY, 0, 0, 0, 0.f, 1.f);
            int len = P.ev_len[e];
            if (u < P.sample_prob && P.active[e] && len > 0) {
                P.replayed[e] += 1;
                int idx = (int)(rng_uniform1<float>(key, QS_SITE_REPLAY, 1, 0, 0, 0.f, 1.f) * (float)len);   // random.randint(0, len - 1)
                idx = idx >= len ? len - 1 : idx;
                P.ev_replayed[(size_t)idx * E + e] += 1;
                act = ACT_RESTORE; src = P.ev_slot[(size_t)idx * E + e];
                P.start_tick[e] = *(const int32_t *)(P.pool + ((size_t)e * SLOTS + src) * P.snap_bytes + P.arr[P.tick_arr].off);
                // ReplayBuffer.cleanup (:50-56): drop the events replayed 10 times, order kept, buffer_idx untouched
                int w = 0;
                for (int q = 0; q < len; ++q) {
                    const int nr = P.ev_replayed[(size_t)q * E + e], sl = P.ev_slot[(size_t)q * E + e];
                    if (nr < 10) { P.ev_replayed[(size_t)w * E + e] = nr; P.ev_slot[(size_t)w * E + e] = sl; ++w; }
                }
                P.ev_len[e] = w;
                P.saved[e] = 1;                    // the event's copy of the env was marked saved_in_replay_buffer (:28)
            } else {
                record_reset(0.f);                 // the wrapper's own env.reset() (:203): crashes_last_episode is 0 again by then
                P.saved[e] = 0;
            }
        } else if (P.active[e] && !P.saved[e]) {
            int cnt = P.ck_count[e], head = P.ck_head[e];
            if (tick % P.cp_every == 0) {          // save_checkpoint (:141-144): deque(maxlen = QS_REPLAY_RING)
                if (cnt < QS_REPLAY_RING) { dst = (head + cnt) % QS_REPLAY_RING; ++cnt; }
                else { dst = head; head = (head + 1) % QS_REPLAY_RING; }
                P.ck_count[e] = cnt; P.ck_head[e] = head;
                act = ACT_SAVE;
            }
            const bool collision = (P.unique_col[e] & ~1ull) != 0 || (P.use_obstacles && P.obst_new[e] != 0);   // `.any()` on the id array
            if (collision && tick > P.grace_ticks && tick - P.last_added[e] > P.min_gap) {
                if (cnt < 3) P.errors[e] += 1;     // the reference raises IndexError here
                else {
                    // checkpoints[-3]; if this very step also saved one, the wave saves first and files afterwards
                    src = (head + cnt - 3) % QS_REPLAY_RING;
                    int len = P.ev_len[e], bi = P.ev_idx[e], k;
                    if (len < QS_REPLAY_EVENTS) {  // append: take a pool slot no listed event uses
                        uint32_t used = 0;
                        for (int q = 0; q < len; ++q) used |= 1u << (P.ev_slot[(size_t)q * E + e] - QS_REPLAY_RING);
                        int fs = 0; while (used >> fs & 1) ++fs;
                        k = len; P.ev_slot[(size_t)k * E + e] = QS_REPLAY_RING + fs; P.ev_len[e] = len + 1;
                    } else k = bi;                 // overwrite buffer[buffer_idx]
                    P.ev_replayed[(size_t)k * E + e] = 0;
                    P.ev_idx[e] = (bi + 1) % QS_REPLAY_EVENTS;
                    P.last_added[e] = tick;
                    dst = (act == ACT_SAVE ? dst : 0) | (P.ev_slot[(size_t)k * E + e] << 8);
                    act = act == ACT_SAVE ? (ACT_SAVE | (ACT_FILE << 4)) : ACT_FILE;
                    s_act[3] = src;
                }
            }
        }
        s_act[0] = act; s_act[1] = src; s_act[2] = dst;
    }
    __syncthreads();
    const int act = s_act[0];
    if (act == ACT_NONE) return;
    char *pool = P.pool + (size_t)e * SLOTS * P.snap_bytes;
    if ((act & 15) == ACT_SAVE) { replay_copy(P, e, pool + (size_t)(s_act[2] & 255) * P.snap_bytes, true, lane); __threadfence_block();
        __syncthreads(); }
    if ((act & 15) == ACT_FILE || (act >> 4) == ACT_FILE) {
        const char *src = pool + (size_t)((act & 15) == ACT_FILE ? s_act[1] : s_act[3]) * P.snap_bytes;
        char *dst = pool + (size_t)(s_act[2] >> 8) * P.snap_bytes;
        for (uint32_t b = lane * 4; b < P.snap_bytes; b += QS_WAVE * 4) *(uint32_t *)(dst + b) = *(const uint32_t *)(src + b);
        // the reference returns the filed checkpoint's observation on this step (quad_experience_replay.py:151 rebinds `obs`)
        const ReplayArr A = P.arr[P.obs_arr];
        for (int idx = lane; idx < (int)A.per_env; idx += QS_WAVE)
            *(uint32_t *)(A.base + ((size_t)e * A.per_env + idx) * A.elem) = *(const uint32_t *)(src + A.off + (size_t)idx * A.elem);
        if (A.elem == 8)
            for (int idx = lane; idx < (int)A.per_env; idx += QS_WAVE)
                *(uint32_t *)(A.base + ((size_t)e * A.per_env + idx) * A.elem + 4) = *(const uint32_t *)(src + A.off + (size_t)idx * A.elem + 4);
    }
    if (act == ACT_RESTORE) {
        replay_copy(P, e, pool + (size_t)s_act[1] * P.snap_bytes, false, lane);
        if (lane == 0) {   // counters the reference zeroes on the replayed env (:180-182)
            P.counters[(size_t)QS_CNT_COLLISIONS * P.E + e] = 0; P.counters[(size_t)QS_CNT_COLLISIONS_AFTER_SETTLE * P.E + e] = 0;
            P.counters[(size_t)QS_CNT_OBST * P.E + e] = 0; P.counters[(size_t)QS_CNT_OBST_AFTER_SETTLE * P.E + e] = 0;
        }
    }
}

// state get/set for one env (qs_get_state / qs_set_state)
template <typename real>
__global__ void qs_state_kernel(Ptrs<real> p, int E, int N, int env, double *buf, int32_t *tick_io, int set) {
    const int i = threadIdx.x;
    if (i >= N) return;
    double *s = buf + (size_t)i * QS_STATE_STRIDE;
    const StateBlk &B = p.blk;
#define QS_S(arr, q) QS_BLK_AT(real, B, arr, q, env, i, N)
    if (!set) {
        for (int q = 0; q < 3; ++q) { s[q] = QS_S(pos, q); s[3 + q] = QS_S(vel, q); s[15 + q] = QS_S(omega, q); s[32 + q] = QS_S(goal, q); }
        for (int q = 0; q < 9; ++q) s[6 + q] = QS_S(rot, q);
        for (int q = 0; q < 4; ++q) { s[18 + q] = QS_S(rot_damp, q); s[22 + q] = QS_S(cmds_damp, q); s[26 + q] = QS_S(ou, q); }
        uint32_t f = QS_BLK_AT(uint32_t, B, flags, 0, env, i, N);
        s[30] = (f & F_ON_FLOOR) ? 1.0 : 0.0;
        s[31] = (double)((f & F_SVD_MASK) >> F_SVD_SHIFT);
        if (i == 0) *tick_io = p.tick[env];
    } else {
        for (int q = 0; q < 3; ++q) { QS_S(pos, q) = (real)s[q]; QS_S(vel, q) = (real)s[3 + q]; QS_S(omega, q) = (real)s[15 + q];
            QS_S(goal, q) = (real)s[32 + q]; }
        for (int q = 0; q < 9; ++q) QS_S(rot, q) = (real)s[6 + q];
        for (int q = 0; q < 4; ++q) { QS_S(rot_damp, q) = (real)s[18 + q]; QS_S(cmds_damp, q) = (real)s[22 + q];
            QS_S(ou, q) = (real)s[26 + q]; }
        uint32_t f = QS_BLK_AT(uint32_t, B, flags, 0, env, i, N) & ~(F_ON_FLOOR | F_SVD_MASK);
        if (s[30] != 0.0) f |= F_ON_FLOOR;
        f |= ((uint32_t)s[31] & 0xffu) << F_SVD_SHIFT;
        QS_BLK_AT(uint32_t, B, flags, 0, env, i, N) = f;
#undef QS_S
        if (i == 0 && *tick_io >= 0) p.tick[env] = *tick_io;
    }
}
#endif
