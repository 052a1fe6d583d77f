// qs_device.h - device-side building blocks of the QuadSwarm stepper (gfx950 / CDNA4, wave64).
//
// Everything here is per-lane code: one lane owns one drone.  Cross-drone data of one environment is
// exchanged through LDS by the kernels in quadswarm_hip.hip.  Templated on `real` (float = production,
// double = parity instantiation).  Reference citations are relative to gym_art/quadrotor_multi/.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/quadswarm.h"

namespace qs {

enum : uint32_t {
    F_ON_FLOOR = 1u << 0, F_CRASH_FLOOR = 1u << 1, F_CRASH_WALL = 1u << 2, F_CRASH_CEIL = 1u << 3,
    F_PREV_WALL = 1u << 4, F_PREV_CEIL = 1u << 5, F_PREV_ROOM = 1u << 6, F_PREV_OBST = 1u << 7,
    F_REACHED = 1u << 8, F_COL_AGENT_OK = 1u << 9, F_COL_OBST_OK = 1u << 10,
    F_IN_COL = 1u << 11,   // this drone's id was in a colliding pair last step (prev_ids of quadrotor_multi.py:440)
    // bookkeeping rows that are only touched when they matter (DESIGN.md 4a "bytes"): the state of those rows in HBM is described by a flag
    // dist_ring of this drone holds maintained values; clear = every entry counts as "far" (see goal_distance_log)
    F_RING_LIVE = 1u << 12,
    // the new_pair_mask word last stored for this drone was non-zero (clear = the word in HBM is 0: no store needed for a 0)
    F_NEWPAIR_NZ = 1u << 13,
    F_SVD_SHIFT = 16, F_SVD_MASK = 0xffu << 16
};

// per-lane event bits exchanged through LDS inside the step kernel
enum : uint32_t { B_OBST_HIT = 1, B_OBST_NEW = 2, B_FLOOR = 4, B_WALL_NEW = 8, B_CEIL_NEW = 16, B_ROOM_NEW = 32, B_DOWNWASH = 64 };

#define QS_PI_D 3.141592653589793

// ------------------------------------------------------------------------------------------------
// math dispatch
// ------------------------------------------------------------------------------------------------
template <typename real> struct M;
template <> struct M<float> {
    // v_sqrt_f32: 1 ulp, one instruction (the correctly rounded sqrtf expands to ~12 with its denormal scaling); the fp32
    // path is specified to 1e-5 and takes ~40 square roots per drone-step (norms, distances)
    static __device__ __forceinline__ float sqrt(float x) { return __builtin_amdgcn_sqrtf(x); }
    static __device__ __forceinline__ float sin(float x) { return sinf(x); }
    static __device__ __forceinline__ float cos(float x) { return cosf(x); }
    static __device__ __forceinline__ void sincos(float x, float *s, float *c) { sincosf(x, s, c); }
    static __device__ __forceinline__ float atan2(float y, float x) { return atan2f(y, x); }
    static __device__ __forceinline__ float log(float x) { return logf(x); }
    // ln(u) for u in (0,1) via v_log_f32 (abs err ~1e-7 * |ln u|: noise amplitudes are 1e-2..1e-4, tolerance 1e-5)
    static __device__ __forceinline__ float log_u01(float u) { return __builtin_amdgcn_logf(u) * 0.69314718055994531f; }
    static __device__ __forceinline__ float rcp(float x) { return __builtin_amdgcn_rcpf(x); }      // 1 ulp; a/b costs ~50 cycles
    static __device__ __forceinline__ float rsqrt(float x) { return __builtin_amdgcn_rsqf(x); }
    // sin(x), 1-cos(x) for |x| <= ~0.4 (the Rodrigues angle is |omega| dt <= 40*sqrt(3)*0.005 = 0.35): Taylor series;
    // computing 1-cos directly avoids the cancellation of 1 - cosf(x) in fp32
    static __device__ __forceinline__ void sin_omcos_small(float x, float *s, float *omc) {
        float x2 = x * x;
        *s = x * (1.0f + x2 * (-1.0f / 6 + x2 * (1.0f / 120 + x2 * (-1.0f / 5040 + x2 * (1.0f / 362880)))));
        *omc = x2 * (0.5f + x2 * (-1.0f / 24 + x2 * (1.0f / 720 + x2 * (-1.0f / 40320 + x2 * (1.0f / 3628800)))));
    }
    static __device__ __forceinline__ float pow(float x, float y) { return powf(x, y); }
    static __device__ __forceinline__ float fabs(float x) { return fabsf(x); }
    static __device__ __forceinline__ float fmax(float a, float b) { return fmaxf(a, b); }
    static __device__ __forceinline__ float fmin(float a, float b) { return fminf(a, b); }
    static __device__ __forceinline__ float floor(float x) { return floorf(x); }
    // sin/cos of 2*pi*u for u in (0,1): v_sin_f32 / v_cos_f32 take their argument in revolutions
    static __device__ __forceinline__ void sincos2pi(float u, float *s, float *c) {
        *s = __builtin_amdgcn_sinf(u); *c = __builtin_amdgcn_cosf(u);
    }
};
template <> struct M<double> {
    static __device__ __forceinline__ double sqrt(double x) { return ::sqrt(x); }
    static __device__ __forceinline__ double sin(double x) { return ::sin(x); }
    static __device__ __forceinline__ double cos(double x) { return ::cos(x); }
    static __device__ __forceinline__ void sincos(double x, double *s, double *c) { *s = ::sin(x); *c = ::cos(x); }
    static __device__ __forceinline__ double atan2(double y, double x) { return ::atan2(y, x); }
    static __device__ __forceinline__ double log(double x) { return ::log(x); }
    static __device__ __forceinline__ double log_u01(double u) { return ::log(u); }
    static __device__ __forceinline__ double rcp(double x) { return 1.0 / x; }
    static __device__ __forceinline__ double rsqrt(double x) { return 1.0 / ::sqrt(x); }
    static __device__ __forceinline__ void sin_omcos_small(double x, double *s, double *omc) { *s = ::sin(x); *omc = 1.0 - ::cos(x); }
    static __device__ __forceinline__ double pow(double x, double y) { return ::pow(x, y); }
    static __device__ __forceinline__ double fabs(double x) { return ::fabs(x); }
    static __device__ __forceinline__ double fmax(double a, double b) { return ::fmax(a, b); }
    static __device__ __forceinline__ double fmin(double a, double b) { return ::fmin(a, b); }
    static __device__ __forceinline__ double floor(double x) { return ::floor(x); }
    static __device__ __forceinline__ void sincos2pi(double u, double *s, double *c) {
        double th = 2.0 * QS_PI_D * u; *s = ::sin(th); *c = ::cos(th);
    }
};

// NaN / overflow test of the reward (quadrotor_single.py:87-90) on the bit pattern: fp32 config-specialised objects are
// built with -ffast-math, under which `x != x` may be folded away
__device__ __forceinline__ bool not_finite(float x) {
    uint32_t b = __float_as_uint(x);
    asm volatile("" : "+v"(b));   // opaque: otherwise the no-NaN assumption travels through the bitcast and folds the test to false
    return (b & 0x7fffffffu) > 0x7f61b1e6u;   // |x| > 3.0e38 (0x7f61b1e6), which includes Inf and every NaN pattern
}
__device__ __forceinline__ bool not_finite(double x) {
    return ((unsigned long long)__double_as_longlong(x) & 0x7ff0000000000000ull) == 0x7ff0000000000000ull || fabs(x) > 3.0e38;
}
__device__ __forceinline__ bool bits_differ(float a, float b) { return __builtin_bit_cast(uint32_t, a) != __builtin_bit_cast(uint32_t, b); }
__device__ __forceinline__ bool bits_differ(double a,
    double b) { return __builtin_bit_cast(uint64_t, a) != __builtin_bit_cast(uint64_t, b); }
template <typename real> __device__ __forceinline__ real clipr(real x, real lo, real hi) { return x < lo ? lo : (x > hi ? hi : x); }
// fp32: one v_med3_f32 instead of two compare + select pairs (identical for every non-NaN x when lo <= hi)
template <> __device__ __forceinline__ float clipr<float>(float x, float lo, float hi) { return __builtin_amdgcn_fmed3f(x, lo, hi); }
template <typename real> __device__ __forceinline__ real norm3(const real v[3]) { return M<real>::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]); }
template <typename real> __device__ __forceinline__ real dot3(const real a[3],
    const real b[3]) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }

// ------------------------------------------------------------------------------------------------
// Philox4x32-10, the stream specified in include/quadswarm.h
// ------------------------------------------------------------------------------------------------
// QS_TAPE builds (qs_tape_kernels.hip: the test-only "noise tape" flavour behind qs_set_noise_tape): a key also carries the
// environment's sequential tape of reference draws (values in final units, in the reference's call order - SURVEY App. B) and
// a lane-local cursor.  Every draw primitive below then pops the tape instead of running Philox; the kernels serialise
// the lanes of an environment wherever the reference's draw order is data-dependent.
struct RngKey { uint32_t k0, k1, env, step;
#ifdef QS_TAPE
                const double *tape; int *cur;
#endif
};
#ifdef QS_TAPE
#define QS_ON_TAPE(k) ((k).tape != nullptr)
__device__ __forceinline__ double tape_pop(const RngKey &k) { return k.tape[(*k.cur)++]; }
__device__ __forceinline__ void tape_skip(const RngKey &k, int n) { *k.cur += n; }
#else
#define QS_ON_TAPE(k) false
__device__ __forceinline__ double tape_pop(const RngKey &) { return 0.0; }
__device__ __forceinline__ void tape_skip(const RngKey &, int) {}
#endif

__device__ __forceinline__ void philox4x32(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, uint32_t w[4]) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        // one 32x32->64 multiply (v_mad_u64_u32) yields both halves of each product
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        c1 = (uint32_t)p1; c3 = (uint32_t)p0; c0 = n0; c2 = n2;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    w[0] = c0; w[1] = c1; w[2] = c2; w[3] = c3;
}
__device__ __forceinline__ void rng_words(const RngKey &k, int site, int slot, int i, int j, uint32_t w[4]) {
    philox4x32(k.env, k.step, (uint32_t)site | ((uint32_t)slot << 8), (uint32_t)i | ((uint32_t)j << 16), k.k0, k.k1, w);
}
// Four Philox blocks advanced in lockstep: the 10-round chains are independent, so issuing them round by round gives
// the in-order wave 8 independent multiplies per round instead of 2 (a lone wave issues dependent VALU ops ~2x slower).
__device__ __forceinline__ void philox4x32_x4(const uint32_t c0_in[4], uint32_t c1, const uint32_t c2_in[4], const uint32_t c3_in[4],
                                              uint32_t k0, uint32_t k1, uint32_t w[4][4]) {
    uint32_t c0[4], c1v[4], c2[4], c3[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) { c0[s] = c0_in[s]; c1v[s] = c1; c2[s] = c2_in[s]; c3[s] = c3_in[s]; }
#pragma unroll
    for (int r = 0; r < 10; ++r) {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const uint64_t p0 = (uint64_t)0xD2511F53u * c0[s], p1 = (uint64_t)0xCD9E8D57u * c2[s];
            const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1v[s] ^ k0, n2 = (uint32_t)(p0 >> 32) ^ c3[s] ^ k1;
            c1v[s] = (uint32_t)p1; c3[s] = (uint32_t)p0; c0[s] = n0; c2[s] = n2;
        }
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
#pragma unroll
    for (int s = 0; s < 4; ++s) { w[s][0] = c0[s]; w[s][1] = c1v[s]; w[s][2] = c2[s]; w[s][3] = c3[s]; }
}

template <typename real> __device__ __forceinline__ real u01(uint32_t x) { return ((real)(x >> 9) + (real)0.5) * (real)(1.0 / 8388608.0); }

// n <= 4 standard normals (Box-Muller on word pairs (0,1),(2,3)); on a tape: the next NN recorded draws (the tape holds the
// reference's values in their final units, so tape-aware callers use rng_normal_s with the scale the reference passed)
template <typename real, int NN> __device__ __forceinline__ void rng_normal(const RngKey &k, int site, int slot, int i, int j, real z[NN]) {
    if (QS_ON_TAPE(k)) {
#pragma unroll
        for (int q = 0; q < NN; ++q) z[q] = (real)tape_pop(k);
        return;
    }
    uint32_t w[4];
    rng_words(k, site, slot, i, j, w);
    real r0 = M<real>::sqrt((real)-2.0 * M<real>::log_u01(u01<real>(w[0]))), s0, c0;
    M<real>::sincos2pi(u01<real>(w[1]), &s0, &c0);
    z[0] = r0 * c0;
    if (NN > 1) z[1] = r0 * s0;
    if (NN > 2) {
        real r1 = M<real>::sqrt((real)-2.0 * M<real>::log_u01(u01<real>(w[2]))), s1, c1;
        M<real>::sincos2pi(u01<real>(w[3]), &s1, &c1);
        z[2] = r1 * c1;
        if (NN > 3) z[3] = r1 * s1;
    }
}
// NN <= 4 draws of normal(0, scale): scale * z from the stream, the recorded value from a tape
template <typename real,
    int NN> __device__ __forceinline__ void rng_normal_s(const RngKey &k, int site, int slot, int i, int j, real scale, real out[NN]) {
    rng_normal<real, NN>(k, site, slot, i, j, out);
    if (!QS_ON_TAPE(k)) {
#pragma unroll
        for (int q = 0; q < NN; ++q) out[q] = scale * out[q];
    }
}
template <typename real> __device__ __forceinline__ void box_muller4(const uint32_t w[4], real z[4]) {
    real r0 = M<real>::sqrt((real)-2.0 * M<real>::log_u01(u01<real>(w[0]))), s0, c0;
    M<real>::sincos2pi(u01<real>(w[1]), &s0, &c0);
    real r1 = M<real>::sqrt((real)-2.0 * M<real>::log_u01(u01<real>(w[2]))), s1, c1;
    M<real>::sincos2pi(u01<real>(w[3]), &s1, &c1);
    z[0] = r0 * c0; z[1] = r0 * s0; z[2] = r1 * c1; z[3] = r1 * s1;
}

template <typename real,
    int NN> __device__ __forceinline__ void rng_uniform(const RngKey &k, int site, int slot, int i, int j, real lo, real hi, real u[NN]) {
    if (QS_ON_TAPE(k)) {
#pragma unroll
        for (int q = 0; q < NN; ++q) u[q] = (real)tape_pop(k);
        return;
    }
    uint32_t w[4];
    rng_words(k, site, slot, i, j, w);
#pragma unroll
    for (int q = 0; q < NN; ++q) u[q] = lo + (hi - lo) * u01<real>(w[q]);
}
template <typename real> __device__ __forceinline__ real rng_uniform1(const RngKey &k, int site, int slot, int i, int j, real lo, real hi) {
    real u[1]; rng_uniform<real, 1>(k, site, slot, i, j, lo, hi, u); return u[0];
}

// ------------------------------------------------------------------------------------------------
// kernel constants (converted once on the host from qs_config)
// ------------------------------------------------------------------------------------------------
template <typename real> struct Consts {
    real inertia[3], inv_inertia[3], arm, mass, inv_mass;
    real prop_cross[4][3], prop_ccw[4], thrust_max[4], torque_max[4];
    real motor_tau_up, motor_tau_down, motor_linearity, vel_damp, damp_omega_quadratic, omega_max;
    real thrust_noise_sigma, ou_theta;
    real dt, control_dt;
    real room_lo[3], room_hi[3];
    real floor_threshold;
    real pos_norm_std, pos_unif_range, vel_norm_std, vel_unif_range, quat_norm_std, quat_unif_range, gyro_noise_density;
    real collision_threshold, collision_falloff_threshold;
    real rew_coeff[QS_REW_COUNT];
    real spawn_box, approach_goal_metric;
    real nbr_clip_pos[3], nbr_clip_vel[3];
    real obst_radius, obst_hit_threshold, obst_size, room_mid_z;
    int32_t sim_steps, ep_len, floor_mode, svd_period, sense_noise, obs_repr, self_dim, obs_dim;
    int32_t num_neighbors, use_downwash, use_obstacles, scenario, num_obstacles, obst_area[2];
    int32_t grace_steps, final_steps, control_freq;
    int32_t cube_fd_all;  // int(N ** (1/3)) for the whole swarm (all scenarios except swarm_vs_swarm)
    int32_t cube_fd[2];   // int(n ** (1/3)) for the two half-swarms, evaluated on the host with libm's pow (scenarios/base.py:98-99)
    uint32_t seed_lo, seed_hi;
    int32_t env_id_offset, num_envs, num_agents;
    real inv_dt, prox_ratio;   // 1/dt; -quadcol_smooth_max / collision_falloff_threshold (collisions/quadrotors.py:97)
    real inv_win[3];           // 1/min(ep_len+1, {1,3,5} s of control steps): episode-stat window sizes
    int32_t write_rew_info;   // 0: skip the 17-term reward-info matrix (it is logging, not part of obs/reward/done)
    int32_t episode_sums;     // 1: per-episode sums of the reward terms and action moments (reward_shaping.py:78-110)
    // --quads_domain_random: per-episode obstacle density / size (include/quadswarm.h); dr_on = 0 folds all of it away
    // (the choice tables live in device memory, Ptrs::dr_*: a run-time index into this struct would move the whole constant block
    // to scratch memory - and every literal of a config-specialised kernel with it: measured 8 -> 21 us per C2 step)
    int32_t dr_on, dr_num_density, dr_num_size;
    real arm_r;
};

// c.cube_fd by value: a pointer INTO the constant block handed to a function that is not inlined would pin the whole block in memory
// (and every literal of a config-specialised kernel with it); the two ints travel in a temporary instead
struct CubeFd { int v[2]; __device__ __forceinline__ operator const int *() const { return v; } };
#define QS_CUBE_FD(c) (CubeFd{{(c).cube_fd[0], (c).cube_fd[1]}})

// per-drone dynamic state held in registers
template <typename real> struct Drone {
    real pos[3], vel[3], rot[9], omega[3];
    real rot_damp[4], cmds_damp[4], ou[4];
    uint32_t flags;
};

template <typename real> __device__ __forceinline__ void matmul3(const real a[9], const real b[9], real o[9]) {
    real t[9];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) t[r * 3 + c] = a[r * 3] * b[c] + a[r * 3 + 1] * b[3 + c] + a[r * 3 + 2] * b[6 + c];
#pragma unroll
    for (int q = 0; q < 9; ++q) o[q] = t[q];
}
template <typename real> __device__ __forceinline__ void yaw_rot(real theta, real r[9]) {
    real s, c; M<real>::sincos(theta, &s, &c);
    r[0] = c; r[1] = -s; r[2] = 0; r[3] = s; r[4] = c; r[5] = 0; r[6] = 0; r[7] = 0; r[8] = 1;
}

// yaw_rot(atan2(y, x)) without the transcendental round trip: cos(atan2(y,x)) = x/hypot(x,y), sin = y/hypot(x,y);
// atan2(0,0) = 0 -> identity yaw.
template <typename real> __device__ __forceinline__ void yaw_rot_xy(real x, real y, real r[9]) {
    real h2 = x * x + y * y, c = 1, s = 0;
    if (h2 > (real)0) { real ih = M<real>::rsqrt(h2); c = x * ih; s = y * ih; }
    r[0] = c; r[1] = -s; r[2] = 0; r[3] = s; r[4] = c; r[5] = 0; r[6] = 0; r[7] = 0; r[8] = 1;
}
template <typename real> __device__ __forceinline__ void unit_xy(real x, real y, real *c, real *s) {
    real h2 = x * x + y * y;
    *c = 1; *s = 0;
    if (h2 > (real)0) { real ih = M<real>::rsqrt(h2); *c = x * ih; *s = y * ih; }
}

// nearest rotation (orthogonal polar factor) == U V^T of the SVD, quadrotor_dynamics.py:546-551.
// Newton iteration X <- (X + X^-T)/2; from a nearly orthogonal start 4 iterations reach round-off.
template <typename real> __device__ __forceinline__ void polar_rotation(real x[9]) {
#pragma unroll 1
    for (int it = 0; it < 5; ++it) {
        real c00 = x[4] * x[8] - x[5] * x[7], c01 = x[5] * x[6] - x[3] * x[8], c02 = x[3] * x[7] - x[4] * x[6];
        real c10 = x[2] * x[7] - x[1] * x[8], c11 = x[0] * x[8] - x[2] * x[6], c12 = x[1] * x[6] - x[0] * x[7];
        real c20 = x[1] * x[5] - x[2] * x[4], c21 = x[2] * x[3] - x[0] * x[5], c22 = x[0] * x[4] - x[1] * x[3];
        real inv = (real)1.0 / (x[0] * c00 + x[1] * c01 + x[2] * c02), h = (real)0.5;
        x[0] = h * (x[0] + c00 * inv); x[1] = h * (x[1] + c01 * inv); x[2] = h * (x[2] + c02 * inv);
        x[3] = h * (x[3] + c10 * inv); x[4] = h * (x[4] + c11 * inv); x[5] = h * (x[5] + c12 * inv);
        x[6] = h * (x[6] + c20 * inv); x[7] = h * (x[7] + c21 * inv); x[8] = h * (x[8] + c22 * inv);
    }
}

// ------------------------------------------------------------------------------------------------
// one physics sub-step: step1_numba quadrotor_dynamics.py:348-383 =
//   calculate_torque_integrate_rotations_and_update_omega :498-566 + room clip :360-367 +
//   floor_interaction_numba :570-639 (numpy semantics :389-457 behind floor_mode) +
//   compute_velocity_and_acceleration :643-649
// ------------------------------------------------------------------------------------------------
template <typename real>
__device__ __forceinline__ void substep(const Consts<real> &c, const RngKey &key, int drone, int sub, Drone<real> &d,
                                        const real cmds[4], real acc[3]) {
    const real dt = c.dt;
    real thrust_z = 0, torque[3] = {0, 0, 0};
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        real cmd = cmds[m];  // already clipped to [0,1] by RawControl
        real tau = (cmd < d.cmds_damp[m]) ? c.motor_tau_down : c.motor_tau_up;
        tau = tau > (real)1 ? (real)1 : tau;
        real trot = M<real>::sqrt(cmd);
        d.rot_damp[m] = tau * (trot - d.rot_damp[m]) + d.rot_damp[m];
        real cd = d.rot_damp[m] * d.rot_damp[m];
        cd = clipr<real>(cd + cmd * d.ou[m], (real)0, (real)1);
        d.cmds_damp[m] = cd;
        real th = c.thrust_max[m] * (((real)1 - c.motor_linearity) * (cd * cd) + c.motor_linearity * cd);
        torque[0] += c.prop_cross[m][0] * th;
        torque[1] += c.prop_cross[m][1] * th;
        torque[2] += c.prop_cross[m][2] * th + c.torque_max[m] * c.prop_ccw[m] * cd;
        thrust_z += th;
    }
    // Rodrigues rotation update (:535-544)
    real *R = d.rot, *om = d.omega;
    real wv[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) wv[r] = R[r * 3] * om[0] + R[r * 3 + 1] * om[1] + R[r * 3 + 2] * om[2];
    real wn = norm3<real>(wv);
    if (wn != (real)0) {
        real iw = M<real>::rcp(wn);
        real kx = wv[0] * iw, ky = wv[1] * iw, kz = wv[2] * iw;
        real K[9] = {0, -kz, ky, kz, 0, -kx, -ky, kx, 0};
        real s, cc; M<real>::sin_omcos_small(wn * dt, &s, &cc);
        real KK[9], dR[9];
        matmul3<real>(K, K, KK);
#pragma unroll
        for (int q = 0; q < 9; ++q) dR[q] = ((q % 4 == 0) ? (real)1 : (real)0) + s * K[q] + cc * KK[q];
        matmul3<real>(dR, R, R);
    }
    // rare re-orthogonalisation (:546-551); counter in flags bits 16..23
    uint32_t svd = ((d.flags & F_SVD_MASK) >> F_SVD_SHIFT) + 1;
    if (__builtin_expect((int)svd >= c.svd_period, 0)) { polar_rotation<real>(R); svd = 0; }   // once in svd_period sub-steps
    d.flags = (d.flags & ~F_SVD_MASK) | (svd << F_SVD_SHIFT);
    // omega update (:555-560)
    real Iw[3] = {c.inertia[0] * om[0], c.inertia[1] * om[1], c.inertia[2] * om[2]};
    real a0 = -om[0], a1 = -om[1], a2 = -om[2];
    real cr[3] = {a1 * Iw[2] - a2 * Iw[1], a2 * Iw[0] - a0 * Iw[2], a0 * Iw[1] - a1 * Iw[0]};
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        real od = c.inv_inertia[q] * (cr[q] + torque[q]);
        real damp = clipr<real>(c.damp_omega_quadratic * (om[q] * om[q]), (real)0, (real)1);
        om[q] = clipr<real>(om[q] + ((real)1 - damp) * dt * od, -c.omega_max, c.omega_max);
    }
    // position, room clip and wall / ceiling flags (:360-367, :563)
    real before[3];
#pragma unroll
    for (int q = 0; q < 3; ++q) { before[q] = d.pos[q] + dt * d.vel[q]; d.pos[q] = clipr<real>(before[q], c.room_lo[q], c.room_hi[q]); }
    uint32_t f = d.flags & ~(F_CRASH_WALL | F_CRASH_CEIL | F_CRASH_FLOOR);
    if (before[0] != d.pos[0] || before[1] != d.pos[1]) f |= F_CRASH_WALL;
    if (before[2] > d.pos[2]) f |= F_CRASH_CEIL;
    // floor interaction
    real force[3] = {R[2] * thrust_z, R[5] * thrust_z, R[8] * thrust_z};
    if (d.pos[2] <= c.floor_threshold) {
        d.pos[2] = c.floor_threshold;
        if (f & F_ON_FLOOR) {
            yaw_rot_xy<real>(R[0] + (real)1e-6, R[3], R);
            real fr = (real)0.6 * (c.mass * (real)9.81 - force[2]);
            real vn = norm3<real>(d.vel);
            bool is_static = (c.floor_mode == QS_FLOOR_NUMPY) ? (vn == (real)0) : (vn < (real)1e-6);
            if (is_static) {
                real fxy = M<real>::sqrt(force[0] * force[0] + force[1] * force[1]);
                fxy = M<real>::fmax(fxy - fr, (real)0);
                if (fxy == (real)0) { force[0] = 0; force[1] = 0; }
                else {
                    real s, cs; unit_xy<real>(force[0], force[1], &cs, &s);
                    force[0] = fxy * cs; force[1] = fxy * s;
                }
            } else {
                real s, cs;
                if (c.floor_mode == QS_FLOOR_NUMPY) unit_xy<real>((real)-1 * d.vel[0], (real)-1 * d.vel[1], &cs, &s);
                else unit_xy<real>(d.vel[0], d.vel[1], &cs, &s);
                force[0] = force[0] - cs * fr; force[1] = force[1] - s * fr;
            }
        } else {
            f |= F_ON_FLOOR | F_CRASH_FLOOR;
#pragma unroll
            for (int q = 0; q < 3; ++q) { d.vel[q] = 0; om[q] = 0; }
            real theta;
            if (__builtin_expect(R[8] < (real)0, 0)) {
                if (c.floor_mode != QS_FLOOR_NUMPY) {
                    theta = rng_uniform1<real>(key, QS_SITE_FLOOR_YAW, sub * 64, drone, 0, (real)-QS_PI_D, (real)QS_PI_D);
                    yaw_rot<real>(theta, R);
                } else {
                    real xy[3] = {-d.pos[0], -d.pos[1], 0}, n = norm3<real>(xy);
                    if (n >= (real)0.00001) { xy[0] /= n; xy[1] /= n; }
                    for (int t = 0; t < 64; ++t) {
                        theta = rng_uniform1<real>(key, QS_SITE_FLOOR_YAW, sub * 64 + t, drone, 0, (real)-QS_PI_D, (real)QS_PI_D);
                        yaw_rot<real>(theta, R);
                        if (!(R[0] * xy[0] + R[3] * xy[1] < (real)0.5)) break;
                    }
                }
            } else {
                yaw_rot_xy<real>(R[0] + (real)1e-6, R[3], R);
            }
#pragma unroll
            for (int m = 0; m < 4; ++m) { d.cmds_damp[m] = 0; d.rot_damp[m] = 0; }
        }
        acc[0] = c.inv_mass * force[0];
        acc[1] = c.inv_mass * force[1];
        acc[2] = M<real>::fmax((real)0, (real)-9.81 + c.inv_mass * force[2]);
    } else {
        f &= ~F_ON_FLOOR;
        acc[0] = c.inv_mass * force[0];
        acc[1] = c.inv_mass * force[1];
        acc[2] = (real)-9.81 + c.inv_mass * force[2];
    }
    d.flags = f;
#pragma unroll
    for (int q = 0; q < 3; ++q) d.vel[q] = ((real)1 - c.vel_damp) * d.vel[q] + dt * acc[q];
}

// rot2quat sensor_noise.py:34-63
template <typename real> __device__ __forceinline__ void rot2quat(const real r[9], real q[4]) {
    real trace = r[0] + r[4] + r[8];
    if (trace > (real)0) {
        real S = M<real>::sqrt(trace + (real)1) * (real)2, iS = M<real>::rcp(S);
        q[0] = (real)0.25 * S; q[1] = (r[7] - r[5]) * iS; q[2] = (r[2] - r[6]) * iS; q[3] = (r[3] - r[1]) * iS;
    } else if (r[0] > r[4] && r[0] > r[8]) {
        real S = M<real>::sqrt((real)1 + r[0] - r[4] - r[8]) * (real)2, iS = M<real>::rcp(S);
        q[0] = (r[7] - r[5]) * iS; q[1] = (real)0.25 * S; q[2] = (r[1] + r[3]) * iS; q[3] = (r[2] + r[6]) * iS;
    } else if (r[4] > r[8]) {
        real S = M<real>::sqrt((real)1 + r[4] - r[0] - r[8]) * (real)2, iS = M<real>::rcp(S);
        q[0] = (r[2] - r[6]) * iS; q[1] = (r[1] + r[3]) * iS; q[2] = (real)0.25 * S; q[3] = (r[5] + r[7]) * iS;
    } else {
        real S = M<real>::sqrt((real)1 + r[8] - r[0] - r[4]) * (real)2, iS = M<real>::rcp(S);
        q[0] = (r[3] - r[1]) * iS; q[1] = (r[2] + r[6]) * iS; q[2] = (r[5] + r[7]) * iS; q[3] = (real)0.25 * S;
    }
}

// Sensor-noise draws of one self observation (sensor_noise.py:235-261 order): additive noise on pos, vel, omega and the
// small-angle rotation noise theta.  Independent of the dynamic state, so the step kernel draws them while the state
// loads are still in flight.
template <typename real> struct SensNoise { real p[3], v[3], w[3], th[3]; };

template <typename real>
__device__ __forceinline__ void sensor_noise_draw(const Consts<real> &c, const RngKey &key, int drone, int pass, SensNoise<real> &n) {
    // on a tape every group of sensor_noise.py:128-168 is present, also the uniform(-0, 0) ones and the accelerometer's 6 draws
    const bool tape = QS_ON_TAPE(key);
    real u[3];
    rng_normal_s<real, 3>(key, QS_SITE_SENS_POS_N, pass, drone, 0, c.pos_norm_std, n.p);
    if (tape || c.pos_unif_range != (real)0) {
        rng_uniform<real, 3>(key, QS_SITE_SENS_POS_U, pass, drone, 0, -c.pos_unif_range, c.pos_unif_range, u);
#pragma unroll
        for (int q = 0; q < 3; ++q) n.p[q] += u[q];
    }
    rng_normal_s<real, 3>(key, QS_SITE_SENS_VEL_N, pass, drone, 0, c.vel_norm_std, n.v);
    if (tape || c.vel_unif_range != (real)0) {
        rng_uniform<real, 3>(key, QS_SITE_SENS_VEL_U, pass, drone, 0, -c.vel_unif_range, c.vel_unif_range, u);
#pragma unroll
        for (int q = 0; q < 3; ++q) n.v[q] += u[q];
    }
    rng_normal_s<real, 3>(key, QS_SITE_SENS_OMEGA_N, pass, drone, 0, c.gyro_noise_density, n.w);
#pragma unroll
    for (int q = 0; q < 3; ++q) n.th[q] = 0;
    if (tape || c.quat_norm_std != (real)0) rng_normal_s<real, 3>(key, QS_SITE_SENS_THETA_N, pass, drone, 0, c.quat_norm_std, n.th);
    if (tape || c.quat_unif_range != (real)0) {
        rng_uniform<real, 3>(key, QS_SITE_SENS_THETA_U, pass, drone, 0, -c.quat_unif_range, c.quat_unif_range, u);
#pragma unroll
        for (int q = 0; q < 3; ++q) n.th[q] += u[q];
    }
    if (tape) tape_skip(key, 6);   // accelerometer noise: drawn by the reference, in no obs_repr
}

// The four always-needed normal groups of a control step (OU thrust noise + sensor pos / vel / omega noise, pass 0)
// from four Philox blocks advanced in lockstep.  Same sites / slots / values as the one-at-a-time draws.
template <typename real>
__device__ __forceinline__ void step_noise_draw(const Consts<real> &c, const RngKey &key, int drone, real zou[4], SensNoise<real> &n) {
    const uint32_t c0[4] = {key.env, key.env, key.env, key.env};
    const uint32_t c2[4] = {QS_SITE_OU, QS_SITE_SENS_POS_N, QS_SITE_SENS_VEL_N, QS_SITE_SENS_OMEGA_N};   // slot (pass) 0
    const uint32_t c3[4] = {(uint32_t)drone, (uint32_t)drone, (uint32_t)drone, (uint32_t)drone};
    uint32_t w[4][4];
    philox4x32_x4(c0, key.step, c2, c3, key.k0, key.k1, w);
    real z[4];
    box_muller4<real>(w[0], zou);
    box_muller4<real>(w[1], z);
#pragma unroll
    for (int q = 0; q < 3; ++q) n.p[q] = c.pos_norm_std * z[q];
    box_muller4<real>(w[2], z);
#pragma unroll
    for (int q = 0; q < 3; ++q) n.v[q] = c.vel_norm_std * z[q];
    box_muller4<real>(w[3], z);
#pragma unroll
    for (int q = 0; q < 3; ++q) { n.w[q] = c.gyro_noise_density * z[q]; n.th[q] = 0; }
    if (c.pos_unif_range != (real)0 || c.vel_unif_range != (real)0 || c.quat_norm_std != (real)0 || c.quat_unif_range != (real)0) {
        real u[3];   // non-default sensor model: the extra terms one group at a time
        if (c.pos_unif_range != (real)0) { rng_uniform<real,
            3>(key, QS_SITE_SENS_POS_U, 0, drone, 0, -c.pos_unif_range, c.pos_unif_range, u); for (int q = 0; q < 3; ++q) n.p[q] += u[q]; }
        if (c.vel_unif_range != (real)0) { rng_uniform<real,
            3>(key, QS_SITE_SENS_VEL_U, 0, drone, 0, -c.vel_unif_range, c.vel_unif_range, u); for (int q = 0; q < 3; ++q) n.v[q] += u[q]; }
        if (c.quat_norm_std != (real)0) { rng_normal<real, 3>(key, QS_SITE_SENS_THETA_N, 0, drone, 0, z);
            for (int q = 0; q < 3; ++q) n.th[q] = c.quat_norm_std * z[q]; }
        if (c.quat_unif_range != (real)0) { rng_uniform<real,
            3>(key, QS_SITE_SENS_THETA_U, 0, drone, 0, -c.quat_unif_range, c.quat_unif_range, u);
        for (int q = 0; q < 3; ++q) n.th[q] += u[q]; }
    }
}

// Self observation: get_state.py:6-72 + sensor_noise.py:112-218.  Writes self_dim values to o[] (LDS row).
template <typename real>
__device__ __forceinline__ void self_obs(const Consts<real> &c, const SensNoise<real> &n, const Drone<real> &d, const real goal[3],
    real *o) {
    real p[3], v[3], w[3], R[9];
    if (!c.sense_noise) {
#pragma unroll
        for (int q = 0; q < 3; ++q) { p[q] = d.pos[q]; v[q] = d.vel[q]; w[q] = d.omega[q]; }
#pragma unroll
        for (int q = 0; q < 9; ++q) R[q] = d.rot[q];
    } else {
#pragma unroll
        for (int q = 0; q < 3; ++q) { p[q] = d.pos[q] + n.p[q]; v[q] = d.vel[q] + n.v[q]; w[q] = d.omega[q] + n.w[q]; }
        // R -> quat -> quat (x) dq(theta) -> R  (sensor_noise.py:205-210; an identity dq still re-derives R)
        if (sizeof(real) == 4 && c.quat_norm_std == (real)0 && c.quat_unif_range == (real)0) {
            // fp32 production path, no rotation noise configured: quat2R(rot2quat(R)) reproduces an orthonormal R to a few
            // ulp (1e-7), far inside the 1e-5 tolerance, so the round trip is skipped.  The f64 parity instantiation keeps it.
#pragma unroll
            for (int k = 0; k < 9; ++k) R[k] = d.rot[k];
        } else {
        real q[4];
        rot2quat<real>(d.rot, q);
        real qw = q[0], qx = q[1], qy = q[2], qz = q[3];
        if (c.quat_norm_std != (real)0 || c.quat_unif_range != (real)0) {
            const real *th = n.th;
            real nt = norm3<real>(th), qsq = nt * nt / (real)4, qt[4];
            if (qsq < (real)1) { qt[0] = M<real>::sqrt((real)1 - qsq); qt[1] = th[0] * (real)0.5; qt[2] = th[1] * (real)0.5;
                qt[3] = th[2] * (real)0.5; }
            else { real ww = (real)1 / M<real>::sqrt((real)1 + qsq), f = (real)0.5 * ww; qt[0] = ww; qt[1] = th[0] * f;
                qt[2] = th[1] * f; qt[3] = th[2] * f; }
            real qn = M<real>::sqrt(qt[0] * qt[0] + qt[1] * qt[1] + qt[2] * qt[2] + qt[3] * qt[3]);
#pragma unroll
            for (int k = 0; k < 4; ++k) qt[k] /= qn;
            qw = q[0] * qt[0] - q[1] * qt[1] - q[2] * qt[2] - q[3] * qt[3];
            qx = q[0] * qt[1] + q[1] * qt[0] - q[2] * qt[3] + q[3] * qt[2];
            qy = q[0] * qt[2] + q[1] * qt[3] + q[2] * qt[0] - q[3] * qt[1];
            qz = q[0] * qt[3] - q[1] * qt[2] + q[2] * qt[1] + q[3] * qt[0];
        }   // theta == 0: quat_from_small_angle gives exactly (1,0,0,0) and q (x) (1,0,0,0) == q bit for bit
        R[0] = (real)1 - 2 * qy * qy - 2 * qz * qz; R[1] = 2 * qx * qy - 2 * qz * qw; R[2] = 2 * qx * qz + 2 * qy * qw;
        R[3] = 2 * qx * qy + 2 * qz * qw; R[4] = (real)1 - 2 * qx * qx - 2 * qz * qz; R[5] = 2 * qy * qz - 2 * qx * qw;
        R[6] = 2 * qx * qz - 2 * qy * qw; R[7] = 2 * qy * qz + 2 * qx * qw; R[8] = (real)1 - 2 * qx * qx - 2 * qy * qy;
        }
    }
#pragma unroll
    for (int q = 0; q < 3; ++q) { o[q] = p[q] - goal[q]; o[3 + q] = v[q]; o[15 + q] = w[q]; }
#pragma unroll
    for (int q = 0; q < 9; ++q) o[6 + q] = R[q];
    if (c.obs_repr == QS_OBS_XYZ_VXYZ_R_OMEGA_FLOOR) o[18] = p[2];
    else if (c.obs_repr == QS_OBS_XYZ_VXYZ_R_OMEGA_WALL) {
#pragma unroll
        for (int q = 0; q < 3; ++q) { o[18 + q] = clipr<real>(p[q] - c.room_lo[q], (real)0, (real)5);
            o[21 + q] = clipr<real>(c.room_hi[q] - p[q], (real)0, (real)5); }
    }
}

// compute_new_vel collisions/utils.py:8-18
template <typename real> __device__ __forceinline__ void compute_new_vel(real max_vel_magn, real vel[3], const real shift[3], real decay) {
    real vn[3] = {vel[0] + shift[0], vel[1] + shift[1], vel[2] + shift[2]};
    real mag = norm3<real>(vn), den = (mag == (real)0) ? mag + (real)1e-5 : mag;
    real nm = M<real>::fmin(mag * decay, max_vel_magn);
#pragma unroll
    for (int q = 0; q < 3; ++q) { real nv = (vn[q] / den) * nm; real sh = nv - vel[q]; vel[q] += sh; }
}
// compute_new_omega collisions/utils.py:22-34
template <typename real> __device__ __forceinline__ void compute_new_omega(const real u[4], real out[3]) {
    real mag = norm3<real>(u), den = (mag == (real)0) ? mag + (real)1e-5 : mag;
#pragma unroll
    for (int q = 0; q < 3; ++q) out[q] = (u[q] / den) * u[3];
}

// perform_collision_with_obstacle collisions/obstacles.py:23-50 (+ :9-20)
template <typename real>
__device__ __forceinline__ void collide_obstacle(const Consts<real> &c, const RngKey &key, int drone, Drone<real> &d, real ox, real oy,
    real obst_size) {
    real n[3] = {d.pos[0] - ox, d.pos[1] - oy, 0};
    real mag = norm3<real>(n), den = (mag == (real)0) ? mag + (real)1e-5 : mag;
    n[0] /= den; n[1] /= den;
    real vmag = norm3<real>(d.vel), nv[3] = {vmag * n[0], vmag * n[1], vmag * n[2]}, noise[3] = {0, 0, 0};
    for (int t = 0; t < 3; ++t) {
        real cons[3], n1[3], tmp[3], chk[3];
        rng_normal_s<real, 3>(key, QS_SITE_OBST_N, t * 2 + 0, drone, 0, (real)0.1, cons);
        rng_normal_s<real, 3>(key, QS_SITE_OBST_N, t * 2 + 1, drone, 0, (real)0.05, n1);
#pragma unroll
        for (int q = 0; q < 3; ++q) { tmp[q] = cons[q] + n1[q]; chk[q] = nv[q] + tmp[q]; }
        if (dot3<real>(chk, n) > (real)0) { noise[0] = tmp[0]; noise[1] = tmp[1]; noise[2] = tmp[2]; break; }
    }
    real diff[3] = {d.pos[0] - ox, d.pos[1] - oy, d.pos[2] - c.room_mid_z};
    bool inside = norm3<real>(diff) < obst_size / (real)2;
    real decay = inside ? rng_uniform1<real>(key, QS_SITE_OBST_U, 0, drone, 0, (real)1, (real)1)
                        : rng_uniform1<real>(key, QS_SITE_OBST_U, 0, drone, 0, (real)0.2, (real)0.8);
    real shift[3] = {nv[0] - d.vel[0] + noise[0], nv[1] - d.vel[1] + noise[1], nv[2] - d.vel[2] + noise[2]};
    compute_new_vel<real>(vmag, d.vel, shift, decay);
    real u[4];
    if (QS_ON_TAPE(key)) { for (int q = 0; q < 4; ++q) u[q] = (real)tape_pop(key); }   // uniform(-1,1,3), uniform(pi/2, pi)
    else {
        uint32_t w[4]; rng_words(key, QS_SITE_OBST_W, 0, drone, 0, w);
        u[0] = (real)-1 + (real)2 * u01<real>(w[0]); u[1] = (real)-1 + (real)2 * u01<real>(w[1]);
        u[2] = (real)-1 + (real)2 * u01<real>(w[2]);
        u[3] = (real)(0.5 * QS_PI_D) + (real)(QS_PI_D - 0.5 * QS_PI_D) * u01<real>(w[3]);
    }
    real dw[3]; compute_new_omega<real>(u, dw);
#pragma unroll
    for (int q = 0; q < 3; ++q) d.omega[q] += dw[q];
}

// perform_collision_with_wall collisions/room.py:6-44 / perform_collision_with_ceiling :91-113
template <typename real>
__device__ __forceinline__ void collide_room(const Consts<real> &c, const RngKey &key, int drone, Drone<real> &d, bool is_wall) {
    int site = is_wall ? QS_SITE_WALL : QS_SITE_CEIL;
    const bool tape = QS_ON_TAPE(key);
    real speed = norm3<real>(d.vel), dir[3], rs;
    uint32_t w[4] = {0, 0, 0, 0}, w1[4] = {0, 0, 0, 0};
    // the reference's call order, room.py:10-41
    if (tape) { rs = (real)tape_pop(key); for (int q = 0; q < 3; ++q) dir[q] = (real)tape_pop(key); }
    else {
        rng_words(key, site, 0, drone, 0, w);
        rs = (real)0.2 * speed + ((real)0.8 * speed - (real)0.2 * speed) * u01<real>(w[0]);
#pragma unroll
        for (int q = 0; q < 3; ++q) dir[q] = (real)-1 + (real)2 * u01<real>(w[1 + q]);
        rng_words(key, site, 1, drone, 0, w1);
    }
    rs = clipr<real>(rs, (real)0.1, (real)6);
    if (is_wall) {
        if (d.pos[0] == c.room_lo[0]) dir[0] = tape ? (real)tape_pop(key) : (real)0.1 + (real)0.9 * u01<real>(w1[0]);
        else if (d.pos[0] == c.room_hi[0]) dir[0] = tape ? (real)tape_pop(key) : (real)-1 + (real)0.9 * u01<real>(w1[0]);
        if (d.pos[1] == c.room_lo[1]) dir[1] = tape ? (real)tape_pop(key) : (real)0.1 + (real)0.9 * u01<real>(w1[1]);
        else if (d.pos[1] == c.room_hi[1]) dir[1] = tape ? (real)tape_pop(key) : (real)-1 + (real)0.9 * u01<real>(w1[1]);
    }
    dir[2] = tape ? (real)tape_pop(key) : (real)-1 + (real)0.5 * u01<real>(w1[2]);
    real dm = norm3<real>(dir);
#pragma unroll
    for (int q = 0; q < 3; ++q) d.vel[q] = rs * (dir[q] / (dm + (real)1e-5));
    real u[4];
    if (tape) { for (int q = 0; q < 4; ++q) u[q] = (real)tape_pop(key); }
    else {
        rng_words(key, site, 2, drone, 0, w);
        u[0] = (real)-1 + (real)2 * u01<real>(w[0]); u[1] = (real)-1 + (real)2 * u01<real>(w[1]);
        u[2] = (real)-1 + (real)2 * u01<real>(w[2]);
        u[3] = (real)(10.0 * QS_PI_D) + (real)(10.0 * QS_PI_D) * u01<real>(w[3]);
    }
    real um = norm3<real>(u);
#pragma unroll
    for (int q = 0; q < 3; ++q) { real dwq = u[q] / (um + (real)1e-5); dwq *= u[3]; d.omega[q] += dwq; }
}

// ------------------------------------------------------------------------------------------------
// scenarios (per-environment, executed by one lane at episode boundaries / goal switches)
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ bool f_is_circle(int f) { return f <= 2; }
__device__ __forceinline__ bool f_is_grid(int f) { return f >= 4 && f <= 6; }
__device__ __forceinline__ int f_suffix(int f) { return (f == 0 || f == 4) ? 0 : ((f == 1 || f == 5) ? 1 : ((f == 2 || f == 6) ? 2 : -1)); }
// Exact quotient / remainder of small non-negative integers (0 <= a < 2^20, b >= 1) by ONE float divide instead of the ~35-instruction
// integer division sequence: (a + 0.5) / b stays 0.5 / b away from every integer, the divide's rounding error is below a / b * 2^-21.
// Used for the per-step `tick % period` tests of the full scenario set (qs_scenarios.h).  (In the formation builders and the swarm_vs_swarm
// switch of the fast kernels the same substitution measured + 0.14 us on the C4 step, profiles/r03y_c4_ab.txt: left as integer code.)
__device__ __forceinline__ int idiv_small(int a, int b) { return (int)__fdividef((float)a + 0.5f, (float)b); }
__device__ __forceinline__ int imod_small(int a, int b) { return a - b * idiv_small(a, b); }
__device__ __forceinline__ void grid_dim(int num, int *d1, int *d2) {  // scenarios/utils.py:117-128
    int a = (int)floorf(sqrtf((float)num));
    while (a * a > num) --a;
    while ((a + 1) * (a + 1) <= num) ++a;
    while (a > 1) { if (num % a == 0) break; --a; }
    *d1 = a; *d2 = num / a;
}
template <typename real> struct Formation { int f, per_layer; real lo, hi, size, layer_dist; };
// The per-episode scenario code below is cold and big; out of line it keeps the step kernels small, but a kernel that CALLS anything
// needs a stack (private segment > 0), and waves with a private segment are dispatched more slowly.  QS_INLINE_COLD=1 inlines it.
#ifndef QS_INLINE_COLD
#ifdef QS_SPEC
#define QS_INLINE_COLD 1   // config-specialised objects: C4 13.86 -> 13.60 us per step (profiles/r03f_bench_lines.txt)
#else
#define QS_INLINE_COLD 0
#endif
#endif
#if QS_INLINE_COLD
#define QS_COLD __forceinline__
#else
#define QS_COLD
#endif

template <typename real> __device__ __forceinline__ void goal_by_formation(int f, real p0, real p1, real layer, real g[3]) {
    int s = f_suffix(f);
    if (s == 0) { g[0] = p0; g[1] = p1; g[2] = layer; }
    else if (s == 1) { g[0] = p0; g[1] = layer; g[2] = p1; }
    else { g[0] = layer; g[1] = p0; g[2] = p1; }
}

// QuadrotorScenario.generate_goals scenarios/base.py:39-113.  out: [rows][3] with stride `ld` reals per row.
template <typename real>
__device__ QS_COLD int generate_goals(const Formation<real> &F, int n, int fd, const real center[3], real *out, int ld) {
    // Opaque n: in a config-specialised build n is a literal and the loops below get fully unrolled + SLP-vectorised; the
    // fp32 grid branch then came out wrong on gfx950 (rows 0 and 2 of the second half-swarm, ROCm 7.2 clang), and this is
    // cold per-episode code where unrolling buys nothing.
    asm volatile("" : "+v"(n));
    int f = F.f, per = F.per_layer, rows = n;
    real size = F.size;
    if (f_is_circle(f)) {
        for (int i = 0; i < n; ++i) {
            int layer = i / per, cur = (n <= per) ? n : ((layer < n / per) ? per : n % per);
            real deg = (real)2 * (real)QS_PI_D * (real)(i % cur) / (real)cur, s, cs;
            M<real>::sincos(deg, &s, &cs);
            real g[3];
            goal_by_formation<real>(f, size * cs, size * s, (real)layer * F.layer_dist, g);
            for (int q = 0; q < 3; ++q) out[i * ld + q] = g[q] + center[q];
        }
    } else if (f == 3) {
        int m = n < 3 ? 3 : n;
        // in double throughout: the azimuth s * x reaches +-(0.1 + 1.2 m) rad (77 rad for 64 drones), where fp32 resolves 4e-6 rad and the
        // fp32 fast-math sin / cos lose 1e-4 - times a formation radius of a few metres that is more than the 1e-5 the fp32 path is
        // specified to (cold per-episode code)
        const double x = 0.1 + 1.2 * m, start = -1.0 + 1.0 / (m - 1.0), inc = (2.0 - 2.0 / (m - 1.0)) / (m - 1.0);
        for (int j = 0; j < m; ++j) {
            const double sj = start + j * inc, sg = (double)((sj > 0) - (sj < 0));
            const double xx = sj * x, yy = (QS_PI_D / 2.) * sg * (1.0 - ::sqrt(1.0 - ::fabs(sj)));
            const double sx = ::sin(xx), cx = ::cos(xx), sy = ::sin(yy), cy = ::cos(yy);
            out[j * ld + 0] = (real)((double)size * (cx * cy) + (double)center[0]);
            out[j * ld + 1] = (real)((double)size * (sx * cy) + (double)center[1]);
            out[j * ld + 2] = (real)((double)size * sy + (double)center[2]);
        }
        rows = m;
    } else if (f_is_grid(f)) {
        real mean[3] = {0, 0, 0};
        for (int i = 0; i < n; ++i) {
            int layer = i / per, d1, d2;
            int cnt = (n <= per) ? n : ((layer < n / per) ? per : n % per);
            grid_dim(cnt, &d1, &d2);
            real g[3];
            goal_by_formation<real>(f, size * (real)(i % d2), size * (real)((i / d2) % d1), (real)layer * F.layer_dist, g);
            for (int q = 0; q < 3; ++q) { out[i * ld + q] = g[q]; mean[q] += g[q]; }
        }
        for (int q = 0; q < 3; ++q) mean[q] /= (real)n;
        for (int i = 0; i < n; ++i) for (int q = 0; q < 3; ++q) out[i * ld + q] = out[i * ld + q] - mean[q] + center[q];
    } else {
        real mean[3] = {0, 0, 0};
        for (int i = 0; i < n; ++i) {
            real g[3] = {center[2] + size * (real)(i / (fd * fd)), size * (real)((i / fd) % fd), size * (real)(i % fd)};
            for (int q = 0; q < 3; ++q) { out[i * ld + q] = g[q]; mean[q] += g[q]; }
        }
        for (int q = 0; q < 3; ++q) mean[q] /= (real)n;
        for (int i = 0; i < n; ++i) for (int q = 0; q < 3; ++q) out[i * ld + q] = out[i * ld + q] - mean[q] + center[q];
    }
    return rows;
}

// QUADS_PARAMS_DICT scenarios/utils.py:33-51: number of candidate formations and [low, high] formation size
__device__ __forceinline__ void scen_params(int scen, int *nform, float *lo, float *hi) {
    *nform = 1; *lo = 0.f; *hi = 0.f;
    if (scen == QS_SCENARIO_STATIC_DIFF_GOAL || scen == QS_SCENARIO_DYNAMIC_DIFF_GOAL || scen == QS_SCENARIO_SWARM_VS_SWARM
        || scen == QS_SCENARIO_RUN_AWAY) { *nform = 8; *lo = 0.25f; *hi = 0.5f; }
    else if (scen == QS_SCENARIO_SWAP_GOALS) { *nform = 8; *lo = 0.4f; *hi = 0.8f; }
    else if (scen == QS_SCENARIO_DYNAMIC_FORMATIONS) { *nform = 8; *lo = 0.f; *hi = 1.0f; }
    else if (scen == QS_SCENARIO_O_SWAP_GOALS) { *nform = 7; *lo = 0.4f; *hi = 0.8f; }
}

template <typename real> __device__ __forceinline__ int rng_index(const RngKey &k, int site, int slot, int n);
// update_formation_and_relate_param scenarios/base.py:123-135
template <typename real>
__device__ QS_COLD void update_formation(int scen, const RngKey &key, int slot, int num_agents, Formation<real> &F) {
    int nform; float lof, hif;
    scen_params(scen, &nform, &lof, &hif);
    // 5*0.05 etc. are evaluated in double by the reference; 0.25/0.5/1.0 are exact, 0.4/0.8 need the double literal
    real lo = (scen == QS_SCENARIO_SWAP_GOALS || scen == QS_SCENARIO_O_SWAP_GOALS) ? (real)(8 * 0.05) : (real)lof;
    real hi = (scen == QS_SCENARIO_SWAP_GOALS || scen == QS_SCENARIO_O_SWAP_GOALS) ? (real)(16 * 0.05) : (real)hif;
    int fi = rng_index<real>(key, QS_SITE_SCEN, slot + 0, nform);
    F.f = fi;
    F.per_layer = f_is_circle(fi) ? 8 : (f_is_grid(fi) ? 50 : 8);
    int n = (scen == QS_SCENARIO_SWARM_VS_SWARM) ? num_agents / 2 : num_agents;
    if (f_is_circle(fi)) {
        real theta = (real)2 * (real)QS_PI_D / (real)F.per_layer, sn = M<real>::sin(theta / (real)2);
        F.lo = ((real)0.5 * lo) / sn; F.hi = ((real)0.5 * hi) / sn;
    } else if (fi == 3) {
        real A = (real)1.75388487222762, B = (real)0.860487305801679, C = (real)10.3632729642351, D = (real)0.0920858134405214;
        real ratio = (A - D) / ((real)1 + M<real>::pow((real)n / C, B)) + D;
        F.lo = lo / ratio; F.hi = hi / ratio;
    } else { F.lo = lo; F.hi = hi; }
    F.size = rng_uniform1<real>(key, QS_SITE_SCEN, slot + 1, 0, 0, F.lo, F.hi);
    F.layer_dist = rng_uniform1<real>(key, QS_SITE_SCEN, slot + 2, 0, 0, F.lo, F.hi);
}

// floor(u * n) and lo + (hi-lo)*u evaluated in double from the exactly-representable uniform: identical to the oracle
// in both precisions (these decide indices / switching ticks, where an fp32 rounding would flip a discrete outcome)
template <typename real> __device__ __forceinline__ int rng_index(const RngKey &k, int site, int slot, int n) {
    if (QS_ON_TAPE(k)) return (int)tape_pop(k);   // the tape holds the index the reference drew (np.random.choice / randint)
    double u = (double)rng_uniform1<real>(k, site, slot, 0, 0, (real)0, (real)1);
    int j = (int)(u * (double)n);
    return j >= n ? n - 1 : j;
}
template <typename real> __device__ __forceinline__ int draw_period(const RngKey &k, int slot, double lo, double hi, int control_freq) {
    if (QS_ON_TAPE(k)) return (int)(tape_pop(k) * (double)control_freq);   // the tape holds uniform(lo, hi) itself
    double u = (double)rng_uniform1<real>(k, QS_SITE_SCEN, slot, 0, 0, (real)0, (real)1);
    return (int)((lo + (hi - lo) * u) * (double)control_freq);
}

// np.random.shuffle on rows [0,n) of buf (stride ld): Fisher-Yates with the QS_SITE_SCEN_SHUFFLE stream
template <typename real>
__device__ QS_COLD void shuffle_rows(const RngKey &key, real *buf, int ld, int n, int slot_base) {
#ifdef QS_TAPE
    if (QS_ON_TAPE(key)) {   // the tape holds the permutation itself: rows[k] = old[perm[k]], applied in place cycle by cycle (n <= 64)
        const double *perm = key.tape + *key.cur;
        uint64_t seen = 0;
        for (int s0 = 0; s0 < n; ++s0) {
            if (seen >> s0 & 1) continue;
            real t[3] = {buf[s0 * ld], buf[s0 * ld + 1], buf[s0 * ld + 2]};
            int k = s0;
            for (;;) {
                seen |= 1ull << k;
                const int src = (int)perm[k];
                if (src == s0) { for (int q = 0; q < 3; ++q) buf[k * ld + q] = t[q]; break; }
                for (int q = 0; q < 3; ++q) buf[k * ld + q] = buf[src * ld + q];
                k = src;
            }
        }
        tape_skip(key, n);
        return;
    }
#endif
    // the oracle builds perm by swapping from the top, then gathers rows[k] = old[perm[k]]; applying the same
    // swaps directly to the rows is the identical permutation.
    for (int i = n - 1; i >= 1; --i) {
        int j = rng_index<real>(key, QS_SITE_SCEN_SHUFFLE, slot_base + i, i + 1);
        for (int q = 0; q < 3; ++q) { real t = buf[i * ld + q]; buf[i * ld + q] = buf[j * ld + q]; buf[j * ld + q] = t; }
    }
}

// Scenario_swarm_vs_swarm.create_formations swarm_vs_swarm.py:52-57 (+ shuffle of update_goals :66-69).
// goals: [>= 2*N+6][3] scratch rows (stride 3); the first N rows are the drones' goals afterwards.
template <typename real>
__device__ QS_COLD void svs_create_formations(const RngKey &key, const Formation<real> &F, int N, const int fd[2], const real c1[3],
                                      const real c2[3], bool do_shuffle, real *goals) {
    int n1 = N / 2, n2 = N - N / 2;
    int r1 = generate_goals<real>(F, n1, fd[0], c1, goals, 3);
    if (do_shuffle) shuffle_rows<real>(key, goals, 3, r1, 0);
    int r2 = generate_goals<real>(F, n2, fd[1], c2, goals + r1 * 3, 3);
    if (do_shuffle) shuffle_rows<real>(key, goals + r1 * 3, 3, r2, 256);
}

// ------------------------------------------------------------------------------------------------
// The same two formations + shuffles (Scenario_swarm_vs_swarm.create_formations / update_goals, swarm_vs_swarm.py:52-69) built by the
// WAVE: lane i of the environment makes goal row i.  The serial form above costs one lane ~25 k cycles (32 rows of sin / cos or
// double-precision sphere points, 30 serial Fisher-Yates draws); with 512 environments that swap goals every 4-6 s, some environment
// of the batch does so on EVERY control step, and that one workgroup set the duration of the whole C4 launch (8.5 us median
// workgroup, 21.8 us kernel: profiles/r03c_wg_c4_steady_*.txt).  Same arithmetic per row, same summation order of the means, the same
// draws (QS_SITE_SCEN_SHUFFLE slots) - results are bit-identical to the serial form.
//   goals: LDS rows [>= 2N][3] of the environment (first N rows = result), scr: N ints of LDS scratch of the environment.
// Must be called by all lanes of the wave with `on` uniform per environment; needs N / 2 >= 3 (every formation has as many rows as drones).
// ------------------------------------------------------------------------------------------------
template <typename real>
__device__ __forceinline__ void goal_row(const Formation<real> &F, int n, int fd, const real center[3], int i, real g[3],
    bool &needs_mean) {
    const int f = F.f, per = F.per_layer;
    const real size = F.size;
    needs_mean = false;
    if (f_is_circle(f)) {
        const int layer = i / per, cur = (n <= per) ? n : ((layer < n / per) ? per : n % per);
        real deg = (real)2 * (real)QS_PI_D * (real)(i % cur) / (real)cur, sn, cs;
        M<real>::sincos(deg, &sn, &cs);
        goal_by_formation<real>(f, size * cs, size * sn, (real)layer * F.layer_dist, g);
#pragma unroll
        for (int q = 0; q < 3; ++q) g[q] = g[q] + center[q];
    } else if (f == 3) {   // (double throughout, like generate_goals)
        const int m = n < 3 ? 3 : n;
        const double x = 0.1 + 1.2 * m, start = -1.0 + 1.0 / (m - 1.0), inc = (2.0 - 2.0 / (m - 1.0)) / (m - 1.0);
        const double sj = start + i * inc, sg = (double)((sj > 0) - (sj < 0));
        const double xx = sj * x, yy = (QS_PI_D / 2.) * sg * (1.0 - ::sqrt(1.0 - ::fabs(sj)));
        const double sx = ::sin(xx), cx = ::cos(xx), sy = ::sin(yy), cy = ::cos(yy);
        g[0] = (real)((double)size * (cx * cy) + (double)center[0]);
        g[1] = (real)((double)size * (sx * cy) + (double)center[1]);
        g[2] = (real)((double)size * sy + (double)center[2]);
    } else if (f_is_grid(f)) {
        const int layer = i / per, cnt = (n <= per) ? n : ((layer < n / per) ? per : n % per);
        int d1, d2;
        grid_dim(cnt, &d1, &d2);
        goal_by_formation<real>(f, size * (real)(i % d2), size * (real)((i / d2) % d1), (real)layer * F.layer_dist, g);
        needs_mean = true;
    } else {
        g[0] = center[2] + size * (real)(i / (fd * fd)); g[1] = size * (real)((i / fd) % fd); g[2] = size * (real)(i % fd);
        needs_mean = true;
    }
}

// One formation (or formation half) of an environment built / shuffled by its lanes: lane (r0 + li) makes row li of the `n` rows that start
// at row r0.  `build`, `do_shuffle`, `n`, `fd`, `cen`, `r0`, `slot_base` are uniform over the lanes of one formation; `on` is uniform per
// environment; environments of one wave may differ in all of them (their lanes share no data).
template <typename real>
__device__ __forceinline__ void formation_rows_wave(const RngKey &key, const Formation<real> &F, bool build, int n, int fd,
    const real cen[3], int li, int r0,
                                                    bool do_shuffle, int slot_base, real *goals, int *scr, int i, bool on) {
    real g[3] = {0, 0, 0};
    bool needs_mean = false;
    if (on && build) {
        goal_row<real>(F, n, fd, cen, li, g, needs_mean);
#pragma unroll
        for (int q = 0; q < 3; ++q) goals[i * 3 + q] = g[q];
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (on && needs_mean) {   // grid / cube: centre the formation; the mean is summed in row order like the serial loop
        real mean[3] = {0, 0, 0};
        for (int k0 = 0; k0 < n; k0 += 4) {   // four rows per LDS round trip; summed in row order like the serial loop
            real row[4][3];
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int q = 0; q < 3; ++q) row[u][q] = goals[(r0 + ((k0 + u < n) ? k0 + u : n - 1)) * 3 + q];
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (k0 + u < n) {
#pragma unroll
                    for (int q = 0; q < 3; ++q) mean[q] += row[u][q];
                }
        }
#pragma unroll
        for (int q = 0; q < 3; ++q) { mean[q] /= (real)n; g[q] = g[q] - mean[q] + cen[q]; }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // (every lane has read the uncentred rows)
    if (on && needs_mean) {
#pragma unroll
        for (int q = 0; q < 3; ++q) goals[i * 3 + q] = g[q];
    }
    // np.random.shuffle of the formation's rows: Fisher-Yates from the top, swap (k, j_k) with j_k = randint(k + 1), k = n-1 .. 1.  Lane
    // li draws j_li; the row that ends at position li is found by sending li back through the swaps in reverse order.
    if (on && do_shuffle) scr[i] = li >= 1 ? rng_index<real>(key, QS_SITE_SCEN_SHUFFLE, slot_base + li, li + 1) : 0;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (on && do_shuffle) {
        int pos = li;
        // eight swap partners per LDS round trip (a lone wave pays ~130 cycles per dependent LDS read: 15 of them were 2 k cycles of the
        // goal swap)
        for (int k0 = 1; k0 < n; k0 += 8) {
            int jj[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) jj[u] = scr[r0 + ((k0 + u < n) ? k0 + u : n - 1)];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int k = k0 + u;
                if (k < n) pos = (pos == k) ? jj[u] : ((pos == jj[u]) ? k : pos);
            }
        }
#pragma unroll
        for (int q = 0; q < 3; ++q) g[q] = goals[(r0 + pos) * 3 + q];
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // (every lane has read its source row)
    if (on && do_shuffle) {
#pragma unroll
        for (int q = 0; q < 3; ++q) goals[i * 3 + q] = g[q];
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
}

template <typename real>
__device__ __forceinline__ void svs_create_formations_wave(const RngKey &key, const Formation<real> &F, int N, int fd0, int fd1,
    const real c1[3], const real c2[3],
                                                           bool do_shuffle, real *goals, int *scr, int i, bool on) {
    const int n1 = N / 2, n2 = N - N / 2;
    const bool second = i >= n1;   // this lane's formation: rows, local row, first row
    formation_rows_wave<real>(key, F, true, second ? n2 : n1, second ? fd1 : fd0, second ? c2 : c1, second ? i - n1 : i, second ? n1 : 0,
        do_shuffle,
                              second ? 256 : 0, goals, scr, i, on);
}

}  // namespace qs
