// qs_step_sem.h - what one control step MEANS, written once: the reference's bookkeeping (QuadrotorEnvMulti.step,
// gym_art/quadrotor_multi/quadrotor_multi.py:413-722) for the part that is the same whichever way the work is spread over waves.
//
// The two step bodies - qs_step_team.inc (a team of 4 / 8 waves per workgroup, latency-bound small batches) and qs_step_kernel.inc (one
// wave per workgroup, throughput) - differ in WHO computes WHAT WHEN: wave roles, barriers, what travels through LDS.  What they compute
// after the pair scan is identical and lives here: room / obstacle event bits, the env-level id sets and counters, the collision
// rewards, the distance-to-goal log, and the four physical interactions in the reference's order (downwash, drone-drone, obstacle,
// wall / ceiling), and scenario.step().  A semantic fix lands here once; both bodies, the generic and the config-specialised builds,
// and the noise-tape test flavour (QS_TAPE, single-wave body only) pick it up.
//
// Every function is __forceinline__ and is entered by all lanes of the calling wave (they contain wave ballots); `Sync` is the
// caller's "every lane of this env has reached this point and sees the others' LDS writes" primitive: a wait on the LDS counter inside
// one wave of a team, __syncthreads() in the single-wave body (whose tape flavour relies on it being a real barrier).
#pragma once

namespace qs {

struct WaveSync { __device__ __forceinline__ void operator()() const { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); } };
struct BlockSync { __device__ __forceinline__ void operator()() const { __syncthreads(); } };

// the tape flavour keeps one read position per env in LDS (s_cur[le]); a lane that is about to draw serially loads it into its RngKey's
// cursor and writes it back afterwards.  No-ops without QS_TAPE.
__device__ __forceinline__ void tape_cursor_in(const RngKey &key, const int *slot) {
#ifdef QS_TAPE
    if (QS_ON_TAPE(key)) *key.cur = *slot;
#else
    (void)key; (void)slot;
#endif
}
__device__ __forceinline__ void tape_cursor_out(const RngKey &key, int *slot) {
#ifdef QS_TAPE
    if (QS_ON_TAPE(key)) *slot = *key.cur;
#else
    (void)key; (void)slot;
#endif
}

// RawControl quadrotor_control.py:53-57 (clip to [-1, 1], map to [0, 1]) and the thrust-noise OU update quad_utils.py:275-279 with this
// step's four normal draws
template <typename real>
__device__ __forceinline__ void control_and_thrust_noise(const Consts<real> &c, const real act[4], const real zou[4], Drone<real> &d,
    real cmds[4]) {
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        cmds[m] = (real)0.5 * (clipr<real>(act[m], (real)-1, (real)1) + (real)1);
        real x = d.ou[m];
        d.ou[m] = x + (c.ou_theta * ((real)0 - x) + c.thrust_noise_sigma * zou[m]);
    }
}

// compute_reward_weighted quadrotor_single.py:34-92 on the post-step state: the five per-drone terms, their sum in the reference's
// order, and the 12 info entries they feed (the obstacle entries start at zero)
template <typename real>
__device__ __forceinline__ void main_reward(const Consts<real> &c, const real *rewc, const Drone<real> &d, const real goal[3],
    const real act[4], real &rew, real *ri) {
    const real dt = c.dt;
    real diff[3] = {goal[0] - d.pos[0], goal[1] - d.pos[1], goal[2] - d.pos[2]};
    real cpr = norm3<real>(diff), cpos = rewc[QS_REW_POS] * cpr;
    real cer = M<real>::sqrt(act[0] * act[0] + act[1] * act[1] + act[2] * act[2] + act[3] * act[3]), cef = rewc[QS_REW_EFFORT] * cer;
    bool on_floor = (d.flags & F_ON_FLOOR) != 0;
    real cor = on_floor ? (real)1 : -d.rot[8], cori = rewc[QS_REW_ORIENT] * cor;
    real csr = M<real>::sqrt(d.omega[0] * d.omega[0] + d.omega[1] * d.omega[1] + d.omega[2] * d.omega[2]), cspin = rewc[QS_REW_SPIN] * csr;
    real ccr = on_floor ? (real)1 : (real)0, ccrash = rewc[QS_REW_CRASH] * ccr;
    rew = -dt * ((((cpos + cef) + ccrash) + cori) + cspin);
    ri[QS_RI_REW_MAIN] = dt * -cpos; ri[QS_RI_REW_POS] = dt * -cpos; ri[QS_RI_REW_ACTION] = dt * -cef;
    ri[QS_RI_REW_CRASH] = dt * -ccrash; ri[QS_RI_REW_ORIENT] = dt * -cori; ri[QS_RI_REW_SPIN] = dt * -cspin;
    ri[QS_RI_RAW_MAIN] = dt * -cpr; ri[QS_RI_RAW_POS] = dt * -cpr; ri[QS_RI_RAW_ACTION] = dt * -cer;
    ri[QS_RI_RAW_CRASH] = dt * -ccr; ri[QS_RI_RAW_ORIENT] = dt * -cor; ri[QS_RI_RAW_SPIN] = dt * -csr;
    ri[QS_RI_REW_QUADCOL_OBST] = 0; ri[QS_RI_RAW_QUADCOL_OBST] = 0;
}

// first obstacle hit of this drone -> event bits (quadrotor_multi.py:462-470: a hit counts as NEW when the drone was not in contact the
// step before)
__device__ __forceinline__ void obstacle_hit_bits(int obst_idx, uint32_t &flags, uint32_t &bits) {
    if (obst_idx >= 0) { bits |= B_OBST_HIT; if (!(flags & F_PREV_OBST)) bits |= B_OBST_NEW; flags |= F_PREV_OBST; }
    else flags &= ~F_PREV_OBST;
}

// calculate_room_collision quadrotor_multi.py:289-302, :491-497 (floor every step; wall / ceiling / "room" only on the step they start)
__device__ __forceinline__ void room_collision_bits(uint32_t &flags, uint32_t &bits) {
    uint32_t f = flags;
    if (f & F_CRASH_FLOOR) bits |= B_FLOOR;
    if ((f & F_CRASH_WALL) && !(f & F_PREV_WALL)) bits |= B_WALL_NEW;
    if ((f & F_CRASH_CEIL) && !(f & F_PREV_CEIL)) bits |= B_CEIL_NEW;
    if ((bits & (B_FLOOR | B_WALL_NEW | B_CEIL_NEW)) && !(f & F_PREV_ROOM)) bits |= B_ROOM_NEW;
    f &= ~(F_PREV_WALL | F_PREV_CEIL | F_PREV_ROOM);
    if (bits & B_WALL_NEW) f |= F_PREV_WALL;
    if (bits & B_CEIL_NEW) f |= F_PREV_CEIL;
    if (bits & B_ROOM_NEW) f |= F_PREV_ROOM;
    flags = f;
}

// env-level id sets of one step, as bit masks over the env's drones (bit i = drone i), identical in every lane of the env
struct EnvEvents {
    uint64_t obst_hit, obst_new, floor, wall, ceil, room;   // drones with that event bit
    uint64_t wave_newpair, newpair_any;                     // lanes of the WAVE / drones of the env that start a new colliding pair
    uint64_t unique;                                        // np.setdiff1d(curr ids, prev ids) (:440)
    int col_tick, obst_cnt, n35, n5, time_remain;
    bool settled;
};

// quadrotor_multi.py:432-459, :462-488: id sets by wave ballots; updates the drone's F_IN_COL / F_COL_*_OK flags.
// myobs: the drone's NOISY self observation (distance_to_goal_3_5 / _5 use its relative position, :474-478).
template <typename real>
__device__ __forceinline__ EnvEvents env_events(const Consts<real> &c, bool active, bool in_curr, uint64_t new_pair, uint32_t bits,
    uint32_t &flags, int base,
                                                uint64_t nmask, int i, int tick, int tick_before, const real *myobs) {
    EnvEvents ev;
    const bool was_in_col = active && (flags & F_IN_COL);
    const uint64_t curr_ids = (__ballot(active && in_curr) >> base) & nmask;
    const uint64_t prev_ids = (__ballot(was_in_col) >> base) & nmask;
    ev.obst_hit = (__ballot(active && (bits & B_OBST_HIT)) >> base) & nmask;
    ev.obst_new = (__ballot(active && (bits & B_OBST_NEW)) >> base) & nmask;
    ev.floor = (__ballot(active && (bits & B_FLOOR)) >> base) & nmask;
    ev.wall = (__ballot(active && (bits & B_WALL_NEW)) >> base) & nmask;
    ev.ceil = (__ballot(active && (bits & B_CEIL_NEW)) >> base) & nmask;
    ev.room = (__ballot(active && (bits & B_ROOM_NEW)) >> base) & nmask;
    ev.wave_newpair = __ballot(active && new_pair != 0);
    ev.newpair_any = (ev.wave_newpair >> base) & nmask;
    ev.unique = curr_ids & ~prev_ids;
    if (in_curr) flags |= F_IN_COL; else flags &= ~F_IN_COL;
    ev.col_tick = __popcll(ev.unique) / 2;                                             // :448
    ev.obst_cnt = __popcll(ev.obst_new);
    ev.settled = tick >= c.grace_steps;
    ev.time_remain = c.ep_len - tick_before;
    if (ev.col_tick > 0 && ev.settled && (ev.unique >> i & 1)) flags &= ~F_COL_AGENT_OK;
    if (ev.obst_cnt > 0 && ev.settled && (bits & B_OBST_NEW)) flags &= ~F_COL_OBST_OK;
    ev.n35 = 0; ev.n5 = 0;
    if (c.use_obstacles) {
        const real qrel = M<real>::sqrt(myobs[0] * myobs[0] + myobs[1] * myobs[1] + myobs[2] * myobs[2]);
        ev.n35 = __popcll((__ballot(active && (bits & B_OBST_NEW) && qrel > (real)3.5) >> base) & nmask);
        ev.n5 = __popcll((__ballot(active && (bits & B_OBST_NEW) && qrel > (real)5.0) >> base) & nmask);
    }
    return ev;
}

// collision / proximity / obstacle terms of the reward (:499-546), added to `rew` in the reference's order
template <typename real>
__device__ __forceinline__ void collision_rewards(const Consts<real> &c, const real *rewc, const EnvEvents &ev, int i, uint32_t bits,
    real prox, real &rew, real *ri) {
    const bool any_nonzero_id = (ev.unique & ~1ull) != 0;   // `.any()` of the id array
    real raw = (any_nonzero_id && (ev.unique >> i & 1)) ? (real)-1 : (real)0;
    real rc = rewc[QS_REW_QUADCOL_BIN] * raw;
    real rp = (real)-1 * (c.control_dt * prox);
    rew += rc;
    rew += rp;
    ri[QS_RI_REW_QUADCOL] = rc; ri[QS_RI_REW_PROXIMITY] = rp; ri[QS_RI_RAW_QUADCOL] = raw;
    if (c.use_obstacles) {
        real ro_raw = (ev.obst_hit && (bits & B_OBST_NEW)) ? (real)-1 : (real)0;
        real ro = rewc[QS_REW_QUADCOL_OBST] * ro_raw;
        rew += ro;
        ri[QS_RI_REW_QUADCOL_OBST] = ro; ri[QS_RI_RAW_QUADCOL_OBST] = ro_raw;
    }
}

// distance-to-goal log, reached_goal (:542-546), windowed sums for the episode stats (:649-661).  metric: approach_goal_metric of the
// env's scenario.
//
// Both logs are PRIVATE state (nothing but reached_goal and the three distance_to_goal_* statistics of a finished episode depends on
// them), kept so that their rows in HBM are touched only on the steps that can matter (the throughput kernels skip the loads / stores
// wave-uniformly; the others keep the same representation so that every kernel of a handle reads what any other wrote):
//  * dist_ring (the last 4 distances; reached_goal = mean of 5 < metric * dt, i.e. within millimetres of the goal): a mean of five
//    non-negative values is below x only if every one of them is below 5 x, so a logged value >= 5.5 * metric * dt can never be part of a
// triggering window (the logged value is the raw position cost, control_dt * distance: "near" means within 2.75 * metric metres).  Such
// entries are not maintained: the ring of a drone is valid only while F_RING_LIVE says so (set while the
//    ring holds at least one "near" entry and the goal has not been reached); otherwise every entry stands for "far" (1e30).  The
//    decisions are the reference's exactly - a window that contains a far entry gives false in both forms, with a margin of 10 %.
//  * dist_sums (sums of the last 1 / 3 / 5 s): zero in HBM from the reset until the 5-s window of the episode opens (they used to be
//    stored as zeros on every step before it), accumulated from there, zeroed again by the step that ends the episode.
template <typename real>
__device__ __forceinline__ bool ring_near(const Consts<real> &c, real metric, real dist) { return dist * c.inv_dt < (real)5.5 * metric; }
template <typename real>
__device__ __forceinline__ bool sums_window_open(const Consts<real> &c, int tick) { return tick > c.ep_len + 1 - 5 * c.control_freq; }
// does this drone's ring row have to be read (and written back) on this step?
template <typename real>
__device__ __forceinline__ bool ring_needed(const Consts<real> &c, real metric, uint32_t flags, const real *ri) {
    return !(flags & F_REACHED) && ((flags & F_RING_LIVE) || ring_near<real>(c, metric, -ri[QS_RI_RAW_POS]));
}
// LAZY: the kernel skips the rows' loads / stores on the steps that cannot matter (the throughput kernels) and therefore decides liveness
// from the values.  The kernels that load and store every row anyway (team, multi-step, gated: the latency regime, where ~40 instructions
// on the physics wave's critical path are 0.1 us of an 8 us step - measured: profiles/r04g_ab_tree_vs_r03.txt) keep every entry exact and
// simply mark the ring live while the goal is not reached; both forms read what the other wrote.
template <typename real, bool LAZY>
__device__ __forceinline__ void goal_distance_log(const Consts<real> &c, real metric, int tick, bool done, uint32_t &flags, real ring[4],
    real sums[3], const real *ri, real eps_dist[3]) {
    const real dnow = -ri[QS_RI_RAW_POS];
    // not maintained = far (also: whatever the row holds after a reset)
    if (!(flags & F_RING_LIVE)) { ring[0] = ring[1] = ring[2] = ring[3] = (real)1e30; }
    if (tick >= 5 && !(flags & F_REACHED)) {
        real mean5 = ((((ring[3] + ring[2]) + ring[1]) + ring[0]) + dnow) * (real)0.2;
        if (mean5 * c.inv_dt < metric) flags |= F_REACHED;
    }
    ring[3] = ring[2]; ring[2] = ring[1]; ring[1] = ring[0]; ring[0] = dnow;
    bool live = !(flags & F_REACHED);
    if (LAZY) live = live && (ring_near<real>(c, metric, ring[0]) | ring_near<real>(c, metric, ring[1]) | ring_near<real>(c, metric,
        ring[2]) | ring_near<real>(c, metric, ring[3]));
    flags = live ? (flags | F_RING_LIVE) : (flags & ~F_RING_LIVE);
    const int total = c.ep_len + 1;
#pragma unroll
    for (int w = 0; w < 3; ++w) {
        const int win = (w == 0 ? 1 : (w == 1 ? 3 : 5)) * c.control_freq;
        // (not LAZY: the row holds zeros outside the window anyway)
        real sum = (tick == 1 || (LAZY && !sums_window_open<real>(c, tick))) ? (real)0 : sums[w];
        if (tick > total - win) sum += dnow;
        eps_dist[w] = c.inv_dt * (sum * c.inv_win[w]);
        sums[w] = done ? (real)0 : sum;
    }
}

// per-env counters (:448-497), kept by lane 0 of the env
template <typename real>
__device__ __forceinline__ void env_counters_add(const Consts<real> &c, int32_t *cnt, const EnvEvents &ev) {
    cnt[QS_CNT_COLLISIONS] += ev.col_tick;
    if (ev.col_tick > 0 && ev.settled) cnt[QS_CNT_COLLISIONS_AFTER_SETTLE] += ev.col_tick;
    if (ev.col_tick > 0 && ev.time_remain <= c.final_steps) cnt[QS_CNT_COLLISIONS_FINAL_5S] += ev.col_tick;
    cnt[QS_CNT_OBST] += ev.obst_cnt;
    if (ev.obst_cnt > 0 && ev.settled) { cnt[QS_CNT_OBST_AFTER_SETTLE] += ev.obst_cnt; cnt[QS_CNT_OBST_DIST_3_5] += ev.n35;
        cnt[QS_CNT_OBST_DIST_5] += ev.n5; }
    if (ev.settled) {
        cnt[QS_CNT_ROOM] += __popcll(ev.room); cnt[QS_CNT_FLOOR] += __popcll(ev.floor);
        cnt[QS_CNT_WALL] += __popcll(ev.wall); cnt[QS_CNT_CEILING] += __popcll(ev.ceil);
    }
}

// ------------------------------------------------------------------------------------------------
// physical interactions, in the reference's order (:548-587)
// ------------------------------------------------------------------------------------------------
// 1) downwash aerodynamics/downwash.py:4-66: this lane is the LOWER drone, ii the upper one
template <typename real>
__device__ __forceinline__ void downwash_apply(const Consts<real> &c, Drone<real> &d, const real *s_pos, const real *s_zax, int B,
    int base, int ii, real ua, real uw,
                                               const real vn[3], const real dirw[3], uint32_t &bits) {
    real rel[3] = {d.pos[0] - s_pos[0 * B + base + ii], d.pos[1] - s_pos[1 * B + base + ii], d.pos[2] - s_pos[2 * B + base + ii]};
    real zx[3] = {s_zax[0 * B + base + ii], s_zax[1 * B + base + ii], s_zax[2 * B + base + ii]};
    real dist = norm3<real>(rel);
    real a = M<real>::fmax((real)1e-6, (real)(6.0 / 17.0) * ((real)-10 * dist + (real)7) + ua);
    real ow = M<real>::fmax((real)1e-6, (real)0.3 * ((dist - (real)1) * (dist - (real)1)) + uw);
    real nz[3];
#pragma unroll
    for (int q = 0; q < 3; ++q) nz[q] = zx[q] + vn[q];
    real mz = norm3<real>(nz), dz = (mz == (real)0) ? mz + (real)1e-6 : mz;
    real mw = norm3<real>(dirw), dwn = (mw == (real)0) ? mw + (real)1e-6 : mw;
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        real down = (real)-1 * (nz[q] / dz);
        d.vel[q] += a * down * c.control_dt;
        d.omega[q] += ow * (dirw[q] / dwn) * c.control_dt;
    }
    bits |= B_DOWNWASH;
}

// all upper drones of this lane (dw_mask: bit ii = drone ii hovers above me), ascending ii = the reference's outer-loop order.
// s_cur / tape_env: the tape flavour's cursor slot of this env and its tape (unused otherwise).
template <typename real, typename Sync>
__device__ __forceinline__ void downwash_phase(const Consts<real> &c, const RngKey &key, Drone<real> &d, const real *s_pos,
    const real *s_zax, int B, int base, uint64_t nmask,
                                               int N, int i, bool active, uint64_t dw_mask, uint32_t &bits, int *s_cur_env,
                                                   const double *tape_env, Sync sync) {
#ifdef QS_TAPE
    if (c.use_downwash && QS_ON_TAPE(key)) {
        // downwash.py:30-62 on a tape: two draws per upper drone ii (always), then six per affected lower drone in ascending order:
        // every lane can compute where its own draws sit
        int off = *s_cur_env;
        for (int ii = 0; ii < N; ++ii) {
            const bool hit = active && ((dw_mask >> ii) & 1);
            const uint64_t aff = (__ballot(hit) >> base) & nmask;
            if (hit) {
                const double *tp = tape_env + off + 2 + 6 * __popcll(aff & ((1ull << i) - 1));
                const real vn[3] = {(real)tp[0], (real)tp[1], (real)tp[2]}, dirw[3] = {(real)tp[3], (real)tp[4], (real)tp[5]};
                downwash_apply<real>(c, d, s_pos, s_zax, B, base, ii, (real)tape_env[off], (real)tape_env[off + 1], vn, dirw, bits);
            }
            off += 2 + 6 * __popcll(aff);
        }
        sync();
        if (active && i == 0) *s_cur_env = off;
        sync();
        return;
    }
#else
    (void)nmask; (void)N; (void)active; (void)s_cur_env; (void)tape_env; (void)sync;
#endif
    if (__builtin_expect(c.use_downwash && dw_mask, 0)) {   // rare
        uint64_t mm = dw_mask;
        while (mm) {
            const int ii = __ffsll((long long)mm) - 1;
            mm &= mm - 1;
            uint32_t w[4];
            rng_words(key, QS_SITE_DW_I, 0, ii, 0, w);
            real ua = (real)-0.1 + (real)0.2 * u01<real>(w[0]), uw = (real)-0.01 + (real)0.02 * u01<real>(w[1]);
            real vn[3], dirw[3];
            rng_words(key, QS_SITE_DW_IJ_V, 0, ii, i, w);
#pragma unroll
            for (int q = 0; q < 3; ++q) vn[q] = (real)-0.1 + (real)0.2 * u01<real>(w[q]);
            rng_words(key, QS_SITE_DW_IJ_W, 0, ii, i, w);
#pragma unroll
            for (int q = 0; q < 3; ++q) dirw[q] = (real)-1 + (real)2 * u01<real>(w[q]);
            downwash_apply<real>(c, d, s_pos, s_zax, B, base, ii, ua, uw, vn, dirw, bits);
        }
    }
}

// 2) drone-drone responses for the NEW pairs in lexicographic order (collisions/quadrotors.py:24-59); order-dependent and rare: one
//    lane per env walks the pair list on LDS-resident vel / omega.  s_mask: one uint64 per lane of LDS scratch.
//    Parallel form (tbl != nullptr: the team kernels; not on a tape, which holds the draws in the serial order): the wave's pairs are
//    enumerated in that order by scalar code (v_readlane over the lanes that own one), QS_DD_CHUNK at a time; lane 11 p + k draws Philox
//    block k of pair p into the table (collide_drones_draw), then lane 11 p applies the response - pairs of different environments at the
//    same time, pairs of one environment one after the other in list order (its r-th pair of the chunk in round r).  Same draws, same
//    arithmetic, same order as the serial form; measured on the C4 shard 12.8 -> 11.x us per step (the slowest workgroup of a step is the
//    one with the most collisions: profiles/r06s_exp_rare_paths.txt, r06r_phase_c4.txt).
template <typename real, typename Sync>
__device__ __forceinline__ void pair_responses(const RngKey &key, Drone<real> &d, const EnvEvents &ev, uint64_t new_pair, bool active,
    int N, int i, int tid, int base, int B,
                                               const real *s_pos, real *s_vel, real *s_om, uint64_t *s_mask, int *s_cur_env, Sync sync,
                                               real *tbl = nullptr) {
    if (__builtin_expect(ev.wave_newpair != 0, 0)) {
#pragma unroll
        for (int q = 0; q < 3; ++q) { s_vel[q * B + tid] = d.vel[q]; s_om[q * B + tid] = d.omega[q]; }
        if (tbl != nullptr && !QS_ON_TAPE(key)) {
            const int lane = tid & (QS_WAVE - 1), grp = lane / QS_DD_CALLS, call = lane - grp * QS_DD_CALLS;
            const uint32_t np_lo = (uint32_t)new_pair, np_hi = (uint32_t)(new_pair >> 32);
            uint64_t owners = ev.wave_newpair, np = 0;   // lanes that own new pairs, still to be enumerated; rest of the current owner's mask
            int own = 0;
            sync();
            while (owners != 0 || np != 0) {
                int cnt = 0, last_base = -1, rank = 0, rounds = 0;
                int ta = 0, tb = 0, tbase = 0, trank = 0;
                uint32_t tenv = 0, tstep = 0;
                while (cnt < QS_DD_CHUNK && (owners != 0 || np != 0)) {   // wave-uniform: the next pair in (env, a, b) order
                    if (np == 0) {
                        own = __ffsll((long long)owners) - 1;
                        owners &= owners - 1;
                        np = (uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)np_lo, own) |
                             ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)np_hi, own) << 32);   // (uint32_t): no sign extension of bit 31
                    }
                    const int b = __ffsll((long long)np) - 1;
                    np &= np - 1;
                    const int obase = __builtin_amdgcn_readlane(base, own);
                    const uint32_t oenv = (uint32_t)__builtin_amdgcn_readlane((int)key.env, own), ostep = (uint32_t)__builtin_amdgcn_readlane((int)key.step, own);
                    rank = (obase == last_base) ? rank + 1 : 0;
                    last_base = obase;
                    rounds = rank + 1 > rounds ? rank + 1 : rounds;
                    if (grp == cnt) { ta = own - obase; tb = b; tbase = obase; trank = rank; tenv = oenv; tstep = ostep; }
                    ++cnt;
                }
                const bool mine = grp < cnt;
                if (mine) {
                    const RngKey kt = {key.k0, key.k1, tenv, tstep};
                    collide_drones_draw<real>(kt, call, ta, tb, tbl + grp * QS_DD_DRAWS);
                }
                sync();
                for (int r = 0; r < rounds; ++r) {
                    if (mine && call == 0 && trank == r) collide_drones_apply<real>(tbl + grp * QS_DD_DRAWS, ta, tb, tbase, B, s_pos, s_vel, s_om);
                    sync();
                }
            }
            if (ev.newpair_any) {
#pragma unroll
                for (int q = 0; q < 3; ++q) { d.vel[q] = s_vel[q * B + tid]; d.omega[q] = s_om[q * B + tid]; }
            }
            return;
        }
        s_mask[tid] = new_pair;
        sync();
        if (active && ev.newpair_any && i == 0) {
            tape_cursor_in(key, s_cur_env);
            for (int a = 0; a < N; ++a) {
                uint64_t np = s_mask[base + a];
                while (np) {
                    int b = __ffsll((long long)np) - 1;
                    np &= np - 1;
                    collide_drones_lds<real>(key, a, b, base, B, s_pos, s_vel, s_om);
                }
            }
            tape_cursor_out(key, s_cur_env);
        }
        sync();
        if (ev.newpair_any) {
#pragma unroll
            for (int q = 0; q < 3; ++q) { d.vel[q] = s_vel[q * B + tid]; d.omega[q] = s_om[q * B + tid]; }
        }
    }
}

// 3) obstacle response, 4) wall then ceiling (:565-587).  s_ox / s_oy: the env's obstacle positions in LDS; obst_size_env: the env's
// obstacle size of the running episode.
template <typename real, typename Sync>
__device__ __forceinline__ void room_obstacle_phase(const Consts<real> *cp, const RngKey &key, Drone<real> &d, bool active, int N, int i,
    uint32_t bits, int obst_idx,
                                                    const real *s_ox, const real *s_oy, const real *obst_size_env, int *s_cur_env,
                                                        Sync sync) {
    const Consts<real> &c = *cp;
#define QS_SEM_OSIZE (c.dr_on ? *obst_size_env : c.obst_size)   /* --quads_domain_random: the running episode's obstacle size */
#ifdef QS_TAPE
    if (QS_ON_TAPE(key)) {   // the reference: all obstacle responses in drone order, then all wall responses, then all ceiling responses
        for (int pass = 0; pass < 3; ++pass) {
            const uint32_t bit = pass == 0 ? B_OBST_NEW : (pass == 1 ? B_WALL_NEW : B_CEIL_NEW);
            if (__ballot(active && (bits & bit)) == 0) continue;
            for (int turn = 0; turn < N; ++turn) {
                sync();
                if (active && i == turn && (bits & bit)) {
                    real ox = 0, oy = 0;
                    if (bit == B_OBST_NEW) { ox = s_ox[obst_idx]; oy = s_oy[obst_idx]; }
                    tape_cursor_in(key, s_cur_env);
                    room_obst_responses<real>(cp, key, i, bit, ox, oy, d.pos, d.vel, d.omega, QS_SEM_OSIZE);
                    tape_cursor_out(key, s_cur_env);
                }
            }
            sync();
        }
        return;
    }
#else
    (void)N; (void)s_cur_env; (void)sync;
#endif
    if (__builtin_expect(active && (bits & (B_OBST_NEW | B_WALL_NEW | B_CEIL_NEW)), 0)) {   // out of line: rare
        real ox = 0, oy = 0;
        if (bits & B_OBST_NEW) { ox = s_ox[obst_idx]; oy = s_oy[obst_idx]; }
        room_obst_responses<real>(cp, key, i, bits, ox, oy, d.pos, d.vel, d.omega, QS_SEM_OSIZE);
    }
#undef QS_SEM_OSIZE
}

// ------------------------------------------------------------------------------------------------
// scenario.step() (scenarios/*.py `step`), after the interactions
// ------------------------------------------------------------------------------------------------
// the full scenario set (qs_scenarios.h); x: the env's scenario context (state in LDS), scr: N ints of LDS scratch of the env
template <typename real, typename Sync>
__device__ __forceinline__ void scenario_phase_full(const Consts<real> &c, const RngKey &key, const ScenCtx<real> &x, bool active, int N,
    int i, int tick, real goal[3], int *scr,
                                                    int *s_cur_env, Sync sync) {
    const int sc = active ? x.si[SI_SCEN] : 0, period = active ? x.si[SI_PERIOD] : 0;
    const bool serial = active && scen_step_serial_needed(sc, period, tick);
    if (__builtin_expect(__ballot(serial) != 0, 0)) {
        if (active) {   // publish the current goals (swap_goals permutes them)
#pragma unroll
            for (int q = 0; q < 3; ++q) x.goals[i * 3 + q] = goal[q];
        }
        sync();
#ifdef QS_TAPE
        const bool wv = false;   // (a tape holds the draws in the serial order)
        (void)scr;
#else
        const bool wv = serial && scen_step_wave_ok(sc, N);   // the env's lanes build the rows (scenario_step_wave, qs_scenarios.h)
        if (__ballot(wv) != 0) scenario_step_wave<real>(c, key, x, sc, scr, i, wv);
#endif
        if (serial && !wv && i == 0) {
            tape_cursor_in(key, s_cur_env);
            scenario_step_serial<real>(c, key, x, sc);
            tape_cursor_out(key, s_cur_env);
        }
        sync();
        if (serial) {
#pragma unroll
            for (int q = 0; q < 3; ++q) goal[q] = x.goals[i * 3 + q];
        }
    }
#ifdef QS_TAPE
    if (QS_ON_TAPE(key)) {   // the lane-local part draws the same values in every lane of the env: every lane reads the same tape positions
        sync();
        if (active) *key.cur = *s_cur_env;
    }
#endif
    if (active) scenario_step_local<real>(c, key, x, sc, period, tick, i == 0, goal);
#ifdef QS_TAPE
    if (QS_ON_TAPE(key)) {
        sync();
        if (active && i == 0) *s_cur_env = *key.cur;
        sync();
    }
#endif
}

// the fast kernels' only stepping scenario: swarm_vs_swarm swaps the two formations every U(4,6) s (swarm_vs_swarm.py:59-79).
// env_goals: the env's goal rows in LDS (>= 2N + 6 rows); the centres live in scen_real[0..5][e].
template <typename real, typename Sync>
// returns (wave-uniform) whether any environment of the wave got new goals on this step
__device__ __forceinline__ bool svs_phase(const Consts<real> &c, const Ptrs<real> &p, const RngKey &key, bool active, int N, int E,
    int e, int i, int tick, int svs_period,
                                          real *env_goals, int *scr, real goal[3], int *s_cur_env, Sync sync) {
    if (c.scenario != QS_SCENARIO_SWARM_VS_SWARM) return false;
    const bool sw = active && svs_period > 0 && tick % svs_period == 0 && tick > 0;
    const bool any_sw = __ballot(sw) != 0;
    if (__builtin_expect(any_sw, 0)) {
#ifndef QS_TAPE
        if (N / 2 >= 3) {   // the wave builds the two formations: lane i makes goal row i (svs_create_formations_wave, qs_device.h)
            real c1[3] = {0, 0, 0}, c2[3] = {0, 0, 0};
            if (sw) { for (int q = 0; q < 3; ++q) { c1[q] = p.scen_real[(3 + q) * E + e]; c2[q] = p.scen_real[q * E + e]; } }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // every lane holds the old centres before lane 0 swaps them
            if (sw && i == 0) { for (int q = 0; q < 3; ++q) { p.scen_real[q * E + e] = c1[q]; p.scen_real[(3 + q) * E + e] = c2[q]; } }
            Formation<real> F;
            RngKey kc = key;                      // the out-of-line callee takes the key by reference: a copy that lives in this cold
            asm volatile("" : "+v"(kc.step));     // block only (otherwise `key` is kept in scratch memory on every step)
            update_formation<real>(c.scenario, kc, 32, N, F);
            svs_create_formations_wave<real>(kc, F, N, c.cube_fd[0], c.cube_fd[1], c1, c2, true, env_goals, scr, i, sw);
        } else
#else
        (void)scr;
#endif
        if (sw && i == 0) {
            tape_cursor_in(key, s_cur_env);
            real c1[3], c2[3];
            for (int q = 0; q < 3; ++q) { c1[q] = p.scen_real[(3 + q) * E + e]; c2[q] = p.scen_real[q * E + e]; }
            for (int q = 0; q < 3; ++q) { p.scen_real[q * E + e] = c1[q]; p.scen_real[(3 + q) * E + e] = c2[q]; }
            Formation<real> F;
#ifdef QS_TAPE
            update_formation<real>(c.scenario, key, 32, N, F);
            svs_create_formations<real>(key, F, N, QS_CUBE_FD(c), c1, c2, true, env_goals);
#else
            RngKey kc = key;
            asm volatile("" : "+v"(kc.step));
            update_formation<real>(c.scenario, kc, 32, N, F);
            svs_create_formations<real>(kc, F, N, QS_CUBE_FD(c), c1, c2, true, env_goals);
#endif
            tape_cursor_out(key, s_cur_env);
        }
        sync();
        if (sw) {
#pragma unroll
            for (int q = 0; q < 3; ++q) goal[q] = env_goals[i * 3 + q];
        }
    }
    return any_sw;
}

}  // namespace qs
