// quadswarm_hip.hip - MI355X (gfx950) QuadSwarm environment stepper: kernels + C ABI (include/quadswarm.h).
//
// Mapping (wave64): one lane = one drone, one workgroup = one wavefront = floor(64/N) whole environments,
// so every cross-drone exchange of an environment (pair scan, neighbour selection, downwash, collision
// responses) goes through LDS inside one wave and needs no inter-workgroup traffic.  State is
// struct-of-arrays in HBM (component-major), so lane l of a wave reads word l of each component row:
// fully coalesced.  Observations are staged in LDS and copied out as one contiguous block per workgroup.
//
// Reference path (gym_art/quadrotor_multi/): quadrotor_multi.py:413-724 (step), :339-411 (reset); the
// per-piece citations are in qs_device.h and next to each phase below.  SURVEY.md Appendix A gives the
// order of operations that the step kernel follows.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "qs_device.h"
#include "qs_scenarios.h"

using namespace qs;

#define QS_WAVE 64

// ------------------------------------------------------------------------------------------------
// device buffers
// ------------------------------------------------------------------------------------------------
template <typename real> struct Ptrs {
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
    uint8_t *reset_mask;   // [E] nonzero => reset kernel re-initialises this env
    unsigned long long *timing;   // [32] phase time stamps of workgroup 0 (only written by -DQS_TIMING builds)
};

#ifdef QS_TIMING
#define QS_STAMP(k) do { if (blockIdx.x == 0 && threadIdx.x == 0) p.timing[k] = clock64(); } while (0)
#else
#define QS_STAMP(k) do { } while (0)
#endif

struct LdsLayout { int off_mask, off_omap, off_si, off_sr, off_envflag, off_scratch, off_pos, off_vel, off_zax, off_om, off_goal, off_obst, off_metric, off_obs, goal_rows, total; };
#define QS_RESET_SCRATCH_INTS 160   // per env: virtual-pool index/value lists (2x64) + two DP rows (2x16)

static LdsLayout lds_layout(int real_size, int B, int N, int epb, int obs_dim, int num_obst, int K) {
    LdsLayout L;
    int o = 0;
    L.off_mask = o; o += 8 * B;                       // u64 per lane: new-pair masks for the serial response path
    L.off_omap = o; o += 8 * 4 * epb;                 // full-scenario kernels: obstacle map bitset, scenario ints / reals per env
    L.off_si = o; o += 4 * SI_COUNT * epb;
    L.off_sr = o; o += real_size * SR_COUNT * epb;
    o = (o + 15) & ~15;
    L.off_envflag = o; o += 4 * ((2 * epb + 3) & ~3);   // [epb] have-spawn-points flags + [epb] swarm_vs_swarm periods
    L.off_scratch = o; o += 4 * QS_RESET_SCRATCH_INTS * epb;
    o = (o + 15) & ~15;
    L.off_pos = o; o += real_size * 3 * B;
    L.off_vel = o; o += real_size * 3 * B;
    L.off_zax = o; o += real_size * 3 * B;            // body z axes (downwash); spawn points in the reset tail
    L.off_om = o; o += real_size * 3 * B;
    L.goal_rows = 2 * N + 8;
    L.off_goal = o; o += real_size * 3 * L.goal_rows * epb;
    L.off_obst = o; o += real_size * 2 * (num_obst > 0 ? num_obst : 1) * epb;   // obstacle xy of the block's envs
    L.off_metric = o; o += (K > 8 && K < N - 1) ? real_size * N * B : 0;         // neighbour metric rows, [N][B]
    o = (o + 15) & ~15;
    L.off_obs = o; o += real_size * obs_dim * B;      // observation staging, row-major, contiguous
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
__device__ __forceinline__ void neighbor_obs(const Consts<real> &c, int N, int i, int base, int B, int tid, const real *s_pos, const real *s_vel,
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
                for (int a = 0; a < 3; ++a) { rp[u][a] = s_pos[a * B + base + j] - mypos[a]; rv[u][a] = s_vel[a * B + base + j] - myvel[a]; }
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
                    for (int u = 0; u < 8; ++u) rank[u] += (mj[k] < mj[u] || (mj[k] == mj[u] && k < u)) ? 1 : 0;
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
                for (int a = 0; a < 3; ++a) { rp[u][a] = s_pos[a * B + base + j] - mypos[a]; rv[u][a] = s_vel[a * B + base + j] - myvel[a]; }
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
                for (int a = 0; a < 3; ++a) { vals[k][a] = s_pos[a * B + base + bi[k]] - mypos[a]; vals[k][3 + a] = s_vel[a * B + base + bi[k]] - myvel[a]; }
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

// get_surround_sdfs obstacles/utils.py:5-27 (obstacle xy of the env in LDS)
template <typename real>
__device__ __forceinline__ void sdf_obs(const Consts<real> &c, const real *ox, const real *oy, int M_, real px, real py, real *o) {
    const real res = (real)0.1;
    real gx[3] = {px - res, px, px + res}, gy[3] = {py - res, py, py + res};
    real mind[9];
#pragma unroll
    for (int q = 0; q < 9; ++q) mind[q] = (real)100;
    for (int k = 0; k < M_; ++k) {
        real x = ox[k], y = oy[k];
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int b = 0; b < 3; ++b) {
                real dx = gx[a] - x, dy = gy[b] - y, dist = M<real>::sqrt(dx * dx + dy * dy);
                mind[a * 3 + b] = dist < mind[a * 3 + b] ? dist : mind[a * 3 + b];
            }
    }
#pragma unroll
    for (int q = 0; q < 9; ++q) o[q] = mind[q] - c.obst_radius;
}

// perform_collision_between_drones collisions/quadrotors.py:24-59 on LDS-resident vel/omega (serial per env)
template <typename real>
__device__ __forceinline__ void collide_drones_lds(const RngKey &key, int i, int j, int base, int B, const real *s_pos, real *s_vel, real *s_om) {
    real p1[3], p2[3], v1[3], v2[3];
    for (int q = 0; q < 3; ++q) { p1[q] = s_pos[q * B + base + i]; p2[q] = s_pos[q * B + base + j]; v1[q] = s_vel[q * B + base + i]; v2[q] = s_vel[q * B + base + j]; }
    real n[3] = {p1[0] - p2[0], p1[1] - p2[1], p1[2] - p2[2]};
    real mag = norm3<real>(n), den = (mag == (real)0) ? mag + (real)1e-5 : mag;
    for (int q = 0; q < 3; ++q) n[q] /= den;
    real v1n = dot3<real>(v1, n), v2n = dot3<real>(v2, n);
    real vc[3] = {(v2n - v1n) * n[0], (v2n - v1n) * n[1], (v2n - v1n) * n[2]};
    real s1[3] = {vc[0], vc[1], vc[2]}, s2[3] = {-vc[0], -vc[1], -vc[2]};
    for (int t = 0; t < 3; ++t) {
        real cons[3], n1[3], n2[3], t1[3], t2[3];
        rng_normal<real, 3>(key, QS_SITE_DD_N, t * 3 + 0, i, j, cons);
        rng_normal<real, 3>(key, QS_SITE_DD_N, t * 3 + 1, i, j, n1);
        rng_normal<real, 3>(key, QS_SITE_DD_N, t * 3 + 2, i, j, n2);
        for (int q = 0; q < 3; ++q) {
            real a = (real)0.8 * cons[q] + (real)0.15 * n1[q], b = -((real)0.8 * cons[q]) + (real)0.15 * n2[q];
            s1[q] = vc[q] + a; s2[q] = -vc[q] + b;
            t1[q] = v1[q] + s1[q]; t2[q] = v2[q] + s2[q];
        }
        if (dot3<real>(t1, n) > (real)0 && (real)0 > dot3<real>(t2, n)) break;
    }
    real maxv = M<real>::fmax(norm3<real>(v1), norm3<real>(v2));
    real dec[2]; rng_uniform<real, 2>(key, QS_SITE_DD_U, 0, i, j, (real)0.2, (real)0.8, dec);
    compute_new_vel<real>(maxv, v1, s1, dec[0]);
    compute_new_vel<real>(maxv, v2, s2, dec[1]);
    uint32_t w[4]; rng_words(key, QS_SITE_DD_W, 0, i, j, w);
    real u[4] = {(real)-1 + (real)2 * u01<real>(w[0]), (real)-1 + (real)2 * u01<real>(w[1]), (real)-1 + (real)2 * u01<real>(w[2]),
                 (real)(10.0 * QS_PI_D) + (real)(20.0 * QS_PI_D - 10.0 * QS_PI_D) * u01<real>(w[3])};
    real dw[3]; compute_new_omega<real>(u, dw);
    for (int q = 0; q < 3; ++q) {
        s_vel[q * B + base + i] = v1[q]; s_vel[q * B + base + j] = v2[q];
        s_om[q * B + base + i] += dw[q]; s_om[q * B + base + j] -= dw[q];
    }
}

// rare per-drone responses kept out of line so the hot path stays compact
template <typename real>
__device__ __forceinline__ void room_obst_responses(const Consts<real> *cp, const RngKey &key, int i, uint32_t bits, real ox, real oy, real pos[3], real vel[3], real omega[3]) {
    const Consts<real> &c = *cp;
    Drone<real> d;
#pragma unroll
    for (int q = 0; q < 3; ++q) { d.pos[q] = pos[q]; d.vel[q] = vel[q]; d.omega[q] = omega[q]; }
    if (bits & B_OBST_NEW) collide_obstacle<real>(c, key, i, d, ox, oy);      // collisions/obstacles.py:23-50
    if (bits & B_WALL_NEW) collide_room<real>(c, key, i, d, true);            // collisions/room.py:6-44
    if (bits & B_CEIL_NEW) collide_room<real>(c, key, i, d, false);           // collisions/room.py:91-113
#pragma unroll
    for (int q = 0; q < 3; ++q) { vel[q] = d.vel[q]; omega[q] = d.omega[q]; }
}

// full-scenario kernels: per-env scenario state HBM <-> LDS (each drone of the env moves a strided part)
template <typename real>
__device__ __forceinline__ void scen_lds_load(const Ptrs<real> &p, const LdsLayout &L, unsigned char *smem, int E, int e, int le, int i, int N) {
    real *sr = (real *)(smem + L.off_sr) + le * SR_COUNT;
    int *si = (int *)(smem + L.off_si) + le * SI_COUNT;
    uint64_t *om = (uint64_t *)(smem + L.off_omap) + le * 4;
    for (int k = i; k < SR_COUNT; k += N) sr[k] = p.scen_real[k * E + e];
    for (int k = i; k < SI_COUNT; k += N) si[k] = p.scen_int[k * E + e];
    for (int k = i; k < 4; k += N) om[k] = p.scen_omap[k * E + e];
}
template <typename real>
__device__ __forceinline__ void scen_lds_store(const Ptrs<real> &p, const LdsLayout &L, unsigned char *smem, int E, int e, int le, int i, int N) {
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
template <typename real, bool FULL>
__device__ __forceinline__ void reset_body(const Consts<real> *cp, const Ptrs<real> *pp, const LdsLayout *Lp, unsigned char *smem, int epb, const RngKey &key,
                                        bool do_reset, Drone<real> *dp, real goal[3], const real stale_vel[3]) {
    const Consts<real> &c = *cp;
    const Ptrs<real> &p = *pp;
    const LdsLayout &L = *Lp;
    Drone<real> &d = *dp;
    const int B = QS_WAVE, N = c.num_agents, E = c.num_envs;
    real *s_pos = (real *)(smem + L.off_pos), *s_vel = (real *)(smem + L.off_vel), *s_spawn = (real *)(smem + L.off_zax);
    real *s_goal = (real *)(smem + L.off_goal), *s_obs = (real *)(smem + L.off_obs), *s_obst = (real *)(smem + L.off_obst);
    real *s_metric = (real *)(smem + L.off_metric);
    uint32_t *s_envflag = (uint32_t *)(smem + L.off_envflag);
    const int tid = threadIdx.x, le = tid / N, i = tid - le * N, e = blockIdx.x * epb + le, base = le * N;
    const int M_ = c.num_obstacles;
    real *myobs = s_obs + tid * c.obs_dim;
    int *tidx = (int *)(smem + L.off_scratch) + le * QS_RESET_SCRATCH_INTS, *tval = tidx + 64, *prev_row = tidx + 128, *cur_row = tidx + 144;

    // ---- per-env part (one lane): obstacle map + scenario.reset() ----
    if (do_reset && i == 0) {
        real *goals = s_goal + le * L.goal_rows * 3;
        uint32_t have_spawn = 0;
        uint64_t omap[4] = {0, 0, 0, 0};   // obstacle map bitset, cell id = rid*W + cid
        const int Lr = c.obst_area[0], W = c.obst_area[1], cells = Lr * W;
        if (FULL) for (int q = 0; q < 4; ++q) ((uint64_t *)(smem + L.off_omap))[le * 4 + q] = 0;
        if (c.use_obstacles) {
            // np.random.choice(cells, M, replace=False): partial Fisher-Yates on a virtual pool
            int nt = 0;
            for (int k = 0; k < M_; ++k) {
                int j = k + rng_index<real>(key, QS_SITE_OBST_MAP, k, cells - k);
                int vk = k, vj = j, pj = -1;
                for (int q = 0; q < nt; ++q) { if (tidx[q] == k) vk = tval[q]; if (tidx[q] == j) { vj = tval[q]; pj = q; } }
                if (pj >= 0) tval[pj] = vk; else { tidx[nt] = j; tval[nt] = vk; ++nt; }   // pool[j] = pool[k]; the pick is old pool[j]
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
            generate_goals<real>(F, N, 1, center, goals, 3);
        } else if (c.scenario == QS_SCENARIO_O_STATIC_SAME_GOAL) {
            // obstacles/o_static_same_goal.py:27-48 + o_base.py:69-81,:124-153
            int nfree = cells - M_;
            int nt = 0;
            for (int k = 0; k < N; ++k) {
                int j = k + rng_index<real>(key, QS_SITE_SCEN, 16 + k, nfree - k);
                int vk = k, vj = j, pj = -1;
                for (int q = 0; q < nt; ++q) { if (tidx[q] == k) vk = tval[q]; if (tidx[q] == j) { vj = tval[q]; pj = q; } }
                if (pj >= 0) tval[pj] = vk; else { tidx[nt] = j; tval[nt] = vk; ++nt; }
                int seen = 0, cell = 0;   // vj-th free cell in row-major order (np.where(obst_map == 0))
                for (int id = 0; id < cells; ++id) if (!(omap[id >> 6] >> (id & 63) & 1)) { if (seen == vj) { cell = id; break; } ++seen; }
                int x = cell / W, y = cell - x * W, index = x + Lr * y, ii = index / W, jj = (W - 1) - (index - ii * W);
                s_spawn[0 * B + base + k] = (real)ii + (real)0.5 - (real)(Lr / 2);
                s_spawn[1 * B + base + k] = (real)jj + (real)0.5 - (real)(W / 2);
                s_spawn[2 * B + base + k] = rng_uniform1<real>(key, QS_SITE_SCEN, 96 + k, 0, 0, (real)1, (real)3);
            }
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
            else if (f == 5 || f == 6) { int rn = N < F.per_layer ? N : F.per_layer, d1, d2; grid_dim(rn, &d1, &d2); zlb = (real)d1 * F.size + (real)0.25; }
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
            svs_create_formations<real>(key, F, N, c.cube_fd, c1, c2, false, goals);
        }
        s_envflag[le] = have_spawn;
    }
    __syncthreads();

    if (do_reset) {
        // ---- per-drone part: QuadrotorSingle._reset quadrotor_single.py:387-447 ----
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
    }
    __syncthreads();
    if (do_reset) {
        neighbor_obs<real>(c, N, i, base, B, tid, s_pos, s_vel, s_metric, d.pos, stale_vel, myobs + c.self_dim);
        if (c.use_obstacles)
            sdf_obs<real>(c, s_obst + (le * 2 + 0) * M_, s_obst + (le * 2 + 1) * M_, M_, d.pos[0], d.pos[1], myobs + c.self_dim + 6 * c.num_neighbors);
    }
}

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

// ------------------------------------------------------------------------------------------------
// reset kernel (qs_reset): resets the envs flagged in reset_mask
// ------------------------------------------------------------------------------------------------
template <typename real, bool FULL>
__global__ void __launch_bounds__(QS_WAVE) qs_reset_kernel(const Consts<real> c, Ptrs<real> p, LdsLayout L, int epb) {
    extern __shared__ __align__(16) unsigned char smem[];
    const Consts<real> *cp = &c;
    const int N = c.num_agents, E = c.num_envs, T = E * N;
    real *s_obs = (real *)(smem + L.off_obs);
    const int tid = threadIdx.x, le = tid / N, i = tid - le * N, e = blockIdx.x * epb + le;
    const bool in_range = (le < epb) && (e < E);
    const bool do_reset = in_range && p.reset_mask[e] != 0;
    const int g = in_range ? e * N + i : 0;
    RngKey key = {c.seed_lo, c.seed_hi, (uint32_t)(c.env_id_offset + (in_range ? e : 0)), in_range ? p.step_ctr[e] : 0u};
    Drone<real> d;
    real goal[3] = {0, 0, 0}, stale_vel[3];
#pragma unroll
    for (int q = 0; q < 3; ++q) stale_vel[q] = p.vel[q * T + g];
    d.flags = p.flags[g];
    if (FULL && in_range) scen_lds_load<real>(p, L, smem, E, e, le, i, N);
    if (FULL) __syncthreads();
    reset_body<real, FULL>(cp, &p, &L, smem, epb, key, do_reset, &d, goal, stale_vel);
    if (FULL) { __syncthreads(); if (do_reset) scen_lds_store<real>(p, L, smem, E, e, le, i, N); }
    if (do_reset) {
#pragma unroll
        for (int q = 0; q < 3; ++q) { p.pos[q * T + g] = d.pos[q]; p.vel[q * T + g] = 0; p.omega[q * T + g] = 0; p.goal[q * T + g] = goal[q]; }
#pragma unroll
        for (int q = 0; q < 9; ++q) p.rot[q * T + g] = d.rot[q];
#pragma unroll
        for (int q = 0; q < 4; ++q) { p.rot_damp[q * T + g] = 0; p.cmds_damp[q * T + g] = 0; p.dist_ring[q * T + g] = 0; }
        p.flags[g] = d.flags;
        p.pair_mask[g] = 0;
        p.new_pair_mask[g] = 0;
        p.obst_hit_idx[g] = -1;
        const real *myobs = s_obs + tid * c.obs_dim;
        real *dst = p.obs + (size_t)g * c.obs_dim;
        for (int q = 0; q < c.obs_dim; ++q) dst[q] = myobs[q];
        if (i == 0) {
            for (int q = 0; q < QS_CNT_COUNT; ++q) p.counters[q * E + e] = 0;
            p.tick[e] = 0;
            p.unique_col[e] = 0; p.obst_new[e] = 0; p.room_new[e] = 0;
            p.reset_mask[e] = 0;
        }
    }
}

// state get/set for one env (qs_get_state / qs_set_state)
template <typename real>
__global__ void qs_state_kernel(Ptrs<real> p, int E, int N, int env, double *buf, int32_t *tick_io, int set) {
    const int i = threadIdx.x, T = E * N;
    if (i >= N) return;
    const int g = env * N + i;
    double *s = buf + (size_t)i * QS_STATE_STRIDE;
    if (!set) {
        for (int q = 0; q < 3; ++q) { s[q] = p.pos[q * T + g]; s[3 + q] = p.vel[q * T + g]; s[15 + q] = p.omega[q * T + g]; s[32 + q] = p.goal[q * T + g]; }
        for (int q = 0; q < 9; ++q) s[6 + q] = p.rot[q * T + g];
        for (int q = 0; q < 4; ++q) { s[18 + q] = p.rot_damp[q * T + g]; s[22 + q] = p.cmds_damp[q * T + g]; s[26 + q] = p.ou[q * T + g]; }
        uint32_t f = p.flags[g];
        s[30] = (f & F_ON_FLOOR) ? 1.0 : 0.0;
        s[31] = (double)((f & F_SVD_MASK) >> F_SVD_SHIFT);
        if (i == 0) *tick_io = p.tick[env];
    } else {
        for (int q = 0; q < 3; ++q) { p.pos[q * T + g] = (real)s[q]; p.vel[q * T + g] = (real)s[3 + q]; p.omega[q * T + g] = (real)s[15 + q]; p.goal[q * T + g] = (real)s[32 + q]; }
        for (int q = 0; q < 9; ++q) p.rot[q * T + g] = (real)s[6 + q];
        for (int q = 0; q < 4; ++q) { p.rot_damp[q * T + g] = (real)s[18 + q]; p.cmds_damp[q * T + g] = (real)s[22 + q]; p.ou[q * T + g] = (real)s[26 + q]; }
        uint32_t f = p.flags[g] & ~(F_ON_FLOOR | F_SVD_MASK);
        if (s[30] != 0.0) f |= F_ON_FLOOR;
        f |= ((uint32_t)s[31] & 0xffu) << F_SVD_SHIFT;
        p.flags[g] = f;
        if (i == 0 && *tick_io >= 0) p.tick[env] = *tick_io;
    }
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
extern "C" int qs_obs_dim(const qs_config *c);
extern "C" int qs_destroy(struct qs_handle *h);
static thread_local std::string g_last_error;
static int fail(int code, const std::string &msg) { g_last_error = msg; return code; }
#define HIP_TRY(expr) do { hipError_t _e = (expr); if (_e != hipSuccess) return fail(QS_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e)); } while (0)

struct qs_handle {
    qs_config cfg;
    int device = 0;
    int real_size = 4;
    int obs_dim = 0, epb = 1, blocks = 0;
    LdsLayout lds;
    bool full = false;     // scenario outside the fast set => kernels compiled with QS_SCEN_FULL
    Consts<float> kf;    // kernel constants, passed by value in the kernarg segment
    Consts<double> kd;
    Ptrs<float> pf;     // same field layout for float/double: only the pointee type differs
    std::vector<void *> allocs;
    qs_buffers bufs;
    void *d_actions = nullptr;
    double *d_state_buf = nullptr;
    int32_t *d_tick_io = nullptr;
    // cached hipGraph of a K-step rollout (qs_step_many): K identical step-kernel nodes
    hipStream_t cap_stream = nullptr;
    hipGraphExec_t graph_exec = nullptr;
    const void *graph_actions = nullptr;
    int32_t graph_k = 0;
    uint8_t *h_mask = nullptr;   // pinned staging for qs_reset masks
    // profiling of the step kernel
    bool profiling = false;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> events;
    size_t events_used = 0;
};

template <typename real> static void fill_consts(const qs_config &c, Consts<real> &k) {
    memset(&k, 0, sizeof k);
    for (int q = 0; q < 3; ++q) { k.inertia[q] = (real)c.inertia[q]; k.inv_inertia[q] = (real)(1.0 / c.inertia[q]); k.room_lo[q] = (real)c.room_lo[q]; k.room_hi[q] = (real)c.room_hi[q];
                                  k.nbr_clip_pos[q] = (real)c.nbr_clip_pos[q]; k.nbr_clip_vel[q] = (real)c.nbr_clip_vel[q]; }
    k.arm = (real)c.arm; k.mass = (real)c.mass; k.inv_mass = (real)(1.0 / c.mass);
    for (int m = 0; m < 4; ++m) { for (int q = 0; q < 3; ++q) k.prop_cross[m][q] = (real)c.prop_cross[m][q];
                                  k.prop_ccw[m] = (real)c.prop_ccw[m]; k.thrust_max[m] = (real)c.thrust_max[m]; k.torque_max[m] = (real)c.torque_max[m]; }
    k.motor_tau_up = (real)c.motor_tau_up; k.motor_tau_down = (real)c.motor_tau_down; k.motor_linearity = (real)c.motor_linearity;
    k.vel_damp = (real)c.vel_damp; k.damp_omega_quadratic = (real)c.damp_omega_quadratic; k.omega_max = (real)c.omega_max;
    k.thrust_noise_sigma = (real)c.thrust_noise_sigma; k.ou_theta = (real)c.ou_theta;
    k.dt = (real)c.dt; k.control_dt = (real)(c.dt * c.sim_steps);
    k.floor_threshold = (real)(c.floor_mode == QS_FLOOR_NUMPY ? 0.05 : c.arm);   // quadrotor_dynamics.py:75 / :378
    k.pos_norm_std = (real)c.pos_norm_std; k.pos_unif_range = (real)c.pos_unif_range; k.vel_norm_std = (real)c.vel_norm_std;
    k.vel_unif_range = (real)c.vel_unif_range; k.quat_norm_std = (real)c.quat_norm_std; k.quat_unif_range = (real)c.quat_unif_range;
    k.gyro_noise_density = (real)c.gyro_noise_density;
    k.collision_threshold = (real)c.collision_threshold; k.collision_falloff_threshold = (real)c.collision_falloff_threshold;
    for (int q = 0; q < QS_REW_COUNT; ++q) k.rew_coeff[q] = (real)c.rew_coeff[q];
    k.spawn_box = (real)c.spawn_box; k.approach_goal_metric = (real)c.approach_goal_metric;
    k.obst_radius = (real)(c.obst_size / 2.0); k.obst_hit_threshold = (real)(c.arm + c.obst_size / 2.0); k.obst_size = (real)c.obst_size;
    k.room_mid_z = (real)((c.room_hi[2] - c.room_lo[2]) / 2.0);
    k.sim_steps = c.sim_steps; k.ep_len = c.ep_len; k.floor_mode = c.floor_mode; k.svd_period = c.svd_period; k.sense_noise = c.sense_noise;
    k.obs_repr = c.obs_repr; k.self_dim = c.obs_repr == 0 ? 18 : (c.obs_repr == 1 ? 19 : 24); k.obs_dim = qs_obs_dim(&c);
    k.num_neighbors = c.num_neighbors; k.use_downwash = c.use_downwash; k.use_obstacles = c.use_obstacles; k.scenario = c.scenario;
    k.num_obstacles = c.num_obstacles; k.obst_area[0] = c.obst_area[0]; k.obst_area[1] = c.obst_area[1];
    const double control_freq = 1.0 / (c.dt * c.sim_steps);
    k.control_freq = (int)(control_freq + 0.5);
    k.grace_steps = (int)std::ceil(1.5 * control_freq - 1e-9);    // tick >= 1.5*control_freq (quadrotor_multi.py:146,:451)
    k.final_steps = (int)std::floor(5.0 * control_freq + 1e-9);   // time_remain <= 5*control_freq (:150,:455)
    k.cube_fd_all = (int)pow((double)c.num_agents, 1.0 / 3);
    k.cube_fd[0] = (int)pow((double)(c.num_agents / 2), 1.0 / 3);
    k.cube_fd[1] = (int)pow((double)(c.num_agents - c.num_agents / 2), 1.0 / 3);
    k.seed_lo = (uint32_t)(c.seed & 0xffffffffu); k.seed_hi = (uint32_t)(c.seed >> 32);
    k.env_id_offset = c.env_id_offset; k.num_envs = c.num_envs; k.num_agents = c.num_agents;
    k.write_rew_info = c.write_rew_info;
    k.inv_dt = (real)(1.0 / c.dt);
    k.prox_ratio = (real)(-c.rew_coeff[QS_REW_QUADCOL_SMOOTH_MAX] / c.collision_falloff_threshold);
    for (int w = 0; w < 3; ++w) {
        const int win = (w == 0 ? 1 : (w == 1 ? 3 : 5)) * k.control_freq, total = c.ep_len + 1;
        k.inv_win[w] = (real)(1.0 / (double)(total < win ? total : win));
    }
}

extern "C" int qs_obs_dim(const qs_config *c);

static int validate(const qs_config *c) {
    if (c->num_envs < 1) return fail(QS_ERR_INVALID, "num_envs must be >= 1");
    if (c->num_agents < 1 || c->num_agents > QS_MAX_AGENTS) return fail(QS_ERR_INVALID, "num_agents must be in [1, 64]");
    if (c->num_neighbors < 0 || c->num_neighbors > c->num_agents - 1) return fail(QS_ERR_INVALID, "Incorrect number of neigbors");
    if (c->precision != QS_PRECISION_F32 && c->precision != QS_PRECISION_F64) return fail(QS_ERR_INVALID, "bad precision");
    if (c->scenario < 0 || c->scenario >= QS_SCENARIO_COUNT) return fail(QS_ERR_UNSUPPORTED, "unsupported scenario");
    if (c->scenario == QS_SCENARIO_SWARM_VS_SWARM && c->num_agents < 2) return fail(QS_ERR_INVALID, "swarm_vs_swarm needs >= 2 drones");
    {
        const bool o_scen = c->scenario == QS_SCENARIO_O_STATIC_SAME_GOAL || c->scenario == QS_SCENARIO_O_RANDOM ||
                            c->scenario == QS_SCENARIO_O_DYNAMIC_SAME_GOAL || c->scenario == QS_SCENARIO_O_SWAP_GOALS;
        if (c->scenario != QS_SCENARIO_MIX && o_scen != (c->use_obstacles != 0)) return fail(QS_ERR_INVALID, "obstacle scenario <=> use_obstacles");
    }
    if (c->use_obstacles) {
        if (c->obst_area[0] < 1 || c->obst_area[1] < 1 || c->obst_area[0] > 16 || c->obst_area[1] > 16) return fail(QS_ERR_UNSUPPORTED, "obst_area must be within [1,16]x[1,16]");
        if (c->num_obstacles < 1 || c->num_obstacles > QS_MAX_OBSTACLES || c->num_obstacles > c->obst_area[0] * c->obst_area[1]) return fail(QS_ERR_INVALID, "bad num_obstacles");
        if (c->obst_area[0] * c->obst_area[1] - c->num_obstacles < c->num_agents) return fail(QS_ERR_INVALID, "not enough free cells to spawn the drones");
    }
    {
        LdsLayout L = lds_layout(c->precision == QS_PRECISION_F64 ? 8 : 4, QS_WAVE, c->num_agents, QS_WAVE / c->num_agents, qs_obs_dim(c), c->num_obstacles, c->num_neighbors);
        if (L.total > 160 * 1024) return fail(QS_ERR_UNSUPPORTED, "observation staging does not fit the 160 KiB LDS of a CU");
    }
    if (c->sim_steps < 1 || c->ep_len < 1 || c->svd_period < 1 || c->svd_period > 255) return fail(QS_ERR_INVALID, "bad sim_steps/ep_len/svd_period");
    return QS_OK;
}

template <typename T> static int dalloc(qs_handle *h, T **ptr, size_t count) {
    void *q = nullptr;
    size_t bytes = count * sizeof(T);
    if (bytes == 0) bytes = sizeof(T);
    HIP_TRY(hipMalloc(&q, bytes));
    HIP_TRY(hipMemset(q, 0, bytes));
    h->allocs.push_back(q);
    *ptr = (T *)q;
    return QS_OK;
}

template <typename real> static int create_typed(qs_handle *h) {
    const qs_config &c = h->cfg;
    const size_t E = c.num_envs, N = c.num_agents, T = E * N, D = h->obs_dim, M_ = c.num_obstacles;
    Ptrs<real> p;
    memset(&p, 0, sizeof p);
    int rc;
#define DA(field, count) if ((rc = dalloc(h, &p.field, (count))) != QS_OK) return rc
    DA(pos, 3 * T); DA(vel, 3 * T); DA(rot, 9 * T); DA(omega, 3 * T); DA(rot_damp, 4 * T); DA(cmds_damp, 4 * T); DA(ou, 4 * T); DA(goal, 3 * T);
    DA(flags, T); DA(pair_mask, T); DA(new_pair_mask, T);
    DA(obs, T * D); DA(reward, T); DA(rew_info, QS_RI_COUNT * T); DA(done, T); DA(obst_hit_idx, T);
    DA(unique_col, E); DA(obst_new, E); DA(room_new, E); DA(counters, QS_CNT_COUNT * E); DA(tick, E); DA(step_ctr, E);
    DA(obst_pos, 2 * E * (M_ ? M_ : 1)); DA(dist_ring, 4 * T); DA(dist_sums, 3 * T); DA(ep_stats, QS_EPS_COUNT * T); DA(ep_counters, QS_CNT_COUNT * E);
    DA(scen_real, SR_COUNT * E); DA(scen_int, SI_COUNT * E); DA(scen_omap, 4 * E); DA(scenario_id, E); DA(ep_scenario, E);
    DA(error_flag, 1); DA(reset_mask, E); DA(timing, 32);
#undef DA
    real *act = nullptr;
    if ((rc = dalloc(h, &act, 4 * T)) != QS_OK) return rc;
    h->d_actions = act;
    static_assert(sizeof(Ptrs<float>) == sizeof(Ptrs<double>), "layout");
    memcpy(&h->pf, &p, sizeof p);
    fill_consts<float>(c, h->kf);
    fill_consts<double>(c, h->kd);
    qs_buffers &b = h->bufs;
    memset(&b, 0, sizeof b);
    b.obs = p.obs; b.reward = p.reward; b.done = p.done; b.rew_info = p.rew_info; b.actions = act;
    b.pos = p.pos; b.vel = p.vel; b.omega = p.omega; b.rot = p.rot; b.thrust_rot_damp = p.rot_damp; b.thrust_cmds_damp = p.cmds_damp;
    b.ou_state = p.ou; b.goal = p.goal; b.flags = p.flags; b.obst_hit_idx = p.obst_hit_idx; b.col_pair_mask = p.pair_mask;
    b.new_pair_mask = p.new_pair_mask; b.unique_col_mask = p.unique_col; b.obst_new_mask = p.obst_new; b.room_new_mask = p.room_new;
    b.counters = p.counters; b.tick = p.tick; b.obst_pos = p.obst_pos; b.ep_stats = p.ep_stats; b.ep_counters = p.ep_counters;
    b.error_flag = p.error_flag; b.scenario_id = p.scenario_id; b.ep_scenario = p.ep_scenario; b.obs_dim = h->obs_dim; b.real_size = sizeof(real);
    return QS_OK;
}

extern "C" {

int qs_version(void) { return QS_VERSION; }
size_t qs_sizeof_config(void) { return sizeof(qs_config); }
const char *qs_last_error(void) { return g_last_error.c_str(); }

int qs_obs_dim(const qs_config *c) {
    int self = c->obs_repr == 0 ? 18 : (c->obs_repr == 1 ? 19 : 24);
    return self + 6 * c->num_neighbors + (c->use_obstacles ? 9 : 0);
}

int qs_default_config(qs_config *c, int32_t num_envs, int32_t num_agents) {
    // Crazyflie constants as derived by the reference at construction time; the python
    // host layer (quad-swarm-rl_amd/airframe.py) re-derives them from the link geometry and overrides these.
    memset(c, 0, sizeof *c);
    c->num_envs = num_envs; c->num_agents = num_agents; c->precision = QS_PRECISION_F32; c->seed = 0;
    c->mass = 0.028000000000000008; c->arm = 0.04596194077712559;
    c->inertia[0] = 1.3669232142857143e-05; c->inertia[1] = 1.4356732142857143e-05; c->inertia[2] = 2.656158333333334e-05;
    const double pc[4][3] = {{-0.0325, -0.0325, 0}, {-0.0325, 0.0325, 0}, {0.0325, 0.0325, 0}, {0.0325, -0.0325, 0}};
    const double ccw[4] = {-1, 1, -1, 1};
    for (int m = 0; m < 4; ++m) {
        for (int q = 0; q < 3; ++q) c->prop_cross[m][q] = pc[m][q];
        c->prop_ccw[m] = ccw[m];
        c->thrust_max[m] = 9.81 * c->mass * 1.9 / 4.0;
        c->torque_max[m] = 0.006 * c->thrust_max[m];
    }
    c->dt = 1.0 / 200.0; c->sim_steps = 2;
    c->motor_tau_up = c->motor_tau_down = 4 * c->dt / (0.15 + 1e-6);
    c->motor_linearity = 1.0; c->vel_damp = 0; c->damp_omega_quadratic = 0; c->omega_max = 40.0; c->gravity = 9.81;
    c->thrust_noise_sigma = 0.2 * 0.05; c->ou_theta = 0.15;
    c->ep_len = (int)(15.0 / (c->dt * c->sim_steps));
    c->room_lo[0] = -5; c->room_lo[1] = -5; c->room_lo[2] = 0; c->room_hi[0] = 5; c->room_hi[1] = 5; c->room_hi[2] = 10;
    c->floor_mode = QS_FLOOR_NUMBA;
    { double s = 0; int n = 0; do { s += c->dt; ++n; } while (!(s > 0.5)); c->svd_period = n; }
    c->sense_noise = 1; c->obs_repr = QS_OBS_XYZ_VXYZ_R_OMEGA;
    c->pos_norm_std = 0.005; c->vel_norm_std = 0.01; c->gyro_noise_density = 0.000175;
    c->num_neighbors = num_agents > 6 ? 6 : num_agents - 1;
    c->use_downwash = 0; c->use_obstacles = 0; c->scenario = QS_SCENARIO_STATIC_SAME_GOAL;
    c->collision_threshold = 2.0 * c->arm; c->collision_falloff_threshold = 4.0 * c->arm;
    const double rc[QS_REW_COUNT] = {1.0, 0.05, 1.0, 1.0, 0.1, 5.0, 4.0, 5.0};
    for (int q = 0; q < QS_REW_COUNT; ++q) c->rew_coeff[q] = rc[q];
    c->spawn_box = 2.0; c->approach_goal_metric = 0.5;
    for (int q = 0; q < 3; ++q) { c->nbr_clip_pos[q] = 10.0; c->nbr_clip_vel[q] = 6.0; }
    c->obst_size = 1.0; c->obst_density = 0.2; c->obst_area[0] = 6; c->obst_area[1] = 6; c->num_obstacles = 0;
    c->write_rew_info = 1;
    return QS_OK;
}

int qs_create(const qs_config *cfg, int device, qs_handle **out) {
    if (!cfg || !out) return fail(QS_ERR_INVALID, "null argument");
    int rc = validate(cfg);
    if (rc != QS_OK) return rc;
    int ndev = 0;
    HIP_TRY(hipGetDeviceCount(&ndev));
    if (device < 0 || device >= ndev) return fail(QS_ERR_HIP, "no such HIP device");
    HIP_TRY(hipSetDevice(device));
    qs_handle *h = new qs_handle();
    h->cfg = *cfg;
    h->device = device;
    h->real_size = cfg->precision == QS_PRECISION_F64 ? 8 : 4;
    h->obs_dim = qs_obs_dim(cfg);
    h->epb = QS_WAVE / cfg->num_agents;
    h->blocks = (cfg->num_envs + h->epb - 1) / h->epb;
    h->lds = lds_layout(h->real_size, QS_WAVE, cfg->num_agents, h->epb, h->obs_dim, cfg->num_obstacles, cfg->num_neighbors);
    h->full = !(cfg->scenario == QS_SCENARIO_STATIC_SAME_GOAL || cfg->scenario == QS_SCENARIO_O_STATIC_SAME_GOAL ||
                cfg->scenario == QS_SCENARIO_SWARM_VS_SWARM);
    rc = (h->real_size == 8) ? create_typed<double>(h) : create_typed<float>(h);
    if (rc == QS_OK) {
        if (hipMalloc((void **)&h->d_state_buf, sizeof(double) * QS_MAX_AGENTS * QS_STATE_STRIDE) != hipSuccess ||
            hipMalloc((void **)&h->d_tick_io, sizeof(int32_t)) != hipSuccess ||
            hipHostMalloc((void **)&h->h_mask, (size_t)cfg->num_envs) != hipSuccess)
            rc = fail(QS_ERR_HIP, "allocation failed");
    }
    if (rc != QS_OK) { qs_destroy(h); return rc; }
    if (h->lds.total > 64 * 1024) {
        const void *fns[] = {(const void *)qs_step_kernel<float>, (const void *)qs_step_kernel<double>, (const void *)qs_rollout_kernel<float>,
                             (const void *)qs_rollout_kernel<double>, (const void *)qs_step_kernel_full<float>, (const void *)qs_step_kernel_full<double>,
                             (const void *)qs_rollout_kernel_full<float>, (const void *)qs_rollout_kernel_full<double>,
                             (const void *)qs_reset_kernel<float, false>, (const void *)qs_reset_kernel<double, false>,
                             (const void *)qs_reset_kernel<float, true>, (const void *)qs_reset_kernel<double, true>};
        for (const void *fn : fns)
            if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, h->lds.total) != hipSuccess) {
                qs_destroy(h);
                return fail(QS_ERR_HIP, "cannot raise dynamic LDS limit");
            }
    }
    *out = h;
    return QS_OK;
}

int qs_destroy(qs_handle *h) {
    if (!h) return QS_OK;
    (void)hipSetDevice(h->device);
    (void)hipDeviceSynchronize();
    for (void *q : h->allocs) (void)hipFree(q);
    if (h->d_state_buf) (void)hipFree(h->d_state_buf);
    if (h->d_tick_io) (void)hipFree(h->d_tick_io);
    if (h->h_mask) (void)hipHostFree(h->h_mask);
    for (auto &ev : h->events) { (void)hipEventDestroy(ev.first); (void)hipEventDestroy(ev.second); }
    if (h->graph_exec) (void)hipGraphExecDestroy(h->graph_exec);
    if (h->cap_stream) (void)hipStreamDestroy(h->cap_stream);
    delete h;
    return QS_OK;
}

static int launch_reset(qs_handle *h, hipStream_t s) {
    if (h->real_size == 8) {
        Ptrs<double> p; memcpy(&p, &h->pf, sizeof p);
        if (h->full) hipLaunchKernelGGL((qs_reset_kernel<double, true>), dim3(h->blocks), dim3(QS_WAVE), h->lds.total, s, h->kd, p, h->lds, h->epb);
        else hipLaunchKernelGGL((qs_reset_kernel<double, false>), dim3(h->blocks), dim3(QS_WAVE), h->lds.total, s, h->kd, p, h->lds, h->epb);
    } else {
        if (h->full) hipLaunchKernelGGL((qs_reset_kernel<float, true>), dim3(h->blocks), dim3(QS_WAVE), h->lds.total, s, h->kf, h->pf, h->lds, h->epb);
        else hipLaunchKernelGGL((qs_reset_kernel<float, false>), dim3(h->blocks), dim3(QS_WAVE), h->lds.total, s, h->kf, h->pf, h->lds, h->epb);
    }
    HIP_TRY(hipGetLastError());
    return QS_OK;
}

int qs_reset(qs_handle *h, const uint8_t *env_mask_host, void *stream) {
    if (!h) return fail(QS_ERR_INVALID, "null handle");
    HIP_TRY(hipSetDevice(h->device));
    hipStream_t s = (hipStream_t)stream;
    const int E = h->cfg.num_envs;
    for (int e = 0; e < E; ++e) h->h_mask[e] = env_mask_host ? (env_mask_host[e] ? 1 : 0) : 1;
    HIP_TRY(hipMemcpyAsync(h->pf.reset_mask, h->h_mask, (size_t)E, hipMemcpyHostToDevice, s));
    int rc = launch_reset(h, s);
    if (rc != QS_OK) return rc;
    HIP_TRY(hipStreamSynchronize(s));   // h_mask is reused by the next call
    return QS_OK;
}

static int launch_step(qs_handle *h, const void *actions, hipStream_t s, int ksteps = 1) {
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (h->profiling) {
        if (h->events_used == h->events.size()) {
            hipEvent_t a, b;
            HIP_TRY(hipEventCreate(&a)); HIP_TRY(hipEventCreate(&b));
            h->events.emplace_back(a, b);
        }
        e0 = h->events[h->events_used].first; e1 = h->events[h->events_used].second;
        ++h->events_used;
        HIP_TRY(hipEventRecord(e0, s));
    }
#define QS_LAUNCH(KERNEL, CONSTS, PTRS, TYPE, ...) hipLaunchKernelGGL(KERNEL<TYPE>, dim3(h->blocks), dim3(QS_WAVE), h->lds.total, s, CONSTS, PTRS, \
                                                                    (const TYPE *)actions, h->lds, h->epb, ##__VA_ARGS__)
    if (h->real_size == 8) {
        Ptrs<double> p; memcpy(&p, &h->pf, sizeof p);
        if (ksteps == 1) { if (h->full) QS_LAUNCH(qs_step_kernel_full, h->kd, p, double); else QS_LAUNCH(qs_step_kernel, h->kd, p, double); }
        else { if (h->full) QS_LAUNCH(qs_rollout_kernel_full, h->kd, p, double, ksteps); else QS_LAUNCH(qs_rollout_kernel, h->kd, p, double, ksteps); }
    } else {
        if (ksteps == 1) { if (h->full) QS_LAUNCH(qs_step_kernel_full, h->kf, h->pf, float); else QS_LAUNCH(qs_step_kernel, h->kf, h->pf, float); }
        else { if (h->full) QS_LAUNCH(qs_rollout_kernel_full, h->kf, h->pf, float, ksteps); else QS_LAUNCH(qs_rollout_kernel, h->kf, h->pf, float, ksteps); }
    }
#undef QS_LAUNCH
    HIP_TRY(hipGetLastError());
    if (h->profiling) HIP_TRY(hipEventRecord(e1, s));
    return QS_OK;   // the auto-reset is the tail of the step kernel itself
}

int qs_step(qs_handle *h, const void *actions_dev, void *stream) {
    if (!h) return fail(QS_ERR_INVALID, "null handle");
    HIP_TRY(hipSetDevice(h->device));
    return launch_step(h, actions_dev ? actions_dev : h->d_actions, (hipStream_t)stream);
}

int qs_step_many(qs_handle *h, const void *actions_dev, int32_t k, void *stream) {
    if (!h || !actions_dev || k < 0) return fail(QS_ERR_INVALID, "bad argument");
    HIP_TRY(hipSetDevice(h->device));
    const size_t stride = (size_t)h->cfg.num_envs * h->cfg.num_agents * 4 * h->real_size;
    if (h->profiling) {   // per-step HIP events: one launch per control step
        for (int32_t t = 0; t < k; ++t) {
            int rc = launch_step(h, (const char *)actions_dev + stride * t, (hipStream_t)stream, 1);
            if (rc != QS_OK) return rc;
        }
        return QS_OK;
    }
    // open-loop rollout: the step kernel keeps the state in registers across up to QS_STEPS_PER_LAUNCH control steps
    const int32_t per = 64;
    for (int32_t t = 0; t < k; t += per) {
        int rc = launch_step(h, (const char *)actions_dev + stride * t, (hipStream_t)stream, (k - t) < per ? (k - t) : per);
        if (rc != QS_OK) return rc;
    }
    return QS_OK;
}

int qs_sync(qs_handle *h, void *stream) {
    if (!h) return fail(QS_ERR_INVALID, "null handle");
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    return QS_OK;
}

int qs_get_buffers(qs_handle *h, qs_buffers *out) {
    if (!h || !out) return fail(QS_ERR_INVALID, "null argument");
    *out = h->bufs;
    return QS_OK;
}

int qs_set_reward_coeffs(qs_handle *h, const double *coeffs) {
    if (!h || !coeffs) return fail(QS_ERR_INVALID, "null argument");
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipDeviceSynchronize());
    for (int q = 0; q < QS_REW_COUNT; ++q) h->cfg.rew_coeff[q] = coeffs[q];
    fill_consts<float>(h->cfg, h->kf);
    fill_consts<double>(h->cfg, h->kd);
    if (h->graph_exec) { (void)hipGraphExecDestroy(h->graph_exec); h->graph_exec = nullptr; }   // constants are baked into the nodes
    return QS_OK;
}

static int state_io(qs_handle *h, int32_t env, double *host, int32_t *tick, int set) {
    if (!h || !host) return fail(QS_ERR_INVALID, "null argument");
    if (env < 0 || env >= h->cfg.num_envs) return fail(QS_ERR_INVALID, "env out of range");
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipDeviceSynchronize());
    const int N = h->cfg.num_agents;
    const size_t bytes = sizeof(double) * N * QS_STATE_STRIDE;
    int32_t t = tick ? *tick : -1;
    if (set) {
        HIP_TRY(hipMemcpy(h->d_state_buf, host, bytes, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(h->d_tick_io, &t, sizeof t, hipMemcpyHostToDevice));
    }
    if (h->real_size == 8) { Ptrs<double> p; memcpy(&p, &h->pf, sizeof p);
        hipLaunchKernelGGL(qs_state_kernel<double>, dim3(1), dim3(QS_WAVE), 0, 0, p, h->cfg.num_envs, N, env, h->d_state_buf, h->d_tick_io, set); }
    else hipLaunchKernelGGL(qs_state_kernel<float>, dim3(1), dim3(QS_WAVE), 0, 0, h->pf, h->cfg.num_envs, N, env, h->d_state_buf, h->d_tick_io, set);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipDeviceSynchronize());
    if (!set) {
        HIP_TRY(hipMemcpy(host, h->d_state_buf, bytes, hipMemcpyDeviceToHost));
        HIP_TRY(hipMemcpy(&t, h->d_tick_io, sizeof t, hipMemcpyDeviceToHost));
        if (tick) *tick = t;
    }
    return QS_OK;
}

int qs_get_state(qs_handle *h, int32_t env, double *state_host, int32_t *tick) { return state_io(h, env, state_host, tick, 0); }
int qs_set_state(qs_handle *h, int32_t env, const double *state_host, int32_t tick) { int32_t t = tick; return state_io(h, env, (double *)state_host, &t, 1); }

int qs_memcpy_d2h(qs_handle *h, void *host_dst, const void *dev_src, size_t bytes) {
    if (!h || !host_dst || !dev_src) return fail(QS_ERR_INVALID, "null argument");
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(host_dst, dev_src, bytes, hipMemcpyDeviceToHost));
    return QS_OK;
}

int qs_memcpy_h2d(qs_handle *h, void *dev_dst, const void *host_src, size_t bytes) {
    if (!h || !dev_dst || !host_src) return fail(QS_ERR_INVALID, "null argument");
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(dev_dst, host_src, bytes, hipMemcpyHostToDevice));
    return QS_OK;
}

/* debug: phase time stamps (shader clock) of workgroup 0; all zero unless built with -DQS_TIMING */
int qs_debug_timing(qs_handle *h, unsigned long long *out32) {
    if (!h || !out32) return fail(QS_ERR_INVALID, "null argument");
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(out32, h->pf.timing, 32 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    return QS_OK;
}

int qs_check_errors(qs_handle *h) {
    if (!h) return fail(QS_ERR_INVALID, "null handle");
    HIP_TRY(hipSetDevice(h->device));
    uint32_t f = 0;
    HIP_TRY(hipMemcpy(&f, h->pf.error_flag, sizeof f, hipMemcpyDeviceToHost));
    if (f) return fail(QS_ERR_NAN_REWARD, "QuadEnv: reward is Nan");
    return QS_OK;
}

int qs_set_profiling(qs_handle *h, int32_t enable) {
    if (!h) return fail(QS_ERR_INVALID, "null handle");
    h->profiling = enable != 0;
    h->events_used = 0;
    return QS_OK;
}

int qs_get_kernel_time(qs_handle *h, double *avg_ms, int64_t *launches) {
    if (!h || !avg_ms || !launches) return fail(QS_ERR_INVALID, "null argument");
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipDeviceSynchronize());
    double total = 0;
    for (size_t k = 0; k < h->events_used; ++k) {
        float ms = 0;
        HIP_TRY(hipEventElapsedTime(&ms, h->events[k].first, h->events[k].second));
        total += ms;
    }
    *launches = (int64_t)h->events_used;
    *avg_ms = h->events_used ? total / (double)h->events_used : 0.0;
    h->events_used = 0;
    return QS_OK;
}

}  // extern "C"
