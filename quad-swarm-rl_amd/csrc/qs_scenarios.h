// qs_scenarios.h - the full scenario set (scenarios/*.py of the reference) for the "full" kernel variants.
//
// The fast kernels (static_same_goal / o_static_same_goal / swarm_vs_swarm, BASELINE configs 2-4) keep their own
// compact code; configurations with any other `quads_mode` (incl. `mix`) run kernels compiled with QS_SCEN_FULL, whose
// per-environment scenario state lives in LDS for the duration of a launch (loaded from / stored to `scen_real`,
// `scen_int`, `scen_omap`).  One lane per env (drone 0) runs the serial pieces; drones pick their goals up from LDS.
#pragma once

#include "qs_device.h"

namespace qs {

// scen_int slots
enum { SI_PERIOD = 0, SI_SCEN, SI_FORM, SI_PER_LAYER, SI_INCREASE, SI_BEZ_VALID, SI_CONSTRUCTED, SI_HAVE_SPAWN, SI_COUNT = 8 };
// scen_real slots
enum { SR_C1 = 0, SR_C2 = 3, SR_LO = 6, SR_HI, SR_SIZE, SR_LAYER, SR_SPEED, SR_BEZ = 11, SR_END = 20, SR_METRIC = 23, SR_COUNT = 24 };

template <typename real> struct ScenCtx {
    real *sr;         // [SR_COUNT] this env's scenario reals (LDS)
    int *si;          // [SI_COUNT] this env's scenario ints (LDS)
    uint64_t *omap;   // [4] obstacle map bitset of this env (LDS), cell id = row*W + col
    real *goals;      // [(2N+8)][3] goal scratch rows of this env (LDS); rows 0..N-1 are the drones' goals
    real *spawn;      // spawn points, component-major with stride B: spawn[q*B + base + k]
    int B, base, N;
};

template <typename real> __device__ __forceinline__ void load_formation(const ScenCtx<real> &x, Formation<real> &F) {
    F.f = x.si[SI_FORM]; F.per_layer = x.si[SI_PER_LAYER]; F.lo = x.sr[SR_LO]; F.hi = x.sr[SR_HI]; F.size = x.sr[SR_SIZE];
    F.layer_dist = x.sr[SR_LAYER];
}
template <typename real> __device__ __forceinline__ void store_formation(const ScenCtx<real> &x, const Formation<real> &F) {
    x.si[SI_FORM] = F.f; x.si[SI_PER_LAYER] = F.per_layer; x.sr[SR_LO] = F.lo; x.sr[SR_HI] = F.hi; x.sr[SR_SIZE] = F.size;
    x.sr[SR_LAYER] = F.layer_dist;
}

// cell centre of grid cell (row x, col y): cell_centers[x + L*y] with the layout of obstacles/utils.py:47-58
template <typename real> __device__ __forceinline__ void cell_center(int Lr, int W, int x, int y, real *px, real *py) {
    int index = x + Lr * y, ii = index / W, jj = (W - 1) - (index - ii * W);
    *px = (real)ii + (real)0.5 - (real)(Lr / 2);
    *py = (real)jj + (real)0.5 - (real)(W / 2);
}
__device__ __forceinline__ int map_count(const uint64_t *omap) { return __popcll(omap[0]) + __popcll(omap[1]) + __popcll(omap[2]) + __popcll(omap[3]); }
// k-th free cell in np.where(obst_map == 0) order
__device__ __forceinline__ int kth_free_cell(const uint64_t *omap, int cells, int k) {
    int seen = 0;
    for (int id = 0; id < cells; ++id) if (!(omap[id >> 6] >> (id & 63) & 1)) { if (seen == k) return id; ++seen; }
    return 0;
}

// Scenario_o_base.generate_pos_obst_map_2 o_base.py:69-81: N distinct free cells (partial Fisher-Yates on a virtual pool
// held in `tidx/tval`, LDS scratch) + z ~ U(1,3).  out: rows of stride `ld` starting at out[0] (x,y,z consecutive) or the
// component-major spawn array when `to_spawn`.
template <typename real>
__device__ QS_COLD void pos_obst_map_2(const Consts<real> &c, const RngKey &key, const ScenCtx<real> &x, int *tidx, int *tval,
    int slot_choice, int slot_z,
                               bool to_spawn) {
    // free cells = cells not in the obstacle map (the episode's obstacle count varies under --quads_domain_random)
    const int Lr = c.obst_area[0], W = c.obst_area[1], cells = Lr * W, nfree = cells - map_count(x.omap), N = x.N;
    const bool on_tape = QS_ON_TAPE(key);   // a tape holds the N chosen free-cell indices, then the N heights (o_base.py:69-81)
    int nt = 0;
    for (int k = 0; k < N; ++k) {
        int vj;
        if (on_tape) vj = (int)tape_pop(key);
        else {
            int j = k + rng_index<real>(key, QS_SITE_SCEN, slot_choice + k, nfree - k);
            int vk = k, pj = -1;
            vj = j;
            for (int q = 0; q < nt; ++q) { if (tidx[q] == k) vk = tval[q]; if (tidx[q] == j) { vj = tval[q]; pj = q; } }
            if (pj >= 0) tval[pj] = vk; else { tidx[nt] = j; tval[nt] = vk; ++nt; }
        }
        int cell = kth_free_cell(x.omap, cells, vj), cx = cell / W, cy = cell - cx * W;
        real px, py, pz = on_tape ? (real)0 : rng_uniform1<real>(key, QS_SITE_SCEN, slot_z + k, 0, 0, (real)1, (real)3);
        cell_center<real>(Lr, W, cx, cy, &px, &py);
        if (to_spawn) { x.spawn[0 * x.B + x.base + k] = px; x.spawn[1 * x.B + x.base + k] = py; x.spawn[2 * x.B + x.base + k] = pz; }
        else { x.goals[k * 3 + 0] = px; x.goals[k * 3 + 1] = py; x.goals[k * 3 + 2] = pz; }
    }
    if (on_tape) for (int k = 0; k < N; ++k) {
        const real pz = (real)tape_pop(key);
        if (to_spawn) x.spawn[2 * x.B + x.base + k] = pz; else x.goals[k * 3 + 2] = pz;
    }
}
// Scenario_o_base.generate_pos_obst_map o_base.py:48-67 (no surroundings check): one free cell + z ~ U(0.75,3)
template <typename real>
__device__ QS_COLD void pos_obst_map_1(const Consts<real> &c, const RngKey &key, const uint64_t *omap, int slot, real out[3]) {
    const int Lr = c.obst_area[0], W = c.obst_area[1], cells = Lr * W, nfree = cells - map_count(omap);
    int idx = rng_index<real>(key, QS_SITE_SCEN, slot, nfree);
    int cell = kth_free_cell(omap, cells, idx), cx = cell / W, cy = cell - cx * W;
    cell_center<real>(Lr, W, cx, cy, &out[0], &out[1]);
    out[2] = rng_uniform1<real>(key, QS_SITE_SCEN, slot + 1, 0, 0, (real)0.75, (real)3);
}
// Scenario_o_base.max_square_area_center o_base.py:124-153 (two-row dynamic programme in LDS scratch rows)
template <typename real>
__device__ QS_COLD void max_square_center(const Consts<real> &c, const RngKey &key, const uint64_t *omap, int *prev_row, int *cur_row,
    int slot_z, real out[3]) {
    const int Lr = c.obst_area[0], W = c.obst_area[1];
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
    out[0] = (real)ii + (real)0.5 - (real)(Lr / 2);
    out[1] = (real)jj + (real)0.5 - (real)(W / 2);
    out[2] = rng_uniform1<real>(key, QS_SITE_SCEN, slot_z, 0, 0, (real)1.5, (real)3);
}

// get_z_value scenarios/utils.py:170-181
template <typename real>
__device__ QS_COLD real get_z_value(const Consts<real> &c, const RngKey &key, const Formation<real> &F, int N, int slot) {
    real box = c.spawn_box;
    real z = rng_uniform1<real>(key, QS_SITE_SCEN, slot, 0, 0, (real)-0.5 * box, (real)0.5 * box) + (real)2, zlb = (real)0.25;
    const int f = F.f;
    if (f == 3 || f == 1 || f == 2) zlb = F.size + (real)0.25;
    else if (f == 5 || f == 6) { int rn = N < F.per_layer ? N : F.per_layer, d1, d2; grid_dim(rn, &d1, &d2);
        zlb = (real)d1 * F.size + (real)0.25; }
    return M<real>::fmax(zlb, z);
}

// QuadrotorScenario.standard_reset scenarios/base.py:153-167
template <typename real>
__device__ QS_COLD void standard_reset(const Consts<real> &c, const RngKey &key, const ScenCtx<real> &x, int scen, const real center[3]) {
    Formation<real> F;
    update_formation<real>(scen, key, 0, x.N, F);
    store_formation<real>(x, F);
    for (int q = 0; q < 3; ++q) x.sr[SR_C1 + q] = center[q];
    int rows = generate_goals<real>(F, x.N, c.cube_fd_all, center, x.goals, 3);
    shuffle_rows<real>(key, x.goals, 3, rows, 0);
}

// u < 0: -(k+1) encodes the list index k the reference drew (noise tape)
__device__ __forceinline__ int mix_pick(int num_agents, bool use_obstacles, double u) {   // scenarios/mix.py:84-90 + utils.py:10-25
    const int LIST_MULTI[9] = {QS_SCENARIO_STATIC_SAME_GOAL, QS_SCENARIO_STATIC_DIFF_GOAL, QS_SCENARIO_EP_LISSAJOUS3D,
        QS_SCENARIO_EP_RAND_BEZIER,
                               QS_SCENARIO_DYNAMIC_SAME_GOAL, QS_SCENARIO_DYNAMIC_DIFF_GOAL, QS_SCENARIO_DYNAMIC_FORMATIONS,
                                   QS_SCENARIO_SWAP_GOALS,
                               QS_SCENARIO_SWARM_VS_SWARM};
    int n, base_list;   // base_list 0: multi / single (a prefix of it), 1: obstacles
    if (num_agents == 1) { if (use_obstacles) { n = 1; base_list = 1; } else { n = 5; base_list = 0; } }
    else if (!use_obstacles) { n = 9; base_list = 0; }
    else { n = 2; base_list = 1; }
    int k = (u < 0.0) ? (int)(-u) - 1 : (int)(u * (double)n);
    if (k >= n) k = n - 1;
    if (base_list == 1) return k == 0 ? QS_SCENARIO_O_RANDOM : QS_SCENARIO_O_STATIC_SAME_GOAL;
    if (n == 5) { const int LIST_SINGLE[5] = {QS_SCENARIO_STATIC_SAME_GOAL, QS_SCENARIO_STATIC_DIFF_GOAL, QS_SCENARIO_EP_LISSAJOUS3D,
                                              QS_SCENARIO_EP_RAND_BEZIER, QS_SCENARIO_DYNAMIC_SAME_GOAL}; return LIST_SINGLE[k]; }
    return LIST_MULTI[k];
}

// scenario.reset() of every scenario (executed by drone 0 of the env); mirrors oracle/quadswarm_oracle.c:scenario_reset
template <typename real>
__device__ QS_COLD void scenario_reset_full(const Consts<real> &c, const RngKey &key, const ScenCtx<real> &x, int *scratch) {
    const int N = x.N;
    int *tidx = scratch, *tval = scratch + 64, *prev_row = scratch + 128, *cur_row = scratch + 144;
    int sc, constructed;
    if (c.scenario == QS_SCENARIO_MIX) {
        // the pick uses a double-precision-equivalent index: u*n with u exact in fp32 (23-bit grid) and n <= 9
        real u = rng_uniform1<real>(key, QS_SITE_SCEN, 288, 0, 0, (real)0, (real)1);   // (a tape holds the list index itself)
        sc = mix_pick(c.num_agents, c.use_obstacles != 0, QS_ON_TAPE(key) ? -((double)(int)u + 1.0) : (double)u);
        constructed = 1;
    } else { sc = c.scenario; constructed = !x.si[SI_CONSTRUCTED]; x.si[SI_CONSTRUCTED] = 1; }
    x.si[SI_SCEN] = sc;
    x.si[SI_HAVE_SPAWN] = 0;
    x.si[SI_BEZ_VALID] = 0;
    if (constructed) {
        x.si[SI_PERIOD] = (int)((sc == QS_SCENARIO_O_SWAP_GOALS ? (real)6 : (real)5) * (real)c.control_freq);
        if (sc == QS_SCENARIO_DYNAMIC_FORMATIONS) x.sr[SR_SPEED] = rng_uniform1<real>(key, QS_SITE_SCEN, 289, 0, 0, (real)1, (real)3);
    }
    const real c002[3] = {0, 0, 2};
    if (sc == QS_SCENARIO_STATIC_SAME_GOAL || sc == QS_SCENARIO_STATIC_DIFF_GOAL || sc == QS_SCENARIO_EP_RAND_BEZIER) {
        standard_reset<real>(c, key, x, sc, c002);
    } else if (sc == QS_SCENARIO_RUN_AWAY) {             // run_away.py:29-40 (= the base reset); step() acts once per second
        x.si[SI_PERIOD] = c.control_freq;
        standard_reset<real>(c, key, x, sc, c002);
    } else if (sc == QS_SCENARIO_DYNAMIC_SAME_GOAL || sc == QS_SCENARIO_DYNAMIC_DIFF_GOAL || sc == QS_SCENARIO_SWAP_GOALS) {
        x.si[SI_PERIOD] = draw_period<real>(key, 8, 4.0, 6.0, c.control_freq);
        standard_reset<real>(c, key, x, sc, c002);
    } else if (sc == QS_SCENARIO_DYNAMIC_FORMATIONS) {
        x.si[SI_INCREASE] = rng_uniform1<real>(key, QS_SITE_SCEN, 290, 0, 0, (real)0, (real)1) < (real)0.5;
        x.sr[SR_SPEED] = rng_uniform1<real>(key, QS_SITE_SCEN, 291, 0, 0, (real)1, (real)3);
        standard_reset<real>(c, key, x, sc, c002);
    } else if (sc == QS_SCENARIO_EP_LISSAJOUS3D) {
        Formation<real> F;
        update_formation<real>(sc, key, 0, N, F);
        store_formation<real>(x, F);
        const real ctr[3] = {-2, 0, 2};
        for (int q = 0; q < 3; ++q) x.sr[SR_C1 + q] = ctr[q];
        F.layer_dist = 0;
        generate_goals<real>(F, N, c.cube_fd_all, ctr, x.goals, 3);
    } else if (sc == QS_SCENARIO_O_STATIC_SAME_GOAL || sc == QS_SCENARIO_O_DYNAMIC_SAME_GOAL) {
        x.si[SI_PERIOD] = draw_period<real>(key, 8, 4.0, 6.0, c.control_freq);
        pos_obst_map_2<real>(c, key, x, tidx, tval, 16, 96, true);
        x.si[SI_HAVE_SPAWN] = 1;
        real end[3];
        max_square_center<real>(c, key, x.omap, prev_row, cur_row, 9, end);
        Formation<real> F;
        update_formation<real>(sc, key, 0, N, F);
        store_formation<real>(x, F);
        for (int q = 0; q < 3; ++q) x.sr[SR_END + q] = end[q];
        for (int k = 0; k < N; ++k) for (int q = 0; q < 3; ++q) x.goals[k * 3 + q] = end[q];
    } else if (sc == QS_SCENARIO_O_EP_RAND_BEZIER) {     // obstacles/o_ep_rand_bezier.py:57-97
        pos_obst_map_2<real>(c, key, x, tidx, tval, 16, 96, true);
        x.si[SI_HAVE_SPAWN] = 1;
        real end[3];
        pos_obst_map_1<real>(c, key, x.omap, 9, end);
        // :72-90 sample ten "trajectory points" that step() never uses: with the counter-based stream there is nothing to skip; a tape
        // holds those draws (np.random.choice over the free cells, rejected while farther than 4 m from an accepted one - the
        // reference indexes its cell-centre table with the free-space index, reproduced here)
        if (QS_ON_TAPE(key)) {
            const int Lr = c.obst_area[0], W = c.obst_area[1];
            int ns = 0;
            while (ns < 10) {
                const int idx = (int)tape_pop(key);
                const int ii = idx / W, jj = (W - 1) - (idx - ii * W);
                const real px = (real)ii + (real)0.5 - (real)(Lr / 2), py = (real)jj + (real)0.5 - (real)(W / 2);
                bool reject = false;
                for (int q = 0; q < ns; ++q) {
                    const int sq = tidx[q], si2 = sq / W, sj2 = (W - 1) - (sq - si2 * W);
                    const real dx = ((real)si2 + (real)0.5 - (real)(Lr / 2)) - px, dy = ((real)sj2 + (real)0.5 - (real)(W / 2)) - py;
                    if (M<real>::sqrt(dx * dx + dy * dy) > (real)4) reject = true;
                }
                if (!reject) tidx[ns++] = idx;
            }
        }
        Formation<real> F;
        update_formation<real>(sc, key, 0, N, F);
        store_formation<real>(x, F);
        for (int q = 0; q < 3; ++q) x.sr[SR_END + q] = end[q];
        for (int k = 0; k < N; ++k) for (int q = 0; q < 3; ++q) x.goals[k * 3 + q] = end[q];
    } else if (sc == QS_SCENARIO_O_RANDOM) {
        if (QS_ON_TAPE(key)) tape_skip(key, 4 * N);   // o_random.py:30-36: N x generate_pos_obst_map() twice, overwritten right below
        pos_obst_map_2<real>(c, key, x, tidx, tval, 16, 96, true);
        x.si[SI_HAVE_SPAWN] = 1;
        pos_obst_map_2<real>(c, key, x, tidx, tval, 160, 224, false);
        (void)rng_uniform1<real>(key, QS_SITE_SCEN, 8, 0, 0, (real)2, (real)4);   // duration_step: its step() re-assigns the same goals
        Formation<real> F;
        update_formation<real>(sc, key, 0, N, F);
        store_formation<real>(x, F);
    } else if (sc == QS_SCENARIO_O_SWAP_GOALS) {
        x.si[SI_PERIOD] = draw_period<real>(key, 8, 4.0, 6.0, c.control_freq);
        Formation<real> F;
        update_formation<real>(sc, key, 0, N, F);
        store_formation<real>(x, F);
        pos_obst_map_2<real>(c, key, x, tidx, tval, 16, 96, true);
        x.si[SI_HAVE_SPAWN] = 1;
        real ctr[3];
        max_square_center<real>(c, key, x.omap, prev_row, cur_row, 9, ctr);
        for (int q = 0; q < 3; ++q) x.sr[SR_C1 + q] = ctr[q];
        int rows = generate_goals<real>(F, N, c.cube_fd_all, ctr, x.goals, 3);
        shuffle_rows<real>(key, x.goals, 3, rows, 0);
    } else {   // swarm_vs_swarm: scenarios/swarm_vs_swarm.py:80-94, :17-50
        x.si[SI_PERIOD] = draw_period<real>(key, 8, 4.0, 6.0, c.control_freq);
        Formation<real> F;
        update_formation<real>(sc, key, 0, N, F);
        store_formation<real>(x, F);
        real box = c.spawn_box, xy[2];
        rng_uniform<real, 2>(key, QS_SITE_SCEN, 9, 0, 0, -box, box, xy);
        real z = get_z_value<real>(c, key, F, N, 10);
        real c1[3] = {xy[0], xy[1], z}, c2[3];
        real dist = rng_uniform1<real>(key, QS_SITE_SCEN, 11, 0, 0, box / (real)4, box);
        real phi = rng_uniform1<real>(key, QS_SITE_SCEN, 12, 0, 0, (real)-QS_PI_D, (real)QS_PI_D);
        real theta = rng_uniform1<real>(key, QS_SITE_SCEN, 13, 0, 0, (real)(-0.5 * QS_PI_D), (real)(0.5 * QS_PI_D));
        real st, ct, sp, cph; M<real>::sincos(theta, &st, &ct); M<real>::sincos(phi, &sp, &cph);
        c2[0] = c1[0] + dist * (st * cph); c2[1] = c1[1] + dist * (st * sp); c2[2] = c1[2] + dist * ct;
        int s = f_suffix(F.f), ax = (s == 0) ? 2 : ((s == 1) ? 1 : ((s == 2) ? 0 : -1));
        if (ax >= 0) {
            real df = c2[ax] - c1[ax];
            if (M<real>::fabs(df) < F.lo) { real sg = (real)((df > 0) - (df < 0)); c2[ax] = sg * F.lo + c1[ax]; }
        }
        for (int q = 0; q < 3; ++q) { x.sr[SR_C1 + q] = c1[q]; x.sr[SR_C2 + q] = c2[q]; }
        svs_create_formations<real>(key, F, N, QS_CUBE_FD(c), c1, c2, false, x.goals);
    }
    x.sr[SR_METRIC] = (sc == QS_SCENARIO_O_STATIC_SAME_GOAL || sc == QS_SCENARIO_O_DYNAMIC_SAME_GOAL || sc == QS_SCENARIO_O_SWAP_GOALS ||
                       sc == QS_SCENARIO_O_EP_RAND_BEZIER) ? (real)1 : (real)0.5;
}

// Does scenario `sc` need the serial (one lane per env, all goals) part of step() at this tick?
__device__ __forceinline__ bool scen_step_serial_needed(int sc, int period, int tick) {
    const bool at_period = period > 0 && tick > 0 && imod_small(tick, period) == 0;
    return sc == QS_SCENARIO_DYNAMIC_FORMATIONS ||
           (at_period && (sc == QS_SCENARIO_SWARM_VS_SWARM || sc == QS_SCENARIO_DYNAMIC_DIFF_GOAL || sc == QS_SCENARIO_SWAP_GOALS
               || sc == QS_SCENARIO_O_SWAP_GOALS ||
                          sc == QS_SCENARIO_RUN_AWAY));
}

// serial part of scenario.step(): rewrites the env's goal rows in LDS (current goals were published there first)
template <typename real>
__device__ QS_COLD void scenario_step_serial(const Consts<real> &c, const RngKey &key, const ScenCtx<real> &x, int sc) {
    const int N = x.N;
    Formation<real> F;
    if (sc == QS_SCENARIO_SWARM_VS_SWARM) {            // swarm_vs_swarm.py:59-79
        real c1[3], c2[3];
        for (int q = 0; q < 3; ++q) { c1[q] = x.sr[SR_C2 + q]; c2[q] = x.sr[SR_C1 + q]; }
        for (int q = 0; q < 3; ++q) { x.sr[SR_C1 + q] = c1[q]; x.sr[SR_C2 + q] = c2[q]; }
        update_formation<real>(sc, key, 32, N, F);
        store_formation<real>(x, F);
        svs_create_formations<real>(key, F, N, QS_CUBE_FD(c), c1, c2, true, x.goals);
    } else if (sc == QS_SCENARIO_DYNAMIC_DIFF_GOAL) {   // dynamic_diff_goal.py:8-34
        load_formation<real>(x, F);
        real box = c.spawn_box, xy[2];
        rng_uniform<real, 2>(key, QS_SITE_SCEN, 40, 0, 0, -box, box, xy);
        real ctr[3] = {xy[0], xy[1], get_z_value<real>(c, key, F, N, 41)};
        for (int q = 0; q < 3; ++q) x.sr[SR_C1 + q] = ctr[q];
        update_formation<real>(sc, key, 32, N, F);
        store_formation<real>(x, F);
        int rows = generate_goals<real>(F, N, c.cube_fd_all, ctr, x.goals, 3);
        shuffle_rows<real>(key, x.goals, 3, rows, 0);
    } else if (sc == QS_SCENARIO_DYNAMIC_FORMATIONS) {  // dynamic_formations.py:16-35
        load_formation<real>(x, F);
        int inc = x.si[SI_INCREASE];
        real speed = x.sr[SR_SPEED];
        if (F.size <= -F.hi) { inc = 1; speed = rng_uniform1<real>(key, QS_SITE_SCEN, 292, 0, 0, (real)1, (real)3); }
        else if (F.size >= F.hi) { inc = 0; speed = rng_uniform1<real>(key, QS_SITE_SCEN, 292, 0, 0, (real)1, (real)3); }
        if (inc) F.size += (real)0.001 * speed; else F.size -= (real)0.001 * speed;
        x.si[SI_INCREASE] = inc; x.sr[SR_SPEED] = speed; x.sr[SR_SIZE] = F.size;
        real ctr[3] = {x.sr[SR_C1], x.sr[SR_C1 + 1], x.sr[SR_C1 + 2]};
        generate_goals<real>(F, N, c.cube_fd_all, ctr, x.goals, 3);
    } else if (sc == QS_SCENARIO_RUN_AWAY) {            // run_away.py:15-27: drones 0 and 1 get the goals of two random others
        const int on_tape = QS_ON_TAPE(key) ? 1 : 0;   // np.random.randint(low=1, high=N, size=2): a tape holds the values themselves
        const int g0 = (1 - on_tape) + rng_index<real>(key, QS_SITE_SCEN, 44, N - 1),
            g1 = (1 - on_tape) + rng_index<real>(key, QS_SITE_SCEN, 45, N - 1);
        for (int q = 0; q < 3; ++q) x.goals[0 * 3 + q] = x.goals[g0 * 3 + q];
        for (int q = 0; q < 3; ++q) x.goals[1 * 3 + q] = x.goals[g1 * 3 + q];
    } else {                                            // swap_goals.py:13-24 / o_swap_goals.py:14-25
        shuffle_rows<real>(key, x.goals, 3, N, 0);
    }
}

// The same rewrites done by the environment's LANES (lane i makes goal row i; formation_rows_wave, qs_device.h) - bit-identical to
// scenario_step_serial (same arithmetic per row, same summation order, same draws).  Why it exists: dynamic_formations rebuilds all goal
// rows on EVERY step and the others at their periods, so with a few hundred environments some lane 0 of the batch was doing serial trig /
// Fisher-Yates on every control step, and the slowest workgroup sets the duration of the launch (`mix`, 1024 x 8: 42 us per step with
// the serial form against 8 us for the same shape on a static scenario; profiles/r03_final_batched_env_host.json).
__device__ __forceinline__ bool scen_step_wave_ok(int sc, int N) {
    return N >= 3 && (sc == QS_SCENARIO_DYNAMIC_FORMATIONS || sc == QS_SCENARIO_DYNAMIC_DIFF_GOAL || sc == QS_SCENARIO_SWAP_GOALS
        || sc == QS_SCENARIO_O_SWAP_GOALS ||
                      (sc == QS_SCENARIO_SWARM_VS_SWARM && N / 2 >= 3));
}
// Called by all lanes of a wave; `on` (uniform per environment): this environment takes the wave path at this tick.  scr: N ints of LDS.
template <typename real>
__device__ __forceinline__ void scenario_step_wave(const Consts<real> &c, const RngKey &key, const ScenCtx<real> &x, int sc, int *scr,
    int i, bool on) {
    const int N = x.N;
    Formation<real> F;
    F.f = 0; F.per_layer = 1; F.lo = F.hi = F.size = F.layer_dist = (real)0;
    real c1[3] = {0, 0, 0}, c2[3] = {0, 0, 0}, speed = 0;
    int inc = 0;
    bool build = false, shuffle = false;
    const bool svs = sc == QS_SCENARIO_SWARM_VS_SWARM, ddg = sc == QS_SCENARIO_DYNAMIC_DIFF_GOAL,
        dyf = sc == QS_SCENARIO_DYNAMIC_FORMATIONS;
    if (on) {   // every lane of the environment derives the new scenario state (identical keys => identical values)
        if (svs) {                                         // swarm_vs_swarm.py:59-79
            for (int q = 0; q < 3; ++q) { c1[q] = x.sr[SR_C2 + q]; c2[q] = x.sr[SR_C1 + q]; }
            update_formation<real>(sc, key, 32, N, F);
            build = shuffle = true;
        } else if (ddg) {                                  // dynamic_diff_goal.py:8-34
            load_formation<real>(x, F);
            real box = c.spawn_box, xy[2];
            rng_uniform<real, 2>(key, QS_SITE_SCEN, 40, 0, 0, -box, box, xy);
            c1[0] = xy[0]; c1[1] = xy[1]; c1[2] = get_z_value<real>(c, key, F, N, 41);
            update_formation<real>(sc, key, 32, N, F);
            build = shuffle = true;
        } else if (dyf) {                                  // dynamic_formations.py:16-35
            load_formation<real>(x, F);
            inc = x.si[SI_INCREASE];
            speed = x.sr[SR_SPEED];
            if (F.size <= -F.hi) { inc = 1; speed = rng_uniform1<real>(key, QS_SITE_SCEN, 292, 0, 0, (real)1, (real)3); }
            else if (F.size >= F.hi) { inc = 0; speed = rng_uniform1<real>(key, QS_SITE_SCEN, 292, 0, 0, (real)1, (real)3); }
            if (inc) F.size += (real)0.001 * speed; else F.size -= (real)0.001 * speed;
            for (int q = 0; q < 3; ++q) c1[q] = x.sr[SR_C1 + q];
            build = true;
        } else shuffle = true;                             // swap_goals.py:13-24 / o_swap_goals.py:14-25
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // every lane has read the old state before lane 0 replaces it
    if (on && i == 0) {
        if (svs) { for (int q = 0; q < 3; ++q) { x.sr[SR_C1 + q] = c1[q]; x.sr[SR_C2 + q] = c2[q]; } store_formation<real>(x, F); }
        else if (ddg) { for (int q = 0; q < 3; ++q) x.sr[SR_C1 + q] = c1[q]; store_formation<real>(x, F); }
        else if (dyf) { x.si[SI_INCREASE] = inc; x.sr[SR_SPEED] = speed; x.sr[SR_SIZE] = F.size; }
    }
    const int n1 = N / 2;
    const bool second = svs && i >= n1;
    // the three cube sizes by VALUE (a select between their addresses would pin the whole constant block in scratch memory: 552 bytes per
    // lane and a 22 us step instead of 9, tools/spec_resources.py)
    int fd0 = c.cube_fd[0], fd1 = c.cube_fd[1], fda = c.cube_fd_all;
    asm volatile("" : "+v"(fd0), "+v"(fd1), "+v"(fda));
    formation_rows_wave<real>(key, F, build, svs ? (second ? N - n1 : n1) : N, svs ? (second ? fd1 : fd0) : fda, second ? c2 : c1,
                              second ? i - n1 : i, second ? n1 : 0, shuffle, second ? 256 : 0, x.goals, scr, i, on);
}

// lane-local part of scenario.step(): scenarios whose drones all share one goal, computed redundantly by every drone
// of the env (identical RNG keys => identical values); `persist` (drone 0) writes the env's state back to LDS.
template <typename real>
__device__ QS_COLD void scenario_step_local(const Consts<real> &c, const RngKey &key, const ScenCtx<real> &x, int sc, int period,
    int tick, bool persist,
                                    real goal[3]) {
    if (sc == QS_SCENARIO_DYNAMIC_SAME_GOAL) {           // dynamic_same_goal.py:16-29 (formation size 0: every goal = the centre)
        if (period > 0 && tick > 0 && imod_small(tick, period) == 0) {
            real box = c.spawn_box, xy[2];
            rng_uniform<real, 2>(key, QS_SITE_SCEN, 40, 0, 0, -box, box, xy);
            real z = M<real>::fmax((real)0.25,
                rng_uniform1<real>(key, QS_SITE_SCEN, 41, 0, 0, (real)-0.5 * box, (real)0.5 * box) + (real)2);
            // generate_goals with size 0: goal = (0*cos, 0*sin, 0) + centre
            goal[0] = (real)0 + xy[0]; goal[1] = (real)0 + xy[1]; goal[2] = (real)0 + z;
            if (persist) { x.sr[SR_C1] = xy[0]; x.sr[SR_C1 + 1] = xy[1]; x.sr[SR_C1 + 2] = z; }
        }
    } else if (sc == QS_SCENARIO_EP_LISSAJOUS3D) {       // ep_lissajous3D.py:8-26
        real t = (real)tick / (real)c.control_freq;
        goal[0] = (real)0.03 * M<real>::sin(t) + goal[0];
        goal[1] = (real)0.01 * M<real>::sin((real)2 * t + (real)90) + goal[1];
        goal[2] = (real)0.01 * M<real>::cos((real)2 * t + (real)90) + goal[2];
    } else if (sc == QS_SCENARIO_EP_RAND_BEZIER || sc == QS_SCENARIO_O_EP_RAND_BEZIER) {
        // ep_rand_bezier.py:8-48 / obstacles/o_ep_rand_bezier.py:16-55 (6 s legs, <= 5 m, goal height in [1.5, 3])
        const bool obst = sc == QS_SCENARIO_O_EP_RAND_BEZIER;
        const int control_steps = (obst ? 6 : 5) * c.control_freq, t = tick % control_steps;
        real fsz = x.sr[SR_SIZE];
        real room[3] = {c.room_hi[0] - c.room_lo[0] - fsz, c.room_hi[1] - c.room_lo[1] - fsz, c.room_hi[2] - c.room_lo[2] - fsz};
        real mx = M<real>::fmax(room[0], M<real>::fmax(room[1], room[2])), max_dist = M<real>::fmin(obst ? (real)5 : (real)30, mx),
            min_dist = max_dist / (real)2;
        real bez[9];
        for (int q = 0; q < 9; ++q) bez[q] = x.sr[SR_BEZ + q];
        if (tick % control_steps == 0 || tick == 1) {
            real low[3] = {-room[0] / (real)2, -room[1] / (real)2, obst ? (real)1.5 : (real)0},
                high[3] = {room[0] / (real)2, room[1] / (real)2, obst ? (real)3 : room[2]};
            real np_[3][2];
            for (int it = 0; it < 100000; ++it) {
                real u[6];
                for (int k = 0; k < 6; ++k) { int ax = k % 3;
                    u[k] = rng_uniform1<real>(key, QS_SITE_SCEN, 300 + 8 * it + k, 0, 0, -high[ax], high[ax]); }
                int lo_i = (int)floorf((float)min_dist), hi_i = (int)max_dist + 1;   // np.random.randint truncates a float low
                // (tape: the randint value itself)
                int r = (QS_ON_TAPE(key) ? 0 : lo_i) + rng_index<real>(key, QS_SITE_SCEN, 300 + 8 * it + 6, hi_i - lo_i);
                bool ok = true;
                for (int col = 0; col < 2; ++col) {
                    real v[3] = {u[0 + col], u[2 + col], u[4 + col]}, n = norm3<real>(v);
                    for (int row = 0; row < 3; ++row) {
                        np_[row][col] = goal[row] + v[row] * (real)r / n;
                        if (!(np_[row][col] > low[row] + (real)0.5) || !(np_[row][col] < high[row] - (real)0.5)) ok = false;
                    }
                }
                if (ok) break;
            }
            for (int row = 0; row < 3; ++row) { bez[row] = goal[row]; bez[3 + row] = np_[row][0]; bez[6 + row] = np_[row][1]; }
            if (persist) { for (int q = 0; q < 9; ++q) x.sr[SR_BEZ + q] = bez[q]; x.si[SI_BEZ_VALID] = 1; }
        }
        if (tick % control_steps != 0 && tick > 1) {     // quadratic Bezier curve at linspace(0,1,control_steps)[t]
            real sv = (real)t / (real)(control_steps - 1), om = (real)1 - sv;
            for (int row = 0; row < 3; ++row) goal[row] = om * om * bez[row] + (real)2 * om * sv * bez[3 + row] + sv * sv * bez[6 + row];
        }
    } else if (sc == QS_SCENARIO_O_DYNAMIC_SAME_GOAL) {  // o_dynamic_same_goal.py:17-28
        if ((period > 0 && imod_small(tick, period) == 0) || tick == 1) {
            real endp[3] = {x.sr[SR_END], x.sr[SR_END + 1], x.sr[SR_END + 2]}, ng[3];
            for (int it = 0; it < 100000; ++it) {
                pos_obst_map_1<real>(c, key, x.omap, 5000 + 2 * it, ng);
                real df[3] = {endp[0] - ng[0], endp[1] - ng[1], endp[2] - ng[2]};
                if (!(norm3<real>(df) > (real)4)) break;
            }
            for (int q = 0; q < 3; ++q) goal[q] = ng[q];
            if (persist) for (int q = 0; q < 3; ++q) x.sr[SR_END + q] = ng[q];
        }
    }
}

}  // namespace qs
