/*
 * quadswarm_oracle.c - CPU oracle of the QuadSwarm env stepper.   *** TEST INFRASTRUCTURE ***
 *
 * Plain C99, float64, sequential restatement of the reference hot path.  Every function cites the
 * reference file:line it follows (paths relative to /root/reference/gym_art/quadrotor_multi/).
 * See quadswarm_oracle.h for the parity status (PINNED against fixtures captured from the reference).
 *
 * Two random sources:
 *   - tape mode  : draws are popped from a sequential tape recorded from the reference, in the
 *                  reference's exact call order (including draws whose value is unused);
 *   - philox mode: the counter-based stream specified in include/quadswarm.h, bit-identical
 *                  uniforms to the HIP stepper.
 */
#include "quadswarm_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define GRAV 9.81
#define EPS_DYN 1e-6 /* quadrotor_dynamics.py:13 */
#define EPS_COL 1e-5 /* quad_utils.py:10 */
#define MAXN QS_MAX_AGENTS
#define MAXG (2 * QS_MAX_AGENTS + 8)
#define PI 3.141592653589793

enum { F_ON_FLOOR = 1u << 0, F_CRASH_FLOOR = 1u << 1, F_CRASH_WALL = 1u << 2, F_CRASH_CEIL = 1u << 3,
       F_PREV_WALL = 1u << 4, F_PREV_CEIL = 1u << 5, F_PREV_ROOM = 1u << 6, F_PREV_OBST = 1u << 7,
       F_REACHED = 1u << 8, F_COL_AGENT_OK = 1u << 9, F_COL_OBST_OK = 1u << 10, F_OMEGA_F32 = 1u << 11 };

typedef struct {
    double pos[3], vel[3], rot[9], omega[3], acc[3];
    double rot_damp[4], cmds_damp[4], ou[4];
    double since_last_svd;
    int32_t svd_count;
    uint32_t flags;
    double goal[3], spawn_point[3];
    double *dist_hist; /* distance_to_goal list (quadrotor_multi.py:542) */
    int32_t dist_len;
    double time_remain;
} drone_t;

struct qso_env {
    qs_config c;
    int32_t env_id;
    uint32_t step_ctr;
    int32_t tick;
    int32_t obs_dim, self_dim;
    drone_t d[MAXN];
    double mpos[MAXN][3], mvel[MAXN][3]; /* QuadrotorEnvMulti.pos / .vel (quadrotor_multi.py:84-85) */
    uint64_t prev_pair[MAXN];
    qso_info info;
    /* obstacles; cur_*: this episode's count / size / density (--quads_domain_random, quad_experience_replay.py:106-118,:191-206) */
    int32_t cur_M;
    double cur_size, cur_density;
    double obst_xy[QS_MAX_OBSTACLES][2];
    uint8_t obst_map[64][64];
    double *cell_centers; /* [L*W][2] */
    /* scenario */
    double goals[MAXG][3];
    int32_t num_goals;
    double spawn_points[MAXN][3];
    int32_t have_spawn_points;
    int32_t formation, per_layer;
    double form_lo, form_hi, form_size, layer_dist;
    double center1[3], center2[3];
    int32_t control_step_for_sec;
    int32_t scen, scen_constructed, duration_step, increase, bez_valid;
    double speed, bez[9], end_point[3], approach_metric;
    /* rng */
    const double *tape;
    int64_t tape_n, tape_i;
    int32_t np126;   /* qso_set_numpy126_quirk: NumPy-1.26 value-based casting of the omega damping term (App. D), off by default */
};

/* ------------------------------------------------------------------------------------------ */
/* Philox4x32-10 (Salmon et al. 2011, "Parallel random numbers: as easy as 1, 2, 3")           */
/* ------------------------------------------------------------------------------------------ */
void qso_philox4x32(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
    uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3], k0 = key[0], k1 = key[1];
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

static double u01(uint32_t x) { return ((double)(x >> 9) + 0.5) * (1.0 / 8388608.0); }

static void philox_group(const qso_env *e, int site, int slot, int i, int j, uint32_t w[4]) {
    uint32_t ctr[4] = {(uint32_t)e->env_id, e->step_ctr, (uint32_t)site | ((uint32_t)slot << 8),
                       (uint32_t)i | ((uint32_t)j << 16)};
    uint32_t key[2] = {(uint32_t)(e->c.seed & 0xffffffffu), (uint32_t)(e->c.seed >> 32)};
    qso_philox4x32(ctr, key, w);
}

static double tape_pop(qso_env *e) {
    if (e->tape_i >= e->tape_n) { e->info.tape_underrun = 1; return 0.0; }
    return e->tape[e->tape_i++];
}
static void tape_skip(qso_env *e, int n) { if (e->tape) for (int k = 0; k < n; ++k) (void)tape_pop(e); }

/* n <= 4 normal(loc, scale) draws */
static void rng_normal(qso_env *e, int site, int slot, int i, int j, int n, double loc, double scale, double *out) {
    if (e->tape) { for (int k = 0; k < n; ++k) out[k] = tape_pop(e); return; }
    uint32_t w[4]; philox_group(e, site, slot, i, j, w);
    double z[4];
    for (int p = 0; p < 2; ++p) {
        double r = sqrt(-2.0 * log(u01(w[2 * p]))), th = 2.0 * PI * u01(w[2 * p + 1]);
        z[2 * p] = r * cos(th); z[2 * p + 1] = r * sin(th);
    }
    for (int k = 0; k < n; ++k) out[k] = loc + scale * z[k];
}
/* n <= 4 uniform(lo, hi) draws */
static void rng_uniform(qso_env *e, int site, int slot, int i, int j, int n, double lo, double hi, double *out) {
    if (e->tape) { for (int k = 0; k < n; ++k) out[k] = tape_pop(e); return; }
    uint32_t w[4]; philox_group(e, site, slot, i, j, w);
    for (int k = 0; k < n; ++k) out[k] = lo + (hi - lo) * u01(w[k]);
}
static double rng_uniform1(qso_env *e, int site, int slot, int i, int j, double lo, double hi) {
    double v; rng_uniform(e, site, slot, i, j, 1, lo, hi, &v); return v;
}

/* ------------------------------------------------------------------------------------------ */
/* small linear algebra                                                                        */
/* ------------------------------------------------------------------------------------------ */
static double norm3(const double v[3]) { return sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]); }
static double dot3(const double a[3], const double b[3]) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
static double clipd(double x, double lo, double hi) { return x < lo ? lo : (x > hi ? hi : x); }
static void matmul3(const double a[9], const double b[9], double o[9]) {
    double t[9];
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) t[r * 3 + c] = a[r * 3] * b[c] + a[r * 3 + 1] * b[3 + c] + a[r * 3 + 2] * b[6 + c];
    memcpy(o, t, sizeof t);
}
static void yaw_rot(double theta, double r[9]) {
    double c = cos(theta), s = sin(theta);
    r[0] = c; r[1] = -s; r[2] = 0; r[3] = s; r[4] = c; r[5] = 0; r[6] = 0; r[7] = 0; r[8] = 1;
}

/* nearest rotation U V^T of np.linalg.svd(rot) (quadrotor_dynamics.py:548-551): the orthogonal polar
 * factor, obtained by Newton's iteration X <- (X + X^-T)/2 which converges quadratically to it. */
void qso_polar_rotation(const double r[9], double out[9]) {
    double x[9]; memcpy(x, r, sizeof x);
    for (int it = 0; it < 20; ++it) {
        double c00 = x[4] * x[8] - x[5] * x[7], c01 = x[5] * x[6] - x[3] * x[8], c02 = x[3] * x[7] - x[4] * x[6];
        double c10 = x[2] * x[7] - x[1] * x[8], c11 = x[0] * x[8] - x[2] * x[6], c12 = x[1] * x[6] - x[0] * x[7];
        double c20 = x[1] * x[5] - x[2] * x[4], c21 = x[2] * x[3] - x[0] * x[5], c22 = x[0] * x[4] - x[1] * x[3];
        double det = x[0] * c00 + x[1] * c01 + x[2] * c02;
        double cof[9] = {c00, c01, c02, c10, c11, c12, c20, c21, c22}; /* X^-T = cof/det */
        double delta = 0;
        for (int k = 0; k < 9; ++k) {
            double nx = 0.5 * (x[k] + cof[k] / det);
            delta += fabs(nx - x[k]); x[k] = nx;
        }
        if (delta < 1e-16) break;
    }
    memcpy(out, x, sizeof x);
}

/* ------------------------------------------------------------------------------------------ */
/* observation sizes: quad_utils.py:30-44, quadrotor_single.py:311-316                         */
/* ------------------------------------------------------------------------------------------ */
static int self_obs_dim(int obs_repr) { return obs_repr == QS_OBS_XYZ_VXYZ_R_OMEGA ? 18 : (obs_repr == QS_OBS_XYZ_VXYZ_R_OMEGA_FLOOR ? 19 : 24); }
int qso_obs_dim(const qs_config *c) { return self_obs_dim(c->obs_repr) + 6 * c->num_neighbors + (c->use_obstacles ? 9 : 0); }

/* ------------------------------------------------------------------------------------------ */
/* per-drone dynamics                                                                          */
/* ------------------------------------------------------------------------------------------ */

/* calculate_torque_integrate_rotations_and_update_omega, quadrotor_dynamics.py:498-566
 * (numpy twin: step1 :225-329).  Returns sum of thrusts (thrust z). */
static double torque_rot_omega(qso_env *e, drone_t *d, const double cmds_in[4], const double thr_noise[4]) {
    const qs_config *c = &e->c;
    double dt = c->dt, thrusts[4], torque[3] = {0, 0, 0};
    double tq[4][3];
    for (int m = 0; m < 4; ++m) {
        double cmd = clipd(cmds_in[m], 0.0, 1.0);
        double tau = c->motor_tau_up;
        if (cmd < d->cmds_damp[m]) tau = c->motor_tau_down;
        if (tau > 1.0) tau = 1.0;
        double trot = pow(cmd, 0.5);
        d->rot_damp[m] = tau * (trot - d->rot_damp[m]) + d->rot_damp[m];
        d->cmds_damp[m] = d->rot_damp[m] * d->rot_damp[m];
        double tn = cmd * thr_noise[m];
        d->cmds_damp[m] = clipd(d->cmds_damp[m] + tn, 0.0, 1.0);
        double w = d->cmds_damp[m];
        thrusts[m] = c->thrust_max[m] * ((1 - c->motor_linearity) * (w * w) + c->motor_linearity * w);
        for (int k = 0; k < 3; ++k) tq[m][k] = c->prop_cross[m][k] * thrusts[m];
        tq[m][2] += c->torque_max[m] * c->prop_ccw[m] * d->cmds_damp[m];
    }
    for (int k = 0; k < 3; ++k) torque[k] = ((tq[0][k] + tq[1][k]) + tq[2][k]) + tq[3][k];
    double thrust_z = ((thrusts[0] + thrusts[1]) + thrusts[2]) + thrusts[3];

    /* rotational dynamics: Rodrigues (:535-544) */
    double *R = d->rot, *om = d->omega;
    double wv[3];
    for (int r = 0; r < 3; ++r) wv[r] = R[r * 3] * om[0] + R[r * 3 + 1] * om[1] + R[r * 3 + 2] * om[2];
    double wn = norm3(wv);
    if (wn != 0) {
        double K[9] = {0, -wv[2] / wn, wv[1] / wn, wv[2] / wn, 0, -wv[0] / wn, -wv[1] / wn, wv[0] / wn, 0};
        double ang = wn * dt, s = sin(ang), cc = 1.0 - cos(ang), KK[9], dR[9];
        matmul3(K, K, KK);
        for (int k = 0; k < 9; ++k) dR[k] = ((k % 4 == 0) ? 1.0 : 0.0) + s * K[k] + cc * KK[k];
        matmul3(dR, R, R);
    }
    /* rare SVD re-orthogonalisation (:546-551) */
    d->since_last_svd += dt;
    d->svd_count += 1;
    if (d->since_last_svd > 0.5) {
        qso_polar_rotation(R, R);
        d->since_last_svd = 0;
        d->svd_count = 0;
    }
    /* omega update (:555-560) */
    double Iw[3] = {c->inertia[0] * om[0], c->inertia[1] * om[1], c->inertia[2] * om[2]};
    double a[3] = {-om[0], -om[1], -om[2]};
    double cr[3] = {a[1] * Iw[2] - a[2] * Iw[1], a[2] * Iw[0] - a[0] * Iw[2], a[0] * Iw[1] - a[1] * Iw[0]};
    /* NB (App. D): in the numpy path omega is a float32 array right after reset / a floor crash.  Under the
     * reference's pinned NumPy 1.26 (value-based casting) that makes (1-damp)*dt evaluate in float32 for ONE
     * sub-step (2e-8 relative); under NumPy >= 2 (the run most golden fixtures come from) damp_omega_quadratic is
     * a float64 scalar and the expression stays float64.  Default: the fixtures' semantics (no dt quirk); the
     * 1.26 behaviour is a switch (below).  The in-place `omega += ...` of the collision responses on such a
     * float32 array IS kept (add_omega). */
    /* qso_set_numpy126_quirk(e, 1): the reference AS PINNED (setup.py:14: numpy 1.26.4).  There `self.damp_omega_quadratic` (an np.float64
     * scalar) times `self.omega ** 2` (a float32 array for the one sub-step after set_state / a floor crash, quadrotor_dynamics.py:188,:430)
     * is a FLOAT32 array by value-based casting, and so are its clip, `1.0 - ...` and the product with the Python float `dt` (:324-325): the
     * factor (1 - damp) * dt is rounded to float32 (dt -> 0.004999999888) before it meets the float64 `omega_dot`.  Pinned by the fixture
     * `c1_single_numpy_np126` (captured under NumPy 2 with that scalar made a Python float, which NEP 50 treats the same way). */
    const int quirk = e->np126 && c->floor_mode == QS_FLOOR_NUMPY && (d->flags & F_OMEGA_F32);
    for (int k = 0; k < 3; ++k) {
        double od = (1.0 / c->inertia[k]) * (cr[k] + torque[k]);
        if (quirk) {
            const float o32 = (float)om[k], sq = o32 * o32;
            float damp32 = (float)c->damp_omega_quadratic * sq;
            damp32 = damp32 < 0.0f ? 0.0f : (damp32 > 1.0f ? 1.0f : damp32);
            const float f32 = (1.0f - damp32) * (float)dt;
            om[k] = clipd(om[k] + (double)f32 * od, -c->omega_max, c->omega_max);
            continue;
        }
        double damp = clipd(c->damp_omega_quadratic * (om[k] * om[k]), 0.0, 1.0);
        om[k] = clipd(om[k] + (1.0 - damp) * dt * od, -c->omega_max, c->omega_max);
    }
    d->flags &= ~F_OMEGA_F32;
    /* position (:563) */
    for (int k = 0; k < 3; ++k) d->pos[k] = d->pos[k] + dt * d->vel[k];
    return thrust_z;
}

/* floor_interaction_numba quadrotor_dynamics.py:570-639 / floor_interaction :389-457 */
static void floor_interaction(qso_env *e, int i, int substep, double thrust_z) {
    const qs_config *c = &e->c;
    drone_t *d = &e->d[i];
    int numpy_mode = c->floor_mode == QS_FLOOR_NUMPY;
    double thr = numpy_mode ? 0.05 : c->arm;
    double *R = d->rot;
    d->flags &= ~F_CRASH_FLOOR;
    double force[3] = {R[2] * thrust_z, R[5] * thrust_z, R[8] * thrust_z};
    if (d->pos[2] <= thr) {
        d->pos[2] = thr;
        if (d->flags & F_ON_FLOOR) {
            double theta = atan2(R[3], R[0] + EPS_DYN);
            yaw_rot(theta, R);
            double fr = 0.6 * (c->mass * GRAV - force[2]);
            double vn = norm3(d->vel);
            int is_static = numpy_mode ? (vn == 0.0) : (vn < EPS_DYN);
            if (is_static) {
                double fxy = sqrt(force[0] * force[0] + force[1] * force[1]);
                fxy = fmax(fxy - fr, 0.0);
                if (fxy == 0.0) { force[0] = 0; force[1] = 0; }
                else {
                    double ang = atan2(force[1], force[0]);
                    force[0] = fxy * cos(ang); force[1] = fxy * sin(ang);
                }
            } else {
                double ang = numpy_mode ? atan2(-1.0 * d->vel[1], -1.0 * d->vel[0]) : atan2(d->vel[1], d->vel[0]);
                force[0] = force[0] - cos(ang) * fr;
                force[1] = force[1] - sin(ang) * fr;
            }
        } else {
            d->flags |= F_ON_FLOOR | F_CRASH_FLOOR;
            for (int k = 0; k < 3; ++k) { d->vel[k] = 0; d->omega[k] = 0; }
            double theta = atan2(R[3], R[0] + EPS_DYN);
            if (R[8] < 0) {
                if (!numpy_mode) {
                    theta = rng_uniform1(e, QS_SITE_FLOOR_YAW, substep * 64, i, 0, -PI, PI);
                    yaw_rot(theta, R);
                } else { /* randyaw() until roughly facing the origin (:434-437) */
                    double xy[3] = {-d->pos[0], -d->pos[1], 0}, n = norm3(xy);
                    if (n >= 0.00001) { xy[0] /= n; xy[1] /= n; }
                    for (int t = 0; t < 64; ++t) {
                        theta = rng_uniform1(e, QS_SITE_FLOOR_YAW, substep * 64 + t, i, 0, -PI, PI);
                        yaw_rot(theta, R);
                        if (!(R[0] * xy[0] + R[3] * xy[1] < 0.5)) break;
                    }
                }
            } else {
                yaw_rot(theta, R);
            }
            if (numpy_mode) d->flags |= F_OMEGA_F32; /* set_state casts omega to float32 (:188,:430) */
            for (int m = 0; m < 4; ++m) { d->cmds_damp[m] = 0; d->rot_damp[m] = 0; }
        }
        d->acc[0] = 0.0 + (1.0 / c->mass) * force[0];
        d->acc[1] = 0.0 + (1.0 / c->mass) * force[1];
        d->acc[2] = -GRAV + (1.0 / c->mass) * force[2];
        d->acc[2] = fmax(0.0, d->acc[2]);
    } else {
        d->flags &= ~F_ON_FLOOR;
        d->acc[0] = 0.0 + (1.0 / c->mass) * force[0];
        d->acc[1] = 0.0 + (1.0 / c->mass) * force[1];
        d->acc[2] = -GRAV + (1.0 / c->mass) * force[2];
    }
}

/* one physics sub-step: step1_numba quadrotor_dynamics.py:348-383 (numpy: step1 :225-346) */
static void substep(qso_env *e, int i, int sub, const double cmds[4], const double thr_noise[4]) {
    const qs_config *c = &e->c;
    drone_t *d = &e->d[i];
    double thrust_z = torque_rot_omega(e, d, cmds, thr_noise);
    double before[3] = {d->pos[0], d->pos[1], d->pos[2]};
    for (int k = 0; k < 3; ++k) d->pos[k] = clipd(d->pos[k], c->room_lo[k], c->room_hi[k]);
    d->flags &= ~(F_CRASH_WALL | F_CRASH_CEIL);
    if (before[0] != d->pos[0] || before[1] != d->pos[1]) d->flags |= F_CRASH_WALL;
    if (before[2] > d->pos[2]) d->flags |= F_CRASH_CEIL;
    floor_interaction(e, i, sub, thrust_z);
    /* compute_velocity_and_acceleration :643-649 (accelerometer is not part of any obs_repr) */
    for (int k = 0; k < 3; ++k) d->vel[k] = (1.0 - c->vel_damp) * d->vel[k] + c->dt * d->acc[k];
}

/* RawControl.step quadrotor_control.py:53-57 + QuadrotorDynamics.step quadrotor_dynamics.py:208-214
 * + OUNoise.noise quad_utils.py:275-279 */
static void dynamics_step(qso_env *e, int i, const double action[4]) {
    const qs_config *c = &e->c;
    drone_t *d = &e->d[i];
    double cmds[4], z[4];
    for (int m = 0; m < 4; ++m) cmds[m] = 0.5 * (clipd(action[m], -1.0, 1.0) + 1.0);
    rng_normal(e, QS_SITE_OU, 0, i, 0, 4, 0.0, 1.0, z);
    for (int m = 0; m < 4; ++m) {
        double x = d->ou[m];
        double dx = c->ou_theta * (0.0 - x) + c->thrust_noise_sigma * z[m];
        d->ou[m] = x + dx;
    }
    for (int s = 0; s < c->sim_steps; ++s) substep(e, i, s, cmds, d->ou);
}

/* compute_reward_weighted quadrotor_single.py:34-92 */
static double reward_single(qso_env *e, int i, const double action[4], double *ri /* QS_RI_COUNT or NULL */) {
    const qs_config *c = &e->c;
    drone_t *d = &e->d[i];
    double dt = c->dt;
    double diff[3] = {d->goal[0] - d->pos[0], d->goal[1] - d->pos[1], d->goal[2] - d->pos[2]};
    double cost_pos_raw = norm3(diff), cost_pos = c->rew_coeff[QS_REW_POS] * cost_pos_raw;
    double cost_effort_raw = sqrt(action[0] * action[0] + action[1] * action[1] + action[2] * action[2] + action[3] * action[3]);
    double cost_effort = c->rew_coeff[QS_REW_EFFORT] * cost_effort_raw;
    int on_floor = (d->flags & F_ON_FLOOR) != 0;
    double cost_orient_raw = on_floor ? 1.0 : -d->rot[8];
    double cost_orient = c->rew_coeff[QS_REW_ORIENT] * cost_orient_raw;
    double cost_spin_raw = pow(d->omega[0] * d->omega[0] + d->omega[1] * d->omega[1] + d->omega[2] * d->omega[2], 0.5);
    double cost_spin = c->rew_coeff[QS_REW_SPIN] * cost_spin_raw;
    double cost_crash_raw = on_floor ? 1.0 : 0.0, cost_crash = c->rew_coeff[QS_REW_CRASH] * cost_crash_raw;
    double reward = -dt * ((((cost_pos + cost_effort) + cost_crash) + cost_orient) + cost_spin);
    if (ri) {
        ri[QS_RI_REW_MAIN] = dt * -cost_pos; ri[QS_RI_REW_POS] = dt * -cost_pos; ri[QS_RI_REW_ACTION] = dt * -cost_effort;
        ri[QS_RI_REW_CRASH] = dt * -cost_crash; ri[QS_RI_REW_ORIENT] = dt * -cost_orient; ri[QS_RI_REW_SPIN] = dt * -cost_spin;
        ri[QS_RI_RAW_MAIN] = dt * -cost_pos_raw; ri[QS_RI_RAW_POS] = dt * -cost_pos_raw; ri[QS_RI_RAW_ACTION] = dt * -cost_effort_raw;
        ri[QS_RI_RAW_CRASH] = dt * -cost_crash_raw; ri[QS_RI_RAW_ORIENT] = dt * -cost_orient_raw; ri[QS_RI_RAW_SPIN] = dt * -cost_spin_raw;
    }
    if (isnan(reward) || !isfinite(reward)) e->info.nan_reward = 1;
    return reward;
}

/* rot2quat sensor_noise.py:34-63 */
static void rot2quat(const double r[9], double q[4]) {
    double trace = r[0] + r[4] + r[8];
    if (trace > 0) {
        double S = pow(trace + 1.0, 0.5) * 2;
        q[0] = 0.25 * S; q[1] = (r[7] - r[5]) / S; q[2] = (r[2] - r[6]) / S; q[3] = (r[3] - r[1]) / S;
    } else if (r[0] > r[4] && r[0] > r[8]) {
        double S = pow(1.0 + r[0] - r[4] - r[8], 0.5) * 2;
        q[0] = (r[7] - r[5]) / S; q[1] = 0.25 * S; q[2] = (r[1] + r[3]) / S; q[3] = (r[2] + r[6]) / S;
    } else if (r[4] > r[8]) {
        double S = pow(1.0 + r[4] - r[0] - r[8], 0.5) * 2;
        q[0] = (r[2] - r[6]) / S; q[1] = (r[1] + r[3]) / S; q[2] = 0.25 * S; q[3] = (r[5] + r[7]) / S;
    } else {
        double S = pow(1.0 + r[8] - r[0] - r[4], 0.5) * 2;
        q[0] = (r[3] - r[1]) / S; q[1] = (r[2] + r[6]) / S; q[2] = (r[5] + r[7]) / S; q[3] = 0.25 * S;
    }
}

/* state_xyz_vxyz_R_omega[_floor|_wall] get_state.py:6-72 + SensorNoise.add_noise[_numba]
 * sensor_noise.py:112-218,:235-261 + quat_from_small_angle :11-23 + quatXquat quad_utils.py:148-159
 * + quat2R :133-138.  `pass` = 0 for the per-drone obs, 1 for the refresh after interactions. */
static void self_obs(qso_env *e, int i, int pass, double *o) {
    const qs_config *c = &e->c;
    drone_t *d = &e->d[i];
    double p[3], v[3], w[3], R[9];
    if (!c->sense_noise) {
        memcpy(p, d->pos, sizeof p); memcpy(v, d->vel, sizeof v); memcpy(w, d->omega, sizeof w); memcpy(R, d->rot, sizeof R);
    } else {
        double n3[3], u3[3] = {0, 0, 0}, th[3], thu[3] = {0, 0, 0};
        rng_normal(e, QS_SITE_SENS_POS_N, pass, i, 0, 3, 0.0, c->pos_norm_std, n3);
        if (e->tape || c->pos_unif_range != 0) rng_uniform(e, QS_SITE_SENS_POS_U, pass, i, 0, 3, -c->pos_unif_range, c->pos_unif_range, u3);
        for (int k = 0; k < 3; ++k) p[k] = d->pos[k] + n3[k] + u3[k];
        rng_normal(e, QS_SITE_SENS_VEL_N, pass, i, 0, 3, 0.0, c->vel_norm_std, n3);
        u3[0] = u3[1] = u3[2] = 0;
        if (e->tape || c->vel_unif_range != 0) rng_uniform(e, QS_SITE_SENS_VEL_U, pass, i, 0, 3, -c->vel_unif_range, c->vel_unif_range, u3);
        for (int k = 0; k < 3; ++k) v[k] = d->vel[k] + n3[k] + u3[k];
        rng_normal(e, QS_SITE_SENS_OMEGA_N, pass, i, 0, 3, 0.0, c->gyro_noise_density, n3);
        for (int k = 0; k < 3; ++k) w[k] = d->omega[k] + n3[k];
        th[0] = th[1] = th[2] = 0;
        if (e->tape || c->quat_norm_std != 0) rng_normal(e, QS_SITE_SENS_THETA_N, pass, i, 0, 3, 0.0, c->quat_norm_std, th);
        if (e->tape || c->quat_unif_range != 0) rng_uniform(e, QS_SITE_SENS_THETA_U, pass, i, 0, 3, -c->quat_unif_range, c->quat_unif_range, thu);
        for (int k = 0; k < 3; ++k) th[k] = th[k] + thu[k];
        tape_skip(e, 6); /* accelerometer noise: drawn by the reference, not part of any obs_repr */
        /* quat_from_small_angle */
        double nt = norm3(th), qsq = nt * nt / 4.0, qt[4];
        if (qsq < 1) { qt[0] = pow(1 - qsq, 0.5); qt[1] = th[0] * 0.5; qt[2] = th[1] * 0.5; qt[3] = th[2] * 0.5; }
        else { double ww = 1.0 / pow(1 + qsq, 0.5), f = 0.5 * ww; qt[0] = ww; qt[1] = th[0] * f; qt[2] = th[1] * f; qt[3] = th[2] * f; }
        double qn = sqrt(qt[0] * qt[0] + qt[1] * qt[1] + qt[2] * qt[2] + qt[3] * qt[3]);
        for (int k = 0; k < 4; ++k) qt[k] /= qn;
        double q[4], nq[4];
        rot2quat(d->rot, q);
        nq[0] = q[0] * qt[0] - q[1] * qt[1] - q[2] * qt[2] - q[3] * qt[3];
        nq[1] = q[0] * qt[1] + q[1] * qt[0] - q[2] * qt[3] + q[3] * qt[2];
        nq[2] = q[0] * qt[2] + q[1] * qt[3] + q[2] * qt[0] - q[3] * qt[1];
        nq[3] = q[0] * qt[3] - q[1] * qt[2] + q[2] * qt[1] + q[3] * qt[0];
        double qw = nq[0], qx = nq[1], qy = nq[2], qz = nq[3];
        R[0] = 1.0 - 2 * qy * qy - 2 * qz * qz; R[1] = 2 * qx * qy - 2 * qz * qw; R[2] = 2 * qx * qz + 2 * qy * qw;
        R[3] = 2 * qx * qy + 2 * qz * qw; R[4] = 1.0 - 2 * qx * qx - 2 * qz * qz; R[5] = 2 * qy * qz - 2 * qx * qw;
        R[6] = 2 * qx * qz - 2 * qy * qw; R[7] = 2 * qy * qz + 2 * qx * qw; R[8] = 1.0 - 2 * qx * qx - 2 * qy * qy;
    }
    for (int k = 0; k < 3; ++k) { o[k] = p[k] - d->goal[k]; o[3 + k] = v[k]; o[15 + k] = w[k]; }
    for (int k = 0; k < 9; ++k) o[6 + k] = R[k];
    if (c->obs_repr == QS_OBS_XYZ_VXYZ_R_OMEGA_FLOOR) o[18] = p[2];
    else if (c->obs_repr == QS_OBS_XYZ_VXYZ_R_OMEGA_WALL)
        for (int k = 0; k < 3; ++k) { o[18 + k] = clipd(p[k] - c->room_lo[k], 0.0, 5.0); o[21 + k] = clipd(c->room_hi[k] - p[k], 0.0, 5.0); }
}

/* ------------------------------------------------------------------------------------------ */
/* per-env pieces                                                                              */
/* ------------------------------------------------------------------------------------------ */

/* neighborhood_indices quadrotor_multi.py:247-274 + extend_obs_space :233-245 (+ :212-231) */
static void neighbor_obs(qso_env *e, double *obs) {
    const qs_config *c = &e->c;
    int N = c->num_agents, K = c->num_neighbors;
    if (K <= 0) return;
    for (int i = 0; i < N; ++i) {
        int idx[MAXN], cnt = 0;
        for (int j = 0; j < N; ++j) if (j != i) idx[cnt++] = j;
        int sel[MAXN];
        if (K == N - 1) { for (int k = 0; k < K; ++k) sel[k] = idx[k]; }
        else {
            double metric[MAXN]; int order[MAXN];
            for (int k = 0; k < cnt; ++k) {
                int j = idx[k];
                double rp[3] = {e->mpos[j][0] - e->mpos[i][0], e->mpos[j][1] - e->mpos[i][1], e->mpos[j][2] - e->mpos[i][2]};
                double rv[3] = {e->mvel[j][0] - e->mvel[i][0], e->mvel[j][1] - e->mvel[i][1], e->mvel[j][2] - e->mvel[i][2]};
                double rd = fmax(norm3(rp), 0.01);
                metric[k] = rd + ((rp[0] / rd) * rv[0] + (rp[1] / rd) * rv[1] + (rp[2] / rd) * rv[2]);
                order[k] = k;
            }
            for (int a = 1; a < cnt; ++a) { /* stable insertion sort == argsort for distinct keys */
                int o = order[a]; int b = a - 1;
                while (b >= 0 && metric[order[b]] > metric[o]) { order[b + 1] = order[b]; --b; }
                order[b + 1] = o;
            }
            for (int k = 0; k < K; ++k) sel[k] = idx[order[k]];
        }
        double *o = obs + (size_t)i * e->obs_dim + e->self_dim;
        for (int k = 0; k < K; ++k) {
            int j = sel[k];
            for (int a = 0; a < 3; ++a) {
                o[k * 6 + a] = clipd(e->mpos[j][a] - e->mpos[i][a], -c->nbr_clip_pos[a], c->nbr_clip_pos[a]);
                o[k * 6 + 3 + a] = clipd(e->mvel[j][a] - e->mvel[i][a], -c->nbr_clip_vel[a], c->nbr_clip_vel[a]);
            }
        }
    }
}

/* get_surround_sdfs obstacles/utils.py:5-27 */
void qso_surround_sdf(const double qxy[2], const double *obst_xy, int32_t m, double radius, double res, double out[9]) {
    double gx[3] = {qxy[0] - res, qxy[0], qxy[0] + res}, gy[3] = {qxy[1] - res, qxy[1], qxy[1] + res};
    for (int a = 0; a < 3; ++a)
        for (int b = 0; b < 3; ++b) {
            double mind = 100.0;
            for (int o = 0; o < m; ++o) {
                double dx = gx[a] - obst_xy[2 * o], dy = gy[b] - obst_xy[2 * o + 1], dist = sqrt(dx * dx + dy * dy);
                if (dist < mind) mind = dist;
            }
            out[a * 3 + b] = mind - radius;
        }
}

/* collision_detection obstacles/utils.py:31-43: lowest obstacle index within quad_radius+obst_radius */
int qso_obst_first_hit(const double qxy[2], const double *obst_xy, int32_t m, double thr) {
    for (int o = 0; o < m; ++o) {
        double dx = qxy[0] - obst_xy[2 * o], dy = qxy[1] - obst_xy[2 * o + 1];
        if (sqrt(dx * dx + dy * dy) <= thr) return o;
    }
    return -1;
}

static void obstacle_obs(qso_env *e, double *obs) { /* MultiObstacles.reset/step obstacles/obstacles.py:15-35 */
    const qs_config *c = &e->c;
    if (!c->use_obstacles) return;
    for (int i = 0; i < c->num_agents; ++i)
        qso_surround_sdf(e->mpos[i], &e->obst_xy[0][0], e->cur_M, e->cur_size / 2.0, 0.1,
                         obs + (size_t)i * e->obs_dim + e->self_dim + 6 * c->num_neighbors);
}

/* get_cell_centers obstacles/utils.py:47-58 (grid_size 1) */
void qso_cell_centers(int32_t L, int32_t W, double *out) {
    int count = 0;
    for (int i = 0; i < L; ++i)
        for (int j = W - 1; j > -1; --j) {
            out[2 * count] = i + 0.5 - (double)(L / 2);
            out[2 * count + 1] = j + 0.5 - (double)(W / 2);
            ++count;
        }
}

/* compute_new_vel collisions/utils.py:8-18 */
static void compute_new_vel(double max_vel_magn, double vel[3], const double shift[3], double decay) {
    double vn[3] = {vel[0] + shift[0], vel[1] + shift[1], vel[2] + shift[2]};
    double mag = norm3(vn), den = (mag == 0.0) ? mag + EPS_COL : mag;
    double dir[3] = {vn[0] / den, vn[1] / den, vn[2] / den};
    mag = fmin(mag * decay, max_vel_magn);
    for (int k = 0; k < 3; ++k) { double nv = dir[k] * mag; double sh = nv - vel[k]; vel[k] += sh; }
}

/* compute_new_omega collisions/utils.py:22-34: u[0..2] direction draws, u[3] magnitude draw */
static void compute_new_omega(const double u[4], double out[3]) {
    double mag = norm3(u), den = (mag == 0.0) ? mag + EPS_COL : mag;
    for (int k = 0; k < 3; ++k) out[k] = (u[k] / den) * u[3];
}

static void add_omega(drone_t *d, const double dw[3], double sign) {
    for (int k = 0; k < 3; ++k) {
        d->omega[k] += sign * dw[k];
        if (d->flags & F_OMEGA_F32) d->omega[k] = (double)(float)d->omega[k]; /* in-place add on a float32 array */
    }
}

/* perform_collision_between_drones collisions/quadrotors.py:24-59 (+ :9-20) */
static void collide_drones(qso_env *e, int i, int j) {
    drone_t *a = &e->d[i], *b = &e->d[j];
    double n[3] = {a->pos[0] - b->pos[0], a->pos[1] - b->pos[1], a->pos[2] - b->pos[2]};
    double mag = norm3(n), den = (mag == 0.0) ? mag + EPS_COL : mag;
    for (int k = 0; k < 3; ++k) n[k] /= den;
    double v1new = dot3(a->vel, n), v2new = dot3(b->vel, n);
    double vc[3] = {(v2new - v1new) * n[0], (v2new - v1new) * n[1], (v2new - v1new) * n[2]};
    double s1[3] = {vc[0], vc[1], vc[2]}, s2[3] = {-vc[0], -vc[1], -vc[2]};
    for (int t = 0; t < 3; ++t) {
        double cons[3], n1[3], n2[3];
        rng_normal(e, QS_SITE_DD_N, t * 3 + 0, i, j, 3, 0.0, 0.8, cons);
        rng_normal(e, QS_SITE_DD_N, t * 3 + 1, i, j, 3, 0.0, 0.15, n1);
        rng_normal(e, QS_SITE_DD_N, t * 3 + 2, i, j, 3, 0.0, 0.15, n2);
        double t1[3], t2[3];
        for (int k = 0; k < 3; ++k) {
            double vn1 = cons[k] + n1[k], vn2 = -cons[k] + n2[k];
            s1[k] = vc[k] + vn1; s2[k] = -vc[k] + vn2;
            t1[k] = a->vel[k] + s1[k]; t2[k] = b->vel[k] + s2[k];
        }
        double d1 = dot3(t1, n), d2 = dot3(t2, n);
        if (d1 > 0 && 0 > d2) break;
    }
    double maxv = fmax(norm3(a->vel), norm3(b->vel));
    double dec[2];
    rng_uniform(e, QS_SITE_DD_U, 0, i, j, 2, 0.2, 0.8, dec);
    compute_new_vel(maxv, a->vel, s1, dec[0]);
    compute_new_vel(maxv, b->vel, s2, dec[1]);
    double u[4], dw[3];
    if (e->tape) { rng_uniform(e, 0, 0, 0, 0, 3, 0, 0, u); u[3] = tape_pop(e); }
    else { uint32_t w[4]; philox_group(e, QS_SITE_DD_W, 0, i, j, w);
           for (int k = 0; k < 3; ++k) u[k] = -1.0 + 2.0 * u01(w[k]);
           u[3] = 10.0 * PI + (20.0 * PI - 10.0 * PI) * u01(w[3]); }
    compute_new_omega(u, dw);
    add_omega(a, dw, 1.0);
    add_omega(b, dw, -1.0);
}

/* exposed for the reference's KAT collisions/test/unit_test/obstacles.py:6-18 */
void qso_collision_obstacle_kat(const double pos[3], const double vel[3], const double opos[3], double *vnew, double n[3]) {
    n[0] = pos[0] - opos[0]; n[1] = pos[1] - opos[1]; n[2] = 0.0;
    double mag = norm3(n), den = (mag == 0.0) ? mag + EPS_COL : mag;
    for (int k = 0; k < 3; ++k) n[k] /= den;
    *vnew = dot3(vel, n);
}

/* perform_collision_with_obstacle collisions/obstacles.py:23-50 (+ :9-20) */
static void collide_obstacle(qso_env *e, int i, int o) {
    const qs_config *c = &e->c;
    drone_t *d = &e->d[i];
    double opos[3] = {e->obst_xy[o][0], e->obst_xy[o][1], (c->room_hi[2] - c->room_lo[2]) / 2.0};
    double n[3], vnew;
    qso_collision_obstacle_kat(d->pos, d->vel, opos, &vnew, n);
    double vmag = norm3(d->vel), nv[3] = {vmag * n[0], vmag * n[1], vmag * n[2]}, noise[3] = {0, 0, 0};
    for (int t = 0; t < 3; ++t) {
        double cons[3], n1[3], tmp[3], chk[3];
        rng_normal(e, QS_SITE_OBST_N, t * 2 + 0, i, 0, 3, 0.0, 0.1, cons);
        rng_normal(e, QS_SITE_OBST_N, t * 2 + 1, i, 0, 3, 0.0, 0.05, n1);
        for (int k = 0; k < 3; ++k) { tmp[k] = cons[k] + n1[k]; chk[k] = nv[k] + tmp[k]; }
        if (dot3(chk, n) > 0) { memcpy(noise, tmp, sizeof noise); break; }
    }
    double diff[3] = {d->pos[0] - opos[0], d->pos[1] - opos[1], d->pos[2] - opos[2]};
    int inside = norm3(diff) < e->cur_size / 2;
    double decay = inside ? rng_uniform1(e, QS_SITE_OBST_U, 0, i, 0, 1.0, 1.0) : rng_uniform1(e, QS_SITE_OBST_U, 0, i, 0, 0.2, 0.8);
    double shift[3] = {nv[0] - d->vel[0] + noise[0], nv[1] - d->vel[1] + noise[1], nv[2] - d->vel[2] + noise[2]};
    compute_new_vel(vmag, d->vel, shift, decay);
    double u[4], dw[3];
    if (e->tape) { rng_uniform(e, 0, 0, 0, 0, 3, 0, 0, u); u[3] = tape_pop(e); }
    else { uint32_t w[4]; philox_group(e, QS_SITE_OBST_W, 0, i, 0, w);
           for (int k = 0; k < 3; ++k) u[k] = -1.0 + 2.0 * u01(w[k]);
           u[3] = 0.5 * PI + (PI - 0.5 * PI) * u01(w[3]); }
    compute_new_omega(u, dw);
    add_omega(d, dw, 1.0);
}

/* perform_collision_with_wall collisions/room.py:6-44 / perform_collision_with_ceiling :91-113 */
static void collide_room(qso_env *e, int i, int is_wall) {
    const qs_config *c = &e->c;
    drone_t *d = &e->d[i];
    int site = is_wall ? QS_SITE_WALL : QS_SITE_CEIL;
    double speed = norm3(d->vel), dir[3], g0[4];
    if (e->tape) {
        g0[0] = tape_pop(e); for (int k = 0; k < 3; ++k) dir[k] = tape_pop(e);
    } else {
        uint32_t w[4]; philox_group(e, site, 0, i, 0, w);
        g0[0] = 0.2 * speed + (0.8 * speed - 0.2 * speed) * u01(w[0]);
        for (int k = 0; k < 3; ++k) dir[k] = -1.0 + 2.0 * u01(w[1 + k]);
    }
    double real_speed = clipd(g0[0], 0.1, 6.0);
    uint32_t w1[4] = {0, 0, 0, 0};
    if (!e->tape) philox_group(e, site, 1, i, 0, w1);
    if (is_wall) {
        int x0 = d->pos[0] == c->room_lo[0], x1 = d->pos[0] == c->room_hi[0];
        int y0 = d->pos[1] == c->room_lo[1], y1 = d->pos[1] == c->room_hi[1];
        if (x0) dir[0] = e->tape ? tape_pop(e) : 0.1 + 0.9 * u01(w1[0]);
        else if (x1) dir[0] = e->tape ? tape_pop(e) : -1.0 + 0.9 * u01(w1[0]);
        if (y0) dir[1] = e->tape ? tape_pop(e) : 0.1 + 0.9 * u01(w1[1]);
        else if (y1) dir[1] = e->tape ? tape_pop(e) : -1.0 + 0.9 * u01(w1[1]);
    }
    dir[2] = e->tape ? tape_pop(e) : -1.0 + 0.5 * u01(w1[2]);
    double dm = norm3(dir);
    for (int k = 0; k < 3; ++k) d->vel[k] = real_speed * (dir[k] / (dm + 1e-5));
    double u[4];
    if (e->tape) { for (int k = 0; k < 4; ++k) u[k] = tape_pop(e); }
    else { uint32_t w[4]; philox_group(e, site, 2, i, 0, w);
           for (int k = 0; k < 3; ++k) u[k] = -1.0 + 2.0 * u01(w[k]);
           u[3] = 10.0 * PI + 10.0 * PI * u01(w[3]); }
    double um = norm3(u), dw[3];
    for (int k = 0; k < 3; ++k) { dw[k] = u[k] / (um + 1e-5); dw[k] *= u[3]; }
    add_omega(d, dw, 1.0);
}

/* perform_downwash aerodynamics/downwash.py:4-51 (+ get_vel_omega_norm :54-66); returns "any applied" */
static int downwash(qso_env *e) {
    const qs_config *c = &e->c;
    int N = c->num_agents, any = 0;
    const double dt = c->dt * c->sim_steps; /* control_dt = 1/control_freq = 0.01 */
    double pos[MAXN][3], zax[MAXN][3];
    for (int i = 0; i < N; ++i) {
        memcpy(pos[i], e->d[i].pos, sizeof pos[i]);
        zax[i][0] = e->d[i].rot[2]; zax[i][1] = e->d[i].rot[5]; zax[i][2] = e->d[i].rot[8];
    }
    for (int i = 0; i < N; ++i) {
        double ui[2];
        if (e->tape) { ui[0] = tape_pop(e); ui[1] = tape_pop(e); }
        else { uint32_t w[4]; philox_group(e, QS_SITE_DW_I, 0, i, 0, w); ui[0] = -0.1 + 0.2 * u01(w[0]); ui[1] = -0.01 + 0.02 * u01(w[1]); }
        for (int j = 0; j < N; ++j) {
            if (i == j) continue;
            double rel[3] = {pos[j][0] - pos[i][0], pos[j][1] - pos[i][1], pos[j][2] - pos[i][2]};
            double dist = norm3(rel);
            double acc = fmax(1e-6, (6.0 / 17.0) * (-10 * dist + 7) + ui[0]);
            double omw = fmax(1e-6, 0.3 * ((dist - 1) * (dist - 1)) + ui[1]);
            double rz = dot3(rel, zax[i]);
            double rxy = sqrt(dist * dist - rz * rz);
            if (-0.7 < rz && rz < 0 && rxy < 0.1) {
                double nz[3], dirw[3];
                if (e->tape) { for (int k = 0; k < 3; ++k) nz[k] = tape_pop(e); for (int k = 0; k < 3; ++k) dirw[k] = tape_pop(e); }
                else {
                    uint32_t w[4]; philox_group(e, QS_SITE_DW_IJ_V, 0, i, j, w);
                    for (int k = 0; k < 3; ++k) nz[k] = -0.1 + 0.2 * u01(w[k]);
                    philox_group(e, QS_SITE_DW_IJ_W, 0, i, j, w);
                    for (int k = 0; k < 3; ++k) dirw[k] = -1.0 + 2.0 * u01(w[k]);
                }
                for (int k = 0; k < 3; ++k) nz[k] = zax[i][k] + nz[k];
                double m = norm3(nz), den = (m == 0.0) ? m + 1e-6 : m;
                double mw = norm3(dirw), denw = (mw == 0.0) ? mw + 1e-6 : mw;
                double dwv[3];
                for (int k = 0; k < 3; ++k) {
                    double down = -1.0 * (nz[k] / den);
                    e->d[j].vel[k] += acc * down * dt;
                    dwv[k] = omw * (dirw[k] / denw) * dt;
                }
                add_omega(&e->d[j], dwv, 1.0);
                any = 1;
            }
        }
    }
    return any;
}

/* ------------------------------------------------------------------------------------------ */
/* scenarios                                                                                   */
/* ------------------------------------------------------------------------------------------ */
static const char *FORMATIONS[8] = {"circle_horizontal", "circle_vertical_xz", "circle_vertical_yz", "sphere",
                                    "grid_horizontal", "grid_vertical_xz", "grid_vertical_yz", "cube"};
static int f_is_circle(int f) { return f <= 2; }
static int f_is_grid(int f) { return f >= 4 && f <= 6; }
static int f_suffix(int f) { /* 0 horizontal, 1 vertical_xz, 2 vertical_yz, -1 none */
    if (f == 0 || f == 4) return 0; if (f == 1 || f == 5) return 1; if (f == 2 || f == 6) return 2; return -1;
}
static void grid_dim(int num, int *d1, int *d2) { /* get_grid_dim_number scenarios/utils.py:117-128 */
    int g = (int)floor(sqrt((double)num)), a = g;
    while (a > 1) { if (num % a == 0) break; --a; }
    *d1 = a; *d2 = num / a;
}
static void goal_by_formation(int f, double p0, double p1, double layer, double g[3]) { /* utils.py:156-167 */
    int s = f_suffix(f);
    if (s == 0) { g[0] = p0; g[1] = p1; g[2] = layer; }
    else if (s == 1) { g[0] = p0; g[1] = layer; g[2] = p1; }
    else { g[0] = layer; g[1] = p0; g[2] = p1; }
}

/* QuadrotorScenario.generate_goals scenarios/base.py:39-113; returns number of rows produced */
static int generate_goals(qso_env *e, int n, const double center[3], double layer_dist, double out[][3]) {
    int f = e->formation, per = e->per_layer, rows = n;
    double size = e->form_size;
    if (f_is_circle(f)) {
        int layers[MAXN], nl = 0;
        if (n <= per) layers[nl++] = n;
        else { for (int k = 0; k < n / per; ++k) layers[nl++] = per; if (n % per > 0) layers[nl++] = n % per; }
        for (int i = 0; i < n; ++i) {
            int cur = layers[i / per];
            double deg = 2 * PI * (i % cur) / cur;
            goal_by_formation(f, size * cos(deg), size * sin(deg), (i / per) * layer_dist, out[i]);
            for (int k = 0; k < 3; ++k) out[i][k] += center[k];
        }
    } else if (f == 3) { /* sphere: generate_points scenarios/utils.py:79-95 (n<3 is cast to 3 rows) */
        int m = n < 3 ? 3 : n;
        double x = 0.1 + 1.2 * m, start = -1. + 1. / (m - 1.), inc = (2. - 2. / (m - 1.)) / (m - 1.);
        for (int j = 0; j < m; ++j) {
            double s = start + j * inc, sg = (s > 0) - (s < 0);
            double xx = s * x, yy = PI / 2. * sg * (1. - sqrt(1. - fabs(s)));
            double p[3] = {cos(xx) * cos(yy), sin(xx) * cos(yy), sin(yy)};
            for (int k = 0; k < 3; ++k) out[j][k] = size * p[k] + center[k];
        }
        rows = m;
    } else if (f_is_grid(f)) {
        int dims[MAXN][2], nl = 0;
        if (n <= per) { grid_dim(n, &dims[0][0], &dims[0][1]); nl = 1; }
        else {
            int m1, m2; grid_dim(per, &m1, &m2);
            for (int k = 0; k < n / per; ++k) { dims[nl][0] = m1; dims[nl][1] = m2; ++nl; }
            if (n % per > 0) { grid_dim(n % per, &dims[nl][0], &dims[nl][1]); ++nl; }
        }
        double mean[3] = {0, 0, 0};
        for (int i = 0; i < n; ++i) {
            int d1 = dims[i / per][0], d2 = dims[i / per][1];
            goal_by_formation(f, size * (i % d2), size * ((int)((double)i / d2) % d1), (i / per) * layer_dist, out[i]);
            for (int k = 0; k < 3; ++k) mean[k] += out[i][k];
        }
        for (int k = 0; k < 3; ++k) mean[k] /= n;
        for (int i = 0; i < n; ++i) for (int k = 0; k < 3; ++k) out[i][k] = out[i][k] - mean[k] + center[k];
    } else { /* cube */
        int fd = (int)pow((double)n, 1.0 / 3);
        double mean[3] = {0, 0, 0};
        for (int i = 0; i < n; ++i) {
            out[i][0] = center[2] + size * (i / (fd * fd));
            out[i][1] = size * ((int)((double)i / fd) % fd);
            out[i][2] = size * (i % fd);
            for (int k = 0; k < 3; ++k) mean[k] += out[i][k];
        }
        for (int k = 0; k < 3; ++k) mean[k] /= n;
        for (int i = 0; i < n; ++i) for (int k = 0; k < 3; ++k) out[i][k] = out[i][k] - mean[k] + center[k];
    }
    return rows;
}

/* np.random.shuffle of the rows: tape holds the permutation, philox uses Fisher-Yates */
static void shuffle_rows(qso_env *e, double rows[][3], int n, int slot_base) {
    int perm[MAXG];
    if (e->tape) { for (int k = 0; k < n; ++k) perm[k] = (int)tape_pop(e); }
    else {
        for (int k = 0; k < n; ++k) perm[k] = k;
        for (int i = n - 1; i >= 1; --i) {
            int j = (int)(rng_uniform1(e, QS_SITE_SCEN_SHUFFLE, slot_base + i, 0, 0, 0.0, 1.0) * (i + 1));
            if (j > i) j = i;
            int t = perm[i]; perm[i] = perm[j]; perm[j] = t;
        }
    }
    double tmp[MAXG][3];
    for (int k = 0; k < n; ++k) memcpy(tmp[k], rows[perm[k]], sizeof tmp[k]);
    for (int k = 0; k < n; ++k) memcpy(rows[k], tmp[k], sizeof tmp[k]);
}

/* per-scenario formation tables: QUADS_PARAMS_DICT scenarios/utils.py:33-51 */
static void scen_params(int scen, int *nform, double *lo, double *hi) {
    *nform = 1; *lo = 0.0; *hi = 0.0;
    switch (scen) {
    case QS_SCENARIO_STATIC_DIFF_GOAL: case QS_SCENARIO_DYNAMIC_DIFF_GOAL: case QS_SCENARIO_SWARM_VS_SWARM: case QS_SCENARIO_RUN_AWAY:
        *nform = 8; *lo = 5 * 0.05; *hi = 10 * 0.05; break;
    case QS_SCENARIO_SWAP_GOALS: *nform = 8; *lo = 8 * 0.05; *hi = 16 * 0.05; break;
    case QS_SCENARIO_DYNAMIC_FORMATIONS: *nform = 8; *lo = 0.0; *hi = 20 * 0.05; break;
    case QS_SCENARIO_O_SWAP_GOALS: *nform = 7; *lo = 8 * 0.05; *hi = 16 * 0.05; break;   /* QUADS_FORMATION_LIST_OBSTACLES has 7 entries */
    default: break;
    }
}

/* update_formation_and_relate_param scenarios/base.py:123-135 (+ utils.py:55-70, :131-153) */
static void update_formation(qso_env *e, int slot) {
    const qs_config *c = &e->c;
    int nform; double lo, hi;
    scen_params(e->scen, &nform, &lo, &hi);
    int fi;
    if (e->tape) fi = (int)tape_pop(e);
    else { fi = (int)(rng_uniform1(e, QS_SITE_SCEN, slot + 0, 0, 0, 0.0, 1.0) * nform); if (fi >= nform) fi = nform - 1; }
    e->formation = fi;
    e->per_layer = f_is_circle(fi) ? 8 : (f_is_grid(fi) ? 50 : 8);
    int n = (e->scen == QS_SCENARIO_SWARM_VS_SWARM) ? c->num_agents / 2 : c->num_agents;
    if (f_is_circle(fi)) { /* get_circle_radius utils.py:110-113 */
        double theta = 2 * PI / e->per_layer;
        e->form_lo = (0.5 * lo) / sin(theta / 2); e->form_hi = (0.5 * hi) / sin(theta / 2);
    } else if (fi == 3) { /* get_sphere_radius utils.py:99-106 */
        double A = 1.75388487222762, B = 0.860487305801679, C = 10.3632729642351, D = 0.0920858134405214;
        double ratio = (A - D) / (1 + pow(n / C, B)) + D;
        e->form_lo = lo / ratio; e->form_hi = hi / ratio;
    } else { e->form_lo = lo; e->form_hi = hi; }
    e->form_size = rng_uniform1(e, QS_SITE_SCEN, slot + 1, 0, 0, e->form_lo, e->form_hi);
    e->layer_dist = rng_uniform1(e, QS_SITE_SCEN, slot + 2, 0, 0, e->form_lo, e->form_hi);
    (void)FORMATIONS;
}

/* Scenario_swarm_vs_swarm.create_formations scenarios/swarm_vs_swarm.py:52-57 */
static void svs_create_formations(qso_env *e, int do_shuffle) {
    int N = e->c.num_agents, n1 = N / 2, n2 = N - N / 2;
    double g1[MAXG][3], g2[MAXG][3];
    int r1 = generate_goals(e, n1, e->center1, e->layer_dist, g1);
    int r2 = generate_goals(e, n2, e->center2, e->layer_dist, g2);
    if (do_shuffle) { shuffle_rows(e, g1, r1, 0); shuffle_rows(e, g2, r2, 256); }
    for (int k = 0; k < r1; ++k) memcpy(e->goals[k], g1[k], sizeof g1[k]);
    for (int k = 0; k < r2; ++k) memcpy(e->goals[r1 + k], g2[k], sizeof g2[k]);
    e->num_goals = r1 + r2;
}

/* QuadrotorScenario.standard_reset scenarios/base.py:153-167 (== .reset :140-151 with the default centre) */
static void standard_reset(qso_env *e, const double center[3]) {
    update_formation(e, 0);
    memcpy(e->center1, center, sizeof e->center1);
    e->num_goals = generate_goals(e, e->c.num_agents, e->center1, e->layer_dist, e->goals);
    shuffle_rows(e, e->goals, e->num_goals, 0);
}

/* free cells in np.where(obst_map == 0) order (row-major) */
static int free_space(const qso_env *e, int fs[][2]) {
    int L = e->c.obst_area[0], W = e->c.obst_area[1], n = 0;
    for (int r = 0; r < L; ++r) for (int q = 0; q < W; ++q) if (!e->obst_map[r][q]) { fs[n][0] = r; fs[n][1] = q; ++n; }
    return n;
}
static void cell_pos(const qso_env *e, const int cell[2], double out[2]) { /* o_base.py:69-73: index = x + width*y */
    int index = cell[0] + e->c.obst_area[0] * cell[1];
    out[0] = e->cell_centers[2 * index]; out[1] = e->cell_centers[2 * index + 1];
}
/* Scenario_o_base.generate_pos_obst_map_2 o_base.py:69-81: N distinct free cells + z ~ U(1,3) */
static void pos_obst_map_2(qso_env *e, double out[][3], int slot_choice, int slot_z) {
    int N = e->c.num_agents;
    int (*fs)[2] = (int (*)[2])malloc(sizeof(int[2]) * 64 * 64);
    int nfree = free_space(e, fs), ids[MAXN];
    if (e->tape) { for (int k = 0; k < N; ++k) ids[k] = (int)tape_pop(e); }
    else {
        int *pool = (int *)malloc(sizeof(int) * 64 * 64);
        for (int k = 0; k < nfree; ++k) pool[k] = k;
        for (int k = 0; k < N; ++k) {
            int j = k + (int)(rng_uniform1(e, QS_SITE_SCEN, slot_choice + k, 0, 0, 0.0, 1.0) * (nfree - k));
            if (j >= nfree) j = nfree - 1;
            int t = pool[k]; pool[k] = pool[j]; pool[j] = t; ids[k] = pool[k];
        }
        free(pool);
    }
    for (int k = 0; k < N; ++k) {
        cell_pos(e, fs[ids[k]], out[k]);
        out[k][2] = rng_uniform1(e, QS_SITE_SCEN, slot_z + k, 0, 0, 1.0, 3.0);
    }
    free(fs);
}
/* Scenario_o_base.generate_pos_obst_map o_base.py:48-67 (check_surroundings=False): one free cell + z ~ U(0.75,3) */
static void pos_obst_map_1(qso_env *e, double out[3], int slot) {
    int (*fs)[2] = (int (*)[2])malloc(sizeof(int[2]) * 64 * 64);
    int nfree = free_space(e, fs), idx;
    if (e->tape) idx = (int)tape_pop(e);
    else { idx = (int)(rng_uniform1(e, QS_SITE_SCEN, slot, 0, 0, 0.0, 1.0) * nfree); if (idx >= nfree) idx = nfree - 1; }
    cell_pos(e, fs[idx], out);
    out[2] = rng_uniform1(e, QS_SITE_SCEN, slot + 1, 0, 0, 0.75, 3.0);
    free(fs);
}
/* Scenario_o_base.max_square_area_center o_base.py:124-153 */
static void max_square_center(qso_env *e, double out[3], int slot_z) {
    int L = e->c.obst_area[0], W = e->c.obst_area[1];
    static int dp[64][64];
    memset(dp, 0, sizeof dp);
    for (int q = 0; q < W; ++q) dp[0][q] = e->obst_map[0][q];
    for (int r = 0; r < L; ++r) dp[r][0] = e->obst_map[r][0];
    int max_size = 0, cx = 0, cy = 0;
    for (int r = 1; r < L; ++r)
        for (int q = 1; q < W; ++q)
            if (e->obst_map[r][q] == 0) {
                int m = dp[r - 1][q] < dp[r][q - 1] ? dp[r - 1][q] : dp[r][q - 1];
                if (dp[r - 1][q - 1] < m) m = dp[r - 1][q - 1];
                dp[r][q] = m + 1;
                if (dp[r][q] > max_size) { max_size = dp[r][q]; cx = r - (max_size - 1) / 2; cy = q - (max_size - 1) / 2; }
            }
    int index = cx + W * cy;
    out[0] = e->cell_centers[2 * index]; out[1] = e->cell_centers[2 * index + 1];
    out[2] = rng_uniform1(e, QS_SITE_SCEN, slot_z, 0, 0, 1.5, 3.0);
}

static int mix_pick(qso_env *e) { /* Scenario_mix.reset scenarios/mix.py:84-90 + utils.py:10-25 */
    static const int LIST_MULTI[9] = {QS_SCENARIO_STATIC_SAME_GOAL, QS_SCENARIO_STATIC_DIFF_GOAL, QS_SCENARIO_EP_LISSAJOUS3D,
                                      QS_SCENARIO_EP_RAND_BEZIER, QS_SCENARIO_DYNAMIC_SAME_GOAL, QS_SCENARIO_DYNAMIC_DIFF_GOAL,
                                      QS_SCENARIO_DYNAMIC_FORMATIONS, QS_SCENARIO_SWAP_GOALS, QS_SCENARIO_SWARM_VS_SWARM};
    static const int LIST_SINGLE[5] = {QS_SCENARIO_STATIC_SAME_GOAL, QS_SCENARIO_STATIC_DIFF_GOAL, QS_SCENARIO_EP_LISSAJOUS3D,
                                       QS_SCENARIO_EP_RAND_BEZIER, QS_SCENARIO_DYNAMIC_SAME_GOAL};
    static const int LIST_OBST[2] = {QS_SCENARIO_O_RANDOM, QS_SCENARIO_O_STATIC_SAME_GOAL};
    const qs_config *c = &e->c;
    const int *list; int n;
    if (c->num_agents == 1) { if (c->use_obstacles) { list = LIST_OBST; n = 1; } else { list = LIST_SINGLE; n = 5; } }
    else if (!c->use_obstacles) { list = LIST_MULTI; n = 9; }
    else { list = LIST_OBST; n = 2; }
    int k;
    if (e->tape) k = (int)tape_pop(e);
    else { k = (int)(rng_uniform1(e, QS_SITE_SCEN, 288, 0, 0, 0.0, 1.0) * n); if (k >= n) k = n - 1; }
    return list[k];
}

/* scenario.reset() of every supported scenario (file:line at each case) */
static void scenario_reset(qso_env *e) {
    const qs_config *c = &e->c;
    int N = c->num_agents;
    double control_freq = 1.0 / (c->dt * c->sim_steps);
    e->have_spawn_points = 0;
    e->bez_valid = 0;
    int constructed = 0;
    if (c->scenario == QS_SCENARIO_MIX) { e->scen = mix_pick(e); constructed = 1; }   /* a new scenario object per episode */
    else { e->scen = c->scenario; constructed = !e->scen_constructed; e->scen_constructed = 1; }
    /* constructors: control_step_for_sec defaults (5 s; o_swap_goals 6 s) and dynamic_formations' speed draw (:13) */
    if (constructed) {
        e->control_step_for_sec = (int)((e->scen == QS_SCENARIO_O_SWAP_GOALS ? 6.0 : 5.0) * control_freq);
        if (e->scen == QS_SCENARIO_DYNAMIC_FORMATIONS) e->speed = rng_uniform1(e, QS_SITE_SCEN, 289, 0, 0, 1.0, 3.0);
    }
    const double c002[3] = {0.0, 0.0, 2.0};
    switch (e->scen) {
    case QS_SCENARIO_STATIC_SAME_GOAL:       /* scenarios/base.py:140-151 */
    case QS_SCENARIO_STATIC_DIFF_GOAL:       /* scenarios/static_diff_goal.py (base reset) */
    case QS_SCENARIO_EP_RAND_BEZIER:         /* scenarios/ep_rand_bezier.py (base reset) */
        standard_reset(e, c002);
        break;
    case QS_SCENARIO_RUN_AWAY:               /* scenarios/run_away.py:29-40 (= the base reset); step() acts once per second */
        e->control_step_for_sec = (int)(1.0 * control_freq);
        standard_reset(e, c002);
        break;
    case QS_SCENARIO_DYNAMIC_SAME_GOAL:      /* scenarios/dynamic_same_goal.py:31-37 */
    case QS_SCENARIO_DYNAMIC_DIFF_GOAL:      /* scenarios/dynamic_diff_goal.py:36-42 */
    case QS_SCENARIO_SWAP_GOALS: {           /* scenarios/swap_goals.py:26-32 */
        double dur = rng_uniform1(e, QS_SITE_SCEN, 8, 0, 0, 4.0, 6.0);
        e->control_step_for_sec = (int)(dur * control_freq);
        standard_reset(e, c002);
        break;
    }
    case QS_SCENARIO_DYNAMIC_FORMATIONS: {   /* scenarios/dynamic_formations.py:37-42 */
        e->increase = rng_uniform1(e, QS_SITE_SCEN, 290, 0, 0, 0.0, 1.0) < 0.5;
        e->speed = rng_uniform1(e, QS_SITE_SCEN, 291, 0, 0, 1.0, 3.0);
        standard_reset(e, c002);
        break;
    }
    case QS_SCENARIO_EP_LISSAJOUS3D: {       /* scenarios/ep_lissajous3D.py:31-38 */
        update_formation(e, 0);
        const double ctr[3] = {-2.0, 0.0, 2.0};
        memcpy(e->center1, ctr, sizeof ctr);
        e->num_goals = generate_goals(e, N, e->center1, 0.0, e->goals);
        break;
    }
    case QS_SCENARIO_O_STATIC_SAME_GOAL:     /* scenarios/obstacles/o_static_same_goal.py:27-48 */
    case QS_SCENARIO_O_DYNAMIC_SAME_GOAL: {  /* scenarios/obstacles/o_dynamic_same_goal.py:30-51 */
        double dur = rng_uniform1(e, QS_SITE_SCEN, 8, 0, 0, 4.0, 6.0);
        e->control_step_for_sec = (int)(dur * control_freq);
        pos_obst_map_2(e, e->spawn_points, 16, 96);
        e->have_spawn_points = 1;
        max_square_center(e, e->end_point, 9);
        update_formation(e, 0);
        for (int k = 0; k < N; ++k) memcpy(e->goals[k], e->end_point, sizeof e->end_point);
        e->num_goals = N;
        break;
    }
    case QS_SCENARIO_O_EP_RAND_BEZIER: {     /* scenarios/obstacles/o_ep_rand_bezier.py:57-97 */
        pos_obst_map_2(e, e->spawn_points, 16, 96);
        e->have_spawn_points = 1;
        pos_obst_map_1(e, e->end_point, 9);
        if (e->tape) {
            /* :72-90 ten "trajectory points" from the free cells, never used afterwards (step() has the line commented out): the
             * loop only consumes draws.  Replayed against the tape; with the counter-based stream there is nothing to skip. */
            int (*fs)[2] = (int (*)[2])malloc(sizeof(int[2]) * 64 * 64);
            int nfree = free_space(e, fs), sampled[10], ns = 0;
            free(fs);
            while (ns < 10) {
                int idx = (int)tape_pop(e), reject = 0;   /* np.random.choice(len(self.free_space)) */
                for (int q = 0; q < ns; ++q) {            /* cell_centers[] is indexed with free-space indices in the reference */
                    double dx = e->cell_centers[2 * sampled[q]] - e->cell_centers[2 * idx], dy = e->cell_centers[2 * sampled[q] + 1] - e->cell_centers[2 * idx + 1];
                    if (sqrt(dx * dx + dy * dy) > 4.0) reject = 1;
                }
                if (reject) continue;
                sampled[ns++] = idx;
                --nfree;                                   /* self.free_space.pop(point_idx) */
            }
            (void)nfree;
        }
        update_formation(e, 0);
        for (int k = 0; k < N; ++k) memcpy(e->goals[k], e->end_point, sizeof e->end_point);
        e->num_goals = N;
        break;
    }
    case QS_SCENARIO_O_RANDOM: {             /* scenarios/obstacles/o_random.py:27-52 */
        if (e->tape) tape_skip(e, 4 * N);    /* N x (generate_pos_obst_map() twice): results are overwritten below */
        pos_obst_map_2(e, e->spawn_points, 16, 96);
        e->have_spawn_points = 1;
        pos_obst_map_2(e, e->goals, 160, 224);
        e->num_goals = N;
        e->duration_step = (int)(rng_uniform1(e, QS_SITE_SCEN, 8, 0, 0, 2.0, 4.0) * control_freq);
        update_formation(e, 0);
        break;
    }
    case QS_SCENARIO_O_SWAP_GOALS: {         /* scenarios/obstacles/o_swap_goals.py:27-52 */
        double dur = rng_uniform1(e, QS_SITE_SCEN, 8, 0, 0, 4.0, 6.0);
        e->control_step_for_sec = (int)(dur * control_freq);
        update_formation(e, 0);
        pos_obst_map_2(e, e->spawn_points, 16, 96);
        e->have_spawn_points = 1;
        max_square_center(e, e->center1, 9);
        e->num_goals = generate_goals(e, N, e->center1, e->layer_dist, e->goals);
        shuffle_rows(e, e->goals, e->num_goals, 0);
        break;
    }
    default: {                               /* swarm_vs_swarm: scenarios/swarm_vs_swarm.py:80-94, :17-50 */
        double dur = rng_uniform1(e, QS_SITE_SCEN, 8, 0, 0, 4.0, 6.0);
        e->control_step_for_sec = (int)(dur * control_freq);
        update_formation(e, 0);
        double box = c->spawn_box, low = e->form_lo, xy[2];
        rng_uniform(e, QS_SITE_SCEN, 9, 0, 0, 2, -box, box, xy);
        /* get_z_value scenarios/utils.py:170-181 */
        double z = rng_uniform1(e, QS_SITE_SCEN, 10, 0, 0, -0.5 * box, 0.5 * box) + 2.0, zlb = 0.25;
        int f = e->formation;
        if (f == 3 || f == 1 || f == 2) zlb = e->form_size + 0.25;
        else if (f == 5 || f == 6) { int rn = c->num_agents < e->per_layer ? c->num_agents : e->per_layer, d1, d2; grid_dim(rn, &d1, &d2); zlb = d1 * e->form_size + 0.25; }
        z = fmax(zlb, z);
        e->center1[0] = xy[0]; e->center1[1] = xy[1]; e->center1[2] = z;
        double dist = rng_uniform1(e, QS_SITE_SCEN, 11, 0, 0, box / 4, box);
        double phi = rng_uniform1(e, QS_SITE_SCEN, 12, 0, 0, -PI, PI);
        double theta = rng_uniform1(e, QS_SITE_SCEN, 13, 0, 0, -0.5 * PI, 0.5 * PI);
        double dv[3] = {sin(theta) * cos(phi), sin(theta) * sin(phi), cos(theta)};
        for (int k = 0; k < 3; ++k) e->center2[k] = e->center1[k] + dist * dv[k];
        int s = f_suffix(f), ax = (s == 0) ? 2 : ((s == 1) ? 1 : ((s == 2) ? 0 : -1));
        if (ax >= 0) {
            double df = e->center2[ax] - e->center1[ax];
            if (fabs(df) < low) { double sg = (df > 0) - (df < 0); e->center2[ax] = sg * low + e->center1[ax]; }
        }
        svs_create_formations(e, 0);
        break;
    }
    }
    /* approch_goal_metric: 0.5 (base.py:31, o_random.py:10), 1.0 for the other obstacle scenarios (o_base.py:16) */
    e->approach_metric = (e->scen == QS_SCENARIO_O_STATIC_SAME_GOAL || e->scen == QS_SCENARIO_O_DYNAMIC_SAME_GOAL ||
                          e->scen == QS_SCENARIO_O_SWAP_GOALS || e->scen == QS_SCENARIO_O_EP_RAND_BEZIER) ? 1.0 : 0.5;
}

static void set_all_goals(qso_env *e) { for (int i = 0; i < e->c.num_agents; ++i) memcpy(e->d[i].goal, e->goals[i], sizeof e->d[i].goal); }

/* get_z_value scenarios/utils.py:170-181 */
static double get_z_value(qso_env *e, int slot) {
    double box = e->c.spawn_box;
    double z = rng_uniform1(e, QS_SITE_SCEN, slot, 0, 0, -0.5 * box, 0.5 * box) + 2.0, zlb = 0.25;
    int f = e->formation;
    if (f == 3 || f == 1 || f == 2) zlb = e->form_size + 0.25;
    else if (f == 5 || f == 6) { int rn = e->c.num_agents < e->per_layer ? e->c.num_agents : e->per_layer, d1, d2; grid_dim(rn, &d1, &d2); zlb = d1 * e->form_size + 0.25; }
    return fmax(zlb, z);
}

/* scenario.step() (called once per control step after the collision responses, quadrotor_multi.py:590) */
static void scenario_step(qso_env *e) {
    const qs_config *c = &e->c;
    int tick = e->tick, N = c->num_agents;
    double control_freq = 1.0 / (c->dt * c->sim_steps);
    int at_period = e->control_step_for_sec > 0 && tick % e->control_step_for_sec == 0 && tick > 0;
    switch (e->scen) {
    case QS_SCENARIO_SWARM_VS_SWARM:         /* scenarios/swarm_vs_swarm.py:59-79 */
        if (at_period) {
            double t[3]; memcpy(t, e->center1, sizeof t); memcpy(e->center1, e->center2, sizeof t); memcpy(e->center2, t, sizeof t);
            update_formation(e, 32);
            svs_create_formations(e, 1);
            set_all_goals(e);
        }
        break;
    case QS_SCENARIO_RUN_AWAY:               /* scenarios/run_away.py:15-27: drones 0 and 1 get the goals of two random others */
        if (at_period) {
            int gi[2];
            for (int k = 0; k < 2; ++k) {            /* np.random.randint(low=1, high=N, size=2) */
                if (e->tape) gi[k] = (int)tape_pop(e);
                else { gi[k] = 1 + (int)(rng_uniform1(e, QS_SITE_SCEN, 44 + k, 0, 0, 0.0, 1.0) * (N - 1)); if (gi[k] > N - 1) gi[k] = N - 1; }
            }
            memcpy(e->goals[0], e->goals[gi[0]], sizeof e->goals[0]);
            memcpy(e->goals[1], e->goals[gi[1]], sizeof e->goals[1]);
            memcpy(e->d[0].goal, e->goals[0], sizeof e->d[0].goal);
            memcpy(e->d[1].goal, e->goals[1], sizeof e->d[1].goal);
        }
        break;
    case QS_SCENARIO_DYNAMIC_SAME_GOAL:      /* scenarios/dynamic_same_goal.py:16-29 */
        if (at_period) {
            double box = c->spawn_box, xy[2];
            rng_uniform(e, QS_SITE_SCEN, 40, 0, 0, 2, -box, box, xy);
            double z = fmax(0.25, rng_uniform1(e, QS_SITE_SCEN, 41, 0, 0, -0.5 * box, 0.5 * box) + 2.0);
            e->center1[0] = xy[0]; e->center1[1] = xy[1]; e->center1[2] = z;
            e->num_goals = generate_goals(e, N, e->center1, 0.0, e->goals);
            set_all_goals(e);
        }
        break;
    case QS_SCENARIO_DYNAMIC_DIFF_GOAL:      /* scenarios/dynamic_diff_goal.py:8-34 */
        if (at_period) {
            double box = c->spawn_box, xy[2];
            rng_uniform(e, QS_SITE_SCEN, 40, 0, 0, 2, -box, box, xy);
            double z = get_z_value(e, 41);
            e->center1[0] = xy[0]; e->center1[1] = xy[1]; e->center1[2] = z;
            update_formation(e, 32);
            e->num_goals = generate_goals(e, N, e->center1, e->layer_dist, e->goals);
            shuffle_rows(e, e->goals, e->num_goals, 0);
            set_all_goals(e);
        }
        break;
    case QS_SCENARIO_DYNAMIC_FORMATIONS:     /* scenarios/dynamic_formations.py:16-35 */
        if (e->form_size <= -e->form_hi) { e->increase = 1; e->speed = rng_uniform1(e, QS_SITE_SCEN, 292, 0, 0, 1.0, 3.0); }
        else if (e->form_size >= e->form_hi) { e->increase = 0; e->speed = rng_uniform1(e, QS_SITE_SCEN, 292, 0, 0, 1.0, 3.0); }
        if (e->increase) e->form_size += 0.001 * e->speed; else e->form_size -= 0.001 * e->speed;
        e->num_goals = generate_goals(e, N, e->center1, e->layer_dist, e->goals);
        set_all_goals(e);
        break;
    case QS_SCENARIO_SWAP_GOALS:             /* scenarios/swap_goals.py:13-24 */
    case QS_SCENARIO_O_SWAP_GOALS:           /* scenarios/obstacles/o_swap_goals.py:14-25 */
        if (at_period) { shuffle_rows(e, e->goals, e->num_goals, 0); set_all_goals(e); }
        break;
    case QS_SCENARIO_EP_LISSAJOUS3D: {       /* scenarios/ep_lissajous3D.py:8-26 (phi = psi = 90 rad, a .03, b = c = .01, n = m = 2) */
        double t = tick / control_freq;
        double x = 0.03 * sin(t), y = 0.01 * sin(2 * t + 90), z = 0.01 * cos(2 * t + 90);
        double g0[3] = {x + e->goals[0][0], y + e->goals[0][1], z + e->goals[0][2]};
        for (int i = 0; i < N; ++i) memcpy(e->goals[i], g0, sizeof g0);
        e->num_goals = N;
        set_all_goals(e);
        break;
    }
    case QS_SCENARIO_EP_RAND_BEZIER:         /* scenarios/ep_rand_bezier.py:8-48 */
    case QS_SCENARIO_O_EP_RAND_BEZIER: {     /* scenarios/obstacles/o_ep_rand_bezier.py:16-55: 6 s legs, <= 5 m, goal height in [1.5, 3] */
        const int obst = e->scen == QS_SCENARIO_O_EP_RAND_BEZIER;
        int control_steps = (int)((obst ? 6 : 5) * control_freq), t = tick % control_steps;
        double room[3] = {c->room_hi[0] - c->room_lo[0] - e->form_size, c->room_hi[1] - c->room_lo[1] - e->form_size, c->room_hi[2] - c->room_lo[2] - e->form_size};
        double mx = fmax(room[0], fmax(room[1], room[2])), max_dist = fmin(obst ? 5.0 : 30.0, mx), min_dist = max_dist / 2;
        if (tick % control_steps == 0 || tick == 1) {
            double low[3] = {-room[0] / 2, -room[1] / 2, obst ? 1.5 : 0}, high[3] = {room[0] / 2, room[1] / 2, obst ? 3.0 : room[2]};
            double np_[3][2];
            for (int it = 0; it < 100000; ++it) {
                double u[6];   /* uniform(low=-high, high=high, size=(2,3)).reshape(3,2): flat order u0..u5 -> [[u0,u1],[u2,u3],[u4,u5]] */
                if (e->tape) { for (int k = 0; k < 6; ++k) u[k] = tape_pop(e); }
                else { for (int k = 0; k < 6; ++k) { int ax = k % 3; u[k] = rng_uniform1(e, QS_SITE_SCEN, 300 + 8 * it + k, 0, 0, -high[ax], high[ax]); } }
                int lo_i = (int)floor(min_dist), hi_i = (int)max_dist + 1, r;   /* np.random.randint(min_dist, max_dist + 1): a float low is truncated */
                if (e->tape) r = (int)tape_pop(e);
                else { r = lo_i + (int)(rng_uniform1(e, QS_SITE_SCEN, 300 + 8 * it + 6, 0, 0, 0.0, 1.0) * (hi_i - lo_i)); if (r >= hi_i) r = hi_i - 1; }
                int ok = 1;
                for (int col = 0; col < 2; ++col) {
                    double v[3] = {u[0 + col], u[2 + col], u[4 + col]}, n = norm3(v);
                    for (int row = 0; row < 3; ++row) {
                        np_[row][col] = e->goals[0][row] + v[row] * r / n;
                        if (!(np_[row][col] > low[row] + 0.5) || !(np_[row][col] < high[row] - 0.5)) ok = 0;
                    }
                }
                if (ok) break;
            }
            for (int row = 0; row < 3; ++row) { e->bez[row] = e->goals[0][row]; e->bez[3 + row] = np_[row][0]; e->bez[6 + row] = np_[row][1]; }
            e->bez_valid = 1;
        }
        if (tick % control_steps != 0 && tick > 1) {
            /* bezier.Curve(nodes, degree=2).evaluate_multi(linspace(0,1,control_steps))[:, t]: quadratic Bernstein form */
            double sv = (double)t / (double)(control_steps - 1), om = 1.0 - sv;
            double g0[3];
            for (int row = 0; row < 3; ++row) g0[row] = om * om * e->bez[row] + 2.0 * om * sv * e->bez[3 + row] + sv * sv * e->bez[6 + row];
            for (int i = 0; i < N; ++i) memcpy(e->goals[i], g0, sizeof g0);
            e->num_goals = N;
            set_all_goals(e);
        }
        break;
    }
    case QS_SCENARIO_O_DYNAMIC_SAME_GOAL:    /* scenarios/obstacles/o_dynamic_same_goal.py:17-28 */
        if ((e->control_step_for_sec > 0 && tick % e->control_step_for_sec == 0) || tick == 1) {
            double ng[3];
            for (int it = 0; it < 100000; ++it) {
                pos_obst_map_1(e, ng, 5000 + 2 * it);
                double df[3] = {e->end_point[0] - ng[0], e->end_point[1] - ng[1], e->end_point[2] - ng[2]};
                if (!(norm3(df) > 4.0)) break;
            }
            memcpy(e->end_point, ng, sizeof ng);
            for (int i = 0; i < N; ++i) memcpy(e->goals[i], ng, sizeof ng);
            set_all_goals(e);
        }
        break;
    default: break;   /* static_same_goal, static_diff_goal, o_static_same_goal, o_random (its step re-assigns the same goals) */
    }
}

/* ------------------------------------------------------------------------------------------ */
/* reset: QuadrotorEnvMulti.reset quadrotor_multi.py:339-411 + QuadrotorSingle._reset            */
/* quadrotor_single.py:387-447 + obst_generation_given_density quadrotor_multi.py:304-325       */
/* ------------------------------------------------------------------------------------------ */
static void env_reset(qso_env *e, double *obs) {
    const qs_config *c = &e->c;
    int N = c->num_agents;
    if (c->use_obstacles) {
        /* --quads_domain_random: the wrapper draws this episode's density / size before reset(obst_density, obst_size) */
        if (c->dr_num_density > 0 && !e->tape) {
            int k = (int)(rng_uniform1(e, QS_SITE_REPLAY, 2, 0, 0, 0.0, 1.0) * c->dr_num_density); if (k >= c->dr_num_density) k = c->dr_num_density - 1;
            e->cur_M = c->dr_obst_count[k]; e->cur_density = c->dr_density[k];
        }
        if (c->dr_num_size > 0 && !e->tape) {
            int k = (int)(rng_uniform1(e, QS_SITE_REPLAY, 3, 0, 0, 0.0, 1.0) * c->dr_num_size); if (k >= c->dr_num_size) k = c->dr_num_size - 1;
            e->cur_size = c->dr_size[k];
        }
        int L = c->obst_area[0], W = c->obst_area[1], M = e->cur_M, cells = L * W;
        for (int k = M; k < c->num_obstacles; ++k) { e->info.obst_pos[k][0] = 1e6; e->info.obst_pos[k][1] = 1e6; }   /* unused slots: parked far away, like the device's */
        qso_cell_centers(L, W, e->cell_centers);
        int ids[QS_MAX_OBSTACLES];
        if (e->tape) { for (int k = 0; k < M; ++k) ids[k] = (int)tape_pop(e); }
        else {
            int pool[64 * 64];
            for (int k = 0; k < cells; ++k) pool[k] = k;
            for (int k = 0; k < M; ++k) {
                int j = k + (int)(rng_uniform1(e, QS_SITE_OBST_MAP, k, 0, 0, 0.0, 1.0) * (cells - k));
                if (j >= cells) j = cells - 1;
                int t = pool[k]; pool[k] = pool[j]; pool[j] = t; ids[k] = pool[k];
            }
        }
        memset(e->obst_map, 0, sizeof e->obst_map);
        for (int k = 0; k < M; ++k) {
            int rid = ids[k] / W, cid = ids[k] - (ids[k] / W) * W;
            e->obst_map[rid][cid] = 1;
            int ci = rid + L * cid;
            e->obst_xy[k][0] = e->cell_centers[2 * ci]; e->obst_xy[k][1] = e->cell_centers[2 * ci + 1];
            e->info.obst_pos[k][0] = e->obst_xy[k][0]; e->info.obst_pos[k][1] = e->obst_xy[k][1];
        }
    }
    scenario_reset(e);
    for (int i = 0; i < N; ++i) {
        drone_t *d = &e->d[i];
        memcpy(d->goal, e->goals[i], sizeof d->goal);
        memcpy(d->spawn_point, e->have_spawn_points ? e->spawn_points[i] : e->goals[i], sizeof d->spawn_point);
        double u[3];
        rng_uniform(e, QS_SITE_SPAWN, 0, i, 0, 3, -c->spawn_box, c->spawn_box, u);
        for (int k = 0; k < 3; ++k) d->pos[k] = u[k] + d->spawn_point[k];
        if (d->pos[2] < 0.75) d->pos[2] = 0.75;
        for (int k = 0; k < 3; ++k) { d->vel[k] = 0; d->omega[k] = 0; d->acc[k] = 0; }
        /* yaw rejection loop quadrotor_single.py:431-434 (to_xyhat / normalize quad_utils.py:80-86,:120-124) */
        double xy[3] = {-d->pos[0], -d->pos[1], 0}, n = norm3(xy);
        if (n >= 0.00001) { xy[0] /= n; xy[1] /= n; }
        for (int t = 0; t < 256; ++t) {
            double th = rng_uniform1(e, QS_SITE_SPAWN_YAW, t, i, 0, -PI, PI);
            yaw_rot(th, d->rot);
            if (!(d->rot[0] * xy[0] + d->rot[3] * xy[1] < 0.5)) break;
        }
        for (int m = 0; m < 4; ++m) { d->rot_damp[m] = 0; d->cmds_damp[m] = 0; }
        d->flags = F_COL_AGENT_OK | F_COL_OBST_OK;
        if (c->floor_mode == QS_FLOOR_NUMPY) d->flags |= F_OMEGA_F32;
        d->dist_len = 0;
        e->tick = 0;
        self_obs(e, i, 0, obs + (size_t)i * e->obs_dim);
        memcpy(e->mpos[i], d->pos, sizeof e->mpos[i]); /* NB: mvel is NOT refreshed here (App. A reset quirk) */
    }
    neighbor_obs(e, obs);
    obstacle_obs(e, obs);
    memset(e->prev_pair, 0, sizeof e->prev_pair);
    memset(e->info.counters, 0, sizeof e->info.counters);
    e->info.num_resets += 1;
}

static int popcount64(uint64_t x) { int c = 0; while (x) { x &= x - 1; ++c; } return c; }

static double mean_tail(const double *h, int len, int w) { /* np.mean(list[-w:]) with pairwise summation */
    int s = len > w ? len - w : 0, n = len - s;
    if (n <= 0) return NAN;
    /* numpy pairwise sum: blocks of <=128 summed with 8 accumulators; fine to tolerance 1e-12 */
    double acc = 0;
    for (int k = s; k < len; ++k) acc += h[k];
    return acc / n;
}

/* ------------------------------------------------------------------------------------------ */
/* step: QuadrotorEnvMulti.step quadrotor_multi.py:413-724 (order of operations: SURVEY App. A) */
/* ------------------------------------------------------------------------------------------ */
void qso_step(qso_env *e, const double *actions, double *obs, double *rew, uint8_t *done, double *rew_info) {
    const qs_config *c = &e->c;
    int N = c->num_agents, any_done = 0;
    double ri_local[MAXN][QS_RI_COUNT];
    double control_dt = c->dt * c->sim_steps;
    e->step_ctr += 1;
    memset(ri_local, 0, sizeof ri_local);

    /* 1. per-drone step: QuadrotorSingle._step quadrotor_single.py:341-357 */
    int tick_before = e->tick;
    double time_remain = c->ep_len - tick_before;
    for (int i = 0; i < N; ++i) {
        dynamics_step(e, i, actions + 4 * i);
        rew[i] = reward_single(e, i, actions + 4 * i, ri_local[i]);
        done[i] = (tick_before + 1) > c->ep_len;
        any_done |= done[i];
        self_obs(e, i, 0, obs + (size_t)i * e->obs_dim);
        memcpy(e->mpos[i], e->d[i].pos, sizeof e->mpos[i]);
    }
    e->tick = tick_before + 1;
    int tick = e->tick;

    /* 2. drone-drone detection: calculate_collision_matrix collisions/quadrotors.py:63-91 + :432-459 */
    uint64_t curr_pair[MAXN], new_pair[MAXN], curr_ids = 0, prev_ids = 0;
    double prox[MAXN];
    int any_near = 0;
    memset(prox, 0, sizeof prox);
    for (int i = 0; i < N; ++i) { curr_pair[i] = 0; }
    for (int i = 0; i < N; ++i)
        for (int j = i + 1; j < N; ++j) {
            double dx = e->mpos[i][0] - e->mpos[j][0], dy = e->mpos[i][1] - e->mpos[j][1], dz = e->mpos[i][2] - e->mpos[j][2];
            double dist = pow(dx * dx + dy * dy + dz * dz, 0.5);
            if (dist <= c->collision_threshold) { curr_pair[i] |= 1ull << j; curr_ids |= (1ull << i) | (1ull << j); }
            if (dist <= c->collision_falloff_threshold) { /* calculate_drone_proximity_penalties :95-103 */
                double pr = -c->rew_coeff[QS_REW_QUADCOL_SMOOTH_MAX] / c->collision_falloff_threshold;
                double pen = pr * dist + c->rew_coeff[QS_REW_QUADCOL_SMOOTH_MAX];
                prox[i] += pen; prox[j] += pen; any_near = 1;
            }
        }
    for (int i = 0; i < N; ++i) {
        new_pair[i] = curr_pair[i] & ~e->prev_pair[i];
        if (e->prev_pair[i]) { prev_ids |= 1ull << i; prev_ids |= e->prev_pair[i]; }
    }
    uint64_t unique = curr_ids & ~prev_ids;              /* np.setdiff1d on flattened ids (:440) */
    int col_tick = popcount64(unique) / 2;                /* :448 */
    e->info.counters[QS_CNT_COLLISIONS] += col_tick;
    if (col_tick > 0 && tick >= 1.5 * (1.0 / control_dt)) {
        e->info.counters[QS_CNT_COLLISIONS_AFTER_SETTLE] += col_tick;
        for (int i = 0; i < N; ++i) if (unique >> i & 1) e->d[i].flags &= ~F_COL_AGENT_OK;
    }
    if (col_tick > 0 && time_remain <= 5.0 * (1.0 / control_dt)) e->info.counters[QS_CNT_COLLISIONS_FINAL_5S] += col_tick;
    for (int i = 0; i < N; ++i) { e->info.col_pair_mask[i] = curr_pair[i]; e->info.new_pair_mask[i] = new_pair[i]; e->prev_pair[i] = curr_pair[i]; }
    e->info.unique_col_mask = unique;

    /* 3. obstacles: MultiObstacles.collision_detection obstacles/obstacles.py:37-47 + :462-488 */
    uint64_t obst_hit = 0, obst_new = 0;
    double rew_obst_raw[MAXN];
    memset(rew_obst_raw, 0, sizeof rew_obst_raw);
    if (c->use_obstacles) {
        uint64_t prev_hit = 0;
        for (int i = 0; i < N; ++i) {
            if (e->d[i].flags & F_PREV_OBST) prev_hit |= 1ull << i;
            int o = qso_obst_first_hit(e->mpos[i], &e->obst_xy[0][0], e->cur_M, c->arm + e->cur_size / 2.0);
            e->info.obst_hit_idx[i] = o;
            if (o >= 0) obst_hit |= 1ull << i;
        }
        obst_new = obst_hit & ~prev_hit;
        int cnt = popcount64(obst_new);
        e->info.counters[QS_CNT_OBST] += cnt;
        if (cnt > 0 && tick >= 1.5 * (1.0 / control_dt)) {
            e->info.counters[QS_CNT_OBST_AFTER_SETTLE] += cnt;
            for (int i = 0; i < N; ++i) if (obst_new >> i & 1) {
                double q = norm3(obs + (size_t)i * e->obs_dim);
                if (q > 3.5) e->info.counters[QS_CNT_OBST_DIST_3_5] += 1;
                if (q > 5.0) e->info.counters[QS_CNT_OBST_DIST_5] += 1;
                e->d[i].flags &= ~F_COL_OBST_OK;
            }
        }
        for (int i = 0; i < N; ++i) { if (obst_hit >> i & 1) e->d[i].flags |= F_PREV_OBST; else e->d[i].flags &= ~F_PREV_OBST; }
        if (obst_hit) for (int i = 0; i < N; ++i) if (obst_new >> i & 1) rew_obst_raw[i] = -1.0;
    }
    e->info.obst_new_mask = obst_new; e->info.obst_hit_mask = obst_hit;

    /* 4. room: calculate_room_collision quadrotor_multi.py:289-302 + :491-497 */
    uint64_t floor_l = 0, wall_l = 0, ceil_l = 0, room_l = 0;
    for (int i = 0; i < N; ++i) {
        uint32_t f = e->d[i].flags;
        if (f & F_CRASH_FLOOR) floor_l |= 1ull << i;
        if ((f & F_CRASH_WALL) && !(f & F_PREV_WALL)) wall_l |= 1ull << i;
        if ((f & F_CRASH_CEIL) && !(f & F_PREV_CEIL)) ceil_l |= 1ull << i;
    }
    for (int i = 0; i < N; ++i) {
        uint32_t *f = &e->d[i].flags;
        int in_room = ((floor_l | wall_l | ceil_l) >> i) & 1;
        if (in_room && !(*f & F_PREV_ROOM)) room_l |= 1ull << i;
    }
    for (int i = 0; i < N; ++i) {
        uint32_t *f = &e->d[i].flags;
        *f &= ~(F_PREV_WALL | F_PREV_CEIL | F_PREV_ROOM);
        if (wall_l >> i & 1) *f |= F_PREV_WALL;
        if (ceil_l >> i & 1) *f |= F_PREV_CEIL;
        if (room_l >> i & 1) *f |= F_PREV_ROOM;
    }
    e->info.room_new_mask = room_l;

    /* 5. rewards (:499-546) */
    if (tick >= 1.5 * (1.0 / control_dt)) {
        e->info.counters[QS_CNT_ROOM] += popcount64(room_l); e->info.counters[QS_CNT_FLOOR] += popcount64(floor_l);
        e->info.counters[QS_CNT_WALL] += popcount64(wall_l); e->info.counters[QS_CNT_CEILING] += popcount64(ceil_l);
    }
    int any_nonzero_id = (unique & ~1ull) != 0; /* `.any()` on the id array: false when the only id is 0 */
    for (int i = 0; i < N; ++i) {
        double raw = (any_nonzero_id && (unique >> i & 1)) ? -1.0 : 0.0;
        double rc = c->rew_coeff[QS_REW_QUADCOL_BIN] * raw;
        double rp = any_near ? -1.0 * (control_dt * prox[i]) : 0.0;
        rew[i] += rc;
        rew[i] += rp;
        ri_local[i][QS_RI_REW_QUADCOL] = rc; ri_local[i][QS_RI_REW_PROXIMITY] = rp; ri_local[i][QS_RI_RAW_QUADCOL] = raw;
        if (c->use_obstacles) {
            double ro = c->rew_coeff[QS_REW_QUADCOL_OBST] * rew_obst_raw[i];
            rew[i] += ro;
            ri_local[i][QS_RI_REW_QUADCOL_OBST] = ro; ri_local[i][QS_RI_RAW_QUADCOL_OBST] = rew_obst_raw[i];
        }
        drone_t *d = &e->d[i];
        d->dist_hist[d->dist_len++] = -ri_local[i][QS_RI_RAW_POS];
        if (d->dist_len >= 5 && mean_tail(d->dist_hist, d->dist_len, 5) / c->dt < e->approach_metric && !(d->flags & F_REACHED))
            d->flags |= F_REACHED;
    }

    /* 6. physical interactions (:548-587) */
    int update_flag = 0;
    if (c->use_downwash) update_flag |= downwash(e);
    for (int i = 0; i < N; ++i)
        for (int j = i + 1; j < N; ++j)
            if (new_pair[i] >> j & 1) { update_flag = 1; collide_drones(e, i, j); }
    if (c->use_obstacles)
        for (int i = 0; i < N; ++i) if (obst_new >> i & 1) { update_flag = 1; collide_obstacle(e, i, e->info.obst_hit_idx[i]); }
    if (wall_l || ceil_l) {
        update_flag = 1;
        for (int i = 0; i < N; ++i) if (wall_l >> i & 1) collide_room(e, i, 1);
        for (int i = 0; i < N; ++i) if (ceil_l >> i & 1) collide_room(e, i, 0);
    }

    /* 7. scenario step, final obs (:590-607) */
    scenario_step(e);
    for (int i = 0; i < N; ++i) { memcpy(e->mpos[i], e->d[i].pos, sizeof e->mpos[i]); memcpy(e->mvel[i], e->d[i].vel, sizeof e->mvel[i]); }
    if (update_flag) for (int i = 0; i < N; ++i) self_obs(e, i, 1, obs + (size_t)i * e->obs_dim);
    neighbor_obs(e, obs);
    obstacle_obs(e, obs);

    for (int i = 0; i < N; ++i) {
        e->info.flags[i] = e->d[i].flags;
        memcpy(e->info.acc[i], e->d[i].acc, sizeof e->info.acc[i]);
        if (rew_info) memcpy(rew_info + (size_t)i * QS_RI_COUNT, ri_local[i], sizeof ri_local[i]);
    }

    /* 8. done: episode stats (:626-718), reset (:720-722) */
    if (any_done) {
        int cf = (int)(1.0 / control_dt + 0.5);
        for (int i = 0; i < N; ++i) {
            drone_t *d = &e->d[i];
            e->info.ep_stats[i][QS_EPS_DIST_1S] = (1.0 / c->dt) * mean_tail(d->dist_hist, d->dist_len, 1 * cf);
            e->info.ep_stats[i][QS_EPS_DIST_3S] = (1.0 / c->dt) * mean_tail(d->dist_hist, d->dist_len, 3 * cf);
            e->info.ep_stats[i][QS_EPS_DIST_5S] = (1.0 / c->dt) * mean_tail(d->dist_hist, d->dist_len, 5 * cf);
            e->info.ep_stats[i][QS_EPS_REACHED_GOAL] = (d->flags & F_REACHED) ? 1.0 : 0.0;
            e->info.ep_stats[i][QS_EPS_COL_AGENT_OK] = (d->flags & F_COL_AGENT_OK) ? 1.0 : 0.0;
            e->info.ep_stats[i][QS_EPS_COL_OBST_OK] = (d->flags & F_COL_OBST_OK) ? 1.0 : 0.0;
            done[i] = 1;
        }
        memcpy(e->info.ep_counters, e->info.counters, sizeof e->info.counters);
        e->info.ep_scenario = e->scen;
        env_reset(e, obs);
        for (int i = 0; i < N; ++i) e->info.flags[i] = e->d[i].flags;
    }
    e->info.tick = e->tick;
    e->info.scenario = e->scen;
}

void qso_reset(qso_env *e, double *obs_out) {
    double *tmp = obs_out ? obs_out : (double *)malloc(sizeof(double) * e->c.num_agents * e->obs_dim);
    e->step_ctr += 1;   /* an explicit reset takes the next draw counter, like a step (include/quadswarm.h: qs_reset) */
    env_reset(e, tmp);
    e->info.tick = e->tick;
    for (int i = 0; i < e->c.num_agents; ++i) e->info.flags[i] = e->d[i].flags;
    e->info.scenario = e->scen;
    if (!obs_out) free(tmp);
}

/* calculate_collision_matrix collisions/quadrotors.py:63-91 as a stand-alone function (reference KAT
 * collisions/test/unit_test/quadrotor.py:6-51): per-drone flag + masks of partners j>i within the threshold */
void qso_collision_matrix(const double *pos, int32_t n, double thr, int32_t *flag, uint64_t *pair_mask) {
    for (int i = 0; i < n; ++i) { flag[i] = 0; pair_mask[i] = 0; }
    for (int i = 0; i < n; ++i)
        for (int j = i + 1; j < n; ++j) {
            double dx = pos[3 * i] - pos[3 * j], dy = pos[3 * i + 1] - pos[3 * j + 1], dz = pos[3 * i + 2] - pos[3 * j + 2];
            if (pow(dx * dx + dy * dy + dz * dz, 0.5) <= thr) { flag[i] = 1; flag[j] = 1; pair_mask[i] |= 1ull << j; }
        }
}

size_t qso_sizeof_config(void) { return sizeof(qs_config); }
size_t qso_sizeof_info(void) { return sizeof(qso_info); }

qso_env *qso_create(const qs_config *cfg, int32_t env_global_id) {
    if (cfg->num_agents < 1 || cfg->num_agents > MAXN || cfg->num_obstacles > QS_MAX_OBSTACLES) return NULL;
    if (cfg->use_obstacles && (cfg->obst_area[0] > 64 || cfg->obst_area[1] > 64)) return NULL;
    qso_env *e = (qso_env *)calloc(1, sizeof *e);
    e->c = *cfg;
    e->env_id = env_global_id;
    e->self_dim = self_obs_dim(cfg->obs_repr);
    e->obs_dim = qso_obs_dim(cfg);
    e->cur_M = cfg->num_obstacles; e->cur_size = cfg->obst_size; e->cur_density = cfg->obst_density;
    for (int i = 0; i < cfg->num_agents; ++i) e->d[i].dist_hist = (double *)calloc((size_t)cfg->ep_len + 8, sizeof(double));
    e->cell_centers = (double *)calloc(64 * 64 * 2, sizeof(double));
    return e;
}

void qso_destroy(qso_env *e) {
    if (!e) return;
    for (int i = 0; i < e->c.num_agents; ++i) free(e->d[i].dist_hist);
    free(e->cell_centers);
    free(e);
}

void qso_set_tape(qso_env *e, const double *tape, int64_t n) { e->tape = tape; e->tape_n = n; e->tape_i = 0; }
int64_t qso_tape_pos(const qso_env *e) { return e->tape_i; }
void qso_get_info(const qso_env *e, qso_info *out) { *out = e->info; }
void qso_set_reward_coeffs(qso_env *e, const double *coeffs) { memcpy(e->c.rew_coeff, coeffs, sizeof e->c.rew_coeff); }
void qso_set_numpy126_quirk(qso_env *e, int32_t on) { e->np126 = on ? 1 : 0; }

void qso_get_state(const qso_env *e, double *s, int32_t *tick) {
    for (int i = 0; i < e->c.num_agents; ++i, s += QS_STATE_STRIDE) {
        const drone_t *d = &e->d[i];
        memcpy(s, d->pos, 24); memcpy(s + 3, d->vel, 24); memcpy(s + 6, d->rot, 72); memcpy(s + 15, d->omega, 24);
        memcpy(s + 18, d->rot_damp, 32); memcpy(s + 22, d->cmds_damp, 32); memcpy(s + 26, d->ou, 32);
        s[30] = (d->flags & F_ON_FLOOR) ? 1.0 : 0.0; s[31] = d->svd_count; memcpy(s + 32, d->goal, 24);
    }
    if (tick) *tick = e->tick;
}

void qso_set_state(qso_env *e, const double *s, int32_t tick) {
    for (int i = 0; i < e->c.num_agents; ++i, s += QS_STATE_STRIDE) {
        drone_t *d = &e->d[i];
        memcpy(d->pos, s, 24); memcpy(d->vel, s + 3, 24); memcpy(d->rot, s + 6, 72); memcpy(d->omega, s + 15, 24);
        memcpy(d->rot_damp, s + 18, 32); memcpy(d->cmds_damp, s + 22, 32); memcpy(d->ou, s + 26, 32);
        if (s[30] != 0.0) d->flags |= F_ON_FLOOR; else d->flags &= ~F_ON_FLOOR;
        d->svd_count = (int32_t)s[31];
        d->since_last_svd = 0; for (int k = 0; k < d->svd_count; ++k) d->since_last_svd += e->c.dt;
        memcpy(d->goal, s + 32, 24);
        d->flags &= ~F_OMEGA_F32;
    }
    if (tick >= 0) e->tick = tick;
}

/* CPU-baseline rollout: every env runs `steps` control steps back to back inside ONE parallel region (envs are
 * independent, so there is no per-step fork/join).  actions: ring[ring_len][num][N*4]; obs/rew/done receive the last
 * step's outputs ([num][N*obs_dim] etc.).  Each env steps into thread-local scratch and publishes its last outputs once:
 * the shared `done` / `rew` arrays hold 8 / 64 bytes per env, so stepping straight into them makes neighbouring envs (=
 * different threads) fight over cache lines on every step.  threads <= 0: the OpenMP default. */
void qso_rollout_batch_threads(qso_env **envs, int32_t num, const double *actions, int32_t ring_len, int32_t steps,
                               double *obs, double *rew, uint8_t *done, int32_t threads) {
#ifdef _OPENMP
    if (threads <= 0) threads = omp_get_max_threads();
#pragma omp parallel num_threads(threads)
#endif
    {
        double *lobs = NULL, lrew[MAXN];
        uint8_t ldone[MAXN];
        size_t lobs_n = 0;
#ifdef _OPENMP
#pragma omp for schedule(dynamic, 1)
#endif
        for (int32_t k = 0; k < num; ++k) {
            qso_env *e = envs[k];
            const int N = e->c.num_agents;
            const size_t per_step = (size_t)num * N * 4, need = (size_t)N * e->obs_dim;
            if (need > lobs_n) { free(lobs); lobs = (double *)malloc(need * sizeof(double)); lobs_n = need; }
            for (int32_t t = 0; t < steps; ++t)
                qso_step(e, actions + per_step * (size_t)(t % ring_len) + (size_t)k * N * 4, lobs, lrew, ldone, NULL);
            if (steps > 0) {
                memcpy(obs + (size_t)k * need, lobs, need * sizeof(double));
                memcpy(rew + (size_t)k * N, lrew, (size_t)N * sizeof(double));
                memcpy(done + (size_t)k * N, ldone, (size_t)N);
            }
        }
        free(lobs);
    }
}

void qso_rollout_batch(qso_env **envs, int32_t num, const double *actions, int32_t ring_len, int32_t steps,
                       double *obs, double *rew, uint8_t *done) {
    qso_rollout_batch_threads(envs, num, actions, ring_len, steps, obs, rew, done, 0);
}

void qso_step_batch(qso_env **envs, int32_t num, const double *actions, double *obs, double *rew, uint8_t *done) {
#ifdef _OPENMP
#pragma omp parallel for schedule(static)
#endif
    for (int32_t k = 0; k < num; ++k) {
        qso_env *e = envs[k];
        int N = e->c.num_agents;
        qso_step(e, actions + (size_t)k * N * 4, obs + (size_t)k * N * e->obs_dim, rew + (size_t)k * N, done + (size_t)k * N, NULL);
    }
}
