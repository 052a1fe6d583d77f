/*
 * quadswarm_oracle.h - CPU oracle of the QuadSwarm env stepper.   *** TEST INFRASTRUCTURE ***
 *
 * A plain-C, float64, one-environment-at-a-time restatement of the reference algorithm
 * (gym_art/quadrotor_multi/ *.py, file:line cited at every function in quadswarm_oracle.c).
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use it - as the
 * checker / CPU baseline, never as a product path.
 *
 * Parity status: PINNED.  tests/test_oracle_vs_reference.py replays the golden fixtures captured
 * from the reference itself (oracle/ref_harness/capture.py; every random draw recorded on a
 * sequential "noise tape") through this code and requires <=1e-9 agreement on all outputs and exact
 * agreement on every flag / index / counter.
 */
#ifndef QUADSWARM_ORACLE_H
#define QUADSWARM_ORACLE_H

#include "../include/quadswarm.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct qso_env qso_env;

typedef struct qso_info {
    uint64_t unique_col_mask, obst_new_mask, obst_hit_mask, room_new_mask;
    uint64_t col_pair_mask[QS_MAX_AGENTS], new_pair_mask[QS_MAX_AGENTS];
    int32_t counters[QS_CNT_COUNT];
    int32_t ep_counters[QS_CNT_COUNT];
    int32_t obst_hit_idx[QS_MAX_AGENTS];
    uint32_t flags[QS_MAX_AGENTS];
    int32_t tick;
    int32_t num_resets;
    double ep_stats[QS_MAX_AGENTS][QS_EPS_COUNT];
    double obst_pos[QS_MAX_OBSTACLES][2];
    double acc[QS_MAX_AGENTS][3];
    int32_t nan_reward;
    int32_t tape_underrun;
    int32_t scenario, ep_scenario;   /* active scenario id (the sub-scenario under `mix`) now / of the last finished episode */
} qso_info;

size_t qso_sizeof_config(void);
size_t qso_sizeof_info(void);
qso_env *qso_create(const qs_config *cfg, int32_t env_global_id);
void qso_destroy(qso_env *e);

/* sequential reference tape (NULL => counter-based Philox, identical to the HIP stepper's stream) */
void qso_set_tape(qso_env *e, const double *tape, int64_t n);
int64_t qso_tape_pos(const qso_env *e);

void qso_reset(qso_env *e, double *obs_out /* [N*obs_dim] */);
void qso_step(qso_env *e, const double *actions /* [N*4] */, double *obs /* [N*obs_dim] */,
              double *rew /* [N] */, uint8_t *done /* [N] */, double *rew_info /* [N*QS_RI_COUNT] or NULL */);

void qso_get_state(const qso_env *e, double *state /* [N*QS_STATE_STRIDE] */, int32_t *tick);
void qso_set_state(qso_env *e, const double *state, int32_t tick);
void qso_get_info(const qso_env *e, qso_info *out);
void qso_set_reward_coeffs(qso_env *e, const double *coeffs);
/* numpy floor mode only: evaluate the omega damping factor the way NumPy 1.26's value-based casting does; default off */
void qso_set_numpy126_quirk(qso_env *e, int32_t on);

/* Batched convenience for the CPU baseline: step `num` independent envs (OpenMP over envs if built with it). */
void qso_step_batch(qso_env **envs, int32_t num, const double *actions, double *obs, double *rew, uint8_t *done);

void qso_rollout_batch(qso_env **envs, int32_t num, const double *actions, int32_t ring_len, int32_t steps,
                       double *obs, double *rew, uint8_t *done);
/* the same on `threads` OpenMP threads (<= 0: the OpenMP default) */
void qso_rollout_batch_threads(qso_env **envs, int32_t num, const double *actions, int32_t ring_len, int32_t steps,
                               double *obs, double *rew, uint8_t *done, int32_t threads);

/* exposed pieces for unit tests (known-answer tests of the reference's own test-suite) */
void qso_philox4x32(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]);
void qso_polar_rotation(const double r[9], double out[9]);
int qso_obs_dim(const qs_config *cfg);
void qso_collision_matrix(const double *pos, int32_t n, double thr, int32_t *flag, uint64_t *pair_mask);
void qso_cell_centers(int32_t length, int32_t width, double *out /* [length*width][2] */);
void qso_surround_sdf(const double qxy[2], const double *obst_xy, int32_t m, double radius, double res, double out[9]);
int qso_obst_first_hit(const double qxy[2], const double *obst_xy, int32_t m, double thr);
void qso_collision_obstacle_kat(const double pos[3], const double vel[3], const double opos[3], double *vnew, double norm[3]);

#ifdef __cplusplus
}
#endif
#endif
