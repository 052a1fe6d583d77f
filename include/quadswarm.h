/*
 * quadswarm.h - C ABI of the MI355X-native vectorised QuadSwarm environment stepper.
 *
 * The reference (Zhehui-Huang/quad-swarm-rl) has NO FFI boundary: its hot path is the Python
 * object protocol `QuadrotorEnvMulti.reset()/step(actions)` registered with Sample Factory
 *   - gym_art/quadrotor_multi/quadrotor_multi.py:339-411  (reset)
 *   - gym_art/quadrotor_multi/quadrotor_multi.py:413-724  (step, auto-reset inside step)
 *   - swarm_rl/env_wrappers/quad_utils.py:20-117          (make_quadrotor_env[_multi], env factory)
 * This header is the boundary a maintainer would bind underneath that protocol (ctypes stub in
 * INTEGRATION.md).  Plain pointers and sizes only; every device pointer is a hipMalloc'd address
 * owned by the library and valid until qs_destroy().
 *
 * E environments x N drones are stepped at once.  Drone d of environment e has flat index e*N+d.
 * Per-drone arrays are struct-of-arrays (coalesced across lanes).  Outputs and statistics are component-major: component c
 * of a field with C components lives at field[c*E*N + e*N + d].  The dynamic STATE (pos ... col_pair_mask) is wave-blocked:
 * see the comment in qs_buffers and qs_state_array_copy.  Observations are what the policy consumes and are stored
 * row-major [E*N, obs_dim] like the reference returns them.
 *
 * The same `qs_config` / entry-point shapes are mirrored by the CPU oracle (oracle/quadswarm_oracle.h,
 * test infrastructure only) so parity tests drive both through identical calls.
 */
#ifndef QUADSWARM_H
#define QUADSWARM_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define QS_VERSION 100            /* 0.1.0 */
#define QS_MAX_AGENTS 64          /* per environment (pair/id sets are 64-bit masks) */
#define QS_MAX_OBSTACLES 64
#define QS_MAX_DR_CHOICES 8       /* choices of obstacle density / size under --quads_domain_random */

/* obs_repr (quad_utils.py:30-34 QUADS_OBS_REPR) */
enum { QS_OBS_XYZ_VXYZ_R_OMEGA = 0, QS_OBS_XYZ_VXYZ_R_OMEGA_FLOOR = 1, QS_OBS_XYZ_VXYZ_R_OMEGA_WALL = 2 };
/* scenario (scenarios/mix.py:31 create_scenario) */
enum { QS_SCENARIO_STATIC_SAME_GOAL = 0, QS_SCENARIO_O_STATIC_SAME_GOAL = 1, QS_SCENARIO_SWARM_VS_SWARM = 2,
       QS_SCENARIO_STATIC_DIFF_GOAL = 3, QS_SCENARIO_DYNAMIC_SAME_GOAL = 4, QS_SCENARIO_DYNAMIC_DIFF_GOAL = 5,
       QS_SCENARIO_DYNAMIC_FORMATIONS = 6, QS_SCENARIO_SWAP_GOALS = 7, QS_SCENARIO_EP_LISSAJOUS3D = 8,
       QS_SCENARIO_EP_RAND_BEZIER = 9, QS_SCENARIO_O_RANDOM = 10, QS_SCENARIO_O_DYNAMIC_SAME_GOAL = 11,
       QS_SCENARIO_O_SWAP_GOALS = 12, QS_SCENARIO_MIX = 13, QS_SCENARIO_O_EP_RAND_BEZIER = 14,
       QS_SCENARIO_RUN_AWAY = 15, QS_SCENARIO_COUNT = 16 };
/* floor_mode: which of the two reference semantics */
enum { QS_FLOOR_NUMBA = 0, QS_FLOOR_NUMPY = 1 };
/* precision of the device state / arithmetic */
enum { QS_PRECISION_F32 = 0, QS_PRECISION_F64 = 1 };
/* status codes */
enum { QS_OK = 0, QS_ERR_INVALID = -1, QS_ERR_HIP = -2, QS_ERR_NAN_REWARD = -3, QS_ERR_UNSUPPORTED = -4 };

/* indices into qs_config.rew_coeff (quadrotor_multi.py:91-94) */
enum { QS_REW_POS = 0, QS_REW_EFFORT, QS_REW_CRASH, QS_REW_ORIENT, QS_REW_SPIN,
       QS_REW_QUADCOL_BIN, QS_REW_QUADCOL_SMOOTH_MAX, QS_REW_QUADCOL_OBST, QS_REW_COUNT };

/* columns of the per-drone reward-info matrix (quadrotor_single.py:68-85, quadrotor_multi.py:533-540) */
enum { QS_RI_REW_MAIN = 0, QS_RI_REW_POS, QS_RI_REW_ACTION, QS_RI_REW_CRASH, QS_RI_REW_ORIENT, QS_RI_REW_SPIN,
       QS_RI_RAW_MAIN, QS_RI_RAW_POS, QS_RI_RAW_ACTION, QS_RI_RAW_CRASH, QS_RI_RAW_ORIENT, QS_RI_RAW_SPIN,
       QS_RI_REW_QUADCOL, QS_RI_REW_PROXIMITY, QS_RI_RAW_QUADCOL, QS_RI_REW_QUADCOL_OBST, QS_RI_RAW_QUADCOL_OBST,
       QS_RI_COUNT };

/* per-environment integer counters (quadrotor_multi.py:143-161, :626-718) */
enum { QS_CNT_COLLISIONS = 0, QS_CNT_COLLISIONS_AFTER_SETTLE, QS_CNT_COLLISIONS_FINAL_5S,
       QS_CNT_ROOM, QS_CNT_FLOOR, QS_CNT_WALL, QS_CNT_CEILING,
       QS_CNT_OBST, QS_CNT_OBST_AFTER_SETTLE, QS_CNT_OBST_DIST_3_5, QS_CNT_OBST_DIST_5, QS_CNT_COUNT };

/* per-drone episode statistics written when an episode ends (quadrotor_multi.py:626-718) */
/* rows of qs_buffers.run_sums / ep_sums: [0, QS_RI_COUNT) = sum of each reward term over the episode,
 * then sum a_k (k = 0..3) and sum a_k^2 of the raw policy actions */
#define QS_SUM_ACT (QS_RI_COUNT)
#define QS_SUM_ACT2 (QS_RI_COUNT + 4)
#define QS_SUM_COUNT (QS_RI_COUNT + 8)
enum { QS_EPS_DIST_1S = 0, QS_EPS_DIST_3S, QS_EPS_DIST_5S, QS_EPS_REACHED_GOAL, QS_EPS_COL_AGENT_OK,
       QS_EPS_COL_OBST_OK, QS_EPS_COUNT };

typedef struct qs_config {
    /* ---- batch ---- */
    int32_t num_envs;          /* E (this shard) */
    int32_t num_agents;        /* N, cfg.quads_num_agents */
    int32_t env_id_offset;     /* global id of local env 0 (multi-GPU sharding; RNG is keyed by global id) */
    int32_t precision;         /* QS_PRECISION_* */
    uint64_t seed;

    /* ---- airframe constants (quad_models.py:1-42 -> inertia.py:182-309 -> quadrotor_dynamics.py:104-166) ---- */
    double mass, inertia[3], arm;
    double prop_cross[4][3];   /* prop_pos x z_hat */
    double prop_ccw[4];
    double thrust_max[4], torque_max[4];
    double motor_tau_up, motor_tau_down;   /* 4*dt/(damp_time+1e-6), quadrotor_dynamics.py:63-64 */
    double motor_linearity, vel_damp, damp_omega_quadratic, omega_max, gravity;
    double thrust_noise_sigma; /* 0.2*thrust_noise_ratio, OU sigma (quadrotor_dynamics.py:168-173) */
    double ou_theta;           /* 0.15 */
                               /* qs_default_config() fills both with their FLOAT32 roundings widened back - (double)(float)0.01, (double)(float)0.15 -
                                * because its floor_mode is QS_FLOOR_NUMBA and the jitted reference keeps them in float32 jitclass members
                                * (numba_utils.py:67-74).  A caller that switches floor_mode to QS_FLOOR_NUMPY (--quads_use_numba=False) must also
                                * store the plain doubles 0.2 * ratio and 0.15 here (the numpy path's OUNoise, quad_utils.py:253-279; 2e-8 / 4e-8
                                * away); the Python binding does both together (config.make_config: numba_float32_ou follows use_numba). */

    /* ---- simulation ---- */
    double dt;                 /* 1/sim_freq = 0.005 */
    int32_t sim_steps;         /* 2 */
    int32_t ep_len;            /* int(ep_time/(dt*sim_steps)) */
    double room_lo[3], room_hi[3];
    int32_t floor_mode;        /* QS_FLOOR_* */
    int32_t svd_period;        /* sub-steps between re-orthogonalisations = first n with fl(sum_n dt) > 0.5 */

    /* ---- sensor noise (sensor_noise.py:69-110); sense_noise=0 bypasses ---- */
    int32_t sense_noise;
    int32_t obs_repr;          /* QS_OBS_* */
    double pos_norm_std, pos_unif_range, vel_norm_std, vel_unif_range;
    double quat_norm_std, quat_unif_range, gyro_noise_density;

    /* ---- multi-drone env (quadrotor_multi.py:24-207) ---- */
    int32_t num_neighbors;     /* K = resolved neighbor_visible_num (N-1 if -1; 0 if obs type 'none') */
    int32_t use_downwash;
    int32_t use_obstacles;
    int32_t scenario;          /* QS_SCENARIO_* */
    double collision_threshold;          /* hitbox_radius * arm */
    double collision_falloff_threshold;  /* falloff_radius * arm */
    double rew_coeff[QS_REW_COUNT];
    double spawn_box;          /* 2.0, or 0.1 with obstacles (quadrotor_single.py:215-218) */
    double approach_goal_metric; /* 0.5, 1.0 for obstacle scenarios */
    double nbr_clip_pos[3], nbr_clip_vel[3]; /* quadrotor_single.py:294-295 */

    /* ---- obstacles (quadrotor_multi.py:117-131, :304-325) ---- */
    double obst_size, obst_density;
    int32_t obst_area[2];      /* int(obst_spawn_area) */
    int32_t num_obstacles;     /* int(density*area0*area1) */

    /* ---- outputs ---- */
    int32_t write_rew_info;    /* 1: fill qs_buffers.rew_info every step (the infos[i]['rewards'] terms the SF
                                  reward-shaping wrapper logs); 0: skip those 17 stores per drone */
    int32_t episode_sums;      /* 1: accumulate, per drone and episode, the sums the SF reward-shaping wrapper keeps on the
                                  host (swarm_rl/env_wrappers/reward_shaping.py:78-110): the 17 reward terms and the
                                  first / second moments of the 4 actions; snapshot into qs_buffers.ep_sums at done */

    /* ---- per-episode obstacle randomisation, --quads_domain_random (quad_experience_replay.py:75-88,:106-118,:191-206 ->
       QuadrotorEnvMulti.reset(obst_density, obst_size), quadrotor_multi.py:339-351).  Every reset of an environment draws one
       of dr_num_density densities (np.arange(min, max, 0.05)) and one of dr_num_size sizes (np.arange(min, max, 0.1)); 0 = that
       quantity is fixed.  num_obstacles above is then the LARGEST count: unused obstacle slots are parked far outside the room. */
    int32_t dr_num_density, dr_num_size;
    int32_t dr_obst_count[QS_MAX_DR_CHOICES];   /* int(cells * density_k) */
    double dr_density[QS_MAX_DR_CHOICES], dr_size[QS_MAX_DR_CHOICES];
} qs_config;

/* Device pointers (element type = float for QS_PRECISION_F32, double for QS_PRECISION_F64 where
 * marked `real`).  Valid until qs_destroy(). */
typedef struct qs_buffers {
    void *obs;            /* real  [E*N, obs_dim] row-major */
    void *reward;         /* real  [E*N] */
    void *done;           /* uint8 [E*N] */
    void *rew_info;       /* real  [QS_RI_COUNT, E*N] */
    void *actions;        /* real  [E*N, 4] staging buffer callers may fill instead of passing their own */
    /* state: WAVE-BLOCKED.  Block b holds the drones of the envs_per_block = 64 / N environments one wavefront steps
       (envs b * envs_per_block ..., lane = local env * N + drone); inside a block the state arrays follow each other (64 lanes x
       components each; blocks state_block_bytes apart).  Each pointer below is its array inside block 0.  Inside an array the element
       order is one of two, fixed per handle (state_lane_major):
           0 - rows: every component is a row of 64 elements
               element (component c, env e, drone i)  =  ptr + (e / envs_per_block) * state_block_bytes
                                                             + (c * 64 + (e % envs_per_block) * N + i) * sizeof(element)
           1 - lane-major: the components of one drone are adjacent
               element (component c, env e, drone i)  =  ptr + (e / envs_per_block) * state_block_bytes
                                                             + (((e % envs_per_block) * N + i) * components + c) * sizeof(element)
       (quad-swarm-rl_amd/native.py: Stepper.to_host / from_host present them as plain [components, E*N] arrays through
       qs_state_array_copy, whatever the order).  Why: everything a wave loads and stores per step is one 11 KB chunk of HBM.  The
       specialised 8-wave team kernels (N <= 8, batches that do not fill the chip: the latency regime) run lane-major handles - one 12- /
       16-byte access per array and lane, 14 per direction instead of 45; the throughput kernels keep the rows (csrc/qs_kernels.h). */
    void *pos, *vel, *omega;  /* real, 3 components */
    void *rot;                /* real, 9 components: row-major R */
    void *thrust_rot_damp, *thrust_cmds_damp, *ou_state; /* real, 4 components */
    void *goal;               /* real, 3 components */
    void *flags;              /* uint32, 1 component: bit0 on_floor, bit1 crashed_floor, bit2 crashed_wall,
                                 bit3 crashed_ceiling, bit4 prev_new_wall, bit5 prev_new_ceiling,
                                 bit6 prev_new_room, bit7 obst_hit_prev */
    void *obst_hit_idx;       /* int32 [E*N]: first obstacle hit this step, -1 none (obstacles/utils.py:31-43) */
    void *col_pair_mask;      /* uint64, 1 component, wave-blocked like the state: bit j set <=> pair (d,j), j>d, within collision_threshold */
    void *new_pair_mask;      /* uint64 [E*N]: pairs new this step (quadrotor_multi.py:437-438) */
    void *unique_col_mask;    /* uint64 [E]: ids of last_step_unique_collisions (quadrotor_multi.py:440) */
    void *obst_new_mask;      /* uint64 [E]: curr_quad_col (quadrotor_multi.py:467) */
    void *room_new_mask;      /* uint64 [E]: room_crash_list (quadrotor_multi.py:492-493) */
    void *counters;           /* int32 [QS_CNT_COUNT, E] */
    void *tick;               /* int32 [E] */
    void *obst_pos;           /* real  [2, E*num_obstacles] */
    void *ep_stats;           /* real  [QS_EPS_COUNT, E*N], ep_counters int32 [QS_CNT_COUNT, E]: snapshot at last done */
    void *ep_counters;
    void *error_flag;         /* uint32 [1]: nonzero if a reward was NaN/Inf (quadrotor_single.py:87-90) */
    void *scenario_id;        /* int32 [E]: active scenario (the sub-scenario chosen by `mix` for this episode) */
    void *ep_scenario;        /* int32 [E]: scenario of the last finished episode (names the per-scenario episode stats) */
    void *run_sums;           /* real [QS_SUM_COUNT, E*N]: running sums of the current episode (episode_sums = 1) */
    void *ep_sums;            /* real [QS_SUM_COUNT, E*N]: the sums of the last finished episode */
    void *obst_count;         /* int32 [E]: obstacles of the running episode (<= num_obstacles; domain randomisation) */
    void *obst_size_env;      /* real  [E]: obstacle size of the running episode */
    void *obst_density_env;   /* real  [E]: obstacle density of the running episode (statistics only) */
    int32_t obs_dim;
    int32_t real_size;        /* 4 or 8 */
    int32_t state_block_bytes;/* bytes between the wave blocks of the state arrays */
    int32_t envs_per_block;   /* 64 / N */
    int32_t state_lane_major; /* element order inside the blocked state arrays (see above) */
} qs_buffers;

typedef struct qs_handle qs_handle;

/* Library version (QS_VERSION). */
int qs_version(void);

/* sizeof(qs_config) as compiled into the library (binding sanity check). */
size_t qs_sizeof_config(void);

/* Last error message of the calling thread's most recent failing call. */
const char *qs_last_error(void);

/* Fill *cfg with the reference defaults for the Crazyflie airframe and the given batch/env sizes
 * (swarm_rl/env_wrappers/quad_utils.py:20-65 + quadrotor_params.py:15-120).  Replaces the
 * QuadrotorEnvMulti.__init__ constant derivation (quadrotor_multi.py:24-207). */
int qs_default_config(qs_config *cfg, int32_t num_envs, int32_t num_agents);

/* obs_dim implied by a config (quad_utils.py:30-44 + quadrotor_single.py:311-316). */
int qs_obs_dim(const qs_config *cfg);

/* Create a stepper on HIP device `device`.  Replaces QuadrotorEnvMulti.__init__. */
int qs_create(const qs_config *cfg, int device, qs_handle **out);
int qs_destroy(qs_handle *h);

/* Episode reset (QuadrotorEnvMulti.reset, quadrotor_multi.py:339-411) of the envs whose byte in
 * env_mask_host is nonzero (NULL = all).  Asynchronous on `stream` (a hipStream_t, NULL = default).
 * Takes the env's next draw counter (like a step), so consecutive resets start different episodes. */
int qs_reset(qs_handle *h, const uint8_t *env_mask_host, void *stream);

/* One control step for all envs (QuadrotorEnvMulti.step, quadrotor_multi.py:413-724), including the
 * in-step auto-reset.  actions_dev: device pointer real[E*N,4] (NULL = use qs_buffers.actions).
 * Asynchronous on `stream`; results are in qs_buffers after the stream is synchronised. */
int qs_step(qs_handle *h, const void *actions_dev, void *stream);

/* K back-to-back control steps with actions_dev = real[K][E*N,4] (rollout with pre-generated actions;
 * one launch sequence, optionally replayed from a captured hipGraph). */
int qs_step_many(qs_handle *h, const void *actions_dev, int32_t k, void *stream);

/*
 * Resident-state stepping: the step kernel stays on the GPU across control steps.
 *
 * The reference steps its environments one `env.step(actions)` call at a time (quadrotor_multi.py:413); here that is one kernel launch
 * per control step, and at the BASELINE shapes (8192 drones = 128 workgroups) a launch is as long as the dependent-launch gap plus one
 * workgroup's critical path, of which ~1.4 us is the state's round trip through HBM.  qs_step_gated(h, k) instead launches ONE kernel for
 * k control steps that keeps the drone state in registers / LDS (the multi-step kernels of qs_step_many) and, per control step and
 * workgroup, WAITS until the step's actions are in the action ring and PUBLISHES when the step's outputs (observation rows, reward,
 * done, collision masks) are in HBM - through sequence words in device memory, so that the producer of the actions (the policy's
 * kernels on another stream) runs concurrently:
 *     producer, for sequence number s (1-based, counted since qs_gate_create), for the workgroups of group g:
 *         wait done_flag[w] >= s - 1 for the group's workgroups w   (closed loop: the policy reads the outputs of step s - 1)
 *         write the actions of the group's drones into  action_ring + ((s - 1) % ring_len) * action_stride_bytes   (real [E*N][4])
 *         make them visible at device scope (write-through stores / release), then act_flag[g] = s
 *     stepper (this library): waits act_flag[g] >= s, steps, writes the outputs through the L2, then done_flag[w] = s.
 * Workgroup w steps the environments [w * envs_per_workgroup, (w + 1) * envs_per_workgroup); group g = w / wg_per_group.  All waits
 * are bounded (QS_GATE_TIMEOUT_MS, default 500 ms of the device wall clock): a missing producer raises status bit 1 and the launch runs on
 * without waiting instead of hanging the GPU.  The producer MUST be able to run while the gated launch is resident: issue it on another,
 * NORMAL-priority stream.  The library runs the gated kernel on a highest-priority stream of its own - a different hardware queue pool -
 * behind `stream` (an event); nothing is made to wait for it unless asked (qs_gate_wait, qs_sync, a device synchronize): the runtime maps
 * the streams of a process onto a few hardware queues, a queue processes its packets in order, and a wait for the gated launch that lands
 * in the producer's queue AHEAD of the producer is a (bounded) deadlock - seen in one of three otherwise identical runs before this rule.  State is written back to HBM at the end of the launch (qs_get_state, snapshots and plain qs_step work between gated
 * launches).  Team kernels only (qs_kernel_flavor); not together with the replay wrapper, a noise tape or the fused exchange.
 * qs_gate_produce: the trivial producer used by bench.py and the tests - k steps of the protocol above with the action batches taken
 * round-robin from a table of n_src batches resident in HBM (closed_loop = 0: runs ahead, bounded only by the ring).
 * qs_gate_produce_verify: the same producer in closed-loop mode that ALSO consumes the stepper's outputs the way a policy would - after seeing
 * done_flag >= s for its workgroups it executes an agent-scope acquire (see "Visibility" below) and reads the observation rows and rewards of
 * step s - and writes sums_dev[t * groups + g] = the sum of their 32-bit words for step t of this call and group g (test instrument).
 * Visibility: the stepper stores its outputs with system-scope write-through stores and drains them before it raises done_flag.  A CONSUMER
 * that runs concurrently with the gated launch and has seen done_flag[w] >= s (a relaxed system-scope load; the flags live in uncached memory)
 * must execute an agent-scope acquire - `__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent")`, i.e. `buffer_inv sc1` - before it reads workgroup
 * w's rows of step s: the L2 of the consumer's XCD is not coherent with the stepper's and may still hold the rows of step s - 1.  A PRODUCER
 * writes its action rows with system-scope write-through stores (`global_store ... sc0 sc1`), waits for them (`s_waitcnt vmcnt(0)`), then stores
 * act_flag (the ring and the flags are uncached memory: no acquire is needed on the stepper's side).
 * qs_gate_status: out[0] = status bits (1 = action wait timed out, 2 = producer wait timed out), out[1] = steps launched,
 * out[2] = min act_flag, out[3] = min done_flag (synchronises the device).
 */
typedef struct qs_gate_info_t {
    void *action_ring; int64_t action_stride_bytes; int32_t ring_len;
    int32_t groups, wg_per_group, workgroups, envs_per_workgroup;
    unsigned long long *act_flag, *done_flag;
    int64_t steps_launched, steps_fed;
} qs_gate_info_t;
int qs_gate_create(qs_handle *h, int32_t ring_len, int32_t wg_per_group);
int qs_gate_info(qs_handle *h, qs_gate_info_t *out);
int qs_step_gated(qs_handle *h, int32_t k, void *stream);   /* ordered BEHIND `stream`; runs on the library's own queue */
int qs_gate_wait(qs_handle *h, void *stream);                /* orders `stream` behind the last gated launch (issue it AFTER the producer's work) */
int qs_gate_produce(qs_handle *h, const void *src_actions_dev, int32_t n_src, int32_t k, int32_t closed_loop, void *stream);
int qs_gate_produce_verify(qs_handle *h, const void *src_actions_dev, int32_t n_src, int32_t k, unsigned long long *sums_dev, void *stream);
int qs_gate_status(qs_handle *h, int64_t out[4]);

int qs_sync(qs_handle *h, void *stream);
int qs_get_buffers(qs_handle *h, qs_buffers *out);

/* Redirect the observation output: qs_step / qs_step_many / qs_reset calls issued after this one write their observation rows
 * (real [E*N, obs_dim] row-major, as QuadrotorEnvMulti.step returns them, quadrotor_multi.py:598-607) to obs_dev instead of
 * qs_buffers.obs; NULL restores the default.  Host-side state only (the pointer is a kernel argument of each launch), so
 * launches captured into a HIP graph keep the target they were recorded with.  Used by the multi-GPU exchange
 * (quadswarm_exchange.h): consecutive steps alternate between two staging buffers, so that the rows of step t can leave over
 * xGMI while step t+1 writes the other buffer.  Not available together with the device-side replay wrapper, whose
 * snapshots hold qs_buffers.obs (QS_ERR_UNSUPPORTED). */
int qs_set_obs_target(qs_handle *h, void *obs_dev);

/* Fused observation exchange (quadswarm_exchange.h): from now on every qs_step launch also stores its observation rows, in the wire
 * type of endpoint `xchg` (a qs_xchg*, fully attached), into slot [seq & 1][rank] of every rank's receive window and raises the
 * sequence flags - the multi-GPU "gather of observations per rollout step" with no launch of its own.  auto_ack != 0: the launch also
 * acts as the consumer (for callers that do not read the gathered rows in place).  xchg = NULL switches it off.  Needs the float32
 * team kernels (batches up to ~8 waves per CU, see qs_kernel_flavor) and rows * cols of the endpoint = E*N * obs_dim;
 * QS_ERR_UNSUPPORTED otherwise (large batches use qs_xchg_push on the rows instead). */
struct qs_xchg;
int qs_set_obs_exchange(qs_handle *h, struct qs_xchg *xchg, int32_t auto_ack);

/* Push new reward coefficients (the SF reward-shaping wrapper mutates env.rew_coeff,
 * swarm_rl/env_wrappers/reward_shaping.py:57-59,111-118). */
int qs_set_reward_coeffs(qs_handle *h, const double *coeffs /* [QS_REW_COUNT] */);

/* Copy the dynamic state of env `env` to / from host doubles (replaces the deepcopy snapshots of
 * quad_experience_replay.py:99-104; also the teacher-forcing hook of the parity tests).
 * Layout per drone d (stride QS_STATE_STRIDE doubles): pos3 vel3 rot9 omega3 rot_damp4 cmds_damp4 ou4
 * on_floor1 svd_count1 goal3.  */
#define QS_STATE_STRIDE 35
int qs_get_state(qs_handle *h, int32_t env, double *state_host /* [N*QS_STATE_STRIDE] */, int32_t *tick);
int qs_set_state(qs_handle *h, int32_t env, const double *state_host, int32_t tick);

/* Plain copies between host memory and any device pointer of qs_buffers (for callers without a HIP
 * runtime binding of their own, e.g. the ctypes stub of INTEGRATION.md).  Synchronous. */
int qs_memcpy_d2h(qs_handle *h, void *host_dst, const void *dev_src, size_t bytes);
int qs_memcpy_h2d(qs_handle *h, void *dev_dst, const void *host_src, size_t bytes);
/* One wave-blocked state array (a qs_buffers pointer: pos ... col_pair_mask) <-> a plain component-major host array
 * [comps, E*N] of `elem`-byte elements.  Synchronous. */
int qs_state_array_copy(qs_handle *h, void *host, void *dev_array, int32_t elem, int32_t comps, int32_t to_device);

/* Check the device NaN flag; returns QS_ERR_NAN_REWARD if set (maps to ValueError('QuadEnv: reward is Nan')). */
int qs_check_errors(qs_handle *h);

/* Time of the dominant kernel: average duration (ms) of the step kernel over the launches issued since
 * the last call, measured with HIP events recorded on the launch stream around each launch when
 * profiling is enabled with qs_set_profiling(h, 1). */
int qs_set_profiling(qs_handle *h, int32_t enable);
int qs_get_kernel_time(qs_handle *h, double *avg_ms, int64_t *launches);

/*
 * Noise tape (test instrument).  tape_host = [num_envs][len_per_env] float64: for every
 * environment the sequence of random draws the REFERENCE made (values in their final units, in its call order -
 * oracle/ref_harness/capture.py records them from the reference's numpy streams).  While a tape is set, qs_reset /
 * qs_step run a flavour of the kernels (of the handle's precision) in which every draw pops the tape instead of the
 * counter-based stream, so that a fixture captured from the reference (actions + tape + outputs) is replayed straight
 * through the HIP arithmetic - the method of gym_art/quadrotor_multi/tests/test_numba_opt.py:59-119, which compares two
 * implementations under identical injected noise: free-running in float64 (tests/test_hip_vs_reference.py, 1e-9), one
 * step at a time from the reference's recorded states in float32 (tests/test_hip_vs_reference_f32.py, 1e-5).  NULL / 0
 * returns to the Philox stream.  qs_get_tape_pos: draws consumed so far, per environment; qs_set_tape_pos: move the
 * cursors (teacher forcing: the recorded tape position of the step about to be replayed).
 */
int qs_set_noise_tape(qs_handle *h, const double *tape_host, int64_t len_per_env);
int qs_get_tape_pos(qs_handle *h, int32_t *pos_host /* [num_envs] */);
int qs_set_tape_pos(qs_handle *h, const int32_t *pos_host /* [num_envs] */);

/*
 * Environment snapshots: device-side deep copies of single environments, replacing `deepcopy(self.env)` of the
 * replay wrapper (gym_art/quadrotor_multi/quad_experience_replay.py:99-104, :176-187).  A snapshot holds every
 * per-drone and per-env array of one environment (state, flags, pair masks, counters, tick, goals, scenario state,
 * obstacle map, running episode sums, and the observation that goes with them) except the position in the noise
 * stream: a restored environment draws fresh noise, as the reference's does.  qs_snapshot_pool() (re)allocates
 * `slots` snapshot slots; save / load / copy are asynchronous on `stream`.
 */
int qs_snapshot_pool(qs_handle *h, int32_t slots);
int qs_snapshot_save(qs_handle *h, int32_t env, int32_t slot, void *stream);
int qs_snapshot_load(qs_handle *h, int32_t slot, int32_t env, void *stream);
int qs_snapshot_copy(qs_handle *h, int32_t src_slot, int32_t dst_slot, void *stream);

/*
 * Batched experience replay on the device: ExperienceReplayWrapper of the reference
 * (gym_art/quadrotor_multi/quad_experience_replay.py:66-209, wired at swarm_rl/env_wrappers/quad_utils.py:67-70) for ALL
 * environments of the handle at once.  Every environment has its own checkpoint ring (one checkpoint per 0.5 s, the last
 * 3 s = QS_REPLAY_RING slots) and its own buffer of QS_REPLAY_EVENTS collision events - what one wrapped env of the reference
 * has - as device-side snapshots (the arrays of qs_snapshot_*).  After qs_replay_enable(), every qs_step() is followed on the
 * same stream by one launch of the replay kernel, which per environment does what the wrapper's step() / new_episode() do:
 *   - an episode ended: count it; with probability sample_prob (if the env's replay buffer is active and not empty) restore a
 *     randomly chosen event - state, observation, counters of collisions zeroed (:176-187), replay count, clean-up of events
 *     replayed 10 times (:50-56) - else keep the fresh episode the step kernel's auto-reset started (the reference resets a
 *     second time there, :191-206: a redundant re-draw unless per-episode obstacle randomisation is on, see below);
 *   - otherwise: every 0.5 s save a checkpoint (:141-144); on a collision after the 1.5 s grace period, at most once per 5 s,
 *     file the checkpoint of 1.5 s ago as an event (:146-165) - and, like the reference (whose `obs` variable is rebound there),
 *     return that checkpoint's observation for this one step;
 *   - `activate_replay_buffer` (quadrotor_multi.py:284-287,:356-359): on after >= 10 recorded episodes with a mean crash
 *     reward above -1, both reset() calls of an episode end recording, as in the reference.
 * Draws come from the counter-based stream (QS_SITE_REPLAY).  Needs episode_sums = 1 (the per-episode crash reward).
 * qs_replay_stats copies out, per environment: [0] episodes, [1] replayed episodes, [2] events in the buffer, [3] sum of their
 * replay counts, [4] replay buffer active, [5] checkpoints in the ring, [6] filing attempts without 3 checkpoints (the
 * reference raises IndexError there), [7] the episode that ended last was a replayed one (its statistics are then the two
 * `*_replay` counters of quadrotor_multi.py:629-633), [8] control steps of that episode (a replayed one starts at its checkpoint's
 * tick).  qs_replay_set_active overrides the activation rule (NULL = all on).
 */
#define QS_REPLAY_RING 6
#define QS_REPLAY_EVENTS 20
#define QS_REPLAY_STATS 9
int qs_replay_enable(qs_handle *h, double sample_prob);
int qs_replay_stats(qs_handle *h, int32_t *stats_host /* [QS_REPLAY_STATS][num_envs] */);
int qs_replay_set_active(qs_handle *h, const uint8_t *active_host /* [num_envs] or NULL */);

/*
 * Config-specialised kernels (no counterpart in the reference; the analogue of Numba compiling the env's hot
 * functions for the argument types it sees, gym_art/quadrotor_multi/quadrotor_dynamics.py:498,:570).  Besides
 * the generic kernels of the library, qs_create() can run a code object compiled for exactly one
 * configuration - every constant a literal - which shortens the per-step critical path by ~25-30 %.
 * Environment: QS_SPEC=jit (default: build on first use with hipcc --genco, ~20 s, cached in
 * QS_SPEC_CACHE or <library dir>/spec_cache), =cache (use only if cached), =off (generic kernels).
 * Results are identical either way.  qs_spec_build() creates the cache entry ahead of time, without a GPU
 * (team: 1 = 4-wave kernels, 0 = single-wave, -1 = what qs_create picks on a 256-CU device).
 * The fallback to the generic kernels is loud: a warning on stderr, the reason from qs_spec_status(), and QS_SPEC=require makes
 * qs_create() fail instead (bench.py and the full-size tests assert that the specialised object is what ran).
 */
int qs_spec_build(const qs_config *cfg, int team, char *path_out, int cap);
/* Code-object verification.  The compiler of this image (ROCm 7.2) can place a VGPR spill, reload, copy or rematerialised constant at the top
 * of a control-flow join block IN FRONT OF the `s_or_b64 exec, exec, s[..]` that restores exec there; a wave that arrives with exec == 0 (it took
 * the branch around the `then` side) then spills nothing and later reloads stale scratch memory.  qs_create() and qs_spec_build() disassemble every
 * specialised object (llvm-objdump from QS_LLVM_BIN, default /opt/rocm/lib/llvm/bin), reject one that shows the pattern, rebuild it with other
 * scheduler settings and use none if all do (generic kernels, loudly); a verified cache entry carries a `.ok` file next to it.
 * qs_spec_verify() is that check for any file: a bundled or plain gfx950 code object, or a shared library with a .hip_fatbin section (every bundle
 * in it).  Returns 0 = clean, 1 = pattern found (one line per place in report_out: kernel <label>: the block's first instructions), < 0 = could not
 * be checked.  QS_SPEC_VERIFY=0 in the environment switches the check off for objects built in that process (tools that study a flagged object). */
int qs_spec_verify(const char *path, char *report_out, int cap);
/* The repair qs_create() / qs_spec_build() apply to an object that shows the pattern (and the Python build applies to the two libraries): the exec
 * restore is moved to the front of its block prologue - straight-line code nobody jumps into, all other instructions keep their relative order -
 * where that provably changes nothing else (no prologue instruction in front of it defines the mask it reads, no v_readlane near the prologue's
 * end feeding a memory / lane instruction right behind it, no DPP / lane operation right behind it, the byte sequence unique in the file).  Rewrites `path` in place; returns the
 * number of places repaired, the ones left (with the reason) in left_out; < 0 on errors.  Callers verify again afterwards. */
int qs_spec_repair(const char *path, char *left_out, int cap);
int qs_is_specialized(qs_handle *h);
/* 1 = config-specialised kernels, 0 = generic kernels with the reason written to why_out (NUL-terminated, at most cap bytes). */
int qs_spec_status(qs_handle *h, char *why_out, int cap);
/* which step kernel the handle launches: bit 0 = config-specialised, bit 1 = team kernels, bit 2 = full scenario set,
 * bits 8..15 = waves per workgroup (1, 4 or 8) */
int qs_kernel_flavor(qs_handle *h);

/*
 * Random numbers.  Every stochastic term of the reference is drawn from a
 * counter-based Philox4x32-10 generator:   bits = philox(counter={env_global, step_ctr, site|slot<<8,
 * i|j<<16}, key={seed_lo, seed_hi});  u = ((bits>>9)+0.5)*2^-23 (exact in fp32 and fp64);
 * normals by Box-Muller on word pairs (0,1) and (2,3).  Results therefore do not depend on how
 * environments are sharded over GPUs, and the CPU oracle generates bit-identical uniforms.
 */
enum { QS_SITE_OU = 0, QS_SITE_SENS_POS_N, QS_SITE_SENS_POS_U, QS_SITE_SENS_VEL_N, QS_SITE_SENS_VEL_U,
       QS_SITE_SENS_OMEGA_N, QS_SITE_SENS_THETA_N, QS_SITE_SENS_THETA_U, QS_SITE_FLOOR_YAW,
       QS_SITE_DW_I, QS_SITE_DW_IJ_V, QS_SITE_DW_IJ_W,
       QS_SITE_DD_N, QS_SITE_DD_U, QS_SITE_DD_W,
       QS_SITE_OBST_N, QS_SITE_OBST_U, QS_SITE_OBST_W,
       QS_SITE_WALL, QS_SITE_CEIL,
       QS_SITE_SPAWN, QS_SITE_SPAWN_YAW, QS_SITE_OBST_MAP, QS_SITE_SCEN, QS_SITE_SCEN_SHUFFLE,
       QS_SITE_REPLAY /* slot 0: replay-or-new-episode draw, slot 1: which event, slot 2 / 3: obstacle density / size choice */ };

#ifdef __cplusplus
}
#endif
#endif /* QUADSWARM_H */
