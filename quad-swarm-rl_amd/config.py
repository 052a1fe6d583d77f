"""`qs_config` (include/quadswarm.h) as a ctypes structure + the reference-style constructor arguments.

`make_config` takes the keyword arguments of `QuadrotorEnvMulti.__init__`
(gym_art/quadrotor_multi/quadrotor_multi.py:24-41) as `make_quadrotor_env_multi` passes them
(swarm_rl/env_wrappers/quad_utils.py:36-65) and derives the constants the reference derives in
`QuadrotorEnvMulti.__init__` / `QuadrotorSingle.__init__` (quadrotor_single.py:99-234).
"""
import ctypes as C

import numpy as np

from . import airframe

QS_MAX_AGENTS = 64
QS_MAX_OBSTACLES = 64
QS_STATE_STRIDE = 35

OBS_REPR = {"xyz_vxyz_R_omega": 0, "xyz_vxyz_R_omega_floor": 1, "xyz_vxyz_R_omega_wall": 2}
OBS_REPR_DIM = {"xyz_vxyz_R_omega": 18, "xyz_vxyz_R_omega_floor": 19, "xyz_vxyz_R_omega_wall": 24}  # quad_utils.py:30-34
SCENARIOS = {"static_same_goal": 0, "o_static_same_goal": 1, "swarm_vs_swarm": 2, "static_diff_goal": 3, "dynamic_same_goal": 4,
             "dynamic_diff_goal": 5, "dynamic_formations": 6, "swap_goals": 7, "ep_lissajous3D": 8, "ep_rand_bezier": 9,
             "o_random": 10, "o_dynamic_same_goal": 11, "o_swap_goals": 12, "mix": 13, "o_ep_rand_bezier": 14, "run_away": 15}
SCENARIO_CLASS_NAMES = {v: "Scenario_" + k for k, v in SCENARIOS.items()}
REW_COEFF_KEYS = ["pos", "effort", "crash", "orient", "spin", "quadcol_bin", "quadcol_bin_smooth_max", "quadcol_bin_obst"]
# quadrotor_multi.py:91-94
REW_COEFF_DEFAULT = dict(pos=1., effort=0.05, action_change=0., crash=1., orient=1., yaw=0., rot=0., attitude=0., spin=0.1,
                         vel=0., quadcol_bin=5., quadcol_bin_smooth_max=4., quadcol_bin_obst=5.)
REW_INFO_KEYS = ["rew_main", "rew_pos", "rew_action", "rew_crash", "rew_orient", "rew_spin",
                 "rewraw_main", "rewraw_pos", "rewraw_action", "rewraw_crash", "rewraw_orient", "rewraw_spin",
                 "rew_quadcol", "rew_proximity", "rewraw_quadcol", "rew_quadcol_obstacle", "rewraw_quadcol_obstacle"]
COUNTER_KEYS = ["collisions", "collisions_after_settle", "collisions_final_5s", "room", "floor", "wall", "ceiling",
                "obst", "obst_after_settle", "obst_dist_3_5", "obst_dist_5"]
EPS_KEYS = ["dist_1s", "dist_3s", "dist_5s", "reached_goal", "col_agent_ok", "col_obst_ok"]

PRECISION = {"f32": 0, "f64": 1}


class QsConfig(C.Structure):
    _fields_ = [
        ("num_envs", C.c_int32), ("num_agents", C.c_int32), ("env_id_offset", C.c_int32), ("precision", C.c_int32),
        ("seed", C.c_uint64),
        ("mass", C.c_double), ("inertia", C.c_double * 3), ("arm", C.c_double),
        ("prop_cross", (C.c_double * 3) * 4), ("prop_ccw", C.c_double * 4),
        ("thrust_max", C.c_double * 4), ("torque_max", C.c_double * 4),
        ("motor_tau_up", C.c_double), ("motor_tau_down", C.c_double),
        ("motor_linearity", C.c_double), ("vel_damp", C.c_double), ("damp_omega_quadratic", C.c_double),
        ("omega_max", C.c_double), ("gravity", C.c_double),
        ("thrust_noise_sigma", C.c_double), ("ou_theta", C.c_double),
        ("dt", C.c_double), ("sim_steps", C.c_int32), ("ep_len", C.c_int32),
        ("room_lo", C.c_double * 3), ("room_hi", C.c_double * 3),
        ("floor_mode", C.c_int32), ("svd_period", C.c_int32),
        ("sense_noise", C.c_int32), ("obs_repr", C.c_int32),
        ("pos_norm_std", C.c_double), ("pos_unif_range", C.c_double), ("vel_norm_std", C.c_double),
        ("vel_unif_range", C.c_double), ("quat_norm_std", C.c_double), ("quat_unif_range", C.c_double),
        ("gyro_noise_density", C.c_double),
        ("num_neighbors", C.c_int32), ("use_downwash", C.c_int32), ("use_obstacles", C.c_int32), ("scenario", C.c_int32),
        ("collision_threshold", C.c_double), ("collision_falloff_threshold", C.c_double),
        ("rew_coeff", C.c_double * 8),
        ("spawn_box", C.c_double), ("approach_goal_metric", C.c_double),
        ("nbr_clip_pos", C.c_double * 3), ("nbr_clip_vel", C.c_double * 3),
        ("obst_size", C.c_double), ("obst_density", C.c_double),
        ("obst_area", C.c_int32 * 2), ("num_obstacles", C.c_int32),
        ("write_rew_info", C.c_int32),
        ("episode_sums", C.c_int32),
        ("dr_num_density", C.c_int32), ("dr_num_size", C.c_int32), ("dr_obst_count", C.c_int32 * 8),
        ("dr_density", C.c_double * 8), ("dr_size", C.c_double * 8),
    ]


def svd_period(dt, limit=0.5):
    """Sub-steps between SVD re-orthogonalisations: first n with fl(sum_n dt) > limit
    (quadrotor_dynamics.py:547-551 accumulates `since_last_svd += dt` in float64)."""
    s, n = 0.0, 0
    while True:
        s += dt
        n += 1
        if s > limit:
            return n


def obs_dim(obs_repr, num_neighbors, use_obstacles):
    return OBS_REPR_DIM[obs_repr] + 6 * num_neighbors + (9 if use_obstacles else 0)


def make_config(num_envs=1, num_agents=8, ep_time=15.0, rew_coeff=None, obs_repr="xyz_vxyz_R_omega",
                neighbor_visible_num=-1, neighbor_obs_type="none",
                collision_hitbox_radius=2.0, collision_falloff_radius=-1.0,
                use_obstacles=False, obst_density=0.2, obst_size=1.0, obst_spawn_area=(6.0, 6.0),
                use_downwash=False, use_numba=False, quads_mode="static_same_goal", room_dims=(10.0, 10.0, 10.0),
                sense_noise="default", thrust_noise_ratio=0.05, sim_freq=200.0, sim_steps=2,
                seed=0, env_id_offset=0, precision="f32", write_rew_info=True, episode_sums=False,
                domain_random=False, obst_density_random=False, obst_size_random=False,
                obst_density_min=0.05, obst_density_max=0.2, obst_size_min=0.3, obst_size_max=0.6, numba_float32_ou=None):
    """Build a QsConfig.  Argument names/defaults follow the reference's `--quads_*` flags
    (swarm_rl/env_wrappers/quadrotor_params.py:15-120) and QuadrotorEnvMulti.__init__."""
    if num_agents < 1 or num_agents > QS_MAX_AGENTS:
        raise ValueError(f"num_agents must be in [1, {QS_MAX_AGENTS}]")
    if quads_mode not in SCENARIOS:
        raise NotImplementedError(f"Unknown/unsupported scenario {quads_mode!r}; supported: {sorted(SCENARIOS)}")
    if obs_repr not in OBS_REPR:
        raise ValueError(f"unknown obs_repr {obs_repr}")
    if quads_mode == "swarm_vs_swarm" and num_agents < 2:
        raise ValueError("swarm_vs_swarm needs >= 2 drones (scenarios/utils.py:12)")
    if quads_mode == "run_away" and num_agents < 2:
        raise ValueError("run_away re-targets drones 0 and 1 (scenarios/run_away.py:21-25): needs >= 2 drones")
    if quads_mode != "mix" and use_obstacles != quads_mode.startswith("o_"):
        raise ValueError("obstacle scenarios (o_*) require use_obstacles=True and vice versa")

    c = QsConfig()
    af = airframe.crazyflie(thrust_noise_ratio=thrust_noise_ratio, dt=1.0 / sim_freq)
    c.num_envs, c.num_agents, c.env_id_offset, c.precision, c.seed = num_envs, num_agents, env_id_offset, PRECISION[precision], seed
    c.mass, c.arm = af["mass"], af["arm"]
    c.inertia[:] = af["inertia"].tolist()
    for m in range(4):
        for k in range(3):
            c.prop_cross[m][k] = float(af["prop_cross"][m][k])
    c.prop_ccw[:] = af["prop_ccw"].tolist()
    c.thrust_max[:] = af["thrust_max"].tolist()
    c.torque_max[:] = af["torque_max"].tolist()
    for k in ("motor_tau_up", "motor_tau_down", "motor_linearity", "vel_damp", "damp_omega_quadratic", "omega_max",
              "gravity", "thrust_noise_sigma", "ou_theta"):
        setattr(c, k, af[k])
    # OUNoiseNumba is a jitclass whose theta / sigma / mu members are declared float32 (numba_utils.py:67-74): the numba path's thrust noise runs
    # with theta = float32(0.15), sigma = float32(0.2 * ratio) widened back to float64 - 4e-8 / 2e-8 away from the numpy path's OUNoise
    # (quad_utils.py:253-279).  Default: follows use_numba; pinned by the `*_f32ou` fixtures (captured with the jitclass stub emulating the float32
    # members); the older fixtures were captured under the plain stub and say numba_float32_ou=False (tests/golden_util.py).
    if numba_float32_ou is None:
        numba_float32_ou = bool(use_numba)
    if numba_float32_ou:
        c.thrust_noise_sigma = float(np.float32(c.thrust_noise_sigma))
        c.ou_theta = float(np.float32(c.ou_theta))
    c.dt = 1.0 / sim_freq
    c.sim_steps = sim_steps
    c.ep_len = int(ep_time / (c.dt * sim_steps))                      # quadrotor_single.py:158
    c.room_lo[:] = [-room_dims[0] / 2.0, -room_dims[1] / 2.0, 0.0]    # quadrotor_single.py:146-147
    c.room_hi[:] = [room_dims[0] / 2.0, room_dims[1] / 2.0, room_dims[2]]
    c.floor_mode = 0 if use_numba else 1
    c.svd_period = svd_period(c.dt)
    if sense_noise == "default":                                     # sensor_noise.py:70-76
        c.sense_noise = 1
        c.pos_norm_std, c.pos_unif_range, c.vel_norm_std, c.vel_unif_range = 0.005, 0.0, 0.01, 0.0
        c.quat_norm_std, c.quat_unif_range, c.gyro_noise_density = 0.0, 0.0, 0.000175
    elif sense_noise is None:
        c.sense_noise = 0
    else:
        raise ValueError("sense_noise must be 'default' or None")
    c.obs_repr = OBS_REPR[obs_repr]
    # quadrotor_multi.py:47-50 (+ quadrotor_single.py:312: neighbour obs only with obs type 'pos_vel')
    k = num_agents - 1 if neighbor_visible_num == -1 else neighbor_visible_num
    if neighbor_obs_type != "pos_vel":
        k = 0
    if k < 0 or k > num_agents - 1:
        raise RuntimeError("Incorrect number of neigbors")             # quadrotor_multi.py:274
    c.num_neighbors = k
    c.use_downwash, c.use_obstacles, c.scenario = int(use_downwash), int(use_obstacles), SCENARIOS[quads_mode]
    c.collision_threshold = collision_hitbox_radius * af["arm"]       # quadrotor_multi.py:154-155
    c.collision_falloff_threshold = collision_falloff_radius * af["arm"]
    coeff = dict(REW_COEFF_DEFAULT)
    if rew_coeff is not None:
        if not set(rew_coeff).issubset(coeff):
            raise AssertionError("unknown reward coefficient")
        coeff.update(rew_coeff)
    c.rew_coeff[:] = [float(coeff[key]) for key in REW_COEFF_KEYS]
    c.spawn_box = 0.1 if use_obstacles else 2.0                       # quadrotor_single.py:215-218
    c.approach_goal_metric = 1.0 if use_obstacles else 0.5            # scenarios/base.py:31, o_base.py:16
    room_range = [room_dims[0], room_dims[1], room_dims[2]]
    c.nbr_clip_pos[:] = room_range                                    # quadrotor_single.py:294
    c.nbr_clip_vel[:] = [2.0 * af["vxyz_max"]] * 3                    # quadrotor_single.py:295
    c.obst_size, c.obst_density = obst_size, obst_density
    c.obst_area[:] = [int(obst_spawn_area[0]), int(obst_spawn_area[1])]
    cells = int(obst_spawn_area[0]) * int(obst_spawn_area[1])
    c.num_obstacles = int(obst_density * obst_spawn_area[0] * obst_spawn_area[1]) if use_obstacles else 0
    # --quads_domain_random (quad_experience_replay.py:75-88): choices as np.arange builds them, counts as
    # obst_generation_given_density does (quadrotor_multi.py:304-313: int(num_room_grids * density))
    c.dr_num_density = c.dr_num_size = 0
    if domain_random and use_obstacles:
        if obst_density_random:
            dens = [float(d) for d in np.arange(obst_density_min, obst_density_max, 0.05)]
            if not 1 <= len(dens) <= 8:
                raise ValueError("obstacle density randomisation: 1..8 choices (np.arange(min, max, 0.05))")
            c.dr_num_density = len(dens)
            for k, d in enumerate(dens):
                c.dr_density[k], c.dr_obst_count[k] = d, int(cells * d)
            if min(c.dr_obst_count[k] for k in range(len(dens))) < 1:
                raise ValueError("obstacle density randomisation: every choice must give at least one obstacle")
            c.num_obstacles = max(c.dr_obst_count[k] for k in range(len(dens)))
        if obst_size_random:
            sizes = [float(x) for x in np.arange(obst_size_min, obst_size_max, 0.1)]
            if not 1 <= len(sizes) <= 8:
                raise ValueError("obstacle size randomisation: 1..8 choices (np.arange(min, max, 0.1))")
            c.dr_num_size = len(sizes)
            for k, x in enumerate(sizes):
                c.dr_size[k] = x
    c.write_rew_info = int(bool(write_rew_info))
    c.episode_sums = int(bool(episode_sums))
    if c.num_obstacles > QS_MAX_OBSTACLES:
        raise ValueError(f"more than {QS_MAX_OBSTACLES} obstacles")
    return c


def config_obs_dim(c):
    dims = {0: 18, 1: 19, 2: 24}
    return dims[c.obs_repr] + 6 * c.num_neighbors + (9 if c.use_obstacles else 0)


def obs_bounds(c):
    """Low/high of the observation Box (quadrotor_single.py:278-335)."""
    rr = np.array([c.room_hi[k] - c.room_lo[k] for k in range(3)])
    vmax, omax = 3.0, c.omega_max
    low = [-rr, -vmax * np.ones(3), -np.ones(9), -omax * np.ones(3)]
    high = [rr, vmax * np.ones(3), np.ones(9), omax * np.ones(3)]
    if c.obs_repr == 1:
        low.append(np.zeros(1)); high.append(c.room_hi[2] * np.ones(1))
    elif c.obs_repr == 2:
        low.append(np.zeros(6)); high.append(5.0 * np.ones(6))
    for _ in range(c.num_neighbors):
        low += [-rr, -2.0 * vmax * np.ones(3)]
        high += [rr, 2.0 * vmax * np.ones(3)]
    if c.use_obstacles:
        low.append(-10 * np.ones(9)); high.append(10 * np.ones(9))
    return np.concatenate(low).astype(np.float32), np.concatenate(high).astype(np.float32)
