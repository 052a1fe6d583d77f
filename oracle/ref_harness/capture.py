#!/usr/bin/env python
"""Golden-vector capture harness.  TEST INFRASTRUCTURE - runs ONLY in the build container.

Imports the reference (`/root/reference/gym_art`) through the stub modules in
`oracle/ref_harness/stubs` (numba/gymnasium/pyglet/bezier are not installed here), records
EVERY random draw the reference makes (the "sequential noise tape") together with the
inputs (actions, forced states) and the outputs (obs, rewards, dones, internal state,
collision bookkeeping, reward-info terms, episode stats) and writes small `.npz`
fixtures to `tests/golden/`.

The fixtures are data only (inputs + expected outputs).  Nothing from the reference
travels to the GPU box; `tests/` replays the tapes through `oracle/quadswarm_oracle.c`
to pin the C restatement against the reference.

Usage:  python oracle/ref_harness/capture.py [case ...]
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(HERE, "stubs"))
sys.path.insert(0, "/root/reference")

import gym_art.quadrotor_multi.quad_utils as ref_quad_utils  # noqa: E402
import gym_art.quadrotor_multi.sensor_noise as ref_sensor_noise  # noqa: E402
from gym_art.quadrotor_multi.quadrotor_multi import QuadrotorEnvMulti  # noqa: E402

GOLDEN_DIR = os.path.join(REPO, "tests", "golden")

REW_KEYS = ["rew_main", "rew_pos", "rew_action", "rew_crash", "rew_orient", "rew_spin",
            "rewraw_main", "rewraw_pos", "rewraw_action", "rewraw_crash", "rewraw_orient", "rewraw_spin",
            "rew_quadcol", "rew_proximity", "rewraw_quadcol", "rew_quadcol_obstacle", "rewraw_quadcol_obstacle"]


# ----------------------------------------------------------------------------------------------
# Recording RNG proxies.  Draw *return values* are recorded (flattened, as float64) in call order.
# ----------------------------------------------------------------------------------------------
class Tape:
    def __init__(self):
        self.vals = []

    def rec(self, x):
        self.vals.extend(np.asarray(x, dtype=np.float64).reshape(-1).tolist())
        return x

    def __len__(self):
        return len(self.vals)


TAPE = Tape()
_ORIG = {k: getattr(np.random, k) for k in ("normal", "uniform", "randn", "rand", "choice", "shuffle", "randint")}


def _normal(loc=0.0, scale=1.0, size=None):
    return TAPE.rec(_ORIG["normal"](loc, scale, size))


def _uniform(low=0.0, high=1.0, size=None):
    return TAPE.rec(_ORIG["uniform"](low, high, size))


def _randn(*shape):
    return TAPE.rec(_ORIG["randn"](*shape))


def _rand(*shape):
    return TAPE.rec(_ORIG["rand"](*shape))


def _randint(low, high=None, size=None, dtype=int):
    return TAPE.rec(_ORIG["randint"](low, high, size, dtype))


def _choice(a, size=None, replace=True, p=None):
    # record the *positions* chosen (a may be a python range / list / int)
    pop = a if isinstance(a, (int, np.integer)) else len(a)
    idx = _ORIG["choice"](pop, size, replace, p)
    TAPE.rec(idx)
    if isinstance(a, (int, np.integer)):
        return idx
    arr = np.asarray(a)
    return arr[idx]


def _shuffle(x):
    # record the permutation: new_x[k] = old_x[perm[k]]
    st = np.random.get_state()
    perm = np.arange(len(x))
    _ORIG["shuffle"](perm)
    np.random.set_state(st)
    _ORIG["shuffle"](x)
    TAPE.rec(perm)


class RecordingNpRandom:
    """Replacement for QuadrotorSingle.np_random (per-drone spawn noise, quadrotor_single.py:394)."""

    def __init__(self, seed):
        self._rs = np.random.RandomState(seed)

    def uniform(self, low=0.0, high=1.0, size=None):
        return TAPE.rec(self._rs.uniform(low, high, size))


def install_recorders():
    np.random.normal = _normal
    np.random.uniform = _uniform
    np.random.randn = _randn
    np.random.rand = _rand
    np.random.randint = _randint
    np.random.choice = _choice
    np.random.shuffle = _shuffle
    # names bound at import time (SURVEY 8c "capture pitfalls")
    ref_sensor_noise.normal = _normal
    ref_sensor_noise.uniform = _uniform
    assert ref_quad_utils.nr is np.random


# ----------------------------------------------------------------------------------------------
def default_cfg(**over):
    cfg = dict(
        num_agents=8, ep_time=15.0, obs_repr="xyz_vxyz_R_omega",
        neighbor_visible_num=6, neighbor_obs_type="pos_vel",
        collision_hitbox_radius=2.0, collision_falloff_radius=4.0,
        use_obstacles=False, obst_density=0.2, obst_size=0.6, obst_spawn_area=[8.0, 8.0],
        use_downwash=True, use_numba=True, quads_mode="static_same_goal", room_dims=[10.0, 10.0, 10.0],
        # reward coefficients as the SF wrapper would push them (reward_shaping.py:57-59)
        rew_coeff=dict(pos=1.0, effort=0.05, spin=0.1, vel=0.0, crash=1.0, orient=1.0, yaw=0.0,
                       quadcol_bin=5.0, quadcol_bin_smooth_max=10.0, quadcol_bin_obst=5.0),
        sense_noise="default", thrust_noise_ratio=0.05,
    )
    cfg.update(over)
    return cfg


def make_env(cfg):
    """Mirrors swarm_rl/env_wrappers/quad_utils.py:20-65 (make_quadrotor_env_multi).
    Two emulation switches for what this container cannot run (numba, NumPy 1.26 - the reference's pins, setup.py:14), both OFF unless a case asks:
      cfg["numba_float32_ou"]  - the jitclass stub rounds OUNoiseNumba's float32 members (theta, sigma, mu) to float32 like numba does
      cfg["numpy126_omega_quirk"] - `damp_omega_quadratic` becomes a Python float, so that NumPy 2's promotion (NEP 50: a Python scalar takes the
                                 array's dtype) evaluates `damp * omega ** 2 ... * dt` on a float32 omega in float32 - what NumPy 1.26's value-
                                 based casting does with the np.float64 scalar the reference holds (quadrotor_dynamics.py:322-325)"""
    import numba.experimental as nbx
    nbx.EMULATE_FLOAT32_FIELDS = bool(cfg.get("numba_float32_ou", False))
    dynamics_change = dict(noise=dict(thrust_noise_ratio=cfg["thrust_noise_ratio"]),
                           damp=dict(vel=0, omega_quadratic=0))
    env = QuadrotorEnvMulti(
        num_agents=cfg["num_agents"], ep_time=cfg["ep_time"], rew_coeff=dict(cfg["rew_coeff"]),
        obs_repr=cfg["obs_repr"],
        neighbor_visible_num=cfg["neighbor_visible_num"], neighbor_obs_type=cfg["neighbor_obs_type"],
        collision_hitbox_radius=cfg["collision_hitbox_radius"],
        collision_falloff_radius=cfg["collision_falloff_radius"],
        use_obstacles=cfg["use_obstacles"], obst_density=cfg["obst_density"], obst_size=cfg["obst_size"],
        obst_spawn_area=cfg["obst_spawn_area"],
        use_downwash=cfg["use_downwash"], use_numba=cfg["use_numba"], quads_mode=cfg["quads_mode"],
        room_dims=cfg["room_dims"], use_replay_buffer=False, quads_view_mode=["global"], quads_render=False,
        dynamics_params="Crazyflie", raw_control=True, raw_control_zero_middle=True,
        dynamics_randomize_every=None, dynamics_change=dynamics_change, dyn_sampler_1=None,
        sense_noise=cfg["sense_noise"], init_random_state=False,
    )
    for i, e in enumerate(env.envs):
        e.np_random = RecordingNpRandom(1000 + i)
        if cfg.get("numpy126_omega_quirk", False):
            e.dynamics.damp_omega_quadratic = float(e.dynamics.damp_omega_quadratic)
    nbx.EMULATE_FLOAT32_FIELDS = False
    return env


def ids_to_mask(ids):
    m = 0
    for i in np.asarray(ids, dtype=np.int64).reshape(-1):
        m |= (1 << int(i))
    return m


def pairs_to_list(pairs, n):
    """pair list [[i,j],...] -> int64[n] per-drone masks of partners j>i."""
    out = np.zeros(n, dtype=np.uint64)
    for p in np.asarray(pairs, dtype=np.int64).reshape(-1, 2):
        out[int(p[0])] |= np.uint64(1 << int(p[1]))
    return out


def snapshot(env):
    dyn = [e.dynamics for e in env.envs]
    return dict(
        pos=np.array([d.pos for d in dyn], dtype=np.float64),
        vel=np.array([d.vel for d in dyn], dtype=np.float64),
        rot=np.array([d.rot for d in dyn], dtype=np.float64),
        omega=np.array([d.omega for d in dyn], dtype=np.float64),
        acc=np.array([d.acc for d in dyn], dtype=np.float64),
        thrust_rot_damp=np.array([d.thrust_rot_damp for d in dyn], dtype=np.float64),
        thrust_cmds_damp=np.array([d.thrust_cmds_damp for d in dyn], dtype=np.float64),
        ou_state=np.array([d.thrust_noise.state for d in dyn], dtype=np.float64),
        since_last_svd=np.array([d.since_last_svd for d in dyn], dtype=np.float64),
        on_floor=np.array([d.on_floor for d in dyn], dtype=np.int8),
        crashed_floor=np.array([d.crashed_floor for d in dyn], dtype=np.int8),
        crashed_wall=np.array([d.crashed_wall for d in dyn], dtype=np.int8),
        crashed_ceiling=np.array([d.crashed_ceiling for d in dyn], dtype=np.int8),
        goal=np.array([e.goal for e in env.envs], dtype=np.float64),
        tick=np.array([e.tick for e in env.envs], dtype=np.int64),
    )


def run_case(name, cfg, steps, seed, action_fn=None, forces=None, with_extra=True, slim=False):
    """forces: dict step_index -> callable(env) mutating dynamics before that step; the resulting
    full (pos, vel, rot, omega) of every drone is stored so the oracle can apply the same override."""
    global TAPE
    TAPE = Tape()
    np.random.seed(seed)
    arng = np.random.RandomState(seed + 7919)  # actions come from a separate stream (not on the tape)
    env = make_env(cfg)
    n = cfg["num_agents"]

    obs0 = np.asarray(env.reset(), dtype=np.float64)
    tape_pos = [len(TAPE)]
    snaps0 = snapshot(env)
    rec = {k: [] for k in ("actions", "obs", "rew", "done", "rew_info", "unique_col", "new_pairs", "curr_pairs",
                           "obst_new", "obst_hit", "obst_hit_idx", "counters", "room_new")}
    # Which obstacle each drone hit (obstacles/utils.py:31-43: obstacles in index order, the first one within reach wins, `break`):
    # quadrotor_multi.py:463 keeps the {drone: obstacle} dict of MultiObstacles.collision_detection in a local, so the call is recorded.
    hit_pairs = []
    if cfg["use_obstacles"]:   # (on the class: reset() builds a new MultiObstacles for every episode, quadrotor_multi.py:349)
        from gym_art.quadrotor_multi.obstacles.obstacles import MultiObstacles
        inner_detect = _ORIG.setdefault("collision_detection", MultiObstacles.collision_detection)

        def recording_detect(self, pos_quads):
            ids, pair = inner_detect(self, pos_quads=pos_quads)
            hit_pairs.append({int(k): int(v) for k, v in pair.items()})
            return ids, pair
        MultiObstacles.collision_detection = recording_detect
    snaps = {k: [] for k in snaps0}
    force_steps, force_state = [], {k: [] for k in ("pos", "vel", "rot", "omega")}
    ep_stats = []
    obst_pos = []
    if cfg["use_obstacles"]:
        obst_pos.append(np.array(env.obstacles.pos_arr, dtype=np.float64))

    for t in range(steps):
        if forces and t in forces:
            forces[t](env)
            force_steps.append(t)
            for k in force_state:
                force_state[k].append(np.array([getattr(e.dynamics, k) for e in env.envs], dtype=np.float64))
        if action_fn is None:
            act = arng.uniform(-1.0, 1.0, size=(n, 4))
        else:
            act = np.asarray(action_fn(t, env, arng), dtype=np.float64)
        prev_pairs = np.array(env.prev_drone_collisions, dtype=np.int64).reshape(-1, 2)
        del hit_pairs[:]
        obs, rew, done, infos = env.step([a for a in act])
        tape_pos.append(len(TAPE))
        assert len(hit_pairs) == (1 if cfg["use_obstacles"] else 0)   # one detection pass per step (quadrotor_multi.py:463)
        rec["obst_hit_idx"].append(np.array([hit_pairs[0].get(i, -1) if hit_pairs else -1 for i in range(n)], dtype=np.int32))
        rec["actions"].append(act)
        rec["obs"].append(np.asarray(obs, dtype=np.float64))
        rec["rew"].append(np.asarray(rew, dtype=np.float64))
        rec["done"].append(np.asarray(done, dtype=np.int8))
        ri = np.zeros((n, len(REW_KEYS)))
        for i in range(n):
            for k, key in enumerate(REW_KEYS):
                ri[i, k] = infos[i]["rewards"].get(key, 0.0)
        rec["rew_info"].append(ri)
        if not any(done):
            # bookkeeping attributes are wiped by the in-step reset() when the episode ends
            rec["unique_col"].append(ids_to_mask(env.last_step_unique_collisions))
            curr = np.array(env.prev_drone_collisions, dtype=np.int64).reshape(-1, 2)  # prev <- curr (:459)
            rec["curr_pairs"].append(pairs_to_list(curr, n))
            old = set(map(tuple, prev_pairs))
            rec["new_pairs"].append(pairs_to_list([p for p in curr if tuple(p) not in old], n))
            rec["obst_new"].append(ids_to_mask(env.curr_quad_col) if cfg["use_obstacles"] else 0)
            rec["obst_hit"].append(ids_to_mask(env.prev_obst_quad_collisions) if cfg["use_obstacles"] else 0)
            rec["room_new"].append(ids_to_mask(env.prev_crashed_room))
        else:
            rec["unique_col"].append(-1)
            rec["curr_pairs"].append(np.zeros(n, dtype=np.uint64))
            rec["new_pairs"].append(np.zeros(n, dtype=np.uint64))
            rec["obst_new"].append(-1)
            rec["obst_hit"].append(-1)
            rec["room_new"].append(-1)
            st = infos[0]["episode_extra_stats"]
            ep_stats.append(dict(step=t, stats={k: float(v) for k, v in st.items()}))
            if cfg["use_obstacles"]:
                obst_pos.append(np.array(env.obstacles.pos_arr, dtype=np.float64))
        rec["counters"].append(np.array([
            env.collisions_per_episode, env.collisions_after_settle, env.collisions_final_5s,
            env.collisions_room_per_episode, env.collisions_floor_per_episode,
            env.collisions_wall_per_episode, env.collisions_ceiling_per_episode,
            getattr(env, "obst_quad_collisions_per_episode", 0),
            getattr(env, "obst_quad_collisions_after_settle", 0),
            getattr(env, "distance_to_goal_3_5", 0), getattr(env, "distance_to_goal_5", 0),
        ], dtype=np.int64))
        s = snapshot(env)
        for k in snaps:
            snaps[k].append(s[k])

    out = dict(
        cfg=np.array(json.dumps(cfg)), rew_keys=np.array(json.dumps(REW_KEYS)),
        ep_stats=np.array(json.dumps(ep_stats)),
        tape=np.array(TAPE.vals, dtype=np.float64), tape_pos=np.array(tape_pos, dtype=np.int64),
        obs0=obs0,
        force_steps=np.array(force_steps, dtype=np.int64),
    )
    for k in force_state:
        out["force_" + k] = np.array(force_state[k], dtype=np.float64).reshape((len(force_steps), n) + {
            "pos": (3,), "vel": (3,), "rot": (3, 3), "omega": (3,)}[k])
    for k, v in rec.items():
        out[k] = np.array(v)
    keep = ("pos", "goal", "tick", "on_floor") if slim else tuple(snaps0)   # slim fixtures: obs/rew/done + a few state rows
    for k, v in snaps0.items():
        if k in keep:
            out["s0_" + k] = v
    for k, v in snaps.items():
        if k in keep:
            out["s_" + k] = np.array(v)
    if slim:
        for k in ("rew_info", "curr_pairs", "new_pairs"):
            out.pop(k)
    if obst_pos:
        out["obst_pos"] = np.array(obst_pos)
    # constants (G1)
    d = env.envs[0].dynamics
    out["const"] = np.array(json.dumps(dict(
        mass=float(d.mass), inertia=[float(x) for x in d.inertia], arm=float(d.arm),
        prop_pos=np.asarray(d.prop_pos).tolist(), prop_crossproducts=np.asarray(d.prop_crossproducts).tolist(),
        prop_ccw=np.asarray(d.prop_ccw).tolist(), thrust_max=np.asarray(d.thrust_max).tolist(),
        torque_max=np.asarray(d.torque_max).tolist(), motor_tau_up=float(d.motor_tau_up),
        motor_tau_down=float(d.motor_tau_down), motor_linearity=float(d.motor_linearity),
        vel_damp=float(d.vel_damp), damp_omega_quadratic=float(d.damp_omega_quadratic),
        omega_max=float(d.omega_max), collision_threshold=float(env.collision_threshold),
        collision_falloff_threshold=float(env.collision_falloff_threshold),
        ep_len=int(env.envs[0].ep_len), obs_dim=int(obs0.shape[1]),
        obs_low=env.observation_space.low.tolist(), obs_high=env.observation_space.high.tolist(),
    )))
    os.makedirs(GOLDEN_DIR, exist_ok=True)
    path = os.path.join(GOLDEN_DIR, name + ".npz")
    np.savez_compressed(path, **out)
    print(f"{name}: steps={steps} tape={len(TAPE)} draws  -> {path} ({os.path.getsize(path) / 1024:.0f} KiB)")


# ----------------------------------------------------------------------------------------------
# forced-state helpers (crafted events)
# ----------------------------------------------------------------------------------------------
def yaw_rot(theta):
    c, s = np.cos(theta), np.sin(theta)
    return np.array([[c, -s, 0.0], [s, c, 0.0], [0.0, 0.0, 1.0]])


def set_dyn(env, i, pos=None, vel=None, rot=None, omega=None):
    d = env.envs[i].dynamics
    if pos is not None:
        d.pos = np.array(pos, dtype=np.float64)
    if vel is not None:
        d.vel = np.array(vel, dtype=np.float64)
    if rot is not None:
        d.rot = np.array(rot, dtype=np.float64)
    if omega is not None:
        d.omega = np.array(omega, dtype=np.float64)


def hover_actions(t, env, arng):
    n = len(env.envs)
    return np.full((n, 4), 0.06) + arng.uniform(-0.05, 0.05, size=(n, 4))


def events_actions(t, env, arng):
    a = hover_actions(t, env, arng)
    if t >= 60:
        a[4] = -0.8 + arng.uniform(-0.05, 0.05, size=4)   # keep drones 4, 5 on the floor after the crash
        a[5] = -0.9 + arng.uniform(-0.05, 0.05, size=4)
    return a


def force_collide(env):
    # drones 1&2 approach head-on, 3&5 already overlapping, 0&4 close (proximity only), 6 above 7 (downwash)
    set_dyn(env, 1, pos=[1.0, 1.0, 3.0], vel=[0.8, 0.0, 0.0])
    set_dyn(env, 2, pos=[1.12, 1.0, 3.0], vel=[-0.8, 0.1, 0.0])
    set_dyn(env, 3, pos=[-1.0, 1.0, 2.5], vel=[0.0, 0.3, 0.0])
    set_dyn(env, 5, pos=[-1.0, 1.05, 2.52], vel=[0.0, -0.3, 0.1])
    set_dyn(env, 0, pos=[2.0, -2.0, 2.0], vel=[0.0, 0.0, 0.0])
    set_dyn(env, 4, pos=[2.0, -1.85, 2.0], vel=[0.0, 0.0, 0.0])
    set_dyn(env, 6, pos=[-2.0, -2.0, 3.0], vel=[0.0, 0.0, 0.0], rot=yaw_rot(0.3))
    set_dyn(env, 7, pos=[-2.03, -1.98, 2.6], vel=[0.0, 0.0, 0.0])


def force_only_drone0_new(env):
    # pairs (0,1) new while (1,2) persists  ->  setdiff1d ids = {0}: the ".any()" quirk (App. A step 9)
    set_dyn(env, 0, pos=[0.0, 0.0, 4.0], vel=[0.0, 0.0, 0.0])
    set_dyn(env, 1, pos=[0.05, 0.0, 4.0], vel=[0.0, 0.0, 0.0])
    set_dyn(env, 2, pos=[0.10, 0.0, 4.0], vel=[0.0, 0.0, 0.0])


def force_pair12(env):
    set_dyn(env, 0, pos=[3.0, 3.0, 4.0], vel=[0.0, 0.0, 0.0])
    set_dyn(env, 1, pos=[0.05, 0.0, 4.0], vel=[0.0, 0.0, 0.0])
    set_dyn(env, 2, pos=[0.10, 0.0, 4.0], vel=[0.0, 0.0, 0.0])


def force_walls(env):
    set_dyn(env, 0, pos=[4.99, 0.0, 3.0], vel=[3.0, 0.5, 0.0])      # +x wall
    set_dyn(env, 1, pos=[-4.995, -4.99, 3.0], vel=[-2.0, -2.5, 0.2])  # corner
    set_dyn(env, 2, pos=[0.0, 1.0, 9.99], vel=[0.2, 0.0, 4.0])      # ceiling
    set_dyn(env, 3, pos=[1.0, 4.99, 9.99], vel=[0.0, 3.0, 3.0])     # wall + ceiling
    set_dyn(env, 4, pos=[0.5, 0.5, 0.2], vel=[0.3, 0.1, -2.0])      # floor, upright
    rot_flip = np.array([[1.0, 0.0, 0.0], [0.0, -1.0, 0.0], [0.0, 0.0, -1.0]])
    set_dyn(env, 5, pos=[-1.5, 0.5, 0.2], vel=[0.0, 0.4, -2.0], rot=rot_flip)  # floor, upside-down


def force_slide(env):
    # drone already resting on the floor gets horizontal velocity -> kinetic friction branch
    d = env.envs[4].dynamics
    set_dyn(env, 4, vel=[0.6, -0.3, 0.0])
    assert d.on_floor


def force_obstacles(env):
    op = np.array(env.obstacles.pos_arr)
    r = env.obst_size / 2.0
    set_dyn(env, 0, pos=[op[3, 0] + r + 0.06, op[3, 1], 2.0], vel=[-1.5, 0.2, 0.0])   # approaching obstacle 3
    set_dyn(env, 1, pos=[op[5, 0] + 0.1, op[5, 1] - 0.05, 1.5], vel=[0.3, 0.0, 0.0])  # inside obstacle 5
    set_dyn(env, 2, pos=[op[0, 0], op[0, 1] - r - 0.03, 3.0], vel=[0.0, 0.5, 0.0])    # grazing obstacle 0


CASES = {}


def case(fn):
    CASES[fn.__name__] = fn
    return fn


@case
def c1_single_numpy():
    run_case("c1_single_numpy", default_cfg(num_agents=1, neighbor_visible_num=0, neighbor_obs_type="none",
                                            use_downwash=False, use_numba=False,
                                            collision_falloff_radius=-1.0,
                                            rew_coeff=dict(pos=1.0, effort=0.05, spin=0.1, vel=0.0, crash=1.0,
                                                           orient=1.0, yaw=0.0, quadcol_bin=0.0,
                                                           quadcol_bin_smooth_max=0.0, quadcol_bin_obst=0.0)),
             steps=1000, seed=11)   # BASELINE config 1 as written: 1000 steps


C1_NUMPY = dict(num_agents=1, neighbor_visible_num=0, neighbor_obs_type="none", use_downwash=False, use_numba=False, collision_falloff_radius=-1.0,
                rew_coeff=dict(pos=1.0, effort=0.05, spin=0.1, vel=0.0, crash=1.0, orient=1.0, yaw=0.0, quadcol_bin=0.0, quadcol_bin_smooth_max=0.0,
                               quadcol_bin_obst=0.0))


@case
def c1_single_numpy_np126():
    # config 1 under the NumPy-1.26 promotion of the omega damping term (see make_env): random actions crash the drone within a few dozen steps and
    # keep it bouncing - every floor crash / reset leaves a float32 omega for one sub-step
    run_case("c1_single_numpy_np126", default_cfg(numpy126_omega_quirk=True, **C1_NUMPY), steps=1000, seed=11)


@case
def c1_single_numba():
    run_case("c1_single_numba", default_cfg(num_agents=1, neighbor_visible_num=0, neighbor_obs_type="none",
                                            use_downwash=False, use_numba=True), steps=1000, seed=12)   # config 1's length


@case
def c1_single_numba_f32ou():
    # the numba path with OUNoiseNumba's float32 theta / sigma (numba_utils.py:67-74) emulated by the jitclass stub
    run_case("c1_single_numba_f32ou", default_cfg(num_agents=1, neighbor_visible_num=0, neighbor_obs_type="none", use_downwash=False, use_numba=True,
                                                  numba_float32_ou=True), steps=1000, seed=12)


@case
def c2_n8_numba_f32ou():
    run_case("c2_n8_numba_f32ou", default_cfg(numba_float32_ou=True), steps=200, seed=21)


@case
def c2_n8_random():
    run_case("c2_n8_random", default_cfg(), steps=300, seed=21)


@case
def c2_n8_hover_svd():
    # gentle actions keep drones airborne past the SVD re-orthogonalisation (sub-step ~101) - G3
    run_case("c2_n8_hover_svd", default_cfg(use_downwash=False), steps=130, seed=22, action_fn=hover_actions)


@case
def c2_n8_events():
    forces = {5: force_collide, 30: force_only_drone0_new, 31: force_pair12, 32: force_only_drone0_new,
              60: force_walls, 90: force_slide}
    run_case("c2_n8_events", default_cfg(), steps=120, seed=23, action_fn=events_actions, forces=forces)


@case
def c2_n8_episode():
    # short episode (ep_time 1.6 s -> ep_len 160) so that done / auto-reset / episode stats are captured twice
    run_case("c2_n8_episode", default_cfg(ep_time=1.6), steps=340, seed=24)


@case
def c2_n8_k2_numpy():
    run_case("c2_n8_k2_numpy", default_cfg(neighbor_visible_num=2, use_numba=False), steps=150, seed=25)


@case
def c2_n8_kall():
    run_case("c2_n8_kall", default_cfg(neighbor_visible_num=-1, obs_repr="xyz_vxyz_R_omega_wall"), steps=60,
             seed=26)


@case
def c3_n8_obst():
    cfg = default_cfg(use_obstacles=True, neighbor_visible_num=2, obs_repr="xyz_vxyz_R_omega_floor",
                      quads_mode="o_static_same_goal",
                      rew_coeff=dict(pos=1.0, effort=0.05, spin=0.1, vel=0.0, crash=1.0, orient=1.0, yaw=0.0,
                                     quadcol_bin=5.0, quadcol_bin_smooth_max=4.0, quadcol_bin_obst=5.0))
    run_case("c3_n8_obst", cfg, steps=200, seed=31, forces={20: force_obstacles, 21: force_obstacles},
             action_fn=hover_actions)


@case
def c3_n8_obst_episode():
    cfg = default_cfg(use_obstacles=True, neighbor_visible_num=2, obs_repr="xyz_vxyz_R_omega_floor",
                      quads_mode="o_static_same_goal", ep_time=2.0,
                      rew_coeff=dict(pos=1.0, effort=0.05, spin=0.1, vel=0.0, crash=1.0, orient=1.0, yaw=0.0,
                                     quadcol_bin=5.0, quadcol_bin_smooth_max=4.0, quadcol_bin_obst=5.0))
    run_case("c3_n8_obst_episode", cfg, steps=420, seed=32, forces={170: force_obstacles})


@case
def c4_n32_svs():
    run_case("c4_n32_svs", default_cfg(num_agents=32, quads_mode="swarm_vs_swarm"), steps=120, seed=41)


@case
def c4_n6_svs_switch():
    # small swarm, long enough to cross the U(4,6) s goal swap (update_goals, swarm_vs_swarm.py:52-72)
    run_case("c4_n6_svs_switch", default_cfg(num_agents=6, neighbor_visible_num=3, quads_mode="swarm_vs_swarm"),
             steps=640, seed=42, action_fn=hover_actions)


@case
def c4_svs_resets():
    # many resets (short episodes) to sweep the formation types of swarm_vs_swarm
    run_case("c4_svs_resets", default_cfg(num_agents=12, neighbor_visible_num=6, quads_mode="swarm_vs_swarm",
                                          ep_time=0.05), steps=96, seed=43)


def _scen_case(name, steps, seed, **over):
    def fn():
        run_case(name, default_cfg(**over), steps=steps, seed=seed, action_fn=hover_actions, slim=True)
    fn.__name__ = name
    CASES[name] = fn


OBST = dict(use_obstacles=True, neighbor_visible_num=2, obs_repr="xyz_vxyz_R_omega_floor",
            rew_coeff=dict(pos=1.0, effort=0.05, spin=0.1, vel=0.0, crash=1.0, orient=1.0, yaw=0.0,
                           quadcol_bin=5.0, quadcol_bin_smooth_max=4.0, quadcol_bin_obst=5.0))
# remaining scenarios: short episodes so that reset + the per-step goal dynamics are both exercised
_scen_case("s_static_diff_goal", 110, 51, quads_mode="static_diff_goal", ep_time=0.5, num_agents=10)
_scen_case("s_dynamic_same_goal", 650, 52, quads_mode="dynamic_same_goal", ep_time=6.2, num_agents=3, neighbor_visible_num=2)
_scen_case("s_dynamic_diff_goal", 650, 53, quads_mode="dynamic_diff_goal", ep_time=6.2, num_agents=5, neighbor_visible_num=2)
_scen_case("s_dynamic_formations", 260, 54, quads_mode="dynamic_formations", ep_time=1.2, num_agents=9)
_scen_case("s_swap_goals", 650, 55, quads_mode="swap_goals", ep_time=6.2, num_agents=4, neighbor_visible_num=2)
_scen_case("s_ep_lissajous3D", 130, 56, quads_mode="ep_lissajous3D", ep_time=0.6, num_agents=3, neighbor_visible_num=2)
_scen_case("s_ep_rand_bezier", 650, 57, quads_mode="ep_rand_bezier", ep_time=6.2, num_agents=2, neighbor_visible_num=1)
_scen_case("s_o_random", 130, 58, quads_mode="o_random", ep_time=0.6, **OBST)
_scen_case("s_o_dynamic_same_goal", 650, 59, quads_mode="o_dynamic_same_goal", ep_time=6.2, **dict(OBST, num_agents=3))
_scen_case("s_o_swap_goals", 650, 60, quads_mode="o_swap_goals", ep_time=6.2, **dict(OBST, num_agents=4))
_scen_case("s_o_ep_rand_bezier", 700, 64, quads_mode="o_ep_rand_bezier", ep_time=6.5, **dict(OBST, num_agents=2, neighbor_visible_num=1))
_scen_case("s_run_away", 330, 65, quads_mode="run_away", ep_time=2.5, num_agents=5, neighbor_visible_num=2)
_scen_case("s_mix", 420, 61, quads_mode="mix", ep_time=0.2, num_agents=6, neighbor_visible_num=3)
_scen_case("s_mix_obst", 200, 62, quads_mode="mix", ep_time=0.2, **dict(OBST, num_agents=4))
_scen_case("s_mix_single", 200, 63, quads_mode="mix", ep_time=0.2, num_agents=1, neighbor_visible_num=0, neighbor_obs_type="none")
# size edges and configuration corners (the same configurations tests/test_hip_parity.py runs on the GPU), few steps each
_scen_case("e_n64_k20", 24, 70, quads_mode="dynamic_formations", ep_time=0.4, num_agents=64, neighbor_visible_num=20, use_downwash=False)
_scen_case("e_n33_k8_numpy_wall", 30, 71, num_agents=33, neighbor_visible_num=8, use_numba=False, obs_repr="xyz_vxyz_R_omega_wall")
_scen_case("e_n40_kall_svs", 10, 72, num_agents=40, neighbor_visible_num=-1, quads_mode="swarm_vs_swarm")
_scen_case("x_n8_blind", 60, 73, neighbor_visible_num=0, neighbor_obs_type="none", ep_time=0.5)
_scen_case("x_no_noise", 60, 74, sense_noise=None, thrust_noise_ratio=0.0)
_scen_case("x_dense_obst", 60, 75, quads_mode="o_random", ep_time=0.4, **dict(OBST, obst_density=0.8, obst_size=0.5))
_scen_case("x_small_room", 80, 76, quads_mode="dynamic_diff_goal", ep_time=0.6, room_dims=[6.0, 6.0, 4.0], obs_repr="xyz_vxyz_R_omega_wall")
_scen_case("x_ep_len2", 40, 77, quads_mode="static_diff_goal", ep_time=0.02, num_agents=5, neighbor_visible_num=2)
_scen_case("x_hitbox", 60, 78, collision_hitbox_radius=3.0, collision_falloff_radius=6.0,
           rew_coeff=dict(pos=0.5, effort=0.1, spin=0.2, vel=0.0, crash=2.0, orient=0.7, yaw=0.0, quadcol_bin=3.0,
                          quadcol_bin_smooth_max=7.0, quadcol_bin_obst=5.0))
_scen_case("x_svs_odd", 50, 79, quads_mode="swarm_vs_swarm", ep_time=0.3, num_agents=9)
_scen_case("e_n17_kall_obst", 50, 80, quads_mode="o_random", ep_time=0.4, **dict(OBST, num_agents=17, neighbor_visible_num=-1))
_scen_case("x_n40_obst", 45, 81, quads_mode="o_random", ep_time=0.4, **dict(OBST, num_agents=40, neighbor_visible_num=6))
_scen_case("e_n2_k1_swap", 60, 82, quads_mode="swap_goals", ep_time=0.5, num_agents=2, neighbor_visible_num=1)
_scen_case("e_n1_obst", 50, 83, quads_mode="o_static_same_goal", ep_time=0.3, **dict(OBST, num_agents=1, neighbor_visible_num=0, neighbor_obs_type="none"))


def force_between_obstacles(env):
    # Every drone halfway between two ADJACENT obstacles (1 m apart, radius 0.475: both within reach), a centimetre nearer to one of them -
    # alternately the lower- and the higher-numbered one: obstacles/utils.py:31-43 reports the LOWER index either way (index order + break).
    op = np.array(env.obstacles.pos_arr)[:, :2]
    pairs = [(a, b) for a in range(len(op)) for b in range(a + 1, len(op)) if abs(np.linalg.norm(op[a] - op[b]) - 1.0) < 1e-9]
    for i in range(len(env.envs)):
        a, b = pairs[(i * 7 + 3) % len(pairs)]
        mid, off = 0.5 * (op[a] + op[b]), (0.01 if i % 2 else -0.01) * (op[b] - op[a])
        set_dyn(env, i, pos=[mid[0] + off[0], mid[1] + off[1], 1.0 + 0.4 * i], vel=[0.1 * (i - 4), 0.05 * i, 0.0])


@case
def x_obst_first_hit():
    # 51 obstacles of 0.95 m: which obstacle a drone hit when two are within reach, on consecutive steps (new-vs-previous bookkeeping) and again later
    cfg = default_cfg(quads_mode="o_random", ep_time=0.5, **dict(OBST, obst_density=0.8, obst_size=0.95))
    run_case("x_obst_first_hit", cfg, steps=60, seed=84, action_fn=hover_actions,
             forces={3: force_between_obstacles, 4: force_between_obstacles, 9: force_between_obstacles, 30: force_between_obstacles})


if __name__ == "__main__":
    install_recorders()
    names = sys.argv[1:] or list(CASES)
    for nm in names:
        CASES[nm]()
