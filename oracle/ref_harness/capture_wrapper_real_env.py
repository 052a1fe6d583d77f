#!/usr/bin/env python
"""Build-container only (needs /root/reference).  The reference's QuadsRewardShapingWrapper (swarm_rl/env_wrappers/reward_shaping.py:19-123)
over the REAL QuadrotorEnvMulti, built the way make_quadrotor_env_multi builds it (swarm_rl/env_wrappers/quad_utils.py:20-110: shaping
scheme from the flags, three annealing schedules when --anneal_collision_steps > 0), with every random draw of the env on the sequential
noise tape of capture.py.  Recorded per step: actions, the training-step counter handed to set_training_info, observations, rewards, dones,
the per-step infos[i]['rewards'] terms, the reward coefficients the env holds after the step; at every episode end, per agent: true_reward
and the complete episode_extra_stats dict (the env's own statistics, the cumulative rew_* / rewraw_* sums, z_action*_mean / _std, the
per-scenario reward keys, z_anneal_*, z_approx_total_training_steps).

tests/test_sf_env_vs_reference_gpu.py hands the tape to the HIP stepper underneath sf_env.SingleQuadSwarm / BatchedQuadSwarm and expects
exactly this record.  Fixtures: tests/golden/wrapper_real_env_*.npz (data only).

Usage: python oracle/ref_harness/capture_wrapper_real_env.py
"""
import json
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)


class TrainingInfoInterface:
    def __init__(self):
        self.training_info = {}

    def set_training_info(self, training_info):
        self.training_info = training_info


class RewardShapingInterface:
    def __init__(self):
        pass


for name in ("sample_factory", "sample_factory.envs"):
    sys.modules[name] = types.ModuleType(name)
m = types.ModuleType("sample_factory.envs.env_utils")
m.TrainingInfoInterface, m.RewardShapingInterface = TrainingInfoInterface, RewardShapingInterface
sys.modules["sample_factory.envs.env_utils"] = m

import capture as cap                                                      # noqa: E402  (installs the stub path + /root/reference)
import gymnasium as gym                                                    # noqa: E402  (the stub)
from swarm_rl.env_wrappers import reward_shaping as ref                    # noqa: E402

if not hasattr(gym.Wrapper, "unwrapped"):
    gym.Wrapper.unwrapped = property(lambda self: self.env.unwrapped)


class Anneal:   # the namedtuple of quad_utils.py:78-83
    def __init__(self, coeff_name, final_value, anneal_env_steps):
        self.coeff_name, self.final_value, self.anneal_env_steps = coeff_name, final_value, anneal_env_steps


def shaping_from_flags(collision_reward, smooth_max, obst_collision_reward, anneal_steps):
    """quad_utils.py:72-107"""
    scheme = dict(quad_rewards=dict(pos=1.0, effort=0.05, spin=0.1, vel=0.0, crash=1.0, orient=1.0, yaw=0.0,
                                    quadcol_bin=collision_reward, quadcol_bin_smooth_max=smooth_max, quadcol_bin_obst=obst_collision_reward))
    annealing = None
    if anneal_steps > 0:
        for k in ("quadcol_bin", "quadcol_bin_smooth_max", "quadcol_bin_obst"):
            scheme["quad_rewards"][k] = 0.0
        annealing = [Anneal("quadcol_bin", collision_reward, anneal_steps), Anneal("quadcol_bin_smooth_max", smooth_max, anneal_steps),
                     Anneal("quadcol_bin_obst", obst_collision_reward, anneal_steps)]
    return scheme, annealing


def run(name, cfg, steps, seed, flags, train_steps_per_step, forces=None, action_fn=None):
    cap.TAPE = cap.Tape()
    np.random.seed(seed)
    arng = np.random.RandomState(seed + 7919)
    # the env starts from DEFAULT_QUAD_REWARD_SHAPING's coefficients (quad_utils.py:28) - the wrapper pushes the scheme on its first step
    cfg = dict(cfg, rew_coeff=dict(ref.DEFAULT_QUAD_REWARD_SHAPING["quad_rewards"]))
    # --quads_use_numba=True as real numba runs it: OUNoiseNumba's theta / sigma / mu are float32 jitclass members (numba_utils.py:67-74; the
    # stub of this container emulates that on request) - what sf_env builds from the same flag (config.make_config: numba_float32_ou follows use_numba)
    cfg["numba_float32_ou"] = bool(cfg["use_numba"])
    env = cap.make_env(cfg)
    scheme, annealing = shaping_from_flags(**flags)
    w = ref.QuadsRewardShapingWrapper(env, reward_shaping_scheme=scheme, annealing=annealing, with_pbt=False)
    n = cfg["num_agents"]
    obs0 = np.asarray(w.reset(), dtype=np.float64)
    tape_pos = [len(cap.TAPE)]
    rec = {k: [] for k in ("actions", "approx_steps", "obs", "rew", "done", "rew_info")}
    coeff_after, ends = [], []
    force_steps, force_state = [], {k: [] for k in ("pos", "vel", "rot", "omega")}
    obst_pos = [np.array(env.obstacles.pos_arr, dtype=np.float64)] if cfg["use_obstacles"] else []
    for t in range(steps):
        if forces and t in forces:
            forces[t](env)
            force_steps.append(t)
            for k in force_state:
                force_state[k].append(np.array([getattr(e.dynamics, k) for e in env.envs], dtype=np.float64))
        act = arng.uniform(-1.0, 1.0, size=(n, 4)) if action_fn is None else np.asarray(action_fn(t, env, arng), dtype=np.float64)
        approx = int(train_steps_per_step * t)
        w.set_training_info({"approx_total_training_steps": approx})
        obs, rew, done, infos = w.step([a for a in act])
        tape_pos.append(len(cap.TAPE))
        rec["actions"].append(act); rec["approx_steps"].append(approx)
        rec["obs"].append(np.asarray(obs, dtype=np.float64)); rec["rew"].append(np.asarray(rew, dtype=np.float64))
        rec["done"].append(np.asarray(done, dtype=np.int8))
        rec["rew_info"].append(np.array([[infos[i]["rewards"].get(k, 0.0) for k in cap.REW_KEYS] for i in range(n)]))
        coeff_after.append({k: float(v) for k, v in sorted(env.rew_coeff.items())})
        if any(done):
            assert all(done) and all("true_reward" in i and "episode_extra_stats" in i for i in infos)
            ends.append(dict(step=t, true_reward=[float(i["true_reward"]) for i in infos],
                             extra=[{k: float(v) for k, v in i["episode_extra_stats"].items()} for i in infos],
                             key_order=[list(i["episode_extra_stats"]) for i in infos][0]))
            if cfg["use_obstacles"]:
                obst_pos.append(np.array(env.obstacles.pos_arr, dtype=np.float64))
    out = dict(cfg=np.array(json.dumps(cfg)), flags=np.array(json.dumps(flags)), ends=np.array(json.dumps(ends)), coeff_after=np.array(json.dumps(coeff_after)),
               tape=np.array(cap.TAPE.vals, dtype=np.float64), tape_pos=np.array(tape_pos, dtype=np.int64), obs0=obs0,
               force_steps=np.array(force_steps, dtype=np.int64))
    for k in force_state:
        out["force_" + k] = np.array(force_state[k], dtype=np.float64).reshape((len(force_steps), n) + {"pos": (3,), "vel": (3,), "rot": (3, 3), "omega": (3,)}[k])
    for k, v in rec.items():
        out[k] = np.array(v)
    if obst_pos:
        out["obst_pos"] = np.array(obst_pos)
    path = os.path.join(cap.GOLDEN_DIR, name + ".npz")
    np.savez_compressed(path, **out)
    print(f"{name}: {steps} steps, {len(ends)} episode ends, tape {len(cap.TAPE)} draws -> {path} ({os.path.getsize(path) / 1024:.0f} KiB)")
    for e in ends:
        print("   end at step", e["step"], "true_reward[0] %.6f" % e["true_reward"][0], {k: round(v, 4) for k, v in e["extra"][0].items() if k.startswith("z_anneal")})


def collide_then_walls(t, env, arng):
    return cap.events_actions(t, env, arng)


if __name__ == "__main__":
    cap.install_recorders()
    # BASELINE config 5's env (train_local.sh: 8 drones, 6 neighbours, downwash, numba, collision reward 5 / smooth max 10 annealed) on short
    # episodes: three episode ends, collisions while the coefficients are still zero, half annealed and saturated
    run("wrapper_real_env_c2_annealed", cap.default_cfg(ep_time=0.7), steps=220, seed=91,
        flags=dict(collision_reward=5.0, smooth_max=10.0, obst_collision_reward=0.0, anneal_steps=90000.0), train_steps_per_step=640,
        forces={20: cap.force_collide, 95: cap.force_collide, 96: cap.force_only_drone0_new, 150: cap.force_collide, 180: cap.force_walls})
    # the obstacle flavour (train_local_obst.sh's shape: 2 neighbours, floor observation, obstacle collision reward), no annealing
    run("wrapper_real_env_c3_obst", cap.default_cfg(ep_time=0.6, use_obstacles=True, neighbor_visible_num=2, obs_repr="xyz_vxyz_R_omega_floor",
                                                    quads_mode="o_static_same_goal"), steps=130, seed=92,
        flags=dict(collision_reward=5.0, smooth_max=4.0, obst_collision_reward=5.0, anneal_steps=0.0), train_steps_per_step=1000,
        forces={15: cap.force_obstacles, 16: cap.force_obstacles, 80: cap.force_obstacles}, action_fn=cap.hover_actions)
    # --quads_mode mix: the per-scenario keys are named after the scenario of the episode that STARTS (reward_shaping.py:95-98 runs after the auto-reset)
    run("wrapper_real_env_mix", cap.default_cfg(ep_time=0.2, num_agents=6, neighbor_visible_num=3, quads_mode="mix"), steps=130, seed=93,
        flags=dict(collision_reward=5.0, smooth_max=10.0, obst_collision_reward=0.0, anneal_steps=30000.0), train_steps_per_step=500)
