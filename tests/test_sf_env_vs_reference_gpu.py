"""SURVEY.md 8f rank 1 on the GPU against the REFERENCE: sf_env.SingleQuadSwarm / BatchedQuadSwarm over the HIP stepper, compared with the
reference's QuadsRewardShapingWrapper (swarm_rl/env_wrappers/reward_shaping.py:52-123) over the reference's real QuadrotorEnvMulti.

tests/golden/wrapper_real_env_*.npz (oracle/ref_harness/capture_wrapper_real_env.py, build container) hold what that stack did: actions, the
training-step counter given to set_training_info, every random draw of the env on a sequential tape, observations / rewards / dones, the
per-step infos[i]['rewards'] terms, the env's reward coefficients after every step (shaping scheme pushed on the first step, collision
coefficients annealed at episode ends), and per agent at every episode end `true_reward` and the complete `episode_extra_stats` dict.
Here the env is built from the same FLAGS through make_quadrotor_env, the tape goes to the float64 noise-tape flavour of the kernels
(qs_set_noise_tape), and everything above must come back: floats to 1e-9 relative (sums over an episode), dones / key sets exact.
The per-episode sums, the action moments and the coefficients that the reference keeps in Python run on the device here (episode_sums,
qs_set_reward_coeffs): this is the test that compares them with the reference instead of with a host twin of themselves."""
import argparse
import json

import numpy as np
import pytest

from quad_swarm_rl_amd import sf_env
from tests import golden_util as gu

pytestmark = pytest.mark.gpu
FIXTURES = ["wrapper_real_env_c2_annealed", "wrapper_real_env_c3_obst", "wrapper_real_env_mix"]
REL = 1e-9


def flags_of(cfgd, fl, num_envs=1):
    """the command line a user of the reference would type for this fixture's env (quadrotor_params.py flag names)"""
    p = argparse.ArgumentParser()
    sf_env.add_quadrotors_env_args("quadrotor_multi", p)
    argv = ["--quads_num_agents", str(cfgd["num_agents"]), "--quads_episode_duration", str(cfgd["ep_time"]), "--quads_obs_repr", cfgd["obs_repr"],
            "--quads_neighbor_visible_num", str(cfgd["neighbor_visible_num"]), "--quads_neighbor_obs_type", cfgd["neighbor_obs_type"],
            "--quads_collision_hitbox_radius", str(cfgd["collision_hitbox_radius"]), "--quads_collision_falloff_radius", str(cfgd["collision_falloff_radius"]),
            "--quads_use_obstacles", str(cfgd["use_obstacles"]), "--quads_obst_density", str(cfgd["obst_density"]), "--quads_obst_size", str(cfgd["obst_size"]),
            "--quads_obst_spawn_area", *[str(x) for x in cfgd["obst_spawn_area"]], "--quads_use_downwash", str(cfgd["use_downwash"]),
            "--quads_use_numba", str(cfgd["use_numba"]), "--quads_mode", cfgd["quads_mode"], "--quads_room_dims", *[str(x) for x in cfgd["room_dims"]],
            "--quads_collision_reward", str(fl["collision_reward"]), "--quads_collision_smooth_max_penalty", str(fl["smooth_max"]),
            "--quads_obst_collision_reward", str(fl["obst_collision_reward"]), "--anneal_collision_steps", str(fl["anneal_steps"]),
            "--quads_precision", "f64", "--quads_num_envs", str(num_envs)]
    return p.parse_args(argv)


def close(got, want, what):
    assert got == pytest.approx(want, rel=REL, abs=1e-9), f"{what}: {got!r} vs the reference's {want!r}"


def check_end(info, end, i, where):
    close(info["true_reward"], end["true_reward"][i], f"{where} true_reward of agent {i}")
    got, want = info["episode_extra_stats"], end["extra"][i]
    assert sorted(got) == sorted(want), f"{where} agent {i}: key sets differ: {sorted(set(got) ^ set(want))}"
    for k, v in want.items():
        close(float(got[k]), v, f"{where} agent {i} episode_extra_stats[{k!r}]")


def apply_force(st, g, k, envs, n):
    for e in range(envs):
        s, tick = st.get_state(e)
        s[:, 0:3] = g["force_pos"][k]; s[:, 3:6] = g["force_vel"][k]
        s[:, 6:15] = g["force_rot"][k].reshape(n, 9); s[:, 15:18] = g["force_omega"][k]
        st.set_state(e, s, tick)


@pytest.mark.parametrize("name", FIXTURES)
def test_single_env_stack_equals_the_reference_wrapper_over_the_real_env(name, monkeypatch):
    """make_quadrotor_env without --quads_num_envs: the reference's list / numpy protocol (SingleQuadSwarm)"""
    monkeypatch.setenv("QS_SPEC", "off")   # a handle with a noise tape launches the library's tape kernels only
    g, cfgd = gu.load(name)
    fl, ends, coeff_after = json.loads(str(g["flags"])), json.loads(str(g["ends"])), json.loads(str(g["coeff_after"]))
    n = cfgd["num_agents"]
    env = sf_env.make_quadrotor_env("quadrotor_multi", flags_of(cfgd, fl))
    assert isinstance(env, sf_env.SingleQuadSwarm) and env.num_agents == n and env.is_multiagent
    st = env._vec.stepper
    st.set_noise_tape(g["tape"][None, :])
    obs, _ = env.reset()
    np.testing.assert_array_equal(st.tape_pos(), g["tape_pos"][0])
    np.testing.assert_allclose(obs, g["obs0"], rtol=0, atol=1e-9)
    force = {int(t): k for k, t in enumerate(g["force_steps"])}
    end_at = {e["step"]: e for e in ends}
    keys = [k for k in gu.qcfg.REW_INFO_KEYS if cfgd["use_obstacles"] or "obstacle" not in k]
    for t in range(g["actions"].shape[0]):
        if t in force:
            apply_force(st, g, force[t], 1, n)
        env.set_training_info({"approx_total_training_steps": int(g["approx_steps"][t])})
        obs, rew, term, trunc, infos = env.step([a for a in g["actions"][t]])
        np.testing.assert_array_equal(st.tape_pos(), g["tape_pos"][t + 1], err_msg=f"step {t}: tape position")
        np.testing.assert_allclose(obs, g["obs"][t], rtol=0, atol=1e-9, err_msg=f"obs step {t}")
        np.testing.assert_allclose(rew, g["rew"][t], rtol=0, atol=1e-9, err_msg=f"reward step {t}")
        np.testing.assert_array_equal(np.asarray(term, dtype=np.int8), g["done"][t], err_msg=f"done step {t}")
        assert not np.asarray(trunc).any() and len(infos) == n
        for i in range(n):   # infos[i]['rewards'] of every step (quadrotor_single.py:68-85, quadrotor_multi.py:533-540)
            assert sorted(infos[i]["rewards"]) == sorted(keys)
            for j, k in enumerate(gu.qcfg.REW_INFO_KEYS):
                if k in infos[i]["rewards"]:
                    close(infos[i]["rewards"][k], float(g["rew_info"][t][i, j]), f"step {t} agent {i} rewards[{k!r}]")
        for k, v in coeff_after[t].items():   # the coefficients the env holds after this step: scheme pushed, annealed at episode ends
            close(float(env.rew_coeff[k]), v, f"step {t} rew_coeff[{k!r}]")
        if t in end_at:
            for i in range(n):
                check_end(infos[i], end_at[t], i, f"step {t}")
        else:
            assert all("true_reward" not in d and "episode_extra_stats" not in d for d in infos), f"step {t}: episode-end keys on a step without an episode end"
    np.testing.assert_array_equal(st.tape_pos(), len(g["tape"]))
    assert len(ends) >= 2
    env.close()


@pytest.mark.parametrize("name", FIXTURES)
def test_batched_env_equals_the_reference_wrapper_over_the_real_env(name, monkeypatch):
    """--quads_num_envs 3: BatchedQuadSwarm (device tensors, E * N agents), every environment on the same tape - each must reproduce the record"""
    import torch
    monkeypatch.setenv("QS_SPEC", "off")
    E = 3
    g, cfgd = gu.load(name)
    fl, ends, coeff_after = json.loads(str(g["flags"])), json.loads(str(g["ends"])), json.loads(str(g["coeff_after"]))
    n = cfgd["num_agents"]
    env = sf_env.make_quadrotor_env("quadrotor_multi", flags_of(cfgd, fl, num_envs=E))
    assert isinstance(env, sf_env.BatchedQuadSwarm) and env.num_agents == E * n
    st = env.vec.stepper
    st.set_noise_tape(np.tile(g["tape"], (E, 1)))
    obs, _ = env.reset()
    np.testing.assert_allclose(obs["obs"].double().cpu().numpy().reshape(E, n, -1), np.tile(g["obs0"], (E, 1, 1)), rtol=0, atol=1e-9)
    force = {int(t): k for k, t in enumerate(g["force_steps"])}
    end_at = {e["step"]: e for e in ends}
    dev = obs["obs"].device
    for t in range(g["actions"].shape[0]):
        if t in force:
            apply_force(st, g, force[t], E, n)
        env.set_training_info({"approx_total_training_steps": int(g["approx_steps"][t])})
        a = torch.as_tensor(np.tile(g["actions"][t], (E, 1)), device=dev, dtype=torch.float64)
        obs, rew, term, trunc, infos = env.step(a)
        torch.cuda.synchronize()
        np.testing.assert_allclose(obs["obs"].double().cpu().numpy().reshape(E, n, -1), np.tile(g["obs"][t], (E, 1, 1)), rtol=0, atol=1e-9, err_msg=f"obs step {t}")
        np.testing.assert_allclose(rew.double().cpu().numpy().reshape(E, n), np.tile(g["rew"][t], (E, 1)), rtol=0, atol=1e-9, err_msg=f"reward step {t}")
        np.testing.assert_array_equal(term.cpu().numpy().reshape(E, n).astype(np.int8), np.tile(g["done"][t], (E, 1)))
        for k, v in coeff_after[t].items():
            close(float(env.rew_coeff[k]), v, f"step {t} rew_coeff[{k!r}]")
        if t in end_at:
            assert len(infos) == E * n
            for e in range(E):
                for i in range(n):
                    check_end(infos[e * n + i], end_at[t], i, f"step {t} env {e}")
        else:
            assert len(infos) == 0
    np.testing.assert_array_equal(st.tape_pos(), len(g["tape"]))
    env.close()
