"""(The experience-replay wrapper is pinned in tests/test_replay_model_vs_reference.py + tests/test_replay_gpu.py.)
The host side of sf_env.BatchedQuadSwarm - shaping scheme pushed into env.rew_coeff, collision-coefficient annealing, `true_reward` and the
episode-end `episode_extra_stats` assembled from per-episode sums - against the reference's QuadsRewardShapingWrapper
(swarm_rl/env_wrappers/reward_shaping.py:19-123).  tests/golden/wrapper_reward_shaping.json holds what the reference wrapper produced
over the scripted env of tests/fake_env.py (oracle/ref_harness/capture_wrappers.py, build container); here BatchedQuadSwarm runs over
the same script through tests/fake_env.FakeVec, which keeps the sums the way the step kernel does.  CPU only."""
import json
import os

import pytest

from quad_swarm_rl_amd import sf_env
from tests.fake_env import FakeQuadEnv, FakeVec, drive_batched

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "wrapper_reward_shaping.json")


def close(a, b):
    if a is None or b is None:
        return a is b
    if isinstance(a, dict):
        return sorted(a) == sorted(b) and all(close(a[k], b[k]) for k in a)
    if isinstance(a, list):
        return len(a) == len(b) and all(close(x, y) for x, y in zip(a, b))
    if isinstance(a, bool) or isinstance(b, bool):
        return a == b
    return a == pytest.approx(b, rel=1e-12, abs=1e-12)


@pytest.mark.parametrize("case,seed", [("annealed", 1), ("plain", 2)])
def test_reward_shaping_wrapper_equals_reference(case, seed):
    want = json.load(open(GOLDEN))[case]
    annealing = [sf_env.AnnealSchedule("quadcol_bin", 5.0, 600000), sf_env.AnnealSchedule("quadcol_bin_smooth_max", 10.0, 300000)] if case == "annealed" else None
    scheme = dict(quad_rewards=dict(pos=1.0, effort=0.05, spin=0.1, vel=0.0, crash=1.0, orient=1.0, yaw=0.0,
                                    quadcol_bin=0.0 if annealing else 5.0, quadcol_bin_smooth_max=0.0 if annealing else 10.0, quadcol_bin_obst=5.0))
    env = FakeQuadEnv(seed=seed)
    vec = FakeVec(env)
    got = drive_batched(sf_env.BatchedQuadSwarm(1, reward_shaping_scheme=scheme, annealing=annealing, _vec=vec), vec, env, steps=30, seed=seed)
    assert len(got["steps"]) == len(want["steps"]) == 30
    for t, (g, w) in enumerate(zip(got["steps"], want["steps"])):
        for key in ("rewards", "dones", "true_reward", "rew_coeff"):
            assert close(g[key], w[key]), (t, key, g[key], w[key])
        # (the scripted env's own {"num_collisions": k} of agent 0 stays in place in the reference's dict: not part of what the wrapper adds)
        w_extra = [None if e is None else {k: v for k, v in e.items() if k != "num_collisions"} for e in w["extra"]]
        assert close(g["extra"], w_extra), (t, "episode_extra_stats", g["extra"], w_extra)
    assert close(got["coeff_seen_by_env"], want["coeff_seen_by_env"])
    # the script crosses four episode ends; the annealed coefficients grow and saturate at their final values
    ends = [s for s in want["steps"] if s["dones"][0]]
    assert len(ends) == 4 and all(e["true_reward"][0] is not None for e in ends)
    if case == "annealed":
        assert ends[0]["rew_coeff"]["quadcol_bin"] < ends[-1]["rew_coeff"]["quadcol_bin"] <= 5.0
        assert ends[-1]["rew_coeff"]["quadcol_bin_smooth_max"] == 10.0


def test_command_line_flags_equal_reference():
    """`--quads_*` surface of swarm_rl/env_wrappers/quadrotor_params.py (tests/golden/flags.json, oracle/ref_harness/capture_flags.py):
    every flag with the same default, choices and arity; this repo only ADDS the four flags of the device path."""
    import argparse
    ref = json.load(open(os.path.join(os.path.dirname(GOLDEN), "flags.json")))
    p = argparse.ArgumentParser()
    sf_env.add_quadrotors_env_args("quadrotor_multi", p)
    mine = {a.dest: a for a in p._actions if a.dest != "help"}
    assert sorted(set(mine) - set(ref["flags"])) == ["quads_backend", "quads_device", "quads_gather_obs", "quads_num_envs", "quads_num_gpus", "quads_obs_transport", "quads_obs_wire", "quads_precision", "quads_seed"]
    assert len(ref["flags"]) == 37
    for name, r in ref["flags"].items():
        a = mine[name]
        assert a.default == r["default"], name
        assert (list(a.choices) if a.choices else None) == r["choices"], name
        assert a.nargs == r["nargs"], name
    p2 = argparse.ArgumentParser()
    for k in ref["override_defaults"]:
        p2.add_argument("--" + k, default=None)
    sf_env.quadrotors_override_defaults("quadrotor_multi", p2)
    assert vars(p2.parse_args([])) == ref["override_defaults"]


@pytest.mark.parametrize("name", ["c2_n8_episode", "c3_n8_obst_episode", "c4_svs_resets"])
def test_episode_extra_stats_dict_equals_reference(name):
    """env.QuadrotorEnvMulti.episode_extra_stats (the host-side dict assembly of quadrotor_multi.py:637-718) over the episode
    snapshot - here produced by the oracle replaying a reference fixture - against infos[0]['episode_extra_stats'] of the
    reference itself: same key set, same values."""
    import numpy as np
    from oracle import oracle as orc
    from quad_swarm_rl_amd import env as qenv
    from tests import golden_util as gu
    g, cfgd = gu.load(name)
    cfg = gu.config_from_golden(cfgd)
    n = cfgd["num_agents"]
    o = orc.OracleEnv(cfg, tape=g["tape"])
    o.reset()
    ends = {d["step"]: d["stats"] for d in json.loads(str(g["ep_stats"]))}
    force = {int(t): k for k, t in enumerate(g["force_steps"])}
    checked = 0
    for t in range(g["actions"].shape[0]):
        if t in force:
            k = force[t]
            s, tick = o.get_state()
            s[:, 0:3] = g["force_pos"][k]; s[:, 3:6] = g["force_vel"][k]
            s[:, 6:15] = g["force_rot"][k].reshape(n, 9); s[:, 15:18] = g["force_omega"][k]
            o.set_state(s, tick)
        o.step(g["actions"][t])
        if t in ends:
            info = o.info()

            class Stepper:   # what the facade reads from the device: ep_stats [6, N] component-major, ep_counters [11, E]
                @staticmethod
                def to_host(what):
                    return np.array(info.ep_stats)[:n].T.copy() if what == "ep_stats" else np.array(info.ep_counters).reshape(-1, 1)

            fac = qenv.QuadrotorEnvMulti.__new__(qenv.QuadrotorEnvMulti)
            fac._vec = type("V", (), {"stepper": Stepper})()
            fac.num_agents, fac.use_obstacles = n, bool(cfgd["use_obstacles"])
            fac.scenario = type("S", (), {"name": staticmethod(lambda finished_episode=False: "Scenario_" + cfgd["quads_mode"])})()
            got = fac.episode_extra_stats()
            assert len(got) == n and sorted(got[0]) == sorted(ends[t]), sorted(set(got[0]) ^ set(ends[t]))
            for key, want in ends[t].items():
                assert got[0][key] == pytest.approx(want, rel=1e-9, abs=1e-12), key
            checked += 1
    assert checked >= 2
