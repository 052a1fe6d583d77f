"""Reference fixtures STRAIGHT through the HIP stepper (no oracle in between).

tests/golden/*.npz were captured from the reference's QuadrotorEnvMulti itself (oracle/ref_harness/capture.py): actions, every
random draw it made on a sequential tape, and its outputs.  Here the tape is handed to the HIP stepper (qs_set_noise_tape: the
float64 noise-tape flavour of the kernels pops the reference's draws instead of running Philox) and the reference's own outputs
are the expectation - the assertions of tests/test_oracle_vs_reference.py, with the HIP stepper in the oracle's place: floats to
1e-9, every flag / mask / counter / tape position exact.  Same method as the reference's numba-vs-numpy test
(gym_art/quadrotor_multi/tests/test_numba_opt.py:59-119): two implementations under identical injected noise.
Each fixture is replayed in 3 environments side by side (same tape): the per-environment cursors must not interact.
"""
import json

import numpy as np
import pytest

from quad_swarm_rl_amd import config as qcfg
from tests import golden_util as gu
from tests.test_oracle_vs_reference import CASES, CASES_FIRST_HIT, EDGE_CASES, SCEN_CASES

pytestmark = pytest.mark.gpu
TOL = 1e-9
E = 3


def soa(a, n):
    """[C, E*N] -> [E, N, C]"""
    return np.ascontiguousarray(a.reshape(a.shape[0], E, n).transpose(1, 2, 0))


def mask_of(bits):
    m = 0
    for i, b in enumerate(bits):
        if b:
            m |= 1 << i
    return m


def check_episode_stats(st, eps, cnt, n, use_obstacles):
    """eps: [N, QS_EPS_COUNT] of the finished episode, cnt: its counters (tests/test_oracle_vs_reference.py:check_episode_stats)"""
    assert cnt[0] == st["num_collisions"] and cnt[3] == st["num_collisions_with_room"]
    assert cnt[4] == st["num_collisions_with_floor"] and cnt[5] == st["num_collisions_with_wall"]
    assert cnt[6] == st["num_collisions_with_ceiling"] and cnt[1] == st["num_collisions_after_settle"]
    assert cnt[2] == st["num_collisions_final_5_s"]
    np.testing.assert_allclose(eps[0, 0:3], [st["distance_to_goal_1s"], st["distance_to_goal_3s"], st["distance_to_goal_5s"]], rtol=1e-9)
    ok = np.logical_and(eps[:, 4], eps[:, 5])
    np.testing.assert_allclose(np.sum(np.logical_and(ok, eps[:, 3])) / n, st["metric/agent_success_rate"])
    np.testing.assert_allclose(np.sum(np.logical_and(ok, 1 - eps[:, 3])) / n, st["metric/agent_deadlock_rate"])
    np.testing.assert_allclose(1.0 - np.sum(ok) / n, st["metric/agent_col_rate"])
    np.testing.assert_allclose(1.0 - np.sum(eps[:, 4]) / n, st["metric/agent_neighbor_col_rate"])
    np.testing.assert_allclose(1.0 - np.sum(eps[:, 5]) / n, st["metric/agent_obst_col_rate"])
    if use_obstacles:
        assert cnt[7] == st["num_collisions_obst_quad"] and cnt[8] == st["num_collisions_obst_quad_after_settle"]
        assert cnt[9] == st["num_collisions_obst_quad_3_5"] and cnt[10] == st["num_collisions_obst_quad_5"]


@pytest.mark.parametrize("name", CASES + EDGE_CASES + SCEN_CASES + CASES_FIRST_HIT)
def test_reference_fixture_through_hip(name, monkeypatch):
    from quad_swarm_rl_amd import native
    # a handle with a noise tape only ever launches the tape kernels of the library (launch_reset / launch_step, quadswarm_hip.hip): the
    # config-specialised object qs_create would compile for each of the 42 configurations on the GPU box would never run
    monkeypatch.setenv("QS_SPEC", "off")
    g, cfgd = gu.load(name)
    cfg = gu.config_from_golden(cfgd, num_envs=E, precision="f64")
    n = cfgd["num_agents"]
    st = native.Stepper(cfg, device=0)
    try:
        st.set_noise_tape(np.tile(g["tape"], (E, 1)))
    except native.QsError as exc:
        st.close()
        if "not replayable yet" in str(exc):
            pytest.skip(str(exc))
        raise
    D = st.obs_dim
    st.reset()
    np.testing.assert_array_equal(st.tape_pos(), g["tape_pos"][0], err_msg="reset consumed a different number of draws than the reference")
    obs0 = st.to_host("obs").reshape(E, n, D)
    for e in range(E):
        np.testing.assert_allclose(obs0[e], g["obs0"], rtol=0, atol=TOL)
    slim = "s0_vel" not in g.files       # scenario fixtures keep obs/rew/done + pos/goal/tick only
    if not slim:
        for e in range(E):
            s, _ = st.get_state(e)
            np.testing.assert_allclose(s[:, :18], gu.state_from_golden(g, "s0_")[:, :18], rtol=0, atol=TOL)

    force = {int(t): k for k, t in enumerate(g["force_steps"])}
    ep_stats = {d["step"]: d["stats"] for d in json.loads(str(g["ep_stats"]))}
    checked_eps, worst = 0, 0.0
    steps = g["actions"].shape[0]
    for t in range(steps):
        if t in force:
            k = force[t]
            for e in range(E):
                s, tick = st.get_state(e)
                s[:, 0:3] = g["force_pos"][k]; s[:, 3:6] = g["force_vel"][k]
                s[:, 6:15] = g["force_rot"][k].reshape(n, 9); s[:, 15:18] = g["force_omega"][k]
                st.set_state(e, s, tick)
        st.from_host("actions", np.tile(g["actions"][t], (E, 1, 1)).reshape(-1, 4))
        st.step()
        st.sync()
        st.check_errors()
        np.testing.assert_array_equal(st.tape_pos(), g["tape_pos"][t + 1], err_msg=f"step {t}: tape position")
        obs, rew, done = st.to_host("obs").reshape(E, n, D), st.to_host("reward").reshape(E, n), st.to_host("done").reshape(E, n)
        ri = soa(st.to_host("rew_info"), n)
        flags, tick = st.to_host("flags").reshape(E, n), st.to_host("tick")
        cp, npm = st.to_host("col_pair_mask").reshape(E, n), st.to_host("new_pair_mask").reshape(E, n)
        uq, on, rn, ohi = st.to_host("unique_col_mask"), st.to_host("obst_new_mask"), st.to_host("room_new_mask"), st.to_host("obst_hit_idx").reshape(E, n)
        cnt, epc, eps = st.to_host("counters"), st.to_host("ep_counters"), soa(st.to_host("ep_stats"), n)
        for e in range(E):
            np.testing.assert_array_equal(done[e], g["done"][t], err_msg=f"done step {t}")
            for nm, a, b in (("obs", obs[e], g["obs"][t]), ("rew", rew[e], g["rew"][t])) + (() if slim else (("rew_info", ri[e], g["rew_info"][t]),)):
                err = np.abs(a - b).max()
                worst = max(worst, err)
                assert err <= TOL, f"{nm} step {t} env {e}: max abs err {err}"
            s, tk = st.get_state(e)
            if slim:
                np.testing.assert_allclose(s[:, 0:3], g["s_pos"][t], atol=TOL, err_msg=f"pos step {t}")
                np.testing.assert_allclose(s[:, 32:35], g["s_goal"][t], atol=TOL, err_msg=f"goal step {t}")
                np.testing.assert_array_equal(s[:, 30], g["s_on_floor"][t], err_msg=f"on_floor step {t}")
            else:
                ref = gu.state_from_golden(g, "s_", t)
                err = np.abs(s[:, :30] - ref[:, :30]).max()
                assert err <= TOL, f"state step {t}: {err}"
                np.testing.assert_array_equal(s[:, 30], ref[:, 30], err_msg=f"on_floor step {t}")
                np.testing.assert_allclose(s[:, 32:35], ref[:, 32:35], atol=TOL, err_msg=f"goal step {t}")
            assert tk == g["s_tick"][t][0] and tick[e] == tk
            if not done[e].any() and not slim:
                assert mask_of(flags[e] & 2) == mask_of(g["s_crashed_floor"][t])
                assert mask_of(flags[e] & 4) == mask_of(g["s_crashed_wall"][t])
                assert mask_of(flags[e] & 8) == mask_of(g["s_crashed_ceiling"][t])
                assert int(uq[e]) == int(g["unique_col"][t]), f"unique collisions step {t}"
                np.testing.assert_array_equal(cp[e], g["curr_pairs"][t])
                np.testing.assert_array_equal(npm[e], g["new_pairs"][t])
                assert int(on[e]) == int(g["obst_new"][t]), f"obstacle collisions step {t}"
                if cfgd["use_obstacles"]:
                    assert mask_of(ohi[e] >= 0) == int(g["obst_hit"][t])
                assert int(rn[e]) == int(g["room_new"][t]), f"room collisions step {t}"
            if cfgd["use_obstacles"] and not done[e].any():   # WHICH obstacle (obstacles/utils.py:31-43: index order, first within reach, break)
                np.testing.assert_array_equal(ohi[e], g["obst_hit_idx"][t], err_msg=f"first-hit obstacle index step {t}")
            np.testing.assert_array_equal(cnt[:, e], g["counters"][t], err_msg=f"counters step {t}")
            if done[e].any():  # episode stats (quadrotor_multi.py:626-718)
                check_episode_stats(ep_stats[t], eps[e], epc[:, e], n, cfgd["use_obstacles"])
                name_now = qcfg.SCENARIO_CLASS_NAMES[int(st.to_host("ep_scenario")[e])][9:]
                assert f"{name_now}/num_collisions" in ep_stats[t], (name_now, sorted(ep_stats[t])[:4])
                checked_eps += 1
    np.testing.assert_array_equal(st.tape_pos(), len(g["tape"]))
    assert checked_eps == E * len(ep_stats)
    st.close()
    print(f"{name}: worst abs err {worst:.3e}")
