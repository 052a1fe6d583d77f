"""The reference-shaped python surface on top of the HIP stepper (needs the GPU)."""
import argparse

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def parse(argv):
    from quad_swarm_rl_amd import sf_env
    p = argparse.ArgumentParser()
    sf_env.add_quadrotors_env_args("quadrotor_multi", p)
    p.add_argument("--with_pbt", default=False, type=sf_env.str2bool)
    return p.parse_args(argv)


def test_make_quadrotor_env_protocol():
    """Mirrors swarm_rl/env_wrappers/tests/test_quads.py:15-31: construct through the factory, random actions, types."""
    from quad_swarm_rl_amd import sf_env
    cfg = parse(["--quads_num_agents=8", "--quads_neighbor_visible_num=6", "--quads_neighbor_obs_type=pos_vel", "--quads_use_numba=True",
                 "--quads_collision_reward=5.0", "--quads_collision_falloff_radius=4.0", "--quads_use_downwash=True",
                 "--quads_episode_duration=0.3", "--anneal_collision_steps=1000"])
    env = sf_env.make_quadrotor_env("quadrotor_multi", cfg=cfg)
    assert env.num_agents == 8 and env.is_multiagent
    assert env.observation_space.shape == (54,) and env.action_space.shape == (4,)
    obs, info = env.reset(seed=123)
    assert obs.shape == (8, 54) and info == {}
    saw_done = False
    for t in range(40):
        actions = [env.action_space.sample() for _ in range(8)]
        obs, rew, term, trunc, infos = env.step(actions)
        assert obs.shape == (8, 54) and len(rew) == 8 and term.shape == (8,) and not trunc.any() and len(infos) == 8
        assert set(infos[0]["rewards"]) >= {"rew_main", "rew_pos", "rew_action", "rew_crash", "rew_orient", "rew_spin", "rewraw_main",
                                            "rew_quadcol", "rew_proximity", "rewraw_quadcol"}
        if term.any():
            assert term.all()                         # all agents finish together (quadrotor_multi.py:720-722)
            st = infos[0]["episode_extra_stats"]
            for k in ("num_collisions", "distance_to_goal_1s", "metric/agent_success_rate", "static_same_goal/agent_col_rate",
                      "z_anneal_quadcol_bin", "rewraw_main"):
                assert k in st, k
            assert "true_reward" in infos[0]
            saw_done = True
    assert saw_done
    with pytest.raises(NotImplementedError):
        sf_env.make_quadrotor_env("no_such_env", cfg=cfg)
    env.close()


def test_vec_env_device_tensors_and_coeff_push():
    import torch
    from quad_swarm_rl_amd.env import QuadSwarmVecEnv
    env = QuadSwarmVecEnv(16, num_agents=8, neighbor_visible_num=6, neighbor_obs_type="pos_vel", use_numba=True,
                          collision_falloff_radius=4.0, seed=3)
    obs = env.reset()
    assert obs.is_cuda and tuple(obs.shape) == (128, 54) and env.num_agents == 128
    a = torch.rand((128, 4), device="cuda") * 2 - 1
    obs, rew, done, _ = env.step(a)
    torch.cuda.synchronize()
    assert torch.isfinite(obs).all() and torch.isfinite(rew).all() and done.sum().item() == 0
    r1 = rew.clone()
    # the SF reward-shaping wrapper mutates env.rew_coeff: the next step must see the new coefficients
    env2 = QuadSwarmVecEnv(16, num_agents=8, neighbor_visible_num=6, neighbor_obs_type="pos_vel", use_numba=True,
                           collision_falloff_radius=4.0, seed=3)
    env2.reset()
    env2.rew_coeff["pos"] = 3.0
    _, rew2, _, _ = env2.step(a)
    torch.cuda.synchronize()
    assert (rew2 < r1 - 1e-4).all()
    env.close(); env2.close()


def test_nan_reward_raises_value_error():
    """quadrotor_single.py:87-90: ValueError('QuadEnv: reward is Nan')"""
    from quad_swarm_rl_amd import config as qcfg, native
    st = native.Stepper(qcfg.make_config(num_envs=2, num_agents=2))
    st.reset()
    a = np.zeros((4, 4), dtype=np.float32)
    a[1, 2] = np.nan
    st.from_host("actions", a)
    st.step()
    st.sync()
    with pytest.raises(ValueError, match="reward is Nan"):
        st.check_errors()
    st.close()


def test_step_many_rollout_equals_stepwise():
    from quad_swarm_rl_amd import config as qcfg, native
    import torch
    kw = dict(num_envs=32, num_agents=8, neighbor_visible_num=6, neighbor_obs_type="pos_vel", use_numba=True, use_downwash=True,
              collision_falloff_radius=4.0, seed=21, ep_time=0.2)
    acts = (torch.rand((16, 256, 4), device="cuda") * 2 - 1).contiguous()
    outs = []
    for graph in (False, True):
        st = native.Stepper(qcfg.make_config(**kw))
        st.reset()
        for rep in range(3):      # 48 steps with ep_len 20: auto-resets inside the captured graph
            if graph:
                st.step_many(acts.data_ptr(), 16)
            else:
                for t in range(16):
                    st.step(acts.data_ptr() + t * 256 * 16)
        st.sync()
        outs.append((st.to_host("obs").copy(), st.to_host("reward").copy(), st.to_host("tick").copy()))
        st.close()
    # the two kernels are separate instantiations: identical algorithm, fp32 results equal up to instruction selection
    np.testing.assert_allclose(outs[0][0], outs[1][0], rtol=0, atol=5e-5)
    np.testing.assert_allclose(outs[0][1], outs[1][1], rtol=0, atol=5e-6)
    np.testing.assert_array_equal(outs[0][2], outs[1][2])


def test_mix_scenario_names_follow_the_episode():
    """Under quads_mode=mix the per-scenario stat keys carry the finished episode's sub-scenario (mix.py:67-71)."""
    from quad_swarm_rl_amd import config as qcfg
    from quad_swarm_rl_amd import sf_env
    cfg = parse(["--quads_num_agents=4", "--quads_neighbor_visible_num=2", "--quads_neighbor_obs_type=pos_vel", "--quads_use_numba=True",
                 "--quads_mode=mix", "--quads_episode_duration=0.1", "--quads_seed=5"])
    env = sf_env.make_quadrotor_env("quadrotor_multi", cfg=cfg)
    env.reset()
    quad = env.unwrapped
    allowed = {"Scenario_" + n for n in qcfg.SCENARIOS if not n.startswith("o_") and n != "mix"}
    seen = set()
    for t in range(120):
        before = quad.scenario.name()
        assert before in allowed
        _, _, term, _, infos = env.step([env.action_space.sample() for _ in range(4)])
        if term.any():
            assert f"{before[9:]}/agent_col_rate" in infos[0]["episode_extra_stats"]
            seen.add(before)
    assert len(seen) >= 3, seen
    env.close()


def test_specialised_and_generic_kernels_agree(monkeypatch):
    """QS_SPEC=jit (default) runs the config-specialised code object, QS_SPEC=off the generic kernels, QS_TEAM picks the
    1-wave / 4-wave flavour: all four combinations give the same f64 trajectory (to reassociation-level rounding)."""
    from quad_swarm_rl_amd import config as qcfg, native
    kw = dict(num_agents=8, neighbor_visible_num=6, neighbor_obs_type="pos_vel", use_downwash=True, use_numba=True,
              collision_falloff_radius=4.0, rew_coeff=dict(quadcol_bin=5.0, quadcol_bin_smooth_max=10.0), ep_time=0.2)
    rng = np.random.RandomState(5)
    acts = rng.uniform(-1, 1, size=(30, 6 * 8, 4))
    runs = {}
    for spec in ("jit", "off"):
        for team in ("1", "0"):
            monkeypatch.setenv("QS_SPEC", spec)
            monkeypatch.setenv("QS_TEAM", team)
            st = native.Stepper(qcfg.make_config(num_envs=6, seed=3, precision="f64", **kw))
            assert st.specialized == (spec == "jit") and st.team == (team == "1")
            assert st.kernel_name == ("qs_spec_step" if spec == "jit" else ("qs_step_team<double>" if team == "1" else "qs_step_kernel<double>"))
            st.reset()
            out = [st.to_host("obs").copy()]
            for a in acts:
                st.from_host("actions", a)
                st.step()
                out += [st.to_host("obs").copy(), st.to_host("reward").copy(), st.to_host("done").copy()]
            st.check_errors()
            st.close()
            runs[(spec, team)] = out
    ref = runs[("off", "0")]
    for key, out in runs.items():
        for a, b in zip(ref, out):
            np.testing.assert_allclose(a, b, rtol=0, atol=1e-9, err_msg=str(key))


def test_episode_sums_on_device_match_host_accumulation():
    """episode_sums=1: the kernel's per-episode sums of the 17 reward terms / action moments equal what a host loop
    accumulates from the per-step rew_info (reward_shaping.py:78-83), including the step on which the episode ends."""
    from quad_swarm_rl_amd import config as qcfg, native
    kw = dict(num_agents=4, neighbor_visible_num=2, neighbor_obs_type="pos_vel", use_numba=True, use_downwash=True,
              collision_falloff_radius=4.0, rew_coeff=dict(quadcol_bin=5.0, quadcol_bin_smooth_max=10.0), ep_time=0.25)
    for precision, tol in (("f64", 1e-10), ("f32", 2e-5)):
        st = native.Stepper(qcfg.make_config(num_envs=5, seed=21, precision=precision, episode_sums=True, write_rew_info=True, **kw))
        st.reset()
        rng = np.random.RandomState(2)
        acc = np.zeros((25, st.T))
        episodes = 0
        for t in range(60):
            a = rng.uniform(-1.2, 1.2, size=(st.T, 4)).astype(st.np_real)
            st.from_host("actions", a)
            st.step()
            st.sync()
            acc[:17] += st.to_host("rew_info")
            acc[17:21] += a.T.astype(np.float64)
            acc[21:25] += (a.T.astype(np.float64)) ** 2
            if st.to_host("done").any():
                assert st.to_host("done").all()
                np.testing.assert_allclose(st.to_host("ep_sums"), acc, rtol=0, atol=tol * (1 + np.abs(acc).max()))
                np.testing.assert_array_equal(st.to_host("run_sums"), 0)
                acc[:] = 0
                episodes += 1
        assert episodes == 2
        st.close()


def test_batched_env_matches_the_single_env_wrapper_stack():
    """BatchedQuadSwarm (E envs, device tensors, on-device sums) and the one-environment stack of make_quadrotor_env (SingleQuadSwarm:
    lists / numpy in and out, per-step infos[i]['rewards']) report the same for the same env, seed and actions - and what they report
    at an episode end is what accumulating the per-step infos[i]['rewards'] on the host gives (reward_shaping.py:78-94: cumulative
    rew_* terms, true_reward = rewraw_main + 1000 * rewraw_quadcol)."""
    import torch
    from quad_swarm_rl_amd import sf_env
    argv = ["--quads_num_agents=4", "--quads_neighbor_visible_num=2", "--quads_neighbor_obs_type=pos_vel", "--quads_use_numba=True",
            "--quads_collision_reward=5.0", "--quads_collision_falloff_radius=4.0", "--quads_episode_duration=0.2",
            "--anneal_collision_steps=1000", "--quads_seed=9", "--quads_precision=f64"]
    single = sf_env.make_quadrotor_env("quadrotor_multi", cfg=parse(argv))
    batched = sf_env.make_quadrotor_env("quadrotor_multi", cfg=parse(argv + ["--quads_num_envs=3"]))
    assert batched.num_agents == 12 and isinstance(batched, sf_env.BatchedQuadSwarm)
    for env in (single, batched):
        env.set_training_info({"approx_total_training_steps": 400})
    obs_s, _ = single.reset()
    obs_b, _ = batched.reset()
    np.testing.assert_allclose(obs_b["obs"][:4].cpu().numpy(), obs_s, atol=1e-12)
    rng = np.random.RandomState(4)
    saw = 0
    acc = [dict() for _ in range(4)]
    for t in range(45):
        a = rng.uniform(-1, 1, size=(12, 4))
        o_s, r_s, term_s, _, inf_s = single.step([a[i] for i in range(4)])
        for i in range(4):
            for k, v in inf_s[i]["rewards"].items():
                acc[i][k] = acc[i].get(k, 0.0) + v
        o_b, r_b, term_b, trunc_b, inf_b = batched.step(torch.as_tensor(a, device="cuda:0", dtype=torch.float64))
        np.testing.assert_allclose(o_b["obs"][:4].cpu().numpy(), o_s, atol=1e-10)
        np.testing.assert_allclose(r_b[:4].cpu().numpy(), r_s, atol=1e-12)
        assert term_b[:4].cpu().numpy().tolist() == list(term_s) and not trunc_b.any()
        if term_s.any():
            saw += 1
            assert len(inf_b) == 12
            for i in range(4):
                assert inf_b[i]["true_reward"] == pytest.approx(inf_s[i]["true_reward"], abs=1e-9)
                assert inf_s[i]["true_reward"] == pytest.approx(acc[i]["rewraw_main"] + 1000 * acc[i]["rewraw_quadcol"], abs=1e-9)
                for k, v in acc[i].items():
                    if k != "rewraw_main":   # (the wrapper overwrites that entry with true_reward, reward_shaping.py:88)
                        assert inf_s[i]["episode_extra_stats"][k] == pytest.approx(v, abs=1e-9), k
                acc[i] = dict()
                es, eb = inf_s[i]["episode_extra_stats"], inf_b[i]["episode_extra_stats"]
                assert set(eb) == set(es), sorted(set(eb) ^ set(es))   # incl. the env's own episode_extra_stats (quadrotor_multi.py:637-718)
                for k, v in eb.items():
                    assert k in es, k
                    assert v == pytest.approx(es[k], abs=1e-9), k
        else:
            assert inf_b == []
    assert saw == 2
    single.close(); batched.close()


STATE_ARRAYS = ("pos", "vel", "rot", "omega", "thrust_rot_damp", "thrust_cmds_damp", "ou_state", "goal", "flags", "col_pair_mask",
                "new_pair_mask", "obst_hit_idx", "run_sums", "obs", "unique_col_mask", "counters", "tick", "scenario_id", "obst_pos")


def _env_slice(st, name, e):
    a = st.to_host(name)
    N, E, T = st.N, st.E, st.T
    if name == "obs":
        return a[e * N:(e + 1) * N].copy()
    if name == "obst_pos":
        M = a.shape[1] // E
        return a[:, e * M:(e + 1) * M].copy()
    if a.shape[-1] == T:
        return a[..., e * N:(e + 1) * N].copy()
    return a[..., e:e + 1].copy()


def test_snapshot_save_load_is_a_deep_copy_of_one_env():
    """qs_snapshot_save / load replace deepcopy(env) of the replay wrapper: every state array of the env comes back bit for bit,
    other envs are untouched, and a snapshot can be loaded into another env index."""
    from quad_swarm_rl_amd import config as qcfg, native
    st = native.Stepper(qcfg.make_config(num_envs=3, num_agents=4, neighbor_visible_num=2, neighbor_obs_type="pos_vel", use_numba=True,
                                         use_obstacles=True, quads_mode="o_random", obst_spawn_area=(8.0, 8.0), obst_density=0.2,
                                         obs_repr="xyz_vxyz_R_omega_floor", episode_sums=True, ep_time=0.6))
    st.snapshot_pool(2)
    st.reset()
    rng = np.random.RandomState(0)

    def run(k):
        for _ in range(k):
            st.from_host("actions", rng.uniform(-1, 1, size=(st.T, 4)).astype(np.float32))
            st.step()
        st.sync()

    run(25)
    saved = {n: _env_slice(st, n, 1) for n in STATE_ARRAYS}
    st.snapshot_save(1, 0)
    run(50)                                       # crosses an episode boundary (61 steps per episode)
    others = {n: (_env_slice(st, n, 0), _env_slice(st, n, 2)) for n in STATE_ARRAYS}
    assert any((saved[n] != _env_slice(st, n, 1)).any() for n in ("pos", "tick", "goal"))
    st.snapshot_load(0, 1)
    st.sync()
    for n in STATE_ARRAYS:
        np.testing.assert_array_equal(_env_slice(st, n, 1), saved[n], err_msg=n)
        np.testing.assert_array_equal(_env_slice(st, n, 0), others[n][0], err_msg=n)
        np.testing.assert_array_equal(_env_slice(st, n, 2), others[n][1], err_msg=n)
    st.snapshot_copy(0, 1)
    st.snapshot_load(1, 2)                        # same state into another env index
    st.sync()
    for n in STATE_ARRAYS:
        np.testing.assert_array_equal(_env_slice(st, n, 2), saved[n], err_msg=n)
    run(5)                                        # and the restored envs keep stepping
    st.check_errors()
    with pytest.raises(native.QsError):
        st.snapshot_save(0, 5)
    st.close()


def test_experience_replay_through_the_facade():
    """quad_experience_replay.py on the device (qs_replay_enable) seen through the single-env facade: checkpoints every 0.5 s, a
    collision after the grace period files the checkpoint from 1.5 s earlier - whose observation that step returns, like the
    reference - and the next episode starts from that checkpoint and reports the replay statistics."""
    from quad_swarm_rl_amd import sf_env
    cfg = parse(["--quads_num_agents=4", "--quads_neighbor_visible_num=2", "--quads_neighbor_obs_type=pos_vel", "--quads_use_numba=True",
                 "--quads_episode_duration=3.0", "--replay_buffer_sample_prob=1.0", "--quads_precision=f64", "--quads_seed=2"])
    env = sf_env.make_quadrotor_env("quadrotor_multi", cfg=cfg)
    quad = env.unwrapped                                # SingleQuadSwarm over a batch of one (replay on the device)
    assert quad.use_replay_buffer and not quad.activate_replay_buffer
    env.reset()
    quad.activate_replay_buffer = True                 # (the reference switches it on after 10 episodes without crashes, :280-287)
    st = quad._vec.stepper
    hover = [np.full(4, 0.06) for _ in range(4)]
    cp_obs = {}
    for t in range(1, 302):
        if t == 211:                                   # two drones on top of each other: a new collision pair at tick 211
            s, tick = st.get_state(0)
            s[1, 0:3] = s[0, 0:3] + np.array([0.03, 0.0, 0.0])
            st.set_state(0, s, tick)
        obs, rew, term, trunc, infos = env.step(hover)
        if t in (50, 100, 150, 200) and not term.any():
            cp_obs[t] = np.array(obs, copy=True)
        if t == 211:
            rs = st.replay_stats()
            assert rs["buffer_len"][0] == 1 and rs["checkpoints"][0] == 4 and rs["errors"][0] == 0
            np.testing.assert_array_equal(obs, cp_obs[100])   # the filed checkpoint's observation (quad_experience_replay.py:151)
        if t < 301:
            assert not term.any()
    assert term.all()                                  # tick 301 > ep_len 300
    st_info = infos[0]["episode_extra_stats"]
    assert st_info["replay/replay_rate"] == 1.0 and st_info["replay/replay_buffer_size"] == 1 and st_info["replay/avg_replayed"] == 1.0
    assert "num_collisions" in st_info and "true_reward" in infos[0]
    # steps_ago = 1.5 / 0.5 = 3 checkpoints before the collision: the one taken at tick 100
    assert quad.envs[0].tick == 100
    np.testing.assert_array_equal(obs, cp_obs[100])
    for t in range(101, 302):                          # the replayed episode runs to its own end
        obs, rew, term, trunc, infos = env.step(hover)
    assert term.all() and "num_collisions_replay" in infos[0]["episode_extra_stats"]
    env.close()
