"""The device-side batched experience replay (qs_replay_enable: qs_replay_kernel behind every step) against its Python twin
tests/replay_model.py - which tests/test_replay_model_vs_reference.py pins against the REFERENCE wrapper's recorded behaviour.

24 environments with different collision histories run for ~4500 control steps: natural activation of the replay buffers (10
episodes with a mean crash reward above -1), checkpoints every 0.5 s, collisions filed once per 5 s from the checkpoint of
1.5 s ago, episodes restarted from events with probability 0.75.  After every step the kernel's per-environment state must equal
the model's (same Philox draws), the observation returned on a filing step must be the filed checkpoint's, and a restored
environment must equal the checkpoint bit for bit."""
import ctypes as C

import numpy as np
import pytest

from tests.replay_model import ReplayModel

pytestmark = pytest.mark.gpu
SITE_REPLAY = 25


def philox_u(seed, env, ctr, slot):
    """u01 of word 0 of the Philox group the kernel draws: counter = {env, step_ctr, site | slot << 8, 0}"""
    from oracle import oracle as orc
    c = (C.c_uint32 * 4)(env, ctr, SITE_REPLAY | (slot << 8), 0)
    k = (C.c_uint32 * 2)(seed & 0xffffffff, seed >> 32)
    out = (C.c_uint32 * 4)()
    orc.lib().qso_philox4x32(c, k, out)
    return np.float32((np.float32(out[0] >> 9) + np.float32(0.5)) * np.float32(1.0 / 8388608.0))


def test_replay_kernel_equals_the_model():
    from quad_swarm_rl_amd import config as qcfg, native
    E, N, seed, prob = 24, 4, 5, 0.75
    cfg = qcfg.make_config(num_envs=E, num_agents=N, neighbor_visible_num=2, neighbor_obs_type="pos_vel", use_numba=True, ep_time=2.6, seed=seed,
                           env_id_offset=7, precision="f64", episode_sums=True, write_rew_info=False)
    st = native.Stepper(cfg, device=0)
    assert native.lib().qs_replay_enable(st._h, 2.0) != 0            # sample_prob outside [0, 1]
    st.replay_enable(prob)
    with pytest.raises(native.QsError):
        st.replay_enable(prob)                                       # once per handle
    st.reset()
    forced = np.array([e % 2 for e in range(E)], dtype=np.uint8)      # half of the buffers switched on by hand, the others by the rule
    st.replay_set_active(forced)
    models = [ReplayModel(prob, control_freq=100, use_obstacles=False, active=bool(forced[e])) for e in range(E)]
    D, T = st.obs_dim, E * N
    rng = np.random.RandomState(1)
    snaps = [dict() for _ in range(E)]                                # per env: pool slot -> (pos, vel, rot, tick, obs) of the snapshot
    ticks = np.zeros(E, dtype=np.int64)
    n_file = n_restore = n_save = 0
    steps = 4600
    for step in range(1, steps + 1):
        # a collision (two drones 3 cm apart) at tick 170 in a third of the envs, in episodes that are not replays
        for e in np.nonzero(ticks == 169)[0]:
            if e % 3 == 0 and not models[e].saved:
                s, tk = st.get_state(int(e))
                s[1, 0:3] = s[0, 0:3] + np.array([0.03, 0.0, 0.0]); s[1, 3:6] = s[0, 3:6]
                st.set_state(int(e), s, tk)
        act = 0.06 + rng.uniform(-0.02, 0.02, size=(T, 4))           # hovering: the drones stay in the air, crash rewards stay at zero
        st.from_host("actions", act)
        st.step()
        st.sync()
        done = st.to_host("done").reshape(E, N)[:, 0].astype(bool)
        ticks = st.to_host("tick").astype(np.int64)
        uq = st.to_host("unique_col_mask")
        ctr = 1 + step                                                # qs_reset took counter 1, step k takes 1 + k
        need_state = False
        acts = []
        crash = st.to_host("ep_sums")[3].reshape(E, N)[:, 0] if done.any() else None
        for e in range(E):
            a = models[e].step(bool(done[e]), int(ticks[e]), int(uq[e]), 0, float(crash[e]) if done[e] else 0.0,
                               lambda: philox_u(seed, 7 + e, ctr, 0),
                               lambda n: min(int(np.float32(philox_u(seed, 7 + e, ctr, 1)) * np.float32(n)), n - 1))
            a = [x for x in a if x[0] != "fresh"]
            acts.append(a)
            need_state |= bool(a)
        rs = st.replay_stats()
        for e in range(E):
            ms = models[e].stats()
            for k in ("episodes", "replayed", "buffer_len", "replayed_sum", "active", "checkpoints", "errors"):
                assert rs[k][e] == ms[k], (step, e, k, rs[k][e], ms[k])
        if need_state:
            pos, vel, rot = st.to_host("pos").reshape(3, E, N), st.to_host("vel").reshape(3, E, N), st.to_host("rot").reshape(9, E, N)
            obs = st.to_host("obs").reshape(E, N, D)
            for e, a in enumerate(acts):
                both = len(a) == 2     # a checkpoint saved AND an event filed on this step: the live observation was overwritten after the save
                for x in a:
                    if x[0] == "save":
                        snaps[e][x[1]] = (pos[:, e].copy(), vel[:, e].copy(), rot[:, e].copy(), int(ticks[e]), None if both else obs[e].copy())
                        n_save += 1
                    elif x[0] == "file":
                        snaps[e][x[2]] = snaps[e][x[1]]
                        if snaps[e][x[1]][4] is not None:
                            np.testing.assert_array_equal(obs[e], snaps[e][x[1]][4])  # the 1.5-s-old observation is returned on this step
                        assert ticks[e] - snaps[e][x[1]][3] >= 100
                        n_file += 1
                    else:
                        p0, v0, r0, t0, o0 = snaps[e][x[1]]
                        np.testing.assert_array_equal(pos[:, e], p0); np.testing.assert_array_equal(vel[:, e], v0)
                        np.testing.assert_array_equal(rot[:, e], r0)
                        if o0 is not None:
                            np.testing.assert_array_equal(obs[e], o0)
                        assert ticks[e] == t0 and rs["ep_steps"][e] >= 1
                        n_restore += 1
            cnt = st.to_host("counters")
            for e, a in enumerate(acts):
                if a and a[-1][0] == "restore":
                    assert cnt[0, e] == 0 and cnt[1, e] == 0                          # collision counters zeroed on the replayed env
    st.check_errors()
    assert all(m.active for m in models)                              # the rule switched the other half on after 10 clean episodes
    assert n_save > 500 and n_file >= 8 and n_restore >= 6, (n_save, n_file, n_restore)
    st.close()


def test_explicit_reset_feeds_the_crash_history():
    """qs_reset while the replay wrapper runs: every explicit reset of an environment records the running episode's crash reward, like
    the reference's reset() does (quadrotor_multi.py:356-359) - ten recorded episodes with a mean above -1 switch the buffer on.  The
    masked-out environments are untouched, and the action statistics of the next finished episode count its steps from tick 0."""
    from quad_swarm_rl_amd import config as qcfg, native
    E, N = 6, 4
    cfg = qcfg.make_config(num_envs=E, num_agents=N, neighbor_visible_num=2, neighbor_obs_type="pos_vel", use_numba=True, ep_time=0.5, seed=3,
                           precision="f32", episode_sums=True, write_rew_info=False)
    st = native.Stepper(cfg, device=0)
    st.replay_enable(0.5)
    st.reset()
    models = [ReplayModel(0.5, control_freq=100) for _ in range(E)]
    act = np.full((E * N, 4), 0.06, dtype=np.float32)
    mask = np.array([1, 0, 1, 1, 0, 0], dtype=np.uint8)
    for rnd in range(9):                                # 1 (first reset) + 9 explicit resets = 10 entries => active
        for _ in range(5):
            st.from_host("actions", act)
            st.step()
        st.sync()
        crash = st.to_host("run_sums")[3].reshape(E, N)[:, 0]
        assert not st.to_host("done").any()
        st.reset(mask)
        for e in np.nonzero(mask)[0]:
            models[e].explicit_reset(float(crash[e]))
        rs = st.replay_stats()
        assert rs["active"].tolist() == [int(m.active) for m in models], (rnd, rs["active"].tolist())
        assert (st.to_host("tick")[mask == 1] == 0).all() and (st.to_host("tick")[mask == 0] == 5 * (rnd + 1)).all()
    assert rs["active"].tolist() == [1, 0, 1, 1, 0, 0]
    # the reset environments finish their next episode after ep_len + 1 steps, counted from the reset
    for _ in range(cfg.ep_len + 1):
        st.from_host("actions", act)
        st.step()
    st.sync()
    rs = st.replay_stats()
    done = st.to_host("done").reshape(E, N)[:, 0]
    assert done[mask == 1].all() and (rs["ep_steps"][mask == 1] == cfg.ep_len + 1).all()
    st.close()


def test_domain_random_through_the_batched_env():
    """--quads_domain_random with the replay wrapper (quad_experience_replay.py:75-88,:106-118,:191-206): every new episode of every
    environment draws its obstacle density and size; the wrapper's statistics report them."""
    import argparse
    import torch
    from quad_swarm_rl_amd import sf_env
    p = argparse.ArgumentParser()
    p.add_argument("--with_pbt", default=False)
    sf_env.add_quadrotors_env_args("quadrotor_multi", p)
    cfg = p.parse_args(["--quads_num_agents=8", "--quads_neighbor_visible_num=2", "--quads_neighbor_obs_type=pos_vel", "--quads_use_numba=True",
                        "--quads_use_obstacles=True", "--quads_obstacle_obs_type=octomap", "--quads_obst_spawn_area", "8", "8", "--quads_obst_size=0.6",
                        "--quads_mode=o_random", "--quads_obs_repr=xyz_vxyz_R_omega_floor", "--quads_episode_duration=0.3", "--quads_num_envs=32",
                        "--replay_buffer_sample_prob=0.5", "--quads_domain_random=True", "--quads_obst_density_random=True", "--quads_obst_size_random=True",
                        "--quads_obst_density_min=0.05", "--quads_obst_density_max=0.2", "--quads_obst_size_min=0.3", "--quads_obst_size_max=0.6"])
    env = sf_env.make_quadrotor_env("quadrotor_multi", cfg=cfg)
    assert env.vec.cfg.dr_num_density == 4 and env.vec.cfg.dr_num_size == 3 and env.vec.cfg.num_obstacles == 12
    env.reset()
    seen_d, seen_s = set(), set()
    act = torch.full((env.num_agents, 4), 0.06, device="cuda")
    for t in range(100):
        obs, rew, term, trunc, infos = env.step(act)
        if term.any():
            for inf in infos:
                seen_d.add(round(inf["episode_extra_stats"]["replay/obst_density"], 6))
                seen_s.add(round(inf["episode_extra_stats"]["replay/obst_size"], 6))
    assert seen_d == {0.05, 0.1, 0.15, 0.2} and seen_s == {0.3, 0.4, 0.5}
    cnt = env.vec.stepper.to_host("obst_count")
    assert set(cnt.tolist()) <= {3, 6, 9, 12} and len(set(cnt.tolist())) > 1
    assert torch.isfinite(obs["obs"]).all()
    env.close()
