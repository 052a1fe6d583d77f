"""Soak: every subsystem at once on the production (float32, config-specialised) kernels for thousands of steps - `mix` scenarios
with obstacles, domain-randomised obstacle density / size, device replay, auto-resets - with uniformly random actions, i.e. crashes,
collisions and wall hits all the time.  Nothing here is compared with the oracle (tests/test_hip_parity.py does that on short
horizons); the point is that nothing goes non-finite, out of range or inconsistent over long runs."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("obstacles", [False, True])
def test_long_random_rollout_stays_sane(obstacles):
    from quad_swarm_rl_amd import config as qcfg, native
    E, N, steps = 384, 8, 3500
    kw = dict(num_agents=N, neighbor_visible_num=6 if not obstacles else 2, neighbor_obs_type="pos_vel", use_numba=True, use_downwash=True,
              collision_falloff_radius=4.0, quads_mode="mix", ep_time=4.0, episode_sums=True, write_rew_info=False)
    if obstacles:
        kw.update(use_obstacles=True, obst_density=0.2, obst_size=0.6, obst_spawn_area=(8.0, 8.0), obs_repr="xyz_vxyz_R_omega_floor",
                  domain_random=True, obst_density_random=True, obst_size_random=True, obst_density_min=0.05, obst_density_max=0.2,
                  obst_size_min=0.3, obst_size_max=0.6)
    cfg = qcfg.make_config(num_envs=E, seed=77, **kw)
    st = native.Stepper(cfg)
    st.replay_enable(0.75)
    st.reset()
    st.replay_set_active(np.ones(E, dtype=np.uint8))
    rng = np.random.RandomState(3)
    ring = rng.uniform(-1, 1, size=(16, E * N, 4)).astype(np.float32)
    resets_seen = 0
    prev_tick = st.to_host("tick").copy()
    for t in range(steps):
        st.from_host("actions", ring[t % 16])
        st.step()
        if t % 250 == 249 or t == steps - 1:
            st.check_errors()                                          # NaN / Inf reward flag of quadrotor_single.py:87-90
            obs, rew, pos, rot = st.to_host("obs"), st.to_host("reward"), st.to_host("pos"), st.to_host("rot")
            assert np.isfinite(obs).all() and np.isfinite(rew).all() and np.isfinite(pos).all() and np.isfinite(rot).all()
            lo, hi = np.array(cfg.room_lo[:]), np.array(cfg.room_hi[:])
            for ax in range(3):                                        # the room clips positions (quadrotor_single.py:146-147)
                assert (pos[ax] >= lo[ax] - 1e-3).all() and (pos[ax] <= hi[ax] + 1e-3).all(), ax
            R = rot.reshape(3, 3, -1)
            ortho = np.einsum("ijn,kjn->ikn", R, R) - np.eye(3)[:, :, None]
            assert np.abs(ortho).max() < 5e-3                          # rotations stay orthonormal (SVD re-orthonormalisation cadence)
            tick = st.to_host("tick")
            assert (tick >= 0).all() and (tick <= cfg.ep_len + 1).all()
            resets_seen += int((tick < prev_tick).sum())
            prev_tick = tick
            cnt = st.to_host("counters")
            assert (cnt >= 0).all()
            if obstacles:
                oc = st.to_host("obst_count")
                assert (oc >= 1).all() and (oc <= cfg.num_obstacles).all()
    stats = st.replay_stats()
    assert all((v >= 0).all() for v in stats.values()) and (stats["errors"] == 0).all() and stats["episodes"].sum() > 0
    assert resets_seen > 0                                             # episodes did end and restart
    st.close()
