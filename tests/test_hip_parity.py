"""GPU parity tests proper: the HIP stepper (through the C ABI) against the CPU oracle on the same seeded inputs.

Both sides draw every stochastic term from the same counter-based Philox stream (include/quadswarm.h), so
"same seed" literally means same noise.  Two modes:
  * f64 instantiation, free-running rollouts: floats to 1e-8, every flag / index / mask / counter exact;
  * f32 instantiation (the production precision), teacher-forced: before each step the device state is
    overwritten with the oracle's, one control step is taken, outputs must agree to 1e-5 (the tolerance
    north_star states for fp32 dynamics state) and the discrete outputs exactly.
"""
import os

import numpy as np
import pytest

from quad_swarm_rl_amd import config as qcfg
from tests import tolerances as tolr

pytestmark = pytest.mark.gpu

REW = dict(pos=1.0, effort=0.05, spin=0.1, crash=1.0, orient=1.0, quadcol_bin=5.0, quadcol_bin_smooth_max=10.0,
           quadcol_bin_obst=5.0)

CASES = {
    "c1_single": dict(num_agents=1, neighbor_visible_num=0, neighbor_obs_type="none", use_numba=False),
    "c2_n8_dw": dict(num_agents=8, neighbor_visible_num=6, neighbor_obs_type="pos_vel", use_downwash=True, use_numba=True,
                     collision_falloff_radius=4.0, rew_coeff=REW),
    "c2_n8_k2_numpy_wall": dict(num_agents=8, neighbor_visible_num=2, neighbor_obs_type="pos_vel", use_downwash=True,
                                use_numba=False, collision_falloff_radius=4.0, rew_coeff=REW, obs_repr="xyz_vxyz_R_omega_wall"),
    "c2_n5_kall_short": dict(num_agents=5, neighbor_visible_num=-1, neighbor_obs_type="pos_vel", use_downwash=True,
                             use_numba=True, collision_falloff_radius=4.0, rew_coeff=REW, ep_time=0.4),
    "c3_n8_obst": dict(num_agents=8, neighbor_visible_num=2, neighbor_obs_type="pos_vel", use_downwash=True, use_numba=True,
                       collision_falloff_radius=4.0, rew_coeff=dict(REW, quadcol_bin_smooth_max=4.0), use_obstacles=True,
                       obst_density=0.2, obst_size=0.6, obst_spawn_area=(8.0, 8.0), quads_mode="o_static_same_goal",
                       obs_repr="xyz_vxyz_R_omega_floor"),
    "c3_n8_obst_short": dict(num_agents=8, neighbor_visible_num=2, neighbor_obs_type="pos_vel", use_downwash=False,
                             use_numba=True, collision_falloff_radius=4.0, rew_coeff=REW, use_obstacles=True,
                             obst_density=0.2, obst_size=0.6, obst_spawn_area=(8.0, 8.0), quads_mode="o_static_same_goal",
                             obs_repr="xyz_vxyz_R_omega_floor", ep_time=0.5),
    "c4_n32_svs": dict(num_agents=32, neighbor_visible_num=6, neighbor_obs_type="pos_vel", use_downwash=True, use_numba=True,
                       collision_falloff_radius=4.0, rew_coeff=REW, quads_mode="swarm_vs_swarm"),
    # the remaining scenarios (full-scenario kernel variants), short episodes so that resets and goal dynamics both occur
    "s_static_diff": dict(num_agents=10, neighbor_visible_num=6, neighbor_obs_type="pos_vel", use_numba=True, collision_falloff_radius=4.0,
                          rew_coeff=REW, quads_mode="static_diff_goal", ep_time=0.3),
    "s_dynamic_same": dict(num_agents=3, neighbor_visible_num=2, neighbor_obs_type="pos_vel", use_numba=True, collision_falloff_radius=4.0,
                           rew_coeff=REW, quads_mode="dynamic_same_goal", ep_time=15.0),
    "s_dynamic_diff": dict(num_agents=5, neighbor_visible_num=2, neighbor_obs_type="pos_vel", use_numba=True, collision_falloff_radius=4.0,
                           rew_coeff=REW, quads_mode="dynamic_diff_goal", ep_time=15.0),
    "s_dynamic_formations": dict(num_agents=9, neighbor_visible_num=6, neighbor_obs_type="pos_vel", use_numba=True,
                                 collision_falloff_radius=4.0, rew_coeff=REW, quads_mode="dynamic_formations", ep_time=0.4),
    "s_swap_goals": dict(num_agents=4, neighbor_visible_num=2, neighbor_obs_type="pos_vel", use_numba=True, collision_falloff_radius=4.0,
                         rew_coeff=REW, quads_mode="swap_goals", ep_time=15.0),
    "s_lissajous": dict(num_agents=3, neighbor_visible_num=2, neighbor_obs_type="pos_vel", use_numba=True, collision_falloff_radius=4.0,
                        rew_coeff=REW, quads_mode="ep_lissajous3D", ep_time=0.4),
    "s_bezier": dict(num_agents=2, neighbor_visible_num=1, neighbor_obs_type="pos_vel", use_numba=True, collision_falloff_radius=4.0,
                     rew_coeff=REW, quads_mode="ep_rand_bezier", ep_time=15.0),
    "s_o_random": dict(num_agents=8, neighbor_visible_num=2, neighbor_obs_type="pos_vel", use_numba=True, collision_falloff_radius=4.0,
                       rew_coeff=REW, use_obstacles=True, obst_density=0.2, obst_size=0.6, obst_spawn_area=(8.0, 8.0), quads_mode="o_random",
                       obs_repr="xyz_vxyz_R_omega_floor", ep_time=0.3),
    "s_o_dynamic_same": dict(num_agents=3, neighbor_visible_num=2, neighbor_obs_type="pos_vel", use_numba=True, collision_falloff_radius=4.0,
                             rew_coeff=REW, use_obstacles=True, obst_density=0.2, obst_size=0.6, obst_spawn_area=(8.0, 8.0),
                             quads_mode="o_dynamic_same_goal", obs_repr="xyz_vxyz_R_omega_floor", ep_time=15.0),
    "s_o_swap": dict(num_agents=4, neighbor_visible_num=2, neighbor_obs_type="pos_vel", use_numba=True, collision_falloff_radius=4.0,
                     rew_coeff=REW, use_obstacles=True, obst_density=0.2, obst_size=0.6, obst_spawn_area=(8.0, 8.0), quads_mode="o_swap_goals",
                     obs_repr="xyz_vxyz_R_omega_floor", ep_time=15.0),
    "s_o_ep_bezier": dict(num_agents=2, neighbor_visible_num=1, neighbor_obs_type="pos_vel", use_numba=True, collision_falloff_radius=4.0,
                          rew_coeff=REW, use_obstacles=True, obst_density=0.2, obst_size=0.6, obst_spawn_area=(8.0, 8.0),
                          quads_mode="o_ep_rand_bezier", obs_repr="xyz_vxyz_R_omega_floor", ep_time=15.0),
    "s_o_ep_bezier_short": dict(num_agents=3, neighbor_visible_num=2, neighbor_obs_type="pos_vel", use_numba=True, collision_falloff_radius=4.0,
                                rew_coeff=REW, use_obstacles=True, obst_density=0.2, obst_size=0.6, obst_spawn_area=(8.0, 8.0),
                                quads_mode="o_ep_rand_bezier", obs_repr="xyz_vxyz_R_omega_floor", ep_time=0.2),
    "s_run_away": dict(num_agents=5, neighbor_visible_num=2, neighbor_obs_type="pos_vel", use_numba=True, collision_falloff_radius=4.0,
                       rew_coeff=REW, quads_mode="run_away", ep_time=2.3),
    "s_mix": dict(num_agents=6, neighbor_visible_num=3, neighbor_obs_type="pos_vel", use_numba=True, use_downwash=True,
                  collision_falloff_radius=4.0, rew_coeff=REW, quads_mode="mix", ep_time=0.12),
    "s_mix_obst": dict(num_agents=4, neighbor_visible_num=2, neighbor_obs_type="pos_vel", use_numba=True, collision_falloff_radius=4.0,
                       rew_coeff=REW, use_obstacles=True, obst_density=0.2, obst_size=0.6, obst_spawn_area=(8.0, 8.0), quads_mode="mix",
                       obs_repr="xyz_vxyz_R_omega_floor", ep_time=0.12),
    # size edges: the 64-drone maximum (pair / id sets fill the 64-bit masks), all-neighbour and K > 8 ranked observations, a drone
    # count that leaves most of a wave idle, the 2- and 1-drone minimum
    "e_n64_k6": dict(num_agents=64, neighbor_visible_num=6, neighbor_obs_type="pos_vel", use_downwash=True, use_numba=True,
                     collision_falloff_radius=4.0, rew_coeff=REW, ep_time=0.5),
    "e_n40_kall": dict(num_agents=40, neighbor_visible_num=-1, neighbor_obs_type="pos_vel", use_downwash=True, use_numba=True,
                       collision_falloff_radius=4.0, rew_coeff=REW, quads_mode="swarm_vs_swarm"),
    "e_n64_k20": dict(num_agents=64, neighbor_visible_num=20, neighbor_obs_type="pos_vel", use_numba=True,
                      collision_falloff_radius=4.0, rew_coeff=REW, quads_mode="dynamic_formations", ep_time=0.4),
    "e_n33_k8": dict(num_agents=33, neighbor_visible_num=8, neighbor_obs_type="pos_vel", use_downwash=True, use_numba=False,
                     collision_falloff_radius=4.0, rew_coeff=REW, obs_repr="xyz_vxyz_R_omega_wall"),
    "e_n17_kall_obst": dict(num_agents=17, neighbor_visible_num=-1, neighbor_obs_type="pos_vel", use_downwash=True, use_numba=True,
                            collision_falloff_radius=4.0, rew_coeff=REW, use_obstacles=True, obst_density=0.2, obst_size=0.6,
                            obst_spawn_area=(8.0, 8.0), quads_mode="o_random", obs_repr="xyz_vxyz_R_omega_floor", ep_time=0.4),
    "e_n2_k1": dict(num_agents=2, neighbor_visible_num=1, neighbor_obs_type="pos_vel", use_downwash=True, use_numba=True,
                    collision_falloff_radius=4.0, rew_coeff=REW, quads_mode="swap_goals", ep_time=0.5),
    "e_n1_obst": dict(num_agents=1, neighbor_visible_num=0, neighbor_obs_type="none", use_numba=True, use_obstacles=True, obst_density=0.2,
                      obst_size=0.6, obst_spawn_area=(8.0, 8.0), quads_mode="o_static_same_goal", obs_repr="xyz_vxyz_R_omega_floor",
                      rew_coeff=REW, ep_time=0.3),
    # configuration corners: blind multi-agent, no noise at all, dense obstacle field, small room, two-step episodes, odd team split,
    # four sub-steps per control step, single-drone mix, wide hitbox, 40 drones among obstacles (52 free cells)
    "x_n8_blind": dict(num_agents=8, neighbor_visible_num=0, neighbor_obs_type="none", use_downwash=True, use_numba=True,
                       collision_falloff_radius=4.0, rew_coeff=REW, ep_time=0.5),
    "x_no_noise": dict(num_agents=8, neighbor_visible_num=6, neighbor_obs_type="pos_vel", use_downwash=True, use_numba=True,
                       collision_falloff_radius=4.0, rew_coeff=REW, sense_noise=None, thrust_noise_ratio=0.0),
    "x_dense_obst": dict(num_agents=8, neighbor_visible_num=2, neighbor_obs_type="pos_vel", use_numba=True, collision_falloff_radius=4.0,
                         rew_coeff=REW, use_obstacles=True, obst_density=0.8, obst_size=0.5, obst_spawn_area=(8.0, 8.0), quads_mode="o_random",
                         obs_repr="xyz_vxyz_R_omega_floor", ep_time=0.4),
    "x_small_room": dict(num_agents=8, neighbor_visible_num=6, neighbor_obs_type="pos_vel", use_downwash=True, use_numba=True,
                         collision_falloff_radius=4.0, rew_coeff=REW, room_dims=(6.0, 6.0, 4.0), obs_repr="xyz_vxyz_R_omega_wall",
                         quads_mode="dynamic_diff_goal", ep_time=0.6),
    "x_ep_len2": dict(num_agents=5, neighbor_visible_num=2, neighbor_obs_type="pos_vel", use_downwash=True, use_numba=True,
                      collision_falloff_radius=4.0, rew_coeff=REW, quads_mode="static_diff_goal", ep_time=0.02),
    "x_svs_odd": dict(num_agents=9, neighbor_visible_num=6, neighbor_obs_type="pos_vel", use_downwash=True, use_numba=True,
                      collision_falloff_radius=4.0, rew_coeff=REW, quads_mode="swarm_vs_swarm", ep_time=0.3),
    "x_sim4": dict(num_agents=4, neighbor_visible_num=-1, neighbor_obs_type="pos_vel", use_downwash=True, use_numba=True,
                   collision_falloff_radius=4.0, rew_coeff=REW, sim_steps=4, sim_freq=400.0, ep_time=0.5),
    "x_mix_single": dict(num_agents=1, neighbor_visible_num=0, neighbor_obs_type="none", use_numba=True, rew_coeff=REW, quads_mode="mix",
                         ep_time=0.1),
    "x_hitbox": dict(num_agents=8, neighbor_visible_num=6, neighbor_obs_type="pos_vel", use_downwash=True, use_numba=True,
                     collision_hitbox_radius=3.0, collision_falloff_radius=6.0,
                     rew_coeff=dict(pos=0.5, effort=0.1, spin=0.2, orient=0.7, crash=2.0, quadcol_bin=3.0, quadcol_bin_smooth_max=7.0)),
    "x_n40_obst": dict(num_agents=40, neighbor_visible_num=6, neighbor_obs_type="pos_vel", use_downwash=True, use_numba=True,
                       collision_falloff_radius=4.0, rew_coeff=REW, use_obstacles=True, obst_density=0.2, obst_size=0.6,
                       obst_spawn_area=(8.0, 8.0), quads_mode="o_random", obs_repr="xyz_vxyz_R_omega_floor", ep_time=0.4),
    # --quads_domain_random: every episode draws one of 4 obstacle densities (3 / 6 / 9 / 12 obstacles of an 8 x 8 area) and one of 3 sizes
    "x_domain_random": dict(num_agents=8, neighbor_visible_num=2, neighbor_obs_type="pos_vel", use_downwash=True, use_numba=True,
                            collision_falloff_radius=4.0, rew_coeff=REW, use_obstacles=True, obst_density=0.2, obst_size=0.6,
                            obst_spawn_area=(8.0, 8.0), quads_mode="o_random", obs_repr="xyz_vxyz_R_omega_floor", ep_time=0.25,
                            domain_random=True, obst_density_random=True, obst_size_random=True),
    "x_domain_random_static": dict(num_agents=5, neighbor_visible_num=2, neighbor_obs_type="pos_vel", use_numba=True,
                                   collision_falloff_radius=4.0, rew_coeff=REW, use_obstacles=True, obst_density=0.2, obst_size=0.6,
                                   obst_spawn_area=(8.0, 8.0), quads_mode="o_static_same_goal", obs_repr="xyz_vxyz_R_omega_floor", ep_time=0.2,
                                   domain_random=True, obst_density_random=True, obst_size_random=False, obst_density_min=0.1, obst_density_max=0.3),
    "c4_n12_svs_short": dict(num_agents=12, neighbor_visible_num=6, neighbor_obs_type="pos_vel", use_downwash=True,
                             use_numba=True, collision_falloff_radius=4.0, rew_coeff=REW, quads_mode="swarm_vs_swarm", ep_time=0.1),
}


def soa(a, E, N):
    """[C, E*N] -> [E, N, C]"""
    return np.ascontiguousarray(a.reshape(a.shape[0], E, N).transpose(1, 2, 0))


def force_events(t, e, s, n, obst_xy=None, obst_r=0.3):
    """Crafted states (same idea as oracle/ref_harness/capture.py) applied to env e before step t."""
    changed = False
    yaw = lambda th: np.array([np.cos(th), -np.sin(th), 0, np.sin(th), np.cos(th), 0, 0, 0, 1.0])
    if n >= 8 and t == 6 and e % 2 == 0:
        s[1, 0:3] = [1.0, 1.0, 3.0]; s[1, 3:6] = [0.8, 0.0, 0.0]
        s[2, 0:3] = [1.12, 1.0, 3.0]; s[2, 3:6] = [-0.8, 0.1, 0.0]
        s[3, 0:3] = [-1.0, 1.0, 2.5]; s[3, 3:6] = [0.0, 0.3, 0.0]
        s[5, 0:3] = [-1.0, 1.05, 2.52]; s[5, 3:6] = [0.0, -0.3, 0.1]
        s[0, 0:3] = [2.0, -2.0, 2.0]; s[4, 0:3] = [2.0, -1.85, 2.0]
        s[6, 0:3] = [-2.0, -2.0, 3.0]; s[6, 6:15] = yaw(0.3)
        s[7, 0:3] = [-2.03, -1.98, 2.6]
        changed = True
    if n >= 3 and t in (14, 16) and e % 2 == 1:      # ids {0}: the `.any()` quirk
        # (heights a few mm apart: at equal height the sign of the body-frame dz that gates the downwash, downwash.py:24-27, is
        # decided by rounding and fp32 and fp64 may disagree)
        s[0, 0:3] = [0.0, 0.0, 4.0]; s[1, 0:3] = [0.05, 0.0, 4.003]; s[2, 0:3] = [0.10, 0.0, 3.996]
        s[0:3, 3:6] = 0
        changed = True
    if n >= 3 and t == 15 and e % 2 == 1:
        s[0, 0:3] = [3.0, 3.0, 4.0]; s[1, 0:3] = [0.05, 0.0, 4.003]; s[2, 0:3] = [0.10, 0.0, 3.996]
        s[0:3, 3:6] = 0
        changed = True
    if t == 22:
        s[0, 0:3] = [4.99, 0.0, 3.0]; s[0, 3:6] = [3.0, 0.5, 0.0]
        if n >= 6:
            s[1, 0:3] = [-4.995, -4.99, 3.0]; s[1, 3:6] = [-2.0, -2.5, 0.2]
            s[2, 0:3] = [0.0, 1.0, 9.99]; s[2, 3:6] = [0.2, 0.0, 4.0]
            s[3, 0:3] = [1.0, 4.99, 9.99]; s[3, 3:6] = [0.0, 3.0, 3.0]
            s[4, 0:3] = [0.5, 0.5, 0.2]; s[4, 3:6] = [0.3, 0.1, -2.0]
            s[5, 0:3] = [-1.5, 0.5, 0.2]; s[5, 3:6] = [0.0, 0.4, -2.0]; s[5, 6:15] = [1, 0, 0, 0, -1, 0, 0, 0, -1.0]
        changed = True
    if obst_xy is not None and t in (30, 31) and n >= 3:
        s[0, 0:3] = [obst_xy[3, 0] + obst_r + 0.06, obst_xy[3, 1], 2.0]; s[0, 3:6] = [-1.5, 0.2, 0.0]
        s[1, 0:3] = [obst_xy[5, 0] + 0.1, obst_xy[5, 1] - 0.05, 5.05]; s[1, 3:6] = [0.3, 0.0, 0.0]   # inside (z ~ room mid)
        s[2, 0:3] = [obst_xy[0, 0], obst_xy[0, 1] - obst_r - 0.03, 3.0]; s[2, 3:6] = [0.0, 0.5, 0.0]
        changed = True
    return changed


class Pair:
    """An oracle batch and a HIP stepper built from the same configuration."""

    def __init__(self, case, E, precision, seed=1234, env_id_offset=3, **over):
        from oracle import oracle as orc
        from quad_swarm_rl_amd import native
        kw = dict(CASES[case], **over)
        self.cfg = qcfg.make_config(num_envs=E, seed=seed, env_id_offset=env_id_offset, precision=precision, **kw)
        self.E, self.N = E, self.cfg.num_agents
        self.context, self.precision = case, precision
        self.oenvs = [orc.OracleEnv(self.cfg, env_global_id=env_id_offset + e) for e in range(E)]
        self.hip = native.Stepper(self.cfg, device=0)
        self.D = self.hip.obs_dim
        self.obs_layout = (self.D - 6 * self.cfg.num_neighbors - (9 if self.cfg.use_obstacles else 0), self.cfg.num_neighbors)   # (self columns, neighbour blocks)

    def reset(self):
        oobs = np.stack([o.reset() for o in self.oenvs])
        self.hip.reset()
        return oobs, self.hip.to_host("obs").reshape(self.E, self.N, self.D)

    def step(self, actions):
        o_obs, o_rew, o_done, o_ri = zip(*[o.step(actions[e]) for e, o in enumerate(self.oenvs)])
        self.hip.from_host("actions", actions.reshape(-1, 4))
        self.hip.step()
        self.hip.sync()
        h = self.hip
        return (np.stack(o_obs), np.stack(o_rew), np.stack(o_done), np.stack(o_ri)), \
               (h.to_host("obs").reshape(self.E, self.N, self.D), h.to_host("reward").reshape(self.E, self.N),
                h.to_host("done").reshape(self.E, self.N), soa(h.to_host("rew_info"), self.E, self.N))

    def compare_discrete(self, t):
        h, E, N = self.hip, self.E, self.N
        flags = h.to_host("flags").reshape(E, N)
        cp, npm = h.to_host("col_pair_mask").reshape(E, N), h.to_host("new_pair_mask").reshape(E, N)
        uq, on, rn = h.to_host("unique_col_mask"), h.to_host("obst_new_mask"), h.to_host("room_new_mask")
        cnt, tick, ohi = h.to_host("counters"), h.to_host("tick"), h.to_host("obst_hit_idx").reshape(E, N)
        for e, o in enumerate(self.oenvs):
            info = o.info()
            assert tick[e] == info.tick, f"tick step {t} env {e}"
            np.testing.assert_array_equal(flags[e] & 0x7ff, np.array(info.flags[:N]) & 0x7ff, err_msg=f"flags step {t} env {e}")
            np.testing.assert_array_equal(cnt[:, e], np.array(info.counters), err_msg=f"counters step {t} env {e}")
            if info.tick != 0:     # (masks are cleared by the reset that ends an episode)
                assert uq[e] == info.unique_col_mask, f"unique collision ids step {t} env {e}"
                assert on[e] == info.obst_new_mask and rn[e] == info.room_new_mask, f"obst/room masks step {t} env {e}"
                np.testing.assert_array_equal(cp[e], np.array(info.col_pair_mask[:N], dtype=np.uint64))
                np.testing.assert_array_equal(npm[e], np.array(info.new_pair_mask[:N], dtype=np.uint64))
                if self.cfg.use_obstacles:
                    np.testing.assert_array_equal(ohi[e], np.array(info.obst_hit_idx[:N]))

    def compare_state(self, t, tol):
        """per quantity (tests/tolerances.py): position, velocity, rotation matrix, goal absolute `tol`, angular velocity tol * max(1, |w|)"""
        h, E, N = self.hip, self.E, self.N
        pos, vel, om, rot = soa(h.to_host("pos"), E, N), soa(h.to_host("vel"), E, N), soa(h.to_host("omega"), E, N), soa(h.to_host("rot"), E, N)
        goal = soa(h.to_host("goal"), E, N)
        worst = 0.0
        for e, o in enumerate(self.oenvs):
            s, _ = o.get_state()
            for nm, a, b in (("pos", pos[e], s[:, 0:3]), ("vel", vel[e], s[:, 3:6]), ("rot", rot[e], s[:, 6:15]),
                             ("omega", om[e], s[:, 15:18]), ("goal", goal[e], s[:, 32:35])):
                worst = max(worst, np.abs(a - b).max())
                allowed = tolr.allowed_vec(b, tol) if nm in ("omega", "vel") else tolr.allowed_abs(b, tol)
                tolr.check(f"{self.context} {self.precision}", nm, a, b, allowed, f"step {t} env {e}")
        return worst

    def compare_ep_stats(self, t, tol):
        h, E, N = self.hip, self.E, self.N
        eps, epc = soa(h.to_host("ep_stats"), E, N), h.to_host("ep_counters")
        for e, o in enumerate(self.oenvs):
            info = o.info()
            np.testing.assert_allclose(eps[e], np.array(info.ep_stats)[:N], rtol=tol, atol=tol, err_msg=f"episode stats step {t} env {e}")
            np.testing.assert_array_equal(epc[:, e], np.array(info.ep_counters))

    def obst_xy(self, e):
        M = self.cfg.num_obstacles
        op = self.hip.to_host("obst_pos")
        return np.stack([op[0, e * M:(e + 1) * M], op[1, e * M:(e + 1) * M]], axis=1)

    def close(self):
        self.hip.close()
        for o in self.oenvs:
            o.close()


def check_floats(t, tol, o, h, what, layout=None):
    """oracle outputs `o` against the stepper's `h`, per quantity (tests/tolerances.py): observation columns absolute `tol` (the angular-
    velocity columns tol * max(1, |w|)), reward and its terms tol * max(1, |r|)"""
    for nm, a, b in zip(("obs", "reward", "done", "rew_info"), o, h):
        if nm == "done":
            np.testing.assert_array_equal(a, b, err_msg=f"done step {t}")
            continue
        allowed = tolr.allowed_obs(a, tol, *(layout or ())) if nm == "obs" else tolr.allowed_rel(a, tol)
        tolr.check(what, nm, b, a, allowed, f"step {t}")


LONG = {"s_run_away": 320, "s_o_ep_bezier": 640, "s_dynamic_same": 640, "s_dynamic_diff": 640, "s_swap_goals": 640, "s_bezier": 560, "s_o_dynamic_same": 640, "s_o_swap": 640}


@pytest.mark.parametrize("case", list(CASES))
def test_rollout_f64_bit_exact_discrete(case):
    E, steps, tol = (3, LONG[case], 1e-7) if case in LONG else (3, 60, 1e-8) if case.startswith(("e_", "x_")) else (11, 70, 1e-8)
    rollout_f64(case, E, steps, tol)


SINGLE_WAVE_CASES = ["c1_single", "c2_n8_dw", "c2_n8_k2_numpy_wall", "c2_n5_kall_short", "c3_n8_obst", "c3_n8_obst_short", "c4_n32_svs",
                     "c4_n12_svs_short", "s_static_diff", "s_mix", "s_mix_obst", "s_o_random", "s_dynamic_formations",
                     "e_n64_k20", "e_n64_k6", "e_n33_k8", "e_n40_kall", "e_n17_kall_obst", "e_n1_obst", "x_ep_len2", "x_svs_odd", "x_n40_obst",
                     "x_n8_blind", "x_domain_random", "x_domain_random_static"]


@pytest.mark.parametrize("case", SINGLE_WAVE_CASES)
def test_rollout_f64_single_wave_kernels(case, monkeypatch):
    """The same rollouts through the one-wave-per-workgroup kernels that large batches select (QS_TEAM=0): streamed observation
    rows, late loads / early stores (qs_step_kernel.inc)."""
    monkeypatch.setenv("QS_TEAM", "0")
    pr = rollout_f64(case, 11 if case.startswith(("c", "s_")) else 3, 45, 1e-8, keep=True)
    assert not pr.hip.team
    pr.close()


@pytest.mark.parametrize("case", SINGLE_WAVE_CASES)
def test_teacher_forced_f32_single_wave_kernels(case, monkeypatch):
    """fp32 production precision through the single-wave kernels (the ones every batch above ~3000 envs runs)."""
    monkeypatch.setenv("QS_TEAM", "0")
    teacher_forced_f32(case, 7, 60, 1e-5, expect_team=False)


def rollout_f64(case, E, steps, tol, keep=False, **over):
    pr = Pair(case, E, "f64", **over)
    rng = np.random.RandomState(5)
    oobs, hobs = pr.reset()
    np.testing.assert_allclose(hobs, oobs, rtol=0, atol=tol)
    if pr.cfg.use_obstacles:
        for e, o in enumerate(pr.oenvs):
            np.testing.assert_allclose(pr.obst_xy(e), np.array(o.info().obst_pos)[:pr.cfg.num_obstacles], atol=1e-12)
    for t in range(steps):
        for e, o in enumerate(pr.oenvs):
            s, tick = o.get_state()
            oxy = pr.obst_xy(e) if pr.cfg.use_obstacles else None
            if force_events(t, e, s, pr.N, oxy, pr.cfg.obst_size / 2):
                o.set_state(s, tick)
                pr.hip.set_state(e, s, tick)
        gentle = (t // 10) % 2 == 1
        act = rng.uniform(-1, 1, size=(E, pr.N, 4)) if not gentle else 0.06 + rng.uniform(-0.05, 0.05, size=(E, pr.N, 4))
        o, h = pr.step(act)
        check_floats(t, tol, o, h, f"{case} f64", pr.obs_layout)
        pr.compare_discrete(t)
        pr.compare_state(t, tol)
        if o[2].any():
            pr.compare_ep_stats(t, 1e-7)
        if case.startswith("s_"):
            sid = pr.hip.to_host("scenario_id")
            for e, oe in enumerate(pr.oenvs):
                assert sid[e] == oe.info().scenario, f"scenario id step {t} env {e}"
    pr.hip.check_errors()
    if pr.cfg.dr_num_density > 0:   # --quads_domain_random: the environments ended up with different obstacle counts, unused slots parked
        cnt = pr.hip.to_host("obst_count")
        assert len(set(cnt.tolist())) > 1 and cnt.max() <= pr.cfg.num_obstacles and cnt.min() >= 1
        for e in range(E):
            xy = pr.obst_xy(e)
            assert (np.abs(xy[:cnt[e]]) < 100).all() and (xy[cnt[e]:] == 1e6).all()
    if keep:
        return pr
    pr.close()


@pytest.mark.parametrize("case", ["c1_single", "c2_n8_dw", "c3_n8_obst", "c4_n32_svs", "c2_n8_k2_numpy_wall",
                                  "s_static_diff", "s_dynamic_formations", "s_lissajous", "s_o_random", "s_mix", "s_mix_obst",
                                  "s_dynamic_diff", "s_bezier", "s_o_swap", "s_o_ep_bezier", "s_o_ep_bezier_short", "s_run_away",
                                  "e_n64_k6", "e_n64_k20", "e_n33_k8", "e_n17_kall_obst", "e_n40_kall",
                                  "x_n8_blind", "x_no_noise", "x_dense_obst", "x_small_room", "x_ep_len2", "x_svs_odd", "x_sim4", "x_hitbox",
                                  "x_n40_obst", "x_domain_random", "x_domain_random_static"])
def test_teacher_forced_f32(case):
    E, steps, tol = (3, LONG[case], 1e-5) if case in LONG else (7, 60, 1e-5)
    teacher_forced_f32(case, E, steps, tol)


def excuse_neighbour_ties(pr, o_obs, h_obs, tol):
    """Neighbour selection (quadrotor_multi.py:247-274) ranks the other drones by a metric; two candidates whose metrics differ by less than float32
    resolves - drones of a formation at EQUAL distances, 16384 of them per step at the full batch sizes - may legitimately swap ranks in float32.
    For every observation row whose neighbour block is outside the tolerance: if each of its K slots holds the relative position / velocity of SOME
    drone of the environment whose float64 metric equals (to 1e-5 * max(1, |m|)) the metric the oracle has at that rank, the row is a tie and is
    replaced by the oracle's (the self / SDF columns stay under the strict check).  Returns the number of rows excused."""
    cfg, N = pr.cfg, pr.N
    self_dim, K = pr.obs_layout
    if K == 0 or K >= N - 1:
        return 0
    blk = slice(self_dim, self_dim + 6 * K)
    bad = np.argwhere((np.abs(h_obs[..., blk] - o_obs[..., blk]) > tol * np.maximum(1.0, np.abs(o_obs[..., blk]))).any(axis=-1))
    excused = 0
    state = {}
    clip_p, clip_v = np.array(list(cfg.nbr_clip_pos)), np.array(list(cfg.nbr_clip_vel))
    for e, i in bad:
        if e not in state:
            st, _ = pr.oenvs[e].get_state()
            state[e] = (st[:, 0:3].copy(), st[:, 3:6].copy())
        pos, vel = state[e]
        rp, rv = pos - pos[i], vel - vel[i]
        rd = np.maximum(np.linalg.norm(rp, axis=1), 0.01)
        m = rd + (rp * rv).sum(axis=1) / rd
        m[i] = np.inf
        order = np.argsort(m, kind="stable")
        rel = np.concatenate([np.clip(rp, -clip_p, clip_p), np.clip(rv, -clip_v, clip_v)], axis=1)     # what a slot holds for drone j
        ok = True
        for k in range(K):
            slot = h_obs[e, i, self_dim + 6 * k:self_dim + 6 * k + 6].astype(np.float64)
            match = np.nonzero((np.abs(rel - slot) <= tol * np.maximum(1.0, np.abs(rel))).all(axis=1))[0]
            match = [j for j in match if j != i]
            want = m[order[k]]
            if not any(abs(m[j] - want) <= 1e-5 * max(1.0, abs(want)) for j in match):
                ok = False
                break
        if ok:
            h_obs[e, i, blk] = o_obs[e, i, blk]
            excused += 1
    return excused


def teacher_forced_f32(case, E, steps, tol, expect_team=None, expect_specialized=False, ties_ok=False, context=None):
    pr = Pair(case, E, "f32")
    if context is not None:
        pr.context = context          # (the name the per-quantity rule and its listed exceptions are looked up under: tests/tolerances.py)
    if expect_team is not None:
        assert bool(pr.hip.team) == expect_team
    if expect_specialized:
        assert pr.hip.specialized, pr.hip.spec_note
    rng = np.random.RandomState(9)
    oobs, hobs = pr.reset()
    tolr.check(f"{pr.context} f32", "obs_reset", hobs, oobs, tolr.allowed_obs(oobs, tol, *pr.obs_layout), "after reset")
    thr = pr.cfg.arm if pr.cfg.floor_mode == 0 else 0.05
    worst, ties = 0.0, 0
    for t in range(steps):
        for e, o in enumerate(pr.oenvs):
            s, tick = o.get_state()
            oxy = pr.obst_xy(e).astype(np.float64) if pr.cfg.use_obstacles else None
            changed = force_events(t, e, s, pr.N, oxy, pr.cfg.obst_size / 2)
            # A drone that lifted off the floor by less than ~1e-6 m sits closer to the `pos_z <= threshold` test
            # (quadrotor_dynamics.py:577) than fp32 resolves (ulp(0.046) = 3.7e-9): which sub-step it lands in is then
            # decided by rounding.  Lift such drones to a representable margin in BOTH states instead of skipping them.
            hover = (s[:, 30] == 0) & (s[:, 2] - thr > 0) & (s[:, 2] - thr < 1e-6)
            if hover.any():
                s[hover, 2] = thr + 1e-4
                changed = True
            # Same for a drone resting exactly on a wall (64-drone formations are wider than the room, so spawn points get clipped
            # onto it): `crashed_wall` is "the clip changed x or y" (quadrotor_dynamics.py:360-367), and while the drone has not
            # tilted yet its drift per sub-step (1e-9 m) is below ulp(5.0) = 4.8e-7 in fp32.  Move it inside by 1 mm in BOTH states.
            for a, half in ((0, pr.cfg.room_hi[0]), (1, pr.cfg.room_hi[1])):
                wall = np.abs(np.abs(s[:, a]) - half) < 1e-5
                if wall.any():
                    s[wall, a] = np.sign(s[wall, a]) * (half - 1e-3)
                    changed = True
            if changed:
                o.set_state(s, tick)
            pr.hip.set_state(e, s, tick)          # teacher forcing: device state <- oracle state (rounded to f32)
        gentle = (t // 10) % 2 == 1
        act = rng.uniform(-1, 1, size=(E, pr.N, 4)) if not gentle else 0.06 + rng.uniform(-0.05, 0.05, size=(E, pr.N, 4))
        act = act.astype(np.float32).astype(np.float64)
        o, h = pr.step(act)
        if ties_ok:
            n_tie = excuse_neighbour_ties(pr, o[0], h[0], tol)
            ties += n_tie
            assert n_tie <= max(2, E * pr.N // 500), f"{case} step {t}: {n_tie} of {E * pr.N} rows differ from the oracle by a neighbour-ranking tie - too many to be ties"
        check_floats(t, tol, o, h, f"{pr.context} f32", pr.obs_layout)
        pr.compare_discrete(t)
        worst = max(worst, pr.compare_state(t, tol))
    print(f"{case}: worst f32 state error {worst:.2e}" + (f"; {ties} of {steps * E * pr.N} observation rows excused as neighbour-ranking ties" if ties_ok else ""))
    pr.hip.check_errors()
    pr.close()


def test_largest_observation_rows():
    """64 drones that all see each other: 396 observation columns.  The float32 stepper stages them in LDS (99 KiB); the float64
    one would need 198 KiB and is refused with a clear error instead of a wrong result."""
    from quad_swarm_rl_amd import native
    kw = dict(num_agents=64, neighbor_visible_num=-1, neighbor_obs_type="pos_vel", use_downwash=True, use_numba=True, collision_falloff_radius=4.0)
    with pytest.raises(native.QsError, match="LDS"):
        native.Stepper(qcfg.make_config(num_envs=2, precision="f64", **kw), device=0)
    st = native.Stepper(qcfg.make_config(num_envs=2, precision="f32", **kw), device=0)
    assert st.obs_dim == 18 + 6 * 63
    st.reset()
    st.from_host("actions", np.zeros((128, 4), dtype=np.float32))
    st.step()
    st.sync()
    obs = st.to_host("obs").reshape(2, 64, -1)
    pos = soa(st.to_host("pos"), 2, 64)
    vel = soa(st.to_host("vel"), 2, 64)
    for d in (0, 17, 63):   # K = N-1: all the others in index order, relative position / velocity (clipped at the room / 2 v_max)
        others = [j for j in range(64) if j != d]
        want = np.concatenate([np.clip(pos[0][others] - pos[0][d], -10, 10), np.clip(vel[0][others] - vel[0][d], -6, 6)], axis=1).reshape(-1)
        np.testing.assert_allclose(obs[0, d, 18:], want, atol=1e-5)
    st.close()


def test_c_abi_error_paths():
    import ctypes as C
    from quad_swarm_rl_amd import native
    L = native.lib()
    cfg = qcfg.make_config(num_envs=2, num_agents=4)
    cfg.num_agents = 100
    h = C.c_void_p()
    assert L.qs_create(C.byref(cfg), 0, C.byref(h)) == -1
    assert b"num_agents" in L.qs_last_error()
    cfg = qcfg.make_config(num_envs=2, num_agents=4)
    assert L.qs_create(C.byref(cfg), 99, C.byref(h)) == -2


def test_determinism_and_sharding_invariance():
    """Same seed twice => bit-identical; env shards with env_id_offset reproduce the un-sharded run."""
    from quad_swarm_rl_amd import native
    kw = dict(CASES["c2_n8_dw"])
    E = 16
    rng = np.random.RandomState(3)
    acts = rng.uniform(-1, 1, size=(20, E, 8, 4)).astype(np.float32)

    def run(num_envs, offset, sl):
        st = native.Stepper(qcfg.make_config(num_envs=num_envs, seed=77, env_id_offset=offset, **kw))
        st.reset()
        out = [st.to_host("obs").copy()]
        for t in range(acts.shape[0]):
            st.from_host("actions", acts[t, sl].reshape(-1, 4))
            st.step()
            out.append(st.to_host("obs").copy())
            out.append(st.to_host("reward").copy())
        st.close()
        return out

    full1, full2 = run(E, 0, slice(0, E)), run(E, 0, slice(0, E))
    for a, b in zip(full1, full2):
        np.testing.assert_array_equal(a, b)
    lo, hi = run(E // 2, 0, slice(0, E // 2)), run(E // 2, E // 2, slice(E // 2, E))
    for a, b, c in zip(full1, lo, hi):
        np.testing.assert_array_equal(a, np.concatenate([b, c], axis=0))


@pytest.mark.parametrize("case,E", [("c2_n8_dw", 1024), ("c3_n8_obst", 1024), ("c4_n32_svs", 512)])
def test_full_size_against_the_oracle(case, E):
    """BASELINE configs[1] / [2] and the per-GPU shard of configs[3] at their FULL batch sizes against the oracle (round 4 compared at most 11
    environments with it and checked the full sizes through oracle-free properties only): the float64 stepper free-running for 36 control steps -
    across the auto-reset of 0.3-s episodes, with the crafted collision / wall / ceiling / floor events of the small-batch test - every
    environment's observations, rewards and state within the per-quantity 1e-8, done / tick / flags / pair masks / unique-id sets / obstacle and
    room masks / counters / obstacle-hit indices / episode statistics exact, for all E environments (8192 / 8192 / 16384 drones)."""
    rollout_f64(case, E, 36, 1e-8, ep_time=0.3)


@pytest.mark.parametrize("case,E", [("c2_n8_dw", 1024), ("c3_n8_obst", 1024), ("c4_n32_svs", 512)])
def test_full_size_f32_production_objects_against_the_oracle(case, E):
    """The float32 PRODUCTION objects (config-specialised team kernels, fast-math, Philox) at the full batch sizes of BASELINE configs[1] / [2]
    and of configs[3]'s per-GPU shard, teacher-forced from the oracle before every one of 34 control steps (round 5 met the oracle with these
    objects at 7 environments only): EVERY environment's observations, rewards, reward terms and post-step state inside the per-quantity 1e-5
    of tests/tolerances.py, done / tick / flags / masks / counters / obstacle-hit indices exact - 8192 / 8192 / 16384 drones per step, the
    crafted collision, `.any()`-quirk, wall / ceiling / floor and obstacle events included."""
    teacher_forced_f32(case, E, 34, 1e-5, expect_team=True, expect_specialized=True, ties_ok=True, context=f"{case}@full")


@pytest.mark.parametrize("case,E", [("c2_n8_dw", 1024), ("c3_n8_obst", 1024), ("c4_n32_svs", 512)])
def test_full_size_properties(case, E):
    """BASELINE configs[1] / [2] and the per-GPU shard of configs[3] at full size (8 x 1024, 8 x 1024 with obstacles, 32 x 512 on the
    4-wave pair-once team kernel; fp32 production kernels): properties that do not need the oracle - sharding invariance (two half-size
    handles == one full-size handle, bit for bit), finite outputs, mask / counter consistency, obstacle-hit indices in range, auto-reset
    cadence."""
    from quad_swarm_rl_amd import native
    kw = dict(CASES[case], ep_time=0.3)
    N, steps = kw["num_agents"], 40
    rng = np.random.RandomState(11)
    acts = rng.uniform(-1, 1, size=(steps, E * N, 4)).astype(np.float32)

    def run(num_envs, offset):
        st = native.Stepper(qcfg.make_config(num_envs=num_envs, seed=5, env_id_offset=offset, **kw))
        assert st.specialized or os.environ.get("QS_SPEC", "jit") in ("off", "0", "cache")
        st.reset()
        lo, hi = offset * N, (offset + num_envs) * N
        out = []
        for t in range(steps):
            st.from_host("actions", acts[t, lo:hi])
            st.step()
            out.append((st.to_host("obs").copy(), st.to_host("reward").copy(), st.to_host("done").copy(),
                        st.to_host("unique_col_mask").copy(), st.to_host("counters").copy(), st.to_host("col_pair_mask").copy(),
                        st.to_host("obst_new_mask").copy(), st.to_host("obst_hit_idx").copy(), st.to_host("room_new_mask").copy()))
        st.check_errors()
        M = st.cfg.num_obstacles
        st.close()
        return out, M

    (full, M), (a, _), (b, _) = run(E, 0), run(E // 2, 0), run(E // 2, E // 2)
    for t in range(steps):
        obs, rew, done, uniq, cnt, pairs, obst_new, ohit, room = full[t]
        for k in range(9):
            axis = 1 if k == 4 else 0
            np.testing.assert_array_equal(full[t][k], np.concatenate([a[t][k], b[t][k]], axis=axis), err_msg=f"step {t} output {k}")
        assert np.isfinite(obs).all() and np.isfinite(rew).all()
        assert done.all() == ((t + 1) % 31 == 0) and done.any() == done.all()          # ep_len 30: done on every 31st step, all envs
        assert (uniq >> np.uint64(N) == 0).all() and (obst_new >> np.uint64(N) == 0).all() and (room >> np.uint64(N) == 0).all()   # ids are drone indices < N
        assert ((ohit >= -1) & (ohit < max(M, 1))).all() and (kw.get("use_obstacles") or (ohit == -1).all())        # first-hit obstacle index
        pm = pairs.reshape(E, N)
        for d in range(N):                                                              # pair bits only above the own index
            assert ((pm[:, d] & ((1 << (d + 1)) - 1)) == 0).all()
        if t > 0 and not full[t - 1][2].any() and not done.any():
            assert (cnt >= full[t - 1][4]).all()                                         # counters only grow inside an episode
            grew = cnt[0] - full[t - 1][4][0]
            pop = np.array([bin(int(u)).count("1") // 2 for u in uniq])
            np.testing.assert_array_equal(grew, pop)                                     # collisions += len(unique ids) // 2
