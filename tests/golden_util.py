"""Helpers shared by the parity tests: load a golden fixture and build the matching qs_config."""
import json
import os

import numpy as np

from quad_swarm_rl_amd import config as qcfg

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    g = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
    cfg = json.loads(str(g["cfg"]))
    return g, cfg


def config_from_golden(cfg, **over):
    kw = dict(
        num_agents=cfg["num_agents"], ep_time=cfg["ep_time"], rew_coeff={k: v for k, v in cfg["rew_coeff"].items()
                                                                          if k in qcfg.REW_COEFF_DEFAULT},
        obs_repr=cfg["obs_repr"], neighbor_visible_num=cfg["neighbor_visible_num"],
        neighbor_obs_type=cfg["neighbor_obs_type"], collision_hitbox_radius=cfg["collision_hitbox_radius"],
        collision_falloff_radius=cfg["collision_falloff_radius"], use_obstacles=cfg["use_obstacles"],
        obst_density=cfg["obst_density"], obst_size=cfg["obst_size"], obst_spawn_area=cfg["obst_spawn_area"],
        use_downwash=cfg["use_downwash"], use_numba=cfg["use_numba"], quads_mode=cfg["quads_mode"],
        room_dims=cfg["room_dims"], sense_noise=cfg["sense_noise"], thrust_noise_ratio=cfg["thrust_noise_ratio"],
        # fixtures captured under the plain numba stub ran OUNoiseNumba with float64 members; the `*_f32ou` ones with numba's float32 members
        numba_float32_ou=bool(cfg.get("numba_float32_ou", False)),
    )
    kw.update(over)
    return qcfg.make_config(**kw)


def state_from_golden(g, prefix, t=None):
    """[N, QS_STATE_STRIDE] state matrix from the snapshot arrays (prefix 's0_' or 's_' with step t)."""
    def get(k):
        a = g[prefix + k]
        return a if t is None else a[t]
    n = get("pos").shape[0]
    s = np.zeros((n, qcfg.QS_STATE_STRIDE))
    s[:, 0:3] = get("pos"); s[:, 3:6] = get("vel"); s[:, 6:15] = get("rot").reshape(n, 9); s[:, 15:18] = get("omega")
    s[:, 18:22] = get("thrust_rot_damp"); s[:, 22:26] = get("thrust_cmds_damp"); s[:, 26:30] = get("ou_state")
    s[:, 30] = get("on_floor")
    s[:, 31] = np.round(get("since_last_svd") / 0.005)
    s[:, 32:35] = get("goal")
    return s
