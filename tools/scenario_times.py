#!/usr/bin/env python
"""Step-kernel time per scenario (`--quads_mode`), 1024 envs x 8 drones with train_local.sh's physics flags: HIP events around K
steps after a warm-up that takes the envs past their first goal changes.  One line per scenario + one JSON line.
usage: python tools/scenario_times.py [num_envs] [steps] [scenario ...]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from quad_swarm_rl_amd import config as qcfg
from quad_swarm_rl_amd.env import QuadSwarmVecEnv

E = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 1200
names = sys.argv[3:] or [n for n in qcfg.SCENARIOS]
res = {}
for name in names:
    obst = name.startswith("o_")
    kw = dict(num_agents=8, neighbor_visible_num=6, neighbor_obs_type="pos_vel", use_numba=True, collision_hitbox_radius=2.0,
              collision_falloff_radius=4.0, use_downwash=True, quads_mode=name, ep_time=15.0, episode_sums=True, write_rew_info=False)
    if obst:
        kw.update(use_obstacles=True, obst_density=0.2, obst_size=0.6, obst_spawn_area=(8.0, 8.0))
    try:
        env = QuadSwarmVecEnv(E, seed=0, **kw)
    except Exception as ex:   # a scenario the flag set does not allow
        print(f"{name:24s} skipped: {ex}")
        continue
    env.reset()
    act = (torch.rand((env.num_agents, 4), device="cuda") * 2 - 1) * 0.2 + 0.1
    for _ in range(700):
        env.step(act)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(steps):
        env.step(act)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / steps
    res[name] = round(us, 2)
    print(f"{name:24s} {us:8.2f} us per step   {getattr(env.stepper, 'spec_note', '')}", flush=True)
    env.close()
print(json.dumps({"workload": f"{E} envs x 8 drones, step kernel us per step by scenario", "us_per_step": res}))
