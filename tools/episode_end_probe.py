#!/usr/bin/env python
"""What the steps around an episode end cost (1024 envs x 8 drones, all environments finish on the same step): GPU time of each step
kernel by events (synchronised per step), then the host side of BatchedQuadSwarm.step on the episode-end step under cProfile.
usage: python tools/episode_end_probe.py [quads_mode] [num_envs]"""
import argparse
import cProfile
import io
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from quad_swarm_rl_amd import sf_env
from quad_swarm_rl_amd.env import QuadSwarmVecEnv

mode = sys.argv[1] if len(sys.argv) > 1 else "mix"
E = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
kw = dict(num_agents=8, neighbor_visible_num=6, neighbor_obs_type="pos_vel", use_numba=True, collision_hitbox_radius=2.0,
          collision_falloff_radius=4.0, use_downwash=True, quads_mode=mode, ep_time=15.0, episode_sums=True, write_rew_info=False)
env = QuadSwarmVecEnv(E, seed=0, **kw)
env.reset()
act = (torch.rand((env.num_agents, 4), device="cuda") * 2 - 1) * 0.2 + 0.1
ep = env.cfg.ep_len + 1
for _ in range(ep - 6):
    env.step(act)
torch.cuda.synchronize()
print(f"{mode}: GPU time of the step kernel around the episode end (step index within the episode: us)")
for k in range(ep - 6, ep + 6):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    env.step(act)
    e1.record()
    torch.cuda.synchronize()
    print(f"  step {k + 1:5d}{'  <- episode end (auto-reset of every env)' if k + 1 == ep else ''}: {e0.elapsed_time(e1) * 1e3:10.1f}")
env.close()

p = argparse.ArgumentParser()
p.add_argument("--with_pbt", default=False)
sf_env.add_quadrotors_env_args("quadrotor_multi", p)
cfg = p.parse_args(["--quads_use_numba=True", f"--quads_mode={mode}", "--quads_episode_duration=15.0", "--quads_neighbor_obs_type=pos_vel",
                    "--quads_collision_hitbox_radius=2.0", "--quads_collision_falloff_radius=4.0", "--quads_collision_reward=5.0",
                    "--quads_collision_smooth_max_penalty=10.0", "--quads_neighbor_visible_num=6", "--quads_use_downwash=True",
                    f"--quads_num_envs={E}", "--anneal_collision_steps=300000000"])
b = sf_env.make_quadrotor_env("quadrotor_multi", cfg=cfg)
b.reset()
act = (torch.rand((b.num_agents, 4), device="cuda") * 2 - 1) * 0.2 + 0.1
for _ in range(ep - 1):
    b.step(act)
torch.cuda.synchronize()
pr = cProfile.Profile()
t0 = time.perf_counter()
pr.enable()
out = b.step(act)
infos = out[-1]
n = len(infos)
first = infos[0] if n else None
pr.disable()
t1 = time.perf_counter()
allv = [infos[i] for i in range(n)]
t2 = time.perf_counter()
print(f"BatchedQuadSwarm.step on the episode-end step (GPU idle before it): {1e3 * (t1 - t0):.2f} ms incl. len(infos) = {n} and infos[0]; reading all infos: {1e3 * (t2 - t1):.2f} ms")
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(14)
print(s.getvalue()[:3000])
b.close()
