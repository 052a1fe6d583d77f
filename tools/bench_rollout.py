#!/usr/bin/env python
"""Closed sampling loop on C2 (8 drones x 1024 envs): [fused policy encoder -> Gaussian action head -> env step] per control step,
eager (one Python iteration per step) vs captured into a HIP graph of T steps.  Prints one JSON line."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from quad_swarm_rl_amd import policy, rollout
from quad_swarm_rl_amd.env import QuadSwarmVecEnv

enc_type = sys.argv[1] if len(sys.argv) > 1 else "attention"
T = int(sys.argv[2]) if len(sys.argv) > 2 else 32
E = int(sys.argv[3]) if len(sys.argv) > 3 else 1024
env = QuadSwarmVecEnv(E, seed=0, num_agents=8, neighbor_visible_num=6, neighbor_obs_type="pos_vel", use_numba=True, collision_falloff_radius=4.0,
                      write_rew_info=False)
env.reset()
enc = policy.FusedQuadEncoder(policy.make_reference_encoder(seed=0, nbr_encoder=enc_type).cuda())
head = rollout.GaussianActionHead(sample=True)
res = {"workload": f"c2 closed loop: {E} envs x 8 drones, {enc_type} encoder, Gaussian head, {T}-step segments"}
for name, graph in (("eager", False), ("graph", True)):
    seg = rollout.GraphedRollout(env, enc, head, steps=T, graph=graph)
    for _ in range(3):
        seg.run()
    torch.cuda.synchronize()
    reps = 20
    t0 = time.perf_counter()
    for _ in range(reps):
        seg.run()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / (reps * T)
    res[name + "_us_per_control_step"] = dt * 1e6
    res[name + "_env_steps_per_s"] = E * 8 * 2 / dt
print(json.dumps(res))
