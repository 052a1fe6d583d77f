#!/usr/bin/env python
"""Host-side rate of the Sample Factory-facing batched env (sf_env.BatchedQuadSwarm.step, train_local.sh's flag set): one Python call
per control step, device tensors in and out.  Prints one JSON line: us per step and agent-steps/s with / without replay."""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from quad_swarm_rl_amd import sf_env

RECIPE = ["--quads_use_numba=True", "--anneal_collision_steps=300000000", "--quads_mode=mix", "--quads_episode_duration=15.0",
          "--quads_neighbor_obs_type=pos_vel", "--quads_collision_hitbox_radius=2.0", "--quads_collision_falloff_radius=4.0",
          "--quads_collision_reward=5.0", "--quads_collision_smooth_max_penalty=10.0", "--quads_neighbor_encoder_type=attention",
          "--quads_neighbor_visible_num=6", "--quads_use_downwash=True"]
E = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 4000
res = {"workload": f"BatchedQuadSwarm.step, {E} envs x 8 drones, mix scenario, train_local.sh flags"}
for name, prob in (("no_replay", 0.0), ("replay_0.75", 0.75)):
    p = argparse.ArgumentParser()
    p.add_argument("--with_pbt", default=False)
    sf_env.add_quadrotors_env_args("quadrotor_multi", p)
    cfg = p.parse_args(RECIPE + [f"--quads_num_envs={E}", f"--replay_buffer_sample_prob={prob}"])
    env = sf_env.make_quadrotor_env("quadrotor_multi", cfg=cfg)
    env.reset()
    act = (torch.rand((env.num_agents, 4), device="cuda") * 2 - 1) * 0.2 + 0.1
    for _ in range(200):
        env.step(act)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n_info, t_info = 0, 0.0
    for _ in range(steps):
        t1 = time.perf_counter()
        _, _, _, _, infos = env.step(act)
        if infos:
            n_info += 1
            t_info += time.perf_counter() - t1
    torch.cuda.synchronize()
    total = time.perf_counter() - t0
    dt = total / steps
    res[name] = {"us_per_step": dt * 1e6, "agent_steps_per_s": env.num_agents / dt, "env_steps_per_s (x2 sim steps)": env.num_agents * 2 / dt, "steps_with_infos": n_info,
                 "ms_per_step_with_infos": 1e3 * t_info / max(n_info, 1), "us_per_step_without_infos": 1e6 * (total - t_info) / max(steps - n_info, 1)}
    if name == "no_replay":   # where a step() call's time goes, layer by layer (same env, same kernel): the C ABI through ctypes alone, the vec env's
        # step(), the batched env's step(); and the kernel's own duration from HIP events (what the GPU needs per step whatever the host does)
        from quad_swarm_rl_amd import native
        vec, st = env.vec, env.vec.stepper
        qs_step, h, ptr = native.lib().qs_step, st._h, act.data_ptr()
        raw = torch._C._cuda_getCurrentRawStream
        layers = {}
        for lname, fn in (("ctypes_qs_step", lambda: qs_step(h, ptr, raw(0))), ("vec_env_step", lambda: vec.step(act)), ("batched_env_step", lambda: env.step(act))):
            for _ in range(200):
                fn()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(2000):
                fn()
            torch.cuda.synchronize()
            layers[lname] = 1e6 * (time.perf_counter() - t0) / 2000
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
        for _ in range(2000):
            qs_step(h, ptr, raw(0))
        ev1.record()
        torch.cuda.synchronize()
        layers["gpu_us_per_step_by_events"] = 1e3 * ev0.elapsed_time(ev1) / 2000
        layers["kernel"] = st.kernel_name
        res["layers_us_per_step"] = layers
    env.close()
print(json.dumps(res))
