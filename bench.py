#!/usr/bin/env python
"""bench.py - env-steps/s of the HIP QuadSwarm stepper on BASELINE.json's metric.

  python bench.py --gpus N --steps K --warmup W
  (N>1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

A "step" is one control step (= 2 physics sub-steps) of all environments of the workload on synthetic
U(-1,1)^4 actions that are already resident in HBM.  Workload at N=1 (BASELINE.json configs[1], "C2"):
8 drones x 1024 envs, static_same_goal, 6 visible neighbours (obs 54), downwash on, Numba-path semantics,
sensor + thrust noise on, auto-resets included.  Each extra GPU gets its own 1024-env shard (weak scaling); the shards
are independent (no cross-env term on this path), so the timed region has no data-path collective.  The optional
variant of north_star / SURVEY.md 8e - ONE RCCL all-gather of the observations after every step - is timed right after
and reported as config.with_obs_allgather (or becomes the headline with --gather): it moves 12.4 MB per GPU per step and
is xGMI-link-bound at >= ~28 us per step whatever the implementation.

metric:  env-steps/s = drones x envs x sim_steps(2) x control-steps/s   (BASELINE.md "Metric")
roofline: HBM-bound; algorithmic bytes per drone-control-step = 500 B (read state 120 + flags 4 +
          goal 12 + action 16, write state 120 + flags 4 + obs 216 + reward 4 + done 4); achieved = 500 B x drones
          per launch / average step-kernel duration, measured here with HIP events on the launch stream.
cpu_baseline: the validated C oracle (oracle/, OpenMP over envs) on this box's host cores, rank 0, N=1 only.
"""
import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

import numpy as np  # noqa: E402

ALGO_BYTES_PER_DRONE_STEP = {"c2": 500, "c3": 456, "c4": 500, "c1": 356}
HBM_PEAK_GBS = 8000.0   # /opt/skills/guides/MI355X_MICROARCH.md: 8 TB/s spec (6.3 TB/s achievable)

WORKLOADS = {
    "c2": dict(num_envs=1024, kw=dict(num_agents=8, neighbor_visible_num=6, neighbor_obs_type="pos_vel", use_downwash=True,
                                      use_numba=True, collision_hitbox_radius=2.0, collision_falloff_radius=4.0,
                                      quads_mode="static_same_goal",
                                      rew_coeff=dict(quadcol_bin=5.0, quadcol_bin_smooth_max=10.0))),
    "c3": dict(num_envs=1024, kw=dict(num_agents=8, neighbor_visible_num=2, neighbor_obs_type="pos_vel", use_downwash=True,
                                      use_numba=True, collision_falloff_radius=4.0, use_obstacles=True, obst_density=0.2,
                                      obst_size=0.6, obst_spawn_area=(8.0, 8.0), quads_mode="o_static_same_goal",
                                      obs_repr="xyz_vxyz_R_omega_floor",
                                      rew_coeff=dict(quadcol_bin=5.0, quadcol_bin_smooth_max=4.0, quadcol_bin_obst=5.0))),
    "c4": dict(num_envs=512, kw=dict(num_agents=32, neighbor_visible_num=6, neighbor_obs_type="pos_vel", use_downwash=True,
                                     use_numba=True, collision_falloff_radius=4.0, quads_mode="swarm_vs_swarm",
                                     rew_coeff=dict(quadcol_bin=5.0, quadcol_bin_smooth_max=10.0))),
    "c1": dict(num_envs=1, kw=dict(num_agents=1, neighbor_visible_num=0, neighbor_obs_type="none", use_numba=False)),
}


def cpu_baseline(workload, seconds):
    """Times the oracle (kind 'port': the validated C restatement of the reference) on the host cores.  Each env runs its
    control steps back to back inside one OpenMP region (envs are independent), float64, all hardware threads."""
    from oracle import oracle as orc
    from quad_swarm_rl_amd import config as qcfg
    w = WORKLOADS[workload]
    threads = os.cpu_count() or 1
    num_envs = w["num_envs"]
    cfg = qcfg.make_config(num_envs=num_envs, seed=0, **w["kw"])
    batch = orc.OracleBatch(cfg, num_envs)
    batch.reset()
    n = cfg.num_agents
    rng = np.random.RandomState(0)
    acts = rng.uniform(-1, 1, size=(16, num_envs, n, 4))
    chunk = 50
    batch.rollout(acts, chunk)                     # warm-up (thread pool, caches)
    steps, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        batch.rollout(acts, chunk)
        steps += chunk
    dt = time.perf_counter() - t0
    return dict(value=num_envs * n * 2 * steps / dt, unit="env-steps/s", cores=threads, kind="port",
                sample=f"{num_envs} envs x {n} drones x {steps} control steps ({dt:.1f} s), C oracle (float64), one OpenMP region, "
                       f"{threads} threads")


def pmc_traffic(workload, num_envs, kernel):
    """HBM bytes per launch of the step kernel from the committed rocprofv3 PMC pass (profiles/r01_pmc_traffic.json, produced by
    tools_pmc.sh: FETCH_SIZE and WRITE_SIZE in separate passes, KiB units and gfx950 corrections of MI355X_MICROARCH.md);
    None when no measurement of this workload / batch / kernel is on file."""
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r01_pmc_traffic.json")
    try:
        with open(path) as f:
            rec = json.load(f).get(f"{workload}:{num_envs}:{kernel}")
        return rec["fetch_bytes"] + rec["write_bytes"] if rec else None
    except (OSError, ValueError, KeyError):
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--workload", default="c2", choices=sorted(WORKLOADS))
    ap.add_argument("--envs-per-gpu", type=int, default=0, help="override the workload's env count per GPU")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="CPU baseline sample length (0 = skip)")
    ap.add_argument("--profile-steps", type=int, default=400, help="steps of the HIP-event kernel-duration pass")
    ap.add_argument("--gather", action="store_true", help="N>1: put ONE RCCL all-gather of the observations after every step inside the timed region "
                                                          "(default: shards step independently; the gather variant is measured separately and reported in config)")
    ap.add_argument("--force-gather", action="store_true", help="run the RCCL obs all-gather path even at N=1 (exercises the multi-GPU code on a 1-GPU box)")
    ap.add_argument("--no-gather", action="store_true", help="skip the separate all-gather measurement at N>1")
    ap.add_argument("--no-overlap", action="store_true", help="N>1: gather on the compute stream instead of overlapping it with the next step")
    ap.add_argument("--rew-info", action="store_true", help="also write the 17-term reward-info matrix every step (logging output)")
    ap.add_argument("--set", action="append", default=[], metavar="KEY=VALUE", help="override a workload keyword (python literal)")
    ap.add_argument("--graph", type=int, default=0, help="headline mode: step the timed region as open-loop rollouts of this many steps per "
                                                          "launch (qs_step_many: state stays in registers between the steps)")
    ap.add_argument("--rollout-steps", type=int, default=64, help="steps per launch of the extra open-loop measurement (0 = skip)")
    args = ap.parse_args()

    import torch
    from quad_swarm_rl_amd import config as qcfg, native

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP stepper has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1 or args.force_gather:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    import ast
    w = WORKLOADS[args.workload]
    kw = dict(w["kw"])
    for item in args.set:
        key, val = item.split("=", 1)
        kw[key] = ast.literal_eval(val)
    E = args.envs_per_gpu or w["num_envs"]
    cfg = qcfg.make_config(num_envs=E, seed=0, env_id_offset=rank * E, precision="f32", write_rew_info=args.rew_info, **kw)
    st = native.Stepper(cfg, device=local_rank)
    N, T, D = cfg.num_agents, E * cfg.num_agents, st.obs_dim
    stream = torch.cuda.current_stream(local_rank)

    # synthetic actions, resident in HBM before the timed region: a ring of pre-drawn U(-1,1)^4 batches
    gen = torch.Generator(device=f"cuda:{local_rank}")
    gen.manual_seed(1234 + rank)
    ring = 64
    actions = (torch.rand((ring, T, 4), device=f"cuda:{local_rank}", generator=gen, dtype=torch.float32) * 2.0 - 1.0).contiguous()
    aptr, astride = actions.data_ptr(), T * 4 * 4
    obs = st.tensor("obs")
    gather = gather_variant = None
    if (world > 1 or args.force_gather) and not args.no_gather:
        from quad_swarm_rl_amd import parallel
        gather_variant = parallel.ObsGather(obs, overlap=not args.no_overlap)   # ONE RCCL all-gather of the obs per rollout step
        if args.gather or args.force_gather:
            gather = gather_variant

    def run(k, offset=0, gather=gather):
        if args.graph > 0 and gather is None:
            # open-loop rollout over the action ring: qs_step_many keeps the state in registers across the steps of a launch
            g = min(args.graph, ring)
            done_steps = 0
            while done_steps + g <= k:
                st.step_many(aptr, g, stream=stream)
                done_steps += g
            for t in range(k - done_steps):
                st.step(aptr + (t % ring) * astride, stream=stream)
            return
        for t in range(k):
            st.step(aptr + ((offset + t) % ring) * astride, stream=stream)
            if gather is not None:
                gather.gather()
        if gather is not None:
            gather.drain()

    st.reset(stream=stream)
    run(args.warmup)
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    # HIP events on the launch stream (= torch's current stream) bracket the timed region: device-side duration of the K
    # back-to-back step-kernel launches, i.e. the average launch-to-launch duration rocprofv3 --kernel-trace also reports
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record(stream)
    run(args.steps, args.warmup)
    ev1.record(stream)
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    region_kernel_ms = ev0.elapsed_time(ev1) / args.steps
    if dist is not None:
        tmax = torch.tensor([elapsed], device=f"cuda:{local_rank}", dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    st.check_errors()

    # N>1: the optional collective variant (north_star: one RCCL all-gather of the observations per rollout step), timed
    # separately with the same bracketing; `value` above is the independent-shard rate unless --gather was given
    with_gather = None
    if gather_variant is not None and gather is None:
        gk = min(args.steps, 1000)
        run(50, 0, gather_variant)
        dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run(gk, 50, gather_variant)
        dist.barrier()
        torch.cuda.synchronize()
        gdt = torch.tensor([time.perf_counter() - t0], device=f"cuda:{local_rank}", dtype=torch.float64)
        dist.all_reduce(gdt, op=dist.ReduceOp.MAX)
        with_gather = {"value": world * T * 2 * gk / float(gdt.item()), "unit": "env-steps/s", "ms_per_step": 1e3 * float(gdt.item()) / gk, "steps": gk,
                       "collective": "rccl all_gather_into_tensor of the obs per step" + ("" if args.no_overlap else ", overlapped with the next step")}

    # extra: the same workload as open-loop rollouts (pre-generated actions, K control steps per launch)
    rollout = None
    if world == 1 and args.rollout_steps > 0 and args.graph == 0:
        kk = min(args.rollout_steps, ring)
        reps = max(1, args.steps // kk)
        st.step_many(aptr, kk, stream=stream)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            st.step_many(aptr, kk, stream=stream)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        rollout = {"steps_per_launch": kk, "value": T * 2 * reps * kk / dt, "unit": "env-steps/s", "us_per_step": 1e6 * dt / (reps * kk)}

    # second view of the same kernel: a HIP event pair around every single launch (includes the event packets themselves)
    st.set_profiling(True)
    for t in range(args.profile_steps):
        st.step(aptr + (t % ring) * astride, stream=stream)
    kernel_ms, launches = st.kernel_time()
    st.set_profiling(False)

    if rank == 0:
        value = world * T * 2 * args.steps / elapsed
        algo = ALGO_BYTES_PER_DRONE_STEP[args.workload]
        achieved = algo * T / (region_kernel_ms * 1e-3) / 1e9 if region_kernel_ms > 0 else 0.0
        out = {
            "metric": "env-steps/s (drones x envs x sim_steps)", "value": value, "unit": "env-steps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{args.workload}: {N} drones x {E} envs per GPU, {WORKLOADS[args.workload]['kw'].get('quads_mode', 'static_same_goal')}, "
                                   f"K={cfg.num_neighbors} neighbours, obs_dim {D}, downwash {bool(cfg.use_downwash)}, sensor+thrust noise on, auto-reset on",
                       "drone_control_steps_per_s": value / 2.0, "envs_per_gpu": E, "num_agents": N,
                       "obs_gather": ("rccl all_gather_into_tensor per step" + ("" if args.no_overlap else ", overlapped with the next step")) if gather is not None
                                     else ("none: env shards are independent, no data-path collective (DESIGN.md 7)" if world > 1 else "none"),
                       "with_obs_allgather": with_gather,
                       "launch": f"open-loop rollout, {min(args.graph, ring)} steps per launch" if args.graph > 0 and world == 1 else "one launch per control step",
                       "open_loop_rollout": rollout, "rew_info": bool(args.rew_info),
                       "overrides": args.set},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": pmc_traffic(args.workload, E, st.kernel_name), "kernel": st.kernel_name,
                         "kernel_flavor": ("config-specialised, " if st.specialized else "generic, ") + f"{st.waves_per_workgroup} wave{'s' if st.waves_per_workgroup > 1 else ''} per workgroup", "kernel_avg_us": region_kernel_ms * 1e3, "kernel_launches": args.steps,
                         "kernel_avg_us_event_pair_per_launch": kernel_ms * 1e3, "event_pair_launches": launches,
                         "algorithmic_bytes_per_launch": algo * T},
        }
        if world == 1 and args.cpu_seconds > 0:
            out["cpu_baseline"] = cpu_baseline(args.workload, args.cpu_seconds)
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out))
    st.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
