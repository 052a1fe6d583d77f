#!/usr/bin/env python
"""bench.py - env-steps/s of the HIP QuadSwarm stepper on BASELINE.json's metric.

  python bench.py --gpus N --steps K --warmup W
  N>1: started directly, bench.py launches its own N ranks (python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr
  127.0.0.1 ... bench.py <same arguments>); started by that launcher (WORLD_SIZE set) it is one rank.  --dry-run: the launch path alone
  (gloo rendezvous + barrier bracket, no GPU; tests/test_bench_launch.py).

A "step" is one control step (= 2 physics sub-steps) of all environments of the workload on synthetic
U(-1,1)^4 actions that are already resident in HBM.

N = 1 (BASELINE.json configs[1], "C2", the configuration the metric is quoted on): 8 drones x 1024 envs,
static_same_goal, 6 visible neighbours (obs 54), downwash on, Numba-path semantics, sensor + thrust noise on,
auto-resets included.
N > 1 (BASELINE.json configs[3], "C4"): 32 drones x 512 envs PER GPU (4096 envs at 8 GPUs), swarm_vs_swarm, and the exchange
of the observation rows after every step INSIDE the timed region (north_star: "a single ... gather of observations over xGMI
per rollout step"): every rank ends each step with the rows of all ranks.  [step -> exchange] x 64 is ONE captured HIP graph per
rank (quad-swarm-rl_amd/parallel.py: ObsExchange); exchange(t) runs on a second stream under step(t+1).  The HEADLINE wire is f32 - 216-byte rows,
bit-exact: every rank holds the reference's float observations of all ranks; the lossy wires are measured on the same shards with the same bracketing
and reported beside it as labelled variants (config.exchange_per_wire: bf16 = 108-byte rows, RNE; q8 = 72-byte rows, bf16 self columns + 8-bit
fixed-point neighbour block, error <= 0.039 m / 0.024 m/s, outside the 1e-5 observation tolerance).  --wire picks another headline wire; the line
names it in its top-level `wire` field.  --transport fused: the step kernel itself stores its rows into every rank's hipIpc-mapped
receive window (include/quadswarm_exchange.h; no launch besides the step), peer: the same windows filled by a push kernel on a second
stream, rccl: RCCL all-gather of the packed rows (stepped eagerly: torch's collective is not recorded into a graph), auto (default): fused (peer for batches that run the single-wave kernels) if every
rank could map its peers' windows, passed the start-up self-check and holds the same rows an RCCL gather delivers (ObsExchange.verify), else rccl; torch: round 2's eager per-step all_gather.  The rate of the
same shards stepping with no exchange is measured right after and reported as config.secondary (--no-gather makes it the
headline; --workload / --envs-per-gpu override the shape).

Timing.  (A scratch handle of the same configuration runs --prewarm steps first: device clocks, code caches.)  W warm-up steps, barrier
+ synchronize, K timed steps, barrier + synchronize.  `value` / `ms_per_step` come from
HIP events recorded on the launch stream right after the opening synchronize and right after the K-th step (for the
gather variant: after the last gather has drained into the stream), MAX over ranks: at K = 20 the timed region is a few
hundred microseconds and the closing host synchronize alone is worth several steps, so the host clock would time the
synchronize, not the steps.  The host-clock figures of the same region are printed beside them (config.host_clock).

metric:  env-steps/s = drones x envs x sim_steps(2) x control-steps/s   (BASELINE.md "Metric")
roofline: HBM-bound; algorithmic bytes per drone-control-step = 500 B (read state 120 + flags 4 +
          goal 12 + action 16, write state 120 + flags 4 + obs 216 + reward 4 + done 4); achieved = 500 B x drones
          per launch / average step-kernel duration, measured here with HIP events on the launch stream.
cpu_baseline: the validated C oracle (oracle/, float64, OpenMP over envs) on this box's host cores, rank 0, N=1 only:
          a sweep over thread counts {1, physical cores, usable hardware threads}; value = the best, single_thread beside it.
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
ALGO_BYTES_F64_PER_DRONE_STEP = {"c2": 988, "c3": 900, "c4": 988, "c1": 700}   # the same arrays with 8-byte reals (flags / done stay 4 / 4 bytes)
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


def _host_cpu():
    """(model string, usable hardware threads, physical cores among them, cgroup CPU quota or None)."""
    usable = sorted(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else list(range(os.cpu_count() or 1))
    model, cores = "unknown", set()
    try:
        with open("/proc/cpuinfo") as f:
            phys = core = proc = None
            for line in f:
                if line.startswith("processor"):
                    proc = int(line.split(":")[1])
                elif line.startswith("model name") and model == "unknown":
                    model = line.split(":", 1)[1].strip()
                elif line.startswith("physical id"):
                    phys = line.split(":")[1].strip()
                elif line.startswith("core id"):
                    core = line.split(":")[1].strip()
                elif not line.strip():
                    if proc in usable and core is not None:
                        cores.add((phys, core))
                    phys = core = proc = None
    except OSError:
        pass
    quota = None
    try:   # cgroup v2 CPU bandwidth limit ("max" or "<quota> <period>")
        with open("/sys/fs/cgroup/cpu.max") as f:
            q, per = f.read().split()
            if q != "max":
                quota = float(q) / float(per)
    except (OSError, ValueError):
        pass
    return model, len(usable), (len(cores) or len(usable)), quota


def cpu_baseline(workload, seconds):
    """Times the oracle (kind 'port': the validated C restatement of the reference, float64) on the host cores.  Each env runs
    its control steps back to back inside one OpenMP region (envs are independent).  Sweep over thread counts {1, physical
    cores, usable hardware threads}: `value` is the best of them, `single_thread` the 1-thread rate."""
    from oracle import oracle as orc
    from quad_swarm_rl_amd import config as qcfg
    w = WORKLOADS[workload]
    model, usable, physical, quota = _host_cpu()
    num_envs = w["num_envs"]
    cfg = qcfg.make_config(num_envs=num_envs, seed=0, **w["kw"])
    batch = orc.OracleBatch(cfg, num_envs)
    batch.reset()
    n = cfg.num_agents
    rng = np.random.RandomState(0)
    acts = rng.uniform(-1, 1, size=(16, num_envs, n, 4))
    counts = {1, physical, usable}
    if quota:
        counts.add(max(1, min(usable, int(round(quota)))))
    if usable > 64:
        counts.add(32)   # one point in between: a box whose cgroup gives this process fewer cores than it shows
    counts = sorted(counts)
    per = max(seconds / len(counts), 1.0)
    sweep, total_steps, total_dt = [], 0, 0.0
    for th in counts:
        batch.rollout(acts, 2, th)                      # warm-up (thread pool, caches)
        t0 = time.perf_counter()
        batch.rollout(acts, 4, th)
        chunk = int(max(4, min(400, 4 * 0.25 / max(time.perf_counter() - t0, 1e-4))))   # ~0.25 s of work per call
        steps, t0 = 0, time.perf_counter()
        while time.perf_counter() - t0 < per:
            batch.rollout(acts, chunk, th)
            steps += chunk
        dt = time.perf_counter() - t0
        sweep.append({"threads": th, "value": num_envs * n * 2 * steps / dt, "control_steps": steps, "seconds": dt})
        total_steps += steps
        total_dt += dt
    best = max(sweep, key=lambda r: r["value"])
    single = sweep[0]["value"]
    return dict(value=best["value"], unit="env-steps/s", cores=best["threads"], kind="port",
                sample=f"{num_envs} envs x {n} drones, C oracle (float64), one OpenMP region over envs; thread sweep {counts} x ~{per:.0f} s each "
                       f"({total_steps} control steps, {total_dt:.1f} s in total); value = best of the sweep",
                single_thread=single, drone_control_steps_per_s_per_thread=single / 2.0, sweep=sweep,
                cpu_model=model, usable_hw_threads=usable, physical_cores=physical, cgroup_cpu_quota=quota,
                os_cpu_count=os.cpu_count())


def c5_record(run_training, iterations=4):
    """BASELINE.json configs[4] ("C5": the C2 batch inside an APPO training loop, train_local.sh:1-18 through swarm_rl/train.py:16-33).
    Where Sample Factory imports, tools/train_c5.py runs the reference's flag set through it and its FPS is recorded.  This image has no
    Sample Factory and no network: the record then holds the exact import error AND a run of the in-tree PPO harness (tools/ppo_c5.py: the
    same flag set, BatchedQuadSwarm with replay / shaping / annealing as the environment, the fixture-pinned encoder restatement as the
    policy, a synchronous PPO learner on PyTorch-ROCm) - `iterations` rollouts of 128 steps x 8192 agents with one update pass each: agent
    steps per second of the whole loop, of the sampling half alone, and the mean reward terms of the first and the last rollout."""
    import subprocess
    sf_error = None
    try:
        import sample_factory  # noqa: F401
    except Exception as exc:   # noqa: BLE001 - the exact error is what is recorded
        sf_error = f"{type(exc).__name__}: {exc}"
    if not run_training:
        return {"status": "training skipped (--no-c5-train)", "sample_factory": sf_error or "importable"}
    script = "train_c5.py" if sf_error is None else "ppo_c5.py"
    argv = [sys.executable, os.path.join(REPO, "tools", script)] + ([] if sf_error is None else [f"--iterations={iterations}", "--quiet"])
    try:
        out = subprocess.run(argv, capture_output=True, text=True, timeout=1200)
        lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
        if not lines:
            return {"status": "failed", "sample_factory": sf_error or "importable", "stderr_tail": out.stderr[-500:]}
        rec = json.loads(lines[-1])
        if sf_error is None:
            return rec
        return {"status": "ran: in-tree PPO harness (tools/ppo_c5.py); Sample Factory itself is not installed", "sample_factory": sf_error,
                "fps_agent_steps_per_s": rec["fps"], "agent_steps": rec["agent_steps"], "seconds": rec["seconds"], "first_rollout": rec["first"],
                "last_rollout": rec["last"], "recipe": "train_local.sh flags: mix, replay 0.75, annealing 3e8, attention encoder, 6 neighbours, lr 1e-4, "
                                                       f"rollout {rec['rollout']}, batch {rec['batch_size']}, 1024 envs x 8 quads",
                "learning_evidence": "tests/test_c5_training_gpu.py trains 1.0e8 agent-steps (minibatches of 8192) and asserts that the per-episode mean reward and "
                                     "rew_pos of the last episode are above the first's; measured curve: profiles/r05b_ppo_c5_b8192.txt (reward -0.0264 -> -0.0069, "
                                     "rew_pos -0.0159 -> -0.0079 per step over 8 episodes, 0.89e6 agent-steps/s)"}
    except Exception as exc:   # noqa: BLE001
        return {"status": "failed", "sample_factory": sf_error or "importable", "error": f"{type(exc).__name__}: {exc}"}


def closed_loop_record(device):
    """The device-resident stand-in for config 5's sampling loop: [fused attention encoder -> Gaussian action head -> env step] on the
    C2 batch, 32 control steps per HIP-graph replay (quad-swarm-rl_amd/rollout.py).  Random-init policy weights of the
    published architecture (QuadMultiEncoder, attention neighbour encoder, 6 neighbours); no learner."""
    import torch
    try:
        from quad_swarm_rl_amd import policy, rollout
        from quad_swarm_rl_amd.env import QuadSwarmVecEnv
        env = QuadSwarmVecEnv(1024, device=device, seed=0, num_agents=8, neighbor_visible_num=6, neighbor_obs_type="pos_vel", use_numba=True, use_downwash=True,
                              collision_falloff_radius=4.0, write_rew_info=False)
        env.reset()
        module = policy.make_reference_encoder(seed=0, nbr_encoder="attention").cuda(device)

        def loop_us(precision):
            enc = policy.FusedQuadEncoder(module, device=device, precision=precision)
            seg = rollout.GraphedRollout(env, enc, rollout.GaussianActionHead(device=device, sample=True), steps=32)
            for _ in range(3):
                seg.run()
            torch.cuda.synchronize(device)
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            reps = 20
            ev0.record()
            for _ in range(reps):
                seg.run()
            ev1.record()
            torch.cuda.synchronize(device)
            return ev0.elapsed_time(ev1) * 1e3 / (reps * 32)

        us, us32 = loop_us("bf16"), loop_us("fp32")
        env.close()
        return {"what": "encoder (attention, bf16 MFMA) -> sampled action -> env step, one HIP graph of 32 control steps, 1024 envs x 8 drones",
                "us_per_control_step": us, "env_steps_per_s": 1024 * 8 * 2 / (us * 1e-6),
                "reference_precision_encoder": {"what": "the same loop with the encoder's operands as fp16 pairs (precision='fp32', DESIGN.md 10: features within 1e-5 of the fp32 module)",
                                                "us_per_control_step": us32, "env_steps_per_s": 1024 * 8 * 2 / (us32 * 1e-6)}}
    except Exception as exc:   # noqa: BLE001 - an optional extra must not cost the bench line
        return {"status": "failed", "error": f"{type(exc).__name__}: {exc}"}


def pmc_traffic(workload, num_envs, kernel):
    """HBM bytes per launch of the step kernel from the committed rocprofv3 PMC pass (profiles/rNN_pmc_traffic.json, produced by
    tools/pmc.sh: FETCH_SIZE and WRITE_SIZE in separate passes, KiB units and gfx950 corrections of MI355X_MICROARCH.md);
    None when no measurement of this workload / batch / kernel is on file."""
    import glob
    return pmc_traffic_record(workload, num_envs, kernel)[0]


def pmc_traffic_record(workload, num_envs, kernel):
    """(bytes per launch, where the figure comes from): it is NOT measured by the run that prints the line - PMC counters need rocprofv3 around
    the process - but read from the committed PMC pass of the same workload / batch / kernel (newest round first)"""
    import glob
    for path in sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r*_pmc_traffic.json")), reverse=True):
        try:
            with open(path) as f:
                rec = json.load(f).get(f"{workload}:{num_envs}:{kernel}")
            if rec:
                return rec["fetch_bytes"] + rec["write_bytes"], f"committed PMC file profiles/{os.path.basename(path)} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this workload, tools/pmc.sh); not measured in this run"
        except (OSError, ValueError, KeyError):
            continue
    return None, None


def make_exchange(st, world, rank, transport, wire, dist, dev, info, protocols=(False, True)):
    """The observation exchange of this rank (quad-swarm-rl_amd/parallel.py).  auto / peer: the peer-store transport, kept only if
    EVERY rank could map its peers' windows and passed the start-up self-check (synthetic rows through both window slots); otherwise
    all ranks fall back to the RCCL all-gather together."""
    import torch
    from quad_swarm_rl_amd import parallel

    def all_agree(flag):
        if world == 1:
            return flag
        t = torch.tensor([1 if flag else 0], device=dev, dtype=torch.int32)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        return bool(t.item())

    if transport in ("auto", "peer", "fused"):
        kind = "peer" if transport == "peer" or not st.team else "fused"   # fused: the step kernel pushes its own rows (team kernels)
        info["attempts"] = []
        for fenced in protocols:   # the relaxed flag protocol, then its fenced variant (QS_XCHG_FENCED, include/quadswarm_exchange.h), then RCCL
            ex, why, att = None, "", {"flags": "fenced" if fenced else "relaxed"}
            try:
                ex = parallel.ObsExchange(st, world, rank, transport=kind, wire=wire, hold=False, fenced=fenced)
            except Exception as exc:   # noqa: BLE001 - recorded; the fallback is a different transport, not a different result
                why = f"{type(exc).__name__}: {exc}"
            ok = all_agree(ex is not None)
            att["windows_mapped"] = "on every rank" if ok else ("failed" + (f" here: {why}" if why else " on another rank"))
            if ok:
                try:
                    ok, why = ex.self_check()
                except Exception as exc:   # noqa: BLE001
                    ok, why = False, f"{type(exc).__name__}: {exc}"
                ok = all_agree(ok)
                att["peer_self_check"] = "passed on every rank" if ok else ("failed" + (f" here: {why}" if why else " on another rank"))
            if ok:   # ... and through the real producer of the rows (fused: the step kernels' epilogue is not what self_check exercises): the rows of
                #      an exchanged reset + two steps against an RCCL gather of the same float32 rows, on every rank.  Every rank-local action is
                #      agreed on before the next collective (verify() does that for its own local part): a rank that raised in reset() / step() while
                #      the others entered verify()'s all-gather would leave mismatched collectives behind
                for t in range(3):
                    try:
                        if t == 0:
                            ex.reset()
                        else:
                            ex.step(info["_actions_ptr"])
                            torch.cuda.synchronize()
                    except Exception as exc:   # noqa: BLE001
                        ok, why = False, f"{type(exc).__name__}: {exc}"
                    ok = all_agree(ok)
                    if not ok:
                        break
                    ok, why = ex.verify()
                    ok = all_agree(ok)
                    if not ok:
                        break
                att["verified_against_rccl_gather_before"] = "equal on every rank" if ok else ("differs" + (f" here: {why}" if why else " on another rank"))
            info["attempts"].append(att)
            info.update({k: v for k, v in att.items() if k != "flags"})
            if ok:
                info["transport"], info["flag_protocol"] = kind, att["flags"]
                return ex
            if ex is not None:
                ex.close()
    info["transport"] = "rccl"
    return parallel.ObsExchange(st, world, rank, transport="rccl", wire=wire, hold=False)


FLAG_PROTOCOLS = {"auto": (False, True), "relaxed": (False,), "fenced": (True,)}


def self_launch(n):
    """Re-runs this command line as `python -m torch.distributed.run --nnodes=1 --nproc-per-node n ... bench.py <same arguments>` (what the
    multi-GPU contract prescribes) and returns its exit code.  A free rendezvous port is picked here; 127.0.0.1 because the container's
    hostname may not resolve.  HSA_ENABLE_IPC_MODE_LEGACY=0: the host driver only supports dmabuf IPC (RCCL, hipIpc windows)."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def dry_run(args, rank, world):
    """--dry-run: everything of the N-rank launch path that needs no GPU - rendezvous (gloo), the barrier bracket of the timed region around
    K empty steps, MAX over ranks, ONE JSON line from rank 0 - so that the launch contract is testable on a CPU-only machine."""
    import torch
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29511")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        pass
    dist.barrier()
    tmax = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    seen = dist.get_world_size()
    workload = args.workload or ("c2" if world == 1 else "c4")
    # what every rank would step, and what the exchange would put on every link - the host arithmetic of the N-rank run, no GPU involved
    from quad_swarm_rl_amd import config as qcfg, native, parallel
    w = WORKLOADS[workload]
    E = args.envs_per_gpu or w["num_envs"]
    lo, hi = parallel.shard_range(world * E, world, rank)
    cfg = qcfg.make_config(num_envs=E, seed=0, env_id_offset=lo, precision="f32", write_rew_info=False, **w["kw"])
    mine = {"rank": rank, "envs": [lo, hi], "env_id_offset": int(cfg.env_id_offset), "drones": E * cfg.num_agents}
    shards = [None] * world
    dist.all_gather_object(shards, mine)
    per_wire = None
    try:
        D = int(native.lib().qs_obs_dim(cfg))
        per_wire = {}
        for wire in ("f32", "bf16", "q8"):
            rb = parallel.wire_row_bytes(D, wire, native.wire_q8_layout(cfg, D) if wire == "q8" else None)
            link = E * cfg.num_agents * rb
            per_wire[wire] = {"row_bytes": rb, "lossy": wire != "f32", "bytes_per_link_per_step": link, "predicted_link_us_per_step_at_77_GB_per_s": 1e6 * link / 77e9,
                              "ms_per_step": None, "exchange_cost_us_per_step": None, "verified_against_rccl_gather_before": None, "verified_against_rccl_gather_after": None}
    except Exception as exc:   # noqa: BLE001 - e.g. the library is not built on this machine
        per_wire = {"error": f"{type(exc).__name__}: {exc}"}
    if rank == 0:
        print(json.dumps({"metric": "env-steps/s (drones x envs x sim_steps)", "value": None, "unit": "env-steps/s", "n_gpus": world, "steps": args.steps,
                          "warmup": args.warmup, "ms_per_step": None, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
                          "data": "synthetic", "dry_run": True, "wire": args.wire if world > 1 and not args.no_gather else None,
                          "config": {"workload": workload, "ranks_seen_by_process_group": seen, "backend": "gloo",
                                     "bracket_host_seconds_max_over_ranks": float(tmax.item()), "shards": shards, "exchange_per_wire": per_wire,
                                     "launch": "self-launched ranks" if os.environ.get("TORCHELASTIC_RUN_ID") else "single process"}}), flush=True)
    dist.destroy_process_group()
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--prewarm", type=int, default=3000, help="steps of a SCRATCH handle of the same configuration run before anything is measured "
                    "(device clocks / code caches: a 20-step timed region is 160 us long); the measured handle still takes exactly "
                    "--warmup untimed steps, then --steps timed ones.  0 = off")
    ap.add_argument("--workload", default=None, choices=sorted(WORKLOADS), help="default: c2 at --gpus 1, c4 (512 envs per GPU) at --gpus N>1")
    ap.add_argument("--envs-per-gpu", type=int, default=0, help="override the workload's env count per GPU")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="CPU baseline sample length in total (0 = skip)")
    ap.add_argument("--profile-steps", type=int, default=400, help="steps of the HIP-event-pair-per-launch pass")
    ap.add_argument("--no-gather", action="store_true", help="N>1: headline = independent shards (no collective in the timed region); the gather "
                                                            "variant is then the secondary measurement")
    ap.add_argument("--gather", action="store_true", help="(default at N>1) ONE RCCL all-gather of the observations after every step inside the timed region")
    ap.add_argument("--force-gather", action="store_true", help="run the RCCL obs all-gather path even at N=1 (exercises the multi-GPU code on a 1-GPU box)")
    ap.add_argument("--no-secondary", action="store_true", help="N>1: skip the secondary measurement (independent shards / gather variant)")
    ap.add_argument("--no-overlap", action="store_true", help="--transport torch: gather on the compute stream instead of overlapping it with the next step")
    ap.add_argument("--transport", default="auto", choices=["auto", "fused", "peer", "rccl", "torch"], help="observation exchange (see the module docstring)")
    ap.add_argument("--wire", default="f32", choices=["q8", "bf16", "f32"], help="wire format of the HEADLINE's exchanged rows.  Default f32: bit-exact, the reference's float "
                                                                                 "observations (the other two are measured as labelled lossy variants in config.exchange_per_wire).  bf16: RNE, 3 significant "
                                                                                 "digits; q8 (bf16 self / SDF columns + 8-bit fixed-point neighbour block, 72 bytes per C4 row, error <= 0.039 m / 0.024 m/s: "
                                                                                 "include/quadswarm_exchange.h): outside the 1e-5 observation tolerance.  The line names the wire in its top-level `wire` field")
    ap.add_argument("--segment", type=int, default=64, help="control steps per captured [step -> exchange] graph (0 = eager launches)")
    ap.add_argument("--no-variants", action="store_true", help="skip config.variants (shaped / rew_info / downwash-off / seeds 1, 2 runs of the same workload)")
    ap.add_argument("--no-c5-train", action="store_true", help="do not run the C5 training (tools/train_c5.py through Sample Factory where it imports, else the in-tree PPO harness tools/ppo_c5.py)")
    ap.add_argument("--no-f64", action="store_true", help="skip the f64 line (same workload through the float64 kernels)")
    ap.add_argument("--no-closed-loop", action="store_true", help="skip config.c5.closed_loop_without_sample_factory (encoder -> action -> step as a HIP graph)")
    ap.add_argument("--rew-info", action="store_true", help="also write the 17-term reward-info matrix every step (logging output)")
    ap.add_argument("--set", action="append", default=[], metavar="KEY=VALUE", help="override a workload keyword (python literal)")
    ap.add_argument("--graph", type=int, default=0, help="headline mode: step the timed region as open-loop rollouts of this many steps per "
                                                          "launch (qs_step_many: state stays in registers between the steps)")
    ap.add_argument("--rollout-steps", type=int, default=64, help="steps per launch of the extra open-loop measurement (0 = skip)")
    ap.add_argument("--flag-protocol", default="auto", choices=["auto", "relaxed", "fenced"], help="flag protocol of the window transports: auto = relaxed flags, the fenced "
                                                                                                       "variant if verify() disagrees, then RCCL; relaxed / fenced = that one only (then RCCL)")
    ap.add_argument("--no-wire-sweep", action="store_true", help="with an observation exchange: measure the headline's wire only (config.exchange_per_wire otherwise holds f32, bf16 and q8)")
    ap.add_argument("--dry-run", action="store_true", help="launch path only: ranks rendezvous over gloo (no GPU needed), take the barriers of the timed "
                                                           "bracket around K empty steps, rank 0 prints one JSON line with value null")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` started directly: become the launcher of its own N ranks (one process per GPU); the ranks re-enter
        # main() with RANK / LOCAL_RANK / WORLD_SIZE set and rank 0 prints the one JSON line on the inherited stdout
        raise SystemExit(self_launch(args.gpus))

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: the launcher's --nproc-per-node must equal --gpus")
    if args.dry_run:
        return dry_run(args, rank, world)

    import torch
    from quad_swarm_rl_amd import config as qcfg, native

    def trace(msg):   # BENCH_TRACE=1: stage markers on stderr (where did a run stop?)
        if os.environ.get("BENCH_TRACE"):
            print(f"[bench rank {rank} +{time.perf_counter() - t_start:.2f}s] {msg}", file=sys.stderr, flush=True)
    t_start = time.perf_counter()

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP stepper has no CPU fallback")
    if torch.cuda.device_count() <= local_rank:
        raise SystemExit(f"--gpus {args.gpus}: rank {rank} wants cuda:{local_rank} but this node shows {torch.cuda.device_count()} GPU(s)")
    torch.cuda.set_device(local_rank)
    dev = f"cuda:{local_rank}"
    use_gather = (world > 1 and not args.no_gather) or args.force_gather or (world == 1 and args.gather)
    dist = None
    if world > 1 or use_gather:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        import datetime
        # (a collective that a failed rank never joins aborts after 3 minutes instead of the default 10: the driver runs N = 1, 2, 4, 8 back to back)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank), timeout=datetime.timedelta(seconds=180))
    ranks_seen = dist.get_world_size() if dist is not None else 1
    trace("process group up")

    import ast
    workload = args.workload or ("c2" if world == 1 else "c4")
    w = WORKLOADS[workload]
    kw = dict(w["kw"])
    for item in args.set:
        key, val = item.split("=", 1)
        kw[key] = ast.literal_eval(val)
    E = args.envs_per_gpu or w["num_envs"]
    cfg = qcfg.make_config(num_envs=E, seed=0, env_id_offset=rank * E, precision="f32", write_rew_info=args.rew_info, **kw)
    st = native.Stepper(cfg, device=local_rank)
    N, T, D = cfg.num_agents, E * cfg.num_agents, st.obs_dim
    trace("stepper created")
    stream = torch.cuda.current_stream(local_rank)

    # synthetic actions, resident in HBM before the timed region: a ring of pre-drawn U(-1,1)^4 batches
    gen = torch.Generator(device=dev)
    gen.manual_seed(1234 + rank)
    ring = 64
    actions = (torch.rand((ring, T, 4), device=dev, generator=gen, dtype=torch.float32) * 2.0 - 1.0).contiguous()
    aptr, astride = actions.data_ptr(), T * 4 * 4
    obs = st.tensor("obs")
    gather_obj, exchange, xinfo = None, None, {"requested": args.transport, "wire": args.wire, "_actions_ptr": aptr}
    if dist is not None and (use_gather or not args.no_secondary):
        from quad_swarm_rl_amd import parallel
        if args.transport == "torch":   # round 2's transport: eager torch.distributed all_gather of the float32 rows, per step
            gather_obj = parallel.ObsGather(obs, overlap=not args.no_overlap)
            xinfo["transport"] = "torch"
        else:
            exchange = make_exchange(st, world, rank, args.transport, args.wire, dist, dev, xinfo, protocols=FLAG_PROTOCOLS[args.flag_protocol])
    xinfo.pop("_actions_ptr", None)
    trace("exchange ready: " + str(xinfo.get("transport")) + "")

    def run(stepper, base_ptr, stride, k, offset=0, gather=None):
        if args.graph > 0 and gather is None:
            # open-loop rollout over the action ring: qs_step_many keeps the state in registers across the steps of a launch
            g = min(args.graph, ring)
            done_steps = 0
            while done_steps + g <= k:
                stepper.step_many(base_ptr, g, stream=stream)
                done_steps += g
            for t in range(k - done_steps):
                stepper.step(base_ptr + (t % ring) * stride, stream=stream)
            return
        for t in range(k):
            stepper.step(base_ptr + ((offset + t) % ring) * stride, stream=stream)
            if gather is not None:
                gather.gather()
        if gather is not None:
            gather.drain()   # the launch stream waits for the in-flight collectives: the closing event sees them

    seg = 0

    def prepare_exchange(info):
        """reset (its rows are exchanged like a step's), then [step -> exchange] x seg as one HIP graph, captured before any timed region (the
        eager steps capture() takes first are untimed)"""
        nonlocal seg
        exchange.reset()
        seg = min(args.segment, ring, max(args.steps, 2)) & ~1   # (a whole number of segments fits the timed region also at --steps 20)
        if seg >= 2 and info.get("transport") == "rccl":
            seg = 0            # torch's collective must not be recorded into a graph (parallel.ObsExchange.capture): eager steps
        if seg >= 2:
            exchange.capture([aptr + t * astride for t in range(seg)])

    if exchange is not None:
        prepare_exchange(xinfo)

    def run_exchange(k):
        """exactly k control steps, each followed by the exchange of its rows: whole segments as graph replays, the rest eagerly"""
        done_steps = 0
        if seg >= 2 and k >= seg:
            exchange.align(aptr)   # (an even number of steps issued before a replay; a no-op inside the timed region, see timed())
            while done_steps + seg <= k:
                exchange.replay()
                done_steps += seg
        for t in range(k - done_steps):
            exchange.step(aptr + (t % ring) * astride)
        exchange.drain()     # the stepping stream waits for the last exchanges: the closing event sees them

    def timed(stepper, base_ptr, stride, warmup, k, gather=None, xchg=False):
        """W untimed steps, then exactly K steps bracketed by barrier + synchronize; HIP events on the launch stream right inside
        the bracket.  Returns (device-event seconds, host-clock seconds), each the MAX over ranks."""
        if xchg:
            run_exchange(warmup)
            exchange.align(aptr)            # an even number of steps issued: the timed region can start with a replay
        else:
            if exchange is not None:
                exchange.pause()
            run(stepper, base_ptr, stride, warmup, 0, gather)
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        ev0.record(stream)
        if xchg:
            run_exchange(k)
        else:
            run(stepper, base_ptr, stride, k, warmup, gather)
        ev1.record(stream)
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        host = time.perf_counter() - t0
        devs = ev0.elapsed_time(ev1) * 1e-3
        if dist is not None:
            tmax = torch.tensor([devs, host], device=dev, dtype=torch.float64)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            devs, host = (float(x) for x in tmax.tolist())
        return devs, host

    episode_end = None
    if args.prewarm > 0:   # a scratch handle (its own state and noise streams): the measured one is untouched by it
        s0 = native.Stepper(cfg, device=local_rank)
        s0.reset(stream=stream)
        for t in range(args.prewarm):
            s0.step(aptr + (t % ring) * astride, stream=stream)
        torch.cuda.synchronize()
        try:   # the step on which EVERY environment ends its episode (what happens once per ep_len + 1 steps), 5 samples by HIP events
            samples = []
            for rep in range(5):
                tick = s0.to_host("tick")
                tick[:] = cfg.ep_len
                s0.from_host("tick", tick)
                for t in range(3):   # (two ordinary steps first: the copy above left the queue cold)
                    if t == 2:
                        tick = s0.to_host("tick"); tick[:] = cfg.ep_len; s0.from_host("tick", tick)
                        torch.cuda.synchronize()
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record(stream)
                    s0.step(aptr + (t % ring) * astride, stream=stream)
                    e1.record(stream)
                    torch.cuda.synchronize()
                    if t == 2:
                        samples.append(1e3 * e0.elapsed_time(e1))
                assert int(s0.to_host("done").min()) == 1
            samples.sort()
            episode_end = {"us_median_of_5": samples[2], "all_envs_end_together": True, "amortised_us_per_step": samples[2] / (cfg.ep_len + 1),
                           "note": "a single launch bracketed by its own event pair (includes ~2 us of event overhead an ordinary step in a stream of steps does not pay)"}
        except Exception as exc:   # noqa: BLE001 - a measurement beside the headline, never fatal
            episode_end = {"error": f"{type(exc).__name__}: {exc}"}
        s0.close()
    if exchange is None:
        st.reset(stream=stream)
    # Auto-resets ("included in timing").  Every environment of the batch starts its episode together, so episodes end together every
    # ep_len + 1 control steps: a timed region of at least that many steps (the default 3000) contains the episode end of every environment; the
    # driver's --steps 20 region contains none - stated in config.workload - and the cost of the step on which all environments end their episodes
    # (statistics snapshot + device-side reset + fresh observation for E envs), measured on the scratch handle above, is reported beside it with
    # its amortised share per step.  (Spreading a K / ep_len share of episode ends over a short region was tried: in the latency regime one
    # resetting environment delays its whole launch, so 14 of 20 steps paid a reset tail - 11.1 us per step where the steady state is 8.0 +
    # 0.02: profiles/r05a_bench_c2_steps20.json.)
    crossing = {"episode_ends_inside_the_timed_region": bool(args.warmup + args.steps > cfg.ep_len), "episode_length_steps": int(cfg.ep_len) + 1,
                "episode_end_step": episode_end}
    trace("timed region")
    head_dev, head_host = timed(st, aptr, astride, args.warmup, args.steps, gather_obj if use_gather else None, xchg=use_gather and exchange is not None)
    trace("timed region done")
    st.check_errors()
    if exchange is not None:
        wire_b = exchange.x.row_bytes / D
        # what the wire costs on a real node: every rank sends its T rows to each of the world-1 peers over that peer's own xGMI link
        link_bytes = T * exchange.x.row_bytes
        xinfo.update(row_bytes=exchange.x.row_bytes, bytes_per_link_per_step=link_bytes,
                     predicted_link_us_per_step={"at_77_GB_per_s_per_direction": 1e6 * link_bytes / 77e9, "at_100_GB_per_s": 1e6 * link_bytes / 100e9,
                                                 "note": "per-link time of one step's rows (point-to-point, all 7 links concurrently); the 8-GPU line is link-bound "
                                                         "when this exceeds the step time printed as secondary.ms_per_step"})
        try:   # the rows of the LAST timed step against an RCCL gather of the same rows (after the timed region: nothing overwrites the slot)
            torch.cuda.synchronize()
            v_ok, v_why = exchange.verify()
            xinfo["verified_against_rccl_gather_after"] = "equal" if v_ok else f"differs: {v_why}"
        except Exception as exc:   # noqa: BLE001
            xinfo["verified_against_rccl_gather_after"] = f"not run: {type(exc).__name__}: {exc}"
        gather_desc = (f"{xinfo.get('transport')} transport, {args.wire} wire: every rank receives the rows of all ranks after each step; "
                       + (f"[step -> exchange] x {seg} per captured HIP graph" if seg >= 2 else "eager launches") + ", exchange(t) on a second stream under step(t+1)")
        xinfo["status"] = exchange.status()
    else:
        wire_b = 4
        gather_desc = "torch.distributed all_gather_into_tensor of the float32 obs per step (eager)" + ("" if args.no_overlap else ", double-buffered: gather(t) overlaps step(t+1)")

    # N>1 / --force-gather: the other variant with the same bracketing (independent shards when the gather is the headline, and
    # the other way round); its step-kernel-only region is also where the kernel duration of the roofline comes from
    secondary = None
    kernel_region_s, kernel_region_steps = (None, 0) if use_gather else (head_dev, args.steps)
    if dist is not None and not args.no_secondary:
        sk = max(args.steps, 50) if use_gather else min(max(args.steps, 50), 1000)
        sdev, shost = timed(st, aptr, astride, min(args.warmup, 50), sk, None if use_gather else gather_obj, xchg=(not use_gather) and exchange is not None)
        secondary = {"value": world * T * 2 * sk / sdev, "unit": "env-steps/s", "ms_per_step": 1e3 * sdev / sk, "steps": sk,
                     "host_clock_ms_per_step": 1e3 * shost / sk,
                     "what": "independent shards: no data-path collective in the timed region" if use_gather else gather_desc}
        if use_gather:
            kernel_region_s, kernel_region_steps = sdev, sk
            secondary["exchange_cost_us_per_step"] = 1e6 * (head_dev / args.steps - sdev / sk)
    # every wire on the same shards, same bracketing: the measured step with the exchange, what the exchange costs over the
    # independent shards, verify() before (make_exchange) and after the timed steps, the predicted per-link time.  f32 is the bit-exact wire;
    # bf16 and q8 are lossy (q8 beyond the 1e-5 observation tolerance): the headline's wire is named in the top-level `wire` field.
    wire_table = None
    if use_gather and exchange is not None and not args.no_wire_sweep and dist is not None:
        base_ms = secondary["ms_per_step"] if secondary else None

        def wire_entry(info, dev_s, k):
            rb = exchange.x.row_bytes
            ent = {"transport": info.get("transport"), "flag_protocol": info.get("flag_protocol"), "row_bytes": rb, "lossy": exchange.wire != "f32",
                   "ms_per_step": 1e3 * dev_s / k, "value": world * T * 2 * k / dev_s,
                   "exchange_cost_us_per_step": (1e3 * (1e3 * dev_s / k - base_ms)) if base_ms is not None else None,
                   "predicted_link_us_per_step_at_77_GB_per_s": 1e6 * T * rb / 77e9,
                   "verified_against_rccl_gather_before": info.get("verified_against_rccl_gather_before"),
                   "verified_against_rccl_gather_after": info.get("verified_against_rccl_gather_after")}
            return ent

        wire_table = {args.wire: wire_entry(xinfo, head_dev, args.steps)}

        def agree(flag):   # every rank-local action of the sweep is agreed on before the next collective: a rank that failed alone must not leave the
            if world == 1:   # others waiting in a barrier (they all skip the wire together instead)
                return flag
            t = torch.tensor([1 if flag else 0], device=dev, dtype=torch.int32)
            dist.all_reduce(t, op=dist.ReduceOp.MIN)
            return bool(t.item())

        for w_name in ("f32", "bf16", "q8"):
            if w_name == args.wire:
                continue
            wi = {"requested": args.transport, "wire": w_name, "_actions_ptr": aptr}
            err, new_exchange = None, None
            try:
                exchange.close()
            except Exception as exc:   # noqa: BLE001 - recorded
                err = f"close: {type(exc).__name__}: {exc}"
            exchange = None
            if agree(err is None):
                try:
                    new_exchange = make_exchange(st, world, rank, args.transport, w_name, dist, dev, wi, protocols=FLAG_PROTOCOLS[args.flag_protocol])   # (agrees on its own steps)
                except Exception as exc:   # noqa: BLE001
                    err = f"create: {type(exc).__name__}: {exc}"
            if agree(err is None and new_exchange is not None):
                exchange = new_exchange
                wi.pop("_actions_ptr", None)
                try:
                    prepare_exchange(wi)
                except Exception as exc:   # noqa: BLE001
                    err = f"prepare: {type(exc).__name__}: {exc}"
            else:
                err = err or "failed on another rank"
            if not agree(err is None):
                wire_table[w_name] = {"error": err or "failed on another rank"}
                if exchange is not None:
                    try:
                        exchange.close()
                    except Exception:   # noqa: BLE001
                        pass
                    exchange = None
                break
            wd, _ = timed(st, aptr, astride, min(args.warmup, 50), args.steps, None, xchg=True)
            torch.cuda.synchronize()
            v_ok, v_why = exchange.verify()
            wi["verified_against_rccl_gather_after"] = "equal" if v_ok else f"differs: {v_why}"
            wire_table[w_name] = wire_entry(wi, wd, args.steps)
            wire_table[w_name]["status"] = exchange.status()
    if kernel_region_s is None:   # gather headline without secondary: a short step-only region for the roofline
        kernel_region_steps = max(args.steps, 50)
        kernel_region_s, _ = timed(st, aptr, astride, 10, kernel_region_steps, None)
    if exchange is not None:
        exchange.pause()
        torch.cuda.synchronize()

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
    st_team = bool(st.team)
    kernel_name, flavor = st.kernel_name, ("config-specialised, " if st.specialized else "generic, ") + f"{st.waves_per_workgroup} wave{'s' if st.waves_per_workgroup > 1 else ''} per workgroup"
    specialized = bool(st.specialized)
    if exchange is not None:
        exchange.close()
        exchange = None
    st.close()

    # the same workload as the callers of the boundary run it (seeds 0 / 1 / 2 with the median, downwash off; VERDICT r02:
    # the reward-shaping sums the Sample Factory env keeps on the device, and the infos['rewards'] matrix), same bracketing, HIP events
    variants = None
    if world == 1 and not args.no_variants and args.graph == 0:
        def quick(seed=0, kw_over=None, extra_bytes=0, **cfg_over):
            k2 = dict(kw)
            k2.update(kw_over or {})
            c2 = qcfg.make_config(num_envs=E, seed=seed, env_id_offset=rank * E, precision="f32", **dict(dict(write_rew_info=args.rew_info), **cfg_over), **k2)
            s2 = native.Stepper(c2, device=local_rank)
            s2.reset(stream=stream)
            n = min(max(args.steps, 200), 2000)
            d, _ = timed(s2, aptr, astride, 100, n)
            s2.check_errors()
            rec = {"value": T * 2 * n / d, "kernel_avg_us": 1e6 * d / n, "steps": n, "specialized": bool(s2.specialized),
                   "algorithmic_bytes_per_drone_step": ALGO_BYTES_PER_DRONE_STEP[workload] + extra_bytes}
            rec["achieved_GBs"] = rec["algorithmic_bytes_per_drone_step"] * T / (d / n) / 1e9
            s2.close()
            return rec
        seeds = [T * 2 * args.steps / head_dev, quick(seed=1)["value"], quick(seed=2)["value"]]
        variants = {
            "seeds_0_1_2": seeds, "seeds_median": float(np.median(seeds)),
            # QuadsRewardShapingWrapper on the device (swarm_rl/env_wrappers/reward_shaping.py:69-83): 25 per-episode sums per drone,
            # read-modify-write = +200 B per drone-step; what sf_env.BatchedQuadSwarm always runs
            "shaped_episode_sums": quick(episode_sums=True, extra_bytes=200),
            # + the 17 infos['rewards'] terms of every step (quadrotor_single.py:68-85, quadrotor_multi.py:533-540): +68 B
            "shaped_episode_sums_and_rew_info": quick(episode_sums=True, write_rew_info=True, extra_bytes=268),
            "downwash_off": quick(kw_over=dict(use_downwash=False)),
        }
        # resident-state stepping (include/quadswarm.h qs_step_gated): ONE launch per 256 control steps keeps the state in registers,
        # waits per step and workgroup for the step's actions and publishes its outputs; the actions come from a producer kernel on another
        # stream (qs_gate_produce: copies each batch of the pre-drawn table into the gate's ring, written through the L2, then raises the flags)
        # - running ahead of the stepper (bounded by the 64-slot ring), or closed loop (the batch of step s only after the outputs of s - 1).
        def gated_line(closed_loop):
            try:
                c2 = qcfg.make_config(num_envs=E, seed=0, env_id_offset=rank * E, precision="f32", write_rew_info=args.rew_info, **kw)
                s2 = native.Stepper(c2, device=local_rank)
                # producer groups: 1 workgroup per group keeps the closed-loop chain shortest (6.98 - 7.25 us per step against 7.5 - 7.8 with 8),
                # 8 per group the producer cheapest when it runs ahead (tools/gated_probe.py, profiles/r04e_gated_probe_c2.txt)
                s2.gate_create(ring_len=ring, wg_per_group=1 if closed_loop else 8)
                s2.reset(stream=stream)
                torch.cuda.synchronize()
                side, feed = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
                k, reps = 256, max(2, min(max(args.steps, 200), 2000) // 256)   # 256 control steps per launch (64: + 0.4 us per step, tools/gated_probe.py)
                s2.step_gated(k, stream=side); s2.gate_produce(aptr, ring, k, closed_loop, stream=feed)   # warm-up launch pair
                torch.cuda.synchronize()
                ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                t0 = time.perf_counter()
                ev0.record(side)
                for _ in range(reps):
                    s2.step_gated(k, stream=side)
                    s2.gate_produce(aptr, ring, k, closed_loop, stream=feed)
                    s2.gate_wait(stream=side)   # `side` behind the launch - issued after the producer's launch (include/quadswarm.h)
                ev1.record(side)
                torch.cuda.synchronize()
                host = time.perf_counter() - t0
                d = ev0.elapsed_time(ev1) * 1e-3
                st2 = s2.gate_status()
                s2.check_errors()
                rec = {"value": T * 2 * reps * k / d, "kernel_avg_us": 1e6 * d / (reps * k), "host_clock_us_per_step": 1e6 * host / (reps * k), "steps": reps * k,
                       "steps_per_launch": k, "specialized": bool(s2.specialized), "gate_status": st2,
                       "achieved_GBs": ALGO_BYTES_PER_DRONE_STEP[workload] * T / (d / (reps * k)) / 1e9,
                       "producer": "closed loop: batch s is written only after the outputs of step s - 1 of the same workgroups are published" if closed_loop
                                   else "runs ahead of the stepper, bounded by the 64-slot action ring"}
                s2.close()
                return rec
            except Exception as exc:   # noqa: BLE001 - an optional extra must not cost the bench line
                return {"status": "failed", "error": f"{type(exc).__name__}: {exc}"}
        if st_team:
            variants["resident_state_gated_producer_ahead"] = gated_line(False)
            variants["resident_state_gated_closed_loop"] = gated_line(True)
        if not kw.get("use_obstacles"):
            # what train_local.sh trains on: --quads_mode=mix (scenarios/mix.py: every env draws one of the non-obstacle scenarios per
            # episode) with the device-side episode sums - the full-scenario kernels; 700 warm-up steps so that the
            # periodic goal changes of the dynamic scenarios are inside the timed region
            def mix_line():
                c2 = qcfg.make_config(num_envs=E, seed=0, env_id_offset=rank * E, precision="f32", write_rew_info=False, episode_sums=True, **dict(kw, quads_mode="mix"))
                s2 = native.Stepper(c2, device=local_rank)
                s2.reset(stream=stream)
                n = min(max(args.steps, 200), 2000)
                d, _ = timed(s2, aptr, astride, 700, n)
                s2.check_errors()
                rec = {"value": T * 2 * n / d, "kernel_avg_us": 1e6 * d / n, "steps": n, "specialized": bool(s2.specialized)}
                s2.close()
                return rec
            variants["mix_scenarios_shaped"] = mix_line()

    # the same workload through the float64 instantiation (the one whose free-running flags are bit-exact against the oracle)
    f64 = None
    if world == 1 and not args.no_f64:
        cfg64 = qcfg.make_config(num_envs=E, seed=0, env_id_offset=rank * E, precision="f64", write_rew_info=args.rew_info, **kw)
        st64 = native.Stepper(cfg64, device=local_rank)
        act64 = actions[:16].double().contiguous()
        st64.reset(stream=stream)
        k64 = min(max(args.steps, 50), 500)
        ring_save, ring = ring, 16
        d64, h64 = timed(st64, act64.data_ptr(), T * 4 * 8, min(args.warmup, 50), k64)
        ring = ring_save
        st64.check_errors()
        f64 = {"value": T * 2 * k64 / d64, "unit": "env-steps/s", "dtype": "f64", "steps": k64, "kernel_avg_us": 1e6 * d64 / k64,
               "host_clock_ms_per_step": 1e3 * h64 / k64, "kernel": st64.kernel_name,
               "achieved_GBs": ALGO_BYTES_F64_PER_DRONE_STEP.get(workload, 0) * T / (d64 / k64) / 1e9}
        st64.close()

    if rank == 0:
        value = world * T * 2 * args.steps / head_dev
        algo = ALGO_BYTES_PER_DRONE_STEP[workload]
        region_kernel_ms = 1e3 * kernel_region_s / kernel_region_steps
        achieved = algo * T / (region_kernel_ms * 1e-3) / 1e9 if region_kernel_ms > 0 else 0.0
        out = {
            "metric": "env-steps/s (drones x envs x sim_steps)", "value": value, "unit": "env-steps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * head_dev / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "wire": (args.wire if use_gather else None),   # the observation exchange's wire format (f32 = bit-exact; bf16 / q8 are lossy), null without an exchange
            "config": {"workload": f"{workload}: {N} drones x {E} envs per GPU ({world * E} envs in total), {kw.get('quads_mode', 'static_same_goal')}, "
                                   f"K={cfg.num_neighbors} neighbours, obs_dim {D}, downwash {bool(cfg.use_downwash)}, sensor+thrust noise on, auto-reset on "
                                   + ("(every environment's episode ends inside the timed region)" if args.warmup + args.steps > cfg.ep_len else
                                      f"(episodes last {cfg.ep_len + 1} steps: none ends inside this {args.steps}-step region; config.auto_reset has the episode-end step's cost)"),
                       "drone_control_steps_per_s": value / 2.0, "envs_per_gpu": E, "num_agents": N, "ranks_seen_by_process_group": ranks_seen,
                       "timing": "HIP events on the launch stream inside the barrier+synchronize bracket of the K timed steps, max over ranks",
                       "host_clock": {"ms_per_step": 1e3 * head_host / args.steps, "value": world * T * 2 * args.steps / head_host,
                                      "note": "perf_counter over the same K steps incl. the closing barrier + synchronize"},
                       "obs_gather": gather_desc if use_gather else ("none: env shards are independent, no data-path collective (--no-gather)" if world > 1 else "none"),
                       "gather_bytes_per_gpu_per_step": int((world - 1) * T * D * wire_b) if use_gather else 0,
                       "exchange": xinfo if (use_gather or secondary) else None,
                       "exchange_per_wire": wire_table,
                       "secondary": secondary,
                       "auto_reset": crossing,
                       "launch": f"open-loop rollout, {min(args.graph, ring)} steps per launch" if args.graph > 0 and not use_gather else "one launch per control step",
                       "device_prewarm_steps_on_a_scratch_handle": args.prewarm, "open_loop_rollout": rollout, "f64": f64, "rew_info": bool(args.rew_info), "variants": variants,
                       "c5": dict(c5_record(not args.no_c5_train), closed_loop_without_sample_factory=closed_loop_record(local_rank)) if world == 1 and not args.no_secondary and not args.no_closed_loop
                       else (c5_record(not args.no_c5_train) if world == 1 else None),
                       "overrides": args.set},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": pmc_traffic_record(workload, E, kernel_name)[0], "traffic_source": pmc_traffic_record(workload, E, kernel_name)[1], "kernel": kernel_name,
                         "kernel_flavor": flavor, "specialized": specialized, "kernel_avg_us": region_kernel_ms * 1e3, "kernel_launches": kernel_region_steps,
                         "kernel_avg_us_event_pair_per_launch": kernel_ms * 1e3, "event_pair_launches": launches,
                         "algorithmic_bytes_per_launch": algo * T},
        }
        if world == 1 and args.cpu_seconds > 0:
            out["cpu_baseline"] = cpu_baseline(workload, args.cpu_seconds)
            cb = out["cpu_baseline"]
            # the ratio is always quoted WITH the thread count it was measured on: a CPU-over-GPU ratio is not a
            # statement about kernel quality - roofline.frac is
            cb["gpu_over_cpu"] = {"ratio": value / cb["value"], "cpu_threads": cb["cores"], "ratio_per_thread": value / cb["single_thread"],
                                  "statement": f"{value / cb['value']:.1f}x the C oracle on {cb['cores']} host threads (cgroup quota {cb['cgroup_cpu_quota']}) of a {cb['cpu_model']}"}
        else:
            out["cpu_baseline"] = None
        line = json.dumps(out)
    else:
        line = None
    if dist is not None:
        dist.destroy_process_group()
    # RCCL writes its version banner to the C-level stdout (NCCL_DEBUG=VERSION in this image), which is flushed when the process ends - BEHIND a
    # line printed from Python.  The one JSON line must be the LAST line of rank 0's stdout: flush the C streams first, then print.
    try:
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except Exception:   # noqa: BLE001
        pass
    if line is not None:
        print(line, flush=True)
    trace("line printed")


if __name__ == "__main__":
    main()
