"""Worker of tests/test_exchange_gpu.py: one rank of a sharded stepper + peer-store exchange, or both ranks in one process.

    python tests/xchg_worker.py --mode proc   --rank R --world W --port P ...   one process per rank (hipIpc-mapped windows); the
                                                                                  window handles travel over a gloo group
    python tests/xchg_worker.py --mode local  --world W ...                     W endpoints of ONE process, wired with attach_local

Every rank steps its shard of a C2-shaped batch (8 drones, K=6) on GPU 0 with the global actions of tests (RandomState(seed)),
records the gathered rows after the reset and after every step (eager) / every replay (graph), and writes them to --out.
"""
import argparse
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

import numpy as np  # noqa: E402

KW = dict(num_agents=8, neighbor_visible_num=6, neighbor_obs_type="pos_vel", use_downwash=True, use_numba=True, collision_falloff_radius=4.0,
          rew_coeff=dict(quadcol_bin=5.0, quadcol_bin_smooth_max=10.0))


def global_actions(steps, total_envs, n, seed=0):
    return np.random.RandomState(seed).uniform(-1, 1, size=(steps, total_envs * n, 4)).astype(np.float32)


# ---- the replay scenario (--replay P): drones that hover (crash rewards stay at zero), a collision planted at tick 169 in every third
# environment, 2.6-s episodes, every replay buffer switched on: checkpoints, filed events and restored episodes within ~600 steps ----
REPLAY_KW = dict(num_agents=4, neighbor_visible_num=2, neighbor_obs_type="pos_vel", use_numba=True, ep_time=2.6)


def replay_actions(steps, total_envs, n, seed=1):
    return (0.06 + np.random.RandomState(seed).uniform(-0.02, 0.02, size=(steps, total_envs * n, 4))).astype(np.float32)


def plant_collisions(stepper, first_global_env):
    """two drones 3 cm apart at tick 169 in the environments whose GLOBAL index is a multiple of 3 (the same envs however the batch is sharded)"""
    ticks = stepper.to_host("tick")
    for e in np.nonzero(ticks == 169)[0]:
        if (first_global_env + int(e)) % 3 == 0:
            s, tk = stepper.get_state(int(e))
            s[1, 0:3] = s[0, 0:3] + np.array([0.03, 0.0, 0.0]); s[1, 3:6] = s[0, 3:6]
            stepper.set_state(int(e), s, tk)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mode", choices=["proc", "local"], required=True)
    ap.add_argument("--rank", type=int, default=0)
    ap.add_argument("--world", type=int, default=2)
    ap.add_argument("--port", type=int, default=29533)
    ap.add_argument("--wire", default="bf16")
    ap.add_argument("--envs", type=int, default=16, help="total envs over all ranks")
    ap.add_argument("--steps", type=int, default=12)
    ap.add_argument("--graph", type=int, default=0, help="steps per captured graph (0 = eager)")
    ap.add_argument("--replays", type=int, default=3)
    ap.add_argument("--hold", type=int, default=1)
    ap.add_argument("--transport", default="peer", choices=["peer", "fused"])
    ap.add_argument("--verify", type=int, default=0)
    ap.add_argument("--replay", type=float, default=0.0, help="device-side replay wrapper with this sample probability (the replay scenario above, eager steps)")
    ap.add_argument("--out", required=True)
    args = ap.parse_args()

    import torch
    from quad_swarm_rl_amd import config as qcfg, native, parallel

    W, E = args.world, args.envs // args.world
    kw = REPLAY_KW if args.replay > 0 else KW
    N = kw["num_agents"]
    acts = replay_actions(args.steps, args.envs, N) if args.replay > 0 else global_actions(max(args.steps, args.graph), args.envs, N)
    ranks = [args.rank] if args.mode == "proc" else list(range(W))
    if args.mode == "proc":
        import torch.distributed as dist
        dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{args.port}", rank=args.rank, world_size=W)

    steppers, exs, act_dev = {}, {}, {}
    for r in ranks:
        cfg = qcfg.make_config(num_envs=E, seed=7, env_id_offset=r * E, precision="f32", write_rew_info=False, episode_sums=args.replay > 0, **kw)
        steppers[r] = native.Stepper(cfg, device=0)
        if args.replay > 0:
            steppers[r].replay_enable(args.replay)
        act_dev[r] = torch.as_tensor(acts[:, r * E * N:(r + 1) * E * N]).cuda().contiguous()
    if args.mode == "proc":
        r = args.rank
        exs[r] = parallel.ObsExchange(steppers[r], W, r, transport=args.transport, wire=args.wire, hold=bool(args.hold))
    else:
        for r in ranks:   # every ObsExchange builds its own endpoint: create them all first, then wire each to the others
            exs[r] = parallel.ObsExchange(steppers[r], W, r, transport=args.transport, wire=args.wire, hold=bool(args.hold), peers=[])
        for r in ranks:
            for q in ranks:
                if q != r:
                    exs[r].x.attach_local(exs[q].x)

    rec = {r: [] for r in ranks}

    def snap():
        rows = {r: exs[r].latest() for r in ranks}          # enqueue every rank's drain before the first host sync
        for r in ranks:
            rec[r].append(rows[r].cpu().numpy() if args.wire == "q8" else rows[r].float().cpu().numpy())   # q8: the raw wire bytes

    for r in ranks:
        exs[r].reset()
    snap()
    if args.replay > 0:
        for r in ranks:
            steppers[r].replay_set_active(None)
    stride = E * N * 4 * 4
    if args.graph:
        warm = 0
        while exs[ranks[0]].k < 2 or exs[ranks[0]].k & 1:   # capture() wants an even number (>= 2) of steps issued: take them here, rank by
            for r in ranks:                                  # rank (torch's capture begins with a device synchronize, which must not wait
                exs[r].step(act_dev[r].data_ptr())           # for a peer of this same process that has not been stepped yet)
            warm += 1
        for r in ranks:
            exs[r].capture([act_dev[r].data_ptr() + t * stride for t in range(args.graph)])
        for _ in range(args.replays):
            for r in ranks:
                exs[r].replay()
            snap()
    else:
        warm = 0
        for t in range(args.steps):
            for r in ranks:
                if args.replay > 0:
                    torch.cuda.synchronize()
                    plant_collisions(steppers[r], r * E)
                exs[r].step(act_dev[r].data_ptr() + t * stride)
            snap()
    torch.cuda.synchronize()
    verified = {}
    if args.verify:   # the gathered rows against an independent gather of the same rows through torch.distributed (ObsExchange.verify)
        for r in ranks:
            verified[r] = exs[r].verify() if args.mode == "proc" else (True, "")
    status = {r: exs[r].status() for r in ranks}
    extra = {}
    if args.replay > 0:
        for r in ranks:
            rs = steppers[r].replay_stats()
            extra[f"replayed{r}"] = int(rs["replayed"].sum())
            extra[f"buffer{r}"] = int(rs["buffer_len"].sum())
    np.savez(args.out, warm=warm, **extra, **{f"rows{r}": np.stack(rec[r]) for r in ranks}, **{f"err{r}": status[r]["error"] for r in ranks},
             **{f"pushes{r}": status[r]["pushes"] for r in ranks}, **{f"verify{r}": (1 if v[0] else 0) for r, v in verified.items()})
    for r in ranks:
        exs[r].close()
        steppers[r].close()
    if args.mode == "proc":
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
