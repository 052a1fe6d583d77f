"""The N>1 path on CPU: world_size-2 `gloo` processes, each stepping its own shard of environments (here with the
oracle standing in for the GPU stepper: this test is about sharding + the per-step observation all-gather, not the
kernels) and gathering the observations every rollout step.  The gathered tensor must equal the un-sharded run."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KW = dict(num_agents=4, neighbor_visible_num=2, neighbor_obs_type="pos_vel", use_downwash=True, use_numba=True,
          collision_falloff_radius=4.0)
TOTAL_ENVS, STEPS, SEED = 6, 12, 31


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _run_shard(lo, hi):
    from oracle import oracle as orc
    from quad_swarm_rl_amd import config as qcfg
    cfg = qcfg.make_config(num_envs=hi - lo, seed=SEED, env_id_offset=lo, **KW)
    envs = [orc.OracleEnv(cfg, env_global_id=e) for e in range(lo, hi)]
    return cfg, envs


def _worker(rank, world, port, overlap, out_dir):
    sys.path.insert(0, REPO)
    from quad_swarm_rl_amd import parallel
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = parallel.shard_range(TOTAL_ENVS, world, rank)
    cfg, envs = _run_shard(lo, hi)
    n, d = cfg.num_agents, envs[0].obs_dim
    local = torch.zeros(((hi - lo) * n, d), dtype=torch.float64)
    gather = parallel.ObsGather(local, overlap=overlap)
    rng = np.random.RandomState(99)
    actions = rng.uniform(-1, 1, size=(STEPS, TOTAL_ENVS, n, 4))     # every rank draws the same global action tape
    local.copy_(torch.from_numpy(np.concatenate([e.reset() for e in envs])))
    results = []
    pending = None
    for t in range(STEPS):
        obs = np.concatenate([e.step(actions[t, lo + k])[0] for k, e in enumerate(envs)])
        local.copy_(torch.from_numpy(obs))
        if overlap:
            if pending is not None:
                results.append(gather.result(pending).clone())
            pending = gather.gather()
        else:
            results.append(gather.gather().clone())
    if overlap:
        results.append(gather.result(pending).clone())
        gather.drain()
    if rank == 0:
        torch.save(torch.stack(results), os.path.join(out_dir, f"gathered_{int(overlap)}.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("overlap", [False, True])
def test_two_rank_gather_matches_unsharded(tmp_path, overlap):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, overlap, str(tmp_path)), nprocs=2, join=True)
    got = torch.load(os.path.join(str(tmp_path), f"gathered_{int(overlap)}.pt")).numpy()
    sys.path.insert(0, REPO)
    cfg, envs = _run_shard(0, TOTAL_ENVS)
    rng = np.random.RandomState(99)
    actions = rng.uniform(-1, 1, size=(STEPS, TOTAL_ENVS, cfg.num_agents, 4))
    for e in envs:
        e.reset()
    for t in range(STEPS):
        ref = np.concatenate([e.step(actions[t, k])[0] for k, e in enumerate(envs)])
        np.testing.assert_array_equal(got[t], ref)


def test_shard_range():
    from quad_swarm_rl_amd import parallel
    assert [parallel.shard_range(4096, 8, r) for r in (0, 7)] == [(0, 512), (3584, 4096)]
    with pytest.raises(ValueError):
        parallel.shard_range(10, 4, 0)


def _handles_worker(rank, world, port, fail_rank, out_dir):
    sys.path.insert(0, REPO)
    from quad_swarm_rl_amd import native, parallel
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    blob = None if rank == fail_rank else bytes([rank]) * 8
    try:
        got = parallel.gather_window_handles(blob, world)
        res = ["ok"] + [b.hex() for b in got]
    except native.QsError as exc:
        res = ["error", str(exc)]
    dist.barrier()   # nobody hangs in the collective, whoever failed
    with open(os.path.join(out_dir, f"handles_{rank}.txt"), "w") as f:
        f.write("\n".join(res))
    dist.destroy_process_group()


@pytest.mark.parametrize("fail_rank", [-1, 0, 1])
def test_window_handle_exchange_fails_on_every_rank_or_on_none(tmp_path, fail_rank):
    """The multi-GPU exchange maps its peers' windows at start-up (parallel.ObsExchange).  A rank whose endpoint could not be created must
    neither hang the others in the handle collective nor leave them on a different transport: every rank raises, naming the rank."""
    port = _free_port()
    mp.spawn(_handles_worker, args=(2, port, fail_rank, str(tmp_path)), nprocs=2, join=True)
    res = [open(os.path.join(str(tmp_path), f"handles_{r}.txt")).read().split("\n") for r in range(2)]
    if fail_rank < 0:
        assert res[0] == res[1] == ["ok", "00" * 8, "01" * 8]
    else:
        assert res[0][0] == res[1][0] == "error" and res[0][1] == res[1][1] and f"[{fail_rank}]" in res[0][1]
