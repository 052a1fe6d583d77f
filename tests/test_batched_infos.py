"""sf_env.assemble_batched_infos (the per-agent `infos` of a batch of finished episodes, built per env and copied) against the
straightforward per-agent construction from env.assemble_episode_extra_stats - the function the single-env facade uses and that
tests/test_facade_gpu.py compares with the reference's dict keys."""
import time

import numpy as np
import pytest


def naive(finished, n, sums, eps, cnt, scen_ids, rs, obst_density, obst_size, approx, keys, use_obstacles, ep_steps, annealed, infos):
    from quad_swarm_rl_amd import config as qcfg
    from quad_swarm_rl_amd.env import assemble_episode_extra_stats
    for f, e in enumerate(finished):
        sl = slice(f * n, (f + 1) * n)
        scenario_name = qcfg.SCENARIO_CLASS_NAMES[int(scen_ids[f])]
        if rs is not None and rs["ep_was_replay"][e]:
            env_stats = [{"num_collisions_replay": int(cnt[0, f]), "num_collisions_obst_replay": int(cnt[7, f])} for _ in range(n)]
        else:
            env_stats = assemble_episode_extra_stats(eps[:, sl], cnt[:, f], scenario_name[9:], n, use_obstacles)
        if rs is not None:
            ep, rp, nb = int(rs["episodes"][e]), int(rs["replayed"][e]), int(rs["buffer_len"][e])
            replay_stats = {"replay/replay_rate": rp / ep, "replay/new_episode_rate": (ep - rp) / ep, "replay/replay_buffer_size": nb,
                            "replay/avg_replayed": (int(rs["replayed_sum"][e]) / nb) if nb else 0,
                            "replay/obst_density": float(obst_density[e]), "replay/obst_size": float(obst_size[e])}
        count = float((int(rs["ep_steps"][e]) if rs is not None else ep_steps) * n)
        a1, a2 = sums[17:21, sl].sum(axis=1) / count, sums[21:25, sl].sum(axis=1) / count
        a_std = np.sqrt(np.maximum(a2 - a1 * a1, 0.0))
        for k in range(n):
            i, col = e * n + k, f * n + k
            cum = {key: float(sums[j, col]) for j, key in enumerate(keys) if use_obstacles or j < 15}
            true_reward = cum["rewraw_main"] + 1000 * cum.get("rewraw_quadcol", 0)
            cum["rewraw_main"] = true_reward
            extra = dict(env_stats[k])
            if rs is not None:
                extra.update(replay_stats)
            extra.update(cum)
            extra["z_approx_total_training_steps"] = approx
            for rew_key in ("rew_pos", "rew_crash"):
                extra[f"{scenario_name}/{rew_key}"] = cum[rew_key]
            for q in range(4):
                extra[f"z_action{q}_mean"], extra[f"z_action{q}_std"] = float(a1[q]), float(a_std[q])
            for key, val in annealed:
                extra[key] = val
            infos[i]["true_reward"] = true_reward
            infos[i]["episode_extra_stats"] = extra


def synthetic(E, n, F, use_obstacles, replay, seed):
    from quad_swarm_rl_amd import config as qcfg
    rng = np.random.RandomState(seed)
    finished = np.sort(rng.choice(E, F, replace=False))
    sums = rng.normal(size=(25, F * n)); sums[21:25] = np.abs(sums[21:25]) * 50 + 30
    eps = np.concatenate([np.abs(rng.normal(size=(3, F * n))), rng.randint(0, 2, size=(3, F * n)).astype(np.float64)])
    cnt = rng.randint(0, 9, size=(11, F))
    scen_ids = rng.randint(0, len(qcfg.SCENARIO_CLASS_NAMES), size=F)
    rs = None
    if replay:
        rs = dict(episodes=rng.randint(1, 9, size=E), replayed=rng.randint(0, 2, size=E), buffer_len=rng.randint(0, 5, size=E), replayed_sum=rng.randint(0, 20, size=E),
                  ep_was_replay=rng.randint(0, 2, size=E), ep_steps=rng.randint(100, 1501, size=E))
    return finished, n, sums, eps, cnt, scen_ids, rs, rng.uniform(0.05, 0.2, size=E), rng.uniform(0.3, 0.6, size=E), 123456, qcfg.REW_INFO_KEYS, use_obstacles, 1501, \
        [("z_anneal_quadcol_bin", 1.25)]


def test_batched_infos_equal_the_per_agent_construction():
    from quad_swarm_rl_amd import sf_env
    for use_obstacles in (False, True):
        for replay in (False, True):
            args = synthetic(64, 8, 40, use_obstacles, replay, seed=3 + use_obstacles + 2 * replay)
            a = [{} for _ in range(64 * 8)]
            b = [{} for _ in range(64 * 8)]
            sf_env.assemble_batched_infos(*args, a)
            naive(*args, b)
            assert a == b
            done = {int(e) * 8 + k for e in args[0] for k in range(8)}
            assert all((i in done) == bool(a[i]) for i in range(64 * 8))
            some = a[int(args[0][0]) * 8]
            assert list(some["episode_extra_stats"]) == list(b[int(args[0][0]) * 8]["episode_extra_stats"])   # same key order too


def test_batched_infos_cost_for_a_full_batch():
    """All 1024 x 8 agents finishing on the same step (the batched env's episodes end together unless replay restarts them): the
    assembly has to stay far below a second; measured here ~0.1 s, amortised over 1500 control steps."""
    from quad_swarm_rl_amd import sf_env
    args = synthetic(1024, 8, 1024, False, True, seed=9)
    infos = [{} for _ in range(8192)]
    t0 = time.perf_counter()
    sf_env.assemble_batched_infos(*args, infos)
    dt = time.perf_counter() - t0
    t0 = time.perf_counter()
    naive(*args, [{} for _ in range(8192)])
    dt_naive = time.perf_counter() - t0
    print(f"assemble_batched_infos: {dt * 1e3:.0f} ms for 8192 agents (per-agent construction: {dt_naive * 1e3:.0f} ms)")
    assert dt < 1.0


def test_lazy_episode_infos_equal_the_eager_list_and_cost_nothing_until_read():
    """sf_env.EpisodeInfos (what BatchedQuadSwarm.step returns): same dicts as the eager assembly, built per finished environment on
    first access; creating it for a full 1024 x 8 batch takes milliseconds (the eager list cost 60-120 ms on that step)."""
    import copy
    import pickle
    from quad_swarm_rl_amd import sf_env
    args = synthetic(64, 8, 40, True, True, seed=11)
    eager = [{} for _ in range(64 * 8)]
    sf_env.assemble_batched_infos(*args, eager)
    lazy = sf_env.EpisodeInfos(64 * 8, 8, sf_env.EpisodeInfoBuilder(*args))
    assert isinstance(lazy, list) and len(lazy) == 512 and bool(lazy)
    e0 = int(args[0][0])
    assert lazy[e0 * 8 + 3] == eager[e0 * 8 + 3] and len(lazy._built) == 1          # one environment built, on demand
    assert lazy[-1] == eager[-1] and lazy[5:9] == eager[5:9]
    assert lazy == eager and list(lazy) == eager and [d for d in lazy] == eager
    assert pickle.loads(pickle.dumps(lazy)) == eager and copy.deepcopy(lazy) == eager
    assert lazy.finished_agents() == [i for i, d in enumerate(eager) if d]
    with pytest.raises(IndexError):
        lazy[512]
    args = synthetic(1024, 8, 1024, False, True, seed=9)
    t0 = time.perf_counter()
    lazy = sf_env.EpisodeInfos(8192, 8, sf_env.EpisodeInfoBuilder(*args))
    dt = time.perf_counter() - t0
    t0 = time.perf_counter()
    one = lazy[4097]
    dt_one = time.perf_counter() - t0
    print(f"EpisodeInfos for 8192 finished agents: {dt * 1e3:.2f} ms to create, {dt_one * 1e6:.0f} us for the first dict of an environment")
    assert dt < 0.010 and "episode_extra_stats" in one
