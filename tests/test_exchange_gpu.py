"""Observation exchange between env shards (include/quadswarm_exchange.h, quad-swarm-rl_amd/parallel.py): the rows every rank ends
up with equal the un-sharded stepper's rows - bit for bit on the float32 wire, equal to the round-to-nearest-even bfloat16 of them
on the bf16 wire - with two PROCESSES sharing the one GPU of the test box (hipIpc-mapped windows), with two endpoints in one
process, eager and as a captured HIP graph; the float32 -> bf16 converter equals torch's; bounded waits report instead of hanging."""
import os
import subprocess
import sys
import tempfile

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKER = os.path.join(REPO, "tests", "xchg_worker.py")


def _expected_rows(total_envs, wire, action_batches):
    """rows of the un-sharded stepper after the reset and after each action batch (an int t = batch t of the global action set)"""
    import torch
    from quad_swarm_rl_amd import config as qcfg, native, parallel
    from tests import xchg_worker as xw
    cfg = qcfg.make_config(num_envs=total_envs, seed=7, precision="f32", write_rew_info=False, **xw.KW)
    st = native.Stepper(cfg, device=0)
    acts = torch.as_tensor(xw.global_actions(max(action_batches) + 1 if action_batches else 1, total_envs, cfg.num_agents)).cuda()
    obs = st.tensor("obs")

    q8 = native.wire_q8_layout(cfg, st.obs_dim)

    def rows():
        torch.cuda.synchronize()
        if wire == "q8":   # the wire bytes of the un-sharded rows by the plain-torch specification of the format
            return parallel.quantize_rows_reference(obs, "q8", q8).cpu().numpy()
        r = obs.to(torch.bfloat16).float() if wire == "bf16" else obs.clone()
        return r.cpu().numpy()

    st.reset()
    out = [rows()]
    for t in action_batches:
        st.step(acts[t].data_ptr())
        out.append(rows())
    st.close()
    return out


def _run_workers(mode, world, wire, graph, steps=10, replays=3, envs=16, timeout=300, transport="peer", hold=1, verify=0, replay=0.0):
    env = dict(os.environ, GPU_MAX_HW_QUEUES="8", HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    with tempfile.TemporaryDirectory() as td:
        port = 29600 + (os.getpid() % 300)
        procs, outs = [], []
        for r in (range(world) if mode == "proc" else [0]):
            out = os.path.join(td, f"r{r}.npz")
            outs.append(out)
            cmd = [sys.executable, WORKER, "--mode", mode, "--rank", str(r), "--world", str(world), "--port", str(port), "--wire", wire, "--envs", str(envs),
                   "--steps", str(steps), "--graph", str(graph), "--replays", str(replays), "--transport", transport, "--hold", str(hold), "--verify", str(verify), "--replay", str(replay), "--out", out]
            procs.append(subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
        logs = []
        for p in procs:
            try:
                log, _ = p.communicate(timeout=timeout)
            except subprocess.TimeoutExpired:
                for q in procs:
                    q.kill()
                raise
            logs.append(log)
        for p, log in zip(procs, logs):
            assert p.returncode == 0, log[-3000:]
        return [dict(np.load(o)) for o in outs]


def _check(results, mode, world, wire, graph, steps, replays, envs=16):
    warm = int(results[0]["warm"])
    if graph:   # reset, `warm` eager steps on action batch 0, then `replays` x the captured batches 0..graph-1; rows recorded after each replay
        seq = [0] * warm + list(range(graph)) * replays
        exp = _expected_rows(envs, wire, seq)
        picks = [0] + [warm + graph * (i + 1) for i in range(replays)]
    else:
        exp = _expected_rows(envs, wire, list(range(steps)))
        picks = list(range(steps + 1))
    ranks = range(world)
    for res in results:
        for r in ranks:
            if f"rows{r}" not in res:
                continue
            assert int(res[f"err{r}"]) == 0, f"exchange status of rank {r}: {int(res[f'err{r}'])}"
            got = res[f"rows{r}"]
            assert got.shape[0] == len(picks)
            for i, k in enumerate(picks):
                assert got[i].shape == exp[k].shape
                assert np.array_equal(got[i], exp[k]), f"{mode} wire={wire} graph={graph}: rank {r}, record {i} differs (max {np.abs(got[i] - exp[k]).max()})"


@pytest.mark.parametrize("wire", ["f32", "bf16", "q8"])
@pytest.mark.parametrize("graph", [0, 6])
def test_two_processes_on_one_gpu_gather_equals_unsharded(wire, graph):
    """north_star's sharding with one process per rank: both processes map each other's windows with hipIpcOpenMemHandle."""
    res = _run_workers("proc", 2, wire, graph)
    _check(res, "proc", 2, wire, graph, 10, 3)


@pytest.mark.parametrize("wire,graph,hold", [("f32", 0, 1), ("bf16", 6, 1), ("bf16", 0, 0), ("f32", 6, 0), ("q8", 0, 1), ("q8", 6, 0)])
def test_two_processes_fused_push_from_the_step_kernel(wire, graph, hold):
    """the FUSED transport (qs_set_obs_exchange): the team step kernels store their rows into both processes' windows themselves;
    hold = 1: reader mode (wait / release launches around the reads), hold = 0: the step launch acknowledges on its own"""
    res = _run_workers("proc", 2, wire, graph, transport="fused", hold=hold)
    _check(res, "proc-fused", 2, wire, graph, 10, 3)


@pytest.mark.parametrize("wire,graph", [("f32", 0), ("bf16", 0)])
def test_two_endpoints_in_one_process(wire, graph):
    """(eager launches only: two captured graphs of ONE process that poll each other's flags are not guaranteed to run concurrently -
    the graph form is covered with one process per rank above, which is the deployment)"""
    res = _run_workers("local", 2, wire, graph)
    _check(res, "local", 2, wire, graph, 10, 3)


def test_two_processes_gather_with_the_device_side_replay_wrapper():
    """train_local.sh trains with --replay_buffer_sample_prob=0.75 (reference train_local.sh:9): with the replay kernel behind every step the
    exchange sends what it leaves in the library's buffer (ObsExchange source="obs") - restored checkpoints and the filed checkpoint's
    observation included.  Two processes, 12 + 12 environments, 620 steps with planted collisions: every rank's gathered rows equal the
    un-sharded replay-wrapped stepper's, and episodes were in fact replayed."""
    import torch
    from quad_swarm_rl_amd import config as qcfg, native
    from tests import xchg_worker as xw
    envs, steps, prob = 24, 620, 0.75
    res = _run_workers("proc", 2, "f32", 0, steps=steps, envs=envs, timeout=600, replay=prob)
    cfg = qcfg.make_config(num_envs=envs, seed=7, precision="f32", write_rew_info=False, episode_sums=True, **xw.REPLAY_KW)
    st = native.Stepper(cfg, device=0)
    st.replay_enable(prob)
    acts = torch.as_tensor(xw.replay_actions(steps, envs, cfg.num_agents)).cuda()
    obs = st.tensor("obs")
    st.reset()
    torch.cuda.synchronize()
    exp = [obs.cpu().numpy().copy()]
    st.replay_set_active(None)
    for t in range(steps):
        torch.cuda.synchronize()
        xw.plant_collisions(st, 0)
        st.step(acts[t].data_ptr())
        torch.cuda.synchronize()
        exp.append(obs.cpu().numpy().copy())
    rs = st.replay_stats()
    assert int(rs["replayed"].sum()) >= 2 and int(rs["buffer_len"].sum()) >= 2, (rs["replayed"], rs["buffer_len"])   # the scenario does replay
    st.close()
    assert sum(int(r[f"replayed{k}"]) for k, r in enumerate(res)) == int(rs["replayed"].sum())
    for k, r in enumerate(res):
        assert int(r[f"err{k}"]) == 0
        got = r[f"rows{k}"]
        assert got.shape[0] == steps + 1
        for i in range(steps + 1):
            assert np.array_equal(got[i], exp[i]), f"rank {k}: gathered rows after step {i} differ from the un-sharded replay-wrapped stepper"


def test_two_processes_verify_against_an_independent_gather():
    """ObsExchange.verify(): the rows in the windows (fused epilogue, q8 wire) equal what torch.distributed gathers from the same float32 rows"""
    res = _run_workers("proc", 2, "q8", 0, steps=4, transport="fused", hold=1, verify=1)
    assert all(int(r[f"verify{k}"]) == 1 for k, r in enumerate(res))


def test_q8_pack_unpack_and_error_bound():
    """qs_obs_pack_rows(QS_WIRE_Q8) == the plain-torch specification bit for bit (C2 and C3 row shapes, an unaligned row count);
    qs_obs_unpack_rows brings the neighbour block back to within clip / 254 and the other columns to their bf16 rounding"""
    import torch
    from quad_swarm_rl_amd import config as qcfg, native, parallel
    g = torch.Generator(device="cuda").manual_seed(5)
    for kw, D in ((dict(num_agents=8, neighbor_visible_num=6, neighbor_obs_type="pos_vel"), 54),
                  (dict(num_agents=8, neighbor_visible_num=2, neighbor_obs_type="pos_vel", use_obstacles=True, obst_density=0.2, obst_size=0.6,
                        quads_mode="o_static_same_goal", obs_repr="xyz_vxyz_R_omega_floor"), 40)):
        cfg = qcfg.make_config(num_envs=1, **kw)
        q8 = native.wire_q8_layout(cfg, D)
        rb = parallel.wire_row_bytes(D, "q8", q8)
        assert rb == (72 if D == 54 else 68)
        for R in (4096, 777):
            x = (torch.rand((R, D), device="cuda", generator=g) * 2 - 1) * 12.0     # beyond the clip range too
            x[:, q8.q0:q8.q1:6] = torch.round(x[:, q8.q0:q8.q1:6] * 12.7) / 12.7 + 0.5 / 12.7   # values on / near rounding ties
            dst = torch.empty((R, rb), dtype=torch.uint8, device="cuda")
            parallel.pack_rows(x, dst, q8=q8)
            want = parallel.quantize_rows_reference(x, "q8", q8)
            torch.cuda.synchronize()
            assert torch.equal(dst, want), (D, R, (dst != want).sum().item())
            back = parallel.unpack_rows(dst, D, "q8", q8)
            torch.cuda.synchronize()
            clip = torch.tensor([q8.clip[a % 6] for a in range(q8.q1 - q8.q0)], device="cuda")
            nb = x[:, q8.q0:q8.q1].clamp(-clip, clip)
            assert ((back[:, q8.q0:q8.q1] - nb).abs() <= clip / 254 * 1.0001 + 1e-6).all()
            other = torch.cat([x[:, :q8.q0], x[:, q8.q1:]], 1)
            assert torch.equal(torch.cat([back[:, :q8.q0], back[:, q8.q1:]], 1), other.to(torch.bfloat16).float())


def test_four_ranks_in_one_process_bf16():
    res = _run_workers("local", 4, "bf16", 0, steps=6)
    _check(res, "local", 4, "bf16", 0, 6, 0)


def test_pack_bf16_equals_torch_rounding():
    import torch
    from quad_swarm_rl_amd import parallel
    g = torch.Generator(device="cuda").manual_seed(3)
    x = torch.randn(100003, device="cuda", generator=g) * 37.0
    bits = torch.randint(-2 ** 31, 2 ** 31 - 1, (50000,), device="cuda", generator=g, dtype=torch.int64).to(torch.int32)
    special = torch.tensor([0.0, -0.0, 1.0, 1.00390625, 1.01171875, float("inf"), -float("inf"), 3.3895314e38, 1e-40, -1e-45, 65504.0], device="cuda")
    src = torch.cat([x, bits.view(torch.float32), special]).contiguous()
    src = src[torch.isfinite(src) | torch.isinf(src)]   # NaN payloads are not part of the contract (obs are finite: NaN rewards raise)
    src = src.contiguous()
    for dt in (torch.bfloat16, torch.float32):
        dst = torch.empty(src.shape, dtype=dt, device="cuda")
        parallel.pack_rows(src, dst)
        torch.cuda.synchronize()
        want = src.to(dt)
        assert torch.equal(dst.view(torch.int16 if dt == torch.bfloat16 else torch.int32), want.view(torch.int16 if dt == torch.bfloat16 else torch.int32))
    odd = src[1:4098]                                    # an address that is not 16-byte aligned takes the element-wise path
    dst = torch.empty(odd.shape, dtype=torch.bfloat16, device="cuda")
    parallel.pack_rows(odd, dst)
    torch.cuda.synchronize()
    assert torch.equal(dst.view(torch.int16), odd.to(torch.bfloat16).view(torch.int16))


@pytest.mark.parametrize("wire", ["bf16", "q8"])
@pytest.mark.parametrize("transport", ["fused", "peer", "rccl"])
def test_world1_graph_capture_matches_plain_stepping(transport, wire):
    """world size 1 (what bench.py --force-gather runs on a 1-GPU box): [step -> exchange] x 8 as one HIP graph, replayed, equals
    the plain stepper on the same actions; the redirected observation output leaves qs_buffers.obs untouched."""
    import torch
    from quad_swarm_rl_amd import config as qcfg, native, parallel
    from tests import xchg_worker as xw
    E, G = 12, 8
    cfg = qcfg.make_config(num_envs=E, seed=7, precision="f32", write_rew_info=False, **xw.KW)
    acts = torch.as_tensor(xw.global_actions(G, E, cfg.num_agents)).cuda()
    stride = acts[0].numel() * 4
    ref = native.Stepper(cfg, device=0)
    st = native.Stepper(cfg, device=0)
    ex = parallel.ObsExchange(st, 1, 0, transport=transport, wire=wire)
    ex.reset()
    ref.reset()
    untouched = st.tensor("obs").clone()
    ex.capture([acts.data_ptr() + t * stride for t in range(G)])
    warm = ex.k - 1
    for _ in range(warm):
        ref.step(acts[0].data_ptr())
    for rep in range(3):
        ex.replay()
        for t in range(G):
            ref.step(acts[t].data_ptr())
        torch.cuda.synchronize()
        want = ref.tensor("obs")
        assert torch.equal(ex.latest(), parallel.quantize_rows_reference(want, wire, ex.q8)), (transport, rep)
        assert torch.equal(ex.local_rows(), want)
        assert ex.verify()[0]
        if wire == "q8":   # the consumer's float32 view: neighbour block within clip / 254 of the float32 rows
            f = ex.latest_f32()
            torch.cuda.synchronize()
            assert (f[:, 18:54] - want[:, 18:54]).abs().max().item() <= 10.0 / 254 * 1.0001
    assert ex.status()["error"] == 0
    if transport != "fused":   # (the fused transport leaves the rows where the library puts them)
        assert torch.equal(st.tensor("obs"), untouched)
    ex.close()
    st.close()
    ref.close()


def test_missing_peer_times_out_and_reports():
    """a rank whose peer never pushes: the bounded wait raises the status word instead of hanging the GPU"""
    import torch
    from quad_swarm_rl_amd import parallel
    os.environ["QS_XCHG_TIMEOUT_MS"] = "50"
    try:
        a = parallel.PeerExchange(64, 8, 2, 0, wire="f32")
        b = parallel.PeerExchange(64, 8, 2, 1, wire="f32")
    finally:
        del os.environ["QS_XCHG_TIMEOUT_MS"]
    a.attach_local(b)
    b.attach_local(a)
    a.staging(0).fill_(1.0)
    torch.cuda.synchronize()
    a.push(a.staging_ptr(0))
    a.wait()            # rank 1 never pushed
    torch.cuda.synchronize()
    assert a.status()["error"] & 2
    assert a.status()["pushes"] == 1 and a.status()["waits"] == 1
    assert torch.equal(b.gathered(1)[:64], a.staging(0))    # the rows themselves did arrive in rank 1's window
    a.close()
    b.close()


def test_obs_target_is_refused_with_device_replay():
    from quad_swarm_rl_amd import config as qcfg, native
    from tests import xchg_worker as xw
    cfg = qcfg.make_config(num_envs=4, seed=1, precision="f32", episode_sums=True, **xw.KW)
    st = native.Stepper(cfg, device=0)
    st.replay_enable(0.5)
    with pytest.raises(native.QsError):
        st.set_obs_target(st.ptr("rew_info"))
    st.close()


def test_exchange_timeout_is_sticky_and_raises():
    """a rank whose peer is late must not go on consuming stale rows silently - the timeout is sticky, check() / the next step() raise"""
    import torch
    from quad_swarm_rl_amd import config as qcfg, native, parallel
    from tests import xchg_worker as xw
    os.environ["QS_XCHG_TIMEOUT_MS"] = "40"
    try:
        cfg = qcfg.make_config(num_envs=8, seed=7, precision="f32", write_rew_info=False, **xw.KW)
        st = native.Stepper(cfg, device=0)
        ex = parallel.ObsExchange(st, 2, 0, transport="peer", wire="bf16", peers=[], hold=True)
        silent = parallel.PeerExchange(st.T, st.obs_dim, 2, 1, wire="bf16")     # rank 1 exists but never pushes
    finally:
        del os.environ["QS_XCHG_TIMEOUT_MS"]
    ex.x.attach_local(silent)
    silent.attach_local(ex.x)
    ex.reset()                         # the wait for rank 1's rows gives up after 40 ms
    torch.cuda.synchronize()
    with pytest.raises(native.QsError, match="timed out"):
        ex.check()
    acts = torch.zeros((st.T, 4), device="cuda")
    with pytest.raises(native.QsError, match="timed out"):
        ex.step(acts.data_ptr())       # sticky: no further steps on a failed exchange
    with pytest.raises(native.QsError):
        ex.latest()
    ex.close(); silent.close(); st.close()


@pytest.mark.parametrize("wire,transport", [("bf16", "rccl"), ("q8", "rccl"), ("q8", "peer")])
def test_batched_env_gathers_with_the_published_recipe_flags(wire, transport):
    """--quads_gather_obs together with --replay_buffer_sample_prob=0.75 (what train_local.sh trains with) at world size 1: the env steps,
    gathered_obs() is the wire form of the rows step() returned, check_exchange() stays quiet"""
    import torch
    from quad_swarm_rl_amd import parallel
    from quad_swarm_rl_amd.sf_env import BatchedQuadSwarm
    env = BatchedQuadSwarm(16, device=0, seed=3, replay_buffer_sample_prob=0.75, num_gpus=1, gather_obs=True, obs_wire=wire, obs_transport=transport,
                           num_agents=8, neighbor_visible_num=6, neighbor_obs_type="pos_vel", use_numba=True, quads_mode="mix", ep_time=0.4)
    assert env.obs_transport == transport and env.vec.exchange.source == "obs"   # (peer: self-check and verify() passed, else it would have fallen back)
    obs, _ = env.reset()
    g = torch.Generator(device="cuda").manual_seed(0)
    for t in range(90):   # two episode ends inside
        act = torch.rand((env.num_agents, 4), device="cuda", generator=g) * 2 - 1
        out = env.step(act)
        rows = out[0]["obs"]
        got = env.gathered_obs()
        torch.cuda.synchronize()
        assert torch.equal(got, parallel.quantize_rows_reference(rows, wire, env.vec.exchange.q8)), t
    f = env.gathered_obs_f32()
    torch.cuda.synchronize()
    assert f.shape == rows.shape and (f[:, :18] - rows[:, :18]).abs().max().item() < 0.05
    env.check_exchange()
    env.close()
