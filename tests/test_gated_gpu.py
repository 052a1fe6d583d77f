"""Resident-state stepping (include/quadswarm.h: qs_gate_create / qs_step_gated / qs_gate_produce): ONE launch for k control steps that
waits per step for the step's actions and publishes its outputs, fed by a producer kernel on another stream.  The result must be the
open-loop multi-step launch's (qs_step_many: the same body without the gate; float64 to 1e-9 with every flag / mask / counter exact), and one-launch-per-step stepping's (which the
oracle parity suite pins against the reference: tests/test_hip_parity.py) with every flag / mask / counter exact in float64 - float64
and float32, closed loop (the producer waits for the previous step's outputs) and running ahead, across several
launches, across auto-resets, with plain steps in between; a missing producer is reported instead of hanging the GPU."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ARRAYS = ("obs", "reward", "done", "pos", "vel", "rot", "omega", "thrust_rot_damp", "thrust_cmds_damp", "ou_state", "goal", "col_pair_mask", "new_pair_mask",
          "unique_col_mask", "obst_new_mask", "room_new_mask", "counters", "tick", "obst_hit_idx", "ep_stats", "ep_counters")


def _pair(case, E, precision, seed=3, **over):
    from quad_swarm_rl_amd import config as qcfg, native
    from tests import test_hip_parity as thp
    kw = dict(thp.CASES[case], **over)
    cfg = qcfg.make_config(num_envs=E, seed=seed, precision=precision, **kw)
    return native.Stepper(cfg, device=0), native.Stepper(cfg, device=0), cfg


FLOATS = ("obs", "reward", "pos", "vel", "rot", "omega", "thrust_rot_damp", "thrust_cmds_damp", "ou_state", "goal", "ep_stats")


def _same(a, b, where, f64=True):
    """The gated kernel is its own instantiation of the multi-step body (instruction selection differs in the last bit): float64 - floats to
    1e-9, every discrete array exact; float32 - floats to 2e-3 * (1 + max|x|) over the 70-odd chaotic steps of the test (a batch consumed
    from the wrong ring slot would be an O(1) difference), done / tick exact."""
    for nm in ARRAYS:
        x, y = a.to_host(nm), b.to_host(nm)
        if nm in FLOATS:
            tol = 1e-9 if f64 else 2e-3 * (1.0 + np.abs(y).max())
            err = np.abs(x.astype(np.float64) - y.astype(np.float64)).max()
            assert err <= tol, f"{where}: {nm} differs by {err}"
        elif f64 or nm in ("done", "tick"):
            assert np.array_equal(x, y), f"{where}: {nm} differs"
    if f64:
        fa, fb = a.to_host("flags"), b.to_host("flags")
        assert np.array_equal(fa & 0xfff, fb & 0xfff) and np.array_equal(fa >> 16, fb >> 16), f"{where}: flags differ"


@pytest.mark.parametrize("case,precision,closed_loop", [("c2_n8_dw", "f64", True), ("c2_n8_dw", "f32", False), ("c4_n32_svs", "f64", False),
                                                        ("c3_n8_obst", "f32", True), ("c2_n8_dw", "f32", True)])
def test_gated_launch_equals_one_launch_per_step(case, precision, closed_loop):
    import torch
    E, K, launches = 40, 24, 3
    plain, gated, cfg = _pair(case, E, precision, ep_time=0.5)   # 50-step episodes: auto-resets inside the second launch
    if not gated.team:
        pytest.skip("team kernels only")
    T = E * cfg.num_agents
    dt = torch.float64 if precision == "f64" else torch.float32
    g = torch.Generator(device="cuda").manual_seed(5)
    table = (torch.rand((K * launches + 4, T, 4), device="cuda", generator=g, dtype=dt) * 2 - 1).contiguous()
    esz = table.element_size()
    gated.gate_create(ring_len=8, wg_per_group=2)
    info = gated.gate_info()
    assert info.ring_len == 8 and info.wg_per_group == 2 and info.groups == (info.workgroups + 1) // 2 and info.envs_per_workgroup == 64 // cfg.num_agents
    side, feed = torch.cuda.Stream(), torch.cuda.Stream()
    from quad_swarm_rl_amd import native
    stepwise = native.Stepper(cfg, device=0)
    plain.reset(); gated.reset(); stepwise.reset()
    torch.cuda.synchronize()
    f64 = precision == "f64"
    _same(plain, gated, "after reset", f64)
    step = 0
    for l in range(launches):
        gated.step_gated(K, stream=side)
        gated.gate_produce(table.data_ptr() + step * T * 4 * esz, K, K, closed_loop=closed_loop, stream=feed)
        gated.gate_wait(stream=side)     # (after the producer's launch: a wait for the gated kernel must never sit in front of its producer)
        plain.step_many(table[step].data_ptr(), K)     # the same multi-step kernel, open loop over the same batches: bit-identical arithmetic
        if precision == "f64" and l == 0:              # ... and one launch per control step: equal up to instruction selection between the two kernels
            for t in range(K):
                stepwise.step(table[step + t].data_ptr())
        torch.cuda.synchronize()
        if precision == "f64" and l == 0:
            for nm in ("done", "tick", "counters", "col_pair_mask", "new_pair_mask", "unique_col_mask", "obst_new_mask", "room_new_mask", "obst_hit_idx"):
                assert np.array_equal(stepwise.to_host(nm), gated.to_host(nm)), f"{nm} differs from one-launch-per-step stepping"
            for nm in ("obs", "reward", "pos", "vel", "rot", "omega"):
                np.testing.assert_allclose(gated.to_host(nm), stepwise.to_host(nm), rtol=0, atol=1e-8, err_msg=nm)
        step += K
        st = gated.gate_status()
        assert st["error"] == 0 and st["min_done_flag"] == step and st["min_act_flag"] == step, st
        _same(plain, gated, f"{case} {precision} after gated launch {l}", f64)
        if l == 0:   # a plain step in between: the state was written back, the gate keeps counting from where it was
            plain.step(table[-1].data_ptr()); gated.step(table[-1].data_ptr())
            torch.cuda.synchronize()
            _same(plain, gated, "plain step between gated launches", f64)
    plain.check_errors(); gated.check_errors()
    plain.close(); gated.close(); stepwise.close()


@pytest.mark.parametrize("spec,mode", [("off", "static_same_goal"), (None, "mix")])
def test_gated_generic_kernels_and_full_scenario_set(spec, mode, monkeypatch):
    """the other instantiations of the gated body: the generic library kernels (QS_SPEC=off) and the full scenario set (`mix`: per-env scenario
    state in LDS across the steps of a launch) - float64, closed loop, against the open-loop multi-step launch"""
    import torch
    from quad_swarm_rl_amd import config as qcfg, native
    from tests import test_hip_parity as thp
    if spec is not None:
        monkeypatch.setenv("QS_SPEC", spec)
    E, K = 24, 20
    kw = dict(thp.CASES["c2_n8_dw"], quads_mode=mode, ep_time=0.3)
    cfg = qcfg.make_config(num_envs=E, seed=9, precision="f64", **kw)
    plain, gated = native.Stepper(cfg, device=0), native.Stepper(cfg, device=0)
    assert gated.team and (spec is None) == bool(gated.specialized)
    T = E * cfg.num_agents
    g = torch.Generator(device="cuda").manual_seed(6)
    table = (torch.rand((2 * K, T, 4), device="cuda", generator=g, dtype=torch.float64) * 2 - 1).contiguous()
    gated.gate_create(ring_len=4, wg_per_group=1)
    side, feed = torch.cuda.Stream(), torch.cuda.Stream()
    plain.reset(); gated.reset()
    for l in range(2):   # the second launch crosses the 30-step episodes' auto-reset
        gated.step_gated(K, stream=side)
        gated.gate_produce(table[l * K].data_ptr(), K, K, closed_loop=True, stream=feed)
        gated.gate_wait(stream=side)
        plain.step_many(table[l * K].data_ptr(), K)
        torch.cuda.synchronize()
        assert gated.gate_status()["error"] == 0
        _same(plain, gated, f"QS_SPEC={spec} {mode} launch {l}", True)
    plain.close(); gated.close()


@pytest.mark.parametrize("case,wg_per_group", [("c2_n8_dw", 1), ("c2_n8_dw", 3), ("c4_n32_svs", 2)])
def test_a_concurrent_consumer_sees_each_steps_outputs(case, wg_per_group):
    """The visibility half of the protocol (include/quadswarm.h "Visibility"): a consumer kernel that runs WHILE the gated launch is resident,
    has seen done_flag >= s and executed an agent-scope acquire must read the observation rows and rewards of step s - not those of step s - 1
    from its own XCD's L2, not rows whose write-through is still in flight.  qs_gate_produce_verify is such a consumer (and the closed-loop
    producer of step s + 1): its per-step, per-group checksums must equal those of a one-launch-per-step twin fed the same actions."""
    import torch
    E, K = 96, 40
    twin, gated, cfg = _pair(case, E, "f32", ep_time=0.3)   # 30-step episodes: an auto-reset inside the launch
    if not gated.team:
        pytest.skip("team kernels only")
    N, D = cfg.num_agents, gated.obs_dim
    T = E * N
    g = torch.Generator(device="cuda").manual_seed(11)
    table = (torch.rand((K, T, 4), device="cuda", generator=g, dtype=torch.float32) * 2 - 1).contiguous()
    # the twin runs the SAME kernel (bit-identical arithmetic) one control step per launch, with a host synchronise after every step
    gated.gate_create(ring_len=4, wg_per_group=wg_per_group)
    twin.gate_create(ring_len=4, wg_per_group=wg_per_group)
    info = gated.gate_info()
    rows_wg = info.envs_per_workgroup * N
    sums = torch.zeros((K, info.groups), dtype=torch.int64, device="cuda")
    side, feed = torch.cuda.Stream(), torch.cuda.Stream()
    twin.reset(); gated.reset()
    torch.cuda.synchronize()
    for rep in range(3):   # (three resident launches back to back: 3 * K hand-overs per group)
        sums.zero_()
        torch.cuda.synchronize()
        gated.step_gated(K, stream=side)
        gated.gate_produce_verify(table.data_ptr(), K, K, sums.data_ptr(), stream=feed)
        gated.gate_wait(stream=side)
        want = torch.zeros((K, info.groups), dtype=torch.int64, device="cuda")
        for t in range(K):
            twin.step_gated(1, stream=side)
            twin.gate_produce(table[t].data_ptr(), 1, 1, closed_loop=False, stream=feed)
            twin.gate_wait(stream=side)
            torch.cuda.synchronize()
            words = torch.cat((twin.tensor("obs").reshape(T, D).view(torch.int32).to(torch.int64) & 0xffffffff,
                               (twin.tensor("reward").reshape(T, 1).view(torch.int32).to(torch.int64) & 0xffffffff)), dim=1).sum(dim=1)   # per drone
            per_wg = torch.zeros(info.workgroups * rows_wg, dtype=torch.int64, device="cuda")
            per_wg[:T] = words
            per_wg = per_wg.reshape(info.workgroups, rows_wg).sum(dim=1)
            pad = torch.zeros(info.groups * wg_per_group, dtype=torch.int64, device="cuda")
            pad[:info.workgroups] = per_wg
            want[t] = pad.reshape(info.groups, wg_per_group).sum(dim=1)
        torch.cuda.synchronize()
        st = gated.gate_status()
        assert st["error"] == 0 and twin.gate_status()["error"] == 0, st
        assert int((want != 0).sum()) == K * info.groups          # (a checksum of nothing would also "agree")
        bad = (sums != want).nonzero()
        assert bad.numel() == 0, f"launch {rep}: {bad.shape[0]} of {K * info.groups} (step, group) checksums differ, first at {bad[0].tolist()}"
    twin.close(); gated.close()


def test_missing_producer_is_reported_not_hung(monkeypatch):
    import time
    import torch
    monkeypatch.setenv("QS_GATE_TIMEOUT_MS", "40")
    a, b, cfg = _pair("c2_n8_dw", 16, "f32")
    a.close()
    b.gate_create(ring_len=4, wg_per_group=1)
    b.reset()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    b.step_gated(50)                     # nobody feeds the ring
    torch.cuda.synchronize()
    assert time.perf_counter() - t0 < 5.0     # ONE bounded wait, then the launch stops waiting
    st = b.gate_status()
    assert st["error"] & 1 and st["min_done_flag"] == 50
    b.close()


def test_gate_is_refused_where_it_cannot_work():
    from quad_swarm_rl_amd import config as qcfg, native
    from tests import test_hip_parity as thp
    cfg = qcfg.make_config(num_envs=8, seed=1, precision="f32", episode_sums=True, **thp.CASES["c2_n8_dw"])
    st = native.Stepper(cfg, device=0)
    with pytest.raises(native.QsError):
        st.step_gated(4)                                  # no gate yet
    st.replay_enable(0.5)
    with pytest.raises(native.QsError):
        st.gate_create()                                  # not with the device-side replay wrapper
    st.close()
