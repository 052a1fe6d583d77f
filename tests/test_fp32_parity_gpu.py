"""What a user of the float32 (production) stepper actually gets, beyond one teacher-forced step through the specialised object:

  (a) the GENERIC float32 kernels - what qs_create falls back to without hipcc / a writable cache (QS_SPEC=off) - teacher-forced
      against the oracle, team and single-wave flavours: 1e-5 on floats, every discrete output exact;
  (b) float32 FREE-RUNNING against the float64 oracle over event-free windows (the method of the reference's
      gym_art/quadrotor_multi/tests/test_numba_opt.py:59-119: two implementations, identical injected noise, compare trajectories):
      hover-ish actions from the spawn, up to 100 control steps, the window of an environment ends at its first event (collision,
      floor / wall / ceiling contact, obstacle hit, proximity, episode end, or a near-tie in the neighbour ranking that float32 cannot
      resolve) - inside it: 1e-4 on state / obs / reward, flags and counters exact;
  (c) float32 vs float64 kernels over one FULL episode of BASELINE configs[1] at full size (8 x 1024, same seeds => same noise):
      the episode statistics the reference reports (quadrotor_multi.py:626-718) agree as distributions - means within a stated
      confidence interval;
  (d) the generic fallback is loud: reason available, QS_SPEC=require refuses.
"""
import numpy as np
import pytest

from quad_swarm_rl_amd import config as qcfg

pytestmark = pytest.mark.gpu

from tests import test_hip_parity as thp   # noqa: E402
from tests import tolerances as tolr   # noqa: E402

BASELINE_SHAPED = ["c1_single", "c2_n8_dw", "c2_n8_k2_numpy_wall", "c2_n5_kall_short", "c3_n8_obst", "c3_n8_obst_short", "c4_n32_svs"]


@pytest.mark.parametrize("team", ["1", "0"])
@pytest.mark.parametrize("case", BASELINE_SHAPED)
def test_teacher_forced_f32_generic_kernels(case, team, monkeypatch):
    """(a) QS_SPEC=off: qs_step_team<float> / qs_step_kernel<float> (and the _full variants) against the oracle"""
    monkeypatch.setenv("QS_SPEC", "off")
    monkeypatch.setenv("QS_TEAM", team)
    pr = thp.Pair(case, 5, "f32")
    assert not pr.hip.specialized and pr.hip.spec_note == "QS_SPEC=off"
    assert bool(pr.hip.team) == (team == "1")
    pr.close()
    thp.teacher_forced_f32(case, 5, 45, 1e-5, expect_team=(team == "1"))


def _event_free(info, n, t, ep_len):
    """no interaction so far in this env: nothing in the oracle's flags / masks / counters, and the episode is not about to end"""
    flags = np.array(info.flags[:n]) & 0xff
    return (not flags.any() and info.unique_col_mask == 0 and info.obst_new_mask == 0 and info.room_new_mask == 0
            and not any(info.col_pair_mask[:n]) and not any(info.counters) and info.tick < ep_len)


def _ranking_margin(s, K):
    """smallest gap between consecutive neighbour metrics among every drone's K + 1 nearest (quadrotor_multi.py:247-274): below the
    float32 resolution of the metric the SELECTION (which drone fills which observation slot) is decided by rounding - a discrete
    event like a contact, and the end of an environment's window"""
    pos, vel = s[:, 0:3], s[:, 3:6]
    n = pos.shape[0]
    if K <= 0 or K >= n - 1:
        return np.inf
    dp, dv = pos[None, :, :] - pos[:, None, :], vel[None, :, :] - vel[:, None, :]
    rd = np.maximum(np.linalg.norm(dp, axis=2), 0.01)
    m = rd + (dp * dv).sum(axis=2) / rd
    m[np.arange(n), np.arange(n)] = np.inf
    top = np.sort(m, axis=1)[:, :K + 1]
    return float(np.diff(top, axis=1).min())


FREE_CASES = [("c1_single", None, None), ("c2_n8_dw", None, None), ("c2_n8_dw", "0", None), ("c2_n8_dw", None, "off"),
              ("c3_n8_obst", None, None), ("c4_n32_svs", None, None), ("c4_n32_svs", "0", None), ("x_no_noise", None, None)]


@pytest.mark.parametrize("case,team,spec", FREE_CASES)
def test_free_running_f32_event_free_windows(case, team, spec, monkeypatch):
    """(b) the float32 stepper free-running from its own state against the float64 oracle, same seeds => same noise"""
    if team is not None:
        monkeypatch.setenv("QS_TEAM", team)
    if spec is not None:
        monkeypatch.setenv("QS_SPEC", spec)
    # north_star's 1e-5 is a per-step tolerance (teacher-forced: tests/test_hip_parity.py, tests/test_hip_vs_reference_f32.py).  Free-running, the
    # rounding of a feedback-free integrator accumulates.  Measured on MI355X against the PER-QUANTITY one-step bounds of tests/tolerances.py
    # (absolute 1e-5 on position / rotation / observation columns; 1e-5 * max(1, |x|) on velocity, angular velocity and reward terms;
    # profiles/r05a_tolerance_report.json): inside the first 60 control steps the worst quantity of the worst case (the angular velocity of the
    # 32-drone case) reaches 2.0 x its bound, position 0.4 x, rotation 0.7 x; between step 60 and 100 a relative-velocity observation column 5.8 x,
    # velocity 3.6 x, angular velocity 2.5 x, rotation 1.2 x, position 0.97 x.  Asserted envelope: 3 x before step 60, 8 x up to step 100.
    E, steps, tol, HORIZON = 6, 100, 1e-5, 60
    DRIFT = (3.0, 8.0)   # the free-running envelope in units of the per-quantity one-step bounds (tests/tolerances.py): measured 2.0 / 5.8
    pr = thp.Pair(case, E, "f32", seed=4321)
    N = pr.N
    rng = np.random.RandomState(21)
    oobs, hobs = pr.reset()
    tolr.check(f"free-running {case} team={team} spec={spec}", "obs_reset", hobs, oobs, tolr.allowed_obs(oobs, tol, *pr.obs_layout), "after reset")
    alive = np.ones(E, dtype=bool)
    window = np.zeros(E, dtype=int)
    worst, by, first = 0.0, {}, {}   # worst relative error per quantity (value, step), first step on which a quantity left the tolerance
    for t in range(steps):
        # hover-ish: thrust-to-weight 1.9 => normalised thrust 0.526 => action 0.053, plus a small per-motor perturbation
        act = (0.055 + rng.uniform(-0.04, 0.04, size=(E, N, 4))).astype(np.float32).astype(np.float64)
        o, h = pr.step(act)
        for e, oe in enumerate(pr.oenvs):
            info = oe.info()
            # proximity (a continuous penalty inside 4 arm lengths) and downwash are interactions too: they show in rew_info / flags
            quiet = _event_free(info, N, t, pr.cfg.ep_len) and not np.any(o[3][e][:, 13] != 0.0)
            # a near-tie in the neighbour ranking is a transient: WHICH drone fills WHICH observation slot is then decided by rounding, but the
            # selection does not feed back into the dynamics - on such a step only the self columns of the row are compared
            tie = _ranking_margin(oe.get_state()[0], pr.cfg.num_neighbors) < 5e-5
            if alive[e] and not quiet:
                alive[e] = False
            if not alive[e]:
                continue
            window[e] = t + 1
            sd = pr.D - 6 * pr.cfg.num_neighbors - (9 if pr.cfg.use_obstacles else 0)
            for nm, a, b in (("obs", o[0][e][:, :sd] if tie else o[0][e], h[0][e][:, :sd] if tie else h[0][e]), ("reward", o[1][e], h[1][e]), ("rew_info", o[3][e], h[3][e])):
                # per quantity (tests/tolerances.py): |err| / allowed, allowed = tol absolute (angular-velocity columns, reward terms: tol * max(1, |x|))
                rel = tol * tolr.excess(b, a, tolr.allowed_obs(a, tol, *pr.obs_layout) if nm == "obs" else tolr.allowed_rel(a, tol))
                worst = max(worst, rel)
                if rel > by.get(nm, (0.0, 0))[0]:
                    by[nm] = (rel, t)
                if tolr.REPORT:
                    tolr.check(f"free-running {case} team={team} spec={spec} {'<' if t < HORIZON else '>='}{HORIZON}", nm, b, a, tolr.allowed_obs(a, tol, *pr.obs_layout) if nm == "obs" else tolr.allowed_rel(a, tol))
                elif rel > tol * (DRIFT[0] if t < HORIZON else DRIFT[1]) and nm not in first:
                    first[nm] = (t, e, rel)
            np.testing.assert_array_equal(o[2][e], h[2][e])
        # discrete outputs of the environments still inside their window
        flags = pr.hip.to_host("flags").reshape(E, N)
        cnt, tick = pr.hip.to_host("counters"), pr.hip.to_host("tick")
        cp = pr.hip.to_host("col_pair_mask").reshape(E, N)
        st_pos, st_vel = thp.soa(pr.hip.to_host("pos"), E, N), thp.soa(pr.hip.to_host("vel"), E, N)
        st_rot, st_om = thp.soa(pr.hip.to_host("rot"), E, N), thp.soa(pr.hip.to_host("omega"), E, N)
        for e, oe in enumerate(pr.oenvs):
            if not alive[e]:
                continue
            info = oe.info()
            assert tick[e] == info.tick
            np.testing.assert_array_equal(flags[e] & 0x7ff, np.array(info.flags[:N]) & 0x7ff, err_msg=f"flags env {e} step {t}")
            np.testing.assert_array_equal(cnt[:, e], np.array(info.counters))
            assert not cp[e].any()
            s, _ = oe.get_state()
            for nm, a, b in (("pos", st_pos[e], s[:, 0:3]), ("vel", st_vel[e], s[:, 3:6]), ("rot", st_rot[e], s[:, 6:15]), ("omega", st_om[e], s[:, 15:18])):
                al = tolr.allowed_vec(b, tol) if nm in ("omega", "vel") else tolr.allowed_abs(b, tol)
                rel = tol * tolr.excess(a, b, al)
                worst = max(worst, rel)
                if rel > by.get(nm, (0.0, 0))[0]:
                    by[nm] = (rel, t)
                if tolr.REPORT:
                    tolr.check(f"free-running {case} team={team} spec={spec} {'<' if t < HORIZON else '>='}{HORIZON}", nm, a, b, al)
                elif rel > tol * (DRIFT[0] if t < HORIZON else DRIFT[1]) and nm not in first:
                    first[nm] = (t, e, rel)
        if not alive.any():
            break
    print(f"{case} team={team} spec={spec}: event-free windows {window.tolist()} steps, worst error in units of its bound x 1e-5: {worst:.2e}; per quantity (error, step): "
          + ", ".join(f"{k} {v[0]:.1e}@{v[1]}" for k, v in sorted(by.items())))
    assert not first, f"{case}: free-running float32 left its per-quantity bound ({tol:g} absolute; velocity / angular velocity / reward terms relative; x {DRIFT[0]:g} before step {HORIZON}, x {DRIFT[1]:g} beyond): first (step, env, error) per quantity {first}; worst per quantity {by}"
    assert window.max() >= 40 and np.median(window) >= 20, f"windows too short to mean anything: {window.tolist()}"
    pr.hip.check_errors()
    pr.close()


def test_f32_vs_f64_episode_statistics_full_size_c2():
    """(c) one full 1501-step episode of 8 x 1024 through the float32 and the float64 kernels, same seeds and actions"""
    import torch
    from quad_swarm_rl_amd import native
    kw = dict(thp.CASES["c2_n8_dw"])
    E, N = 1024, 8
    g = torch.Generator(device="cuda").manual_seed(17)
    # hover-ish thrust with a wide per-motor spread: drones drift, meet, touch the floor and the walls - every statistic moves
    acts32 = (0.06 + 0.5 * (torch.rand((64, E * N, 4), device="cuda", generator=g) - 0.5)).float().contiguous()
    acts64 = acts32.double().contiguous()
    out = {}
    for prec, acts in (("f32", acts32), ("f64", acts64)):
        st = native.Stepper(qcfg.make_config(num_envs=E, seed=99, precision=prec, write_rew_info=False, **kw), device=0)
        assert st.specialized
        st.reset()
        ep_len = st.cfg.ep_len
        stride = acts[0].numel() * acts.element_size()
        for t in range(ep_len + 1):
            st.step(acts.data_ptr() + (t % 64) * stride)
        st.sync()
        st.check_errors()
        assert st.to_host("done").all()
        eps, cnt = st.to_host("ep_stats").reshape(6, E, N).astype(np.float64), st.to_host("ep_counters").astype(np.float64)
        ok = eps[4] * eps[5]
        out[prec] = {"num_collisions": cnt[0], "num_collisions_after_settle": cnt[1], "collisions_with_room": cnt[3], "collisions_with_floor": cnt[4],
                     "distance_to_goal_1s": eps[0].mean(axis=1), "distance_to_goal_3s": eps[1].mean(axis=1), "distance_to_goal_5s": eps[2].mean(axis=1),
                     "agent_success_rate": (ok * eps[3]).mean(axis=1), "agent_col_rate": 1.0 - ok.mean(axis=1)}
        st.close()
    lines = []
    for key in out["f32"]:
        a, b = out["f32"][key], out["f64"][key]
        # per-env values; the two runs share every random draw, so this is a paired comparison: mean difference against its standard error
        # (plus an absolute floor for statistics that are almost constant), 4 sigma
        d = a - b
        se = d.std(ddof=1) / np.sqrt(E)
        bound = 4.0 * se + 1e-3 * (1.0 + abs(b.mean()))
        lines.append(f"  {key:32s} f32 {a.mean():10.5f}  f64 {b.mean():10.5f}  diff {d.mean():+.2e}  bound {bound:.2e}")
        assert abs(d.mean()) <= bound, f"{key}: f32 mean {a.mean()} vs f64 mean {b.mean()} (paired diff {d.mean()}, bound {bound})\n" + "\n".join(lines)
    print("episode statistics over 1024 envs, float32 vs float64 kernels:\n" + "\n".join(lines))
    assert out["f64"]["num_collisions"].sum() > 0 and out["f64"]["collisions_with_room"].sum() > 0   # the episode did exercise the event paths


def test_generic_fallback_is_loud(monkeypatch, capfd):
    """(d)"""
    from quad_swarm_rl_amd import native
    kw = dict(thp.CASES["c2_n8_dw"])
    monkeypatch.setenv("QS_SPEC", "cache")
    monkeypatch.setenv("QS_SPEC_CACHE", "/tmp/qs_empty_spec_cache_for_test")
    st = native.Stepper(qcfg.make_config(num_envs=4, seed=1, **kw), device=0)
    assert not st.specialized and "no cached code object" in st.spec_note
    st.close()
    assert "GENERIC step kernels" in capfd.readouterr().err
    monkeypatch.setenv("QS_SPEC", "require")
    monkeypatch.setenv("HIPCC", "/nonexistent/hipcc")
    with pytest.raises(native.QsError, match="QS_SPEC=require"):
        native.Stepper(qcfg.make_config(num_envs=4, seed=1, **kw), device=0)
