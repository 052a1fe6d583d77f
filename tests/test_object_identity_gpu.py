"""Two code objects of the same configuration must produce the same BITS.

(1) Idle lanes.  A wave holds floor(64 / N) whole environments; with N = 17 that leaves 13 of 64 lanes without a drone, and the last
    workgroup of a batch may have whole environments missing.  The debug build `-DQS_POISON_IDLE=1|2` (qs_kernels.h) starts those lanes
    from NaN / 3e30 / all-ones instead of a copy of drone 0 and fills the dynamic LDS with the same pattern before its first use.  If any
    active lane's result depended on an idle lane's registers or on an LDS word nobody wrote, the two builds would differ.
(2) Scheduling flags.  A compiler flag that is only supposed to reorder instructions (register-pressure trackers, post-RA scheduler,
    max-ILP strategy) is admitted only on exact equality with the default build: round 5 had one float32 parity case fail with
    `-amdgpu-use-amdgpu-trackers` on the single-wave objects and could not say why (DESIGN.md 5.3 has the answer).

Both run from identical states with identical Philox streams, crafted events included (the same ones as tests/test_hip_parity.py).  (2) compares
every output and state array BIT FOR BIT after every control step of a free-running rollout: scheduler settings do not touch the IR.  (1) cannot
be bit-exact - the poison build's extra selects change which multiply-adds the compiler fuses, an ulp here and there even in float64 (measured:
profiles/r06b_flag_diff_all_variants.txt, profiles/r06i_poison_f64_first_differences.txt) - so the poisoned handle restarts every step from the plain
handle's state, floats may differ by a few ulps of one step, and everything discrete must be identical.
"""
import os

import numpy as np
import pytest

from quad_swarm_rl_amd import config as qcfg
from tests import test_hip_parity as thp

pytestmark = pytest.mark.gpu

ARRAYS = ["obs", "reward", "done", "rew_info", "pos", "vel", "rot", "omega", "goal", "thrust_rot_damp", "thrust_cmds_damp", "ou_state", "flags",
          "col_pair_mask", "new_pair_mask", "obst_hit_idx", "counters", "tick", "unique_col_mask", "obst_new_mask", "room_new_mask",
          "ep_stats", "ep_counters"]
POISON_CASES = ["e_n17_kall_obst", "e_n33_k8", "c2_n5_kall_short", "c4_n12_svs_short"]   # 13 / 31 / 4 / 4 idle lanes per wave; 7 envs: a partial last block
# (float64 rows of 17 drones that all see each other do not fit the 64 KiB of LDS a module-loaded kernel gets: 6 neighbours there)
POISON_OVERRIDES = {"e_n17_kall_obst": dict(neighbor_visible_num=6)}


def _bits(a):
    a = np.ascontiguousarray(a)
    return a.view({1: np.uint8, 4: np.uint32, 8: np.uint64}[a.dtype.itemsize])


def make_pair(case, E, var, value, precision="f32", seed=77):
    """(plain handle, handle built with `var=value` in the environment) of one parity case"""
    from quad_swarm_rl_amd import native
    kw = dict(thp.CASES[case], **(POISON_OVERRIDES.get(case, {}) if "POISON" in value else {}))
    cfg = qcfg.make_config(num_envs=E, seed=seed, env_id_offset=2, precision=precision, **kw)
    old = os.environ.pop(var, None)
    try:
        a = native.Stepper(cfg, device=0)
        os.environ[var] = value
        b = native.Stepper(cfg, device=0)
    finally:
        os.environ.pop(var, None)
        if old is not None:
            os.environ[var] = old
    assert a.specialized and b.specialized, (a.spec_note, b.spec_note)
    return cfg, a, b


SYNC = ["pos", "vel", "rot", "omega", "thrust_rot_damp", "thrust_cmds_damp", "ou_state", "goal", "flags", "col_pair_mask"]
FLOAT_ULPS = {4: 2e-6, 8: 1e-12}   # "the same arithmetic up to how the compiler contracted it": a few ulps of ONE control step


def identical_rollout(case, E, steps, var, value, expect_team=None, precision="f32", exact=True):
    """exact: every array bit for bit, free-running.  Not exact (builds whose SOURCE differs - the poison builds: fused multiply-adds come out
    differently around the extra selects, an ulp here and there even in float64): B restarts every step from A's state, floats may differ by a
    few ulps of one step, everything discrete must still be identical - an idle lane's NaN / 3e30 / all-ones leaking into an active lane is
    not an ulp."""
    cfg, a, b = make_pair(case, E, var, value, precision=precision)
    if expect_team is not None:
        assert bool(a.team) == expect_team and bool(b.team) == expect_team
    N = cfg.num_agents
    a.reset(); b.reset()
    rng = np.random.RandomState(3)
    M = cfg.num_obstacles
    for t in range(-1, steps):
        if t >= 0:
            for e in range(E):   # the crafted collision / wall / ceiling / floor / obstacle states of the parity suite, applied to both
                s, tick = a.get_state(e)
                oxy = None
                if cfg.use_obstacles:
                    op = a.to_host("obst_pos")
                    oxy = np.stack([op[0, e * M:(e + 1) * M], op[1, e * M:(e + 1) * M]], axis=1).astype(np.float64)
                if thp.force_events(t, e, s, N, oxy, cfg.obst_size / 2):
                    a.set_state(e, s, tick); b.set_state(e, s, tick)
            if not exact:
                for nm in SYNC:
                    b.from_host(nm, a.to_host(nm))
            gentle = (t // 10) % 2 == 1
            act = rng.uniform(-1, 1, size=(E * N, 4)) if not gentle else 0.06 + rng.uniform(-0.05, 0.05, size=(E * N, 4))
            for st in (a, b):
                st.from_host("actions", act)
                st.step()
                st.sync()
        for nm in ARRAYS:
            xa, xb = a.to_host(nm), b.to_host(nm)
            if not exact and xa.dtype.kind == "f":
                assert np.isfinite(xb).all() or not np.isfinite(xa).all(), f"{case} {var}={value}: non-finite values in `{nm}` after step {t}"
                tol = FLOAT_ULPS[xa.dtype.itemsize]
                ne = np.argwhere(np.abs(xa.astype(np.float64) - xb.astype(np.float64)) > tol * np.maximum(1.0, np.abs(xa.astype(np.float64))))
            else:
                ne = np.argwhere(_bits(xa) != _bits(xb))
            if len(ne):
                i = tuple(int(v) for v in ne[0])
                raise AssertionError(f"{case} {var}={value}: `{nm}` differs after step {t} at {i} ({len(ne)} words): {xa[i]!r} vs {xb[i]!r}; "
                                     f"drone index {i[-1] % N if nm != 'obs' else i[0] % N}, env {(i[-1] if nm != 'obs' else i[0]) // N if xa.shape[-1] != E else i[-1]}")
    a.check_errors(); b.check_errors()
    a.close(); b.close()


@pytest.mark.parametrize("mode", [1, 2])
@pytest.mark.parametrize("case", POISON_CASES)
def test_single_wave_kernels_ignore_idle_lanes(case, mode, monkeypatch):
    monkeypatch.setenv("QS_TEAM", "0")
    identical_rollout(case, 7, 45, "QS_SPEC_EXTRA_FLAGS", f"-DQS_POISON_IDLE={mode}", expect_team=False, precision="f64", exact=False)


@pytest.mark.parametrize("mode", [1, 2])
@pytest.mark.parametrize("case", POISON_CASES)
def test_team_kernels_ignore_idle_lanes(case, mode, monkeypatch):
    monkeypatch.setenv("QS_TEAM", "1")
    # (float64 team layouts of more than 8 drones exceed the 64 KiB of LDS a module-loaded kernel gets: those cases run the float32 objects)
    precision = "f64" if thp.CASES[case]["num_agents"] <= 8 else "f32"
    identical_rollout(case, 7, 45, "QS_SPEC_EXTRA_FLAGS", f"-DQS_POISON_IDLE={mode}", expect_team=True, precision=precision, exact=False)


SCHED_SINGLE = ["e_n17_kall_obst", "c3_n8_obst", "c2_n8_dw", "c4_n32_svs", "x_n40_obst"]


@pytest.mark.parametrize("case", SCHED_SINGLE)
def test_single_wave_objects_are_schedule_independent(case, monkeypatch):
    """the register-pressure trackers on the single-wave objects (the flag of round 5's unexplained failure): same bits"""
    monkeypatch.setenv("QS_TEAM", "0")
    identical_rollout(case, 7, 45, "QS_SPEC_SINGLE_FLAGS", "-mllvm -amdgpu-use-amdgpu-trackers", expect_team=False)


@pytest.mark.parametrize("case", ["c2_n8_dw", "c3_n8_obst", "c4_n32_svs"])
def test_team_objects_are_schedule_independent(case, monkeypatch):
    """the team objects' scheduler settings against the compiler's defaults (no max-ILP strategy, post-RA scheduler on): same bits"""
    monkeypatch.setenv("QS_TEAM", "1")
    identical_rollout(case, 7, 45, "QS_SPEC_TEAM_FLAGS", "", expect_team=True)


RESPONSE_CASES = ["c2_n8_dw", "c4_n32_svs", "e_n17_kall_obst", "c2_n5_kall_short"]


@pytest.mark.parametrize("precision", ["f64", "f32"])
@pytest.mark.parametrize("case", RESPONSE_CASES)
def test_parallel_pair_responses_equal_the_serial_form(case, precision, monkeypatch):
    """Drone-drone collision responses (collisions/quadrotors.py:24-59) are order-dependent - a later pair reads the velocities an earlier one
    left.  The team kernels draw the random numbers of all new pairs of a wave in parallel (one lane per Philox block) and apply the responses
    in list order (qs_step_sem.h pair_responses, parallel form); `-DQS_SERIAL_PAIR_RESPONSES` keeps the serial form - one lane per environment
    walking the list.  Same draws, same arithmetic, same order: floats to a few ulps of one step (two builds of different source fuse different
    multiply-adds), everything discrete identical.  The crafted events put 2-3 disjoint pairs into every other environment at t = 6 (more than
    one chunk of 5 pairs per wave at N <= 8) and three mutually colliding drones into the others at t = 14-16 (three pairs that share drones)."""
    monkeypatch.setenv("QS_TEAM", "1")
    if precision == "f64" and thp.CASES[case]["num_agents"] > 8:
        pytest.skip("float64 team layouts of more than 8 drones exceed the 64 KiB of LDS a module-loaded kernel gets")
    identical_rollout(case, 7, 45, "QS_SPEC_EXTRA_FLAGS", "-DQS_SERIAL_PAIR_RESPONSES", expect_team=True, precision=precision, exact=False)


SELECT_CASES = ["c4_n32_svs", "e_n33_k8", "x_n40_obst", "e_n64_k6", "c4_n12_svs_short", "x_svs_odd"]   # 32 / 33 / 40 / 64 / 12 / 9 drones: 5, 6, 6, 6, 4, 4 index bits


@pytest.mark.parametrize("flag", ["-DQS_EXACT_NBR_SELECT", "-DQS_NBR_TRUNC_BITS=16"])
@pytest.mark.parametrize("case", SELECT_CASES)
def test_key_selection_of_neighbours_equals_the_exact_selection(case, flag, monkeypatch):
    """The float32 single-wave kernels keep the K + 1 nearest as one integer key each - the metric's ordered bit pattern with the drone index
    in its low bits, one v_med3_i32 per slot and candidate (qs_kernels.h nbr_select_keys) - and take the exact (metric, index) insertion only
    where two neighbouring entries share a truncated metric.  `-DQS_EXACT_NBR_SELECT` sends every drone down the exact path;
    `-DQS_NBR_TRUNC_BITS=16` truncates 16 bits instead of 3-6, so that the key order is wrong often and only the tie report keeps the result
    right (the exact path runs for some lanes of most waves).  All three must choose the same neighbours in the same order: observation rows to
    a few ulps of one step (builds of different source fuse different multiply-adds), everything discrete identical."""
    monkeypatch.setenv("QS_TEAM", "0")
    identical_rollout(case, 7, 45, "QS_SPEC_EXTRA_FLAGS", flag, expect_team=False, precision="f32", exact=False)


REDO_CASES = ["c4_n32_svs", "c4_n12_svs_short", "x_svs_odd", "s_static_diff", "s_dynamic_formations"]   # 32 / 12 / 9 / 10 / 9 drones, K = 6


@pytest.mark.parametrize("case", REDO_CASES)
def test_redoing_only_the_changed_rows_of_the_metric_matrix_equals_redoing_the_environment(case, monkeypatch):
    """Team kernels, more than 8 drones (pair-once): when an interaction changes a velocity behind the pair scan, the default build re-evaluates
    only the metrics of pairs with a CHANGED drone (rows / columns of the matrix in LDS, one more workgroup barrier) and ranks from the matrix
    again; `-DQS_REDO_ROWS=0` re-evaluates every metric of the environment (qs_step_team.inc).  Same neighbours, same order, same rows: the
    crafted events put collisions, wall / ceiling hits and downwash into the environments at t = 6, 14-16, 22."""
    monkeypatch.setenv("QS_TEAM", "1")
    identical_rollout(case, 7, 45, "QS_SPEC_EXTRA_FLAGS", "-DQS_REDO_ROWS=0", expect_team=True, precision="f32", exact=False)


STASH_CASES = ["c2_n8_dw", "c3_n8_obst", "c4_n32_svs", "e_n17_kall_obst", "c2_n5_kall_short", "x_svs_odd"]


@pytest.mark.parametrize("case", STASH_CASES)
def test_sensor_noise_drawn_ahead_equals_sensor_noise_drawn_on_demand(case, monkeypatch):
    """A step whose interactions changed something observes twice (quadrotor_multi.py:598-599): fresh sensor noise, new self observation.  The
    team kernels' default build lets wave 2 draw that noise while it waits for barrier 1 and parks the 12 values in the row's neighbour columns;
    wave 0 - the critical path - only reads them.  `-DQS_SN_STASH=0`: wave 0 draws them itself when they are needed.  Philox is keyed by (env,
    step, site, pass, drone): the same values either way, every array identical."""
    monkeypatch.setenv("QS_TEAM", "1")
    identical_rollout(case, 7, 45, "QS_SPEC_EXTRA_FLAGS", "-DQS_SN_STASH=0", expect_team=True, precision="f32", exact=False)
