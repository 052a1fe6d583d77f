"""The reference's fixtures through the FLOAT32 arithmetic of the HIP stepper, one control step at a time (the
production precision was pinned against the reference only through the oracle).

For every fixture of tests/golden (captured from the reference's QuadrotorEnvMulti, oracle/ref_harness/capture.py) two steppers replay
the reference's own random draws (qs_set_noise_tape): the float64 one free-running - it IS the reference to 1e-9, with every flag /
mask / counter exact (tests/test_hip_vs_reference.py) - and the float32 one TEACHER-FORCED: before every control step its dynamic state
(position, velocity, rotation, angular velocity, motor filters, OU state, floor contact, SVD counter, goal, tick) is overwritten with
the float64 twin's - for the 13 fixtures that carry per-step states that is the reference's recorded state to 1e-9 - and its tape
cursor is set to the reference's recorded position.  Expectations of the float32 step:
  * obs / reward: the REFERENCE's recorded outputs, |err| <= 1e-5 * (1 + max|x|)   (north_star: 1e-5 fp32 on dynamics state; the same
    rule as tests/test_hip_parity.py::teacher_forced_f32);
  * post-step state: the twin's, same rule;
  * done, tick, crashed / collision / obstacle / room masks, pair masks, counters, number of draws consumed: exact.
Where float32 cannot follow a float64 branch decision (a drone less than 1e-6 m above the floor-contact threshold - or touching
down within 1e-4 m of it inside the step -, a drone
resting exactly on a wall, the downwash sign test between two drones at the same height, a drone hovering AT the reached-goal distance) the
step is recognised from the pre-step
state and excused: counted, bounded to a small share of the steps, discrete bookkeeping re-synchronised from the twin.
"""
import numpy as np
import pytest

from tests import golden_util as gu
from tests import tolerances as tolr
from tests.test_oracle_vs_reference import CASES, CASES_FIRST_HIT, EDGE_CASES, SCEN_CASES

pytestmark = pytest.mark.gpu
TOL = 1e-5
MAX_EXCUSED = 3          # per fixture, whatever its length (rounds 1-4: 5 % of the steps)
EXCUSED = {}             # fixture -> (excused, steps), printed by the last test of the module
DISCRETE = ("flags", "col_pair_mask", "new_pair_mask", "unique_col_mask", "obst_new_mask", "room_new_mask", "counters", "tick", "obst_hit_idx", "done")
RESYNC = ("flags", "col_pair_mask", "counters")


def fp32_boundary(s, cfg):
    """pre-step state [N, 35] in which float32 may legitimately take another branch than float64 (cases i - iii)"""
    thr = cfg.arm if cfg.floor_mode == 0 else 0.05
    hover = (s[:, 30] == 0) & (s[:, 2] - thr > 0) & (s[:, 2] - thr < 1e-6)
    wall = np.zeros(len(s), bool)
    for a, half in ((0, cfg.room_hi[0]), (1, cfg.room_hi[1])):
        wall |= np.abs(np.abs(s[:, a]) - half) < 1e-5
    same_height = False
    if cfg.use_downwash and len(s) > 1:
        d = s[:, None, 0:3] - s[None, :, 0:3]
        close = (np.hypot(d[..., 0], d[..., 1]) < 0.12) & (np.abs(d[..., 2]) < 1e-5) & ~np.eye(len(s), dtype=bool)
        same_height = bool(close.any())
    # (iv) reached_goal = "mean of the last five goal distances * control_dt / dt < approach_goal_metric" (quadrotor_multi.py:542-546): a drone
    # hovering within 2 % of that distance crosses the threshold on a step decided by rounding
    near_goal = np.abs(np.linalg.norm(s[:, 0:3] - s[:, 32:35], axis=1) * (cfg.dt * cfg.sim_steps) / cfg.dt - cfg.approach_goal_metric) < 0.02 * cfg.approach_goal_metric
    return bool(hover.any() or wall.any() or same_height or near_goal.any())


@pytest.mark.parametrize("name", CASES + EDGE_CASES + SCEN_CASES + CASES_FIRST_HIT)
def test_reference_fixture_teacher_forced_through_f32(name, monkeypatch):
    from quad_swarm_rl_amd import native
    monkeypatch.setenv("QS_SPEC", "off")   # tape handles launch the library's tape kernels only: no per-configuration code object to compile
    g, cfgd = gu.load(name)
    n = cfgd["num_agents"]
    cfg64 = gu.config_from_golden(cfgd, num_envs=1, precision="f64")
    cfg32 = gu.config_from_golden(cfgd, num_envs=1, precision="f32")
    st64, st32 = native.Stepper(cfg64, device=0), native.Stepper(cfg32, device=0)
    try:
        for st in (st64, st32):
            st.set_noise_tape(g["tape"][None, :])
    except native.QsError as exc:
        st64.close(); st32.close()
        pytest.skip(str(exc))
    D = st64.obs_dim

    ctx = f"fixture {name} f32"
    layout = (D - 6 * cfg64.num_neighbors - (9 if cfg64.use_obstacles else 0), cfg64.num_neighbors)
    st64.reset(); st32.reset()
    np.testing.assert_array_equal(st32.tape_pos(), g["tape_pos"][0], err_msg="float32 reset consumed a different number of draws than the reference")
    tolr.check(ctx, "obs_reset", st32.to_host("obs").reshape(n, D), g["obs0"], tolr.allowed_obs(g["obs0"], TOL, *layout), "after reset")

    force = {int(t): k for k, t in enumerate(g["force_steps"])}
    steps = g["actions"].shape[0]
    excused, worst = 0, 0.0
    for t in range(steps):
        s, tick = st64.get_state(0)
        if t in force:
            k = force[t]
            s[:, 0:3] = g["force_pos"][k]; s[:, 3:6] = g["force_vel"][k]
            s[:, 6:15] = g["force_rot"][k].reshape(n, 9); s[:, 15:18] = g["force_omega"][k]
            st64.set_state(0, s, tick)
        st32.set_state(0, s, tick)                       # teacher forcing (rounded to float32 by the library)
        st32.set_tape_pos(g["tape_pos"][t])
        boundary = fp32_boundary(s, cfg64)
        a = g["actions"][t].reshape(-1, 4)
        st64.from_host("actions", a); st32.from_host("actions", a)
        st64.step(); st32.step()
        st64.sync(); st32.sync()
        st32.check_errors()
        ok = True
        why = ""
        if st32.tape_pos()[0] != g["tape_pos"][t + 1]:
            ok, why = False, f"draws consumed {st32.tape_pos()[0]} vs {g['tape_pos'][t + 1]}"
        for nm in DISCRETE:
            if ok and not np.array_equal(st32.to_host(nm), st64.to_host(nm)):
                if nm == "flags":   # bits 0-11: the reference's flags; 12-13 describe private rows (F_RING_LIVE: a rounding-level threshold), 16-23 the SVD counter (forced)
                    f32f, f64f = st32.to_host(nm) & 0xfff, st64.to_host(nm) & 0xfff
                    if np.array_equal(f32f, f64f):
                        continue
                    # case (i) seen from the other side: a drone that TOUCHES DOWN inside this control step - its height after the step within
                    # 1e-4 m of the contact threshold in either precision - lands one sub-step earlier or later depending on rounding
                    thr = cfg64.arm if cfg64.floor_mode == 0 else 0.05
                    z32, z64 = st32.get_state(0)[0][:, 2], st64.get_state(0)[0][:, 2]
                    diff = (f32f ^ f64f) != 0
                    landing = (np.minimum(np.abs(z32 - thr), np.abs(z64 - thr)) < 1e-4) & ((f32f ^ f64f) & ~np.uint32(0x3) == 0)   # only on_floor / crashed_floor differ
                    if diff.any() and landing[diff].all():
                        boundary = True
                    ok, why = False, f"flags differ from the float64 twin: xor {[hex(int(x)) for x in (f32f ^ f64f)]}, heights after the step {np.round(z32, 6).tolist()} / {np.round(z64, 6).tolist()}"
                    continue
                ok, why = False, f"{nm} differs from the float64 twin"
        if not ok:
            assert boundary, f"{name} step {t}: {why} - and the pre-step state is none of the documented float32 boundary cases"
            excused += 1
            for nm in RESYNC:
                v = st64.to_host(nm)
                if nm == "flags":   # (bits 12-13 describe the float32 stepper's own private rows - distance ring, new-pair word: kept)
                    v = (v & ~np.uint32(0x3000)) | (st32.to_host(nm) & np.uint32(0x3000))
                st32.from_host(nm, v)
            continue
        # per quantity (tests/tolerances.py): observation columns absolute 1e-5 (angular-velocity columns 1e-5 * max(1, |w|)), reward
        # 1e-5 * max(1, |r|) - against the REFERENCE's recorded outputs; post-step state against the twin, the same rule per state column
        obs, rew = st32.to_host("obs").reshape(n, D), st32.to_host("reward")
        floats = (("obs", obs, g["obs"][t], tolr.allowed_obs(g["obs"][t], TOL, *layout)), ("reward", rew, g["rew"][t], tolr.allowed_rel(g["rew"][t], TOL)))
        if boundary and any(tolr.excess(got, ref, al) > 1.0 for _, got, ref, al in floats):
            excused += 1
        else:
            for nm, got, ref, al in floats:
                worst = max(worst, tolr.check(ctx, nm, got, ref, al, f"step {t}: differs from the REFERENCE's recorded output"))
        s32, s64 = st32.get_state(0)[0], st64.get_state(0)[0]
        if not boundary:
            worst = max(worst, tolr.check(ctx, "state", s32[:, :30], s64[:, :30], tolr.allowed_state(s64[:, :30], TOL), f"step {t}: post-step state"))
    EXCUSED[name] = (excused, steps)
    assert excused <= MAX_EXCUSED or tolr.REPORT, f"{name}: {excused} of {steps} steps excused as float32 boundary cases (cap {MAX_EXCUSED})"
    print(f"{name}: worst float32 |err| / allowed = {worst:.2f}, {excused} of {steps} steps excused as documented boundary cases")
    st64.close(); st32.close()


def test_excused_steps_summary():
    """the excused-step counts of every fixture in one place (run with -s to see them): a fixed small cap per fixture, stated in the output"""
    if not EXCUSED:
        pytest.skip("run together with the fixture tests")
    total = sum(e for e, _ in EXCUSED.values())
    print("\nexcused float32 boundary steps per fixture: " + ", ".join(f"{k} {e}/{n}" for k, (e, n) in sorted(EXCUSED.items()) if e) + f"; {total} in total over {len(EXCUSED)} fixtures")
    assert all(e <= MAX_EXCUSED for e, _ in EXCUSED.values()) or tolr.REPORT
    if tolr.REPORT:
        tolr._worst["excused_steps|max_per_fixture"] = float(max(e for e, _ in EXCUSED.values()))
