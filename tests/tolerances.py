"""Per-quantity tolerances of the float comparisons in the GPU parity tests.

north_star: "within 1e-5 fp32 on dynamics state".  Rounds 1-4 compared whole mixed-quantity arrays with `tol * (1 + max|x|)`: with an
angular velocity of 40 rad/s in the same array that allowed 4e-4 on a rotation-matrix entry.  Here every quantity has its own bound:

    rotation matrix, position, goal, motor filters, position / SDF observation columns:                    |err| <= tol          (absolute)
    angular velocity (state columns 15:18, observation columns 15:18):                                     |err| <= tol * max(1, |ref|)
    velocity (state columns 3:6, observation columns 3:6 and the relative-velocity columns of every neighbour): |err| <= tol * max(1, |ref|)
    reward and its terms (collision penalties reach O(10)):                                                |err| <= tol * max(1, |ref|)

tol = 1e-5 for float32 ONE control step from a forced state (north_star's claim), 1e-8 / 1e-7 for the float64 rollouts.  Why velocity is relative:
the first GPU run of these bounds (profiles/r05a_tolerance_report.json, every parity case) found every quantity of every case inside its
absolute bound - worst rot 0.03, pos 0.15, omega 0.53 of the bound - except the velocity behind an obstacle-collision response in the two
densest obstacle cases (2.1e-5 / 1.8e-5): the response sets speeds of several m/s (collisions/obstacles.py:8-40), where float32 carries ~1e-6
per unit.  Free-running float32 (tests/test_fp32_parity_gpu.py) is not the per-step claim: rounding accumulates in a feedback-free integrator,
and that test states its own envelope - 3 x these bounds inside the first 60 control steps, 8 x up to step 100 (measured: 2.0 x / 5.8 x).

    EXTRA = {quantity: factor}   (empty: nothing needs more than the rules above)

QS_TOL_REPORT=<path>: nothing is asserted by check(); the worst err / allowed per (context, quantity) is merged into <path> as JSON - how the
bounds above were checked against every parity case before they became assertions."""
import atexit
import json
import os

import numpy as np

REPORT = os.environ.get("QS_TOL_REPORT")
EXTRA = {}
_worst = {}


def _allowed(ref, tol, rel_cols=(), relative=False):
    ref = np.asarray(ref, dtype=np.float64)
    if relative:
        return tol * np.maximum(1.0, np.abs(ref))
    a = np.full(ref.shape, tol)
    for cols in rel_cols:
        if ref.shape[-1] >= cols.stop:
            a[..., cols] = tol * np.maximum(1.0, np.abs(ref[..., cols]))
    return a


def allowed_obs(ref, tol, self_dim=18, num_nbr=None):
    """observation rows [..., D] = [self block (pos 0:3, vel 3:6, rot 6:15, omega 15:18, ...) | num_nbr x (rel pos 3, rel vel 3) | SDF 9]: the
    velocity, angular-velocity and relative-velocity columns relative, everything else absolute.  num_nbr=None: as many 6-column neighbour
    blocks as fit behind the self block (a trailing 9-column SDF block is then left absolute only if the rest is a multiple of 6: callers that
    have the configuration pass num_nbr)."""
    D = np.asarray(ref).shape[-1]
    if num_nbr is None:
        rest = D - self_dim
        num_nbr = (rest - 9) // 6 if (rest % 6 != 0 and rest >= 9 and (rest - 9) % 6 == 0) else rest // 6
    cols = [slice(3, 6), slice(15, 18)] + [slice(self_dim + 6 * k + 3, self_dim + 6 * k + 6) for k in range(max(num_nbr, 0))]
    return _allowed(ref, tol, rel_cols=cols)


def allowed_state(ref, tol):
    """qs_get_state rows [..., >= 30]: pos 0:3, vel 3:6, rot 6:15, omega 15:18, motor filters / OU state 18:30"""
    return _allowed(ref, tol, rel_cols=(slice(3, 6), slice(15, 18)))


def allowed_rel(ref, tol):
    return _allowed(ref, tol, relative=True)


def allowed_abs(ref, tol):
    return _allowed(ref, tol)


def excess(got, ref, allowed):
    """max over elements of |got - ref| / allowed (<= 1 means inside the bound)"""
    got, ref = np.asarray(got, dtype=np.float64), np.asarray(ref, dtype=np.float64)
    if got.size == 0:
        return 0.0
    return float((np.abs(got - ref) / allowed).max())


def check(context, quantity, got, ref, allowed, msg=""):
    """assert |got - ref| <= allowed elementwise (times EXTRA[quantity] where listed); returns err / allowed.  In report mode: records only."""
    x = excess(got, ref, allowed * EXTRA.get(quantity, 1.0))
    if REPORT:
        key = f"{context}|{quantity}"
        _worst[key] = max(_worst.get(key, 0.0), x)
        return x
    assert x <= 1.0, f"{context}: {quantity} {msg}: |err| / allowed = {x:.3g} (max abs err {np.abs(np.asarray(got, dtype=np.float64) - np.asarray(ref, dtype=np.float64)).max():.3g})"
    return x


def _dump():
    if not REPORT or not _worst:
        return
    old = {}
    try:
        with open(REPORT) as f:
            old = json.load(f)
    except (OSError, ValueError):
        pass
    for k, v in _worst.items():
        old[k] = max(old.get(k, 0.0), v)
    os.makedirs(os.path.dirname(os.path.abspath(REPORT)), exist_ok=True)
    with open(REPORT, "w") as f:
        json.dump(old, f, indent=0, sort_keys=True)


atexit.register(_dump)
