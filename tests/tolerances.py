"""Per-quantity tolerances of the float comparisons in the GPU parity tests.

north_star: "within 1e-5 fp32 on dynamics state".  Rounds 1-4 compared whole mixed-quantity arrays with `tol * (1 + max|x|)`: with an
angular velocity of 40 rad/s in the same array that allowed 4e-4 on a rotation-matrix entry.  Here every quantity has its own bound:

    rotation matrix, position, velocity, goal, motor filters, self / neighbour / SDF observation columns:  |err| <= tol          (absolute)
    angular velocity (state columns 15:18, observation columns 15:18):                                     |err| <= tol * max(1, |ref|)
    reward and its terms (collision penalties reach O(10)):                                                |err| <= tol * max(1, |ref|)

tol = 1e-5 for float32 (one control step from a forced state, or the stated horizon of a free-running window), 1e-8 / 1e-7 for the float64
rollouts.  Quantities that need more than that are listed HERE, with the reason, instead of being hidden in a looser global rule:

    EXTRA = {quantity: factor}   (empty unless a GPU run showed the need; see profiles/r05*_tolerance_report.json)

QS_TOL_REPORT=<path>: nothing is asserted by check(); the worst err / allowed per (context, quantity) is merged into <path> as JSON - how the
bounds above were checked against every parity case before they became assertions."""
import atexit
import json
import os

import numpy as np

REPORT = os.environ.get("QS_TOL_REPORT")
EXTRA = {}
_worst = {}


def _allowed(ref, tol, omega_cols=None, relative=False):
    ref = np.asarray(ref, dtype=np.float64)
    if relative:
        return tol * np.maximum(1.0, np.abs(ref))
    a = np.full(ref.shape, tol)
    if omega_cols is not None and ref.shape[-1] >= omega_cols.stop:
        a[..., omega_cols] = tol * np.maximum(1.0, np.abs(ref[..., omega_cols]))
    return a


def allowed_obs(ref, tol):
    """observation rows [..., D]: columns 15:18 of the self block are the (noisy) angular velocity"""
    return _allowed(ref, tol, omega_cols=slice(15, 18))


def allowed_state(ref, tol):
    """qs_get_state rows [..., >= 30]: pos 0:3, vel 3:6, rot 6:15, omega 15:18, motor filters / OU state 18:30"""
    return _allowed(ref, tol, omega_cols=slice(15, 18))


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
