"""jitclass stub.  By default the class is returned unchanged (every field a Python / float64 value).  With EMULATE_FLOAT32_FIELDS = True
(set by oracle/ref_harness/capture.py for the `*_f32ou` fixtures) a scalar field that the spec declares `float32` is ROUNDED TO FLOAT32 when it
is assigned - what numba's jitclass does when it stores a Python float into a float32 member (OUNoiseNumba.theta / .sigma / .mu,
gym_art/quadrotor_multi/numba_utils.py:67-74); reading it back gives the widened value, and float32 * float64-array is float64 in numba's typing
as in NumPy's, so the arithmetic downstream is the float64 arithmetic of the real jitclass."""
import numpy as np

EMULATE_FLOAT32_FIELDS = False


def _wrap(cls, spec):
    f32 = {name for name, t in (spec or []) if getattr(t, "name", "") == "float32"}
    if not f32:
        return cls
    base_setattr = cls.__setattr__

    def __setattr__(self, name, value):
        if EMULATE_FLOAT32_FIELDS and name in f32:
            value = float(np.float32(value))
        base_setattr(self, name, value)

    cls.__setattr__ = __setattr__
    return cls


def jitclass(*a, **k):
    if len(a) == 1 and isinstance(a[0], type):
        return a[0]
    spec = a[0] if a else k.get("spec")

    def wrap(cls):
        return _wrap(cls, spec)
    return wrap
