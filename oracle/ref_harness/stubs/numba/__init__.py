"""Stub of numba for THIS container only (numba is not installed, no network).

Identity decorators: the reference's @njit bodies are plain NumPy and run
un-jitted, giving the same float64 arithmetic.
Test infrastructure only; never shipped to the GPU box as a product path.
"""
import numpy as np


def _identity_decorator(*dargs, **dkwargs):
    if len(dargs) == 1 and callable(dargs[0]) and not dkwargs:
        return dargs[0]

    def wrap(fn):
        return fn
    return wrap


njit = jit = _identity_decorator


def vectorize(*dargs, **dkwargs):
    if len(dargs) == 1 and callable(dargs[0]) and not dkwargs:
        return np.vectorize(dargs[0])

    def wrap(fn):
        return np.vectorize(fn)
    return wrap


class _T:
    def __init__(self, name="any"):
        self.name = name

    def __getitem__(self, item):
        return _T(self.name + "[]")

    def __call__(self, *a, **k):
        return self


int32 = double = boolean = float64 = int64 = _T()
float32 = _T("float32")   # (its own object: the jitclass stub can tell which fields numba would store as float32)


class types:
    Integer = Float = NoneType = Array = type("X", (), {})


random = np.random
