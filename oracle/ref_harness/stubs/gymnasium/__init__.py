"""Minimal gymnasium stub (gymnasium is not installed in this container)."""
from . import spaces, utils, error  # noqa


class Env:
    metadata = {}

    def __init__(self):
        pass

    @property
    def unwrapped(self):
        return self

    def close(self):
        pass


class Wrapper(Env):
    def __init__(self, env):
        self.env = env

    def __getattr__(self, name):
        return getattr(self.env, name)

    @property
    def unwrapped(self):
        return self.env.unwrapped
