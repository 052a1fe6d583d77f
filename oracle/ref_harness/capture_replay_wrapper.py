#!/usr/bin/env python
"""Build-container only (needs /root/reference): runs the reference's ExperienceReplayWrapper
(gym_art/quadrotor_multi/quad_experience_replay.py:66-209) over the scripted env of tests/fake_env.py (FakeReplayEnv) and writes the
trajectory it produces - which episodes were replays, from which checkpoint, the replay statistics - together with the random
draws it made (one U(0,1) per new episode, one index per sampled event) to tests/golden/wrapper_experience_replay.json."""
import json
import os
import random
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(HERE, "stubs"))
sys.path.insert(0, "/root/reference")
sys.path.insert(0, REPO)

import gymnasium as gym                                                             # noqa: E402  (the stub)
from gym_art.quadrotor_multi import quad_experience_replay as ref                    # noqa: E402
from tests.fake_env import FakeReplayEnv, drive_replay                               # noqa: E402

if not hasattr(gym.Wrapper, "__getattr__"):
    gym.Wrapper.__getattr__ = lambda self, name: getattr(self.__dict__["env"], name)
ref.print = lambda *a, **k: None

draws = {"uniform": [], "randint": []}
_uniform, _randint = np.random.uniform, random.randint


def rec_uniform(*a, **k):
    v = float(_uniform(*a, **k))
    draws["uniform"].append(v)
    return v


def rec_randint(a, b):
    v = _randint(a, b)
    draws["randint"].append(int(v))
    return v


ref.np.random.uniform = rec_uniform
ref.random.randint = rec_randint
np.random.seed(7)
random.seed(7)
env = FakeReplayEnv(seed=3)
w = ref.ExperienceReplayWrapper(env, 0.75, 0.2, 0.6)
steps = 9000
rec = drive_replay(w, steps)
ref.np.random.uniform, ref.random.randint = _uniform, _randint
out = {"steps": steps, "draws": draws, "trajectory": rec}
path = os.path.join(REPO, "tests", "golden", "wrapper_experience_replay.json")
json.dump(out, open(path, "w"), sort_keys=True)
print("wrote", path, os.path.getsize(path), "bytes; episodes", len(rec["ends"]), "uniform draws", len(draws["uniform"]), "sampled events", len(draws["randint"]))
