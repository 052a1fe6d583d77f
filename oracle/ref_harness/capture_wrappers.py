#!/usr/bin/env python
"""Build-container only (needs /root/reference): runs the reference's QuadsRewardShapingWrapper
(swarm_rl/env_wrappers/reward_shaping.py:19-123) over the scripted env of tests/fake_env.py and writes everything it adds or changes
(true_reward, episode_extra_stats incl. per-scenario and action statistics, the annealed reward coefficients) to
tests/golden/wrapper_reward_shaping.json.  Sample Factory is not installed: its two mix-in interfaces are stubbed with what the
wrapper uses of them (a `training_info` dict and `set_training_info`)."""
import json
import os
import sys
import types

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(HERE, "stubs"))
sys.path.insert(0, "/root/reference")
sys.path.insert(0, REPO)


class TrainingInfoInterface:
    def __init__(self):
        self.training_info = {}

    def set_training_info(self, training_info):
        self.training_info = training_info


class RewardShapingInterface:
    def __init__(self):
        pass


for name in ("sample_factory", "sample_factory.envs"):
    sys.modules[name] = types.ModuleType(name)
m = types.ModuleType("sample_factory.envs.env_utils")
m.TrainingInfoInterface, m.RewardShapingInterface = TrainingInfoInterface, RewardShapingInterface
sys.modules["sample_factory.envs.env_utils"] = m

import gymnasium as gym                                                    # noqa: E402  (the stub)
from swarm_rl.env_wrappers import reward_shaping as ref                    # noqa: E402
from tests.fake_env import FakeQuadEnv, drive                              # noqa: E402

if not hasattr(gym.Wrapper, "unwrapped"):
    gym.Wrapper.unwrapped = property(lambda self: self.env.unwrapped)


class Anneal:   # the namedtuple-like object the reference builds in quad_utils.py:78-83
    def __init__(self, coeff_name, final_value, anneal_env_steps):
        self.coeff_name, self.final_value, self.anneal_env_steps = coeff_name, final_value, anneal_env_steps


out = {}
for case, (annealing, seed) in {"annealed": ([Anneal("quadcol_bin", 5.0, 600000), Anneal("quadcol_bin_smooth_max", 10.0, 300000)], 1),
                                "plain": (None, 2)}.items():
    env = FakeQuadEnv(seed=seed)
    scheme = dict(quad_rewards=dict(pos=1.0, effort=0.05, spin=0.1, vel=0.0, crash=1.0, orient=1.0, yaw=0.0,
                                    quadcol_bin=0.0 if annealing else 5.0, quadcol_bin_smooth_max=0.0 if annealing else 10.0, quadcol_bin_obst=5.0))
    w = ref.QuadsRewardShapingWrapper(env, reward_shaping_scheme=scheme, annealing=annealing, with_pbt=False)
    out[case] = drive(w, env, steps=30, seed=seed)
path = os.path.join(REPO, "tests", "golden", "wrapper_reward_shaping.json")
json.dump(out, open(path, "w"), indent=0, sort_keys=True)
print("wrote", path, os.path.getsize(path), "bytes")
