"""The replay kernel's decision logic (tests/replay_model.py, the Python twin of qs_replay_kernel) against the REFERENCE wrapper.

tests/golden/wrapper_experience_replay.json holds what the reference's ExperienceReplayWrapper
(gym_art/quadrotor_multi/quad_experience_replay.py:66-209) did over the scripted env of tests/fake_env.py - 9000 steps, 18 episodes,
14 replays - together with the random draws it made (oracle/ref_harness/capture_replay_wrapper.py).  Here the model makes the
decisions, a small harness carries them out on the same scripted env (checkpoint = the env's copyable core), and the trajectory and
the replay statistics at every episode end must equal the reference's.  CPU only; tests/test_replay_gpu.py then requires the
device kernel to equal the model."""
import json
import os

import pytest

from tests.fake_env import FakeReplayEnv
from tests.replay_model import EVENTS, RING, ReplayModel

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "wrapper_experience_replay.json")


def test_model_reproduces_the_reference_wrapper():
    want = json.load(open(GOLDEN))
    u, idx = list(want["draws"]["uniform"]), list(want["draws"]["randint"])
    env = FakeReplayEnv(seed=3)
    model = ReplayModel(0.75, control_freq=100, use_obstacles=False, active=True)   # the scripted env has its replay buffer active from the start
    pool = [None] * (RING + EVENTS)

    def draw_idx(n):
        v = idx.pop(0)
        assert 0 <= v < n
        return v

    rec = {"x": [], "tick": [], "episode": [], "ends": {}}
    obs = env.reset()
    for t in range(want["steps"]):
        obs, rewards, dones, infos = env.step(None)     # (the scripted env resets itself at done, like QuadrotorEnvMulti.step)
        ids = env.last_step_unique_collisions
        mask = sum(1 << int(i) for i in ids)
        # `.any()` on the id array [0, 1] is True: the mask test (mask & ~1) != 0 sees id 1
        for act in model.step(bool(dones[0]), env.envs[0].tick, mask, 0, 0.0, lambda: u.pop(0), draw_idx):
            if act[0] == "save":
                pool[act[1]] = (env.core(), [o.copy() for o in obs])
            elif act[0] == "file":
                pool[act[2]] = pool[act[1]]
                obs = [o.copy() for o in pool[act[1]][1]]           # the reference returns the filed checkpoint's observation
            elif act[0] == "fresh":
                obs = env.reset()                                   # the wrapper's own reset() of a non-replayed episode
            else:
                core, ck_obs = pool[act[1]]
                env.set_core(core)
                env.collisions_per_episode = env.collisions_after_settle = 0
                obs = [o.copy() for o in ck_obs]
        rec["x"].append(float(obs[0][0])); rec["tick"].append(int(obs[0][1])); rec["episode"].append(int(obs[0][2]))
        if dones[0]:
            s = model.stats()
            stats = dict(infos[0]["episode_extra_stats"])
            stats.update({"replay/replay_rate": s["replayed"] / s["episodes"], "replay/new_episode_rate": (s["episodes"] - s["replayed"]) / s["episodes"],
                          "replay/replay_buffer_size": s["buffer_len"], "replay/avg_replayed": (s["replayed_sum"] / s["buffer_len"]) if s["buffer_len"] else 0,
                          "replay/obst_density": 0.2, "replay/obst_size": 0.6})
            rec["ends"][str(t)] = {k: float(v) for k, v in sorted(stats.items())}
    ref = want["trajectory"]
    assert rec["tick"] == ref["tick"] and rec["episode"] == ref["episode"]
    assert rec["x"] == pytest.approx(ref["x"], rel=0, abs=0)
    assert sorted(rec["ends"]) == sorted(ref["ends"]) and len(ref["ends"]) >= 15
    for t in ref["ends"]:
        assert rec["ends"][t] == pytest.approx(ref["ends"][t]), (t, rec["ends"][t], ref["ends"][t])
    assert not u and not idx            # every draw of the reference was asked for, in order
    assert model.errors == 0
