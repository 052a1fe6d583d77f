"""BASELINE.json configs[4] ("C5") end to end on the GPU: the reference's training recipe (train_local.sh:1-18 - mix scenarios, replay
0.75, collision annealing, attention encoder, 6 neighbours, lr 1e-4, rollout 128, gamma 0.99, lambda 1, clip 0.1 / value clip 5, grad-norm
5) with `sf_env.BatchedQuadSwarm` (1024 envs x 8 quads on the HIP stepper) as the environment and the in-tree PPO harness (tools/ppo_c5.py) in
Sample Factory's place - SF is not in this image.  One deviation, for wall clock: minibatches of 8192 samples instead of the recipe's 1024
(128 instead of 1024 optimiser steps per million samples: 0.9e6 instead of 0.25e6 agent-steps/s; the recipe's own batch size learns the same
way per optimiser step - profiles/r05a_ppo_c5_curve_batch1024.txt - but needs 4 x the time per sample).

The run must LEARN.  All 1024 environments start their 1500-step episodes together, so rollout means ride on the episode phase (drones fall
during an episode); episodes are compared with episodes: the mean shaped reward and the mean position reward `rew_pos` (= -distance to the
goal, quadrotor_single.py:41-44) over the LAST episode must be above those over the FIRST by margins far outside their noise.  Measured on
MI355X (profiles/r05b_ppo_c5_b8192.txt, 8 episodes = 1.0e8 agent-steps, 113 s): reward -0.0264 -> -0.0069 per step, rew_pos -0.0159 ->
-0.0079 (it first gets slightly worse while the policy learns not to crash and to stay level - crash -0.0037 -> -0.00005, orientation -0.0027
-> +0.0046 - and improves from the sixth episode on).  tests/test_ppo_harness.py covers the learner itself on the CPU."""
import os
import sys

import numpy as np
import pytest

torch = pytest.importorskip("torch")
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))

pytestmark = pytest.mark.gpu

EPISODES = 8
PER_EPISODE = 12          # rollouts of 128 steps per 1500-step episode (the 36 steps that spill over are ignored)
ITERATIONS = EPISODES * PER_EPISODE           # x 128 steps x 8192 agents = 1.0e8 agent-steps
REW_MARGIN, POS_MARGIN = 0.008, 0.003         # measured improvements 0.0195 / 0.0080 per step


@pytest.fixture(scope="module")
def run():
    import ppo_c5
    cfg = ppo_c5.parse([f"--iterations={ITERATIONS}", "--batch_size=8192", "--seed=0"])
    recs, summary = ppo_c5.train(cfg)
    return cfg, recs, summary


def test_the_recipe_is_the_references(run):
    cfg, recs, summary = run
    assert (cfg.quads_mode, cfg.replay_buffer_sample_prob, cfg.quads_neighbor_encoder_type, cfg.quads_neighbor_visible_num) == ("mix", 0.75, "attention", 6)
    assert (cfg.anneal_collision_steps, cfg.quads_collision_reward, cfg.quads_use_downwash, cfg.quads_use_numba) == (300000000, 5.0, True, True)
    assert (cfg.learning_rate, cfg.rollout, cfg.gae_lambda, cfg.gamma, cfg.ppo_clip_ratio, cfg.ppo_clip_value, cfg.max_grad_norm, cfg.reward_clip) == \
        (1e-4, 128, 1.0, 0.99, 0.1, 5.0, 5.0, 10.0)
    assert summary["agents"] == 8192 and summary["agent_steps"] == ITERATIONS * 128 * 8192
    assert all(r["updates"] == 128 * 8192 // cfg.batch_size for r in recs)
    assert recs[-1]["episodes"] >= (EPISODES - 1) * 8192          # auto-resets (and replayed episodes) went through the batched env's infos


def test_training_improves_reward_and_position_reward(run):
    _, recs, summary = run
    pos = np.array([r["terms"]["rew_pos"] for r in recs]).reshape(EPISODES, PER_EPISODE).mean(axis=1)
    rew = np.array([r["reward_mean"] for r in recs]).reshape(EPISODES, PER_EPISODE).mean(axis=1)
    crash = np.array([r["terms"]["rew_crash"] for r in recs]).reshape(EPISODES, PER_EPISODE).mean(axis=1)
    assert np.isfinite(pos).all() and np.isfinite(rew).all() and all(np.isfinite(r["value_loss"]) and np.isfinite(r["policy_loss"]) for r in recs)
    print(f"\nC5 harness: {summary['fps']:.0f} agent-steps/s over {summary['agent_steps']:.2e} agent-steps; per-episode means - reward {rew.round(4).tolist()}, "
          f"rew_pos {pos.round(4).tolist()}, rew_crash {crash.round(5).tolist()}")
    assert rew[-1] > rew[0] + REW_MARGIN, rew.round(5).tolist()
    assert pos[-1] > pos[0] + POS_MARGIN, pos.round(5).tolist()
    assert crash[-1] > 0.5 * crash[0], crash.round(5).tolist()        # (crash is a penalty <= 0: the last episode's is at most half the first's)
    assert recs[-1]["action_std"][0] < 1.0                            # the free log-std moved off its initial value
