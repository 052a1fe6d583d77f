"""BASELINE.json configs[4] ("C5") end to end on the GPU: the reference's training recipe (train_local.sh:1-18 - mix scenarios, replay
0.75, collision annealing, attention encoder, 6 neighbours, lr 1e-4, rollout 128, batch 1024) with `sf_env.BatchedQuadSwarm` (1024 envs x
8 quads on the HIP stepper) as the environment and the in-tree PPO harness (tools/ppo_c5.py) in Sample Factory's place - SF is not in this
image.  The run must LEARN: the mean per-step position reward (`rew_pos`, quadrotor_single.py:41-44) and the mean shaped reward of the last
rollouts are above those of the first ones by a margin far outside the rollout-to-rollout noise.  tests/test_ppo_harness.py covers the
learner itself on the CPU."""
import os
import sys

import numpy as np
import pytest

torch = pytest.importorskip("torch")
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))

pytestmark = pytest.mark.gpu

ITERATIONS = 16           # x 128 steps x 8192 agents = 16.8e6 agent-steps


@pytest.fixture(scope="module")
def run():
    import ppo_c5
    cfg = ppo_c5.parse([f"--iterations={ITERATIONS}", "--seed=0"])
    recs, summary = ppo_c5.train(cfg)
    return cfg, recs, summary


def test_the_recipe_is_the_references(run):
    cfg, recs, summary = run
    assert (cfg.quads_mode, cfg.replay_buffer_sample_prob, cfg.quads_neighbor_encoder_type, cfg.quads_neighbor_visible_num) == ("mix", 0.75, "attention", 6)
    assert (cfg.learning_rate, cfg.rollout, cfg.batch_size, cfg.gae_lambda, cfg.ppo_clip_value, cfg.max_grad_norm) == (1e-4, 128, 1024, 1.0, 5.0, 5.0)
    assert summary["agents"] == 8192 and summary["agent_steps"] == ITERATIONS * 128 * 8192 >= 2_000_000
    assert all(r["updates"] == 128 * 8192 // 1024 for r in recs)


def test_training_improves_the_position_reward(run):
    _, recs, summary = run
    pos = np.array([r["terms"]["rew_pos"] for r in recs])
    rew = np.array([r["reward_mean"] for r in recs])
    assert np.isfinite(pos).all() and np.isfinite(rew).all() and all(np.isfinite(r["value_loss"]) and np.isfinite(r["policy_loss"]) for r in recs)
    first_pos, last_pos = pos[:2].mean(), pos[-3:].mean()
    first_rew, last_rew = rew[:2].mean(), rew[-3:].mean()
    print(f"\nC5 harness: {summary['fps']:.0f} agent-steps/s; rew_pos {first_pos:.4f} -> {last_pos:.4f}, reward {first_rew:.4f} -> {last_rew:.4f}")
    assert last_pos > first_pos + POS_MARGIN, (pos.round(4).tolist(),)
    assert last_rew > first_rew + REW_MARGIN, (rew.round(4).tolist(),)


POS_MARGIN = 0.02         # per-step rew_pos (= -distance to the goal in metres), see profiles/r05*_ppo_c5_curve.txt for the measured curve
REW_MARGIN = 0.05
