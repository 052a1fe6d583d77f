"""tools/ppo_c5.py (the in-tree PPO harness that runs BASELINE config 5 where Sample Factory is absent) on the CPU: flag handling, and
the learner itself over a scripted stand-in with BatchedQuadSwarm's call protocol - the loop must run and must LEARN (a point mass
whose reward is minus its distance from the origin).  The real thing - the HIP stepper as the environment - is
tests/test_c5_training_gpu.py."""
import os
import sys

import numpy as np
import pytest

torch = pytest.importorskip("torch")
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
import ppo_c5  # noqa: E402


class PointMassVec:
    """E*N agents, obs = [pos(3), zeros...] of the width the encoder expects, action[:3] is a velocity command, reward = -|pos|;
    episodes of `ep_len` steps with auto-reset: the protocol of sf_env.BatchedQuadSwarm (tensors in, 5-tuple out)."""

    def __init__(self, agents=64, obs_dim=54, ep_len=16, seed=0):
        self.A, self.D, self.ep_len = agents, obs_dim, ep_len
        self.g = torch.Generator().manual_seed(seed)
        self.t = 0
        self.training_info = {}

    def _spawn(self):
        return torch.rand((self.A, 3), generator=self.g) * 2.0 - 1.0

    def _obs(self):
        o = torch.zeros((self.A, self.D))
        o[:, :3] = self.pos
        return o

    def set_training_info(self, info):
        self.training_info = info

    def reset(self):
        self.pos, self.t = self._spawn(), 0
        return {"obs": self._obs()}, {}

    def step(self, actions):
        self.pos = self.pos + 0.2 * actions[:, :3].clamp(-1, 1)
        self.t += 1
        rew = -self.pos.norm(dim=1)
        done = torch.full((self.A,), self.t >= self.ep_len)
        infos = []
        if self.t >= self.ep_len:
            self.pos, self.t = self._spawn(), 0
            infos = [{"episode_extra_stats": {}} for _ in range(self.A)]
        return {"obs": self._obs()}, rew, done, torch.zeros_like(done), infos

    def close(self):
        pass


def small_cfg(*extra):
    return ppo_c5.parse(["--rnn_size=16", "--quads_neighbor_hidden_size=16", "--rollout=16", "--batch_size=256", "--learning_rate=0.003",
                         "--quads_neighbor_encoder_type=mean_embed", *extra])


def test_flags_follow_train_local_and_ignore_sample_factory_only_flags():
    cfg = ppo_c5.parse(["--algo=APPO", "--num_workers=4", "--with_vtrace=False", "--iterations=3"])
    assert cfg.ignored_flags == ["--algo=APPO", "--num_workers=4", "--with_vtrace=False"]
    # train_local.sh:1-18
    assert (cfg.learning_rate, cfg.ppo_clip_value, cfg.gae_lambda, cfg.max_grad_norm, cfg.rollout, cfg.batch_size, cfg.reward_clip) == \
        (1e-4, 5.0, 1.0, 5.0, 128, 1024, 10.0)
    assert (cfg.quads_mode, cfg.replay_buffer_sample_prob, cfg.anneal_collision_steps, cfg.quads_neighbor_encoder_type, cfg.quads_neighbor_visible_num) == \
        ("mix", 0.75, 300000000, "attention", 6)
    assert cfg.quads_use_downwash is True and cfg.quads_use_obstacles is False and cfg.quads_collision_reward == 5.0
    assert ppo_c5.parse(["--quads_mode=static_same_goal"]).quads_mode == "static_same_goal"   # the caller's flags win


def test_actor_critic_matches_the_recipe():
    cfg = small_cfg()
    ac = ppo_c5.make_actor_critic(cfg, 54, torch.device("cpu"))
    assert ac.actor_encoder is not ac.critic_encoder                                        # --actor_critic_share_weights=False
    assert all(float(m.bias.detach().abs().max()) == 0.0 for m in ac.modules() if isinstance(m, torch.nn.Linear))
    assert torch.allclose(ac.log_std.exp(), torch.ones(4))                                  # --initial_stddev=1.0, state-independent
    obs = torch.randn(10, 54)
    assert ac.act_mean(obs).shape == (10, 4) and ac.values(obs).shape == (10,)
    lp = ppo_c5.gaussian_logp(torch.zeros(5, 4), torch.zeros(4), torch.ones(5, 4))
    assert torch.allclose(lp, torch.distributions.Normal(0.0, 1.0).log_prob(torch.ones(5, 4)).sum(-1))


def test_gae_cuts_the_bootstrap_at_an_episode_end():
    cfg = small_cfg("--rollout=4", "--gamma=0.5", "--gae_lambda=1.0")
    lr = ppo_c5.Learner(cfg, PointMassVec(agents=2))
    lr.rew[:] = 1.0
    lr.val[:] = 0.0
    lr.val[4] = 8.0
    lr.done[:] = 0.0
    lr.done[1, 0] = 1.0                     # agent 0's episode ends with step 1
    adv, ret = lr.advantages()
    assert np.allclose(adv[:, 1].tolist(), [1 + 0.5 * (1 + 0.5 * (1 + 0.5 * (1 + 0.5 * 8.0))), 1 + 0.5 * (1 + 0.5 * (1 + 4.0)), 1 + 0.5 * 5.0, 5.0])
    assert np.allclose(adv[:, 0].tolist(), [1.5, 1.0, 1 + 0.5 * 5.0, 5.0])
    assert torch.equal(ret, adv + lr.val[:4])


def test_the_learner_learns_on_a_scripted_env():
    cfg = small_cfg("--iterations=30", "--seed=1")
    recs, summary = ppo_c5.train(cfg, env=PointMassVec(agents=128, ep_len=16))
    assert len(recs) == 30 and summary["agent_steps"] == 30 * 16 * 128
    first = np.mean([r["reward_mean"] for r in recs[:3]])
    last = np.mean([r["reward_mean"] for r in recs[-3:]])
    assert last > first + 0.3, (first, last)
    assert recs[-1]["episodes"] == 30 * 128 and recs[-1]["updates"] == 16 * 128 // 256
