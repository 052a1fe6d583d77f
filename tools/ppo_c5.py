#!/usr/bin/env python
"""BASELINE.json configs[4] ("C5") without Sample Factory: the reference's training recipe (train_local.sh:1-18, entered through
swarm_rl/train.py:16-33) run end to end on PyTorch-ROCm by a minimal synchronous PPO learner.  A HARNESS, not product: Sample
Factory is not in this image (and cannot be installed), so this file plays its part - sampler and learner - around the pieces that
ARE the product: `sf_env.BatchedQuadSwarm` (the HIP stepper behind the reference's wrapper stack: device-side replay, reward
shaping sums, collision-coefficient annealing, reward_shaping.py:52-123) and the encoder restatement pinned by the reference
fixtures (`policy.encoder_from_cfg`, swarm_rl/models/quad_multi_model.py:250-370).

What it takes from train_local.sh / Sample Factory's defaults (flag names kept): --learning_rate 1e-4, --ppo_clip_ratio 0.1 (SF
default), --ppo_clip_value 5.0, --gamma 0.99 (SF default), --gae_lambda 1.0, --max_grad_norm 5.0, --exploration_loss_coeff 0,
--rollout 128, --batch_size 1024, --num_epochs 1 (SF default), --reward_clip 10, --value_loss_coeff 0.5 (SF default),
--nonlinearity tanh, --policy_initialization xavier_uniform, --actor_critic_share_weights False (two encoders), --adaptive_stddev
False (a state-independent log-std parameter, --initial_stddev 1.0), --normalize_input / --normalize_returns False, Adam
(eps 1e-6, betas 0.9 / 0.999: SF defaults), no RNN, no V-trace; every --quads_* flag of train_local.sh through
sf_env.add_quadrotors_env_args.  What differs from APPO: synchronous (collect `rollout` steps from all E*N agents, then one pass
over the samples in minibatches of `batch_size`) - the policy lag inside a pass is bounded by the pass, where APPO's is bounded by
--max_policy_lag.

    python tools/ppo_c5.py --iterations 10                       # one JSON line per iteration, then a summary line
    python tools/ppo_c5.py --iterations 10 --batch_size 8192     # fewer, larger minibatches (faster wall clock, fewer updates)
"""
import argparse
import json
import math
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

TRAIN_LOCAL = ["--quads_use_numba=True", "--anneal_collision_steps=300000000", "--replay_buffer_sample_prob=0.75", "--quads_mode=mix",
               "--quads_episode_duration=15.0", "--quads_obs_repr=xyz_vxyz_R_omega", "--quads_neighbor_hidden_size=256",
               "--quads_neighbor_obs_type=pos_vel", "--quads_collision_hitbox_radius=2.0", "--quads_collision_falloff_radius=4.0",
               "--quads_collision_reward=5.0", "--quads_collision_smooth_max_penalty=10.0", "--quads_neighbor_encoder_type=attention",
               "--quads_neighbor_visible_num=6", "--quads_use_obstacles=False", "--quads_use_downwash=True"]   # train_local.sh:8-17


def make_parser():
    from quad_swarm_rl_amd import sf_env
    p = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    # the learner's flags, Sample Factory's names; defaults = train_local.sh where it sets them, SF's defaults elsewhere
    p.add_argument("--learning_rate", type=float, default=1e-4)
    p.add_argument("--ppo_clip_ratio", type=float, default=0.1)
    p.add_argument("--ppo_clip_value", type=float, default=5.0)
    p.add_argument("--gamma", type=float, default=0.99)
    p.add_argument("--gae_lambda", type=float, default=1.0)
    p.add_argument("--max_grad_norm", type=float, default=5.0)
    p.add_argument("--exploration_loss_coeff", type=float, default=0.0)
    p.add_argument("--value_loss_coeff", type=float, default=0.5)
    p.add_argument("--rollout", type=int, default=128)
    p.add_argument("--batch_size", type=int, default=1024)
    p.add_argument("--num_epochs", type=int, default=1)
    p.add_argument("--reward_clip", type=float, default=10.0)
    p.add_argument("--reward_scale", type=float, default=1.0)
    p.add_argument("--initial_stddev", type=float, default=1.0)
    p.add_argument("--adam_eps", type=float, default=1e-6)
    p.add_argument("--rnn_size", type=int, default=256)
    p.add_argument("--nonlinearity", type=str, default="tanh")
    p.add_argument("--seed", type=int, default=0)
    # the harness's own
    p.add_argument("--iterations", type=int, default=10, help="rollout + update passes; agent-steps = iterations * rollout * envs * quads")
    p.add_argument("--train_for_env_steps", type=int, default=0, help="alternative to --iterations (SF's flag): agent-steps to train for")
    p.add_argument("--quiet", action="store_true")
    p.add_argument("--graph_update", type=lambda v: str(v).lower() in ("1", "true", "yes"), default=True,
                   help="record the minibatch step (forward, losses, backward, gradient clipping, Adam) once into a HIP graph and replay it per minibatch "
                        "(same arithmetic, no per-kernel launch cost); falls back to eager steps where capture is not available")
    p.add_argument("--fused_sampler", type=lambda v: str(v).lower() in ("1", "true", "yes"), default=True,
                   help="collect the rollouts on the product's closed loop - the fused MFMA policy encoder + action head + step kernel recorded into ONE HIP "
                        "graph per rollout (rollout.GraphedRollout over the env's stepper; bf16 encoder) - instead of stepping the torch actor eagerly; the "
                        "behaviour policy's log-probabilities come from the action means the segment records, the values from the float32 critic in "
                        "one batched pass.  Falls back to the eager sampler where the fused kernels do not apply (CPU stand-in envs, float64, widths)")
    p.add_argument("--sampler_precision", choices=("bf16", "fp32"), default="fp32",
                   help="operands of the fused sampler's encoder: fp32 = reference precision (fp16 pairs, three MFMAs per product; the behaviour policy IS the "
                        "learner's float32 actor to ~1e-6, what PPO's ratio assumes), bf16 = the faster kernels (action means ~1e-2 away from the learner's)")
    sf_env.add_quadrotors_env_args(None, p)
    p.set_defaults(quads_num_envs=1024)
    return p


def parse(argv=None):
    """train_local.sh's environment flags first, then the caller's (later flags win); SF-only flags (--algo, --num_workers ...) are ignored"""
    p = make_parser()
    cfg, unknown = p.parse_known_args(TRAIN_LOCAL + list(argv if argv is not None else sys.argv[1:]))
    cfg.ignored_flags = unknown
    return cfg


def make_env(cfg):
    """sf_env.make_quadrotor_env_batched plus the per-step reward terms (the learning-curve columns below are means of them)"""
    from quad_swarm_rl_amd import sf_env
    reward_shaping, annealing = sf_env._shaping_from_cfg(cfg)
    return sf_env.BatchedQuadSwarm(cfg.quads_num_envs, reward_shaping_scheme=reward_shaping, annealing=annealing, write_rew_info=True,
                                   **sf_env._env_kwargs_from_cfg(cfg))


def make_actor_critic(cfg, obs_dim, device):
    """Two encoders (--actor_critic_share_weights=False), a Linear(512, 4) mean head with a free log-std, a Linear(512, 1) value head;
    every Linear xavier_uniform with zero bias (--policy_initialization=xavier_uniform)."""
    import torch
    from torch import nn
    from quad_swarm_rl_amd import policy

    class ActorCritic(nn.Module):
        def __init__(self):
            super().__init__()
            self.actor_encoder = policy.encoder_from_cfg(cfg, seed=None)
            self.critic_encoder = policy.encoder_from_cfg(cfg, seed=None)
            feat = 2 * cfg.rnn_size
            self.action_mean = nn.Linear(feat, 4)
            self.log_std = nn.Parameter(torch.full((4,), math.log(cfg.initial_stddev)))
            self.value = nn.Linear(feat, 1)
            for m in self.modules():
                if isinstance(m, nn.Linear):
                    nn.init.xavier_uniform_(m.weight, gain=1.0)
                    nn.init.zeros_(m.bias)

        def act_mean(self, obs):
            return self.action_mean(self.actor_encoder(obs))

        def values(self, obs):
            return self.value(self.critic_encoder(obs)).squeeze(-1)

    torch.manual_seed(cfg.seed)
    return ActorCritic().to(device)


def gaussian_logp(mean, log_std, act):
    z = (act - mean) * (-log_std).exp()
    return (-0.5 * z * z - log_std - 0.5 * math.log(2.0 * math.pi)).sum(-1)


class Learner:
    """collect() = `rollout` control steps of all agents with the current policy; update() = one PPO pass over them"""

    def __init__(self, cfg, env, device=None):
        import torch
        self.torch, self.cfg, self.env = torch, cfg, env
        obs, _ = env.reset()
        obs = obs["obs"]
        self.device = device if device is not None else obs.device
        self.A, self.D = obs.shape
        self.ac = make_actor_critic(cfg, self.D, self.device)
        self._want_graph = bool(getattr(cfg, "graph_update", False)) and self.device.type == "cuda"
        # (capturable: step counter and learning rate live on the device, so that optimiser steps can be recorded into a HIP graph)
        self.opt = torch.optim.Adam(self.ac.parameters(), lr=torch.tensor(cfg.learning_rate, device=self.device) if self._want_graph else cfg.learning_rate,
                                    betas=(0.9, 0.999), eps=cfg.adam_eps, capturable=self._want_graph)
        self._graph, self._graph_error, self._static = None, None, None
        self._acc = torch.zeros(4, device=self.device)
        T, A, D = cfg.rollout, self.A, self.D
        f32 = dict(dtype=torch.float32, device=self.device)
        self.obs = torch.zeros((T + 1, A, D), **f32)
        self.act = torch.zeros((T, A, 4), **f32)
        self.logp = torch.zeros((T, A), **f32)
        self.val = torch.zeros((T + 1, A), **f32)
        self.rew = torch.zeros((T, A), **f32)
        self.done = torch.zeros((T, A), **f32)
        self.obs[0].copy_(obs)
        self.agent_steps = 0
        self.episodes = 0
        self.terms = None       # [17] means of the per-step reward terms over the last rollout (config.REW_INFO_KEYS order)
        self._rew_info = getattr(getattr(env, "vec", None), "reward_info", None)
        self.segment, self.sampler_note, self.sampler_gap = None, "eager torch actor", None
        if bool(getattr(cfg, "fused_sampler", False)) and self.device.type == "cuda":
            self._make_segment()

    def _make_segment(self):
        """the product's closed loop as the sampler: fused encoder (the actor's weights, bf16) + Gaussian head + step kernel, one HIP graph per rollout"""
        torch, cfg = self.torch, self.cfg
        try:
            from quad_swarm_rl_amd import policy, rollout
            self.fused = policy.FusedQuadEncoder(self.ac.actor_encoder, device=self.device.index or 0, precision=getattr(cfg, "sampler_precision", "fp32"))
            self.head = rollout.GaussianActionHead(in_features=self.fused.out_dim, device=self.device.index or 0, seed=cfg.seed, sample=True)
            self.head.weight, self.head.bias = self.ac.action_mean.weight, self.ac.action_mean.bias   # set_head copies them; refresh() follows them
            self.head.log_std = self.ac.log_std.detach().clone()
            self.segment = rollout.GraphedRollout(self.env.vec, self.fused, self.head, steps=cfg.rollout)
            self.sampler_note = f"fused encoder ({self.fused.precision} operands) + head + step, one HIP graph of {cfg.rollout} control steps (rollout.GraphedRollout)"
        except Exception as exc:   # noqa: BLE001 - the eager sampler is always there
            self.segment, self.sampler_note = None, f"eager torch actor (fused sampler unavailable: {type(exc).__name__}: {exc})"

    def collect_fused(self):
        """one rollout = one replay of the captured segment; then, in batched float32 passes: log-probabilities of the recorded actions under the
        recorded means (the behaviour policy as it really sampled), values of all T + 1 observation sets from the critic"""
        torch, cfg, seg = self.torch, self.cfg, self.segment
        T = cfg.rollout
        with torch.no_grad():
            self.fused.refresh()                                  # the weights the last update left, into the buffers the graph points at
            self.head.log_std.copy_(self.ac.log_std.detach())
            if self.fused.attention:
                seg.recapture()                                   # (the attention score bias travels in the launch arguments)
            self.env.set_training_info({"approx_total_training_steps": self.agent_steps})
            self.env.segment_begin()
            st = self.env.vec.stepper
            run_before = st.tensor("run_sums")[:17].clone()       # the step kernel's running per-episode sums of the 17 reward terms (episode_sums)
            out = seg.run()
            self.obs[:T].copy_(out["obs"]); self.obs[T].copy_(out["last_obs"])
            self.act.copy_(out["actions"]); self.rew.copy_(out["rewards"]); self.done.copy_(out["dones"])
            self.logp.copy_(gaussian_logp(out["means"], self.head.log_std, out["actions"]))
            # how far the behaviour policy (fused kernels) is from the learner's float32 actor on the same observations: |mean difference|, first step
            self.sampler_gap = float((self.ac.act_mean(out["obs"][0]) - out["means"][0]).abs().max())
            flat = self.obs.reshape((T + 1) * self.A, self.D)
            vals = self.val.reshape(-1)
            for s0 in range(0, flat.shape[0], 65536):
                vals[s0:s0 + 65536] = self.ac.values(flat[s0:s0 + 65536])
            infos = self.env.segment_end(out["dones"])
            if infos:
                self.episodes += len(infos.finished_agents())
            self.agent_steps += T * self.A
            # means of the per-step reward terms over the rollout, from the running sums: end - start, plus the finished episode's total
            # for the agents whose episode ended (and restarted the running sum) inside the segment
            ended = out["dones"].any(dim=0).float()
            total = st.tensor("run_sums")[:17] - run_before + ended[None, :] * st.tensor("ep_sums")[:17]
            self.terms = total.float().mean(dim=1) / T

    def collect(self):
        if self.segment is not None:
            return self.collect_fused()
        torch, cfg = self.torch, self.cfg
        T = cfg.rollout
        term_sum = None
        with torch.no_grad():
            std = self.ac.log_std.exp()
            for t in range(T):
                o = self.obs[t]
                mean = self.ac.act_mean(o)
                a = mean + std * torch.randn_like(mean)
                self.act[t].copy_(a)
                self.logp[t].copy_(gaussian_logp(mean, self.ac.log_std, a))
                self.val[t].copy_(self.ac.values(o))
                # Sample Factory hands the sampled action to the env as is; RawControl clips it (quadrotor_control.py:53-57)
                self.env.set_training_info({"approx_total_training_steps": self.agent_steps})   # what the annealing schedule reads
                nxt, rew, term, trunc, infos = self.env.step(self.act[t])
                self.obs[t + 1].copy_(nxt["obs"])
                self.rew[t].copy_(rew)
                self.done[t].copy_(term)
                if self._rew_info is not None:
                    ri = self._rew_info().float().mean(dim=1)
                    term_sum = ri if term_sum is None else term_sum + ri
                if infos:   # (EpisodeInfos builds its dicts lazily: counting the finished agents builds none)
                    self.episodes += len(infos.finished_agents()) if hasattr(infos, "finished_agents") else sum(1 for d in infos if d)
                self.agent_steps += self.A
            self.val[T].copy_(self.ac.values(self.obs[T]))
        self.terms = None if term_sum is None else (term_sum / T)

    def advantages(self):
        """GAE (lambda = --gae_lambda) on the clipped, scaled rewards; an episode end (time limit, quadrotor_single.py:352-353)
        cuts the bootstrap the way SF does with --value_bootstrap=False"""
        torch, cfg = self.torch, self.cfg
        T = cfg.rollout
        rew = (self.rew * cfg.reward_scale).clamp(-cfg.reward_clip, cfg.reward_clip)
        adv = torch.zeros_like(self.rew)
        last = torch.zeros_like(self.rew[0])
        for t in reversed(range(T)):
            nd = 1.0 - self.done[t]
            delta = rew[t] + cfg.gamma * self.val[t + 1] * nd - self.val[t]
            last = delta + cfg.gamma * cfg.gae_lambda * nd * last
            adv[t] = last
        return adv, adv + self.val[:T]

    def _minibatch_step(self, o, a, lp_old, v_old, ad, rt, acc):
        """one optimiser step on a minibatch (plain tensors in, statistics accumulated into `acc` on the device): what update() runs eagerly, or
        records once into a HIP graph (--graph_update) and replays"""
        torch, cfg = self.torch, self.cfg
        hi = 1.0 + cfg.ppo_clip_ratio
        lo = 1.0 / hi
        mean = self.ac.act_mean(o)
        logp = gaussian_logp(mean, self.ac.log_std, a)
        v = self.ac.values(o)
        ad = (ad - ad.mean()) / ad.std().clamp_min(1e-7)   # SF normalises advantages per batch
        ratio = (logp - lp_old).exp()
        ploss = -torch.min(ratio * ad, ratio.clamp(lo, hi) * ad).mean()
        vclip = v_old + (v - v_old).clamp(-cfg.ppo_clip_value, cfg.ppo_clip_value)
        vloss = torch.max((v - rt) ** 2, (vclip - rt) ** 2).mean()
        entropy = (self.ac.log_std + 0.5 * math.log(2.0 * math.pi * math.e)).sum()
        loss = ploss + cfg.value_loss_coeff * vloss - cfg.exploration_loss_coeff * entropy
        self.opt.zero_grad(set_to_none=False)
        loss.backward()
        if cfg.max_grad_norm > 0:
            torch.nn.utils.clip_grad_norm_(self.ac.parameters(), cfg.max_grad_norm, foreach=True)
        self.opt.step()
        with torch.no_grad():
            acc += torch.stack((ploss.detach(), vloss.detach(), (lp_old - logp).mean().detach(), ((ratio < lo) | (ratio > hi)).float().mean()))

    def _capture(self, acc):
        """the minibatch step as ONE HIP graph over static input buffers: the eager step is ~ 300 small kernels at ~ 10 us of launch cost each
        (3.6 ms per 1024-sample minibatch on MI355X), the replay runs them back to back.  Returns False (and stays eager) where capture fails."""
        torch, cfg = self.torch, self.cfg
        B = cfg.batch_size
        f32 = dict(dtype=torch.float32, device=self.device)
        self._static = [torch.zeros((B, self.D), **f32), torch.zeros((B, 4), **f32)] + [torch.zeros((B,), **f32) for _ in range(4)]
        try:
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):   # warm-up on a side stream (allocator, autograd buffers, optimiser state) - three REAL but zero-data steps would
                for _ in range(3):          # move the weights, so the warm-up runs at a learning rate of zero
                    for g in self.opt.param_groups:
                        g["lr"].zero_() if torch.is_tensor(g["lr"]) else None
                    self._minibatch_step(*self._static, acc)
            torch.cuda.current_stream().wait_stream(side)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                self._minibatch_step(*self._static, acc)
            for g in self.opt.param_groups:
                g["lr"].fill_(cfg.learning_rate)
            for st in self.opt.state.values():   # the warm-up steps moved no weight (learning rate 0) but did feed the moment estimates: back to a fresh
                for v in st.values():            # optimiser, in place (the graph holds these tensors' addresses)
                    if torch.is_tensor(v):
                        v.zero_()
            acc.zero_()
            self._graph = graph
            return True
        except Exception as exc:   # noqa: BLE001 - the eager path is always there
            self._graph, self._graph_error = None, f"{type(exc).__name__}: {exc}"
            return False

    def update(self):
        torch, cfg = self.torch, self.cfg
        T, A = cfg.rollout, self.A
        adv, ret = self.advantages()
        n = T * A
        obs, act = self.obs[:T].reshape(n, self.D), self.act.reshape(n, 4)
        logp_old, val_old, adv, ret = self.logp.reshape(n), self.val[:T].reshape(n), adv.reshape(n), ret.reshape(n)
        stats = dict(policy_loss=0.0, value_loss=0.0, kl=0.0, clip_frac=0.0, updates=0)
        acc = self._acc
        acc.zero_()
        if self._want_graph and self._graph is None and self._graph_error is None:
            self._capture(acc)
        for _ in range(cfg.num_epochs):
            perm = torch.randperm(n, device=self.device)
            for s in range(0, n - cfg.batch_size + 1, cfg.batch_size):
                idx = perm[s:s + cfg.batch_size]
                if self._graph is not None:
                    for dst, src in zip(self._static, (obs, act, logp_old, val_old, adv, ret)):
                        torch.index_select(src, 0, idx, out=dst)
                    self._graph.replay()
                else:
                    self._minibatch_step(obs[idx], act[idx], logp_old[idx], val_old[idx], adv[idx], ret[idx], acc)
                stats["updates"] += 1
        u = max(stats["updates"], 1)
        pl, vl, kl, cf = (acc / u).tolist()
        stats.update(policy_loss=pl, value_loss=vl, kl=kl, clip_frac=cf, graph_update=self._graph is not None)
        self.obs[0].copy_(self.obs[T])
        return stats


def train(cfg, env=None, log=None):
    """-> list of per-iteration records (dicts).  `env`: a ready BatchedQuadSwarm (or a stand-in with its call protocol)."""
    import torch
    from quad_swarm_rl_amd import config as qcfg
    own = env is None
    if own:
        env = make_env(cfg)
    lr = Learner(cfg, env)
    per_iter = cfg.rollout * lr.A
    iters = cfg.iterations if not cfg.train_for_env_steps else max(1, -(-cfg.train_for_env_steps // per_iter))
    cuda = lr.device.type == "cuda"
    recs = []
    t_start = time.time()
    for it in range(iters):
        if cuda:
            torch.cuda.synchronize()
        t0 = time.time()
        lr.collect()
        if cuda:
            torch.cuda.synchronize()
        t1 = time.time()
        st = lr.update()
        if cuda:
            torch.cuda.synchronize()
        t2 = time.time()
        rec = dict(iteration=it, agent_steps=lr.agent_steps, reward_mean=float(lr.rew.mean()), value_mean=float(lr.val.mean()),
                   action_std=[round(float(x), 4) for x in lr.ac.log_std.detach().exp()], collect_s=round(t1 - t0, 3), update_s=round(t2 - t1, 3),
                   fps=round(per_iter / (t2 - t0), 1), sample_fps=round(per_iter / (t1 - t0), 1), episodes=lr.episodes, **{k: round(v, 6) if isinstance(v, float) else v for k, v in st.items()})
        if lr.sampler_gap is not None:
            rec["sampler_action_mean_gap"] = lr.sampler_gap
        if lr.terms is not None:
            tm = lr.terms.tolist()
            rec["terms"] = {k: round(tm[j], 6) for j, k in enumerate(qcfg.REW_INFO_KEYS[:len(tm)])}
        recs.append(rec)
        if log is not None:
            log(rec)
    total = time.time() - t_start
    summary = dict(c5="ran (in-tree PPO harness: Sample Factory is not installed)", iterations=iters, agent_steps=lr.agent_steps,
                   seconds=round(total, 2), fps=round(lr.agent_steps / total, 1), agents=lr.A, rollout=cfg.rollout, batch_size=cfg.batch_size,
                   first=_brief(recs[0]), last=_brief(recs[-1]), graph_update=bool(recs[-1].get("graph_update")), graph_error=lr._graph_error,
                   sampler=lr.sampler_note, sampler_action_mean_gap_max=max((r.get("sampler_action_mean_gap", 0.0) for r in recs), default=None),
                   sample_fps_median=sorted(r["sample_fps"] for r in recs)[len(recs) // 2])
    if own:
        env.close()
    return recs, summary


def _brief(rec):
    out = dict(reward_mean=round(rec["reward_mean"], 5))
    for k in ("rew_pos", "rew_crash", "rew_orient", "rew_spin", "rew_action"):
        if "terms" in rec and k in rec["terms"]:
            out[k] = round(rec["terms"][k], 5)
    return out


def main(argv=None):
    cfg = parse(argv)
    log = None if cfg.quiet else (lambda rec: print(json.dumps(rec), flush=True))
    _, summary = train(cfg, log=log)
    print(json.dumps(summary), flush=True)
    return 0


if __name__ == "__main__":
    sys.exit(main())
