"""Sample-Factory facing surface, mirroring swarm_rl/env_wrappers/ of the reference:

  quad_utils.py:20-117        make_quadrotor_env_multi / make_quadrotor_env   -> make_quadrotor_env (same signature)
  quadrotor_params.py:4-120   quadrotors_override_defaults / add_quadrotors_env_args (same flag names and defaults)
  reward_shaping.py:52-123    QuadsRewardShapingWrapper   -> BatchedQuadSwarm: the per-episode sums run in the step kernel (episode_sums),
                                                             the host keeps the shaping scheme, the annealing schedule and the assembly
                                                             of the episode-end `infos` (EpisodeInfoBuilder)
  compatibility.py:21-50      QuadEnvCompatibility        -> the 5-tuple / `reset(seed, options)` surface of BatchedQuadSwarm and of
                                                             SingleQuadSwarm (one environment, lists / numpy in and out)
  swarm_rl/train.py:16-19     register_swarm_components

The env underneath is always the HIP stepper - one environment is a batch of one; nothing here falls back to a CPU simulator.
"""
import copy

import numpy as np


def str2bool(v):
    if isinstance(v, bool):
        return v
    if str(v).lower() in ("true", "1", "yes", "y", "t"):
        return True
    if str(v).lower() in ("false", "0", "no", "n", "f"):
        return False
    raise ValueError(f"boolean flag expected, got {v!r}")


def quadrotors_override_defaults(env, parser):
    parser.set_defaults(encoder_type="mlp", encoder_subtype="mlp_quads", rnn_size=256, encoder_extra_fc_layers=0, env_frameskip=1)


def add_quadrotors_env_args(env, parser):
    """Same names/defaults as swarm_rl/env_wrappers/quadrotor_params.py:15-120 (+ the stepper's own three flags)."""
    p = parser
    p.add_argument("--quads_num_agents", default=8, type=int)
    p.add_argument("--quads_obs_repr", default="xyz_vxyz_R_omega", type=str,
                   choices=["xyz_vxyz_R_omega", "xyz_vxyz_R_omega_floor", "xyz_vxyz_R_omega_wall"])
    p.add_argument("--quads_episode_duration", default=15.0, type=float)
    p.add_argument("--quads_encoder_type", default="corl", type=str)
    p.add_argument("--quads_neighbor_visible_num", default=-1, type=int)
    p.add_argument("--quads_neighbor_obs_type", default="none", type=str, choices=["none", "pos_vel"])
    p.add_argument("--quads_neighbor_hidden_size", default=256, type=int)
    p.add_argument("--quads_neighbor_encoder_type", default="attention", type=str,
                   choices=["attention", "mean_embed", "mlp", "no_encoder"])
    p.add_argument("--quads_collision_reward", default=0.0, type=float)
    p.add_argument("--quads_collision_hitbox_radius", default=2.0, type=float)
    p.add_argument("--quads_collision_falloff_radius", default=-1.0, type=float)
    p.add_argument("--quads_collision_smooth_max_penalty", default=10.0, type=float)
    p.add_argument("--quads_use_obstacles", default=False, type=str2bool)
    p.add_argument("--quads_obstacle_obs_type", default="none", type=str, choices=["none", "octomap"])
    p.add_argument("--quads_obst_density", default=0.2, type=float)
    p.add_argument("--quads_obst_size", default=1.0, type=float)
    p.add_argument("--quads_obst_spawn_area", nargs="+", default=[6.0, 6.0], type=float)
    p.add_argument("--quads_domain_random", default=False, type=str2bool)
    p.add_argument("--quads_obst_density_random", default=False, type=str2bool)
    p.add_argument("--quads_obst_density_min", default=0.05, type=float)
    p.add_argument("--quads_obst_density_max", default=0.2, type=float)
    p.add_argument("--quads_obst_size_random", default=False, type=str2bool)
    p.add_argument("--quads_obst_size_min", default=0.3, type=float)
    p.add_argument("--quads_obst_size_max", default=0.6, type=float)
    p.add_argument("--quads_obst_hidden_size", default=256, type=int)
    p.add_argument("--quads_obst_encoder_type", default="mlp", type=str)
    p.add_argument("--quads_obst_collision_reward", default=0.0, type=float)
    p.add_argument("--quads_use_downwash", default=False, type=str2bool)
    p.add_argument("--quads_use_numba", default=False, type=str2bool)
    p.add_argument("--quads_mode", default="static_same_goal", type=str,   # the reference's choices verbatim (quadrotor_params.py:77-83):
                   choices=["static_same_goal", "static_diff_goal", "dynamic_same_goal", "dynamic_diff_goal", "ep_lissajous3D",   # they name
                            "ep_rand_bezier", "swarm_vs_swarm", "swap_goals", "dynamic_formations", "mix",   # four scenarios that have no
                            "o_uniform_same_goal_spawn", "o_random", "o_dynamic_diff_goal", "o_dynamic_same_goal", "o_diagonal",   # file and
                            "o_static_same_goal", "o_static_diff_goal", "o_swap_goals", "o_ep_rand_bezier"])   # omit run_away
    p.add_argument("--quads_room_dims", nargs="+", default=[10.0, 10.0, 10.0], type=float)
    p.add_argument("--replay_buffer_sample_prob", default=0.0, type=float)
    p.add_argument("--anneal_collision_steps", default=0.0, type=float)
    p.add_argument("--quads_view_mode", nargs="+", default=["topdown", "chase", "global"], type=str,
                   choices=["topdown", "chase", "side", "global", "corner0", "corner1", "corner2", "corner3", "topdownfollow"])
    p.add_argument("--quads_render", default=False, type=bool)
    p.add_argument("--visualize_v_value", action="store_true")
    p.add_argument("--quads_sim2real", default=False, type=str2bool)
    # the stepper's own flags
    p.add_argument("--quads_seed", default=0, type=int, help="seed of the counter-based noise stream")
    p.add_argument("--quads_num_envs", default=1, type=int, help="> 1: one device-resident batched env of this many environments "
                                                                 "(num_agents = envs x quads) instead of the single-env facade")
    p.add_argument("--quads_device", default=0, type=int, help="HIP device index")
    p.add_argument("--quads_precision", default="f32", type=str, choices=["f32", "f64"])
    p.add_argument("--quads_backend", default="hip", type=str, help="'hip': the MI355X stepper.  There is no CPU simulator behind this flag: "
                                                                    "anything else is rejected")
    p.add_argument("--quads_num_gpus", default=1, type=int, help="> 1: --quads_num_envs is the GLOBAL batch, sharded over this many processes "
                                                                 "(one per GPU, launched with torchrun: RANK / LOCAL_RANK / WORLD_SIZE); each steps "
                                                                 "its contiguous env range on GPU LOCAL_RANK")
    p.add_argument("--quads_gather_obs", default=False, type=str2bool, help="with --quads_num_gpus > 1: after every step every rank also "
                                                                            "receives the observation rows of all shards (env.gathered_obs())")
    p.add_argument("--quads_obs_wire", default="bf16", type=str, choices=["q8", "bf16", "f32"], help="wire format of the gathered rows (q8: bf16 self columns + 8-bit fixed-point neighbour block, include/quadswarm_exchange.h; env.gathered_obs_f32() dequantises)")
    p.add_argument("--quads_obs_transport", default="rccl", type=str, choices=["rccl", "peer"],
                   help="how the gathered rows travel: rccl = torch.distributed all-gather (default: the transport that is validated wherever RCCL is); "
                        "peer = the library's peer stores into hipIpc-mapped windows (opt-in: start-up self-check + a comparison against an RCCL "
                        "gather of the same rows on every rank, else all ranks fall back to rccl together)")


DEFAULT_QUAD_REWARD_SHAPING = dict(quad_rewards=dict(pos=1.0, effort=0.05, spin=0.1, vel=0.0, crash=1.0, orient=1.0, yaw=0.0,
                                                     quadcol_bin=0.0, quadcol_bin_smooth_max=0.0, quadcol_bin_obst=0.0))


def shard_spec(num_envs, num_gpus, environ):
    """(envs of this shard, env_id_offset, rank, local_rank) of the calling process for a global batch of `num_envs` environments
    sharded over `num_gpus` processes, one per GPU (contiguous env ranges; the noise stream is keyed by the global
    env id, so the shards together are the un-sharded batch).  `environ`: os.environ of a torchrun-launched process."""
    if num_gpus < 1:
        raise ValueError("--quads_num_gpus must be >= 1")
    if num_gpus == 1:
        return num_envs, 0, 0, None
    world = int(environ.get("WORLD_SIZE", "1"))
    if world != num_gpus:
        raise ValueError(f"--quads_num_gpus={num_gpus} shards the batch over {num_gpus} processes (one per GPU), but WORLD_SIZE={world}: launch with "
                         f"python -m torch.distributed.run --nnodes=1 --nproc-per-node {num_gpus} --master-addr 127.0.0.1 ...")
    if num_envs % num_gpus:
        raise ValueError("--quads_num_envs must be divisible by --quads_num_gpus")
    rank, per = int(environ.get("RANK", "0")), num_envs // num_gpus
    return per, rank * per, rank, int(environ.get("LOCAL_RANK", str(rank)))


class AnnealSchedule:
    def __init__(self, coeff_name, final_value, anneal_env_steps):
        self.coeff_name, self.final_value, self.anneal_env_steps = coeff_name, final_value, anneal_env_steps


class EpisodeInfoBuilder:
    """Per-agent `infos` of a batch of finished episodes - {'true_reward', 'episode_extra_stats'} as the reference's wrapper stack
    attaches them (quadrotor_multi.py:626-718, quad_experience_replay.py:126-138, reward_shaping.py:85-118) - from host copies of the
    device arrays: sums [25, F*n] (reward-term sums and action moments of the finished episodes), eps [6, F*n] / cnt [11, F] (the env's
    episode statistics), scen_ids [F], rs (qs_replay_stats dict or None).  Everything that can be computed for all finished
    environments at once is (a handful of numpy reductions in the constructor); `env_dicts(f)` then builds the n dicts of ONE finished
    environment - on demand, so a step on which 1024 environments finish does not build 8192 dicts before anybody asks for one."""

    def __init__(self, finished, n, sums, eps, cnt, scen_ids, rs, obst_density, obst_size, approx, keys, use_obstacles, ep_steps, annealed, cur_scen_ids=None):
        from . import config as qcfg
        F = len(finished)
        self.finished, self.n, self.rs, self.approx, self.annealed, self.use_obstacles = finished, n, rs, approx, annealed, use_obstacles
        nkeys = len(keys) if use_obstacles else 15
        self.rew_keys = list(keys[:nkeys])
        self.i_main = self.rew_keys.index("rewraw_main")
        self.i_quadcol = self.rew_keys.index("rewraw_quadcol") if "rewraw_quadcol" in self.rew_keys else -1
        self.i_pos, self.i_crash = self.rew_keys.index("rew_pos"), self.rew_keys.index("rew_crash")
        self.sums_rows = np.ascontiguousarray(sums[:nkeys].T).reshape(F, n, nkeys)       # per agent: the reward-term sums
        self.dist = np.ascontiguousarray(eps[:3].T).reshape(F, n, 3)
        e3 = eps[3:6].reshape(3, F, n)
        ok = np.logical_and(e3[1], e3[2])
        self.succ = np.sum(np.logical_and(ok, e3[0]), axis=1) / n
        self.dead = np.sum(np.logical_and(ok, 1 - e3[0]), axis=1) / n
        self.col, self.ncol, self.ocol = 1.0 - np.sum(ok, axis=1) / n, 1.0 - np.sum(e3[1], axis=1) / n, 1.0 - np.sum(e3[2], axis=1) / n
        self.cnt = np.asarray(cnt)
        self.names = [qcfg.SCENARIO_CLASS_NAMES[int(x)] for x in scen_ids]
        # The shaping wrapper names its two per-scenario keys AFTER env.step() returned (reward_shaping.py:95-98), i.e. after the
        # auto-reset: under `mix` that is the scenario of the episode that STARTS, not of the one the sums belong to.  Reproduced.
        self.cur_names = self.names if cur_scen_ids is None else [qcfg.SCENARIO_CLASS_NAMES[int(x)] for x in cur_scen_ids]
        # action moments over agents x steps of the episode (np.mean / np.std of reward_shaping.py:103-108); a replayed episode starts at
        # its checkpoint's tick and is shorter than ep_len + 1 steps
        steps = np.asarray(rs["ep_steps"])[np.asarray(finished, dtype=np.int64)] if rs is not None else np.full(F, ep_steps)
        count = steps.astype(np.float64) * n
        a1 = sums[17:21].reshape(4, F, n).sum(axis=2) / count
        a2 = sums[21:25].reshape(4, F, n).sum(axis=2) / count
        self.a_mean, self.a_std = a1, np.sqrt(np.maximum(a2 - a1 * a1, 0.0))
        self.obst_density, self.obst_size = obst_density, obst_size

    def env_dicts(self, f):
        e, n, rs = int(self.finished[f]), self.n, self.rs
        scenario_name = self.names[f]
        name = scenario_name[9:]
        replayed_episode = rs is not None and bool(rs["ep_was_replay"][e])
        c = [int(x) for x in self.cnt[:, f]]
        if replayed_episode:
            base = {"num_collisions_replay": c[0], "num_collisions_obst_replay": c[7]}
        else:   # env-level part of assemble_episode_extra_stats (env.py); the per-agent distances are added below
            succ, dead, col, ncol, ocol = float(self.succ[f]), float(self.dead[f]), float(self.col[f]), float(self.ncol[f]), float(self.ocol[f])
            base = {"num_collisions": c[0], "num_collisions_with_room": c[3], "num_collisions_with_floor": c[4], "num_collisions_with_wall": c[5],
                    "num_collisions_with_ceiling": c[6], "num_collisions_after_settle": c[1], f"{name}/num_collisions": c[1],
                    "num_collisions_final_5_s": c[2], f"{name}/num_collisions_final_5_s": c[2],
                    "distance_to_goal_1s": 0.0, "distance_to_goal_3s": 0.0, "distance_to_goal_5s": 0.0,
                    f"{name}/distance_to_goal_1s": 0.0, f"{name}/distance_to_goal_3s": 0.0, f"{name}/distance_to_goal_5s": 0.0,
                    "metric/agent_success_rate": succ, f"{name}/agent_success_rate": succ, "metric/agent_deadlock_rate": dead, f"{name}/agent_deadlock_rate": dead,
                    "metric/agent_col_rate": col, f"{name}/agent_col_rate": col, "metric/agent_neighbor_col_rate": ncol, f"{name}/agent_neighbor_col_rate": ncol,
                    "metric/agent_obst_col_rate": ocol, f"{name}/agent_obst_col_rate": ocol}
            if self.use_obstacles:
                base.update({"num_collisions_obst_quad": c[7], "num_collisions_obst_quad_after_settle": c[8], f"{name}/num_collisions_obst": c[7],
                             "num_collisions_obst_quad_3_5": c[9], f"{name}/num_collisions_obst_quad_3_5": c[9], "num_collisions_obst_quad_5": c[10],
                             f"{name}/num_collisions_obst_quad_5": c[10]})
        if rs is not None:
            ep, rp, nb = int(rs["episodes"][e]), int(rs["replayed"][e]), int(rs["buffer_len"][e])
            base.update({"replay/replay_rate": rp / ep, "replay/new_episode_rate": (ep - rp) / ep, "replay/replay_buffer_size": nb,
                         "replay/avg_replayed": (int(rs["replayed_sum"][e]) / nb) if nb else 0,
                         "replay/obst_density": float(self.obst_density[e]), "replay/obst_size": float(self.obst_size[e])})
        rew_keys = self.rew_keys
        for key in rew_keys:      # placeholders keep the key order of the reference's dicts: env stats, replay stats, reward sums, z_* keys
            base[key] = 0.0
        base["z_approx_total_training_steps"] = self.approx
        k_pos, k_crash = f"{self.cur_names[f]}/rew_pos", f"{self.cur_names[f]}/rew_crash"
        base[k_pos] = base[k_crash] = 0.0
        for q in range(4):
            base[f"z_action{q}_mean"], base[f"z_action{q}_std"] = float(self.a_mean[q, f]), float(self.a_std[q, f])
        for key, val in self.annealed:
            base[key] = val
        dist_keys = None if replayed_episode else ("distance_to_goal_1s", "distance_to_goal_3s", "distance_to_goal_5s",
                                                   f"{name}/distance_to_goal_1s", f"{name}/distance_to_goal_3s", f"{name}/distance_to_goal_5s")
        rows, dist = self.sums_rows[f].tolist(), self.dist[f].tolist()
        i_main, i_quadcol, i_pos, i_crash = self.i_main, self.i_quadcol, self.i_pos, self.i_crash
        out = []
        for k in range(n):
            row = rows[k]
            true_reward = row[i_main] + (1000 * row[i_quadcol] if i_quadcol >= 0 else 0)
            extra = dict(base)
            if dist_keys is not None:
                d1, d3, d5 = dist[k]
                extra[dist_keys[0]] = extra[dist_keys[3]] = d1
                extra[dist_keys[1]] = extra[dist_keys[4]] = d3
                extra[dist_keys[2]] = extra[dist_keys[5]] = d5
            extra.update(zip(rew_keys, row))
            extra["rewraw_main"] = true_reward
            extra[k_pos], extra[k_crash] = row[i_pos], row[i_crash]
            out.append({"true_reward": true_reward, "episode_extra_stats": extra})
        return out


def assemble_batched_infos(finished, n, sums, eps, cnt, scen_ids, rs, obst_density, obst_size, approx, keys, use_obstacles, ep_steps, annealed, infos):
    """eager form of EpisodeInfoBuilder: fills infos[e * n + k] for every agent of every finished environment"""
    b = EpisodeInfoBuilder(finished, n, sums, eps, cnt, scen_ids, rs, obst_density, obst_size, approx, keys, use_obstacles, ep_steps, annealed)
    for f, e in enumerate(finished):
        for k, d in enumerate(b.env_dicts(f)):
            infos[int(e) * n + k].update(d)


class EpisodeInfos(list):
    """What BatchedQuadSwarm.step returns as `infos` on a step where episodes ended: a list with one entry per agent, {} for the agents
    still flying, {'true_reward', 'episode_extra_stats'} for those whose episode ended.  The dicts of a finished environment are built
    when one of its agents is first asked for (indexing or iterating) - a sampler that looks at the finished agents pays for exactly
    those, nobody pays 60-120 ms on the step where all 8192 agents of a batch finish together.  The underlying list storage is only
    filled by `materialize()` (copying / pickling the object calls it)."""

    def __init__(self, num_agents, n, builder):
        super().__init__()
        self._num, self._n, self._b = num_agents, n, builder
        self._slot = {int(e): f for f, e in enumerate(builder.finished)}
        self._built = {}

    def _env(self, e):
        f = self._slot.get(e)
        if f is None:
            return None
        d = self._built.get(f)
        if d is None:
            d = self._built[f] = self._b.env_dicts(f)
        return d

    def __len__(self):
        return self._num

    def __bool__(self):
        return self._num > 0

    def __getitem__(self, i):
        if isinstance(i, slice):
            return [self[j] for j in range(*i.indices(self._num))]
        if i < 0:
            i += self._num
        if not 0 <= i < self._num:
            raise IndexError("list index out of range")
        e, k = divmod(i, self._n)
        d = self._env(e)
        return {} if d is None else d[k]

    def __iter__(self):
        n = self._n
        for e in range(self._num // n):
            d = self._env(e)
            if d is None:
                for _ in range(n):
                    yield {}
            else:
                yield from d

    def __eq__(self, other):
        return list(self) == other

    def __ne__(self, other):
        return not self == other

    __hash__ = None

    def finished_agents(self):
        """indices of the agents whose episode ended on this step"""
        n = self._n
        return [e * n + k for e in sorted(self._slot) for k in range(n)]

    def materialize(self):
        """a plain list of dicts (everything built)"""
        return list(self)

    def __reduce__(self):
        return (list, (self.materialize(),))

    def __repr__(self):
        return f"EpisodeInfos({self._num} agents, {len(self._slot)} finished envs, {len(self._built)} built)"


class BatchedQuadSwarm:
    """E environments x N drones behind one object with the batched-sampling shape of Sample Factory (num_agents = E*N,
    device tensors in and out) and the semantics of the reference's wrapper stack - ExperienceReplayWrapper
    (quad_experience_replay.py:66-209), QuadsRewardShapingWrapper (reward_shaping.py:22-123), QuadEnvCompatibility
    (compatibility.py:21-50) - with the per-step bookkeeping on the GPU: the step kernel keeps the per-episode sums of the reward
    terms and action moments (`episode_sums`), the replay kernel keeps every environment's checkpoint ring / event buffer /
    activation state (include/quadswarm.h: qs_replay_enable).  A control step costs one kernel launch (two with replay) and no
    device->host traffic; the host knows from the environments' ticks when the next episode can end and touches the device only
    on those steps, to build the `infos` dicts of the finished environments (gathered on the device first, so the transfer is
    proportional to what finished).  SURVEY.md 8f rank 1 / 3.
    Call protocol (Sample Factory's vectorised-env convention, tests/test_sf_protocol_gpu.py): `reset() -> (obs_dict, info)`,
    `step(actions) -> (obs_dict, rewards, terminated, truncated, infos)`, torch tensors of leading dimension num_agents; `infos`
    is a list of num_agents dicts on steps where an episode ended (empty dicts for the others) and `[]` otherwise."""

    def __init__(self, num_envs, reward_shaping_scheme=None, annealing=None, device=0, seed=0, replay_buffer_sample_prob=0.0,
                 num_gpus=1, gather_obs=False, obs_wire="bf16", obs_transport="rccl", write_rew_info=False, _vec=None, **env_kwargs):
        """num_gpus > 1: `num_envs` is the global batch; this process (one per GPU under torchrun) steps its contiguous shard on GPU
        LOCAL_RANK.  gather_obs: the observation rows of all shards are exchanged after every step (parallel.ObsExchange) and available
        from gathered_obs()."""
        import os
        import torch
        from . import config as qcfg
        from .env import QuadSwarmVecEnv
        self._torch = torch
        self.global_num_envs = num_envs
        num_envs, env_id_offset, self.rank, local_rank = shard_spec(num_envs, num_gpus, os.environ)
        self.num_gpus = num_gpus
        if local_rank is not None:
            device = local_rank
        # (_vec: a ready-made vec env - the CPU test of the shaping / annealing / infos logic drives this class over a scripted stand-in)
        self.vec = _vec if _vec is not None else QuadSwarmVecEnv(num_envs, device=device, seed=seed, env_id_offset=env_id_offset, episode_sums=True,
                                                                 write_rew_info=write_rew_info, **env_kwargs)
        self.use_replay_buffer = replay_buffer_sample_prob > 0.0        # quad_utils.py:34
        self.replay_buffer_sample_prob = float(replay_buffer_sample_prob)
        if self.use_replay_buffer:
            self.vec.stepper.replay_enable(self.replay_buffer_sample_prob)
        if gather_obs:   # (after the replay wrapper: with it the exchange sends the rows the replay kernel leaves in the library's buffer -
            #               restored checkpoints included - instead of redirecting the step kernel's output: parallel.ObsExchange source="obs")
            self.vec.attach_exchange(self._make_exchange(num_gpus, obs_wire, obs_transport))
        self.num_envs, self.agents_per_env = num_envs, self.vec.num_agents_per_env
        self.num_agents = self.vec.num_agents
        self.is_multiagent = True
        self.observation_space, self.action_space = self.vec.observation_space, self.vec.action_space
        self.rew_coeff = self.vec.rew_coeff
        self.scenario = self.vec.scenario
        self.reward_shaping_scheme = reward_shaping_scheme or copy.deepcopy(DEFAULT_QUAD_REWARD_SHAPING)
        self.reward_shaping_updated = True
        self.annealing = annealing
        self.training_info = {}
        self._keys = qcfg.REW_INFO_KEYS
        self._warm = False                                # the torch kernels of the episode-end path are loaded by the first reset()
        self._ep_steps = self.vec.cfg.ep_len + 1          # an episode ends by time: tick > ep_len (quadrotor_single.py:353)
        self._steps_to_done = self._ep_steps              # control steps until the earliest possible episode end
        self._truncated = None

    def _make_exchange(self, world, wire, transport="rccl"):
        """The exchange of the observation rows between the shards.  Default: RCCL all-gather of the packed rows.  transport="peer"
        (opt-in, --quads_obs_transport peer): the library's peer stores into hipIpc-mapped windows (include/quadswarm_exchange.h), kept
        only if on EVERY rank the windows could be mapped, the start-up self-check passed and the rows of a first exchanged reset equal
        what an RCCL gather of the same rows delivers (ObsExchange.verify); otherwise all ranks fall back to RCCL together."""
        import torch
        import torch.distributed as dist
        from . import parallel
        st = self.vec.stepper
        torch.cuda.set_device(st.device)
        if world > 1 and not dist.is_initialized():
            dist.init_process_group("nccl", device_id=torch.device("cuda", st.device))
        self.obs_transport = "rccl"
        if transport == "rccl":
            return parallel.ObsExchange(st, world, self.rank, transport="rccl", wire=wire, hold=True)

        def all_agree(flag):
            if world == 1:
                return flag
            t = torch.tensor([1 if flag else 0], device=f"cuda:{st.device}", dtype=torch.int32)
            dist.all_reduce(t, op=dist.ReduceOp.MIN)
            return bool(t.item())

        ex = None
        for fenced in (False, True):   # the relaxed flag protocol first, its fenced variant (QS_XCHG_FENCED) second, RCCL last
            ex = None
            try:
                ex = parallel.ObsExchange(st, world, self.rank, transport="peer", wire=wire, hold=True, fenced=fenced)
            except Exception:   # noqa: BLE001 - every rank takes the same decision below
                ex = None
            ok = all_agree(ex is not None)
            if ok:
                try:
                    ok = ex.self_check()[0]
                except Exception:   # noqa: BLE001
                    ok = False
                ok = all_agree(ok)
            if ok:   # ... and through the real producer of the rows: an exchanged reset, compared with an RCCL gather of the same rows.  The
                try:  # reset (no collective inside) is agreed on first; verify() agrees on its own local part before it enters its all-gather
                    ex.reset()
                except Exception:   # noqa: BLE001
                    ok = False
                ok = all_agree(ok)
            if ok:
                ok = all_agree(ex.verify()[0])
            if ok:
                break
            if ex is not None:
                ex.close()
                ex = None
        if ex is None:
            ex = parallel.ObsExchange(st, world, self.rank, transport="rccl", wire=wire, hold=True)
        self.obs_protocol = "rccl" if ex.transport == "rccl" else ("fenced flags" if ex.x.fenced else "relaxed flags")
        self.obs_transport = ex.transport
        return ex

    def gathered_obs(self):
        """[global envs * N, obs_dim] observation rows of ALL shards after the most recent step (rank order = global env order), in the
        wire format (--quads_obs_wire); valid until the next step().  Needs gather_obs=True."""
        if self.vec.exchange is None:
            raise RuntimeError("create the env with gather_obs=True (--quads_gather_obs=True)")
        return self.vec.exchange.latest()

    def gathered_obs_f32(self):
        """gathered_obs() as float32 [global envs * N, obs_dim] whatever the wire (bf16 widened, q8 dequantised on the device)"""
        if self.vec.exchange is None:
            raise RuntimeError("create the env with gather_obs=True (--quads_gather_obs=True)")
        return self.vec.exchange.latest_f32()

    def check_exchange(self):
        """raises QsError if the observation exchange ever timed out (parallel.ObsExchange.check); step() calls it on episode-end steps"""
        if self.vec.exchange is not None:
            self.vec.exchange.check()

    @property
    def unwrapped(self):
        return self

    def set_training_info(self, training_info):
        self.training_info = training_info

    # Sample Factory's RewardShapingInterface as the reference implements it (reward_shaping.py:36-47)
    def get_default_reward_shaping(self):
        return dict(quad_rewards=dict())

    def get_current_reward_shaping(self, agent_idx):
        return dict(quad_rewards=dict())

    def set_reward_shaping(self, reward_shaping, unused_agent_idx):
        self.reward_shaping_scheme = dict(quad_rewards=dict())
        self.reward_shaping_updated = True

    def render(self, *a, **k):
        return None

    def reset(self, seed=None, options=None):
        obs = self.vec.reset()
        self._steps_to_done = self._ep_steps
        if not self._warm:   # load the torch kernels of the episode-end path (gather / cast / copy) now: 35 - 80 ms the first time they
            self._warm = True   # run, which would otherwise land on the first step on which episodes end (tools/episode_end_probe.py)
            torch, st = self._torch, self.vec.stepper
            idx = torch.zeros(1, dtype=torch.long, device=st.tensor("ep_sums").device)
            for name, dim in (("ep_sums", 1), ("ep_stats", 1), ("ep_counters", 1), ("ep_scenario", 0), ("scenario_id", 0)):
                st.tensor(name).index_select(dim, idx).double().cpu()
            (idx[:, None] * self.agents_per_env + torch.arange(self.agents_per_env, device=idx.device)[None, :]).reshape(-1)
        return {"obs": obs}, {}

    def step(self, actions):
        torch = self._torch
        if self.reward_shaping_updated:   # reward_shaping.py:55-61
            for key, weight in self.reward_shaping_scheme["quad_rewards"].items():
                if key in self.rew_coeff:
                    self.rew_coeff[key] = weight
            self.reward_shaping_updated = False
        obs, rew, done, _ = self.vec.step(actions)
        if self._truncated is None:
            self._truncated = torch.zeros_like(done, dtype=torch.bool)
            self._terminated = done.view(torch.bool)   # the uint8 0 / 1 buffer reinterpreted: no kernel, no allocation - and, like `obs`
                                                       # and `rew`, a view of the live buffer that the next step() overwrites
        infos = []
        self._steps_to_done -= 1
        if self._steps_to_done <= 0:      # the only steps on which the host looks at the device
            st, n = self.vec.stepper, self.agents_per_env
            self.check_exchange()         # (a timed-out exchange means stale gathered rows since: raise instead of training on them)
            finished = np.nonzero(st.to_host("done").reshape(self.num_envs, n)[:, 0])[0]
            if len(finished):
                infos = self._episode_infos(finished)
            # every environment's tick after this step (a replayed episode starts at its checkpoint's tick): next possible end
            self._steps_to_done = int(self._ep_steps - st.to_host("tick").max())
        return {"obs": obs}, rew, self._terminated, self._truncated, infos

    # ---- rollout segments recorded into a HIP graph (rollout.GraphedRollout over self.vec): the host duties of step(), per SEGMENT ----
    def segment_begin(self):
        """Before a captured segment runs: what step() does on the host before the kernel - the shaping scheme pushed into the env's
        coefficients (reward_shaping.py:55-61), the coefficients pushed to the device (the kernels read them from device memory on every
        launch, so replays of a graph that was captured earlier see them)."""
        if self.reward_shaping_updated:
            for key, weight in self.reward_shaping_scheme["quad_rewards"].items():
                if key in self.rew_coeff:
                    self.rew_coeff[key] = weight
            self.reward_shaping_updated = False
        self.vec._sync_rew_coeff()

    def segment_end(self, dones):
        """After a captured segment of T control steps (dones: its [T, E*N] done flags): what step() does on the host behind the kernel, for
        all steps of the segment at once - the episode-end `infos` of every environment whose episode ended inside the segment (the sums of a
        finished episode stay in the device buffers until the environment's NEXT episode ends: segments are far shorter than episodes) and the
        annealing of the collision coefficients (reward_shaping.py:111-118), which therefore takes effect at the next segment instead of the
        next step.  Returns the EpisodeInfos (or [])."""
        torch = self._torch
        st, n = self.vec.stepper, self.agents_per_env
        self.check_exchange()
        ended = dones.reshape(dones.shape[0], self.num_envs, n)[:, :, 0].any(dim=0)
        finished = torch.nonzero(ended).reshape(-1).cpu().numpy()
        infos = self._episode_infos(finished) if len(finished) else []
        self.vec._sync_rew_coeff()
        self._steps_to_done = int(self._ep_steps - st.to_host("tick").max())
        return infos

    def _episode_infos(self, finished):
        """What the wrapper stack attaches at an episode end - the env's own episode_extra_stats (quadrotor_multi.py:626-718, or
        the two `*_replay` counters of a replayed episode, :629-633), the replay wrapper's statistics
        (quad_experience_replay.py:126-138), the reward-shaping wrapper's sums / action moments / annealed coefficients
        (reward_shaping.py:85-118) - for the agents of the finished envs."""
        from . import config as qcfg
        from .env import assemble_episode_extra_stats
        torch = self._torch
        st, n = self.vec.stepper, self.agents_per_env
        dev = st.tensor("ep_sums").device
        env_idx = torch.as_tensor(finished, device=dev, dtype=torch.long)
        agent_idx = (env_idx[:, None] * n + torch.arange(n, device=dev)[None, :]).reshape(-1)
        # gather on the device, then one transfer each: [25, F*n] sums, [6, F*n] episode stats, [11, F] counters, [F] scenario ids
        sums = st.tensor("ep_sums").index_select(1, agent_idx).double().cpu().numpy()
        eps = st.tensor("ep_stats").index_select(1, agent_idx).double().cpu().numpy()
        cnt = st.tensor("ep_counters").index_select(1, env_idx).cpu().numpy()
        scen_ids = st.tensor("ep_scenario").index_select(0, env_idx).cpu().numpy()
        cur_scen = st.tensor("scenario_id").index_select(0, env_idx).cpu().numpy()   # after the auto-reset (and a replay restore)
        rs = st.replay_stats() if self.use_replay_buffer else None
        if rs is not None:   # what the wrapper calls curr_obst_density / curr_obst_size: the values of the episode that starts now
            obst_density, obst_size = st.to_host("obst_density_env"), st.to_host("obst_size_env")
        approx = self.training_info.get("approx_total_training_steps", 0)
        annealed = []
        if self.annealing:   # :111-118, once per step on which episodes ended (same values for every agent)
            for sched in self.annealing:
                self.rew_coeff[sched.coeff_name] = min(sched.final_value * approx / sched.anneal_env_steps, sched.final_value)
                annealed.append((f"z_anneal_{sched.coeff_name}", self.rew_coeff[sched.coeff_name]))
        builder = EpisodeInfoBuilder(finished, n, sums, eps, cnt, scen_ids, rs, obst_density if rs is not None else None, obst_size if rs is not None else None,
                                     approx, self._keys, bool(self.vec.cfg.use_obstacles), self._ep_steps, annealed, cur_scen_ids=cur_scen)
        return EpisodeInfos(self.num_agents, n, builder)

    def close(self):
        try:
            self.check_exchange()
        finally:
            self.vec.close()


def _domain_random_kwargs(cfg, use_replay_buffer):
    """--quads_domain_random and its companions (quadrotor_params.py:52-60).  In the reference the per-episode draw of obstacle density
    / size is made by the replay wrapper (quad_experience_replay.py:75-88,:106-118,:191-206), which only exists with
    --replay_buffer_sample_prob > 0 (quad_utils.py:67-70): without it the flags have no effect there, and none here."""
    if not (use_replay_buffer and getattr(cfg, "quads_domain_random", False) and cfg.quads_use_obstacles):
        return {}
    return dict(domain_random=True, obst_density_random=bool(getattr(cfg, "quads_obst_density_random", False)),
                obst_size_random=bool(getattr(cfg, "quads_obst_size_random", False)),
                obst_density_min=cfg.quads_obst_density_min, obst_density_max=cfg.quads_obst_density_max,
                obst_size_min=cfg.quads_obst_size_min, obst_size_max=cfg.quads_obst_size_max)


def _shaping_from_cfg(cfg):
    """reward-shaping scheme and annealing schedule of quad_utils.py:72-107"""
    reward_shaping = copy.deepcopy(DEFAULT_QUAD_REWARD_SHAPING)
    reward_shaping["quad_rewards"]["quadcol_bin"] = cfg.quads_collision_reward
    reward_shaping["quad_rewards"]["quadcol_bin_smooth_max"] = cfg.quads_collision_smooth_max_penalty
    reward_shaping["quad_rewards"]["quadcol_bin_obst"] = cfg.quads_obst_collision_reward
    annealing = None
    if cfg.anneal_collision_steps > 0:
        for k in ("quadcol_bin", "quadcol_bin_smooth_max", "quadcol_bin_obst"):
            reward_shaping["quad_rewards"][k] = 0.0
        annealing = [AnnealSchedule("quadcol_bin", cfg.quads_collision_reward, cfg.anneal_collision_steps),
                     AnnealSchedule("quadcol_bin_smooth_max", cfg.quads_collision_smooth_max_penalty, cfg.anneal_collision_steps),
                     AnnealSchedule("quadcol_bin_obst", cfg.quads_obst_collision_reward, cfg.anneal_collision_steps)]
    return reward_shaping, annealing


def _env_kwargs_from_cfg(cfg):
    """QuadrotorEnvMulti's constructor arguments as make_quadrotor_env_multi derives them from the flags (quad_utils.py:20-65)"""
    use_replay_buffer = getattr(cfg, "replay_buffer_sample_prob", 0.0) > 0.0
    return dict(
        replay_buffer_sample_prob=getattr(cfg, "replay_buffer_sample_prob", 0.0),
        device=getattr(cfg, "quads_device", 0), seed=getattr(cfg, "quads_seed", 0), precision=getattr(cfg, "quads_precision", "f32"),
        num_agents=cfg.quads_num_agents, ep_time=cfg.quads_episode_duration, rew_coeff=dict(DEFAULT_QUAD_REWARD_SHAPING["quad_rewards"]),
        obs_repr=cfg.quads_obs_repr, neighbor_visible_num=cfg.quads_neighbor_visible_num, neighbor_obs_type=cfg.quads_neighbor_obs_type,
        collision_hitbox_radius=cfg.quads_collision_hitbox_radius, collision_falloff_radius=cfg.quads_collision_falloff_radius,
        use_obstacles=cfg.quads_use_obstacles, obst_density=cfg.quads_obst_density, obst_size=cfg.quads_obst_size,
        obst_spawn_area=cfg.quads_obst_spawn_area, use_downwash=cfg.quads_use_downwash, use_numba=cfg.quads_use_numba,
        quads_mode=cfg.quads_mode, room_dims=cfg.quads_room_dims,
        **_domain_random_kwargs(cfg, use_replay_buffer))


class SingleQuadSwarm:
    """`make_quadrotor_env_multi`'s product for ONE environment: the call protocol of the reference's wrapper stack
    (QuadrotorEnvMulti -> ExperienceReplayWrapper -> QuadsRewardShapingWrapper -> QuadEnvCompatibility, quad_utils.py:20-110) - lists /
    numpy arrays in and out, per-step infos[i]['rewards'], `true_reward` and `episode_extra_stats` at an episode end - over a
    BatchedQuadSwarm of one environment: the same kernels, the same on-device episode sums, the same host-side assembly as the
    batched env (tests/test_facade_gpu.py compares the two).  A compatibility path: every step copies actions in and observations,
    rewards and reward terms out."""

    is_multiagent = True

    def __init__(self, batched):
        from . import config as qcfg
        self._b, self._vec = batched, batched.vec
        self.num_agents = batched.agents_per_env
        self.observation_space, self.action_space = batched.observation_space, batched.action_space
        self.rew_coeff, self.scenario = batched.rew_coeff, batched.scenario
        self.use_replay_buffer = batched.use_replay_buffer
        self.use_obstacles = bool(self._vec.cfg.use_obstacles)
        self.control_freq = 1.0 / (self._vec.cfg.dt * self._vec.cfg.sim_steps)
        self._keys = qcfg.REW_INFO_KEYS if self.use_obstacles else qcfg.REW_INFO_KEYS[:15]
        self.envs = [self]                                   # `env.envs[0].tick` of the reference (the replay wrapper reads it)

    # what the reference's wrappers and tests reach for on the inner env
    @property
    def unwrapped(self):
        return self

    @property
    def tick(self):
        return int(self._vec.stepper.to_host("tick")[0])

    @property
    def activate_replay_buffer(self):
        return bool(self.use_replay_buffer and self._vec.stepper.replay_stats()["active"][0])

    @activate_replay_buffer.setter
    def activate_replay_buffer(self, value):
        self._vec.stepper.replay_set_active([1 if value else 0])

    def set_training_info(self, training_info):
        self._b.set_training_info(training_info)

    def get_default_reward_shaping(self):
        return self._b.get_default_reward_shaping()

    def get_current_reward_shaping(self, agent_idx):
        return self._b.get_current_reward_shaping(agent_idx)

    def set_reward_shaping(self, reward_shaping, agent_idx):
        self._b.set_reward_shaping(reward_shaping, agent_idx)

    def render(self, *a, **k):
        return None

    def reset(self, seed=None, options=None):
        obs, info = self._b.reset()
        return obs["obs"].double().cpu().numpy(), info

    def step(self, actions):
        import torch
        st = self._vec.stepper
        a = torch.as_tensor(np.ascontiguousarray(np.asarray(actions, dtype=st.np_real).reshape(self.num_agents, 4)), device=f"cuda:{st.device}")
        obs, rew, term, trunc, ep_infos = self._b.step(a)
        st.sync(stream=torch.cuda.current_stream(st.device))
        st.check_errors()                                    # ValueError('QuadEnv: reward is Nan'), quadrotor_single.py:87-90
        ri = st.to_host("rew_info")                          # [17, N]: the infos[i]['rewards'] terms of this step (quadrotor_single.py:68-85)
        infos = [{"rewards": {k: float(ri[j, i]) for j, k in enumerate(self._keys)}} for i in range(self.num_agents)]
        for i, d in enumerate(ep_infos):
            infos[i].update(d)
        done = term.cpu().numpy().copy()
        return obs["obs"].double().cpu().numpy(), [float(r) for r in rew.cpu().numpy()], done, np.zeros_like(done, dtype=bool), infos

    def close(self):
        self._b.close()


def make_quadrotor_env_multi(cfg, render_mode=None, **kwargs):
    """quad_utils.py:20-110 for one environment: a batch of one behind the reference's list / numpy protocol"""
    reward_shaping, annealing = _shaping_from_cfg(cfg)
    return SingleQuadSwarm(BatchedQuadSwarm(1, reward_shaping_scheme=reward_shaping, annealing=annealing, write_rew_info=True, **_env_kwargs_from_cfg(cfg)))


def make_quadrotor_env_batched(cfg, **kwargs):
    """`--quads_num_envs E` > 1: the device-resident batched env (E*N agents) with the same flags and shaping schedule."""
    reward_shaping, annealing = _shaping_from_cfg(cfg)
    return BatchedQuadSwarm(
        cfg.quads_num_envs, reward_shaping_scheme=reward_shaping, annealing=annealing,
        num_gpus=getattr(cfg, "quads_num_gpus", 1), gather_obs=getattr(cfg, "quads_gather_obs", False), obs_wire=getattr(cfg, "quads_obs_wire", "bf16"), obs_transport=getattr(cfg, "quads_obs_transport", "rccl"),
        **_env_kwargs_from_cfg(cfg))


def make_quadrotor_env(env_name, cfg=None, _env_config=None, render_mode=None, **kwargs):
    if env_name == "quadrotor_multi":
        backend = getattr(cfg, "quads_backend", "hip")
        if backend != "hip":
            raise NotImplementedError(f"--quads_backend={backend!r}: this package only has the MI355X HIP stepper (no CPU simulator to fall back to)")
        if getattr(cfg, "quads_num_gpus", 1) > 1 and getattr(cfg, "quads_num_envs", 1) <= 1:
            raise ValueError("--quads_num_gpus > 1 shards the batched env: set --quads_num_envs to the global number of environments")
        if getattr(cfg, "quads_num_envs", 1) > 1:
            return make_quadrotor_env_batched(cfg, **kwargs)
        return make_quadrotor_env_multi(cfg, render_mode, **kwargs)
    raise NotImplementedError


def register_swarm_components():
    """swarm_rl/train.py:16-19: the env factory and the encoder factory.  Needs sample_factory (not part of this repo)."""
    from sample_factory.envs.env_utils import register_env   # raises ImportError where SF is not installed
    from .sf_models import register_models
    register_env("quadrotor_multi", make_quadrotor_env)
    register_models()


def parse_swarm_cfg(argv=None, evaluation=False):
    """swarm_rl/train.py:22-27"""
    from sample_factory.cfg.arguments import parse_full_cfg, parse_sf_args
    parser, partial_cfg = parse_sf_args(argv=argv, evaluation=evaluation)
    add_quadrotors_env_args(partial_cfg.env, parser)
    quadrotors_override_defaults(partial_cfg.env, parser)
    return parse_full_cfg(parser, argv)
