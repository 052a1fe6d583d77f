"""Host-side mirror of the reference's env interface on top of the HIP stepper.

  * `QuadSwarmVecEnv`   - E envs x N drones, device tensors in / out (the production, batched view);
  * `QuadrotorEnvMulti` - drop-in for gym_art/quadrotor_multi/quadrotor_multi.py:23 (one env, python lists /
                          numpy in and out, same constructor keywords, same `infos` keys, auto-reset in step).

Both raise if the HIP extension or a GPU is missing: there is no CPU fallback on the product path.
"""
import numpy as np

from . import config as qcfg
from . import native

try:  # gymnasium is optional on the box that only steps envs
    from gymnasium import spaces as _spaces
    _Box = _spaces.Box
except Exception:  # pragma: no cover - minimal stand-in with the attributes SF / the wrappers read
    class _Box:
        def __init__(self, low, high, shape=None, dtype=np.float32):
            self.low = np.asarray(low, dtype=dtype)
            self.high = np.asarray(high, dtype=dtype)
            self.shape = self.low.shape
            self.dtype = np.dtype(dtype)

        def sample(self):
            return np.random.uniform(self.low, self.high).astype(self.dtype)


class _Scenario:
    """Just enough of scenarios/base.py for the wrappers (`scenario.name()`, `.approch_goal_metric`)."""

    def __init__(self, cfg, stepper=None):
        self._cfg, self._stepper = cfg, stepper
        self.approch_goal_metric = cfg.approach_goal_metric

    def name(self, finished_episode=False):
        """Class name of the active scenario; under `mix` that is the sub-scenario of env 0's current episode
        (scenarios/mix.py:67-71), or of its last finished episode for the episode statistics."""
        sid = self._cfg.scenario
        if sid == qcfg.SCENARIOS["mix"] and self._stepper is not None:
            sid = int(self._stepper.to_host("ep_scenario" if finished_episode else "scenario_id")[0])
        return qcfg.SCENARIO_CLASS_NAMES[sid]


class _SingleView:
    """`env.envs[0]` of the reference, as far as the wrappers look at it: `.tick`, `.control_freq`."""

    def __init__(self, multi):
        self._multi = multi

    @property
    def tick(self):
        return int(self._multi._vec.stepper.to_host("tick")[0])

    @property
    def control_freq(self):
        return self._multi.control_freq


class _RewCoeff(dict):
    """env.rew_coeff: the reward-shaping wrapper assigns into it from outside (swarm_rl/env_wrappers/reward_shaping.py:57-59,111-118);
    every write raises `dirty`, so that a step pushes the coefficients to the device only when one changed - no per-step comparison."""
    dirty = True

    def __setitem__(self, k, v):
        dict.__setitem__(self, k, v)
        self.dirty = True

    def update(self, *a, **k):
        dict.update(self, *a, **k)
        self.dirty = True

    def setdefault(self, k, default=None):
        self.dirty = True
        return dict.setdefault(self, k, default)

    def pop(self, *a):
        self.dirty = True
        return dict.pop(self, *a)

    def __delitem__(self, k):
        dict.__delitem__(self, k)
        self.dirty = True

    def __deepcopy__(self, memo):
        return _RewCoeff(self)


class QuadSwarmVecEnv:
    """Batched env: all E*N agents stepped by one kernel launch; observations are born in HBM.

    reset() -> obs[E*N, D];  step(actions[E*N, 4]) -> obs, rewards[E*N], dones[E*N] (uint8), infos (lazy)
    `actions` is a torch tensor on the stepper's device with the stepper's dtype (float32 by default).
    """

    def __init__(self, num_envs, device=0, seed=0, env_id_offset=0, precision="f32", **env_kwargs):
        self.cfg = qcfg.make_config(num_envs=num_envs, seed=seed, env_id_offset=env_id_offset, precision=precision, **env_kwargs)
        self.stepper = native.Stepper(self.cfg, device=device)
        self.num_envs = num_envs
        self.num_agents_per_env = self.cfg.num_agents
        self.num_agents = num_envs * self.cfg.num_agents
        self.is_multiagent = True
        low, high = qcfg.obs_bounds(self.cfg)
        self.observation_space = _Box(low, high, dtype=np.float32)
        self.action_space = _Box(-np.ones(4), np.ones(4), dtype=np.float32)   # quadrotor_control.py:37-49
        self._rew_coeff = _RewCoeff(qcfg.REW_COEFF_DEFAULT)
        self._rew_coeff.update({k: self.cfg.rew_coeff[i] for i, k in enumerate(qcfg.REW_COEFF_KEYS)})
        self._pushed_coeff = [self._rew_coeff[k] for k in qcfg.REW_COEFF_KEYS]
        self._rew_coeff.dirty = False
        self.scenario = _Scenario(self.cfg, self.stepper)
        self._t = self.stepper.tensor
        # the per-step call path, resolved once: the library entry point, the handle, the stream getter, the output views
        self._qs_step, self._h = native.lib().qs_step, self.stepper._h
        self._n_act, self._act_bytes = self.num_agents * 4, self.stepper.real_size
        self._views = None
        self.exchange = None   # parallel.ObsExchange when this env is one shard of a multi-GPU batch whose rows are exchanged

    @property
    def rew_coeff(self):
        return self._rew_coeff

    @rew_coeff.setter
    def rew_coeff(self, value):
        """`env.rew_coeff = {...}` from outside: re-wrapped (the dirty flag lives on the wrapper) and pushed by the next step"""
        self._rew_coeff = value if isinstance(value, _RewCoeff) else _RewCoeff(value)
        self._rew_coeff.dirty = True

    def attach_exchange(self, exchange):
        """From now on every reset / step writes its observation rows into the exchange's staging buffers and sends them to the other
        shards (parallel.ObsExchange); reset() / step() return this shard's float32 rows, `exchange.latest()` the rows of all shards."""
        self.exchange = exchange

    def _sync_rew_coeff(self):
        cur = [float(self.rew_coeff[k]) for k in qcfg.REW_COEFF_KEYS]
        if cur != self._pushed_coeff:   # the SF wrapper mutates env.rew_coeff in place (reward_shaping.py:57-59)
            self.stepper.set_reward_coeffs(cur)
            self._pushed_coeff = cur
        self.rew_coeff.dirty = False

    def reset(self, env_mask=None):
        import torch
        if self.exchange is not None:
            if env_mask is not None:
                raise ValueError("masked resets are not available on a shard whose observations are exchanged")
            self.exchange.reset()
            return self.exchange.local_rows()
        self.stepper.reset(env_mask, stream=torch.cuda.current_stream(self.stepper.device))
        return self._t("obs")

    def step(self, actions):
        """One control step.  The returned tensors are VIEWS of the library's buffers (no copy, no allocation): the next step()
        overwrites them - a caller that keeps one across steps copies it."""
        if self.rew_coeff.dirty:
            self._sync_rew_coeff()
        if not (actions.is_cuda and actions.is_contiguous() and actions.numel() == self._n_act and actions.element_size() == self._act_bytes):
            raise ValueError("actions: contiguous device tensor [E*N, 4] of the stepper's precision")
        if self._views is None:
            import torch
            self._views = (self._t("obs"), self._t("reward"), self._t("done"))
            # the raw stream handle of the CURRENT stream, resolved per call (a caller may switch streams) without building a torch.cuda.Stream
            # object for it: ~0.3 us instead of ~3 us of the ~20 us a step() call cost the interpreter in round 4 (tools/bench_batched_env.py)
            raw = getattr(torch._C, "_cuda_getCurrentRawStream", None)
            self._raw_stream = raw if raw is not None else (lambda dev: torch.cuda.current_stream(dev).cuda_stream)
            self._dev = self.stepper.device
        if self.exchange is not None:
            self.exchange.step(actions.data_ptr())
            return self.exchange.local_rows(), self._views[1], self._views[2], None
        rc = self._qs_step(self._h, actions.data_ptr(), self._raw_stream(self._dev))
        if rc:
            native._check(rc)
        v = self._views
        return v[0], v[1], v[2], None

    def reward_info(self):
        """[17, E*N] device tensor of the `infos[i]['rewards']` terms (row order: config.REW_INFO_KEYS)."""
        return self._t("rew_info")

    def episode_sums(self):
        """[25, E*N] device tensor, valid after a done: per-agent sums over the finished episode of the 17 reward terms
        (rows 0-16, config.REW_INFO_KEYS order), of the 4 raw actions (17-20) and of their squares (21-24).  Needs
        `episode_sums=True`; the device-side counterpart of the reward-shaping wrapper's bookkeeping
        (swarm_rl/env_wrappers/reward_shaping.py:78-110)."""
        if not self.cfg.episode_sums:
            raise RuntimeError("create the env with episode_sums=True")
        return self._t("ep_sums")

    def close(self):
        if self.exchange is not None:
            self.exchange.close()
            self.exchange = None
        self.stepper.close()


def assemble_episode_extra_stats(eps, cnt, name, n, use_obstacles):
    """infos[i]['episode_extra_stats'] of quadrotor_multi.py:637-718 from the episode snapshot the step kernel leaves behind:
    eps[6, n] = per-agent distance_to_goal_{1,3,5}s, reached-goal / no-drone-collision / no-obstacle-collision flags;
    cnt[11] = episode counters (config.COUNTER_KEYS order); name = scenario name without its `Scenario_` prefix."""
    ok = np.logical_and(eps[4], eps[5])
    succ = float(np.sum(np.logical_and(ok, eps[3])) / n)
    dead = float(np.sum(np.logical_and(ok, 1 - eps[3])) / n)
    col, ncol, ocol = float(1.0 - np.sum(ok) / n), float(1.0 - np.sum(eps[4]) / n), float(1.0 - np.sum(eps[5]) / n)
    out = []
    for i in range(n):
        d = {
            "num_collisions": int(cnt[0]), "num_collisions_with_room": int(cnt[3]), "num_collisions_with_floor": int(cnt[4]),
            "num_collisions_with_wall": int(cnt[5]), "num_collisions_with_ceiling": int(cnt[6]),
            "num_collisions_after_settle": int(cnt[1]), f"{name}/num_collisions": int(cnt[1]),
            "num_collisions_final_5_s": int(cnt[2]), f"{name}/num_collisions_final_5_s": int(cnt[2]),
            "distance_to_goal_1s": float(eps[0, i]), "distance_to_goal_3s": float(eps[1, i]), "distance_to_goal_5s": float(eps[2, i]),
            f"{name}/distance_to_goal_1s": float(eps[0, i]), f"{name}/distance_to_goal_3s": float(eps[1, i]),
            f"{name}/distance_to_goal_5s": float(eps[2, i]),
            "metric/agent_success_rate": succ, f"{name}/agent_success_rate": succ,
            "metric/agent_deadlock_rate": dead, f"{name}/agent_deadlock_rate": dead,
            "metric/agent_col_rate": col, f"{name}/agent_col_rate": col,
            "metric/agent_neighbor_col_rate": ncol, f"{name}/agent_neighbor_col_rate": ncol,
            "metric/agent_obst_col_rate": ocol, f"{name}/agent_obst_col_rate": ocol,
        }
        if use_obstacles:
            d.update({"num_collisions_obst_quad": int(cnt[7]), "num_collisions_obst_quad_after_settle": int(cnt[8]),
                      f"{name}/num_collisions_obst": int(cnt[7]), "num_collisions_obst_quad_3_5": int(cnt[9]),
                      f"{name}/num_collisions_obst_quad_3_5": int(cnt[9]), "num_collisions_obst_quad_5": int(cnt[10]),
                      f"{name}/num_collisions_obst_quad_5": int(cnt[10])})
        out.append(d)
    return out


class QuadrotorEnvMulti:
    """Reference-compatible single environment (constructor keywords of quadrotor_multi.py:24-41)."""

    def __init__(self, num_agents, ep_time, rew_coeff, obs_repr,
                 neighbor_visible_num, neighbor_obs_type, collision_hitbox_radius, collision_falloff_radius,
                 use_obstacles, obst_density, obst_size, obst_spawn_area,
                 use_downwash, use_numba, quads_mode, room_dims, use_replay_buffer=False, quads_view_mode=None,
                 quads_render=False,
                 dynamics_params="Crazyflie", raw_control=True, raw_control_zero_middle=True,
                 dynamics_randomize_every=None, dynamics_change=None, dyn_sampler_1=None,
                 sense_noise="default", init_random_state=False, render_mode="human",
                 seed=0, device=0, precision="f32", replay_buffer_sample_prob=0.0, **domain_random_kwargs):
        if dynamics_params != "Crazyflie" or not raw_control or not raw_control_zero_middle or init_random_state \
                or dynamics_randomize_every is not None or dyn_sampler_1 is not None:
            raise NotImplementedError("only the configuration hard-coded by make_quadrotor_env_multi is supported "
                                      "(swarm_rl/env_wrappers/quad_utils.py:22-31)")
        if quads_render:
            raise NotImplementedError("rendering is out of scope of the stepper")
        tnr = 0.05
        if dynamics_change is not None:
            tnr = dynamics_change.get("noise", {}).get("thrust_noise_ratio", tnr)
        self._vec = QuadSwarmVecEnv(
            1, device=device, seed=seed, precision=precision,
            num_agents=num_agents, ep_time=ep_time, rew_coeff=rew_coeff, obs_repr=obs_repr,
            neighbor_visible_num=neighbor_visible_num, neighbor_obs_type=neighbor_obs_type,
            collision_hitbox_radius=collision_hitbox_radius, collision_falloff_radius=collision_falloff_radius,
            use_obstacles=use_obstacles, obst_density=obst_density, obst_size=obst_size, obst_spawn_area=obst_spawn_area,
            use_downwash=use_downwash, use_numba=use_numba, quads_mode=quads_mode, room_dims=room_dims,
            sense_noise=sense_noise, thrust_noise_ratio=tnr, episode_sums=bool(use_replay_buffer), **domain_random_kwargs)
        v = self._vec
        self.num_agents = num_agents
        self.is_multiagent = True
        self.observation_space, self.action_space = v.observation_space, v.action_space
        self.rew_coeff = v.rew_coeff          # same dict object: outside mutation reaches the device constants
        self.scenario = v.scenario
        self.use_obstacles, self.use_replay_buffer = use_obstacles, use_replay_buffer
        self.room_dims, self.quads_mode = room_dims, quads_mode
        self.control_freq = 1.0 / (v.cfg.dt * v.cfg.sim_steps)
        self.last_step_unique_collisions = np.array([], dtype=np.int64)
        self.curr_quad_col = np.array([], dtype=np.int64)
        self._real = v.stepper.np_real
        # Experience replay (gym_art/quadrotor_multi/quad_experience_replay.py, wired at swarm_rl/env_wrappers/quad_utils.py:67-70): the
        # wrapper's bookkeeping and the env attributes it reads (activate_replay_buffer, saved_in_replay_buffer, the crash history of
        # quadrotor_multi.py:166-175,:280-287) live on the device, per environment - qs_replay_enable, include/quadswarm.h
        self.replay_buffer_sample_prob = float(replay_buffer_sample_prob)
        if use_replay_buffer:
            v.stepper.replay_enable(self.replay_buffer_sample_prob)
        self.collisions_grace_period_seconds = 1.5
        self.obst_density, self.obst_size = obst_density, obst_size
        self.envs = [_SingleView(self)]

    @property
    def unwrapped(self):
        return self

    @property
    def activate_replay_buffer(self):
        return bool(self.use_replay_buffer and self._vec.stepper.replay_stats()["active"][0])

    @activate_replay_buffer.setter
    def activate_replay_buffer(self, value):
        self._vec.stepper.replay_set_active([1 if value else 0])

    def reset(self, obst_density=None, obst_size=None):
        """(obst_density / obst_size: the reference's replay wrapper passes its per-episode draw here; on the stepper the draw is
        made by the reset itself - --quads_domain_random, include/quadswarm.h - so explicit values are not accepted)"""
        if obst_density is not None or obst_size is not None:
            raise NotImplementedError("pass --quads_domain_random / the domain_random keywords instead: the reset draws density and size itself")
        self._vec.stepper.reset()
        return self._vec.stepper.to_host("obs").astype(np.float64)

    def _read_masks(self, st):
        ids = int(st.to_host("unique_col_mask")[0])
        self.last_step_unique_collisions = np.array([i for i in range(self.num_agents) if ids >> i & 1], dtype=np.int64)
        if self.use_obstacles:
            ids = int(st.to_host("obst_new_mask")[0])
            self.curr_quad_col = np.array([i for i in range(self.num_agents) if ids >> i & 1], dtype=np.int64)

    def step(self, actions):
        st = self._vec.stepper
        self._vec._sync_rew_coeff()
        a = np.ascontiguousarray(np.asarray(actions, dtype=self._real).reshape(self.num_agents, 4))
        st.from_host("actions", a)
        st.step()
        st.sync()
        st.check_errors()          # ValueError('QuadEnv: reward is Nan'), quadrotor_single.py:87-90
        obs = st.to_host("obs").astype(np.float64)
        rewards = [float(r) for r in st.to_host("reward")]
        dones = [bool(d) for d in st.to_host("done")]
        ri = st.to_host("rew_info")
        keys = qcfg.REW_INFO_KEYS if self.use_obstacles else qcfg.REW_INFO_KEYS[:15]
        infos = [{"rewards": {k: float(ri[j, i]) for j, k in enumerate(keys)}} for i in range(self.num_agents)]
        self._read_masks(st)
        if any(dones):
            rs = st.replay_stats() if self.use_replay_buffer else None
            if rs is not None and rs["ep_was_replay"][0]:   # quadrotor_multi.py:629-633: a replayed episode reports these two only
                cnt = st.to_host("ep_counters")[:, 0]
                stats = [{"num_collisions_replay": int(cnt[0]), "num_collisions_obst_replay": int(cnt[7])} for _ in range(self.num_agents)]
            else:
                stats = self.episode_extra_stats()
            if rs is not None:   # what ExperienceReplayWrapper.step adds at an episode end (quad_experience_replay.py:126-138)
                ep, rp, n = int(rs["episodes"][0]), int(rs["replayed"][0]), int(rs["buffer_len"][0])
                extra = {"replay/replay_rate": rp / ep, "replay/new_episode_rate": (ep - rp) / ep, "replay/replay_buffer_size": n,
                         "replay/avg_replayed": (int(rs["replayed_sum"][0]) / n) if n else 0,
                         "replay/obst_density": float(st.to_host("obst_density_env")[0]), "replay/obst_size": float(st.to_host("obst_size_env")[0])}
                for d in stats:
                    d.update(extra)
            for i in range(self.num_agents):
                infos[i]["episode_extra_stats"] = stats[i]
        return obs, rewards, dones, infos

    def episode_extra_stats(self):
        """Per-agent dicts with the keys of quadrotor_multi.py:637-718, from the device-side episode snapshot."""
        st = self._vec.stepper
        eps, cnt = st.to_host("ep_stats").astype(np.float64), st.to_host("ep_counters")[:, 0]
        return assemble_episode_extra_stats(eps, cnt, self.scenario.name(finished_episode=True)[9:], self.num_agents, self.use_obstacles)

    def render(self, *a, **k):
        raise NotImplementedError("rendering is out of scope of the stepper")

    def close(self):
        self._vec.close()
