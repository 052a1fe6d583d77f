"""A scripted multi-agent env for wrapper tests: the outputs of step() are a pure function of (seed, step index), so the reference's
wrappers (run once in the build container, oracle/ref_harness/capture_wrappers.py) and this repo's wrappers see the same stream."""
import numpy as np

REW_KEYS = ("rew_main", "rew_pos", "rew_action", "rew_crash", "rew_orient", "rew_spin", "rewraw_main", "rewraw_pos", "rewraw_action",
            "rewraw_crash", "rewraw_orient", "rewraw_spin", "rew_quadcol", "rew_proximity", "rewraw_quadcol")


class _Scenario:
    def __init__(self, names):
        self._names, self.i = names, 0

    def name(self):
        return self._names[self.i % len(self._names)]


class FakeQuadEnv:
    is_multiagent = True

    def __init__(self, num_agents=3, ep_len=7, seed=0):
        self.num_agents, self.ep_len, self.seed = num_agents, ep_len, seed
        self.rew_coeff = dict(pos=1.0, effort=0.05, spin=0.1, vel=0.0, crash=1.0, orient=1.0, yaw=0.0, quadcol_bin=0.0,
                              quadcol_bin_smooth_max=0.0, quadcol_bin_obst=0.0)
        self.scenario = _Scenario(["static_same_goal", "swarm_vs_swarm", "dynamic_formations"])
        self.t = 0
        self.coeff_log = []

    @property
    def unwrapped(self):
        return self

    def _obs(self):
        rng = np.random.RandomState(self.seed * 100003 + self.t)
        return [rng.uniform(-1, 1, 18) for _ in range(self.num_agents)]

    def reset(self):
        return self._obs()

    def step(self, action):
        rng = np.random.RandomState(self.seed * 7919 + 13 * self.t + 1)
        self.t += 1
        done = self.t % self.ep_len == 0
        infos = []
        for _ in range(self.num_agents):
            vals = rng.uniform(-0.2, 0.05, len(REW_KEYS))
            rew = dict(zip(REW_KEYS, (float(v) for v in vals)))
            rew["rewraw_quadcol"] = -float(rng.randint(0, 2))         # collisions are counted in whole units
            rew["not_a_reward_key"] = 123.0                            # only keys starting with 'rew' are accumulated
            infos.append({"rewards": rew})
        if done:
            infos[0]["episode_extra_stats"] = {"num_collisions": int(rng.randint(0, 5))}   # the env's own stats stay in place
            self.scenario.i += 1
        self.coeff_log.append(dict(self.rew_coeff))                    # what the env would have used for this step
        rewards = [info["rewards"]["rew_main"] for info in infos]
        return self._obs(), rewards, [done] * self.num_agents, infos


class FakeVec:
    """What sf_env.BatchedQuadSwarm needs from env.QuadSwarmVecEnv, over ONE FakeQuadEnv and with CPU torch tensors where the real one has
    device tensors: step() keeps the per-episode sums of the reward terms and the action moments the way the step kernel does
    (`episode_sums`), the "stepper" hands them out.  Lets the CPU suite drive the host side of the batched env - shaping scheme, annealing,
    episode-end infos - over the script the reference wrapper was recorded on."""

    def __init__(self, env):
        import types
        import torch
        from quad_swarm_rl_amd import config as qcfg
        self.env, self._torch = env, torch
        n = env.num_agents
        self.num_agents_per_env = self.num_agents = n
        self.observation_space = self.action_space = None
        self.rew_coeff, self.scenario = env.rew_coeff, env.scenario
        self.cfg = types.SimpleNamespace(ep_len=env.ep_len - 1, use_obstacles=0)
        self.exchange = None
        self._scen_id = lambda: qcfg.SCENARIOS[env.scenario.name()]
        self.run = np.zeros((25, n))
        self.bufs = {"ep_sums": torch.zeros((25, n), dtype=torch.float64), "ep_stats": torch.zeros((6, n), dtype=torch.float64),
                     "ep_counters": torch.zeros((11, 1), dtype=torch.int32), "ep_scenario": torch.zeros(1, dtype=torch.int32),
                     "scenario_id": torch.zeros(1, dtype=torch.int32), "done": torch.zeros(n, dtype=torch.uint8), "tick": torch.zeros(1, dtype=torch.int32)}
        vec = self

        class Stepper:
            device = "cpu"

            @staticmethod
            def tensor(name):
                return vec.bufs[name]

            @staticmethod
            def to_host(name):
                return vec.bufs[name].numpy().copy()

        self.stepper = Stepper()

    def reset(self, env_mask=None):
        self.bufs["scenario_id"][0] = self._scen_id()
        return self._torch.as_tensor(np.stack(self.env.reset()))

    def step(self, actions):
        torch = self._torch
        a = np.asarray(actions, dtype=np.float64)
        scen_before = self._scen_id()
        obs, rewards, dones, infos = self.env.step([a[i] for i in range(self.num_agents)])
        for i, info in enumerate(infos):
            self.run[:15, i] += [info["rewards"][k] for k in REW_KEYS]
            self.run[17:21, i] += a[i]
            self.run[21:25, i] += a[i] ** 2
        self.env_infos = infos
        done = bool(dones[0])
        self.bufs["done"][:] = int(done)
        self.bufs["tick"][0] = 0 if done else self.env.t % self.env.ep_len
        self.bufs["scenario_id"][0] = self._scen_id()
        if done:
            self.bufs["ep_sums"].copy_(torch.as_tensor(self.run))
            self.bufs["ep_scenario"][0] = scen_before
            self.run[:] = 0
        return torch.as_tensor(np.stack(obs)), torch.as_tensor(np.array(rewards)), self.bufs["done"], None

    def close(self):
        pass


def drive_batched(batched, vec, env, steps=30, seed=0):
    """drive()'s script through sf_env.BatchedQuadSwarm over a FakeVec; the record keeps what the shaping wrapper adds (the env's own
    episode statistics - zeros here - are dropped, the scenario prefix of the class names is stripped: the scripted env's scenarios
    are called by their bare names)"""
    import torch
    rng = np.random.RandomState(seed + 555)
    batched.reset()
    rec = []
    for t in range(steps):
        batched.set_training_info({"approx_total_training_steps": 40000 * t})
        actions = np.stack([rng.uniform(-1, 1, 4) for _ in range(env.num_agents)])
        _, rewards, term, _, infos = batched.step(torch.as_tensor(actions))
        extra = [None] * env.num_agents
        if len(infos):
            for i in range(env.num_agents):
                ex = infos[i]["episode_extra_stats"]
                # drop what only the env's own statistics contribute (neither a reward sum, nor a z_* key, nor a per-scenario reward key)
                extra[i] = {k.replace("Scenario_", ""): float(v) for k, v in sorted(ex.items())
                            if k.startswith(("rew", "z_")) or k.endswith(("/rew_pos", "/rew_crash"))}
        rec.append({"rewards": [float(r) for r in rewards], "dones": [bool(d) for d in term],
                    "true_reward": [float(infos[i]["true_reward"]) if len(infos) else None for i in range(env.num_agents)],
                    "extra": extra, "rew_coeff": {k: float(v) for k, v in sorted(env.rew_coeff.items())}})
    return {"steps": rec, "coeff_seen_by_env": [{k: float(v) for k, v in sorted(c.items())} for c in env.coeff_log]}


def drive(wrapper, env, steps=30, seed=0):
    """The fixed script both sides run; returns a JSON-able record of everything the wrapper adds or changes."""
    rng = np.random.RandomState(seed + 555)
    wrapper.reset()
    rec = []
    for t in range(steps):
        wrapper.set_training_info({"approx_total_training_steps": 40000 * t})
        actions = [rng.uniform(-1, 1, 4) for _ in range(env.num_agents)]
        _, rewards, dones, infos = wrapper.step(actions)
        rec.append({
            "rewards": [float(r) for r in rewards], "dones": [bool(d) for d in dones],
            "true_reward": [float(i["true_reward"]) if "true_reward" in i else None for i in infos],
            "extra": [{k: float(v) for k, v in sorted(i["episode_extra_stats"].items())} if "episode_extra_stats" in i else None for i in infos],
            "rew_coeff": {k: float(v) for k, v in sorted(env.rew_coeff.items())},
        })
    return {"steps": rec, "coeff_seen_by_env": [{k: float(v) for k, v in sorted(c.items())} for c in env.coeff_log]}


# ---------------------------------------------------------------------------------------------------------------------------------
# scripted env for the experience-replay wrapper: its whole future is a function of its (copyable) state, so "deep copy the env"
# (reference) and "snapshot slot + host attributes" (this repo) must lead to the same trajectories
# ---------------------------------------------------------------------------------------------------------------------------------
class _Single:
    def __init__(self):
        self.tick, self.control_freq = 0, 100


class _FakeStepper:
    def __init__(self, env):
        self.env = env

    def snapshot_pool(self, n):
        self.env._slots = [None] * n

    def snapshot_copy(self, src, dst):
        self.env._slots[dst] = dict(self.env._slots[src])


class FakeReplayEnv:
    """collisions at pseudo-random ticks, 6-second episodes; `core()` is everything a deep copy would carry"""
    num_agents, is_multiagent, use_replay_buffer, use_obstacles = 2, True, True, False
    collisions_grace_period_seconds = 1.5
    _HOST = ("activate_replay_buffer", "saved_in_replay_buffer", "obst_density")

    def __init__(self, seed=0, ep_len=600):
        self.seed, self.ep_len = seed, ep_len
        self.envs = [_Single()]
        self.scenes = []
        self.activate_replay_buffer, self.saved_in_replay_buffer, self.collision_occurred = True, False, False
        self.obst_density, self.obst_size = 0.2, 0.6
        self.curr_quad_col = []
        self.last_step_unique_collisions = np.array([], dtype=np.int64)
        self.collisions_per_episode = self.collisions_after_settle = 0
        self.obst_quad_collisions_per_episode = self.obst_quad_collisions_after_settle = 0
        self.episode, self.x = 0, 0.0
        self._slots = []
        self._vec = type("V", (), {})()
        self._vec.stepper = _FakeStepper(self)

    @property
    def unwrapped(self):
        return self

    # --- the state a deep copy carries (besides the host attributes) ---
    def core(self):
        return dict(tick=self.envs[0].tick, x=self.x, episode=self.episode, cpe=self.collisions_per_episode, cas=self.collisions_after_settle,
                    ids=[int(i) for i in self.last_step_unique_collisions])

    def set_core(self, c):
        self.envs[0].tick, self.x, self.episode = c["tick"], c["x"], c["episode"]
        self.collisions_per_episode, self.collisions_after_settle = c["cpe"], c["cas"]
        self.last_step_unique_collisions = np.array(c["ids"], dtype=np.int64)

    def __deepcopy__(self, memo):   # what the reference's deepcopy(env) sees
        other = FakeReplayEnv(self.seed, self.ep_len)
        other.set_core(self.core())
        for k in self._HOST + ("collision_occurred",):
            setattr(other, k, getattr(self, k))
        return other

    # --- this repo's checkpoint protocol (env.QuadrotorEnvMulti.save_checkpoint / load_checkpoint / zero_collision_counters) ---
    def save_checkpoint(self, slot):
        self._slots[slot] = self.core()
        return {k: getattr(self, k) for k in self._HOST}

    def load_checkpoint(self, slot, host):
        self.set_core(self._slots[slot])
        for k in self._HOST:
            setattr(self, k, host[k])

    def zero_collision_counters(self):
        self.collisions_per_episode = self.collisions_after_settle = 0

    def obs(self):
        return [np.array([self.x, float(self.envs[0].tick), float(self.episode)]) for _ in range(self.num_agents)]

    def reset(self, obst_density=None, obst_size=None):
        self.episode += 1
        self.envs[0].tick = 0
        self.x = float((self.seed * 31 + self.episode * 17) % 89)
        self.last_step_unique_collisions = np.array([], dtype=np.int64)
        self.collisions_per_episode = self.collisions_after_settle = 0
        return self.obs()

    def step(self, action):
        s = self.envs[0]
        s.tick += 1
        self.x = (self.x * 1.0009765625 + 0.37 * s.tick) % 97.0
        hit = int(self.x * 64) % 173 == 0
        self.last_step_unique_collisions = np.array([0, 1] if hit else [], dtype=np.int64)
        if hit:
            self.collisions_per_episode += 1
            self.collisions_after_settle += int(s.tick > 150)
        done = s.tick >= self.ep_len
        infos = [{"episode_extra_stats": {"num_collisions": self.collisions_per_episode} if done else {}} for _ in range(self.num_agents)]
        rewards = [-0.01 * self.x] * self.num_agents
        if done:
            obs = self.reset()      # the real env resets itself inside step()
        else:
            obs = self.obs()
        return obs, rewards, [done] * self.num_agents, infos


def drive_replay(wrapper, steps):
    """returns, per step, what the wrapper's env looks like from outside (x, tick, episode - whatever object that env currently is)
    and the replay statistics it attaches at episode ends"""
    rec = {"x": [], "tick": [], "episode": [], "ends": {}}
    wrapper.reset()
    for t in range(steps):
        obs, rewards, dones, infos = wrapper.step([np.zeros(4)] * 2)
        rec["x"].append(float(obs[0][0])); rec["tick"].append(int(obs[0][1])); rec["episode"].append(int(obs[0][2]))
        if dones[0]:
            rec["ends"][str(t)] = {k: float(v) for k, v in sorted(infos[0]["episode_extra_stats"].items())}
    return rec
