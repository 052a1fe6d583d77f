"""ExperienceReplayWrapper of the reference (gym_art/quadrotor_multi/quad_experience_replay.py:9-209) on the HIP stepper.

Same control flow, constants and statistics; the two `deepcopy(env)` sites become device-side snapshots
(`qs_snapshot_save / load / copy`, include/quadswarm.h): a checkpoint is a slot of the stepper's snapshot pool plus the few
host-side attributes of the env.  Slots 0..5 hold the last 3 s of checkpoints of the running episode (one per 0.5 s), slots
6..25 the replay buffer (20 collision events).  Per-episode obstacle density / size randomisation (`domain_random`) is not
supported: the obstacle configuration is fixed when the stepper is created.
"""
from collections import deque

import numpy as np


class ReplayBufferEvent:
    def __init__(self, slot, host, obs):
        self.slot, self.host, self.obs = slot, host, obs
        self.num_replayed = 0


class ReplayBuffer:
    def __init__(self, control_frequency, first_slot, cp_step_size=0.5, buffer_size=20, rng=None):
        self.control_frequency = control_frequency
        self.cp_step_size_sec = cp_step_size
        self.cp_step_size_freq = self.cp_step_size_sec * self.control_frequency
        self.buffer_idx = 0
        self.maxlen = buffer_size
        self.buffer = []
        self.free_slots = list(range(first_slot, first_slot + buffer_size))
        self.rng = rng

    def write_cp_to_buffer(self, env, ring_slot, host, obs):
        """:24-38 - the checkpoint from X seconds ago becomes a buffer event (its copy of the env is marked as saved)."""
        host = dict(host, saved_in_replay_buffer=True)
        if len(self.buffer) < self.maxlen:
            evt = ReplayBufferEvent(self.free_slots.pop(0), host, obs)
            self.buffer.append(evt)
        else:
            evt = ReplayBufferEvent(self.buffer[self.buffer_idx].slot, host, obs)
            self.buffer[self.buffer_idx] = evt
        env.unwrapped._vec.stepper.snapshot_copy(ring_slot, evt.slot)
        self.buffer_idx = (self.buffer_idx + 1) % self.maxlen

    def sample_event(self):
        idx = int(self.rng.randint(0, len(self.buffer)))   # random.randint(0, len-1) of the reference
        self.buffer[idx].num_replayed += 1
        return self.buffer[idx]

    def cleanup(self):   # :50-56
        keep = []
        for event in self.buffer:
            if event.num_replayed < 10:
                keep.append(event)
            else:
                self.free_slots.append(event.slot)
        self.buffer = keep

    def avg_num_replayed(self):
        return float(np.mean([e.num_replayed for e in self.buffer])) if self.buffer else 0

    def __len__(self):
        return len(self.buffer)


class ExperienceReplayWrapper:
    def __init__(self, env, replay_buffer_sample_prob, default_obst_density, defulat_obst_size, domain_random=False, seed=0, **unused):
        if domain_random:
            raise NotImplementedError("domain_random (per-episode obstacle density / size) is not part of the stepper")
        self.env = env
        self.rng = np.random.RandomState(seed)
        self.ring_slots = int(3.0 / 0.5)
        self.replay_buffer = ReplayBuffer(env.envs[0].control_freq, first_slot=self.ring_slots, rng=self.rng)
        env.unwrapped._vec.stepper.snapshot_pool(self.ring_slots + self.replay_buffer.maxlen)
        self.replay_buffer_sample_prob = replay_buffer_sample_prob
        self.curr_obst_density, self.curr_obst_size = default_obst_density, defulat_obst_size
        self.max_episode_checkpoints_to_keep = int(3.0 / self.replay_buffer.cp_step_size_sec)   # the last 3 seconds
        self.episode_checkpoints = deque([], maxlen=self.max_episode_checkpoints_to_keep)
        self._next_ring = 0
        self.save_time_before_collision_sec = 1.5
        self.last_tick_added_to_buffer = -1e9
        self.replayed_events = 0
        self.episode_counter = 0

    def __getattr__(self, name):
        return getattr(self.env, name)

    @property
    def unwrapped(self):
        return self.env.unwrapped

    def save_checkpoint(self, obs):   # :99-104
        slot = self._next_ring
        self._next_ring = (self._next_ring + 1) % self.ring_slots
        self.episode_checkpoints.append((slot, self.env.save_checkpoint(slot), np.array(obs, copy=True)))

    def reset(self):
        return self.env.reset(None, None)

    def step(self, action):
        env = self.env
        obs, rewards, dones, infos = env.step(action)
        if any(dones):
            obs = self.new_episode()
            for i in range(len(infos)):
                if not infos[i].get("episode_extra_stats"):
                    infos[i]["episode_extra_stats"] = dict()
                tag = "replay"
                infos[i]["episode_extra_stats"].update({
                    f"{tag}/replay_rate": self.replayed_events / self.episode_counter,
                    f"{tag}/new_episode_rate": (self.episode_counter - self.replayed_events) / self.episode_counter,
                    f"{tag}/replay_buffer_size": len(self.replay_buffer),
                    f"{tag}/avg_replayed": self.replay_buffer.avg_num_replayed(),
                    f"{tag}/obst_density": self.curr_obst_density,
                    f"{tag}/obst_size": self.curr_obst_size,
                })
        else:
            tick = env.envs[0].tick
            if env.use_replay_buffer and env.activate_replay_buffer and not env.saved_in_replay_buffer \
                    and tick % self.replay_buffer.cp_step_size_freq == 0:
                self.save_checkpoint(obs)
            collision_flag = env.last_step_unique_collisions.any()
            if env.use_obstacles:
                collision_flag = collision_flag or len(env.curr_quad_col) > 0
            if collision_flag and env.use_replay_buffer and env.activate_replay_buffer \
                    and tick > env.collisions_grace_period_seconds * env.envs[0].control_freq and not env.saved_in_replay_buffer:
                if tick - self.last_tick_added_to_buffer > 5 * env.envs[0].control_freq:
                    steps_ago = int(self.save_time_before_collision_sec / self.replay_buffer.cp_step_size_sec)
                    if steps_ago > len(self.episode_checkpoints):
                        raise IndexError(f"Tried to read past the boundary of checkpoint_history. Steps ago: {steps_ago}, "
                                         f"episode checkpoints: {len(self.episode_checkpoints)}, {tick}")
                    slot, host, cp_obs = self.episode_checkpoints[-steps_ago]
                    self.replay_buffer.write_cp_to_buffer(env, slot, host, cp_obs)
                    # the reference unpacks the checkpoint into its local `obs` (`env, obs = self.episode_checkpoints[-steps_ago]`,
                    # :151) and returns that: on the step that files an event the caller gets the 1.5-s-old observation of the
                    # checkpoint, not the current one.  Reproduced (pinned by tests/test_wrappers_vs_reference.py).
                    obs = np.array(cp_obs, copy=True)
                    env.collision_occurred = False
                    self.last_tick_added_to_buffer = tick
        return obs, rewards, dones, infos

    def new_episode(self):   # :162-209
        env = self.env
        self.episode_counter += 1
        self.last_tick_added_to_buffer = -1e9
        self.episode_checkpoints = deque([], maxlen=self.max_episode_checkpoints_to_keep)
        if self.rng.uniform(0, 1) < self.replay_buffer_sample_prob and self.replay_buffer and env.activate_replay_buffer \
                and len(self.replay_buffer) > 0:
            self.replayed_events += 1
            event = self.replay_buffer.sample_event()
            env.load_checkpoint(event.slot, event.host)
            self.curr_obst_density = env.obst_density
            env.zero_collision_counters()
            self.replay_buffer.cleanup()
            return np.array(event.obs, copy=True)
        obs = env.reset(None, None)
        env.saved_in_replay_buffer = False
        return obs
