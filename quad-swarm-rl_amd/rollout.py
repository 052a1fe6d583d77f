"""A rollout segment - T control steps of [policy encoder -> action head -> environment step] - captured once into a HIP graph.

The stepper's launch (`qs_step`: one hipModuleLaunchKernel, its per-environment step counters live in device memory) and the fused
encoder's launches are stream-ordered and allocation-free, so the whole closed loop can be recorded on a capturing stream and
replayed with one host call per segment: no Python, no per-launch host cost (about 4 us per launch on this platform) inside the
segment.  This is the device-resident counterpart of a Sample Factory rollout worker's inner loop (the reference steps its
environments from `swarm_rl/train.py` through SF's sampler); it needs no part of SF.

    seg = GraphedRollout(env, encoder, head, steps=32)
    out = seg.run()          # dict of device tensors: obs[T, A, D], actions[T, A, 4], means[T, A, 4], rewards[T, A], dones[T, A], last_obs[A, D]
"""
from . import native


class GaussianActionHead:
    """mean = Linear(512, 4)(features), action = mean + exp(log_std) * N(0, 1)  (SF's continuous action parameterisation with a
    state-independent std, `--adaptive_stddev=False` in the reference's runs); `sample=False` returns the mean."""

    def __init__(self, in_features=512, device=0, seed=0, sample=True):
        import torch
        g = torch.Generator().manual_seed(seed)
        dev = torch.device("cuda", device)
        self.weight = (torch.randn((4, in_features), generator=g) * (1.0 / in_features ** 0.5)).to(dev)
        self.bias = torch.zeros(4, device=dev)
        self.log_std = torch.full((4,), -0.5, device=dev)
        self.sample = sample
        self.seed = seed

    def __call__(self, features):
        import torch
        return self.from_mean(torch.addmm(self.bias, features, self.weight.t()))

    def from_mean(self, mean):
        """action from the Linear's output; GraphedRollout gets that from the encoder's fused epilogue (FusedQuadEncoder.forward_head)"""
        import torch
        if not self.sample:
            return mean
        return mean + torch.exp(self.log_std) * torch.randn_like(mean)


class GraphedRollout:
    def __init__(self, env, encoder, head, steps, graph=True):
        """env: QuadSwarmVecEnv (float32), encoder: policy.FusedQuadEncoder, head: features[A, 512] -> actions[A, 4].
        Construction runs ONE control step eagerly (library warm-up outside the capture) before recording.

        With the library's GaussianActionHead the segment has no glue launches at all: the encoder's epilogue evaluates the head AND
        samples the action into actions[t] (qs_enc_params.sample_*); the step writes its observation rows straight into obs[t + 1]
        (qs_set_obs_target); its rewards / done flags are copied into rewards[t] / dones[t] by the first kernel of the NEXT step's forward pass
        (qs_enc_params.traj_*; round 5 - a launch of its own, qs_rollout_post, until round 4 and still for the segment's last step).
        2 (attention: 3) dependent graph nodes per control step, and no observation copy.  Measured on C2, us per control step
        (profiles/r05h_bench_rollout.txt): mean_embed 31.5, attention 82.5; with a copy launch of its own per step 33.0 (round 4); the copy on a
        second stream as a parallel graph branch 46.3 - a fork / join costs more than the launch it hides; rewards / done redirected inside the
        step kernel 31.8, but + 0.08 us on every step of every user of the headline kernel - not taken.
        `means[t]` keeps the action head's output of every step - the mean of the Gaussian the action was drawn from - so that a learner has the
        behaviour policy's log-probabilities without a second forward pass (tools/ppo_c5.py)."""
        import torch
        if not torch.cuda.is_available():
            raise native.QsError("GraphedRollout needs a GPU")
        if env.stepper.real_size != 4:
            raise ValueError("the policy path is float32")
        self.env, self.encoder, self.head, self.steps = env, encoder, head, steps
        st = env.stepper
        self._obs, self._rew, self._done = st.tensor("obs"), st.tensor("reward"), st.tensor("done")
        A, D = self._obs.shape
        dev = self._obs.device
        self.obs = torch.empty((steps, A, D), device=dev)
        self.actions = torch.empty((steps, A, 4), device=dev)
        self.rewards = torch.empty((steps, A), device=dev)
        self.dones = torch.empty((steps, A), device=dev, dtype=torch.uint8)
        self._fused_head = hasattr(head, "from_mean") and hasattr(head, "weight") and hasattr(encoder, "set_head")
        self._glue = False
        if self._fused_head:   # the Linear runs in the encoder's epilogue: the [A, 512] features are never written
            encoder.set_head(head.weight, head.bias)
            self.means = torch.empty((steps, A, 4), device=dev)   # the action head's output per step
            # ... and with the library's head so do sampling and the trajectory writes (no copy / sampling kernels between the steps)
            self._glue = isinstance(head, GaussianActionHead) and self.dones.dtype == torch.uint8 and self._done.dtype == torch.uint8
            # (the device-side replay wrapper restores observations into the library's buffer and reads its done flags: a handle with
            # replay enabled keeps its outputs there, and the segment copies them - the two glue launches of qs_rollout_pre / _post)
            self._in_place = self._glue and not getattr(st, "replay_on", False)
            if self._glue:
                self._counter = torch.zeros(1, device=dev, dtype=torch.int32)
                self._scratch_counter = torch.zeros(1, device=dev, dtype=torch.int32)   # (qs_rollout_post's own counter: unused here)
                self._seed = int(getattr(head, "seed", 0)) & 0xffffffffffffffff
        else:
            self._feat = torch.empty((A, encoder.out_dim), device=dev)
        self.graph = None
        try:
            if graph:
                side = torch.cuda.Stream(device=dev)
                side.wait_stream(torch.cuda.current_stream(dev))
                with torch.cuda.stream(side):
                    self._segment(1)     # one step, eagerly: library warm-up outside the capture
                torch.cuda.current_stream(dev).wait_stream(side)
                self.recapture()
        finally:
            self._restore_targets()

    def recapture(self):
        """Record the segment again.  Needed when something that travels in LAUNCH ARGUMENTS changed - the attention encoder's score bias after
        FusedQuadEncoder.refresh(), the head's sampling switch; weights, log-std, reward coefficients live in device memory and are picked up
        by replays of the old graph.  Capturing runs nothing: the environments stay where they are."""
        import torch
        dev = self._obs.device
        torch.cuda.synchronize(dev)
        try:
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph):
                self._segment(self.steps)
        finally:
            self._restore_targets()

    def warmup(self):
        """one control step outside any capture (what the constructor does before it records a graph)"""
        try:
            self._segment(1)
        finally:
            self._restore_targets()

    def _restore_targets(self):
        """the stepper's output redirections are host-side launch state: an env.step() outside the segment writes the library's buffers"""
        if self._glue and self._in_place:
            self.env.stepper.set_obs_target(None)

    def _segment(self, n):
        """n control steps from the environments' current state"""
        if self._glue and not self._in_place:
            for t in range(n):
                self._step_copying(t)
            return
        if self._glue:
            import ctypes as C
            import torch
            from . import policy
            st, last, A = self.env.stepper, n - 1, self._obs.shape[0]
            main = self._torch_stream()
            self.obs[0].copy_(self._obs)      # the one observation copy of the segment: the rows the environments are in
            for t in range(n):
                # the reward / done flags of step t - 1 ride on this forward pass's first kernel (qs_enc_params.traj_*): two dependent launches per
                # control step (three with `attention`) instead of three (four); only the LAST step of the segment needs a copy launch of its own
                traj = (self._rew, self.rewards[t - 1], self._done, self.dones[t - 1]) if t > 0 else None
                if self.head.sample:
                    self.encoder.forward_head(self.obs[t], head_out=self.means[t], sample=(self.head.log_std, self.actions[t], self._counter, t, self._seed), traj=traj)
                else:   # deterministic policy: the head's output IS the action
                    self.encoder.forward_head(self.obs[t], head_out=self.actions[t], traj=traj)
                # step t: observation rows -> obs[t + 1] (the last step: the library's buffer, where the next segment starts)
                st.set_obs_target(self.obs[t + 1].data_ptr() if t < last else None)
                st.step(self.actions[t].data_ptr(), stream=main)
            rc = policy.lib().qs_rollout_post(C.c_void_p(self._rew.data_ptr()), C.c_void_p(self.rewards[last].data_ptr()), C.c_void_p(self._done.data_ptr()),
                                              C.c_void_p(self.dones[last].data_ptr()), A, C.c_void_p(self._scratch_counter.data_ptr()), C.c_void_p(main.cuda_stream))
            if rc != 0:
                raise native.QsError(f"qs_rollout_post failed ({rc})")
            self._counter.add_(n)             # the next replay draws fresh noise
            return
        for t in range(n):
            self._step(t)

    def _torch_stream(self):
        import torch
        return torch.cuda.current_stream(self._obs.device)

    def _step_copying(self, t):
        """the segment step of a handle whose outputs stay in the library's buffers: sampling + observation copy in one launch before
        the step, reward / done copies + counter in one launch behind it"""
        import ctypes as C
        from . import policy
        L = policy.lib()
        stream = C.c_void_p(self._torch_stream().cuda_stream)
        A = self._obs.shape[0]
        self.encoder.forward_head(self._obs, head_out=self.means[t])
        log_std = C.c_void_p(self.head.log_std.data_ptr()) if self.head.sample else None
        rc = L.qs_rollout_pre(C.c_void_p(self._obs.data_ptr()), C.c_void_p(self.obs[t].data_ptr()), self._obs.numel(), C.c_void_p(self.means[t].data_ptr()), log_std,
                              C.c_void_p(self.actions[t].data_ptr()), A, C.c_uint64(self._seed), C.c_void_p(self._counter.data_ptr()), stream)
        if rc != 0:
            raise native.QsError(f"qs_rollout_pre failed ({rc})")
        self.env.stepper.step(self.actions[t].data_ptr(), stream=self._torch_stream())
        rc = L.qs_rollout_post(C.c_void_p(self._rew.data_ptr()), C.c_void_p(self.rewards[t].data_ptr()), C.c_void_p(self._done.data_ptr()),
                               C.c_void_p(self.dones[t].data_ptr()), A, C.c_void_p(self._counter.data_ptr()), stream)
        if rc != 0:
            raise native.QsError(f"qs_rollout_post failed ({rc})")

    def _step(self, t):
        self.obs[t].copy_(self._obs)
        if self._fused_head:
            self.encoder.forward_head(self._obs, head_out=self.means[t])
            self.actions[t].copy_(self.head.from_mean(self.means[t]))
        else:
            self.encoder(self._obs, out=self._feat)
            self.actions[t].copy_(self.head(self._feat))
        self.env.stepper.step(self.actions[t].data_ptr(), stream=self._torch_stream())
        self.rewards[t].copy_(self._rew)
        self.dones[t].copy_(self._done)

    def run(self):
        """One segment of `steps` control steps from the environments' current state (stream-ordered, asynchronous).  The returned
        tensors are the segment's buffers: the next run() overwrites them."""
        if self.graph is not None:
            self.graph.replay()
        else:
            try:
                self._segment(self.steps)
            finally:
                self._restore_targets()
        out = {"obs": self.obs, "actions": self.actions, "rewards": self.rewards, "dones": self.dones, "last_obs": self._obs}
        if self._fused_head:
            out["means"] = self.means if getattr(self.head, "sample", True) else self.actions
        return out
