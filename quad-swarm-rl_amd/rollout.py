"""A rollout segment - T control steps of [policy encoder -> action head -> environment step] - captured once into a HIP graph.

The stepper's launch (`qs_step`: one hipModuleLaunchKernel, its per-environment step counters live in device memory) and the fused
encoder's launches are stream-ordered and allocation-free, so the whole closed loop can be recorded on a capturing stream and
replayed with one host call per segment: no Python, no per-launch host cost (about 4 us per launch on this platform) inside the
segment.  This is the device-resident counterpart of a Sample Factory rollout worker's inner loop (the reference steps its
environments from `swarm_rl/train.py` through SF's sampler); it needs no part of SF.

    seg = GraphedRollout(env, encoder, head, steps=32)
    out = seg.run()          # dict of device tensors: obs[T, A, D], actions[T, A, 4], rewards[T, A], dones[T, A], last_obs[A, D]
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
        Construction runs ONE control step eagerly (library warm-up outside the capture) before recording."""
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
            self._mean = torch.empty((A, 4), device=dev)
            # ... and sampling + the trajectory copies are two launches of the library's glue kernels instead of eight torch kernels
            self._glue = isinstance(head, GaussianActionHead) and self.dones.dtype == torch.uint8 and self._done.dtype == torch.uint8
            if self._glue:
                self._counter = torch.zeros(1, device=dev, dtype=torch.int32)
                self._seed = int(getattr(head, "seed", 0)) & 0xffffffffffffffff
        else:
            self._feat = torch.empty((A, encoder.out_dim), device=dev)
        self.graph = None
        if graph:
            side = torch.cuda.Stream(device=dev)
            side.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(side):
                self._step(0)
            torch.cuda.current_stream(dev).wait_stream(side)
            torch.cuda.synchronize(dev)
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph):
                for t in range(steps):
                    self._step(t)

    def _step(self, t):
        import torch
        if self._glue:
            import ctypes as C
            from . import policy
            L = policy.lib()
            stream = C.c_void_p(torch.cuda.current_stream(self._obs.device).cuda_stream)
            A = self._obs.shape[0]
            self.encoder.forward_head(self._obs, head_out=self._mean)
            log_std = C.c_void_p(self.head.log_std.data_ptr()) if self.head.sample else None
            rc = L.qs_rollout_pre(C.c_void_p(self._obs.data_ptr()), C.c_void_p(self.obs[t].data_ptr()), self._obs.numel(), C.c_void_p(self._mean.data_ptr()), log_std,
                                  C.c_void_p(self.actions[t].data_ptr()), A, C.c_uint64(self._seed), C.c_void_p(self._counter.data_ptr()), stream)
            if rc != 0:
                raise native.QsError(f"qs_rollout_pre failed ({rc})")
            self.env.stepper.step(self.actions[t].data_ptr(), stream=torch.cuda.current_stream(self._obs.device))
            rc = L.qs_rollout_post(C.c_void_p(self._rew.data_ptr()), C.c_void_p(self.rewards[t].data_ptr()), C.c_void_p(self._done.data_ptr()),
                                   C.c_void_p(self.dones[t].data_ptr()), A, C.c_void_p(self._counter.data_ptr()), stream)
            if rc != 0:
                raise native.QsError(f"qs_rollout_post failed ({rc})")
            return
        self.obs[t].copy_(self._obs)
        if self._fused_head:
            self.encoder.forward_head(self._obs, head_out=self._mean)
            self.actions[t].copy_(self.head.from_mean(self._mean))
        else:
            self.encoder(self._obs, out=self._feat)
            self.actions[t].copy_(self.head(self._feat))
        self.env.stepper.step(self.actions[t].data_ptr(), stream=torch.cuda.current_stream(self._obs.device))
        self.rewards[t].copy_(self._rew)
        self.dones[t].copy_(self._done)

    def run(self):
        """One segment of `steps` control steps from the environments' current state (stream-ordered, asynchronous)."""
        if self.graph is not None:
            self.graph.replay()
        else:
            for t in range(self.steps):
                self._step(t)
        return {"obs": self.obs, "actions": self.actions, "rewards": self.rewards, "dones": self.dones, "last_obs": self._obs}
