"""Multi-GPU sharding of the stepper: one process per GPU, contiguous env ranges, ONE RCCL all-gather of the
observations per rollout step (the reference itself has no collective anywhere).

Environments are fully independent (no cross-env term in quadrotor_multi.py), so rank r owns the global envs
[r*E, (r+1)*E) and steps them with `env_id_offset=r*E`; the counter-based RNG is keyed by the GLOBAL env id, so the
union of the shards is bit-identical to one un-sharded run (tests/test_hip_parity.py::test_determinism_and_sharding_invariance).
xGMI is point-to-point (7 links/GPU): the only traffic is each rank's observation shard going to its peers, no reduction.
"""
import torch
import torch.distributed as dist


def shard_range(total_envs, world_size, rank):
    """Contiguous env range [lo, hi) of `rank`; a whole env never straddles GPUs."""
    if total_envs % world_size:
        raise ValueError("total_envs must be divisible by the number of ranks")
    per = total_envs // world_size
    return rank * per, (rank + 1) * per


class ObsGather:
    """All-gather of the local observation tensor [T, D] into [world*T, D], optionally overlapped with the next steps.

    overlap=False : gather() enqueues the collective behind the step on the current stream and returns the result.
    overlap=True  : the obs are copied into one of two staging buffers and gathered asynchronously (RCCL's own
                    stream); result() of step t is waited for (stream-side) only when it is consumed or when its
                    staging buffer is needed again at step t+2, so gather(t) overlaps compute(t+1).
    """

    def __init__(self, local_obs, group=None, overlap=False):
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.local = local_obs
        self.overlap = overlap
        nbuf = 2 if overlap else 1
        shape = (self.world * local_obs.shape[0],) + tuple(local_obs.shape[1:])
        self.out = [torch.empty(shape, dtype=local_obs.dtype, device=local_obs.device) for _ in range(nbuf)]
        self.stage = [torch.empty_like(local_obs) for _ in range(nbuf)] if overlap else None
        self.work = [None] * nbuf
        self.t = 0

    def _all_gather(self, out, src, async_op):
        try:
            return dist.all_gather_into_tensor(out, src, group=self.group, async_op=async_op)
        except (RuntimeError, NotImplementedError):   # backend without the fused op: list form, same result
            chunks = list(out.chunk(self.world, dim=0))
            return dist.all_gather(chunks, src, group=self.group, async_op=async_op)

    def gather(self):
        """Call right after the step that produced `local_obs`.  Returns the gathered tensor (overlap=False) or the
        index of the in-flight buffer (overlap=True; fetch with result(idx))."""
        if not self.overlap:
            self._all_gather(self.out[0], self.local, async_op=False)
            return self.out[0]
        k = self.t & 1
        if self.work[k] is not None:
            self.work[k].wait()                 # buffer k was used at step t-2
        self.stage[k].copy_(self.local)         # snapshot: the stepper overwrites its obs buffer at the next step
        self.work[k] = self._all_gather(self.out[k], self.stage[k], async_op=True)
        self.t += 1
        return k

    def result(self, k):
        if self.work[k] is not None:
            self.work[k].wait()
            self.work[k] = None
        return self.out[k]

    def drain(self):
        for k in range(len(self.work)):
            if self.work[k] is not None:
                self.work[k].wait()
                self.work[k] = None
