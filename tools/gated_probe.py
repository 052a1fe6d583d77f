#!/usr/bin/env python
"""Resident-state stepping (qs_step_gated) swept over the producer's group size and the steps per launch: us per control step, producer
running ahead / closed loop, next to one launch per step and the open-loop multi-step launch.  usage: tools/gated_probe.py [workload]"""
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import bench  # noqa: E402
import torch  # noqa: E402
from quad_swarm_rl_amd import config as qcfg, native  # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "c2"
w = bench.WORKLOADS[wl]
E = w["num_envs"]
cfg = qcfg.make_config(num_envs=E, seed=0, precision="f32", write_rew_info=False, **w["kw"])
T = E * cfg.num_agents
g = torch.Generator(device="cuda").manual_seed(1234)
ring = 64
acts = (torch.rand((ring, T, 4), device="cuda", generator=g) * 2 - 1).contiguous()
aptr = acts.data_ptr()
side, feed = torch.cuda.Stream(), torch.cuda.Stream()


def timed(fn, steps):
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record(side)
    fn()
    ev1.record(side)
    torch.cuda.synchronize()
    return ev0.elapsed_time(ev1) * 1e3 / steps


st = native.Stepper(cfg, device=0)
st.reset()
for t in range(2000):
    st.step(aptr + (t % ring) * T * 16, stream=side)
print(f"{wl}: one launch per control step: {timed(lambda: [st.step(aptr + (t % ring) * T * 16, stream=side) for t in range(2000)], 2000):.3f} us per step")
print(f"{wl}: open-loop 64-step launches:  {timed(lambda: [st.step_many(aptr, 64, stream=side) for _ in range(30)], 1920):.3f} us per step")
st.close()
for wgpg in (1, 2, 4, 8, 16):
    for k in (16, 64, 256):
        s2 = native.Stepper(cfg, device=0)
        s2.gate_create(ring_len=ring, wg_per_group=wgpg)
        s2.reset()
        out = []
        for closed in (False, True):
            reps = max(2, 2048 // k)
            s2.step_gated(k, stream=side); s2.gate_produce(aptr, ring, k, closed, stream=feed)
            torch.cuda.synchronize()

            def run():
                for _ in range(reps):
                    s2.step_gated(k, stream=side)
                    s2.gate_produce(aptr, ring, k, closed, stream=feed)
                    s2.gate_wait(stream=side)
            out.append(timed(run, reps * k))
        stt = s2.gate_status()
        print(f"{wl}: gated, {wgpg:2d} workgroups per producer group, {k:3d} steps per launch: producer ahead {out[0]:.3f} us, closed loop {out[1]:.3f} us per step (status {stt['error']})")
        s2.close()
