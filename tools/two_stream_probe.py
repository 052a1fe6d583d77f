#!/usr/bin/env python
"""Does a batch stepped as TWO independent half-batches on two streams beat one launch?  (The dispatch ramp and the tail of one
launch would overlap with the body of the other.)  Each variant is K steps captured into one HIP graph per stream and replayed, so
the host is out of the picture.  usage: python tools/two_stream_probe.py c2|c4"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from quad_swarm_rl_amd import config as qcfg, native

wl = sys.argv[1] if len(sys.argv) > 1 else "c2"
kw = dict(bench.WORKLOADS[wl]["kw"])
E = bench.WORKLOADS[wl]["num_envs"]
K, REPS = 200, 20


def make(envs, offset):
    cfg = qcfg.make_config(num_envs=envs, seed=0, env_id_offset=offset, write_rew_info=False, **kw)
    st = native.Stepper(cfg)
    st.reset()
    return st


def graph_of(st, stream, acts):
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(stream):
        for t in range(3):
            st.step(acts.data_ptr(), stream=stream.cuda_stream)
        stream.synchronize()
        with torch.cuda.graph(g, stream=stream):
            for t in range(K):
                st.step(acts.data_ptr(), stream=stream.cuda_stream)
    return g


def run(parts):
    sts, streams, graphs = [], [], []
    off = 0
    for envs in parts:
        st = make(envs, off)
        off += envs
        s = torch.cuda.Stream()
        acts = ((torch.rand((st.T, 4), device="cuda") * 2 - 1) * 0.2 + 0.1).contiguous()
        sts.append((st, acts))
        streams.append(s)
        graphs.append(graph_of(st, s, acts))
    for g, s in zip(graphs, streams):
        with torch.cuda.stream(s):
            g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(REPS):
        for g, s in zip(graphs, streams):
            with torch.cuda.stream(s):
                g.replay()
    torch.cuda.synchronize()
    us = (time.perf_counter() - t0) / (REPS * K) * 1e6
    for st, _ in sts:
        st.close()
    return us


for parts in ([E], [E // 2, E // 2], [E // 4] * 4, [E // 2]):
    us = run(parts)
    print(f"{wl}: {len(parts)} stream(s) x {parts[0]} envs: {us:7.2f} us per step of {sum(parts)} envs  ({sum(parts) * kw['num_agents'] * 2 / us / 1e3:.3f} G env-steps/s)", flush=True)
