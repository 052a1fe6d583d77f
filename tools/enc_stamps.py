#!/usr/bin/env python
"""Phase stamps of workgroup 0 / wave 0 of the fused encoder (build the library with -DENC_TIMING, point QS_ENC_LIB at it)."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from quad_swarm_rl_amd import policy

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
ref = policy.make_reference_encoder(seed=0, num_nbr=6, obst_dim=0).cuda()
fused = policy.FusedQuadEncoder(ref)
obs = torch.rand((B, fused.params.obs_dim), device="cuda") * 2 - 1
out = torch.empty((B, 512), device="cuda")
for _ in range(5):
    fused(obs, out=out)
torch.cuda.synchronize()
st = (C.c_ulonglong * 16)()
policy.lib().qs_enc_stamps(st)
names = ["start", "obs staged", "self enc", "obst enc", "n1 gemm (last pass)", "n1 stored+barrier", "n2 gemm", "nbr done", "ff gemm", "end"]
if os.environ.get("QS_ENC_PP", "1") != "0":   # pp_body: wave 0 (the early half)
    names = ["start", "obs staged", "n1(A) s1", "n2(A)", "n1(B)", "s2", "n2(B)", "[o1 - o2] -", "f(lo) + G f(hi)", "end"]
us = fused.benchmark(obs, out, 200) * 1e6
print(f"kernel {us:.1f} us per forward; stamps span {st[9] - st[0]} ticks -> {us * 1e3 / max(st[9] - st[0], 1):.2f} ns per tick (if workgroup 0 spans the kernel)")
t0 = st[0]
prev = t0
for i, n in enumerate(names):
    print(f"{n:24s} {st[i] - t0:8d} (+{st[i] - prev})")
    prev = st[i]

# start / end of every workgroup (100 MHz clock): how many run at once, and for how long each
n = min((B + 15) // 16, 8192)
wt = (C.c_ulonglong * (2 * n))()
if hasattr(policy.lib(), "qs_enc_wg_times") and policy.lib().qs_enc_wg_times(wt, n) == 0:
    import numpy as np
    t = np.array(list(wt), dtype=np.float64).reshape(n, 2)
    t = t[t[:, 1] > 0]
    t0w = t[:, 0].min()
    dur = (t[:, 1] - t[:, 0]) * 0.01
    print(f"{len(t)} workgroups: start spread {(t[:, 0].max() - t0w) * 0.01:.1f} us, duration min/median/max {dur.min():.1f}/{np.median(dur):.1f}/{dur.max():.1f} us, "
          f"last end {(t[:, 1].max() - t0w) * 0.01:.1f} us; started within the first 2 us: {(t[:, 0] - t0w < 200).sum()}")
