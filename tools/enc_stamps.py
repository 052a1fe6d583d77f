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
t0 = st[0]
prev = t0
for i, n in enumerate(names):
    print(f"{n:24s} {st[i] - t0:8d} (+{st[i] - prev})")
    prev = st[i]
