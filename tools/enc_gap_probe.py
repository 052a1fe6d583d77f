#!/usr/bin/env python
"""What does the fused encoder's [B, 512] fp32 feature write-out cost?  Back-to-back forwards (HIP events) of mean_embed at 8192 agents:
features written (16.8 MB per pass) / only a 4-wide fused head written (what a rollout segment does)."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from quad_swarm_rl_amd import policy

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
att = len(sys.argv) > 2 and sys.argv[2] == "attention"
ref = policy.make_reference_encoder(seed=0, num_nbr=6, obst_dim=0, attention=att).cuda()
fused = policy.FusedQuadEncoder(ref)
obs = torch.rand((B, fused.params.obs_dim), device="cuda") * 2 - 1
out = torch.empty((B, 512), device="cuda")
us_feat = fused.benchmark(obs, out, 300) * 1e6
fused.set_head(torch.randn((4, 512), device="cuda") * 0.05, torch.zeros(4, device="cuda"))
head_out = torch.empty((B, 4), device="cuda")
fused.forward_head(obs, head_out=head_out)
P = fused.params
w, b = fused._head
P.head_w, P.head_b, P.head_out, P.head_dim = w.data_ptr(), b.data_ptr(), head_out.data_ptr(), 4
ms = C.c_double(0)
L = policy.lib()
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
rc = L.qs_enc_benchmark(obs.data_ptr(), B, C.byref(P), C.c_void_p(None), st, 300, C.byref(ms))
assert rc == 0, L.qs_enc_last_error()
us_head = ms.value * 1e3
rc = L.qs_enc_benchmark(obs.data_ptr(), B, C.byref(P), out.data_ptr(), st, 300, C.byref(ms))
us_both = ms.value * 1e3
P.head_dim = 0
print(f"{'attention' if att else 'mean_embed'} {B} agents: features written {us_feat:.2f} us | head only (no feature stores) {us_head:.2f} us | both {us_both:.2f} us")
