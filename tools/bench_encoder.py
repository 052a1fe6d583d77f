#!/usr/bin/env python
"""Fused policy encoder (qs_policy_encoder.hip) vs PyTorch on the observations of BASELINE config 2 (8192 agents, obs 54).
Prints one JSON line: us per forward, TFLOP/s (algorithmic FLOPs of the network) and the fraction of the dense bf16 MFMA peak; for
QuadMultiEncoder also the reference-precision kernels (precision="fp32": fp16-pair operands) with their distance from the module in float64.

    python tools/bench_encoder.py [agents] [attention | mha | sim2real]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from quad_swarm_rl_amd import policy

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
ATT = len(sys.argv) > 2 and sys.argv[2] == "attention"
S2R = len(sys.argv) > 2 and sys.argv[2] == "sim2real"   # QuadSingleHeadAttentionEncoder_Sim2Real, same observations
MHA = len(sys.argv) > 2 and sys.argv[2] in ("mha", "sim2real")   # QuadMultiHeadAttentionEncoder on C3's observations (obs 40: self 19, 2 neighbours, 9 SDF cells)
shape = dict(num_nbr=6, obst_dim=0, attention=ATT)
make = policy.make_reference_encoder
if MHA:
    shape = dict(num_nbr=2, obst_dim=9, self_dim=19)
    make = policy.make_reference_sim2real_encoder if S2R else policy.make_reference_mha_encoder
ref = make(seed=0, **shape).cuda()
fused = policy.FusedQuadEncoder(ref)
D = fused.params.obs_dim
obs = torch.rand((B, D), device="cuda") * 2 - 1
out = torch.empty((B, fused.out_dim), device="cuda")
H = 256
macs = 18 * H + H * H + shape["num_nbr"] * (6 * H + H * H) + (2 * H) * (2 * H)      # per agent, unpadded
if ATT:   # embedding on [self | nbr], value MLP, score MLP (e_mean half once per agent)
    macs = 18 * H + H * H + shape["num_nbr"] * ((18 + 6) * H + H * H + 2 * H * H + 2 * H * H + H) + H * H + (2 * H) * (2 * H)
if MHA:
    macs = (19 + 12 + 9) * H + 3 * H * H + 2 * 3 * H * 4 * H + 2 * 4 * H * H + 2 * 4 * 2 * H + 3 * H * 2 * H
if S2R:   # one-layer embeddings; q, k, v, fc 256 x 256 on two tokens; 2 x 2 scores and mixes; feed forward 768 -> 256
    macs = (19 + 12 + 9) * H + 2 * 3 * H * H + 2 * H * H + 2 * 2 * 2 * H + 3 * H * H
flops = 2.0 * macs * B


def timeit(fn, iters=300):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters


with torch.no_grad():
    t_fused_host = timeit(lambda: fused(obs, out=out))   # one python call per pass: includes the host-side call overhead
    t_fused = fused.benchmark(obs, out, 300)             # device-side rate: back-to-back launches, HIP events
    t_fp32 = timeit(lambda: ref(obs), 100)
    ref16 = make(seed=0, **shape).cuda().to(torch.bfloat16)
    obs16 = obs.to(torch.bfloat16)
    t_bf16 = timeit(lambda: ref16(obs16), 100)
PEAK = 2500.0   # TFLOP/s dense bf16 MFMA (MI355X_MICROARCH.md)
extra = {}
if True:   # reference precision: three fp16 MFMAs per product (same peak as bf16), against the fp32 module it stands in for
    import copy
    fused32 = policy.FusedQuadEncoder(ref, precision="fp32")
    with torch.no_grad():
        t32 = fused32.benchmark(obs, out, 300)
        want64 = copy.deepcopy(ref).double()(obs.double())
        e_sp = (fused32(obs).double() - want64).abs().max().item()
        e_bf = (fused(obs).double() - want64).abs().max().item()
        e_t32 = (ref(obs).double() - want64).abs().max().item()
    extra = {"reference_precision_us": t32 * 1e6, "reference_precision_speedup_vs_torch_fp32": t_fp32 / t32,
             "reference_precision_mfma_tflops": 3 * flops / t32 / 1e12, "reference_precision_frac_of_f16_mfma_peak": 3 * flops / t32 / 1e12 / PEAK,
             "max_abs_error_vs_float64_module": {"reference_precision": e_sp, "bf16": e_bf, "torch_fp32": e_t32}}
print(json.dumps({"kernel": "qs_encoder_s2r_kernel" if S2R else "qs_encoder_mha_kernel" if MHA else "qs_encoder_embed_kernel + qs_encoder_attn_kernel" if ATT else "qs_encoder_kernel", "agents": B, "obs_dim": D, "algorithmic_gflop": flops / 1e9,
                  "fused_us": t_fused * 1e6, "fused_us_one_python_call_per_pass": t_fused_host * 1e6, "fused_tflops": flops / t_fused / 1e12, "frac_of_bf16_mfma_peak": flops / t_fused / 1e12 / PEAK,
                  "torch_fp32_eager_us": t_fp32 * 1e6, "torch_bf16_eager_us": t_bf16 * 1e6,
                  "speedup_vs_torch_fp32": t_fp32 / t_fused, "speedup_vs_torch_bf16": t_bf16 / t_fused, **extra}))
