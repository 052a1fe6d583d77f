#!/usr/bin/env python
"""Emit the ISA of the config-specialised kernels of a bench workload (no GPU needed) and a per-barrier-region
instruction histogram of qs_spec_step:  python tools/spec_asm.py c2 [team] [out.s]"""
import collections
import os
import re
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import bench
from quad_swarm_rl_amd import config as qcfg, native

wl = sys.argv[1] if len(sys.argv) > 1 else "c2"
team = int(sys.argv[2]) if len(sys.argv) > 2 else -1
out = sys.argv[3] if len(sys.argv) > 3 else f"/tmp/spec_{wl}.s"
kw = dict(bench.WORKLOADS[wl]["kw"])
cfg = qcfg.make_config(num_envs=bench.WORKLOADS[wl]["num_envs"], seed=0, write_rew_info=False, **kw)
path = native.spec_build(cfg, team)
hdr = path.replace(".hsaco", ".h")
subprocess.check_call(f"/opt/rocm/bin/hipcc --genco --offload-arch=gfx950 -O3 -std=c++17 -S -DQS_SPEC_FILE='\"{hdr}\"' "
                      f"{native.CSRC}/qs_spec_kernels.hip -o {out} 2>/dev/null", shell=True)


def cls(k):
    if k in ("v_readlane_b32", "v_writelane_b32"): return "spill"
    if re.match(r"v_(sqrt|rcp|rsq|sin|cos|log|exp)_f(32|64)", k): return "trans"
    if re.match(r"v_(mul_lo_u32|mul_hi_u32|mad_u64_u32)", k): return "imul"
    if k.startswith("v_accvgpr"): return "agpr"
    if k.startswith("v_mov") or k.startswith("v_pk_mov"): return "vmov"
    if k.startswith("v_"): return "valu"
    if k.startswith("s_cbranch") or k.startswith("s_branch"): return "branch"
    if k.startswith("s_waitcnt"): return "wait"
    if k.startswith("s_nop"): return "nop"
    if k.startswith("s_load"): return "sload"
    if k.startswith("s_"): return "salu"
    if k.startswith("ds_"): return "lds"
    if k.split("_")[0] in ("global", "buffer", "flat", "scratch"): return "vmem"
    return "other"


lines = open(out).read().split("\n")
for kern in ("qs_spec_step", "qs_spec_rollout"):
    start = next(i for i, l in enumerate(lines) if l.startswith(kern + ":"))
    end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith(".section"))
    tot, cur, n = collections.Counter(), collections.Counter(), 0
    print(kern)
    for l in lines[start:end]:
        if not l.startswith("\t"):
            continue
        t = l.strip()
        if not t or t[0] in ".;":
            continue
        k = t.split()[0]
        if k == "s_barrier":
            print("  region", n, sum(cur.values()), dict(cur)); n += 1; cur = collections.Counter(); continue
        cur[cls(k)] += 1; tot[cls(k)] += 1
    print("  region", n, sum(cur.values()), dict(cur))
    print("  total", sum(tot.values()), dict(tot))
print(out)
