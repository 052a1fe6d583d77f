#!/usr/bin/env python
"""Register / LDS footprint of the config-specialised kernels of a bench workload (no GPU needed):
  python tools/spec_resources.py c2 [team] [extra hipcc flags...]
Compiles qs_spec_kernels.hip exactly like qs_create does (same flags, incl. the fp32 fast-math set) to assembly and prints,
per kernel, the code-object notes: VGPRs, AGPRs, SGPRs, spills, scratch, LDS, and the waves/SIMD the VGPR count allows
(gfx950: 512 VGPRs per SIMD lane, allocation granule 8)."""
import os
import re
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import bench
from quad_swarm_rl_amd import config as qcfg, native


def resources(wl="c2", team=-1, extra="", precision="f32", out=None, envs=None, **overrides):
    kw = dict(bench.WORKLOADS[wl]["kw"], **overrides)   # e.g. quads_mode="mix": the full-scenario kernels on the workload's shape
    cfg = qcfg.make_config(num_envs=envs or bench.WORKLOADS[wl]["num_envs"], seed=0, write_rew_info=False, precision=precision, **kw)
    if extra:
        os.environ["QS_SPEC_EXTRA_FLAGS"] = extra
    path = native.spec_build(cfg, team)
    hdr = path.replace(".hsaco", ".h")
    out = out or f"/tmp/spec_{wl}_{team}.s"
    fm = "-ffast-math -fno-slp-vectorize" if precision == "f32" else ""
    if team > 0:   # the team objects' scheduler flags (kSpecFlagsTeam / kSpecFlagsTeam8 in quadswarm_hip.hip)
        fm += " -mllvm -amdgpu-sched-strategy=max-ilp" + (" -mllvm -enable-post-misched=0" if team == 8 else "")
    elif team == 0 and precision == "f32":   # the single-wave fp32 objects' default: the RP trackers (kSpecFlagsSingleF32 in quadswarm_hip.hip)
        fm += " " + os.environ.get("QS_SPEC_SINGLE_FLAGS", "-mllvm -amdgpu-use-amdgpu-trackers")
    subprocess.check_call(f"/opt/rocm/bin/hipcc --genco --offload-arch=gfx950 -O3 -std=c++17 {fm} {extra} -S -DQS_SPEC_FILE='\"{hdr}\"' "
                          f"{native.CSRC}/qs_spec_kernels.hip -o {out} 2>/dev/null", shell=True)
    text = open(out).read()
    res = {}
    for m in re.finditer(r"\.amdhsa_kernel (\w+)(.*?)\.end_amdhsa_kernel", text, re.S):
        name, body = m.group(1), m.group(2)
        def g(key):
            mm = re.search(r"\.amdhsa_" + key + r"\s+(\d+)", body)
            return int(mm.group(1)) if mm else None
        res[name] = dict(next_free_vgpr=g("next_free_vgpr"), accum_offset=g("accum_offset"), next_free_sgpr=g("next_free_sgpr"),
                         lds=g("group_segment_fixed_size"), scratch=g("private_segment_fixed_size"))
    for m in re.finditer(r"; Function info:|^(\w+):\n", text, re.M):
        pass
    # per-kernel comment block: "; NumVgprs: N", "; NumAgprs", "; ScratchSize", "; Occupancy", "; SGPRSpill"/"VGPRSpill"
    for m in re.finditer(r"\.size\s+(\w+), \.Lfunc_end\d+-\w+\n(.*?)(?=\n\t\.(?:text|section))", text, re.S):
        name, blk = m.group(1), m.group(2)
        if name not in res:
            continue
        for key in ("NumSgprs", "NumVgprs", "NumAgprs", "TotalNumVgprs", "ScratchSize", "Occupancy", "LDSByteSize", "sgpr_spill_count", "vgpr_spill_count"):
            mm = re.search(r"; " + key + r": (\d+)", blk)
            if mm:
                res[name][key] = int(mm.group(1))
        mm = re.search(r"codeLenInByte = (\d+)", blk)
        if mm:
            res[name]["code_bytes"] = int(mm.group(1))
    return res, out


if __name__ == "__main__":
    wl = sys.argv[1] if len(sys.argv) > 1 else "c2"
    team = int(sys.argv[2]) if len(sys.argv) > 2 else -1
    extra = " ".join(sys.argv[3:])
    res, out = resources(wl, team, extra)
    for k, v in res.items():
        print(k, v)
    print(out)
