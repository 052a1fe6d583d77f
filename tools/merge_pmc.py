#!/usr/bin/env python
"""gpurun_out/pmc_<tag>_<wl>[_E<envs>]_traffic.json (tools/pmc.sh) -> profiles/<round>_pmc_traffic.json, the file bench.py reads roofline.traffic
from (key "<workload>:<envs per GPU>:<kernel>"), and copies the per-shape PMC summaries next to it.  usage: tools/merge_pmc.py <tag> <round>"""
import glob
import json
import os
import re
import shutil
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import bench  # noqa: E402

tag, rnd = sys.argv[1], sys.argv[2]
out = {}
for path in sorted(glob.glob(os.path.join(REPO, "gpurun_out", f"pmc_{tag}_*_traffic.json"))):
    m = re.match(rf"pmc_{tag}_(c\d)(?:_E(\d+))?_traffic\.json", os.path.basename(path))
    if not m:
        continue
    wl, E = m.group(1), int(m.group(2) or bench.WORKLOADS[m.group(1)]["num_envs"])
    summ = path.replace("_traffic.json", "_summary.txt")
    dst = os.path.join(REPO, "profiles", f"{rnd}_pmc_{wl}" + (f"_E{E}" if m.group(2) else "") + ".txt")
    if os.path.exists(summ):
        shutil.copy(summ, dst)
    for kernel, rec in json.load(open(path)).items():
        T = E * bench.WORKLOADS[wl]["kw"]["num_agents"]
        algo = bench.ALGO_BYTES_PER_DRONE_STEP[wl] * T
        rec = dict(rec, algorithmic_bytes=algo, traffic_over_algorithmic=(rec["fetch_bytes"] + rec["write_bytes"]) / algo,
                   read_bytes_per_drone=rec["fetch_bytes"] / T, write_bytes_per_drone=rec["write_bytes"] / T,
                   source=f"profiles/{os.path.basename(dst)}: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes with --kernel-trace only (tools/pmc.sh), KiB x 1024; "
                          "FETCH_SIZE x 2 and WRITE_SIZE x 1 by the calibration on known byte counts in this access pattern (profiles/r04_pmc_calibration.txt)")
        out[f"{wl}:{E}:{kernel}"] = rec
json.dump(out, open(os.path.join(REPO, "profiles", f"{rnd}_pmc_traffic.json"), "w"), indent=1)
for k, v in out.items():
    print(f"{k}: read {v['read_bytes_per_drone']:.1f} + written {v['write_bytes_per_drone']:.1f} B per drone-step = {v['traffic_over_algorithmic']:.3f} x algorithmic")
