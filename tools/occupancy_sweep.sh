#!/bin/bash
# Throughput (single-wave) kernel at a chip-filling batch: occupancy-hint sweep.  usage (GPU box): bash tools/occupancy_sweep.sh [tag]
# Writes gpurun_out/occ_<tag>.txt: one line per (workload, envs, waves_per_eu hint).
tag=${1:-sweep}
out=gpurun_out/occ_${tag}.txt
mkdir -p gpurun_out
fmt='import json,sys
for line in sys.stdin:
    if not line.startswith("{"): continue
    d=json.loads(line)
    r=d["roofline"]
    print(sys.argv[1], round(d["value"]/1e9,2), "G env-steps/s  kernel_us", round(r["kernel_avg_us"],2), "frac", round(r["frac"],4), r["kernel_flavor"])'
: > $out
for spec in "c2 131072" "c3 131072" "c4 32768"; do
  set -- $spec
  for occ in default 0 3 4 5; do
    if [ $occ = default ]; then unset QS_SPEC_EXTRA_FLAGS; else export QS_SPEC_EXTRA_FLAGS="-DQS_WAVES_PER_EU=$occ"; fi
    python bench.py --workload $1 --envs-per-gpu $2 --cpu-seconds 0 --steps 200 --warmup 20 --rollout-steps 0 --profile-steps 0 --no-f64 --no-closed-loop 2>&1 | python -c "$fmt" "$1 E=$2 waves_per_eu=$occ" | tee -a $out
  done
done
unset QS_SPEC_EXTRA_FLAGS
