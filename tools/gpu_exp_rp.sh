#!/bin/bash
# rows-per-pass sweep of the single-wave kernel at a chip-filling batch.  usage: bash tools/gpu_exp_rp.sh <tag>
tag=$1
mkdir -p gpurun_out
out=gpurun_out/exp_rp_${tag}.txt
fmt='import json,sys
for line in sys.stdin:
    if not line.startswith("{"): continue
    d=json.loads(line)
    r=d["roofline"]
    print(sys.argv[1], "|", round(d["value"]/1e9,2), "G env-steps/s  kernel_us", round(r["kernel_avg_us"],2), "frac", round(r["frac"],4), r["kernel_flavor"])'
: > $out
for spec in "c2 131072" "c3 131072" "c4 32768"; do
  set -- $spec
  for rp in 16 32 64; do
    export QS_OBS_RP=$rp
    python bench.py --workload $1 --envs-per-gpu $2 --cpu-seconds 0 --steps 200 --warmup 20 --rollout-steps 0 --profile-steps 0 --no-f64 --no-closed-loop 2>&1 | python -c "$fmt" "$1 E=$2 rows_per_pass=$rp" | tee -a $out
  done
  unset QS_OBS_RP
done
export QS_SPEC_EXTRA_FLAGS="-DQS_EXP_NOFLUSH"
python bench.py --workload c2 --envs-per-gpu 131072 --cpu-seconds 0 --steps 200 --warmup 20 --rollout-steps 0 --profile-steps 0 --no-f64 --no-closed-loop 2>&1 | python -c "$fmt" "c2 E=131072 rows_per_pass=16 NOFLUSH" | tee -a $out
unset QS_SPEC_EXTRA_FLAGS
