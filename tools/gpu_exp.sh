#!/bin/bash
# A/B experiments on the throughput kernel at a chip-filling batch + PMC of the default build.  usage: bash tools/gpu_exp.sh <tag> "<flagset1>" "<flagset2>" ...
tag=$1; shift
mkdir -p gpurun_out
out=gpurun_out/exp_${tag}.txt
fmt='import json,sys
for line in sys.stdin:
    if not line.startswith("{"): continue
    d=json.loads(line)
    r=d["roofline"]
    print(sys.argv[1], "|", round(d["value"]/1e9,2), "G env-steps/s  kernel_us", round(r["kernel_avg_us"],2), "frac", round(r["frac"],4), r["kernel_flavor"])'
: > $out
WL=${WL:-c2}; ENVS=${ENVS:-131072}
for flags in "$@"; do
  if [ "$flags" = default ]; then unset QS_SPEC_EXTRA_FLAGS; else export QS_SPEC_EXTRA_FLAGS="$flags"; fi
  python bench.py --workload $WL --envs-per-gpu $ENVS --cpu-seconds 0 --steps 200 --warmup 20 --rollout-steps 0 --profile-steps 0 --no-f64 --no-closed-loop 2>&1 | python -c "$fmt" "$WL E=$ENVS [$flags]" | tee -a $out
done
unset QS_SPEC_EXTRA_FLAGS
