#!/bin/bash
tag=${1:-r02e}
mkdir -p gpurun_out
( timeout 900 python -m pytest tests -m gpu -q --deselect tests/test_hip_vs_reference.py 2>&1 | tail -12 ) > gpurun_out/${tag}_pytest.txt; tail -12 gpurun_out/${tag}_pytest.txt
fmt='import json,sys
for line in sys.stdin:
    if not line.startswith("{"): continue
    d=json.loads(line)
    r=d["roofline"]
    print(sys.argv[1], "|", round(d["value"]/1e9,3), "G env-steps/s  kernel_us", round(r["kernel_avg_us"],2), "frac", round(r["frac"],4), r["kernel_flavor"])'
out=gpurun_out/exp_c4_${tag}.txt; : > $out
for team in 4 8 0; do for flags in default "-DQS_EXP_NOBIG"; do
  export QS_TEAM=$team; if [ "$flags" = default ]; then unset QS_SPEC_EXTRA_FLAGS; else export QS_SPEC_EXTRA_FLAGS="$flags"; fi
  [ $team = 0 ] && [ "$flags" != default ] && continue
  python bench.py --workload c4 --cpu-seconds 0 --steps 2000 --warmup 100 --rollout-steps 0 --profile-steps 0 --no-f64 --no-closed-loop 2>&1 | python -c "$fmt" "c4 512 envs QS_TEAM=$team [$flags]" | tee -a $out
done; done
unset QS_TEAM QS_SPEC_EXTRA_FLAGS
for team in 8 4 0; do export QS_TEAM=$team; python bench.py --workload c2 --cpu-seconds 0 --steps 2000 --warmup 100 --rollout-steps 0 --profile-steps 0 --no-f64 --no-closed-loop 2>&1 | python -c "$fmt" "c2 1024 envs QS_TEAM=$team" | tee -a $out; done
unset QS_TEAM
