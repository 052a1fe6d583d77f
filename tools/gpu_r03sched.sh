#!/bin/bash
# round 3: the AMDGPU machine scheduler's max-ILP strategy on the specialised kernels (lone waves: ILP, not occupancy, is what they lack); same box, alternating.
tag=${1:-r03sched}
mkdir -p gpurun_out
export TMPDIR=/tmp
fmt='import json,sys
for l in sys.stdin:
    if l.startswith("{"):
        d=json.loads(l); v=d["config"].get("variants") or {}
        print(sys.argv[1], round(d["roofline"]["kernel_avg_us"],3), "shaped", (v.get("shaped_episode_sums") or {}).get("kernel_avg_us"), "mix", (v.get("mix_scenarios_shaped") or {}).get("kernel_avg_us"))'
Q="--cpu-seconds 0 --no-f64 --no-closed-loop --rollout-steps 0 --profile-steps 0 --prewarm 1000"
out=gpurun_out/${tag}_ab.txt; : > $out
for rep in 1 2; do for flags in "" "-mllvm -amdgpu-sched-strategy=max-ilp"; do
  if [ -z "$flags" ]; then unset QS_SPEC_EXTRA_FLAGS; else export QS_SPEC_EXTRA_FLAGS="$flags"; fi
  timeout 300 python bench.py --workload c2 --steps 3000 --warmup 200 $Q 2>/dev/null | python -c "$fmt" "c2 [$flags]" | tee -a $out
  for wl in c3 c4; do timeout 300 python bench.py --workload $wl --steps 3000 --warmup 200 $Q --no-variants 2>/dev/null | python -c "$fmt" "$wl [$flags]" | tee -a $out; done
  if [ $rep = 1 ]; then
    timeout 300 python bench.py --workload c2 --envs-per-gpu 131072 --steps 600 --warmup 100 $Q --no-variants 2>/dev/null | python -c "$fmt" "c2 E=131072 [$flags]" | tee -a $out
    timeout 300 python bench.py --workload c4 --envs-per-gpu 32768 --steps 600 --warmup 100 $Q --no-variants 2>/dev/null | python -c "$fmt" "c4 E=32768 [$flags]" | tee -a $out
  fi
done; done
