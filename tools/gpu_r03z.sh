#!/bin/bash
# round 3, GPU call Z: C2 / C4 same-box A/B against the tree of commit c480c8b (_old) after the partial revert; mix / dynamic_formations step time.
tag=${1:-r03z}
mkdir -p gpurun_out
export TMPDIR=/tmp
fmt='import json,sys
for l in sys.stdin:
    if l.startswith("{"):
        d=json.loads(l); print(sys.argv[1], round(d["roofline"]["kernel_avg_us"],3))'
Q="--cpu-seconds 0 --no-f64 --no-closed-loop --no-variants --rollout-steps 0 --profile-steps 0"
out=$PWD/gpurun_out/${tag}_ab.txt; : > $out
root=$PWD
for rep in 1 2 3; do for tree in _old .; do
  cd $root/$tree
  for wl in c2 c4; do timeout 300 python bench.py --workload $wl --steps 3000 --warmup 200 $Q 2>/dev/null | python -c "$fmt" "$wl [$tree]" | tee -a $out; done
done; done
cd $root
timeout 300 python tools/scenario_times.py 1024 1200 static_same_goal static_diff_goal dynamic_formations mix 2>&1 | grep -v amdgpu | tee gpurun_out/${tag}_scenario_times.txt
