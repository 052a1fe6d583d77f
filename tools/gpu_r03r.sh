#!/bin/bash
# round 3, GPU call R: same-box A/B of the C2 / C4 step between the tree before the semantics refactor (_old = commit c480c8b) and HEAD.
tag=${1:-r03r}
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
