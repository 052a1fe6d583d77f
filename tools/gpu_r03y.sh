#!/bin/bash
# round 3, GPU call Y: C4 same-box A/B against the tree of commit c480c8b (_old), the rccl transport stepped eagerly.
tag=${1:-r03y}
mkdir -p gpurun_out
export TMPDIR=/tmp
fmt='import json,sys
for l in sys.stdin:
    if l.startswith("{"):
        d=json.loads(l); s=d["config"].get("secondary") or {}; print(sys.argv[1], round(d["roofline"]["kernel_avg_us"],3), round(d["ms_per_step"]*1e3,2), s.get("exchange_cost_us_per_step"), (d["config"].get("exchange") or {}).get("transport"))'
Q="--cpu-seconds 0 --no-f64 --no-closed-loop --no-variants --rollout-steps 0 --profile-steps 0"
out=$PWD/gpurun_out/${tag}_ab.txt; : > $out
root=$PWD
for rep in 1 2 3; do for tree in _old .; do
  cd $root/$tree
  for wl in c4; do timeout 300 python bench.py --workload $wl --steps 3000 --warmup 200 $Q 2>/dev/null | python -c "$fmt" "$wl [$tree]" | tee -a $out; done
done; done
cd $root
for wl in c2 c4; do timeout 300 python bench.py --workload $wl --force-gather --transport rccl --steps 2048 --warmup 128 $Q 2>>gpurun_out/${tag}_err.txt > gpurun_out/${tag}_bench_${wl}_gather_rccl.json; python -c "$fmt" "$wl force-gather rccl (eager)" < gpurun_out/${tag}_bench_${wl}_gather_rccl.json | tee -a $out; done
