#!/bin/bash
# usage: tools/ab_tree.sh <tag> [workloads...]   (GPU box)  same-box A/B of the step kernels of two source trees: this one vs build_exp/old
# (git worktree add -f build_exp/old <commit>; build its library + spec objects there first); interleaved, 3 rounds, us per step by HIP events.
tag=$1; shift
wls=${@:-"c2 c4"}
mkdir -p gpurun_out
R=$PWD
fmt='import json,sys
d=json.loads([l for l in sys.stdin if l.startswith("{")][-1]); r=d["config"].get("open_loop_rollout") or {}
print(sys.argv[1], round(d["ms_per_step"]*1e3,3), "us per step; open-loop", round(r.get("us_per_step",0),3))'
for rep in 1 2 3; do
  for tree in . build_exp/old; do
    for wl in $wls; do
      ( cd $R/$tree; python bench.py --workload $wl --cpu-seconds 0 --no-f64 --no-closed-loop --no-variants --no-c5-train --steps 3000 2>/dev/null | python -c "$fmt" "$tree $wl" ) | tee -a $R/gpurun_out/${tag}_ab_tree.txt
    done
  done
done
