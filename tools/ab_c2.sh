#!/bin/bash
# same-box A/B of the C2 / C4 step kernels of two source trees: this one vs build_exp/old (create it with
#   git worktree add -f build_exp/old <commit>  and build its library + spec objects there); runs interleaved
fmt='import json,sys
d=json.loads(sys.stdin.readline()); r=d["config"].get("open_loop_rollout") or {}
print(sys.argv[1], round(d["ms_per_step"]*1e3,3), "us", "rollout", round(r.get("us_per_step",0),3))'
for rep in 1 2 3; do
  for tree in . build_exp/old; do
    ( cd $tree; python bench.py --cpu-seconds 0 --no-f64 --no-closed-loop 2>/dev/null | python -c "$fmt" "$tree c2";
      python bench.py --workload c4 --cpu-seconds 0 --no-f64 --no-closed-loop --steps 1500 2>/dev/null | python -c "$fmt" "$tree c4" )
  done
done
