#!/bin/bash
# same-box A/B of two source trees (this one vs build_exp/old = a git worktree of an earlier commit): interleaved bench runs
fmt='import json,sys
d=json.loads(sys.stdin.readline()); print(sys.argv[1], round(d["ms_per_step"]*1e3,2), "us", round(d["roofline"]["frac"],4), d["roofline"]["kernel_flavor"])'
for rep in 1 2; do
  for tree in . build_exp/old; do
    ( cd $tree; python bench.py --workload c2 --envs-per-gpu 131072 --cpu-seconds 0 --no-f64 --no-closed-loop --rollout-steps 0 2>/dev/null | python -c "$fmt" "$tree c2 E=131072 3000 steps";
      python bench.py --workload c2 --envs-per-gpu 131072 --cpu-seconds 0 --no-f64 --no-closed-loop --rollout-steps 0 --steps 200 --warmup 20 2>/dev/null | python -c "$fmt" "$tree c2 E=131072 200 steps";
      python bench.py --cpu-seconds 0 --no-f64 --no-closed-loop 2>/dev/null | python -c "$fmt" "$tree c2 default" )
  done
done
