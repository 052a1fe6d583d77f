#!/bin/bash
# round 3, GPU call P: two builds of the same source (noise floor of the A/B) after the output redirection moved into buffer resources on the headline kernels (same box, alternating), rollout tests.
tag=${1:-r03p}
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 600 python -m pytest tests/test_rollout_gpu.py -m gpu -q --maxfail=10 --timeout=500 -p no:cacheprovider 2>&1 | tail -5 ) > gpurun_out/${tag}_pytest.txt; tail -3 gpurun_out/${tag}_pytest.txt
fmt='import json,sys
for l in sys.stdin:
    if l.startswith("{"):
        d=json.loads(l); print(sys.argv[1], round(d["roofline"]["kernel_avg_us"],3))'
Q="--cpu-seconds 0 --no-f64 --no-closed-loop --no-variants --rollout-steps 0 --profile-steps 0"
out=gpurun_out/${tag}_ab.txt; : > $out
for rep in 1 2 3; do for flags in "" "-DQS_AB_DUMMY=1"; do
  if [ -z "$flags" ]; then unset QS_SPEC_EXTRA_FLAGS; else export QS_SPEC_EXTRA_FLAGS="$flags"; fi
  for wl in c2 c4; do timeout 300 python bench.py --workload $wl --steps 3000 --warmup 200 $Q 2>/dev/null | python -c "$fmt" "$wl [$flags]" | tee -a $out; done
done; done
