#!/bin/bash
# compile-flag experiments on the config-specialised kernels (GPU box): QS_SPEC_EXTRA_FLAGS is part of the cache key
fmt='import json,sys
for line in sys.stdin:
    if not line.startswith("{"): continue
    d=json.loads(line); r=d["config"].get("open_loop_rollout") or {}
    print(sys.argv[1], "| us/step", round(d["ms_per_step"]*1e3,2), "| rollout us/step", round(r.get("us_per_step",0),2), "|", d["roofline"]["kernel"])'
while IFS= read -r flags; do
  QS_SPEC_EXTRA_FLAGS="$flags" python bench.py --workload ${1:-c2} --cpu-seconds 0 --steps 2048 2>&1 | python -c "$fmt" "[$flags]"
done <<'LIST'

-fno-signed-zeros -fno-trapping-math
-fno-signed-zeros -fno-trapping-math -freciprocal-math
-fno-signed-zeros -fno-trapping-math -fassociative-math
-ffinite-math-only
-fapprox-func
-fno-signed-zeros -fno-trapping-math -fassociative-math -freciprocal-math -ffinite-math-only
-ffast-math -fno-slp-vectorize
LIST
