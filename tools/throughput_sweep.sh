#!/bin/bash
# team (QS_TEAM=1) vs single-wave (QS_TEAM=0) step kernels over the batch size, config-specialised objects (GPU box)
fmt='import json,sys
for line in sys.stdin:
    if not line.startswith("{"): continue
    d=json.loads(line)
    print(sys.argv[1], round(d["value"]/1e6,1), "M env-steps/s  us/step", round(d["ms_per_step"]*1e3,2), "frac", round(d["roofline"]["frac"],4), d["roofline"]["kernel_flavor"])'
for E in ${@:-1024 1536 2048 3072}; do for t in 0 1; do
QS_TEAM=$t python bench.py --workload ${WL:-c2} --envs-per-gpu $E --cpu-seconds 0 --steps 500 --warmup 20 --rollout-steps 0 2>&1 | python -c "$fmt" "E$E team=$t"
done; done
