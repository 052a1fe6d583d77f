fmt='import json,sys
for line in sys.stdin:
    if not line.startswith("{"): continue
    d=json.loads(line)
    print(sys.argv[1], round(d["value"]/1e6,1), "M env-steps/s  us/step", round(d["ms_per_step"]*1e3,2), "frac", round(d["roofline"]["frac"],4), d["roofline"]["kernel_flavor"])'
for occ in 2 3 4 5; do
QS_SPEC_EXTRA_FLAGS="-DQS_WAVES_PER_EU=$occ" python bench.py --workload c2 --envs-per-gpu 131072 --cpu-seconds 0 --steps 200 --warmup 20 --rollout-steps 0 2>&1 | python -c "$fmt" "occ=$occ"
done
