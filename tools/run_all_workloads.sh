fmt='import json,sys
for line in sys.stdin:
    if not line.startswith("{"): continue
    d=json.loads(line); r=d["config"].get("open_loop_rollout") or {}
    print(sys.argv[1], round(d["value"]/1e6,1), "M env-steps/s  us/step", round(d["ms_per_step"]*1e3,2), "kernel", round(d["roofline"]["kernel_avg_us"],2), "frac", round(d["roofline"]["frac"],4), "| rollout us/step", round(r.get("us_per_step",0),2), "M/s", round(r.get("value",0)/1e6,1))'
for wl in c1 c2 c3 c4; do python bench.py --workload $wl --cpu-seconds 0 --steps 1024 2>&1 | python -c "$fmt" $wl; done
for E in 16384 131072; do python bench.py --workload c2 --envs-per-gpu $E --cpu-seconds 0 --steps 256 --warmup 20 2>&1 | python -c "$fmt" c2_E$E; QS_TEAM=1 python bench.py --workload c2 --envs-per-gpu $E --cpu-seconds 0 --steps 256 --warmup 20 2>&1 | python -c "$fmt" c2_E${E}_team; done
