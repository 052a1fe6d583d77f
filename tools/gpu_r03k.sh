#!/bin/bash
# round 3, GPU call K: the step semantics in one header (qs_step_sem.h) used by both step bodies: whole gpu suite, bench lines, step time per scenario.
tag=${1:-r03k}
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 1500 python -m pytest tests -m gpu -q --maxfail=20 --timeout=900 -p no:cacheprovider 2>&1 | tail -60 ) > gpurun_out/${tag}_pytest.txt
tail -5 gpurun_out/${tag}_pytest.txt
fmt='import json,sys
for line in sys.stdin:
    if not line.startswith("{"): continue
    d=json.loads(line)
    r=d["roofline"]; v=d["config"].get("variants") or {}
    print(sys.argv[1], "|", round(d["value"]/1e9,3), "G env-steps/s  ms_per_step", round(d["ms_per_step"]*1e3,2), "us  kernel_us", round(r["kernel_avg_us"],2), "frac", round(r["frac"],4), r["kernel_flavor"],
          "| shaped_us", (v.get("shaped_episode_sums") or {}).get("kernel_avg_us"))'
Q="--cpu-seconds 0 --no-f64 --no-closed-loop --rollout-steps 0 --profile-steps 0"
out=gpurun_out/${tag}_lines.txt; : > $out
for wl in c2 c3 c4; do timeout 300 python bench.py --workload $wl --steps 2000 --warmup 100 $Q 2>>gpurun_out/${tag}_err.txt | python -c "$fmt" "$wl" | tee -a $out; done
for wl in c2 c3; do timeout 300 python bench.py --workload $wl --envs-per-gpu 131072 --steps 600 --warmup 100 $Q --no-variants 2>>gpurun_out/${tag}_err.txt | python -c "$fmt" "$wl E=131072" | tee -a $out; done
timeout 300 python bench.py --workload c4 --envs-per-gpu 32768 --steps 600 --warmup 100 $Q --no-variants 2>>gpurun_out/${tag}_err.txt | python -c "$fmt" "c4 E=32768" | tee -a $out
timeout 300 python bench.py --workload c4 --force-gather --transport fused --steps 2048 --warmup 128 $Q --no-variants 2>>gpurun_out/${tag}_err.txt | python -c "$fmt" "c4 force-gather fused" | tee -a $out
timeout 600 python tools/scenario_times.py 1024 1200 > gpurun_out/${tag}_scenario_times.txt 2>&1; tail -18 gpurun_out/${tag}_scenario_times.txt
timeout 600 python tools/bench_batched_env.py > gpurun_out/${tag}_batched_env_host.json 2>gpurun_out/${tag}_batched_env_host.err; cat gpurun_out/${tag}_batched_env_host.json
tail -5 gpurun_out/${tag}_err.txt
