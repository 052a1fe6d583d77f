#!/bin/bash
# round 3, GPU call X (final state): whole gpu suite, default bench line (+ --steps 20), bench lines of the other shapes, exchange lines, per-scenario
# step time, the batched SF env on mix, closed loop, rocprofv3 kernel stats + PMC passes.
tag=${1:-r03x}
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 1500 python -m pytest tests -m gpu -q --maxfail=20 --timeout=900 -p no:cacheprovider 2>&1 | tail -40 ) > gpurun_out/${tag}_pytest.txt
tail -5 gpurun_out/${tag}_pytest.txt
timeout 600 python bench.py > gpurun_out/${tag}_bench_c2_default.json 2> gpurun_out/${tag}_bench_c2_default.err
timeout 300 python bench.py --steps 20 --warmup 5 > gpurun_out/${tag}_bench_c2_steps20.json 2>> gpurun_out/${tag}_err.txt
fmt='import json,sys
for line in sys.stdin:
    if not line.startswith("{"): continue
    d=json.loads(line)
    r=d["roofline"]; s=d["config"].get("secondary") or {}; v=d["config"].get("variants") or {}
    print(sys.argv[1], "|", round(d["value"]/1e9,3), "G env-steps/s  ms_per_step", round(d["ms_per_step"]*1e3,2), "us  kernel_us", round(r["kernel_avg_us"],2), "frac", round(r["frac"],4), r["kernel_flavor"],
          "| secondary_us", round(s.get("ms_per_step",0)*1e3,2), "exchange_cost_us", s.get("exchange_cost_us_per_step"), (d["config"].get("exchange") or {}).get("transport"), (d["config"].get("exchange") or {}).get("peer_self_check"),
          "| shaped_us", (v.get("shaped_episode_sums") or {}).get("kernel_avg_us"), "mix_us", (v.get("mix_scenarios_shaped") or {}).get("kernel_avg_us"))'
Q="--cpu-seconds 0 --no-f64 --no-closed-loop --no-variants --rollout-steps 0 --profile-steps 0"
out=gpurun_out/${tag}_lines.txt; : > $out
python -c "$fmt" "c2 default" < gpurun_out/${tag}_bench_c2_default.json | tee -a $out
python -c "$fmt" "c2 default --steps 20" < gpurun_out/${tag}_bench_c2_steps20.json | tee -a $out
for wl in c3 c4; do timeout 300 python bench.py --workload $wl --steps 2000 --warmup 100 $Q 2>>gpurun_out/${tag}_err.txt | python -c "$fmt" "$wl" | tee -a $out; done
for wl in c2 c3; do timeout 300 python bench.py --workload $wl --envs-per-gpu 131072 --steps 600 --warmup 100 $Q 2>>gpurun_out/${tag}_err.txt | python -c "$fmt" "$wl E=131072" | tee -a $out; done
timeout 300 python bench.py --workload c4 --envs-per-gpu 32768 --steps 600 --warmup 100 $Q 2>>gpurun_out/${tag}_err.txt | python -c "$fmt" "c4 E=32768" | tee -a $out
for wl in c2 c4; do
  for tr in fused rccl; do
    timeout 300 python bench.py --workload $wl --force-gather --transport $tr --steps 2048 --warmup 128 $Q > gpurun_out/${tag}_bench_${wl}_gather_$tr.json 2>>gpurun_out/${tag}_err.txt
    python -c "$fmt" "$wl force-gather $tr bf16" < gpurun_out/${tag}_bench_${wl}_gather_$tr.json | tee -a $out
  done
done
timeout 300 python bench.py --workload c4 --force-gather --steps 20 --warmup 5 $Q > gpurun_out/${tag}_bench_c4_gather_auto_steps20.json 2>>gpurun_out/${tag}_err.txt
python -c "$fmt" "c4 force-gather auto steps=20" < gpurun_out/${tag}_bench_c4_gather_auto_steps20.json | tee -a $out
timeout 600 python tools/scenario_times.py 1024 1200 > gpurun_out/${tag}_scenario_times.txt 2>&1; tail -18 gpurun_out/${tag}_scenario_times.txt
timeout 300 python tools/bench_batched_env.py 1024 3200 > gpurun_out/${tag}_batched_env_host.json 2>>gpurun_out/${tag}_err.txt; cat gpurun_out/${tag}_batched_env_host.json
for e in mean_embed attention; do timeout 200 python tools/bench_rollout.py $e 32 2>/dev/null | tee gpurun_out/${tag}_closed_loop_$e.json; done
timeout 900 bash tools/profile_round.sh ${tag} 2>&1 | tail -30
tail -5 gpurun_out/${tag}_err.txt
