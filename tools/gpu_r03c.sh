#!/bin/bash
# round 3, GPU call C: steady-state per-workgroup phase timing of the C4 launch, the whole gpu suite, the bench lines after
# (a) logging outputs on a helper wave, (b) nt observation stores by default.
tag=${1:-r03c}
mkdir -p gpurun_out
export TMPDIR=/tmp
for v in 0 1; do
  QS_WG_WARM=1200 QS_TIMING_EXTRA="-DQS_PAIR_ONCE=$v" timeout 300 python tools/wg_times.py c4 > gpurun_out/${tag}_wg_c4_steady_paironce$v.txt 2>&1; tail -22 gpurun_out/${tag}_wg_c4_steady_paironce$v.txt
done
QS_WG_WARM=1200 timeout 300 python tools/wg_times.py c2 > gpurun_out/${tag}_wg_c2_steady.txt 2>&1; tail -20 gpurun_out/${tag}_wg_c2_steady.txt
( timeout 1500 python -m pytest tests -m gpu -q --maxfail=20 --timeout=900 -p no:cacheprovider 2>&1 | tail -80 ) > gpurun_out/${tag}_pytest.txt
tail -6 gpurun_out/${tag}_pytest.txt
fmt='import json,sys
for line in sys.stdin:
    if not line.startswith("{"): continue
    d=json.loads(line)
    r=d["roofline"]; v=d["config"].get("variants") or {}
    print(sys.argv[1], "|", round(d["value"]/1e9,3), "G env-steps/s  kernel_us", round(r["kernel_avg_us"],2), "frac", round(r["frac"],4), r["kernel_flavor"],
          "| shaped_us", (v.get("shaped_episode_sums") or {}).get("kernel_avg_us"), "shaped+rew_info_us", (v.get("shaped_episode_sums_and_rew_info") or {}).get("kernel_avg_us"),
          "dw_off_us", (v.get("downwash_off") or {}).get("kernel_avg_us"))'
Q="--cpu-seconds 0 --no-f64 --no-closed-loop --rollout-steps 0 --profile-steps 0"
out=gpurun_out/${tag}_lines.txt; : > $out
timeout 600 python bench.py > gpurun_out/${tag}_bench_c2_default.json 2> gpurun_out/${tag}_bench_c2_default.err
python -c "$fmt" "c2 default" < gpurun_out/${tag}_bench_c2_default.json | tee -a $out
for wl in c3 c4; do timeout 300 python bench.py --workload $wl --steps 2000 --warmup 100 $Q 2>>gpurun_out/${tag}_err.txt | python -c "$fmt" "$wl" | tee -a $out; done
for v in 0 1; do
  QS_SPEC_EXTRA_FLAGS="-DQS_PAIR_ONCE=$v" timeout 300 python bench.py --workload c4 --steps 2000 --warmup 100 $Q --no-variants 2>>gpurun_out/${tag}_err.txt | python -c "$fmt" "c4 QS_PAIR_ONCE=$v" | tee -a $out
done
for wl in c2 c3; do timeout 300 python bench.py --workload $wl --envs-per-gpu 131072 --steps 600 --warmup 100 $Q --no-variants 2>>gpurun_out/${tag}_err.txt | python -c "$fmt" "$wl E=131072" | tee -a $out; done
timeout 300 python bench.py --workload c4 --envs-per-gpu 32768 --steps 600 --warmup 100 $Q --no-variants 2>>gpurun_out/${tag}_err.txt | python -c "$fmt" "c4 E=32768" | tee -a $out
tail -5 gpurun_out/${tag}_err.txt
