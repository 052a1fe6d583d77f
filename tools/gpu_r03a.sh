#!/bin/bash
# round 3, GPU call A: the whole gpu suite (new: exchange, fp32 parity; changed: pair-once team kernels), then the bench lines that
# price this round's changes.  usage (through gpurun): bash tools/gpu_r03a.sh [tag]
tag=${1:-r03a}
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 1500 python -m pytest tests -m gpu -q --maxfail=20 --timeout=900 -p no:cacheprovider 2>&1 | tail -120 ) > gpurun_out/${tag}_pytest.txt
tail -5 gpurun_out/${tag}_pytest.txt
fmt='import json,sys
for line in sys.stdin:
    if not line.startswith("{"): continue
    d=json.loads(line)
    r=d["roofline"]; s=d["config"].get("secondary") or {}
    print(sys.argv[1], "|", round(d["value"]/1e9,3), "G env-steps/s  ms_per_step", round(d["ms_per_step"]*1e3,2), "us  kernel_us", round(r["kernel_avg_us"],2), "frac", round(r["frac"],4), r["kernel_flavor"],
          "| secondary_us", round(s.get("ms_per_step",0)*1e3,2), "exchange_cost_us", s.get("exchange_cost_us_per_step"), (d["config"].get("exchange") or {}).get("transport"), (d["config"].get("exchange") or {}).get("peer_self_check"))'
Q="--cpu-seconds 0 --no-f64 --no-closed-loop --no-variants --rollout-steps 0 --profile-steps 0"
out=gpurun_out/${tag}_lines.txt; : > $out
# default line (what the driver runs), full
timeout 600 python bench.py > gpurun_out/${tag}_bench_c2_default.json 2> gpurun_out/${tag}_bench_c2_default.err
python -c "$fmt" "c2 default" < gpurun_out/${tag}_bench_c2_default.json | tee -a $out
# C4 shard: pair-once (default) vs the round-2 scan, 4 vs 8 waves
for team in 4 8; do for flags in default "-DQS_PAIR_ONCE=0"; do
  export QS_TEAM=$team; if [ "$flags" = default ]; then unset QS_SPEC_EXTRA_FLAGS; else export QS_SPEC_EXTRA_FLAGS="$flags"; fi
  timeout 300 python bench.py --workload c4 --steps 2000 --warmup 100 $Q 2>>gpurun_out/${tag}_err.txt | python -c "$fmt" "c4 512 envs QS_TEAM=$team [$flags]" | tee -a $out
done; done
unset QS_TEAM QS_SPEC_EXTRA_FLAGS
timeout 300 python bench.py --workload c3 --steps 2000 --warmup 100 $Q 2>>gpurun_out/${tag}_err.txt | python -c "$fmt" "c3" | tee -a $out
# the exchange at world size 1: captured graph (peer / rccl), eager (segment 0), round 2's torch gather; f32 and bf16 wire
for wl in c2; do
  for tr in peer rccl torch; do
    timeout 300 python bench.py --workload $wl --force-gather --transport $tr --steps 2048 --warmup 128 $Q > gpurun_out/${tag}_bench_${wl}_gather_$tr.json 2>>gpurun_out/${tag}_err.txt
    python -c "$fmt" "$wl force-gather $tr bf16 graph" < gpurun_out/${tag}_bench_${wl}_gather_$tr.json | tee -a $out
  done
  timeout 300 python bench.py --workload $wl --force-gather --transport peer --segment 0 --steps 2048 --warmup 128 $Q 2>>gpurun_out/${tag}_err.txt | python -c "$fmt" "$wl force-gather peer bf16 eager" | tee -a $out
  timeout 300 python bench.py --workload $wl --force-gather --transport peer --wire f32 --steps 2048 --warmup 128 $Q 2>>gpurun_out/${tag}_err.txt | python -c "$fmt" "$wl force-gather peer f32 graph" | tee -a $out
done
timeout 300 python bench.py --workload c4 --force-gather --transport peer --steps 2048 --warmup 128 $Q 2>>gpurun_out/${tag}_err.txt | python -c "$fmt" "c4 force-gather peer bf16 graph" | tee -a $out
timeout 300 python bench.py --force-gather --steps 20 --warmup 5 $Q 2>>gpurun_out/${tag}_err.txt | python -c "$fmt" "c2 force-gather auto steps=20" | tee -a $out
# phase timing of the C4 team kernel (new and old scan)
timeout 300 python tools/phase_timing.py c4 > gpurun_out/${tag}_phase_c4.txt 2>&1
QS_TIMING_EXTRA="-DQS_PAIR_ONCE=0" timeout 300 python tools/phase_timing.py c4 > gpurun_out/${tag}_phase_c4_oldscan.txt 2>&1
timeout 300 python tools/phase_timing.py c2 > gpurun_out/${tag}_phase_c2.txt 2>&1
head -20 gpurun_out/${tag}_phase_c4.txt
tail -3 gpurun_out/${tag}_err.txt
timeout 200 tools/ubench_hbm > gpurun_out/${tag}_ubench_hbm.txt 2>&1; grep BEST gpurun_out/${tag}_ubench_hbm.txt
