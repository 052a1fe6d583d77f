#!/bin/bash
# round 3, GPU call F: (1) cold scenario code inlined (no private segment) vs out of line, C4; (2) the exchange after dropping the per-access fences.
tag=${1:-r03f}
mkdir -p gpurun_out
export TMPDIR=/tmp
fmt='import json,sys
for line in sys.stdin:
    if not line.startswith("{"): continue
    d=json.loads(line)
    r=d["roofline"]; s=d["config"].get("secondary") or {}
    print(sys.argv[1], "|", round(d["value"]/1e9,3), "G env-steps/s  ms_per_step", round(d["ms_per_step"]*1e3,2), "us  kernel_us", round(r["kernel_avg_us"],2), "frac", round(r["frac"],4), r["kernel_flavor"],
          "| secondary_us", round(s.get("ms_per_step",0)*1e3,2), "exchange_cost_us", s.get("exchange_cost_us_per_step"), (d["config"].get("exchange") or {}).get("transport"), (d["config"].get("exchange") or {}).get("peer_self_check"))'
Q="--cpu-seconds 0 --no-f64 --no-closed-loop --no-variants --rollout-steps 0 --profile-steps 0"
out=gpurun_out/${tag}_lines.txt; : > $out
for v in 0 1; do
  export QS_SPEC_EXTRA_FLAGS="-DQS_INLINE_COLD=$v"
  timeout 300 python bench.py --workload c4 --steps 2000 --warmup 100 $Q 2>>gpurun_out/${tag}_err.txt | python -c "$fmt" "c4 QS_INLINE_COLD=$v" | tee -a $out
  QS_WG_WARM=1200 QS_TIMING_EXTRA="-DQS_INLINE_COLD=$v" timeout 300 python tools/wg_times.py c4 > gpurun_out/${tag}_wg_c4_steady_inline$v.txt 2>&1; sed -n 3,4p gpurun_out/${tag}_wg_c4_steady_inline$v.txt
done
unset QS_SPEC_EXTRA_FLAGS
( timeout 600 python -m pytest tests/test_exchange_gpu.py -m gpu -q --maxfail=20 --timeout=600 -p no:cacheprovider 2>&1 | tail -12 ) > gpurun_out/${tag}_pytest.txt
tail -3 gpurun_out/${tag}_pytest.txt
for wl in c2 c4; do
  for tr in fused peer; do
    timeout 300 python bench.py --workload $wl --force-gather --transport $tr --steps 2048 --warmup 128 $Q > gpurun_out/${tag}_bench_${wl}_gather_$tr.json 2>>gpurun_out/${tag}_err.txt
    python -c "$fmt" "$wl force-gather $tr bf16" < gpurun_out/${tag}_bench_${wl}_gather_$tr.json | tee -a $out
  done
done
timeout 300 python bench.py --workload c4 --force-gather --steps 20 --warmup 5 $Q 2>>gpurun_out/${tag}_err.txt | python -c "$fmt" "c4 force-gather auto steps=20" | tee -a $out
tail -5 gpurun_out/${tag}_err.txt
