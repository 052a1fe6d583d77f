#!/bin/bash
# round 3, GPU call B: (1) where / when the workgroups of the C4 launch run (tools/wg_times.py), (2) the tests touched since call A,
# (3) the exchange at world size 1 (graph-captured), (4) nt stores of the observation rows at 131072 envs with PMC traffic.
tag=${1:-r03b}
mkdir -p gpurun_out
export TMPDIR=/tmp
for v in 0 1; do
  QS_TIMING_EXTRA="-DQS_PAIR_ONCE=$v" timeout 300 python tools/wg_times.py c4 > gpurun_out/${tag}_wg_c4_paironce$v.txt 2>&1; tail -6 gpurun_out/${tag}_wg_c4_paironce$v.txt
done
timeout 300 python tools/wg_times.py c2 > gpurun_out/${tag}_wg_c2.txt 2>&1; tail -6 gpurun_out/${tag}_wg_c2.txt
timeout 300 python tools/wg_times.py c2 num_envs=2048 > gpurun_out/${tag}_wg_c2_2048.txt 2>&1; tail -6 gpurun_out/${tag}_wg_c2_2048.txt
QS_TEAM=8 timeout 300 python tools/wg_times.py c4 > gpurun_out/${tag}_wg_c4_team8.txt 2>&1; tail -6 gpurun_out/${tag}_wg_c4_team8.txt
( timeout 1200 python -m pytest tests/test_exchange_gpu.py tests/test_fp32_parity_gpu.py tests/test_replay_gpu.py tests/test_facade_gpu.py tests/test_sf_protocol_gpu.py \
    tests/test_policy_encoder_gpu.py::test_multi_head_kernels_are_run_to_run_identical_above_one_workgroup_per_cu tests/test_rollout_gpu.py \
    -m gpu -q --maxfail=20 --timeout=600 -p no:cacheprovider 2>&1 | tail -80 ) > gpurun_out/${tag}_pytest.txt
tail -4 gpurun_out/${tag}_pytest.txt
fmt='import json,sys
for line in sys.stdin:
    if not line.startswith("{"): continue
    d=json.loads(line)
    r=d["roofline"]; s=d["config"].get("secondary") or {}
    print(sys.argv[1], "|", round(d["value"]/1e9,3), "G env-steps/s  ms_per_step", round(d["ms_per_step"]*1e3,2), "us  kernel_us", round(r["kernel_avg_us"],2), "frac", round(r["frac"],4), r["kernel_flavor"],
          "| secondary_us", round(s.get("ms_per_step",0)*1e3,2), "exchange_cost_us", s.get("exchange_cost_us_per_step"), (d["config"].get("exchange") or {}).get("transport"), (d["config"].get("exchange") or {}).get("peer_self_check"))'
Q="--cpu-seconds 0 --no-f64 --no-closed-loop --no-variants --rollout-steps 0 --profile-steps 0"
out=gpurun_out/${tag}_lines.txt; : > $out
for wl in c2 c4; do
  for tr in peer rccl; do
    timeout 300 python bench.py --workload $wl --force-gather --transport $tr --steps 2048 --warmup 128 $Q > gpurun_out/${tag}_bench_${wl}_gather_$tr.json 2>>gpurun_out/${tag}_err.txt
    python -c "$fmt" "$wl force-gather $tr bf16 graph" < gpurun_out/${tag}_bench_${wl}_gather_$tr.json | tee -a $out
  done
done
timeout 300 python bench.py --force-gather --transport peer --wire f32 --steps 2048 --warmup 128 $Q 2>>gpurun_out/${tag}_err.txt | python -c "$fmt" "c2 force-gather peer f32 graph" | tee -a $out
timeout 300 python bench.py --force-gather --steps 20 --warmup 5 $Q 2>>gpurun_out/${tag}_err.txt | python -c "$fmt" "c2 force-gather auto steps=20" | tee -a $out
# bandwidth regime: plain vs nt stores of the observation rows, bench + PMC traffic
for nt in 0 1; do
  if [ $nt = 1 ]; then export QS_SPEC_EXTRA_FLAGS="-DQS_NT_OBS=1"; else unset QS_SPEC_EXTRA_FLAGS; fi
  timeout 300 python bench.py --workload c2 --envs-per-gpu 131072 --steps 600 --warmup 100 $Q 2>>gpurun_out/${tag}_err.txt | python -c "$fmt" "c2 E=131072 QS_NT_OBS=$nt" | tee -a $out
  timeout 600 bash tools/pmc.sh ${tag}_E131072_nt$nt c2 --envs-per-gpu 131072 --no-variants 2>>gpurun_out/${tag}_err.txt | tail -2
done
unset QS_SPEC_EXTRA_FLAGS
tail -5 gpurun_out/${tag}_err.txt
