#!/bin/bash
# One entry point for the GPU box (through gpurun): tools/gpu.sh <tag> <task> [task ...]
#   suite            whole `-m gpu` test suite + smoke                      -> gpurun_out/<tag>_pytest.txt
#   quick            the parity / facade / exchange tests only (fast gate)   -> gpurun_out/<tag>_pytest_quick.txt
#   bench            default bench line + the driver's --steps 20 line       -> gpurun_out/<tag>_bench_c2_{default,steps20}.json
#   lines            bench lines of c2 / c3 / c4 and the three 2^20-drone shapes (no extras) -> gpurun_out/<tag>_bench_lines.txt
#   kstats           rocprofv3 --kernel-trace --stats of c2 / c3 / c4         -> gpurun_out/kt_<tag>_<wl>_stats.txt
#   calib            FETCH_SIZE / WRITE_SIZE against known byte counts        -> gpurun_out/pmc_calib_<tag>.{txt,json}
#   pmc              PMC passes: c2, c3, c4 and their 2^20-drone shapes        -> gpurun_out/pmc_<tag>_*_{summary.txt,traffic.json}
#   pmc:<wl>[:E]     one PMC pass set (e.g. pmc:c2:131072)
#   wg:<wl>          per-workgroup timing probe (tools/wg_times.py)           -> gpurun_out/<tag>_wg_<wl>.txt
#   run:<script.py>  python <script.py>                                       -> gpurun_out/<tag>_<script>.txt
#   tol              the parity tests in REPORT mode: worst |err| / allowed per case and quantity (tests/tolerances.py) -> gpurun_out/<tag>_tol_report.json
#   abtree[:wls]     same-box A/B of the step kernels against another tree in build_exp/old (tools/ab_tree.sh)         -> gpurun_out/<tag>_ab_tree.txt
#   c5[:iters[:batch[:bf16|fp32]]]  the in-tree PPO harness on BASELINE config 5 (tools/ppo_c5.py), learning curve                 -> gpurun_out/<tag>_ppo_c5.txt
#   gather           bench.py --workload c4 --force-gather: the exchange at world size 1, every wire                   -> gpurun_out/<tag>_bench_c4_gather_w1.json
#   diff:<case>[:var:flags]  two code objects of one parity case side by side, bit for bit + against the oracle (tools/flag_diff.py; QS_SPEC_VERIFY=0:
#                    the flagged object as the compiler delivers it)                                                      -> gpurun_out/<tag>_flag_diff_<case>.txt
#   gdb:<case>[:var:flags]   the same under rocgdb with precise memory faults: faulting instruction, registers            -> gpurun_out/<tag>_rocgdb_<case>.txt
#   noise[:shapes]   what the random draws cost at run time: the throughput shapes with the sensor + thrust noise configured off (27 + 4 normal draws per
#                    drone-step not made) beside the default, same box, interleaved                                      -> gpurun_out/<tag>_noise_share.txt
#   enc              the policy-encoder tests (tests/test_policy_encoder_gpu.py, test_encoder_fixtures.py) + tools/bench_encoder.py lines (mean_embed, attention:
#                    bf16 and reference precision)                                                                    -> gpurun_out/<tag>_enc_{pytest,bench}.txt
#   sweep:<n>        scheduler / switch sweep <n> of tools/sched_sweep.py (objects prebuilt with `SWEEP=<n> python tools/sched_sweep.py build`)  -> gpurun_out/<tag>_sched_sweep.txt
tag=$1; shift
mkdir -p gpurun_out
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
BIG="c2:131072 c3:131072 c4:32768"
kt() { # rocprofv3 kernel stats of one workload
  wl=$1
  ( cd /tmp && rocprofv3 --kernel-trace --stats -d /tmp/rocprof_kt_${tag}_$wl -o k -- python $R/bench.py --workload $wl --steps 2000 --warmup 100 --cpu-seconds 0 --no-f64 --no-closed-loop --no-variants --no-c5-train > $R/gpurun_out/kt_${tag}_$wl.json 2> $R/gpurun_out/kt_${tag}_$wl.err )
  db=$(ls /tmp/rocprof_kt_${tag}_$wl/*.db /tmp/rocprof_kt_${tag}_$wl/*/*.db 2>/dev/null | head -1)
  [ -n "$db" ] && python $R/tools/rocprof_summary.py $db "rocprofv3 --kernel-trace --stats -- python bench.py --workload $wl --steps 2000 --warmup 100 --cpu-seconds 0 --no-f64 --no-closed-loop --no-variants --no-c5-train" > $R/gpurun_out/kt_${tag}_${wl}_stats.txt
  head -4 $R/gpurun_out/kt_${tag}_${wl}_stats.txt
}
for task in "$@"; do
  echo "== $task"
  case $task in
    suite)
      ( timeout 1700 python -m pytest tests -m gpu -q -rf --tb=line --durations=25 --maxfail=40 --timeout=900 -p no:cacheprovider 2>&1 | tail -120 ) > gpurun_out/${tag}_pytest.txt
      tail -4 gpurun_out/${tag}_pytest.txt
      timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 ;;
    quick)
      ( timeout 900 python -m pytest tests/test_hip_parity.py tests/test_fp32_parity_gpu.py tests/test_facade_gpu.py tests/test_exchange_gpu.py tests/test_rollout_gpu.py -m gpu -q -x --timeout=600 -p no:cacheprovider 2>&1 | tail -15 ) > gpurun_out/${tag}_pytest_quick.txt
      tail -3 gpurun_out/${tag}_pytest_quick.txt ;;
    bench)
      timeout 600 python bench.py > gpurun_out/${tag}_bench_c2_default.json 2> gpurun_out/${tag}_bench_c2_default.err; tail -c 700 gpurun_out/${tag}_bench_c2_default.json
      timeout 300 python bench.py --steps 20 --warmup 5 > gpurun_out/${tag}_bench_c2_steps20.json 2>> gpurun_out/${tag}_err.txt ;;
    lines)
      : > gpurun_out/${tag}_bench_lines.txt
      for wl in c2 c3 c4; do timeout 300 python bench.py --workload $wl --cpu-seconds 0 --no-f64 --no-closed-loop --no-variants --no-c5-train >> gpurun_out/${tag}_bench_lines.txt 2>> gpurun_out/${tag}_err.txt; done
      for we in $BIG; do timeout 300 python bench.py --workload ${we%%:*} --envs-per-gpu ${we##*:} --steps 400 --warmup 50 --prewarm 200 --rollout-steps 0 --cpu-seconds 0 --no-f64 --no-closed-loop --no-variants --no-c5-train >> gpurun_out/${tag}_bench_lines.txt 2>> gpurun_out/${tag}_err.txt; done
      python - <<PY
import json
for l in open("gpurun_out/${tag}_bench_lines.txt"):
    if l.startswith("{"):
        d = json.loads(l); r = d["roofline"]
        print(d["config"]["workload"][:40], "| us/step %.3f | frac %.4f | traffic %s" % (1e3 * d["ms_per_step"], r["frac"], r["traffic"]))
PY
      ;;
    kstats) for wl in c2 c3 c4; do kt $wl; done ;;
    calib) bash tools/pmc_calib.sh $tag | tail -12 ;;
    pmc)
      for wl in c2 c3 c4; do bash tools/pmc.sh ${tag}_$wl $wl | tail -1; done
      for we in $BIG; do bash tools/pmc.sh ${tag}_${we%%:*}_E${we##*:} ${we%%:*} --envs-per-gpu ${we##*:} | tail -1; done ;;
    pmc:*)
      IFS=: read -r _ wl E <<< "$task"
      if [ -n "$E" ]; then bash tools/pmc.sh ${tag}_${wl}_E$E $wl --envs-per-gpu $E | tail -1; else bash tools/pmc.sh ${tag}_$wl $wl | tail -1; fi ;;
    wg:*) wl=${task#wg:}; QS_WG_WARM=1000 timeout 600 python tools/wg_times.py $wl > gpurun_out/${tag}_wg_$wl.txt 2>&1; tail -12 gpurun_out/${tag}_wg_$wl.txt ;;
    tol)
      rm -f gpurun_out/${tag}_tol_report.json
      ( QS_TOL_REPORT=$PWD/gpurun_out/${tag}_tol_report.json timeout 900 python -m pytest tests/test_hip_parity.py tests/test_hip_vs_reference_f32.py tests/test_fp32_parity_gpu.py -m gpu -q --timeout=900 -p no:cacheprovider 2>&1 | tail -5 ) > gpurun_out/${tag}_tol_pytest.txt; tail -2 gpurun_out/${tag}_tol_pytest.txt ;;
    abtree*) wls=${task#abtree}; wls=${wls#:}; bash tools/ab_tree.sh $tag ${wls//,/ } 2>&1 | tail -20 ;;
    c5*)
      IFS=: read -r _ iters batch prec <<< "$task"
      timeout 900 python tools/ppo_c5.py --iterations ${iters:-24} --batch_size ${batch:-1024} --sampler_precision ${prec:-fp32} > gpurun_out/${tag}_ppo_c5${prec:+_$prec}.txt 2> gpurun_out/${tag}_ppo_c5.err; tail -1 gpurun_out/${tag}_ppo_c5${prec:+_$prec}.txt | cut -c1-900 ;;
    gather)
      timeout 300 python bench.py --workload c4 --force-gather --cpu-seconds 0 --no-f64 --no-closed-loop --no-variants --no-c5-train > gpurun_out/${tag}_bench_c4_gather_w1.json 2> gpurun_out/${tag}_bench_c4_gather_w1.err
      python -c "import json; d=json.loads(open('gpurun_out/${tag}_bench_c4_gather_w1.json').read().strip().splitlines()[-1]); print({w: (round(v['ms_per_step']*1e3,2), v['verified_against_rccl_gather_after']) for w, v in d['config']['exchange_per_wire'].items()})" ;;
    diff:*|gdb:*)
      IFS=: read -r kind case var flags <<< "$task"
      var=${var:-QS_SPEC_SINGLE_FLAGS}; flags=${flags:--mllvm -amdgpu-use-amdgpu-trackers}
      if [ $kind = diff ]; then
        ( QS_SPEC_VERIFY=0 QS_TEAM=${QS_TEAM:-0} FLAGS_VAR="$var" FLAGS_B="$flags" timeout 300 python -u tools/flag_diff.py $case 7 40 ) > gpurun_out/${tag}_flag_diff_$case.txt 2>&1
        grep -v "^A \|^B " gpurun_out/${tag}_flag_diff_$case.txt | tail -6 | cut -c1-300
      else
        ( QS_SPEC_VERIFY=0 QS_TEAM=${QS_TEAM:-0} FLAGS_VAR="$var" FLAGS_B="$flags" timeout 300 rocgdb -batch -ex "set pagination off" -ex "set confirm off" -ex "set amdgpu precise-memory on" -ex run \
            -ex "info threads" -ex "x/40i \$pc-120" -ex "info registers" --args python -u tools/flag_diff.py $case 7 40 ) > gpurun_out/${tag}_rocgdb_$case.txt 2>&1
        grep -n "received signal\|=> " gpurun_out/${tag}_rocgdb_$case.txt | head -4
      fi ;;
    noise*)
      shapes=${task#noise}; shapes=${shapes#:}; shapes=${shapes:-$BIG}
      out=gpurun_out/${tag}_noise_share.txt
      echo "# bench.py --steps 400, us per step: default configuration / sense_noise=None + thrust_noise_ratio=0 (no Philox + Box-Muller draws on the per-drone path)" > $out
      for rep in 1 2; do for we in ${shapes//,/ }; do for v in default nonoise; do
        extra=""; [ $v = nonoise ] && extra="--set sense_noise=None --set thrust_noise_ratio=0.0"
        us=$(timeout 300 python bench.py --workload ${we%%:*} --envs-per-gpu ${we##*:} --steps 400 --warmup 50 --prewarm 200 --rollout-steps 0 --cpu-seconds 0 --no-f64 --no-closed-loop --no-variants --no-c5-train $extra 2>>gpurun_out/${tag}_err.txt | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('%.3f' % (1e3*d['ms_per_step']))")
        echo "$we rep $rep $v: $us" | tee -a $out
      done; done; done ;;
    enc)
      ( timeout 1500 python -m pytest tests/test_policy_encoder_gpu.py tests/test_encoder_fixtures.py -m gpu -q -rf --tb=short --maxfail=12 --timeout=600 -p no:cacheprovider 2>&1 | tail -80 ) > gpurun_out/${tag}_enc_pytest.txt
      tail -5 gpurun_out/${tag}_enc_pytest.txt
      : > gpurun_out/${tag}_enc_bench.txt
      for a in "8192" "8192 attention" "4096" "131072" "8192 mha" "8192 sim2real"; do timeout 300 python tools/bench_encoder.py $a >> gpurun_out/${tag}_enc_bench.txt 2>> gpurun_out/${tag}_err.txt; done
      python - <<PYEOF
import json
for l in open("gpurun_out/${tag}_enc_bench.txt"):
    d = json.loads(l)
    print(d["kernel"], d["agents"], "bf16 %.1f us (%.3f of peak)" % (d["fused_us"], d["frac_of_bf16_mfma_peak"]), "| reference precision %.1f us (%.3f), torch fp32 %.1f us |" % (d.get("reference_precision_us", 0), d.get("reference_precision_frac_of_f16_mfma_peak", 0), d["torch_fp32_eager_us"]), d.get("max_abs_error_vs_float64_module"))
PYEOF
      ;;
    sweep:*) SWEEP=${task#sweep:} SWEEP_TAG=$tag python tools/sched_sweep.py run 2 2>&1 | tail -30 ;;
    run:*) sc=${task#run:}; timeout 900 python $sc > gpurun_out/${tag}_$(basename $sc .py).txt 2>&1; tail -25 gpurun_out/${tag}_$(basename $sc .py).txt ;;
    *) echo "unknown task $task" ;;
  esac
done
