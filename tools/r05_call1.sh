#!/bin/bash
# GPU call 1 of round 5 (through gpurun): correctness gate of the step-kernel changes (counter-load fix, cooperative state store), the
# tolerance report of the per-quantity bounds, same-box A/Bs, the C5 harness's learning curve, the bench line.
tag=r05a
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== gate"; date
( timeout 600 python -m pytest tests/test_gated_gpu.py tests/test_facade_gpu.py tests/test_rollout_gpu.py tests/test_replay_gpu.py -m gpu -q -x --timeout=600 -p no:cacheprovider 2>&1 | tail -15 ) > gpurun_out/${tag}_gate.txt; tail -4 gpurun_out/${tag}_gate.txt
echo "== tolerance report"; date
rm -f gpurun_out/${tag}_tol_report.json
( QS_TOL_REPORT=$PWD/gpurun_out/${tag}_tol_report.json timeout 900 python -m pytest tests/test_hip_parity.py tests/test_hip_vs_reference_f32.py tests/test_fp32_parity_gpu.py -m gpu -q --timeout=900 -p no:cacheprovider 2>&1 | tail -25 ) > gpurun_out/${tag}_tol_pytest.txt; tail -4 gpurun_out/${tag}_tol_pytest.txt
python - <<'PY'
import json
try:
    d = json.load(open("gpurun_out/r05a_tol_report.json"))
except Exception as e:
    print("no report", e); raise SystemExit
by = {}
for k, v in d.items():
    ctx, q = k.rsplit("|", 1)
    prec = "f64" if " f64" in ctx else ("free" if ctx.startswith("free-running") else "f32")
    by.setdefault((prec, q), []).append((v, ctx))
for (prec, q), lst in sorted(by.items()):
    lst.sort(reverse=True)
    print(f"{prec:5s} {q:10s} worst |err|/allowed {lst[0][0]:8.3f} ({lst[0][1]}); above 1: {sum(1 for v, _ in lst if v > 1)} of {len(lst)}")
PY
echo "== A/B against round 4's tree"; date
bash tools/ab_tree.sh $tag c2 c4 2>&1 | tail -14
echo "== A/B cooperative store off"; date
bash tools/ab_flag.sh ${tag}_coop "-DQS_COOP_STORE=0" c2:1024 c4:512 2>&1 | tail -10
echo "== C5 harness curve"; date
timeout 600 python tools/ppo_c5.py --iterations 24 > gpurun_out/${tag}_ppo_c5_curve.txt 2> gpurun_out/${tag}_ppo_c5.err; tail -2 gpurun_out/${tag}_ppo_c5_curve.txt | cut -c1-600; tail -3 gpurun_out/${tag}_ppo_c5.err
echo "== batched env layers"; date
timeout 300 python tools/bench_batched_env.py > gpurun_out/${tag}_batched_env.txt 2>&1; tail -1 gpurun_out/${tag}_batched_env.txt | cut -c1-1500
echo "== wg probe c2"; date
QS_WG_WARM=1000 timeout 300 python tools/wg_times.py c2 > gpurun_out/${tag}_wg_c2.txt 2>&1; tail -22 gpurun_out/${tag}_wg_c2.txt
echo "== bench"; date
timeout 600 python bench.py > gpurun_out/${tag}_bench_c2_default.json 2> gpurun_out/${tag}_bench_c2_default.err; tail -c 1500 gpurun_out/${tag}_bench_c2_default.json
timeout 300 python bench.py --steps 20 --warmup 5 --no-c5-train --no-closed-loop --no-variants --no-f64 > gpurun_out/${tag}_bench_c2_steps20.json 2>> gpurun_out/${tag}_err.txt; python -c "
import json; d=json.loads(open('gpurun_out/${tag}_bench_c2_steps20.json').read().strip().splitlines()[-1]); print('steps20:', d['ms_per_step']*1e3, 'us', d['roofline']['frac'], d['config']['auto_reset'])"
date
