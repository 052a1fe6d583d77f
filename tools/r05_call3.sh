#!/bin/bash
# GPU call 3 of round 5: A/B of the early helper-wave stores, the whole `-m gpu` suite (assert mode) + smoke, the bench lines, rocprofv3
# kernel stats, PMC traffic of C2, the per-wire exchange sweep at world size 1
tag=r05c
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== A/B against round 4's tree"; date
bash tools/ab_tree.sh $tag c2 c3 c4 2>&1 | tail -20
echo "== A/B early stores off"; date
bash tools/ab_flag.sh ${tag}_early "-DQS_EARLY_STORES=0" c2:1024 c3:1024 2>&1 | tail -8
echo "== suite"; date
bash tools/gpu.sh $tag suite 2>&1 | tail -8
echo "== bench"; date
bash tools/gpu.sh $tag bench 2>&1 | tail -c 600
python -c "
import json
for f in ('gpurun_out/${tag}_bench_c2_default.json','gpurun_out/${tag}_bench_c2_steps20.json'):
    d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, d['ms_per_step']*1e3, 'us', d['value'], d['roofline']['frac'], d['roofline']['traffic'], (d.get('cpu_baseline') or {}).get('gpu_over_cpu',{}).get('ratio'), json.dumps(d['config'].get('c5'))[:600], d['config']['auto_reset'])
"
echo "== lines"; date
bash tools/gpu.sh $tag lines 2>&1 | tail -8
echo "== kstats"; date
bash tools/gpu.sh $tag kstats 2>&1 | tail -14
echo "== pmc c2"; date
bash tools/gpu.sh $tag pmc:c2 2>&1 | tail -3
echo "== exchange per wire, world 1"; date
timeout 300 python bench.py --workload c4 --force-gather --cpu-seconds 0 --no-f64 --no-closed-loop --no-variants --no-c5-train > gpurun_out/${tag}_bench_c4_gather_w1.json 2> gpurun_out/${tag}_bench_c4_gather_w1.err; python -c "
import json; d=json.loads(open('gpurun_out/${tag}_bench_c4_gather_w1.json').read().strip().splitlines()[-1]); print(d['ms_per_step']*1e3, d.get('wire')); print(json.dumps(d['config']['exchange_per_wire'], indent=1)[:3000]); print(json.dumps(d['config']['exchange'])[:800])"; tail -3 gpurun_out/${tag}_bench_c4_gather_w1.err
echo "== batched env"; date
timeout 300 python tools/bench_batched_env.py > gpurun_out/${tag}_batched_env.txt 2>&1; tail -1 gpurun_out/${tag}_batched_env.txt | cut -c1-1200
date
