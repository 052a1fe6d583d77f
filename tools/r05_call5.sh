#!/bin/bash
# GPU call 5 of round 5: the --force-gather path with stage tracing, the whole `-m gpu` suite + smoke on the final tree, final bench lines and kernel stats
tag=r05e
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== exchange per wire, world 1 (traced)"; date
BENCH_TRACE=1 timeout 300 python -X faulthandler bench.py --workload c4 --force-gather --cpu-seconds 0 --no-f64 --no-closed-loop --no-variants --no-c5-train > gpurun_out/${tag}_bench_c4_gather_w1.json 2> gpurun_out/${tag}_bench_c4_gather_w1.err; echo "rc=$?"; tail -25 gpurun_out/${tag}_bench_c4_gather_w1.err | cut -c1-300; wc -c gpurun_out/${tag}_bench_c4_gather_w1.json
python -c "
import json; d=json.loads(open('gpurun_out/${tag}_bench_c4_gather_w1.json').read().strip().splitlines()[-1]); print(d['ms_per_step']*1e3, d.get('wire')); print(json.dumps(d['config']['exchange_per_wire'], indent=1)[:3000]); print(json.dumps(d['config']['exchange'])[:900])" 2>&1 | tail -60
echo "== suite"; date
bash tools/gpu.sh $tag suite 2>&1 | tail -8
echo "== bench"; date
bash tools/gpu.sh $tag bench 2>&1 | tail -c 300
python -c "
import json
for f in ('gpurun_out/${tag}_bench_c2_default.json','gpurun_out/${tag}_bench_c2_steps20.json'):
    d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, d['ms_per_step']*1e3, 'us', d['value'], d['roofline']['frac'], d['roofline']['traffic'], (d.get('cpu_baseline') or {}).get('gpu_over_cpu',{}).get('ratio'), json.dumps(d['config'].get('c5'))[:900])
"
echo "== kstats"; date
bash tools/gpu.sh $tag kstats 2>&1 | tail -14
date
