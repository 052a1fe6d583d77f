#!/bin/bash
# GPU call 4 of round 5: the final step kernel (counter-load fix only) against round 4's tree, the tolerance cases that failed in call 3,
# the per-wire exchange sweep at world size 1 (with faulthandler), kernel stats / PMC of the final kernel, the bench lines
tag=r05d
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== A/B against round 4's tree"; date
bash tools/ab_tree.sh $tag c2 c3 c4 2>&1 | tail -20
echo "== tolerance cases"; date
( timeout 600 python -m pytest tests/test_hip_parity.py -k "x_dense_obst or x_n40_obst or c3_n8_obst or e_n17" -m gpu -q --timeout=600 -p no:cacheprovider 2>&1 | tail -8 ) > gpurun_out/${tag}_tol_cases.txt; tail -3 gpurun_out/${tag}_tol_cases.txt
echo "== exchange per wire, world 1"; date
timeout 300 python -X faulthandler bench.py --workload c4 --force-gather --cpu-seconds 0 --no-f64 --no-closed-loop --no-variants --no-c5-train > gpurun_out/${tag}_bench_c4_gather_w1.json 2> gpurun_out/${tag}_bench_c4_gather_w1.err; echo "rc=$?"; tail -25 gpurun_out/${tag}_bench_c4_gather_w1.err | cut -c1-300
python -c "
import json; d=json.loads(open('gpurun_out/${tag}_bench_c4_gather_w1.json').read().strip().splitlines()[-1]); print(d['ms_per_step']*1e3, d.get('wire')); print(json.dumps(d['config']['exchange_per_wire'], indent=1)[:3000]); print(json.dumps(d['config']['exchange'])[:900])" 2>&1 | tail -60
echo "== no sweep"; date
timeout 300 python -X faulthandler bench.py --workload c4 --force-gather --no-wire-sweep --cpu-seconds 0 --no-f64 --no-closed-loop --no-variants --no-c5-train > gpurun_out/${tag}_bench_c4_gather_w1_nosweep.json 2> gpurun_out/${tag}_bench_c4_gather_w1_nosweep.err; echo "rc=$?"; tail -5 gpurun_out/${tag}_bench_c4_gather_w1_nosweep.err | cut -c1-300
echo "== bench"; date
bash tools/gpu.sh $tag bench 2>&1 | tail -c 300
python -c "
import json
for f in ('gpurun_out/${tag}_bench_c2_default.json','gpurun_out/${tag}_bench_c2_steps20.json'):
    d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, d['ms_per_step']*1e3, 'us', d['value'], d['roofline']['frac'], d['roofline']['traffic'], (d.get('cpu_baseline') or {}).get('gpu_over_cpu',{}).get('ratio'), d['config']['auto_reset'])
"
echo "== lines"; date
bash tools/gpu.sh $tag lines 2>&1 | tail -8
echo "== kstats"; date
bash tools/gpu.sh $tag kstats 2>&1 | tail -14
echo "== pmc c2"; date
bash tools/gpu.sh $tag pmc:c2 2>&1 | tail -2
QS_SPEC_EXTRA_FLAGS="-DQS_SKIP_ROWS=1" bash tools/pmc.sh ${tag}skip_c2 c2 2>&1 | tail -1
date
echo "== C5 harness: graphed vs eager minibatch step"; date
timeout 300 python tools/ppo_c5.py --iterations 4 > gpurun_out/r05d_ppo_graph.txt 2> gpurun_out/r05d_ppo_graph.err; tail -1 gpurun_out/r05d_ppo_graph.txt | cut -c1-700; tail -3 gpurun_out/r05d_ppo_graph.err | cut -c1-300
timeout 300 python tools/ppo_c5.py --iterations 2 --graph_update=false > gpurun_out/r05d_ppo_eager.txt 2>> gpurun_out/r05d_ppo_graph.err; tail -1 gpurun_out/r05d_ppo_eager.txt | cut -c1-500
timeout 300 python tools/ppo_c5.py --iterations 12 --batch_size 8192 > gpurun_out/r05d_ppo_graph_b8192.txt 2>> gpurun_out/r05d_ppo_graph.err; tail -1 gpurun_out/r05d_ppo_graph_b8192.txt | cut -c1-700
date
