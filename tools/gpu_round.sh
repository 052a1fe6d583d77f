#!/bin/bash
# One GPU-box session: parity suite, bench lines, occupancy sweep.  usage: gpurun -- 'bash tools/gpu_round.sh <tag>'
tag=${1:-r02a}
mkdir -p gpurun_out
( timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 ) > gpurun_out/${tag}_pytest.txt
tail -3 gpurun_out/${tag}_pytest.txt
python bench.py > gpurun_out/${tag}_bench_default.json 2> gpurun_out/${tag}_bench_default.err; tail -c 3000 gpurun_out/${tag}_bench_default.json
python bench.py --steps 20 --warmup 5 --cpu-seconds 0 > gpurun_out/${tag}_bench_20.json 2>/dev/null
python bench.py --workload c4 --force-gather --cpu-seconds 0 --steps 500 > gpurun_out/${tag}_bench_c4_gather1.json 2> gpurun_out/${tag}_bench_c4_gather1.err
for wl in c3 c4 c1; do python bench.py --workload $wl --cpu-seconds 0 --steps 2000 > gpurun_out/${tag}_bench_$wl.json 2>/dev/null; done
bash tools/occupancy_sweep.sh $tag
