#!/bin/bash
# round 3, last GPU call: whole gpu suite + smoke + default bench on the final tree, kernel stats of c2 / c3 / c4.
tag=${1:-r03fin}
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 1500 python -m pytest tests -m gpu -q --maxfail=20 --timeout=900 -p no:cacheprovider 2>&1 | tail -40 ) > gpurun_out/${tag}_pytest.txt
tail -4 gpurun_out/${tag}_pytest.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 600 python bench.py > gpurun_out/${tag}_bench_c2_default.json 2> gpurun_out/${tag}_bench_c2_default.err; tail -c 600 gpurun_out/${tag}_bench_c2_default.json
timeout 300 python bench.py --steps 20 --warmup 5 > gpurun_out/${tag}_bench_c2_steps20.json 2>> gpurun_out/${tag}_err.txt
R=$PWD
cd /tmp
for wl in c2 c3 c4; do
  rocprofv3 --kernel-trace --stats -d /tmp/rocprof_kt_${tag}_$wl -o k -- python $R/bench.py --workload $wl --steps 2000 --warmup 100 --cpu-seconds 0 --no-f64 --no-closed-loop --no-variants --prewarm 0 > $R/gpurun_out/kt_${tag}_$wl.json 2> $R/gpurun_out/kt_${tag}_$wl.err
  db=$(ls /tmp/rocprof_kt_${tag}_$wl/*.db /tmp/rocprof_kt_${tag}_$wl/*/*.db 2>/dev/null | head -1)
  [ -n "$db" ] && python $R/tools/rocprof_summary.py $db "rocprofv3 --kernel-trace --stats -- python bench.py --workload $wl --steps 2000 --warmup 100 --cpu-seconds 0 --no-f64 --no-closed-loop --no-variants" > $R/gpurun_out/kt_${tag}_${wl}_stats.txt
  head -3 $R/gpurun_out/kt_${tag}_${wl}_stats.txt
done
