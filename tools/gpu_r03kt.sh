#!/bin/bash
# rocprofv3 kernel stats of the three bench shapes on the final tree (no scratch pre-warm inside the traced process)
tag=${1:-r03kt}
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$PWD
cd /tmp
for wl in c2 c3 c4; do
  rocprofv3 --kernel-trace --stats -d /tmp/rocprof_kt_${tag}_$wl -o k -- python $R/bench.py --workload $wl --steps 2000 --warmup 100 --cpu-seconds 0 --no-f64 --no-closed-loop --no-variants --prewarm 0 > $R/gpurun_out/kt_${tag}_$wl.json 2> $R/gpurun_out/kt_${tag}_$wl.err
  db=$(ls /tmp/rocprof_kt_${tag}_$wl/*.db /tmp/rocprof_kt_${tag}_$wl/*/*.db 2>/dev/null | head -1)
  [ -n "$db" ] && python $R/tools/rocprof_summary.py $db "rocprofv3 --kernel-trace --stats -- python bench.py --workload $wl --steps 2000 --warmup 100 --cpu-seconds 0 --no-f64 --no-closed-loop --no-variants --prewarm 0" > $R/gpurun_out/kt_${tag}_${wl}_stats.txt
  head -3 $R/gpurun_out/kt_${tag}_${wl}_stats.txt
  python -c "
import json,sys
for l in open('$R/gpurun_out/kt_${tag}_$wl.json'):
    if l.startswith('{'):
        d=json.loads(l); print('$wl events:', round(d['roofline']['kernel_avg_us'],3), 'frac', round(d['roofline']['frac'],4))"
done
