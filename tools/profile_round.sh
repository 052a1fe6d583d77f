#!/bin/bash
# One profiling pass for profiles/: rocprofv3 --kernel-trace --stats of the default bench command + PMC passes per workload.
# usage (GPU box): bash tools/profile_round.sh <tag>
tag=$1
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for wl in c2 c3 c4; do
  rocprofv3 --kernel-trace --stats -d /tmp/rocprof_kt_${tag}_$wl -o k -- python $R/bench.py --workload $wl --steps 2000 --warmup 100 --cpu-seconds 0 --no-f64 --no-closed-loop --no-variants --prewarm 0 > $R/gpurun_out/kt_${tag}_$wl.json 2> $R/gpurun_out/kt_${tag}_$wl.err
  db=$(ls /tmp/rocprof_kt_${tag}_$wl/*.db /tmp/rocprof_kt_${tag}_$wl/*/*.db 2>/dev/null | head -1)
  [ -n "$db" ] && python $R/tools/rocprof_summary.py $db "rocprofv3 --kernel-trace --stats -- python bench.py --workload $wl --steps 2000 --warmup 100 --cpu-seconds 0 --no-f64 --no-closed-loop --no-variants" > $R/gpurun_out/kt_${tag}_${wl}_stats.txt
  cat $R/gpurun_out/kt_${tag}_${wl}_stats.txt
done
cd $R
for wl in c2 c4; do bash tools/pmc.sh ${tag}_$wl $wl | tail -3; done
bash tools/pmc.sh ${tag}_c2_E131072 c2 --envs-per-gpu 131072 | tail -3
