#!/bin/bash
# rocprofv3 --kernel-trace --stats + bench lines of the fused policy encoder (both neighbour-encoder variants) for profiles/.
# usage (GPU box): bash tools/profile_encoder.sh <tag>
tag=$1
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for v in mean attention mha; do
  rocprofv3 --kernel-trace --stats -d /tmp/rocprof_enc_${tag}_$v -o k -- python $R/tools/bench_encoder.py 8192 $v > $R/gpurun_out/enc_${tag}_${v}_prof.json 2> $R/gpurun_out/enc_${tag}_$v.err
  db=$(ls /tmp/rocprof_enc_${tag}_$v/*.db /tmp/rocprof_enc_${tag}_$v/*/*.db 2>/dev/null | head -1)
  [ -n "$db" ] && python $R/tools/rocprof_summary.py $db "rocprofv3 --kernel-trace --stats -- python tools/bench_encoder.py 8192 $v" | head -12 > $R/gpurun_out/enc_${tag}_${v}_stats.txt
  cat $R/gpurun_out/enc_${tag}_${v}_stats.txt
  for b in 8192 131072; do python $R/tools/bench_encoder.py $b $v 2>/dev/null | tail -1 > $R/gpurun_out/enc_${tag}_${v}_bench_$b.json; cat $R/gpurun_out/enc_${tag}_${v}_bench_$b.json | cut -c1-300; done
done
