#!/bin/bash
# usage: tools/ab_flags.sh <tag> "<shapes>" "<flags of variant 1>" ["<flags of variant 2>" ...]   (GPU box, through gpurun)
# Same-box comparison of compile-time switches of the config-specialised kernels (QS_SPEC_EXTRA_FLAGS is part of the cache key): variant 0 is
# the default build; the variants alternate, twice each, per shape (workload:envs).  us per step = HIP events over bench.py's timed region.
tag=$1; shapes=$2; shift; shift
mkdir -p gpurun_out
out=gpurun_out/${tag}_ab.txt
echo "# variant 0 = default build; variant k = QS_SPEC_EXTRA_FLAGS of argument k; bench.py --steps 400 --warmup 50, us per step" > $out
k=1; for f in "$@"; do echo "# variant $k: $f" >> $out; k=$((k+1)); done
for we in $shapes; do
  wl=${we%%:*}; E=${we##*:}
  for rep in 1 2; do
    k=0
    for f in "" "$@"; do
      if [ -n "$f" ]; then export QS_SPEC_EXTRA_FLAGS="$f"; else unset QS_SPEC_EXTRA_FLAGS; fi
      us=$(timeout 300 python bench.py --workload $wl --envs-per-gpu $E --steps 400 --warmup 50 --prewarm 200 --rollout-steps 0 --cpu-seconds 0 --no-f64 --no-closed-loop --no-variants --no-c5-train 2>>gpurun_out/${tag}_ab.err | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('%.3f frac %.4f' % (1e3*d['ms_per_step'], d['roofline']['frac']))")
      echo "$wl E=$E rep $rep variant $k: $us" | tee -a $out
      k=$((k+1))
    done
  done
done
unset QS_SPEC_EXTRA_FLAGS
