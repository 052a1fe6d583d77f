#!/bin/bash
# usage: tools/ab_flag.sh <tag> "<extra hipcc flags of variant B>" [workload:envs ...]   (GPU box, through gpurun)
# Same-box A/B of a compile-time switch of the config-specialised kernels (QS_SPEC_EXTRA_FLAGS is part of the cache key): the two
# variants alternate, twice each, per shape.  Prints us per step (HIP events over bench.py's timed region).  AB_B_ENV="NAME=value": an environment
# variable set for variant B only (e.g. QS_PERSIST=16 next to -DQS_PERSIST_LOOP).
tag=$1; flags=$2; shift; shift
shapes=${@:-"c2:131072 c3:131072 c4:32768"}
mkdir -p gpurun_out
out=gpurun_out/${tag}_ab.txt
echo "# A = default build, B = QS_SPEC_EXTRA_FLAGS='$flags'; bench.py --steps 400 --warmup 50, us per step" > $out
for we in $shapes; do
  wl=${we%%:*}; E=${we##*:}
  for rep in 1 2; do
    for v in A B; do
      if [ $v = B ]; then export QS_SPEC_EXTRA_FLAGS="$flags"; [ -n "$AB_B_ENV" ] && export $AB_B_ENV; else unset QS_SPEC_EXTRA_FLAGS; [ -n "$AB_B_ENV" ] && unset ${AB_B_ENV%%=*}; fi
      us=$(timeout 300 python bench.py --workload $wl --envs-per-gpu $E --steps 400 --warmup 50 --prewarm 200 --rollout-steps 0 --cpu-seconds 0 --no-f64 --no-closed-loop --no-variants --no-c5-train 2>>gpurun_out/${tag}_ab.err | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('%.3f frac %.4f' % (1e3*d['ms_per_step'], d['roofline']['frac']))")
      echo "$wl E=$E rep $rep variant $v: $us" | tee -a $out
    done
  done
done
unset QS_SPEC_EXTRA_FLAGS
