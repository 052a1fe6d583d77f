#!/bin/bash
# GPU call 6 of round 5: the full-size oracle comparison, the stdout order of the --force-gather line
tag=r05f
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== full size against the oracle"; date
( timeout 900 python -m pytest tests/test_hip_parity.py -k "full_size" -m gpu -q --durations=8 --timeout=900 -p no:cacheprovider 2>&1 | tail -16 ) > gpurun_out/${tag}_full_size.txt; tail -14 gpurun_out/${tag}_full_size.txt
echo "== gather line order"; date
timeout 300 python bench.py --workload c4 --force-gather --steps 200 --warmup 20 --no-wire-sweep --cpu-seconds 0 --no-f64 --no-closed-loop --no-variants --no-c5-train > gpurun_out/${tag}_gather.out 2> gpurun_out/${tag}_gather.err; echo "rc=$?"; tail -n 3 gpurun_out/${tag}_gather.out | cut -c1-120
date
