#!/bin/bash
# GPU call 7 of round 5: SDF with one square root per cell - parity of every obstacle case, same-box A/B against round 4's tree, the C3 shapes
tag=r05g
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== obstacle cases"; date
( timeout 900 python -m pytest tests/test_hip_parity.py tests/test_hip_vs_reference.py tests/test_hip_vs_reference_f32.py tests/test_fp32_parity_gpu.py -k "obst or o_ or mix or domain or full_size or c3" -m gpu -q --timeout=900 -p no:cacheprovider 2>&1 | tail -8 ) > gpurun_out/${tag}_obst_cases.txt; tail -4 gpurun_out/${tag}_obst_cases.txt
echo "== A/B against round 4's tree"; date
bash tools/ab_tree.sh $tag c3 c2 2>&1 | tail -14
echo "== lines"; date
bash tools/gpu.sh $tag lines 2>&1 | tail -8
date
