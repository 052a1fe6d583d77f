#!/bin/bash
tag=${1:-r02d}
mkdir -p gpurun_out
( timeout 600 python -m pytest tests/test_hip_vs_reference.py -q 2>&1 | tail -70 ) > gpurun_out/${tag}_tape_all.txt; tail -70 gpurun_out/${tag}_tape_all.txt
./tools/ubench_hbm > gpurun_out/${tag}_ubench_hbm.txt 2>&1; cat gpurun_out/${tag}_ubench_hbm.txt
( timeout 900 python -m pytest tests -m gpu -q -x --deselect tests/test_hip_vs_reference.py 2>&1 | tail -5 ) > gpurun_out/${tag}_pytest.txt; tail -5 gpurun_out/${tag}_pytest.txt
bash tools/pmc.sh ${tag}_c2_E131072 c2 --envs-per-gpu 131072 --no-f64 --no-closed-loop | tail -24
for wl in c2 c4; do timeout 300 python tools/phase_timing.py $wl > gpurun_out/${tag}_phase_$wl.txt 2>&1; tail -40 gpurun_out/${tag}_phase_$wl.txt; done
cd /tmp && rocprofv3 --list-avail > $GRAFT_REPO_ROOT/gpurun_out/${tag}_counters_avail.txt 2>&1; cd $GRAFT_REPO_ROOT
