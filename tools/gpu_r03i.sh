#!/bin/bash
# round 3, GPU call I: step time per scenario at 1024 x 8 and the per-workgroup phases of the `mix` launch.
tag=${1:-r03i}
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python tools/scenario_times.py 1024 1200 > gpurun_out/${tag}_scenario_times.txt 2>&1; tail -20 gpurun_out/${tag}_scenario_times.txt
MIXKW=""
QS_WG_WARM=1200 timeout 300 python tools/wg_times.py c2 "quads_mode='mix'" $MIXKW > gpurun_out/${tag}_wg_mix_steady.txt 2>&1; tail -22 gpurun_out/${tag}_wg_mix_steady.txt
QS_WG_WARM=1200 timeout 300 python tools/wg_times.py c2 "quads_mode='dynamic_formations'" $MIXKW > gpurun_out/${tag}_wg_dynform_steady.txt 2>&1; tail -22 gpurun_out/${tag}_wg_dynform_steady.txt
QS_WG_WARM=1200 timeout 300 python tools/wg_times.py c2 "quads_mode='ep_lissajous3D'" $MIXKW > gpurun_out/${tag}_wg_liss_steady.txt 2>&1; tail -22 gpurun_out/${tag}_wg_liss_steady.txt
