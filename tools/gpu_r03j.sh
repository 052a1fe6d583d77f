#!/bin/bash
# round 3, GPU call J: full-scenario kernels with the per-episode scenario code inlined in the specialised objects (constant block out
# of scratch): parity of every scenario, step time per scenario, the batched SF env on `mix`.
tag=${1:-r03j}
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_hip_parity.py tests/test_fp32_parity_gpu.py tests/test_soak_gpu.py tests/test_facade_gpu.py tests/test_sf_protocol_gpu.py tests/test_replay_gpu.py tests/test_rollout_gpu.py -m gpu -q --maxfail=20 --timeout=600 -p no:cacheprovider 2>&1 | tail -60 ) > gpurun_out/${tag}_pytest_scen.txt
tail -5 gpurun_out/${tag}_pytest_scen.txt
timeout 600 python tools/scenario_times.py 1024 1200 > gpurun_out/${tag}_scenario_times.txt 2>&1; tail -18 gpurun_out/${tag}_scenario_times.txt
timeout 600 python tools/bench_batched_env.py > gpurun_out/${tag}_batched_env_host.json 2>gpurun_out/${tag}_batched_env_host.err; cat gpurun_out/${tag}_batched_env_host.json
QS_WG_WARM=1200 timeout 300 python tools/wg_times.py c2 "quads_mode='mix'" > gpurun_out/${tag}_wg_mix_steady.txt 2>&1; tail -20 gpurun_out/${tag}_wg_mix_steady.txt
