#!/bin/bash
# round 3, GPU call H: scenario.step() rewrites done by the env's lanes in the full-scenario kernels (dynamic_formations every step, the
# others at their periods): parity of every scenario, then what the `mix` batch costs per step now, per scenario, and that the
# headline shapes did not move.
tag=${1:-r03h}
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_hip_parity.py tests/test_fp32_parity_gpu.py tests/test_soak_gpu.py tests/test_facade_gpu.py tests/test_sf_protocol_gpu.py -m gpu -q --maxfail=20 --timeout=600 -p no:cacheprovider 2>&1 | tail -60 ) > gpurun_out/${tag}_pytest_scen.txt
tail -5 gpurun_out/${tag}_pytest_scen.txt
timeout 600 python tools/scenario_times.py 1024 1200 > gpurun_out/${tag}_scenario_times.txt 2>&1; cat gpurun_out/${tag}_scenario_times.txt | tail -20
timeout 600 python tools/bench_batched_env.py > gpurun_out/${tag}_batched_env_host.json 2>gpurun_out/${tag}_batched_env_host.err; cat gpurun_out/${tag}_batched_env_host.json
fmt='import json,sys
for line in sys.stdin:
    if not line.startswith("{"): continue
    d=json.loads(line)
    r=d["roofline"]
    print(sys.argv[1], "|", round(d["value"]/1e9,3), "G env-steps/s  ms_per_step", round(d["ms_per_step"]*1e3,2), "us  kernel_us", round(r["kernel_avg_us"],2), "frac", round(r["frac"],4), r["kernel_flavor"])'
Q="--cpu-seconds 0 --no-f64 --no-closed-loop --no-variants --rollout-steps 0 --profile-steps 0"
out=gpurun_out/${tag}_lines.txt; : > $out
for wl in c2 c3 c4; do timeout 300 python bench.py --workload $wl --steps 2000 --warmup 100 $Q 2>>gpurun_out/${tag}_err.txt | python -c "$fmt" "$wl" | tee -a $out; done
timeout 300 python bench.py --workload c4 --envs-per-gpu 32768 --steps 600 --warmup 100 $Q 2>>gpurun_out/${tag}_err.txt | python -c "$fmt" "c4 E=32768" | tee -a $out
( timeout 1500 python -m pytest tests -m gpu -q --maxfail=20 --timeout=900 -p no:cacheprovider --deselect tests/test_hip_parity.py --deselect tests/test_fp32_parity_gpu.py --deselect tests/test_soak_gpu.py --deselect tests/test_facade_gpu.py --deselect tests/test_sf_protocol_gpu.py 2>&1 | tail -40 ) > gpurun_out/${tag}_pytest_rest.txt
tail -4 gpurun_out/${tag}_pytest_rest.txt
tail -5 gpurun_out/${tag}_err.txt
