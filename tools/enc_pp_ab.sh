#!/bin/bash
# usage (GPU box): bash tools/enc_pp_ab.sh <tag>   the 32-agent mean_embed encoder kernels: tests, same-box A/B of the ping-pong against the lock-step
# schedule (QS_ENC_PP=1 / 0, tools/bench_encoder.py), phase stamps of both (build_exp/libenc_timing.so = the library built with -DENC_TIMING, if present)
tag=$1
out=gpurun_out/${tag}_enc_pp_ab.txt
( timeout 900 python -m pytest tests/test_policy_encoder_gpu.py -m gpu -q -x -k "pingpong or wide_workgroup" --timeout=600 -p no:cacheprovider 2>&1 | tail -15 ) > gpurun_out/${tag}_enc_pp_pytest.txt
tail -5 gpurun_out/${tag}_enc_pp_pytest.txt
echo "# tools/bench_encoder.py <agents>: mean_embed, bf16, us per forward (HIP events, back-to-back launches); QS_ENC_PP=1 ping-pong / 0 lock-step; same box, interleaved" > $out
for rep in 1 2; do for B in 8192 131072 4096; do for pp in 1 0; do
  QS_ENC_PP=$pp timeout 300 python tools/bench_encoder.py $B 2>>gpurun_out/${tag}_err.txt | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('agents $B rep $rep pp=$pp: %.2f us  frac %.3f' % (d['fused_us'], d['frac_of_bf16_mfma_peak']))" | tee -a $out
done; done; done
if [ -f build_exp/libenc_timing.so ]; then
  for pp in 1 0; do echo "== QS_ENC_PP=$pp"; QS_ENC_PP=$pp QS_ENC_LIB=$PWD/build_exp/libenc_timing.so python tools/enc_stamps.py 8192 2>&1 | grep -v amdgpu.ids; done > gpurun_out/${tag}_enc_stamps.txt; cat gpurun_out/${tag}_enc_stamps.txt
fi
