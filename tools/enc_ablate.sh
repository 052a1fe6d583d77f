#!/bin/bash
# timing ablations of the fused encoder (GPU box): every build_exp/libenc_<variant>.so (built by hand with -DENC_EXP_* flags) on
# the attention / mean_embed / multi-head encoders at 8192 agents
for lib in build_exp/libenc_*.so; do
  for m in ${ENC_MODELS:-attention mean_embed mha}; do
    QS_ENC_LIB=$PWD/$lib python tools/bench_encoder.py ${1:-8192} $m 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$lib'.split('libenc_')[1][:-3].ljust(28), '$m'.ljust(10), round(d['fused_us'],1), 'us', round(d['frac_of_bf16_mfma_peak'],3))"
  done
done
