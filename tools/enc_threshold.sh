#!/bin/bash
# 16-agent vs 32-agent workgroup kernels of the fused encoder across batch sizes (GPU box): where does the wide variant start to pay?
for m in mean_embed attention; do
  for b in 1024 2048 3072 4096 5120 6144 8192 16384; do
    for w in 0 1; do
      QS_ENC_WIDE_MIN=$w python tools/bench_encoder.py $b $m 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$m', 'B=$b', 'wide' if $w else 'narrow', round(d['fused_us'],1), 'us')"
    done
  done
done
