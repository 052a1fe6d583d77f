for wl in c2 c3; do
  for v in shared own; do
    if [ $v = own ]; then export QS_TIMING_EXTRA="-DQS_AB_OWN_CTR"; else unset QS_TIMING_EXTRA; fi
    timeout 300 python tools/phase_timing.py $wl > gpurun_out/r05r_phase_${wl}_${v}.txt 2>&1
    echo "== $wl $v"; grep -v "amdgpu.ids" gpurun_out/r05r_phase_${wl}_${v}.txt | head -24
  done
done
