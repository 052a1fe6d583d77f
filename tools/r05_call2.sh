#!/bin/bash
# GPU call 2 of round 5: step-kernel A/Bs (direct stores + row skipping vs round 4's tree, the counter-load fix and the row skipping each on
# their own), phase timing of both trees, a longer C5 harness run with a larger minibatch
tag=r05b
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== A/B against round 4's tree"; date
bash tools/ab_tree.sh $tag c2 c4 2>&1 | tail -14
echo "== A/B flags"; date
bash tools/ab_flag.sh ${tag}_oldctr "-DQS_AB_OLD_CTR" c2:1024 c4:512 2>&1 | tail -8
bash tools/ab_flag.sh ${tag}_noskip "-DQS_SKIP_ROWS=0" c2:1024 c4:512 2>&1 | tail -8
echo "== phase timing"; date
timeout 200 python tools/phase_timing.py c2 > gpurun_out/${tag}_phase_c2_new.txt 2>&1; head -18 gpurun_out/${tag}_phase_c2_new.txt
( cd build_exp/old && timeout 200 python tools/phase_timing.py c2 ) > gpurun_out/${tag}_phase_c2_old.txt 2>&1; head -18 gpurun_out/${tag}_phase_c2_old.txt
echo "== gate"; date
( QS_SPEC=off timeout 600 python -m pytest tests/test_hip_parity.py -k "c2_n8_dw or c4_n32_svs or c3_n8_obst or s_mix" -m gpu -q -x --timeout=600 -p no:cacheprovider 2>&1 | tail -6 ) > gpurun_out/${tag}_gate.txt; tail -3 gpurun_out/${tag}_gate.txt
echo "== C5 harness, batch 8192"; date
timeout 400 python tools/ppo_c5.py --iterations 96 --batch_size 8192 > gpurun_out/${tag}_ppo_c5_b8192.txt 2> gpurun_out/${tag}_ppo_c5.err; tail -1 gpurun_out/${tag}_ppo_c5_b8192.txt | cut -c1-700; tail -3 gpurun_out/${tag}_ppo_c5.err
python - <<'PY'
import json
rows=[json.loads(l) for l in open("gpurun_out/r05b_ppo_c5_b8192.txt") if l.startswith("{") and "iteration" in l]
for k in range(0, len(rows), 12):
    ch = rows[k:k+12]
    if len(ch) < 12: break
    m = lambda f: sum(f(r) for r in ch) / len(ch)
    print(f"episode {k//12}: reward {m(lambda r: r['reward_mean']):.5f} pos {m(lambda r: r['terms']['rew_pos']):.5f} crash {m(lambda r: r['terms']['rew_crash']):.5f} orient {m(lambda r: r['terms']['rew_orient']):.5f} spin {m(lambda r: r['terms']['rew_spin']):.5f} std {ch[-1]['action_std'][0]} fps {ch[-1]['fps']}")
PY
echo "== bench steps20 + default short"; date
timeout 300 python bench.py --steps 20 --warmup 5 --no-c5-train --no-closed-loop --no-variants --no-f64 > gpurun_out/${tag}_bench_c2_steps20.json 2>> gpurun_out/${tag}_err.txt; python -c "
import json; d=json.loads(open('gpurun_out/${tag}_bench_c2_steps20.json').read().strip().splitlines()[-1]); print('steps20:', d['ms_per_step']*1e3, 'us', d['roofline']['frac'], d['config']['auto_reset'], d['config']['workload'])"
date
