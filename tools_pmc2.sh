cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for pmc in "SQ_INSTS_SMEM SQ_INST_LEVEL_SMEM SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY" "SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY"; do
  n=$(echo $pmc | cut -c1-14 | tr " " "_")
  rocprofv3 --kernel-trace --pmc $pmc -d $R/gpurun_out/pmc_v3_$n -o p --output-format csv -- python $R/bench.py --steps 200 --warmup 20 --profile-steps 10 --cpu-seconds 0 > /dev/null 2> $R/gpurun_out/pmc_v3_$n.err
done
python - <<PY
import csv, collections, glob
for d in sorted(glob.glob("$R/gpurun_out/pmc_v3_*/p_counter_collection.csv")):
    rows=list(csv.DictReader(open(d)))
    acc=collections.defaultdict(list)
    for r in rows:
        if 'qs_step_kernel' in r['Kernel_Name']:
            acc[r['Counter_Name']].append(float(r['Counter_Value']))
    for k,v in sorted(acc.items()):
        print(f'{k:28s} launches={len(v):4d} mean/wave={sum(v)/len(v)/128:12.1f}')
PY
