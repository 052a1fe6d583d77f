#!/bin/bash
# usage: tools_pmc.sh <tag> [bench args...]   (run on the GPU box through gpurun); PMC passes are separate runs with
# --kernel-trace only, as the node pool requires.
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for pmc in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM" "FETCH_SIZE" "WRITE_SIZE GRBM_GUI_ACTIVE"; do
  n=$(echo $pmc | cut -c1-14 | tr " " "_")
  rocprofv3 --kernel-trace --pmc $pmc -d $R/gpurun_out/pmc_${tag}_$n -o p --output-format csv -- python $R/bench.py --steps 200 --warmup 20 --profile-steps 10 --cpu-seconds 0 "$@" > /dev/null 2> $R/gpurun_out/pmc_${tag}_$n.err
done
python - <<PY
import csv, collections, glob
out=["# PMC summary of qs_step_kernel, tag ${tag}, args: $@ (per launch means)"]
for d in sorted(glob.glob("$R/gpurun_out/pmc_${tag}_*/p_counter_collection.csv")):
    rows=list(csv.DictReader(open(d)))
    acc=collections.defaultdict(list)
    for r in rows:
        if 'qs_step_kernel' in r['Kernel_Name']:
            acc[r['Counter_Name']].append(float(r['Counter_Value']))
    for k,v in sorted(acc.items()):
        out.append(f'{k:28s} launches={len(v):4d} mean={sum(v)/len(v):16.1f}')
open("$R/gpurun_out/pmc_${tag}_summary.txt","w").write("\n".join(out)+"\n")
print("\n".join(out))
PY
