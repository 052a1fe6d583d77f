#!/bin/bash
# One rocprofv3 PMC pass (wait / active breakdown) over the fused encoder kernels; usage (GPU box): bash tools/pmc_encoder.sh <tag> <mean|attention|mha>
tag=$1; v=$2
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_LEVEL_VMEM -d /tmp/rocprof_pmcenc_${tag}_$v -o p --output-format csv -- python $R/tools/bench_encoder.py 8192 $v > $R/gpurun_out/pmcenc_${tag}_$v.out 2> $R/gpurun_out/pmcenc_${tag}_$v.err
python - <<PY
import csv, collections, glob
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for d in sorted(glob.glob("/tmp/rocprof_pmcenc_${tag}_$v/**/p_counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(d)):
        k = r['Kernel_Name']
        if k.startswith('qs_encoder'):
            acc[k.split('(')[0]][r['Counter_Name']].append(float(r['Counter_Value']))
out = ["# rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_LEVEL_VMEM -- python tools/bench_encoder.py 8192 $v (per-launch means)"]
for k, cs in acc.items():
    out.append(f"kernel {k}")
    for c, vv in sorted(cs.items()):
        out.append(f"  {c:24s} launches={len(vv):4d} mean={sum(vv)/len(vv):16.1f}")
open("$R/gpurun_out/pmcenc_${tag}_${v}_summary.txt", "w").write("\n".join(out) + "\n")
print("\n".join(out))
PY
