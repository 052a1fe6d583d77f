#!/bin/bash
# usage: tools/pmc.sh <tag> <workload> [bench args...]   (run on the GPU box through gpurun)
# rocprofv3 PMC passes of the step kernel, each in its own run with --kernel-trace only (the node pool refuses PMC together
# with other trace domains).  Writes gpurun_out/pmc_<tag>_summary.txt and gpurun_out/pmc_<tag>_traffic.json
# (per-launch means; FETCH_SIZE / WRITE_SIZE are reported in KiB by rocprofv3).  --no-f64 --no-closed-loop: the float64 line of bench.py runs a kernel
# of the same name with twice the bytes, which would be averaged into the float32 kernel's counters.
tag=$1; wl=$2; shift; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for pmc in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INSTS_SMEM SQ_INST_LEVEL_VMEM" "FETCH_SIZE" "WRITE_SIZE GRBM_GUI_ACTIVE"; do
  n=$(echo $pmc | cut -c1-14 | tr " " "_")
  rocprofv3 --kernel-trace --pmc $pmc -d /tmp/rocprof_pmc_${tag}_$n -o p --output-format csv -- python $R/bench.py --workload $wl --steps 200 --warmup 20 --profile-steps 10 --rollout-steps 0 --cpu-seconds 0 --no-f64 --no-closed-loop --no-variants --no-c5-train "$@" > $R/gpurun_out/pmc_${tag}_$n.out 2> $R/gpurun_out/pmc_${tag}_$n.err
done
python - <<PY
import csv, collections, glob, json
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for d in sorted(glob.glob("/tmp/rocprof_pmc_${tag}_*/p_counter_collection.csv")):
    for r in csv.DictReader(open(d)):
        k = r['Kernel_Name']
        if k.startswith('qs_spec_step') or k.startswith('void qs_step') or k.startswith('qs_step'):
            acc[k.split('(')[0]][r['Counter_Name']].append(float(r['Counter_Value']))
out = ["# rocprofv3 PMC summary, tag ${tag}: bench.py --workload $wl $@ (per-launch means of the step kernel)"]
traffic = {}
for k, cs in acc.items():
    out.append(f"kernel {k}")
    for c, v in sorted(cs.items()):
        out.append(f"  {c:24s} launches={len(v):4d} mean={sum(v)/len(v):16.1f}")
    if 'FETCH_SIZE' in cs and 'WRITE_SIZE' in cs:
        # calibration (tools/pmc_calib.sh, profiles/r04_pmc_calibration.txt): on gfx950 FETCH_SIZE counts exactly 1/2 of the bytes read, for the
        # stepper's 4-byte-per-lane buffer_load rows as for 16-byte streams; WRITE_SIZE is exact (1.000x, plain and nt stores)
        raw = 1024 * sum(cs['FETCH_SIZE']) / len(cs['FETCH_SIZE'])
        traffic[k] = {"fetch_bytes": 2.0 * raw, "fetch_counter_bytes": raw, "fetch_correction": 2.0, "write_bytes": 1024 * sum(cs['WRITE_SIZE']) / len(cs['WRITE_SIZE'])}
open("$R/gpurun_out/pmc_${tag}_summary.txt", "w").write("\n".join(out) + "\n")
json.dump(traffic, open("$R/gpurun_out/pmc_${tag}_traffic.json", "w"), indent=1)
print("\n".join(out)); print(json.dumps(traffic))
PY
