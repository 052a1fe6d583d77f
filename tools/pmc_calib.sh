#!/bin/bash
# usage: tools/pmc_calib.sh <tag>   (run on the GPU box through gpurun)
# FETCH_SIZE / WRITE_SIZE of rocprofv3 against KNOWN byte counts in the stepper's own access pattern (tools/ubench_hbm.hip `calib`):
# separate --pmc passes with --kernel-trace only.  Writes gpurun_out/pmc_calib_<tag>.txt / .json: counted bytes, known bytes, ratio per kernel.
tag=$1
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
[ -x $R/tools/ubench_hbm ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o $R/tools/ubench_hbm $R/tools/ubench_hbm.hip
for pmc in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $pmc -d /tmp/rocprof_calib_${tag}_$pmc -o p --output-format csv -- $R/tools/ubench_hbm calib > $R/gpurun_out/pmc_calib_${tag}_$pmc.out 2> $R/gpurun_out/pmc_calib_${tag}_$pmc.err
done
python - <<PY
import csv, collections, glob, json
T = 1 << 22
known = {"k_calib_read4<47>": (47 * T * 4, 0), "k_calib_read4<8>": (8 * T * 4, 0), "k_calib_write4<54>": (0, 54 * T * 4), "k_calib_write4<8>": (0, 8 * T * 4),
         "k_calib_write16nt": (0, 54 * T * 4), "k_calib_read16": (47 * T * 4, 0), "k_calib_write16": (0, 54 * T * 4)}
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for d in sorted(glob.glob("/tmp/rocprof_calib_${tag}_*/p_counter_collection.csv")):
    for r in csv.DictReader(open(d)):
        k = r['Kernel_Name'].replace('void ', '').split('(')[0]
        acc[k][r['Counter_Name']].append(float(r['Counter_Value']))
out, rec = ["# rocprofv3 FETCH_SIZE / WRITE_SIZE (KiB x 1024) vs the bytes the kernel is KNOWN to move; tools/ubench_hbm calib, 2^22 lanes, sets > 512 MB"], {}
for k, (rb, wb) in known.items():
    cs = acc.get(k, {})
    f = 1024 * sum(cs.get('FETCH_SIZE', [0])) / max(len(cs.get('FETCH_SIZE', [])), 1)
    w = 1024 * sum(cs.get('WRITE_SIZE', [0])) / max(len(cs.get('WRITE_SIZE', [])), 1)
    rec[k] = {"known_read": rb, "known_write": wb, "FETCH_SIZE_bytes": f, "WRITE_SIZE_bytes": w,
              "fetch_over_known": (f / rb if rb else None), "write_over_known": (w / wb if wb else None)}
    out.append(f"{k:22s} known read {rb:12d} write {wb:12d} | FETCH_SIZE {f:14.0f} ({(f / rb if rb else float('nan')):5.3f}x)  WRITE_SIZE {w:14.0f} ({(w / wb if wb else float('nan')):5.3f}x)")
out.append("# kernels seen: " + ", ".join(sorted(acc)))
open("$R/gpurun_out/pmc_calib_${tag}.txt", "w").write("\n".join(out) + "\n")
json.dump(rec, open("$R/gpurun_out/pmc_calib_${tag}.json", "w"), indent=1)
print("\n".join(out))
PY
