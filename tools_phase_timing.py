#!/usr/bin/env python
"""Per-phase cycle breakdown of the step kernel (workgroup 0), using a -DQS_TIMING build (s_memtime stamps)."""
import ctypes as C
import os
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
from quad_swarm_rl_amd import config as qcfg, native

lib_t = os.path.join(native.CSRC, "libquadswarm_hip_timing.so")
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-DQS_TIMING",
                       "-o", lib_t, native.SOURCES[0]])
native.LIB_PATH = lib_t
import bench
args = sys.argv[1:]
wl = args[0] if args else "c2"
kw = dict(bench.WORKLOADS[wl]["kw"])
import ast
for item in args[1:]:
    k, v = item.split("=", 1)
    kw[k] = ast.literal_eval(v)
E = bench.WORKLOADS[wl]["num_envs"]
cfg = qcfg.make_config(num_envs=E, seed=0, write_rew_info=False, **kw)
st = native.Stepper(cfg)
L = native.lib()
L.qs_debug_timing.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong)]
st.reset()
rng = np.random.RandomState(0)
names = ["loads-issue", "ou-rng", "2 substeps", "reward", "self-obs", "publish+pairscan", "ballots/reward2", "downwash", "responses/scen",
         "final obs+nbr", "reset-check", "barrier", "obs copy-out", "state stores"]
acc = np.zeros(13)
n = 0
for t in range(60):
    st.from_host("actions", rng.uniform(-1, 1, size=(st.T, 4)))
    st.step()
    st.sync()
    buf = (C.c_ulonglong * 32)()
    L.qs_debug_timing(st._h, buf)
    ts = np.array(buf[:14], dtype=np.float64)
    extra = np.array(buf[14:17], dtype=np.float64)
    if t >= 10:
        acc += np.diff(ts)
        ex = ex + np.array([extra[0] - ts[9], extra[1] - extra[0], extra[2] - extra[1], ts[10] - extra[2]]) if t > 10 else np.array([extra[0] - ts[9], extra[1] - extra[0], extra[2] - extra[1], ts[10] - extra[2]])
        n += 1
acc /= n
print(f"workload {wl} {args[1:]}: per-phase shader cycles (workgroup 0, lane 0), total {acc.sum():.0f}")
for nm, v in zip(names, acc):
    print(f"  {nm:22s} {v:9.0f}")
print("  final-obs split: publish vel %.0f | refresh self-obs %.0f | barrier %.0f | neighbour+sdf obs %.0f" % tuple(ex / n))
