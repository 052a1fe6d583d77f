#!/usr/bin/env python
"""Per-phase cycle breakdown of the step kernel (workgroup 0), using a -DQS_TIMING build (s_memtime stamps)."""
import ctypes as C
import os
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from quad_swarm_rl_amd import config as qcfg, native

# the phase stamps are compiled into the config-specialised code object (QS_SPEC=jit is the default)
os.environ["QS_SPEC_EXTRA_FLAGS"] = "-DQS_TIMING " + os.environ.get("QS_TIMING_EXTRA", "")
os.environ.setdefault("QS_SPEC", "jit")
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
team = os.environ.get("QS_TEAM", "1") != "0" and st.T // cfg.num_agents <= 1024 * (64 // cfg.num_agents)
assert st.specialized
if team:
    order = [0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 14, 15, 10, 11, 12, 13]
    names = ["loads+ou-rng", "ou update", "2 substeps", "publish+reward", "wait barrier 1", "pair share (+wait barrier 2)", "combine/ballots/reward2",
             "downwash", "responses/scen", "publish vel + barrier 3", "metrics + barrier 4", "ranks/nbr rows + barrier 5", "reset-check",
             "barrier 6 + obs copy-out", "outputs + state stores"]
else:
    order = [0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 14, 15, 16, 10, 11, 12, 13]
    names = ["loads+rng", "ou update", "2 substeps", "reward", "self-obs", "publish+pairscan", "ballots/reward2", "downwash", "responses/scen",
             "publish vel", "refresh self-obs", "barrier", "neighbour+sdf obs", "reset-check", "barrier + obs copy-out", "outputs + state stores"]
acc = np.zeros(len(order) - 1)
n = 0
for t in range(60):
    st.from_host("actions", rng.uniform(-1, 1, size=(st.T, 4)))
    st.step()
    st.sync()
    buf = (C.c_ulonglong * 128)()
    L.qs_debug_timing(st._h, buf)
    ts = np.array([buf[k] for k in order], dtype=np.float64)
    if t >= 10:
        acc += np.diff(ts)
        n += 1
        if team:   # helper waves: arrival at / release from each barrier relative to wave 0's kernel entry
            hw = np.array([[buf[32 * w + k] for k in range(18)] for w in (1, 2, 3)], dtype=np.float64) - float(buf[0])
            w0 = np.array([buf[k] for k in (0, 4, 5, 6, 9, 14, 14, 15, 15, 10, 10, 11, 12, 12)], dtype=np.float64) - float(buf[0])
            hacc = hw if t == 10 else hacc + hw
            w0acc = w0 if t == 10 else w0acc + w0
acc /= n
print(f"workload {wl} {args[1:]} team={team}: per-phase s_memtime ticks (workgroup 0, wave 0 lane 0), total {acc.sum():.0f}")
for nm, v in zip(names, acc):
    print(f"  {nm:32s} {v:9.0f}")
if team:
    print("helper waves (ticks since wave 0 entered the kernel): stamp = entry, [arrive, leave] x barriers 1..6, copy-out done")
    print("  wave 0 (nearest stamps)", np.round(w0acc / n).astype(int).tolist())
    for w in range(3):
        print(f"  wave {w + 1}", np.round(hacc[w] / n).astype(int).tolist())
