#!/usr/bin/env python
"""Where and when every workgroup of the step kernel ran (-DQS_TIMING build): start / end of wave 0 of each workgroup on the shader clock
(s_memtime) and on the constant 100 MHz wall clock, plus HW_ID / XCC_ID.  Answers: do all workgroups of a 256-workgroup launch run at
once, one per CU?  how long does a workgroup take compared with the kernel?  what does a shader-clock tick last?
usage (GPU box): python tools/wg_times.py c4 [KEY=VALUE ...]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
os.environ["QS_SPEC_EXTRA_FLAGS"] = "-DQS_TIMING " + os.environ.get("QS_TIMING_EXTRA", "")
os.environ.setdefault("QS_SPEC", "jit")
import bench
from quad_swarm_rl_amd import config as qcfg, native
import ast
args = sys.argv[1:]
wl = args[0] if args else "c4"
kw = dict(bench.WORKLOADS[wl]["kw"])
E = bench.WORKLOADS[wl]["num_envs"]
for item in args[1:]:
    k, v = item.split("=", 1)
    if k == "num_envs":
        E = int(v)
    else:
        kw[k] = ast.literal_eval(v)
cfg = qcfg.make_config(num_envs=E, seed=0, write_rew_info=False, **kw)
st = native.Stepper(cfg)
assert st.specialized
L = native.lib()
L.qs_debug_wg_times.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong), C.c_int32]
st.reset()
rng = np.random.RandomState(0)
blocks = (E + (64 // cfg.num_agents) - 1) // (64 // cfg.num_agents)
warm = int(os.environ.get("QS_WG_WARM", "0"))          # control steps before the measured ones (steady state of a random-action rollout: ~1000)
if warm:
    import torch
    acts = (torch.rand((64, st.T, 4), device="cuda") * 2 - 1).contiguous()
    for t in range(warm):
        st.step(acts.data_ptr() + (t % 64) * st.T * 16)
    st.sync()
rows = []
S = 16
GATED = int(os.environ.get("QS_WG_GATED", "0"))   # > 0: resident-state launches of this many control steps (qs_step_gated, producer running ahead)
if GATED:
    import torch
    tab = (torch.rand((64, st.T, 4), device="cuda") * 2 - 1).contiguous()
    st.gate_create(ring_len=64, wg_per_group=8)
    side, feed = torch.cuda.Stream(), torch.cuda.Stream()
    per_wg, spans, skews = [], [], []
    for rep in range(12):
        st.step_gated(GATED, stream=side); st.gate_produce(tab.data_ptr(), 64, GATED, False, stream=feed); st.gate_wait(stream=side)
        torch.cuda.synchronize()
        buf = (C.c_ulonglong * (S * blocks))()
        n = L.qs_debug_wg_times(st._h, buf, blocks)
        a = np.array(buf[:S * n], dtype=np.uint64).reshape(n, S).astype(np.int64)
        if rep >= 2:
            a = a[(a[:, 3] & 0xf) < 8]   # (a workgroup whose stamps were not written reads as garbage: XCC ids are 0..7)
            per_wg.append(np.median((a[:, 5] - a[:, 4]) * 10.0 / GATED)); spans.append((a[:, 5].max() - a[:, 4].min()) * 10.0 / GATED); skews.append((a[:, 4].max() - a[:, 4].min()) * 10.0)
    per_wg = np.array(per_wg)
    print(f"workload {wl}: resident-state launches of {GATED} control steps, {n} workgroups x {st.waves_per_workgroup} waves (producer ahead)")
    print(f"  per control step: workgroup wall time median {np.median(per_wg) / 1e3:.3f} us (min {per_wg.min() / 1e3:.3f}, max {per_wg.max() / 1e3:.3f}); "
          f"first start -> last end of the launch / steps: median {np.median(spans) / 1e3:.3f} us")
    print(f"  start skew of the launch (last workgroup's start after the first): median {np.median(skews) / 1e3:.2f} us ONCE per {GATED} steps = {np.median(skews) / 1e3 / GATED:.3f} us per step; "
          f"state loads / stores: once per launch")
    sys.exit(0)
for t in range(60):
    st.from_host("actions", rng.uniform(-1, 1, size=(st.T, 4)))
    st.step()
    st.sync()
    buf = (C.c_ulonglong * (S * blocks))()
    n = L.qs_debug_wg_times(st._h, buf, blocks)
    a = np.array(buf[:S * n], dtype=np.uint64).reshape(n, S).astype(np.int64)
    if t >= 10:
        rows.append(a)
a = rows[-1]
dur = np.stack([r[:, 1] - r[:, 0] for r in rows]).astype(np.float64)           # shader-clock ticks per workgroup
wall = np.stack([r[:, 5] - r[:, 4] for r in rows]).astype(np.float64) * 10.0   # ns per workgroup (100 MHz)
start_spread = np.stack([(r[:, 4] - r[:, 4].min()) for r in rows]).astype(np.float64) * 10.0
span = np.array([(r[:, 5].max() - r[:, 4].min()) * 10.0 for r in rows])
hw, xcc = a[:, 2], a[:, 3] & 0xf
cu_key = (xcc << 16) | (((hw >> 13) & 7) << 8) | (((hw >> 12) & 1) << 4) | ((hw >> 8) & 0xf)
uniq, counts = np.unique(cu_key, return_counts=True)
print(f"workload {wl} {args[1:]}: {n} workgroups x {st.waves_per_workgroup} waves, kernel {st.kernel_name}")
print(f"  workgroup duration: shader clock ticks median {np.median(dur):.0f} (min {dur.min():.0f}, max {dur.max():.0f}); wall median {np.median(wall) / 1e3:.2f} us "
      f"(min {wall.min() / 1e3:.2f}, max {wall.max() / 1e3:.2f}) => {np.median(dur) / np.median(wall):.2f} ticks per ns")
print(f"  first start -> last end: median {np.median(span) / 1e3:.2f} us; start of the last workgroup after the first: median {np.median(start_spread.max(axis=1)) / 1e3:.2f} us")
print(f"  placement (last step): {len(uniq)} distinct (xcc, se, sh, cu) for {n} workgroups; workgroups per CU histogram: "
      f"{dict(zip(*np.unique(counts, return_counts=True)))}; per XCC: {dict(zip(*np.unique(xcc, return_counts=True)))}")
slow = np.argsort(-wall[-1])[:8]
# phases of wave 0 (shader clock ticks): start | sub-steps done | barrier 1 | barrier 2 | bookkeeping | responses | barrier 3 | barrier 4 | barrier 5 | reset | copy-out | end
names = ["loads+substeps", "publish+b1", "pair scan+b2", "bookkeeping", "downwash+responses+scenario", "publish vel+b3", "metrics+b4", "rank/rows+b5", "reset check/tail", "b6+copy-out", "outputs+stores"]
idx = [0, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 1]
ph_med = np.zeros(len(names)); ph_slow = np.zeros(len(names)); ph_max = np.zeros(len(names))
for r in rows:
    ph = np.diff(r[:, idx].astype(np.float64), axis=1)            # [blocks, phases]
    ph_med += np.median(ph, axis=0); ph_max += ph.max(axis=0)
    ph_slow += ph[np.argmax(r[:, 1] - r[:, 0])]
k = len(rows)
print(f"  phases of wave 0, shader clock ticks (mean over {k} steps): median workgroup | the step's slowest workgroup | per-phase maximum over workgroups")
for nm, a1, a2, a3 in zip(names, ph_med / k, ph_slow / k, ph_max / k):
    print(f"    {nm:32s} {a1:8.0f} {a2:8.0f} {a3:8.0f}")
print(f"    {'total':32s} {ph_med.sum() / k:8.0f} {ph_slow.sum() / k:8.0f}")
print("  slowest workgroups (last step):", [(int(b), f"{wall[-1][b] / 1e3:.2f} us", f"xcc {int(xcc[b])} cu {int((hw[b] >> 8) & 15)} se {int((hw[b] >> 13) & 7)}", int(counts[list(uniq).index(cu_key[b])])) for b in slow])
