#!/usr/bin/env python
"""How often does a WAVE need its bookkeeping rows (DESIGN.md 4a "bytes")?  Steps a C2-shaped batch on random actions like bench.py does and
prints, per probe point, the share of 64-drone waves in which at least one drone carries F_IN_COL (pair-mask row needed), F_RING_LIVE
(distance ring), F_NEWPAIR_NZ (new-pair word), and the tick range (distance sums: window open above ep_len + 1 - 5 s)."""
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import bench  # noqa: E402
import torch  # noqa: E402
from quad_swarm_rl_amd import config as qcfg, native  # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "c2"
E = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
w = bench.WORKLOADS[wl]
cfg = qcfg.make_config(num_envs=E, seed=0, precision="f32", write_rew_info=False, **w["kw"])
st = native.Stepper(cfg, device=0)
T = E * cfg.num_agents
g = torch.Generator(device="cuda").manual_seed(1234)
acts = (torch.rand((64, T, 4), device="cuda", generator=g) * 2 - 1).contiguous()
st.reset()
done_steps = 0
for upto in (50, 200, 450, 900, 1100, 1400, 1600, 3000):
    for t in range(done_steps, upto):
        st.step(acts[t % 64].data_ptr())
    done_steps = upto
    torch.cuda.synchronize()
    f = st.to_host("flags").reshape(-1)
    pad = (-len(f)) % 64
    fw = np.concatenate([f, np.zeros(pad, f.dtype)]).reshape(-1, 64)
    share = lambda bit: float(((fw & bit) != 0).any(axis=1).mean())
    lane = lambda bit: float(((f & bit) != 0).mean())
    tick = st.to_host("tick")
    print(f"{wl} E={E} after {upto:5d} steps: tick {tick.min()}..{tick.max()} | waves with F_IN_COL {share(1 << 11):.3f} (lanes {lane(1 << 11):.4f}) | "
          f"F_RING_LIVE {share(1 << 12):.3f} | F_NEWPAIR_NZ {share(1 << 13):.3f} | on floor lanes {lane(1):.3f}")
st.close()
