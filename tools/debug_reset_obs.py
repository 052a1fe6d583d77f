import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import test_hip_parity as thp
case = sys.argv[1] if len(sys.argv) > 1 else "e_n64_k20"
os.environ["QS_TEAM"] = "0"
pr = thp.Pair(case, 7, "f32")
oobs, hobs = pr.reset()
E, N = pr.E, pr.N
pos, goal, rot = thp.soa(pr.hip.to_host("pos"), E, N), thp.soa(pr.hip.to_host("goal"), E, N), thp.soa(pr.hip.to_host("rot"), E, N)
for e, o in enumerate(pr.oenvs):
    s, _ = o.get_state()
    dp, dg, dr = np.abs(pos[e] - s[:, 0:3]).max(axis=1), np.abs(goal[e] - s[:, 32:35]).max(axis=1), np.abs(rot[e] - s[:, 6:15]).max(axis=1)
    print("env", e, "scenario", o.info().scenario, "max pos err", dp.max(), "at drone", dp.argmax(), "max goal err", dg.max(), "at", dg.argmax(), "max rot err", dr.max())
    if dp.max() > 1e-5:
        d = dp.argmax()
        print("   hip pos", pos[e, d], "oracle", s[d, 0:3], "goal hip", goal[e, d], "oracle", s[d, 32:35])
        print("   drones with pos err > 1e-5:", np.nonzero(dp > 1e-5)[0], "goal err > 1e-5:", np.nonzero(dg > 1e-5)[0])
so = np.abs(hobs[:, :, :18] - oobs[:, :, :18]).max(axis=2)
print("self-obs err max per env", so.max(axis=1))
pr.close()
