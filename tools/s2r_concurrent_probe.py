"""Does the Sim2Real / multi-head kernel stay run-to-run deterministic while OTHER kernels with large LDS footprints share the GPU
(a second stream running the mean_embed encoder)?  (GPU box)"""
import sys, os
sys.path.insert(0, os.getcwd())
import torch
from quad_swarm_rl_amd import policy
for name, make in (("sim2real", policy.make_reference_sim2real_encoder), ("mha", policy.make_reference_mha_encoder)):
    fused = policy.FusedQuadEncoder(make(seed=11, num_nbr=6).cuda())
    other = policy.FusedQuadEncoder(policy.make_reference_encoder(seed=1, nbr_encoder="mean_embed").cuda())
    B = 8192
    obs = torch.rand((B, fused.params.obs_dim), device="cuda") * 2 - 1
    obs2 = torch.rand((2048, other.params.obs_dim), device="cuda") * 2 - 1
    ref = fused(obs).clone()
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    bad = 0
    for rep in range(40):
        with torch.cuda.stream(side):
            for _ in range(6):
                other(obs2)          # 128 narrow workgroups (80 KB LDS each) scattered over the CUs while the kernel under test runs
        got = fused(obs)
        torch.cuda.synchronize()
        bad += int(not torch.equal(got, ref))
    print(name, "runs differing from the undisturbed result:", bad, "of 40")
