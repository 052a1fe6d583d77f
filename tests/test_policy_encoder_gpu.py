"""Fused MFMA policy encoder (csrc/qs_policy_encoder.hip) against the plain PyTorch fp32 module with the same weights."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def bf16_emulation(module, obs):
    """The same forward pass with every matmul input rounded to bf16 (weights and activations) and fp32 accumulation:
    what the kernel computes, up to the summation order and its exp-based tanh."""
    import torch
    r = lambda t: t.to(torch.bfloat16).float()

    def mlp(seq, x):
        x = torch.tanh(r(x) @ r(seq[0].weight).T + seq[0].bias)
        return torch.tanh(r(x) @ r(seq[2].weight).T + seq[2].bias)

    B = obs.shape[0]
    emb = [r(mlp(module.self_encoder, obs[:, :module.self_dim]))]
    nb = module.nbr_dim * module.num_nbr
    if module.neighbor_encoder is not None:
        e = mlp(module.neighbor_encoder, obs[:, module.self_dim:module.self_dim + nb].reshape(-1, module.nbr_dim))
        emb.append(r(e.reshape(B, -1, e.shape[-1]).mean(dim=1)))
    if module.obstacle_encoder is not None:
        emb.append(r(mlp(module.obstacle_encoder, obs[:, module.self_dim + nb:])))
    x = torch.cat(emb, dim=1)
    return torch.tanh(x @ r(module.feed_forward[0].weight).T + module.feed_forward[0].bias)


@pytest.mark.parametrize("shape", [dict(num_nbr=6, obst_dim=0), dict(num_nbr=2, obst_dim=9, self_dim=19), dict(num_nbr=0, obst_dim=0),
                                   dict(num_nbr=8, obst_dim=0), dict(num_nbr=0, obst_dim=9)])
@pytest.mark.parametrize("batch", [1, 16, 77, 8192])
def test_fused_encoder_matches_torch(shape, batch):
    import torch
    from quad_swarm_rl_amd import policy
    ref = policy.make_reference_encoder(seed=3, **shape).cuda()
    with torch.no_grad():   # larger weights than the default init so that the tanh layers are exercised away from the linear regime
        for p in ref.parameters():
            p.mul_(2.5)
    fused = policy.FusedQuadEncoder(ref)
    g = torch.Generator(device="cuda").manual_seed(batch)
    D = fused.params.obs_dim
    obs = (torch.rand((batch, D), device="cuda", generator=g) * 2 - 1) * torch.tensor([3.0] * 3 + [1.0] * (D - 3), device="cuda")
    with torch.no_grad():
        want32, want16 = ref(obs), bf16_emulation(ref, obs)
    got = fused(obs)
    torch.cuda.synchronize()
    assert got.shape == (batch, 512) and torch.isfinite(got).all()
    # bf16 tolerance against the fp32 module; much tighter against the bf16-rounded restatement of the same arithmetic
    assert (got - want32).abs().max().item() < 6e-2, (got - want32).abs().max().item()
    assert (got - want16).abs().max().item() < 8e-3, (got - want16).abs().max().item()


def test_encoder_reads_the_stepper_observation_buffer():
    """obs tensor of the stepper -> fused encoder, no copy in between; rows of padded workgroups are not written."""
    import torch
    from quad_swarm_rl_amd import policy
    from quad_swarm_rl_amd.env import QuadSwarmVecEnv
    env = QuadSwarmVecEnv(5, num_agents=8, neighbor_visible_num=6, neighbor_obs_type="pos_vel", use_numba=True, collision_falloff_radius=4.0)
    obs = env.reset()
    ref = policy.make_reference_encoder(seed=1).cuda()
    fused = policy.FusedQuadEncoder(ref)
    out = torch.full((obs.shape[0] + 3, 512), 7.0, device="cuda")
    fused(obs, out=out[:obs.shape[0]])
    with torch.no_grad():
        want = ref(obs)
    torch.cuda.synchronize()
    assert (out[:obs.shape[0]] - want).abs().max().item() < 6e-2
    assert (out[obs.shape[0]:] == 7.0).all()
    env.close()
