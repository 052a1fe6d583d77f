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
        return torch.tanh(r(x) @ r(seq[2].weight).T + seq[2].bias) if len(seq) > 2 else x   # the Sim2Real class has one-layer embeddings

    B = obs.shape[0]
    emb = [r(mlp(module.self_encoder, obs[:, :module.self_dim]))]
    nb = module.nbr_dim * module.num_nbr
    if getattr(module, "attention", False):
        K = module.num_nbr
        nbrs = obs[:, module.self_dim:module.self_dim + nb].reshape(-1, module.nbr_dim)
        e = mlp(module.neighbor_encoder, torch.cat((obs[:, :module.self_dim].repeat(K, 1), nbrs), dim=1))
        e_mean = r(e.reshape(B, K, -1).mean(dim=1))       # mean of the fp32 e_i, then bf16 as a matmul input
        e = r(e)                                            # e_i travels between the two launches as bf16
        h = mlp(module.neighbor_value_mlp, e)
        a = module.attention_mlp
        W1 = r(a[0].weight)
        x = torch.tanh(e @ W1[:, :e.shape[1]].T + (e_mean @ W1[:, e.shape[1]:].T).repeat(K, 1) + a[0].bias)
        x = torch.tanh(r(x) @ r(a[2].weight).T + a[2].bias)
        alpha = (x @ a[4].weight.T + a[4].bias).view(B, K)     # last layer in fp32 from the accumulators
        emb.append(r((torch.softmax(alpha, dim=1).view(-1, 1) * h).view(B, K, -1).sum(dim=1)))
    elif getattr(module, "nbr_encoder", "") == "mlp":
        x = obs[:, module.self_dim:module.self_dim + nb]
        for i in (0, 2, 4):
            x = torch.tanh(r(x) @ r(module.neighbor_encoder[i].weight).T + module.neighbor_encoder[i].bias)
        emb.append(r(x))
    elif module.neighbor_encoder is not None:
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
    got = fused(obs).clone()
    again = fused(obs)
    torch.cuda.synchronize()
    assert got.shape == (batch, 512) and torch.isfinite(got).all()
    assert torch.equal(got, again)                                     # run-to-run identical, also with two workgroups per CU
    # bf16 tolerance against the fp32 module; much tighter against the bf16-rounded restatement of the same arithmetic
    assert (got - want32).abs().max().item() < 6e-2, (got - want32).abs().max().item()
    assert (got - want16).abs().max().item() < 8e-3, (got - want16).abs().max().item()


@pytest.mark.parametrize("shape", [dict(num_nbr=6, obst_dim=0), dict(num_nbr=2, obst_dim=9, self_dim=19), dict(num_nbr=8, obst_dim=0),
                                   dict(num_nbr=1, obst_dim=0), dict(num_nbr=5, obst_dim=9), dict(num_nbr=3, obst_dim=0), dict(num_nbr=7, obst_dim=9)])
@pytest.mark.parametrize("batch", [1, 31, 33, 77, 4111, 8192])
@pytest.mark.parametrize("attention", [False, True])
def test_wide_workgroup_kernels_match_torch_and_the_narrow_kernels(shape, batch, attention):
    """The 32-agents-per-workgroup kernels (weight ring carried across layers), forced for every batch size, against the PyTorch
    module, its bf16 restatement and the 16-agent kernels."""
    import torch
    from quad_swarm_rl_amd import policy
    ref = policy.make_reference_encoder(seed=13, attention=attention, **shape).cuda()
    with torch.no_grad():
        for p in ref.parameters():
            p.mul_(2.5)
    fused = policy.FusedQuadEncoder(ref)
    g = torch.Generator(device="cuda").manual_seed(batch + 77)
    D = fused.params.obs_dim
    obs = (torch.rand((batch, D), device="cuda", generator=g) * 2 - 1) * torch.tensor([3.0] * 3 + [1.0] * (D - 3), device="cuda")
    with torch.no_grad():
        want32, want16 = ref(obs), bf16_emulation(ref, obs)
    prev = policy.lib().qs_enc_set_wide_min(1)
    try:
        wide = fused(obs).clone()
        assert torch.equal(fused(obs), wide)                           # run-to-run identical
        policy.lib().qs_enc_set_wide_min(0)
        narrow = fused(obs).clone()
        assert torch.equal(fused(obs), narrow)
    finally:
        policy.lib().qs_enc_set_wide_min(prev)
    torch.cuda.synchronize()
    assert torch.isfinite(wide).all()
    assert (wide - want32).abs().max().item() < 8e-2, (wide - want32).abs().max().item()
    assert (wide - want16).abs().max().item() < (2e-2 if attention else 8e-3), (wide - want16).abs().max().item()
    assert (wide - narrow).abs().max().item() < (2e-2 if attention else 8e-3), (wide - narrow).abs().max().item()


@pytest.mark.parametrize("shape", [dict(num_nbr=6, obst_dim=0), dict(num_nbr=6, obst_dim=9), dict(num_nbr=2, obst_dim=9, self_dim=19), dict(num_nbr=2, obst_dim=0),
                                   dict(num_nbr=4, obst_dim=0), dict(num_nbr=5, obst_dim=9)])
@pytest.mark.parametrize("batch", [1, 33, 4111, 8192])
@pytest.mark.parametrize("head", [0, 4])
def test_pingpong_schedule_equals_the_lockstep_schedule_bit_for_bit(shape, batch, head):
    """mean_embed on the 32-agent workgroups with 2, 4, 5 or 6 neighbours runs the two waves of every SIMD half a layer apart (pp_body: one in a
    K loop, the other in a tanh epilogue; qs_enc_set_pingpong).  It is a schedule, not arithmetic: the same MFMAs in the same order per
    output as wide_body - features (and a fused head's outputs) identical bit for bit, run to run as well."""
    import torch
    from quad_swarm_rl_amd import policy
    ref = policy.make_reference_encoder(seed=29, **shape).cuda()
    with torch.no_grad():
        for p in ref.parameters():
            p.mul_(2.0)
    fused = policy.FusedQuadEncoder(ref)
    g = torch.Generator(device="cuda").manual_seed(batch + 5)
    D = fused.params.obs_dim
    obs = (torch.rand((batch, D), device="cuda", generator=g) * 2 - 1) * torch.tensor([3.0] * 3 + [1.0] * (D - 3), device="cuda")
    if head:
        fused.set_head(torch.randn((head, 512), device="cuda", generator=g) * 0.05, torch.randn((head,), device="cuda", generator=g))

    def run():
        feats = torch.empty((batch, 512), device="cuda")
        if not head:
            return fused(obs, out=feats), None
        return feats, fused.forward_head(obs, features=feats).clone()

    prev = policy.lib().qs_enc_set_wide_min(1)
    prev_pp = policy.lib().qs_enc_set_pingpong(1)
    try:
        pp, pp_head = run()
        for _ in range(3):
            again, again_head = run()
            assert torch.equal(again, pp) and (not head or torch.equal(again_head, pp_head))
        policy.lib().qs_enc_set_pingpong(0)
        lock, lock_head = run()
    finally:
        policy.lib().qs_enc_set_wide_min(prev)
        policy.lib().qs_enc_set_pingpong(prev_pp)
    torch.cuda.synchronize()
    assert torch.isfinite(pp).all()
    assert torch.equal(pp, lock), (pp - lock).abs().max().item()
    if head:
        assert torch.equal(pp_head, lock_head)


@pytest.mark.parametrize("shape", [dict(num_nbr=6, obst_dim=0), dict(num_nbr=2, obst_dim=9, self_dim=19), dict(num_nbr=8, obst_dim=0),
                                   dict(num_nbr=1, obst_dim=0), dict(num_nbr=5, obst_dim=9)])
@pytest.mark.parametrize("batch", [1, 16, 77, 8192])
def test_fused_attention_encoder_matches_torch(shape, batch):
    """`attention` neighbour encoder (quad_multi_model.py:46-101), including the row pairing of its two .repeat() calls:
    row (agent a, neighbour k) sees the self observation and the mean embedding of agent (a*K + k) mod batch."""
    import torch
    from quad_swarm_rl_amd import policy
    ref = policy.make_reference_encoder(seed=5, attention=True, **shape).cuda()
    with torch.no_grad():
        for p in ref.parameters():
            p.mul_(2.5)
        ref.attention_mlp[4].weight.mul_(4.0)   # spread the scores so that the softmax is far from uniform
    fused = policy.FusedQuadEncoder(ref)
    assert fused.params.nbr_encoder == 1
    g = torch.Generator(device="cuda").manual_seed(batch + 1000)
    D = fused.params.obs_dim
    obs = (torch.rand((batch, D), device="cuda", generator=g) * 2 - 1) * torch.tensor([3.0] * 3 + [1.0] * (D - 3), device="cuda")
    with torch.no_grad():
        want32, want16 = ref(obs), bf16_emulation(ref, obs)
    got = fused(obs)
    again = fused(obs)   # scratch buffers are reused
    torch.cuda.synchronize()
    assert got.shape == (batch, 512) and torch.isfinite(got).all()
    assert torch.equal(got, again)
    # the sharpened softmax amplifies bf16 rounding of the scores: measured max 1.0e-2 / mean 9e-5 against the bf16 restatement
    # over 4M outputs (3e-3 / 6e-5 with the unsharpened scores), 4.7e-2 against fp32 - which the bf16 restatement itself shows too
    assert (got - want32).abs().max().item() < 8e-2, (got - want32).abs().max().item()
    assert (got - want16).abs().max().item() < 2e-2, (got - want16).abs().max().item()
    assert (got - want16).abs().mean().item() < 3e-4, (got - want16).abs().mean().item()


@pytest.mark.parametrize("nbr_encoder", ["mlp", "no_encoder"])
@pytest.mark.parametrize("shape", [dict(num_nbr=6, obst_dim=0), dict(num_nbr=2, obst_dim=9, self_dim=19), dict(num_nbr=8, obst_dim=9)])
@pytest.mark.parametrize("batch", [1, 77, 8192])
def test_fused_mlp_and_blind_encoders_match_torch(nbr_encoder, shape, batch):
    """--quads_neighbor_encoder_type=mlp (quad_multi_model.py:104-122) and =no_encoder (:289-291; train_local_obst.sh: the
    neighbour columns sit between the self and the obstacle columns but only those two are encoded)."""
    import torch
    from quad_swarm_rl_amd import policy
    ref = policy.make_reference_encoder(seed=7, nbr_encoder=nbr_encoder, **shape).cuda()
    with torch.no_grad():
        for p in ref.parameters():
            p.mul_(2.5)
    fused = policy.FusedQuadEncoder(ref)
    assert fused.params.nbr_encoder == policy.NBR_ENCODERS.index(nbr_encoder)
    g = torch.Generator(device="cuda").manual_seed(batch + 7)
    D = fused.params.obs_dim
    obs = (torch.rand((batch, D), device="cuda", generator=g) * 2 - 1) * torch.tensor([3.0] * 3 + [1.0] * (D - 3), device="cuda")
    with torch.no_grad():
        want32, want16 = ref(obs), bf16_emulation(ref, obs)
    got = fused(obs)
    torch.cuda.synchronize()
    assert got.shape == (batch, 512) and torch.isfinite(got).all()
    assert (got - want32).abs().max().item() < 6e-2, (got - want32).abs().max().item()
    assert (got - want16).abs().max().item() < 8e-3, (got - want16).abs().max().item()
    if nbr_encoder == "no_encoder":   # blind to the neighbour columns
        obs2 = obs.clone()
        obs2[:, ref.self_dim:ref.self_dim + ref.nbr_dim * ref.num_nbr] = 9.0
        assert torch.equal(fused(obs2), got)


def mha_bf16_emulation(module, obs):
    """QuadMultiHeadAttentionEncoderRef with every matmul input rounded to bf16 like the kernel: the tokens enter the projections
    and the concatenated heads enter fc as bf16; scores, softmax, residual and LayerNorm stay fp32."""
    import torch
    r = lambda t: t.to(torch.bfloat16).float()

    def mlp(seq, x):
        x = torch.tanh(r(x) @ r(seq[0].weight).T + seq[0].bias)
        return torch.tanh(r(x) @ r(seq[2].weight).T + seq[2].bias) if len(seq) > 2 else x   # the Sim2Real class has one-layer embeddings

    B, nb = obs.shape[0], module.nbr_dim * module.num_nbr
    s = mlp(module.self_encoder, obs[:, :module.self_dim])
    x = torch.stack((mlp(module.neighbor_encoder, obs[:, module.self_dim:module.self_dim + nb]), mlp(module.obstacle_encoder, obs[:, module.self_dim + nb:])), dim=1)
    a = module.attention_layer
    H = a.w_qs.weight.shape[0] // 256   # heads: 4, or 1 (OneHeadAttention)
    q = (r(x) @ r(a.w_qs.weight).T).view(B, 2, H, 256).transpose(1, 2)
    k = (r(x) @ r(a.w_ks.weight).T).view(B, 2, H, 256).transpose(1, 2)
    v = (r(x) @ r(a.w_vs.weight).T).view(B, 2, H, 256).transpose(1, 2)
    p = torch.softmax(torch.matmul(q, k.transpose(2, 3)) / 16.0, dim=-1)
    o = torch.matmul(p, v).transpose(1, 2).contiguous().view(B, 2, -1)
    y = a.layer_norm(r(o) @ r(a.fc.weight).T + x)
    cat = torch.cat((r(s), r(y.reshape(B, -1))), dim=1)
    return torch.tanh(cat @ r(module.feed_forward[0].weight).T + module.feed_forward[0].bias)


@pytest.mark.parametrize("shape", [dict(num_nbr=2), dict(num_nbr=6), dict(num_nbr=8, self_dim=18), dict(num_nbr=1, obst_dim=25)])
@pytest.mark.parametrize("batch", [1, 16, 77, 8192])
@pytest.mark.parametrize("sim2real", [False, True])
def test_fused_multi_head_attention_encoder_matches_torch(shape, batch, sim2real):
    """QuadMultiHeadAttentionEncoder (--quads_encoder_type=attention; quad_multi_model.py:124-196, attention_layer.py:12-56) and
    its --quads_sim2real subclass QuadSingleHeadAttentionEncoder_Sim2Real (:199-248, attention_layer.py:56-97)."""
    import torch
    from quad_swarm_rl_amd import policy
    ref = (policy.make_reference_sim2real_encoder if sim2real else policy.make_reference_mha_encoder)(seed=11, **shape).cuda()
    with torch.no_grad():
        for p in ref.parameters():
            p.mul_(2.0)
        a = ref.attention_layer
        a.w_qs.weight.mul_(1.5)   # scores of a few units: the 2-way softmax weights spread over 0.04 .. 0.96
        a.layer_norm.weight.uniform_(0.5, 1.5)
        a.layer_norm.bias.uniform_(-0.3, 0.3)
    fused = policy.FusedQuadEncoder(ref)
    assert fused.params.nbr_encoder == (5 if sim2real else 4)
    g = torch.Generator(device="cuda").manual_seed(batch + 2000)
    D = fused.params.obs_dim
    obs = (torch.rand((batch, D), device="cuda", generator=g) * 2 - 1) * torch.tensor([3.0] * 3 + [1.0] * (D - 3), device="cuda")
    with torch.no_grad():
        want32, want16 = ref(obs), mha_bf16_emulation(ref, obs)
    got = fused(obs).clone()
    again = fused(obs)
    torch.cuda.synchronize()
    assert got.shape == (batch, 256 if sim2real else 512) and torch.isfinite(got).all()
    assert torch.equal(got, again)
    assert (got - want32).abs().max().item() < 8e-2, (got - want32).abs().max().item()
    assert (got - want16).abs().max().item() < 2e-2, (got - want16).abs().max().item()
    assert (got - want16).abs().mean().item() < 5e-4, (got - want16).abs().mean().item()


REFERENCE_PRECISION_TOL = 1e-5   # "an fp32-accurate variant ... tested at <= 1e-5 against the fp32 module"


@pytest.mark.parametrize("nbr_encoder", ["mean_embed", "attention", "mlp", "no_encoder"])
@pytest.mark.parametrize("shape", [dict(num_nbr=6, obst_dim=0), dict(num_nbr=2, obst_dim=9, self_dim=19), dict(num_nbr=8, obst_dim=9), dict(num_nbr=5, obst_dim=0)])
@pytest.mark.parametrize("batch", [1, 77, 8192])
def test_reference_precision_encoder_matches_the_fp32_module(nbr_encoder, shape, batch):
    """precision="fp32" (fp16-pair operands, three MFMAs per product; include/quadswarm_encoder.h): the features of the reference's fp32
    modules to 1e-5 - against the module evaluated in float64 (the value both fp32 evaluations approximate) AND against the fp32 module
    as torch runs it on the GPU, whose own distance from the float64 value is printed beside ours in the failure message."""
    import copy
    import torch
    from quad_swarm_rl_amd import policy
    ref = policy.make_reference_encoder(seed=5, nbr_encoder=nbr_encoder, **shape).cuda()
    with torch.no_grad():
        for p in ref.parameters():
            p.mul_(2.5)
    fused = policy.FusedQuadEncoder(ref, precision="fp32")
    g = torch.Generator(device="cuda").manual_seed(batch + 1)
    D = fused.params.obs_dim
    obs = (torch.rand((batch, D), device="cuda", generator=g) * 2 - 1) * torch.tensor([3.0] * 3 + [1.0] * (D - 3), device="cuda")
    with torch.no_grad():
        want32 = ref(obs)
        want64 = copy.deepcopy(ref).double()(obs.double())
    got = fused(obs).clone()
    again = fused(obs)
    torch.cuda.synchronize()
    assert got.shape == (batch, 512) and torch.isfinite(got).all() and torch.equal(got, again)
    e64, e32, t32 = (got.double() - want64).abs().max().item(), (got - want32).abs().max().item(), (want32.double() - want64).abs().max().item()
    assert e64 < REFERENCE_PRECISION_TOL and e32 < REFERENCE_PRECISION_TOL, f"fused vs float64 {e64:.2e}, fused vs torch fp32 {e32:.2e}, torch fp32 vs float64 {t32:.2e}"
    # and the bf16 kernels really are two to three orders of magnitude further away: the mode is not a relabelled bf16 path
    far = (policy.FusedQuadEncoder(ref)(obs).double() - want64).abs().max().item()
    assert far > 50 * e64, (far, e64)


def test_reference_precision_head_sampling_and_refresh():
    """the fused head, the Gaussian sampling epilogue and refresh() in reference precision; an unknown precision code is refused"""
    import torch
    from quad_swarm_rl_amd import native, policy
    ref = policy.make_reference_encoder(seed=9, num_nbr=6).cuda()
    fused = policy.FusedQuadEncoder(ref, precision="fp32")
    head = torch.nn.Linear(512, 4).cuda()
    fused.set_head(head.weight, head.bias)
    obs = torch.rand((333, fused.params.obs_dim), device="cuda") * 2 - 1
    feats = torch.empty((333, 512), device="cuda")
    with torch.no_grad():
        want = head(ref(obs))
    got = fused.forward_head(obs, features=feats).clone()
    assert (got - want).abs().max().item() < REFERENCE_PRECISION_TOL and (feats - ref(obs)).abs().max().item() < REFERENCE_PRECISION_TOL
    with torch.no_grad():
        for p in ref.parameters():
            p.add_(0.01 * torch.randn_like(p))
    fused.refresh()
    with torch.no_grad():
        want2 = head(ref(obs))
    assert (fused.forward_head(obs) - want2).abs().max().item() < REFERENCE_PRECISION_TOL and (want2 - want).abs().max().item() > 1e-3
    bad = policy.FusedQuadEncoder(policy.make_reference_mha_encoder().cuda())
    bad.params.precision = 2
    with pytest.raises(native.QsError, match="precision"):
        bad(torch.zeros((4, bad.params.obs_dim), device="cuda"))


@pytest.mark.parametrize("shape", [dict(num_nbr=2), dict(num_nbr=6), dict(num_nbr=8, self_dim=18), dict(num_nbr=1, obst_dim=25)])
@pytest.mark.parametrize("batch", [1, 77, 8192])
@pytest.mark.parametrize("sim2real", [False, True])
def test_reference_precision_multi_head_encoder_matches_the_fp32_module(shape, batch, sim2real):
    """precision="fp32" for QuadMultiHeadAttentionEncoder and its Sim2Real subclass (quad_multi_model.py:124-248, attention_layer.py:12-97):
    embeddings, q / k / v projections, the heads' output projection and the feed-forward layer on fp16 pairs; scores, softmax, residual and
    LayerNorm in fp32.  Same bar as the neighbour encoders: 1e-5 from the module in float64 and from torch's fp32 forward."""
    import copy
    import torch
    from quad_swarm_rl_amd import policy
    ref = (policy.make_reference_sim2real_encoder if sim2real else policy.make_reference_mha_encoder)(seed=11, **shape).cuda()
    with torch.no_grad():
        for p in ref.parameters():
            p.mul_(2.0)
        a = ref.attention_layer
        a.w_qs.weight.mul_(1.5)   # scores of a few units: the 2-way softmax weights spread over 0.04 .. 0.96
        a.layer_norm.weight.uniform_(0.5, 1.5)
        a.layer_norm.bias.uniform_(-0.3, 0.3)
    fused = policy.FusedQuadEncoder(ref, precision="fp32")
    assert fused.params.nbr_encoder == (5 if sim2real else 4) and fused.params.precision == 1
    g = torch.Generator(device="cuda").manual_seed(batch + 2000)
    D = fused.params.obs_dim
    obs = (torch.rand((batch, D), device="cuda", generator=g) * 2 - 1) * torch.tensor([3.0] * 3 + [1.0] * (D - 3), device="cuda")
    with torch.no_grad():
        want32 = ref(obs)
        want64 = copy.deepcopy(ref).double()(obs.double())
    got = fused(obs).clone()
    again = fused(obs)
    torch.cuda.synchronize()
    assert got.shape == (batch, 256 if sim2real else 512) and torch.isfinite(got).all() and torch.equal(got, again)
    e64, e32, t32 = (got.double() - want64).abs().max().item(), (got - want32).abs().max().item(), (want32.double() - want64).abs().max().item()
    assert e64 < REFERENCE_PRECISION_TOL and e32 < REFERENCE_PRECISION_TOL, f"fused vs float64 {e64:.2e}, fused vs torch fp32 {e32:.2e}, torch fp32 vs float64 {t32:.2e}"
    far = (policy.FusedQuadEncoder(ref)(obs).double() - want64).abs().max().item()
    assert far > 50 * e64, (far, e64)


@pytest.mark.parametrize("sim2real", [False, True])
@pytest.mark.parametrize("batch", [4112, 16384])
def test_multi_head_kernels_are_run_to_run_identical_above_one_workgroup_per_cu(batch, sim2real):
    """The multi-head / Sim2Real kernel body is one workgroup per CU by construction (its LDS request is more than half a CU's LDS,
    tests/test_c_abi.py).  From 4097 agents on a CU runs several workgroups one after the other: five forwards of the same batch must be
    bit-identical (a two-workgroups-per-CU experiment of this body was not, and was never shipped)."""
    import torch
    from quad_swarm_rl_amd import policy
    ref = (policy.make_reference_sim2real_encoder if sim2real else policy.make_reference_mha_encoder)(seed=5, num_nbr=6).cuda()
    fused = policy.FusedQuadEncoder(ref)
    g = torch.Generator(device="cuda").manual_seed(batch)
    obs = torch.rand((batch, fused.params.obs_dim), device="cuda", generator=g) * 2 - 1
    first = fused(obs).clone()
    for _ in range(4):
        assert torch.equal(fused(obs), first)
    torch.cuda.synchronize()


def test_attention_encoder_is_not_the_per_agent_pairing():
    """Guards the quirk: with the 'natural' pairing (row (a,k) with agent a) the result differs measurably for batch > 1."""
    import torch
    from quad_swarm_rl_amd import policy
    ref = policy.make_reference_encoder(seed=6, attention=True).cuda()
    with torch.no_grad():
        for p in ref.parameters():
            p.mul_(2.5)
    fused = policy.FusedQuadEncoder(ref)
    obs = torch.rand((64, fused.params.obs_dim), device="cuda") * 2 - 1
    got = fused(obs)
    with torch.no_grad():
        K, B = ref.num_nbr, obs.shape[0]
        nbrs = obs[:, ref.self_dim:ref.self_dim + K * ref.nbr_dim].reshape(-1, ref.nbr_dim)
        e = ref.neighbor_encoder(torch.cat((obs[:, :ref.self_dim].repeat_interleave(K, dim=0), nbrs), dim=1))
        h = ref.neighbor_value_mlp(e)
        em = e.reshape(B, K, -1).mean(dim=1).repeat_interleave(K, dim=0)
        w = torch.softmax(ref.attention_mlp(torch.cat((e, em), dim=1)).view(B, K), dim=1).view(-1, 1)
        natural = ref.feed_forward(torch.cat((ref.self_encoder(obs[:, :ref.self_dim]), (w * h).view(B, K, -1).sum(dim=1)), dim=1))
        want = ref(obs)
    torch.cuda.synchronize()
    assert (got - want).abs().max().item() < 8e-2
    assert (got - natural).abs().max().item() > 0.2


@pytest.mark.parametrize("model", ["attention", "mean_embed", "mha", "sim2real"])
@pytest.mark.parametrize("hdim,batch", [(4, 77), (1, 8192), (8, 16)])
def test_fused_linear_head(model, hdim, batch):
    """Linear head in the encoder's epilogue == the same Linear applied to the features the kernel writes; features optional."""
    import torch
    from quad_swarm_rl_amd import policy
    ref = (policy.make_reference_mha_encoder(seed=3) if model == "mha" else policy.make_reference_sim2real_encoder(seed=3) if model == "sim2real"
           else policy.make_reference_encoder(seed=3, nbr_encoder=model)).cuda()
    fused = policy.FusedQuadEncoder(ref)
    g = torch.Generator(device="cuda").manual_seed(hdim)
    obs = torch.rand((batch, fused.params.obs_dim), device="cuda", generator=g) * 2 - 1
    W = torch.randn((hdim, fused.out_dim), device="cuda", generator=g) * 0.1
    b = torch.randn((hdim,), device="cuda", generator=g)
    feats = fused(obs)
    fused.set_head(W, b)
    got = torch.full((batch + 2, hdim), 7.0, device="cuda")
    fused.forward_head(obs, head_out=got[:batch])                           # no feature tensor at all
    feats2 = torch.zeros_like(feats)
    got2 = fused.forward_head(obs, features=feats2)
    torch.cuda.synchronize()
    want = feats.double() @ W.double().T + b.double()
    assert (got[:batch].double() - want).abs().max().item() < 1e-4
    assert (got[batch:] == 7.0).all()
    assert torch.equal(got2, got[:batch]) and torch.equal(feats2, feats)
    assert torch.equal(fused(obs), feats)                                   # the plain forward is unaffected afterwards


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
