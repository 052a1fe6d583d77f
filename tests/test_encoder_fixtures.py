"""The policy-encoder restatements (quad-swarm-rl_amd/policy.py) and the fused MFMA kernel against the REFERENCE classes' outputs.

tests/golden/encoder_*.npz come from oracle/ref_harness/capture_encoders.py: the reference's QuadMultiEncoder (every neighbour
encoder type, with / without obstacles), QuadMultiHeadAttentionEncoder and QuadSingleHeadAttentionEncoder_Sim2Real instantiated under a torch seed and run on a fixed
batch.  The restatements create their parameters in the reference's order, so the seed reproduces the reference's weights (the
fixture's per-tensor checksums prove it on this machine); then
  * CPU: the restatement's output must equal the reference class's output (1e-6; it was 0.0 when captured);
  * GPU: the fused kernel (bf16 weights / activations, fp32 accumulation) must match it to the bf16 tolerance of
    tests/test_policy_encoder_gpu.py.
"""
import glob
import os

import numpy as np
import pytest

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
NAMES = sorted(os.path.basename(p)[len("encoder_"):-len(".npz")] for p in glob.glob(os.path.join(GOLDEN, "encoder_*.npz")))


def build(name):
    import torch
    from quad_swarm_rl_amd import policy
    g = np.load(os.path.join(GOLDEN, f"encoder_{name}.npz"))
    seed, K = int(g["seed"]), int(g["num_nbr"])
    if str(g["cls"]) == "multi":
        m = policy.make_reference_encoder(seed=seed, nbr_encoder=str(g["nbr_encoder"]), num_nbr=K, obst_dim=int(g["obst_dim"]), self_dim=int(g["self_dim"]))
    else:
        m = (policy.make_reference_mha_encoder if str(g["cls"]) == "mha" else policy.make_reference_sim2real_encoder)(seed=seed, num_nbr=K)
        with torch.no_grad():
            m.attention_layer.layer_norm.weight.copy_(torch.from_numpy(g["ln"][0]))
            m.attention_layer.layer_norm.bias.copy_(torch.from_numpy(g["ln"][1]))
    return g, m


def reference_names(module):
    """state-dict keys as the reference class names them (the fixture's checksums are sorted by those)"""
    from quad_swarm_rl_amd import policy
    mha = hasattr(module, "attention_layer")
    inv = [(b, a) for a, b in (policy._KEYMAP_MHA if mha else policy._KEYMAP_MULTI)]
    kind = getattr(module, "nbr_encoder", "")
    out = {}
    for k, v in module.state_dict().items():
        if mha:
            for b, a in inv:
                if k.startswith(b):
                    k = a + k[len(b):]
                    break
        elif k.startswith("neighbor_encoder."):
            k = ("neighbor_encoder.neighbor_mlp." if kind == "mlp" else "neighbor_encoder.embedding_mlp.") + k[len("neighbor_encoder."):]
        elif k.startswith("neighbor_value_mlp.") or k.startswith("attention_mlp."):
            k = "neighbor_encoder." + k
        out[k] = v
    return out


def test_fixtures_present():
    assert len(NAMES) >= 11 and {"attention", "mean_embed", "mlp", "none", "mha", "sim2real"} <= set(NAMES)


@pytest.mark.parametrize("name", NAMES)
def test_restatement_reproduces_the_reference_class(name):
    import torch
    g, m = build(name)
    sums = np.array([[float(v.double().sum()), float(v.double().abs().sum())] for _, v in sorted(reference_names(m).items())])
    np.testing.assert_allclose(sums, g["weight_sums"], rtol=1e-12, atol=1e-12, err_msg="the seed does not reproduce the reference's weights on this torch build")
    with torch.no_grad():
        out = m(torch.from_numpy(g["obs"])).numpy()
    assert np.abs(out - g["out"]).max() <= 1e-6


@pytest.mark.parametrize("name", NAMES)
def test_state_dict_loader_round_trip(name):
    """encoder_from_state_dict: reference / Sample Factory key names (any prefix) -> the same module"""
    import torch
    from quad_swarm_rl_amd import policy
    g, m = build(name)
    sd = {"actor_critic.encoder." + k: v for k, v in reference_names(m).items()}
    m2 = policy.encoder_from_state_dict(sd, num_nbr=int(g["num_nbr"]))
    with torch.no_grad():
        assert torch.equal(m2(torch.from_numpy(g["obs"])), m(torch.from_numpy(g["obs"])))


@pytest.mark.gpu
@pytest.mark.parametrize("name", NAMES)
def test_fused_kernel_against_the_reference_class(name):
    import torch
    from quad_swarm_rl_amd import policy
    g, m = build(name)
    enc = policy.FusedQuadEncoder(m.cuda())
    out = enc(torch.from_numpy(g["obs"]).cuda().contiguous()).cpu().numpy()
    err = np.abs(out - g["out"])
    assert err.max() <= 8e-2 and err.mean() <= 1e-2, (err.max(), err.mean())
