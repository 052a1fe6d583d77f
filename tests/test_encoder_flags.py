"""The encoder factory honours the reference's width / nonlinearity flags (swarm_rl/models/quad_multi_model.py:250-370): for flag
sets that differ from the published runs, policy.encoder_from_cfg builds the module the reference's QuadMultiEncoder builds - same
parameter names (through the checkpoint key map), same shapes, same outputs from the same torch seed.  Fixture:
tests/golden/encoder_flag_variants.json (oracle/ref_harness/capture_encoder_shapes.py, from the reference classes).  Flag
combinations the reference itself cannot run raise instead of silently building a different network."""
import json
import os
import types

import numpy as np
import pytest
import torch

from quad_swarm_rl_amd import policy

HERE = os.path.dirname(os.path.abspath(__file__))


def _variants():
    with open(os.path.join(HERE, "golden", "encoder_flag_variants.json")) as f:
        return json.load(f)


def _to_mine(key):
    for a, b in policy._KEYMAP_MULTI:
        if key.startswith(a):
            return b + key[len(a):]
    return key


@pytest.mark.parametrize("idx", range(4))
def test_factory_builds_the_reference_architecture_for_non_default_flags(idx):
    v = _variants()[idx]
    cfg = types.SimpleNamespace(**v["flags"])
    mine = policy.encoder_from_cfg(cfg, seed=v["seed"])
    got = {k: list(p.shape) for k, p in mine.state_dict().items()}
    want = {_to_mine(k): shape for k, shape in v["params"]}
    assert got == want
    assert mine.feed_forward[0].out_features == v["out_size"] == 2 * cfg.rnn_size
    with torch.no_grad():
        y = mine(torch.tensor(v["obs"], dtype=torch.float32))
    np.testing.assert_allclose(y.numpy(), np.array(v["out"], dtype=np.float32), rtol=0, atol=1e-6)   # same seed => same weights (creation order)


def test_flag_combinations_the_reference_cannot_run_are_rejected():
    base = dict(quads_obs_repr="xyz_vxyz_R_omega", quads_use_obstacles=False, quads_neighbor_visible_num=6, quads_num_agents=8, quads_neighbor_obs_type="pos_vel",
                quads_neighbor_encoder_type="attention", quads_neighbor_hidden_size=256, quads_obst_hidden_size=256, rnn_size=256, nonlinearity="tanh")
    with pytest.raises(NotImplementedError, match="obstacle"):
        policy.encoder_from_cfg(types.SimpleNamespace(**dict(base, quads_encoder_type="attention")))       # multi-head attention without obstacle columns
    with pytest.raises(NotImplementedError):
        policy.encoder_from_cfg(types.SimpleNamespace(**dict(base, quads_encoder_type="corl", nonlinearity="gelu")))
    with pytest.raises(NotImplementedError):
        policy.encoder_from_cfg(types.SimpleNamespace(**dict(base, quads_encoder_type="corl", quads_neighbor_encoder_type="transformer")))


def test_unseeded_factory_draws_from_the_callers_generator():
    """Sample Factory seeds torch once and then builds the model: seed=None must not reseed"""
    v = _variants()[1]
    cfg = types.SimpleNamespace(**v["flags"])
    torch.manual_seed(v["seed"])
    a = policy.encoder_from_cfg(cfg)
    b = policy.encoder_from_cfg(cfg, seed=v["seed"])
    for (ka, pa), (kb, pb) in zip(a.state_dict().items(), b.state_dict().items()):
        assert ka == kb and torch.equal(pa, pb)
    c = policy.encoder_from_cfg(cfg)          # the generator has moved on
    assert not torch.equal(c.self_encoder[0].weight, a.self_encoder[0].weight)


def test_reference_precision_operand_format():
    """precision="fp32" packs every weight as a PAIR of fp16 numbers (include/quadswarm_encoder.h, policy.split_fp16): h + l / 2048 gives the
    fp32 value back to ~2^-22 relative (2^-26 absolute below 2^-14), h is never a subnormal fp16, and the fragment layout is the bf16 one with two planes per fragment."""
    rng = np.random.RandomState(0)
    x = np.concatenate([rng.standard_normal(20000) * 10.0 ** rng.uniform(-9, 3, 20000), [0.0, 1.0, -1.0, 65504.0, 1e6, -1e6, 6e-5, 6.2e-5, 1e-7]]).astype(np.float32)
    h, l = policy.split_fp16(x)
    assert h.dtype == np.float16 and l.dtype == np.float16 and np.isfinite(h).all() and np.isfinite(l).all()
    assert ((h == 0) | (np.abs(h.astype(np.float32)) >= 2.0 ** -14)).all()
    back = h.astype(np.float64) + l.astype(np.float64) / 2048.0
    want = np.clip(x, -65504.0, 65504.0).astype(np.float64)
    err = np.abs(back - want)
    # (below the smallest normal fp16, 2^-14, the value lives in l alone: 11 bits of something that small, <= 2^-26 absolute)
    assert (err <= np.maximum(np.abs(want) * 2.0 ** -21, 2.0 ** -26)).all(), (err / np.maximum(np.abs(want), 1e-30)).max()
    lin = torch.nn.Linear(40, 24)
    w16, b, M, K = policy.pack_linear(lin, "cpu")
    w2, b2, M2, K2 = policy.pack_linear(lin, "cpu", split=True)
    assert (M, K) == (M2, K2) == (32, 64) and w2.shape == (M // 16, K // 32, 2, 64, 8) and w2.dtype == torch.float16 and torch.equal(b, b2)
    rebuilt = (w2[:, :, 0].double() + w2[:, :, 1].double() / 2048.0)
    assert torch.equal(rebuilt.to(torch.bfloat16), w16)                       # same elements in the same fragment slots ...
    W = np.zeros((M, K)); W[:24, :40] = lin.weight.detach().numpy()
    lane = np.arange(64)
    frag = W[(lane & 15)[:, None] + 16 * 1, (8 * (lane >> 4))[:, None] + np.arange(8)[None, :] + 32 * 1]   # tile 1, K-step 1
    assert np.abs(rebuilt[1, 1].numpy() - frag).max() <= np.abs(frag).max() * 2.0 ** -21   # ... to fp32 accuracy
