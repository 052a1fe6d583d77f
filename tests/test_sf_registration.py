"""Sample Factory registration surface (swarm_rl/train.py:16-27, swarm_rl/models/quad_multi_model.py:355-370) against a stand-in of
the five sample_factory entry points it touches.  The image has no sample_factory, so the real registration cannot run (bench.py
records that as config.c5); here a minimal stand-in package is put on sys.modules for the duration of the test - register_env,
global_model_factory().register_encoder_factory, model.encoder.Encoder, parse_sf_args / parse_full_cfg with argparse semantics - and
the repository's side is exercised for real: the env factory that gets registered is `sf_env.make_quadrotor_env`, the encoder
factory builds the three reference encoder classes (restatements pinned by tests/golden/encoder_*.npz) with the reference's
output sizes, and `parse_swarm_cfg` yields the reference's flag set with its defaults."""
import argparse
import sys
import types

import pytest


@pytest.fixture
def fake_sample_factory(monkeypatch):
    import torch
    reg = {"envs": {}, "encoder_factories": []}

    def mod(name, **kw):
        m = types.ModuleType(name)
        m.__dict__.update(kw)
        monkeypatch.setitem(sys.modules, name, m)
        return m

    class Encoder(torch.nn.Module):          # sample_factory.model.encoder.Encoder: nn.Module constructed with cfg
        def __init__(self, cfg):
            super().__init__()
            self.cfg = cfg

    class Factory:
        def register_encoder_factory(self, fn):
            reg["encoder_factories"].append(fn)

    factory = Factory()

    def parse_sf_args(argv=None, evaluation=False):
        p = argparse.ArgumentParser()
        p.add_argument("--env", default=None)
        p.add_argument("--rnn_size", default=256, type=int)
        p.add_argument("--with_pbt", default=False)
        partial, _ = p.parse_known_args(argv)
        return p, partial

    for name in ("sample_factory", "sample_factory.envs", "sample_factory.algo", "sample_factory.algo.utils", "sample_factory.model", "sample_factory.cfg"):
        mod(name)
    mod("sample_factory.envs.env_utils", register_env=lambda name, fn: reg["envs"].__setitem__(name, fn))
    mod("sample_factory.algo.utils.context", global_model_factory=lambda: factory)
    mod("sample_factory.model.encoder", Encoder=Encoder)
    mod("sample_factory.cfg.arguments", parse_sf_args=parse_sf_args, parse_full_cfg=lambda parser, argv=None: parser.parse_args(argv))
    monkeypatch.delitem(sys.modules, "quad_swarm_rl_amd.sf_models", raising=False)
    return reg


def test_register_swarm_components_and_encoder_factory(fake_sample_factory):
    import torch
    from quad_swarm_rl_amd import sf_env
    sf_env.register_swarm_components()
    assert fake_sample_factory["envs"] == {"quadrotor_multi": sf_env.make_quadrotor_env}
    (make_encoder,) = fake_sample_factory["encoder_factories"]
    cfg = sf_env.parse_swarm_cfg(argv=["--env=quadrotor_multi", "--quads_neighbor_encoder_type=attention", "--quads_neighbor_visible_num=6"])
    # reference defaults survive the round trip (swarm_rl/env_wrappers/quadrotor_params.py)
    assert cfg.quads_num_agents == 8 and cfg.quads_obs_repr == "xyz_vxyz_R_omega" and cfg.quads_neighbor_visible_num == 6 and cfg.rnn_size == 256
    cfg.quads_neighbor_obs_type = "pos_vel"
    cases = [  # (overrides, observation width, encoder output)
        (dict(quads_encoder_type="corl", quads_neighbor_encoder_type="attention", quads_use_obstacles=False), 18 + 36, 512),
        (dict(quads_encoder_type="corl", quads_neighbor_encoder_type="mean_embed", quads_use_obstacles=False), 18 + 36, 512),
        (dict(quads_encoder_type="corl", quads_neighbor_encoder_type="mlp", quads_use_obstacles=False), 18 + 36, 512),
        (dict(quads_encoder_type="corl", quads_neighbor_encoder_type="no_encoder", quads_use_obstacles=True, quads_obs_repr="xyz_vxyz_R_omega_floor"), 19 + 36 + 9, 512),
        (dict(quads_encoder_type="attention", quads_sim2real=False, quads_use_obstacles=True, quads_obs_repr="xyz_vxyz_R_omega_floor", quads_neighbor_visible_num=2), 19 + 12 + 9, 512),
        (dict(quads_encoder_type="attention", quads_sim2real=True, quads_use_obstacles=True, quads_obs_repr="xyz_vxyz_R_omega_floor", quads_neighbor_visible_num=2), 19 + 12 + 9, 256),
    ]
    for over, width, out_size in cases:
        c = argparse.Namespace(**dict(vars(cfg), **over))
        enc = make_encoder(c, None)
        assert enc.get_out_size() == out_size, over
        y = enc({"obs": torch.zeros(5, width)})
        assert tuple(y.shape) == (5, out_size)


def test_registration_fails_loudly_without_sample_factory():
    for name in list(sys.modules):
        assert not name.startswith("sample_factory"), "a previous test leaked the stand-in"
    from quad_swarm_rl_amd import sf_env
    try:
        import sample_factory  # noqa: F401
    except ImportError:
        with pytest.raises(ImportError):
            sf_env.register_swarm_components()
