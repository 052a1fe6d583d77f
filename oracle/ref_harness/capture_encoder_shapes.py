#!/usr/bin/env python
"""Fixture tests/golden/encoder_flag_variants.json: what the REFERENCE's encoder factory builds for non-default flags.  Build
container only (imports /root/reference through the stubs of capture_encoders.py).

For each flag set - --quads_neighbor_hidden_size / --quads_obst_hidden_size different from --rnn_size, --nonlinearity elu / relu -
the reference class (swarm_rl/models/quad_multi_model.py:250-350) is instantiated under a fixed torch seed; the fixture keeps the
flags, every parameter's name and shape, and the class's output on a fixed observation batch.  tests/test_encoder_flags.py builds
the same thing with quad-swarm-rl_amd/policy.encoder_from_cfg and compares (names through policy's key map, shapes, outputs)."""
import json
import os
import sys
import types

import torch
from torch import nn

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import capture_encoders as ce   # noqa: E402  (sets up the sample_factory stubs and imports the reference's model file)

ACT = {"tanh": nn.Tanh, "elu": lambda: nn.ELU(inplace=True), "relu": lambda: nn.ReLU(inplace=True)}
sys.modules["sample_factory.model.model_utils"].nonlinearity = lambda cfg: ACT[cfg.nonlinearity]()   # what SF's nonlinearity(cfg) returns
ce.ref_model.nonlinearity = sys.modules["sample_factory.model.model_utils"].nonlinearity

VARIANTS = [
    dict(quads_neighbor_encoder_type="attention", quads_neighbor_hidden_size=128, quads_obst_hidden_size=64, rnn_size=256, nonlinearity="tanh", obst=True, K=2),
    dict(quads_neighbor_encoder_type="mean_embed", quads_neighbor_hidden_size=64, quads_obst_hidden_size=256, rnn_size=128, nonlinearity="elu", obst=False, K=6),
    dict(quads_neighbor_encoder_type="mlp", quads_neighbor_hidden_size=96, quads_obst_hidden_size=32, rnn_size=256, nonlinearity="relu", obst=True, K=3),
    dict(quads_neighbor_encoder_type="no_encoder", quads_neighbor_hidden_size=256, quads_obst_hidden_size=48, rnn_size=64, nonlinearity="tanh", obst=True, K=2),
]


def main():
    out = []
    for idx, v in enumerate(VARIANTS):
        seed = 2000 + idx
        obst, K = v["obst"], v["K"]
        flags = dict(quads_obs_repr="xyz_vxyz_R_omega_floor" if obst else "xyz_vxyz_R_omega", quads_use_obstacles=obst, quads_neighbor_visible_num=K,
                     quads_num_agents=8, quads_neighbor_obs_type="pos_vel", quads_obstacle_obs_type="octomap", quads_encoder_type="corl",
                     **{k: v[k] for k in ("quads_neighbor_encoder_type", "quads_neighbor_hidden_size", "quads_obst_hidden_size", "rnn_size", "nonlinearity")})
        cfg = types.SimpleNamespace(**flags)
        torch.manual_seed(seed)
        theirs = ce.ref_model.QuadMultiEncoder(cfg, None)
        D = (19 if obst else 18) + 6 * K + (9 if obst else 0)
        g = torch.Generator().manual_seed(seed + 1)
        obs = torch.rand((3, D), generator=g) * 2 - 1
        with torch.no_grad():
            y = theirs({"obs": obs})
        out.append(dict(flags=flags, seed=seed, params=[[k, list(p.shape)] for k, p in theirs.state_dict().items()],
                        obs=obs.tolist(), out=y.tolist(), out_size=theirs.get_out_size()))
        print(flags["quads_neighbor_encoder_type"], "params", len(out[-1]["params"]), "out", tuple(y.shape))
    path = os.path.join(ce.REPO, "tests", "golden", "encoder_flag_variants.json")
    with open(path, "w") as f:
        json.dump(out, f)
    print("wrote", path)


if __name__ == "__main__":
    main()
