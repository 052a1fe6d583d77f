"""Sample Factory model registration (swarm_rl/models/quad_multi_model.py:355-370): `make_quadmulti_encoder` hands SF an Encoder whose
body is one of the restatements of quad-swarm-rl_amd/policy.py - the modules tests/golden/encoder_*.npz pin against the reference
classes.  Training differentiates through the torch module; rollouts outside SF (rollout.py) run the same weights on the fused
kernel (policy.FusedQuadEncoder).  Importing this module needs sample_factory."""
from sample_factory.algo.utils.context import global_model_factory
from sample_factory.model.encoder import Encoder

from . import policy

class QuadEncoder(Encoder):
    def __init__(self, cfg, obs_space):
        super().__init__(cfg)
        # the reference's factory, flag for flag (widths, nonlinearity, encoder class): quad_multi_model.py:250-370
        self.body = policy.encoder_from_cfg(cfg)
        self.encoder_out_size = self.body.feed_forward[0].out_features

    def forward(self, obs_dict):
        return self.body(obs_dict["obs"])

    def get_out_size(self):
        return self.encoder_out_size


def make_quadmulti_encoder(cfg, obs_space):
    return QuadEncoder(cfg, obs_space)


def register_models():
    global_model_factory().register_encoder_factory(make_quadmulti_encoder)
