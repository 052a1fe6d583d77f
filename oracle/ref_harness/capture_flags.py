#!/usr/bin/env python
"""Build-container only (needs /root/reference): dumps the reference's `--quads_*` command-line surface
(swarm_rl/env_wrappers/quadrotor_params.py: names, defaults, choices, and the defaults it overrides) to tests/golden/flags.json.
Sample Factory's `str2bool` is the only import the module needs; stubbed with the usual truthy-string parser."""
import argparse
import json
import os
import sys
import types

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, "/root/reference")


def str2bool(v):
    if isinstance(v, bool):
        return v
    if isinstance(v, str) and v.lower() in ("true", "1", "yes", "y", "t"):
        return True
    if isinstance(v, str) and v.lower() in ("false", "0", "no", "n", "f"):
        return False
    raise argparse.ArgumentTypeError("Boolean value expected")


for name in ("sample_factory", "sample_factory.utils"):
    sys.modules[name] = types.ModuleType(name)
m = types.ModuleType("sample_factory.utils.utils")
m.str2bool = str2bool
sys.modules["sample_factory.utils.utils"] = m
for name in ("swarm_rl", "swarm_rl.env_wrappers"):      # import the one file, not the package __init__ (which pulls in Sample Factory)
    pkg = types.ModuleType(name)
    pkg.__path__ = [os.path.join("/root/reference", *name.split("."))]
    sys.modules[name] = pkg
from swarm_rl.env_wrappers import quadrotor_params as ref   # noqa: E402

p = argparse.ArgumentParser()
ref.add_quadrotors_env_args("quadrotor_multi", p)
flags = {}
for a in p._actions:
    if a.dest == "help":
        continue
    flags[a.dest] = {"default": a.default, "choices": list(a.choices) if a.choices else None, "nargs": a.nargs,
                     "type": getattr(a.type, "__name__", None) if a.type else None}
p2 = argparse.ArgumentParser()
for k in ("encoder_type", "encoder_subtype", "rnn_size", "encoder_extra_fc_layers", "env_frameskip"):
    p2.add_argument("--" + k, default=None)
ref.quadrotors_override_defaults("quadrotor_multi", p2)
overrides = {k: v for k, v in vars(p2.parse_args([])).items()}
path = os.path.join(REPO, "tests", "golden", "flags.json")
json.dump({"flags": flags, "override_defaults": overrides}, open(path, "w"), indent=0, sort_keys=True)
print("wrote", path, len(flags), "flags")
