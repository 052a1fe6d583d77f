"""Importable alias of the `quad-swarm-rl_amd/` package directory.

The product sources live in `quad-swarm-rl_amd/` (a hyphenated name cannot be imported), so this
shim package simply extends its search path to that directory:
`import quad_swarm_rl_amd.env` loads `quad-swarm-rl_amd/env.py`.
"""
import os as _os

__path__.append(_os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "quad-swarm-rl_amd"))

__version__ = "0.1.0"
