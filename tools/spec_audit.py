#!/usr/bin/env python
"""Which config-specialised code objects does the `-m gpu` test suite ask for?  (no GPU needed)

    python tools/spec_audit.py [pytest args...]        -> tests/spec_audit.json

Runs the GPU-marked tests HERE with `native.Stepper.__init__` replaced by a recorder: the qs_config the test would have created its
handle with (and the QS_TEAM / QS_SPEC environment at that moment) is written down, then the test is skipped.  Every distinct
(configuration, kernel flavour) is then resolved to its code object with qs_spec_build - the ones `__graft_entry__.build()` does not
know yet are compiled on first use ON THE GPU BOX (hipcc, 10-30 s each, inside the suite's wall time).  `__graft_entry__._spec_jobs`
reads tests/spec_audit.json, so that build() compiles them ahead of time and keeps them when it prunes the cache.
Tests that start their handles in a subprocess (tests/xchg_worker.py, bench.py) are not seen; their configurations are listed in
_spec_jobs by hand.
"""
import base64
import ctypes as C
import json
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
OUT = os.path.join(REPO, "tests", "spec_audit.json")


def team_request():
    """the flavour qs_create would pick under the current environment: 0 / 4 / 8 forced by QS_TEAM, -1 = its default rule"""
    ev = os.environ.get("QS_TEAM", "")
    if ev[:1] == "0":
        return 0
    if ev[:1] in ("4", "8"):
        return int(ev[:1])
    if ev[:1] == "1":
        return 1
    return -1


class Recorder:
    def __init__(self):
        self.seen = {}

    def pytest_configure(self, config):
        import pytest
        from quad_swarm_rl_amd import native
        rec = self

        def fake_init(self_, cfg, device=0):
            if os.environ.get("QS_SPEC", "") not in ("off", "0"):
                key = (bytes(cfg), team_request())
                rec.seen.setdefault(key, os.environ.get("PYTEST_CURRENT_TEST", "?").split(" ")[0])
            self_._h = C.c_void_p()

        def fake_getattr(self_, name):   # a test may create several handles before it uses one: the skip comes with the first use
            if name in ("_h", "close", "__del__", "__class__", "__dict__"):
                return object.__getattribute__(self_, name)
            pytest.skip("spec audit: configuration recorded")

        native.Stepper.__getattribute__ = fake_getattr
        native.Stepper.__init__ = fake_init
        # tests that move a policy / a tensor to the GPU before they create their handle: let them get as far as the handle
        import torch
        from quad_swarm_rl_amd import policy
        torch.nn.Module.cuda = lambda self_, *a, **k: self_
        torch.Tensor.cuda = lambda self_, *a, **k: self_
        policy.FusedQuadEncoder.__init__ = lambda self_, *a, **k: None
        from quad_swarm_rl_amd import rollout
        rollout.GaussianActionHead.__init__ = lambda self_, *a, **k: None

        def on_cpu(fn):
            def wrapped(*a, **k):
                k.pop("device", None)
                return fn(*a, **k)
            return wrapped
        for name in ("rand", "randn", "zeros", "ones", "empty", "full", "tensor", "arange", "as_tensor", "randint"):
            setattr(torch, name, on_cpu(getattr(torch, name)))
        gen = torch.Generator
        torch.Generator = lambda device=None: gen()


def resolve(item):
    from quad_swarm_rl_amd import config as qcfg, native
    raw, team = item
    cfg = qcfg.QsConfig.from_buffer_copy(raw)
    try:
        return native.spec_build(cfg, team)
    except native.QsError as exc:
        return f"!{exc}"


def main():
    import pytest
    from quad_swarm_rl_amd import native
    cache = os.path.join(native.CSRC, "spec_cache")
    before = set(os.listdir(cache)) if os.path.isdir(cache) else set()
    rec = Recorder()
    pytest.main(["tests", "-m", "gpu", "-q", "-p", "no:cacheprovider", "--no-header", "-rN"] + sys.argv[1:], plugins=[rec])
    items = sorted(rec.seen, key=lambda k: rec.seen[k])
    print(f"{len(items)} distinct (qs_config, flavour) requests")
    from concurrent.futures import ProcessPoolExecutor
    with ProcessPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as ex:
        paths = list(ex.map(resolve, items))
    by_path = {}
    for it, p in zip(items, paths):
        if p.startswith("!"):
            print("no specialised object:", rec.seen[it], p[1:])
            continue
        by_path.setdefault(os.path.basename(p), (it, rec.seen[it]))
    new = sorted(f for f in by_path if f not in before)
    print(f"{len(by_path)} code objects, {len(new)} of them were not in the cache before this run")
    for f in new:
        print("  new:", f, "first asked for by", by_path[f][1])
    # only what the hand-written list of __graft_entry__._spec_jobs does not produce anyway goes into the file
    import __graft_entry__ as ge
    with ProcessPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as ex:
        hand = {os.path.basename(p) for p in ex.map(ge._spec_one, ge._spec_jobs(with_audit=False)) if p}
    print(f"{len(hand)} objects come from _spec_jobs' own list; {len(set(by_path) - hand)} more are asked for by the tests")
    by_path = {f: v for f, v in by_path.items() if f not in hand}
    doc = {"note": "written by tools/spec_audit.py: the qs_config structs (base64) + kernel flavour the -m gpu tests create handles with; "
                   "read by __graft_entry__._spec_jobs so that build() compiles their code objects ahead of time",
           "sizeof_qs_config": C.sizeof(__import__("quad_swarm_rl_amd.config", fromlist=["QsConfig"]).QsConfig),
           "objects": [{"config": base64.b64encode(it[0]).decode(), "team": it[1], "first_test": t} for f, (it, t) in sorted(by_path.items())]}
    json.dump(doc, open(OUT, "w"), indent=0)
    print("wrote", OUT, len(doc["objects"]), "objects")


if __name__ == "__main__":
    main()
