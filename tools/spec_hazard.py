#!/usr/bin/env python
"""Scan code objects / libraries for the compiler defect of DESIGN.md 5.3 - a VGPR spill, reload, copy or rematerialised constant at the top of a
control-flow join block IN FRONT OF the `s_or_b64 exec, exec, s[..]` that restores exec there - with the library's own checker (qs_spec_verify,
include/quadswarm.h; no GPU needed):

    python tools/spec_hazard.py [--repair] [cache dir | object | library ...]      default: the spec cache and the two libraries

One line per file: clean / HAZARD + the block prologues found; --repair applies qs_spec_repair first (in place) and reports what was moved.
profiles/r06e_spec_hazard_scan_of_the_r05_style_cache.txt is this scan over a cache built the way round 5 built it (no check): 18 of 269 objects.
"""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from quad_swarm_rl_amd import native  # noqa: E402


def main():
    args = [a for a in sys.argv[1:] if a != "--repair"]
    repair = "--repair" in sys.argv[1:]
    if not args:
        args = [os.path.join(native.CSRC, "spec_cache"), native.LIB_PATH, os.path.join(native.CSRC, "libquadswarm_encoder.so")]
    paths = []
    for a in args:
        paths += sorted(os.path.join(a, f) for f in os.listdir(a) if f.endswith((".hsaco", ".so"))) if os.path.isdir(a) else [a]
    bad = 0
    for p in paths:
        note = ""
        if repair:
            fixed, left = native.spec_repair(p)
            note = f"  ({fixed} exec restore(s) moved" + (f"; not repaired:\n{left}" if left else "") + ")" if fixed or left else ""
        rc, report = native.spec_verify(p)
        bad += rc
        print(os.path.basename(p), "clean" if rc == 0 else "HAZARD", note)
        for line in report.splitlines():
            print("    " + line)
    print(f"{len(paths)} files, {bad} with a VGPR spill / copy in front of an exec restore")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
