"""The code-object verifier of the library (include/quadswarm.h: qs_spec_verify), on the CPU: hipcc cross-compiles and
llvm-objdump disassembles without a GPU.

Round 5 ended with one float32 parity case of the single-wave kernels failing under a scheduler flag and nobody knowing why.  Round 6 found
it on the GPU (rocgdb: the environment index of `p.counters[q * E + e]` held a float) and in the ISA: a VGPR spill in front of the
`s_or_b64 exec, exec, s[0:1]` of a join block - ROCm 7.2's register allocator, not the kernel source.  These tests pin the defence:
the scanner flags exactly that object, the repair (the exec restore moved in front of the spills) makes it clean by permuting a few dozen bytes,
qs_spec_build() never hands out an object that shows the pattern, and what the build step left in the cache (and the libraries themselves) is
verified."""
import glob
import os

import pytest

from quad_swarm_rl_amd import config as qcfg, native
from tests import test_hip_parity as thp

TRACKERS = "-mllvm -amdgpu-use-amdgpu-trackers"


def n17_cfg():
    return qcfg.make_config(num_envs=7, precision="f32", **thp.CASES["e_n17_kall_obst"])


def test_the_scanner_flags_round5s_object(tmp_path, monkeypatch):
    """the single-wave float32 object of e_n17_kall_obst built with the RP-tracker flag: `scratch_store ...; s_or_b64 exec, exec, s[..]` at a join"""
    monkeypatch.setenv("QS_SPEC_CACHE", str(tmp_path))
    monkeypatch.setenv("QS_SPEC_SINGLE_FLAGS", TRACKERS)
    monkeypatch.setenv("QS_SPEC_VERIFY", "0")          # take the object as the compiler delivers it
    path = native.spec_build(n17_cfg(), 0)
    assert not os.path.exists(path.replace(".hsaco", ".ok"))
    rc, report = native.spec_verify(path)
    assert rc == 1, "the compiler no longer produces the pattern for this object: re-derive the test case (tools/spec_hazard.py over a cache)"
    assert "qs_spec_step <L" in report and "scratch_store_" in report and report.rstrip().endswith("]") and "s_or_b64 exec, exec, s[" in report


def test_the_repair_moves_the_restore_in_front_of_the_spills(tmp_path, monkeypatch):
    """qs_spec_repair on round 5's object: same bytes, the `s_or_b64 exec` first in its block prologue - and nothing else changed"""
    monkeypatch.setenv("QS_SPEC_CACHE", str(tmp_path))
    monkeypatch.setenv("QS_SPEC_SINGLE_FLAGS", TRACKERS)
    monkeypatch.setenv("QS_SPEC_VERIFY", "0")
    path = native.spec_build(n17_cfg(), 0)
    before = open(path, "rb").read()
    fixed, left = native.spec_repair(path)
    assert fixed >= 1 and left == "", (fixed, left)
    after = open(path, "rb").read()
    assert native.spec_verify(path)[0] == 0
    assert len(after) == len(before) and sorted(before) == sorted(after)                       # a permutation of the same bytes ...
    diff = [k for k in range(len(before)) if before[k] != after[k]]
    assert 0 < len(diff) and diff[-1] - diff[0] < 256 * fixed                                    # ... inside a few block prologues
    assert native.spec_repair(path) == (0, "")                                                   # nothing left to do


def test_a_flagged_object_is_repaired_or_rebuilt_before_it_is_used(tmp_path, monkeypatch):
    monkeypatch.setenv("QS_SPEC_CACHE", str(tmp_path))
    monkeypatch.setenv("QS_SPEC_SINGLE_FLAGS", TRACKERS)
    path = native.spec_build(n17_cfg(), 0)             # verification on (the default)
    stamp = open(path.replace(".hsaco", ".ok")).read()
    assert "exec restores moved" in stamp or "the configured ones were rejected" in stamp, stamp
    assert native.spec_verify(path)[0] == 0
    # ... and an unverified object that got into the cache (QS_SPEC_VERIFY=0 above, an older tree) is checked when it is asked for
    os.remove(path.replace(".hsaco", ".ok"))
    assert native.spec_build(n17_cfg(), 0) == path and os.path.exists(path.replace(".hsaco", ".ok"))


def test_the_libraries_and_the_prebuilt_cache_are_clean():
    from quad_swarm_rl_amd import policy
    for lib in (native.LIB_PATH, policy.ENC_LIB_PATH):
        if os.path.exists(lib):
            rc, report = native.spec_verify(lib)
            assert rc == 0, f"{lib}:\n{report}"
    cache = os.path.join(native.CSRC, "spec_cache")
    objs = sorted(glob.glob(os.path.join(cache, "*.hsaco")))
    if not objs:
        pytest.skip("no prebuilt cache (run __graft_entry__.build())")
    missing = [o for o in objs if not os.path.exists(o.replace(".hsaco", ".ok"))]
    assert not missing, f"{len(missing)} cached objects without a verification stamp, e.g. {missing[:3]}"
    for o in objs[::max(1, len(objs) // 10)]:          # a sample, re-checked here (the build verified each when it wrote the stamp)
        assert native.spec_verify(o)[0] == 0, o
