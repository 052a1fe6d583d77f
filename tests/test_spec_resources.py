"""Register / scratch footprint of the config-specialised bench kernels (hipcc cross-compiles for gfx950 without a GPU).

A regression guard, not a tuning tool: a run-time index into the kernel-constant block once moved the whole block to scratch
memory (564 bytes per lane) and turned the 8 us C2 step into 21 us without failing a single parity test.  The throughput
(single-wave) kernels must stay at <= 128 VGPRs: that is what lets 4 waves share a SIMD.  The team kernels are
scheduled for ILP (one wave per SIMD with 4 waves per workgroup: 512 registers to spend; two with 8 waves: 256)."""
import os
import sys

import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))


@pytest.mark.parametrize("workload,team,max_vgpr,max_scratch", [("c2", 8, 176, 0), ("c2", 0, 128, 0), ("c4", 4, 256, 128), ("c4", 0, 128, 64)])
def test_no_scratch_and_register_budget(workload, team, max_vgpr, max_scratch, tmp_path):
    import spec_resources
    res, _ = spec_resources.resources(workload, team, out=str(tmp_path / "k.s"))
    step = res["qs_spec_step"]
    # (the 128-register cap of the throughput kernels may spill a few dwords at N = 32; the swarm_vs_swarm goal swap of the team kernel - a cold
    # block that calls two out-of-line functions - keeps its arguments in 96 bytes of stack; the constant block would be 560)
    assert step["scratch"] <= max_scratch, step
    assert step["next_free_vgpr"] <= max_vgpr, step
    assert res["qs_spec_reset"]["scratch"] == 0, res["qs_spec_reset"]


@pytest.mark.parametrize("workload,team,max_vgpr,max_scratch", [("c2", 8, 224, 0), ("c3", 8, 224, 0), ("c2", 0, 128, 64)])   # (8 waves: 256 to spend; round 5 keeps 12 loaded values for the row-skipping store)
def test_full_scenario_kernels_keep_the_constant_block_out_of_scratch(workload, team, max_vgpr, max_scratch, tmp_path):
    """`--quads_mode mix` (train_local.sh) runs the full-scenario kernels.  Their per-episode scenario code once stayed out of line in the
    specialised objects and took the constant block by reference: 552 bytes of scratch per lane, every literal a memory load, 22 us per
    step for EVERY scenario of that set against 9 us for the same shape on static_same_goal (profiles/history/r03i_scenario_times_before.txt)."""
    import spec_resources
    res, _ = spec_resources.resources(workload, team, out=str(tmp_path / "k.s"), quads_mode="mix")
    assert res["qs_spec_step"]["scratch"] <= max_scratch, res["qs_spec_step"]
    assert res["qs_spec_step"]["next_free_vgpr"] <= max_vgpr, res["qs_spec_step"]
    assert res["qs_spec_reset"]["scratch"] == 0, res["qs_spec_reset"]


def _step_kernel_body(path):
    text = open(path).read()
    a = text.index("qs_spec_step:")
    return text[a:text.index("s_endpgm", a)]


def test_state_accesses_are_wide_on_lane_major_blocks_and_narrow_on_rows(tmp_path):
    """the specialised 8-wave team kernel moves a lane-major state block with one 12- / 16-byte buffer access per array and lane
    (5 x dwordx3 + 6 x dwordx4 + flags + pair mask per direction in float32); the single-wave throughput kernel of the same configuration and the
    4-wave team kernel keep the 4-byte rows.  Read off the ISA: a change that silently falls back to 45 narrow accesses (or widens the
    throughput kernels, which measured 1-3 % slower with it) fails here, without a GPU."""
    import re
    import spec_resources
    _, path = spec_resources.resources("c2", 8, out=str(tmp_path / "team8.s"))
    body = _step_kernel_body(path)
    loads = re.findall(r"\bbuffer_load_(dword(?:x\d)?)\b", body)
    stores = re.findall(r"\bbuffer_store_(dword(?:x\d)?)\b", body)
    assert loads.count("dwordx3") == 5 and loads.count("dwordx4") == 6 and loads.count("dword") <= 3, sorted(loads)
    assert stores.count("dwordx3") == 5 and stores.count("dwordx4") == 6, sorted(stores)
    for wl, team in (("c2", 0), ("c4", 4)):
        _, path = spec_resources.resources(wl, team, out=str(tmp_path / f"{wl}_{team}.s"))
        body = _step_kernel_body(path)
        wide = re.findall(r"\bbuffer_(?:load|store)_dwordx[34]\b", body)
        assert not wide, (wl, team, wide)
        assert len(re.findall(r"\bbuffer_load_dword\b", body)) >= 37, (wl, team)
