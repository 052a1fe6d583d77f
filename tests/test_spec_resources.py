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
