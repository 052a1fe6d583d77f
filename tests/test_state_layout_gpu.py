"""The two element orders of the wave-blocked state arrays (include/quadswarm.h, `qs_buffers`): which handles get which, that the
raw device bytes follow the documented formula for both, that `qs_state_array_copy` (Stepper.to_host / from_host) hides the order, and that the
same rollout gives the same state whatever the order (the 8-wave team kernels on lane-major blocks against the single-wave kernels on rows).
Configurations of tests/test_hip_parity.py whose code objects `__graft_entry__.build()` prebuilds."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import test_hip_parity as thp  # noqa: E402
from quad_swarm_rl_amd import config as qcfg  # noqa: E402

pytestmark = pytest.mark.gpu

ARRAYS = [("pos", 3), ("vel", 3), ("omega", 3), ("rot", 9), ("thrust_rot_damp", 4), ("thrust_cmds_damp", 4), ("ou_state", 4), ("goal", 3), ("flags", 1), ("col_pair_mask", 1)]


def make(case, E, precision="f32"):
    from quad_swarm_rl_amd import native
    cfg = qcfg.make_config(num_envs=E, seed=77, env_id_offset=2, precision=precision, **thp.CASES[case])
    return native.Stepper(cfg, device=0)


def raw_blocks(st, name):
    """The bytes of one state array as they lie on the device: [blocks, 64 lanes x comps] elements (block pitch = state_block_bytes)."""
    from quad_swarm_rl_amd import native
    shape, kind = st._shapes[name]
    comps = shape[0] if len(shape) == 2 else 1
    dt = np.dtype(st._dtype(kind))
    epb, pitch = st.bufs.envs_per_block, st.bufs.state_block_bytes
    nblk = (st.E + epb - 1) // epb
    span = (nblk - 1) * pitch + 64 * comps * dt.itemsize
    buf = np.empty(span, dtype=np.uint8)
    native._check(native.lib().qs_memcpy_d2h(st._h, buf.ctypes.data_as(C.c_void_p), C.c_void_p(st.ptr(name)), span))
    return np.stack([buf[b * pitch: b * pitch + 64 * comps * dt.itemsize].view(dt) for b in range(nblk)]), comps


def by_formula(st, name):
    """[comps, E * N] from the raw blocks through the element formula of include/quadswarm.h."""
    blocks, comps = raw_blocks(st, name)
    epb, N = st.bufs.envs_per_block, st.N
    out = np.empty((comps, st.T), dtype=blocks.dtype)
    for e in range(st.E):
        for i in range(N):
            lane = (e % epb) * N + i
            for c in range(comps):
                out[c, e * N + i] = blocks[e // epb, lane * comps + c] if st.bufs.state_lane_major else blocks[e // epb, c * 64 + lane]
    return out


@pytest.mark.parametrize("case,team,lane_major", [("c2_n8_dw", None, 1), ("c3_n8_obst", None, 1), ("c2_n8_dw", "0", 0), ("c4_n32_svs", None, 0)])
def test_which_handles_are_lane_major(case, team, lane_major, monkeypatch):
    if team is not None:
        monkeypatch.setenv("QS_TEAM", team)
    st = make(case, 8)
    assert st.specialized
    assert st.bufs.state_lane_major == lane_major, (case, team, st.bufs.state_lane_major)
    assert st.bufs.state_block_bytes == 40 * 64 * st.real_size + 64 * 4 + 64 * 8 and st.bufs.envs_per_block == 64 // st.N
    st.close()


@pytest.mark.parametrize("case,team", [("c2_n8_dw", None), ("c2_n8_dw", "0"), ("c4_n32_svs", None), ("c2_n5_kall_short", None)])
def test_raw_blocks_follow_the_documented_formula(case, team, monkeypatch):
    """After a reset and a few steps every state array read raw from the device and unscrambled by the header's formula equals what
    qs_state_array_copy hands out; a host array written through it reads back the same, and lands where the formula says."""
    if team is not None:
        monkeypatch.setenv("QS_TEAM", team)
    st = make(case, 11)   # (11 environments: a partial last block for every drone count here)
    rng = np.random.RandomState(5)
    st.reset()
    for _ in range(3):
        st.from_host("actions", rng.uniform(-1, 1, size=(st.T, 4)))
        st.step()
    st.sync()
    for name, comps in ARRAYS:
        got = st.to_host(name).reshape(comps, st.T)
        assert np.array_equal(by_formula(st, name), got), (case, team, name)
    for name, comps in ARRAYS[:8]:
        new = rng.standard_normal((comps, st.T)).astype(st.np_real)
        st.from_host(name, new)
        assert np.array_equal(st.to_host(name).reshape(comps, st.T), new), name
        assert np.array_equal(by_formula(st, name), new), name
    st.close()


def test_the_same_rollout_on_both_orders(monkeypatch):
    """8-wave team kernels on lane-major blocks against the single-wave kernels on rows: the same state after 40 float64 steps (<= 1e-9, flags and
    pair masks exact; tests/test_hip_parity.py runs every case through both flavours against the oracle)."""
    rng = np.random.RandomState(9)
    acts = rng.uniform(-1, 1, size=(40, 8 * 8, 4))
    states = {}
    for team in (None, "0"):
        if team is not None:
            monkeypatch.setenv("QS_TEAM", team)
        st = make("c2_n8_dw", 8, precision="f64")
        st.reset()
        for t in range(40):
            st.from_host("actions", acts[t])
            st.step()
        st.sync()
        states[team] = (st.bufs.state_lane_major, {name: st.to_host(name).copy() for name, _ in ARRAYS})
        st.close()
    assert states[None][0] == 1 and states["0"][0] == 0
    for name, _ in ARRAYS:
        a, b = states[None][1][name], states["0"][1][name]
        if a.dtype.kind == "f":   # (two code shapes of the same float64 arithmetic: contraction into FMAs may differ)
            assert np.allclose(a, b, rtol=0, atol=1e-9), (name, float(np.abs(a - b).max()))
        elif name == "flags":   # bits 12 / 13 are the kernels' private bookkeeping (distance ring live, new-pair word non-zero: DESIGN.md 3 / 4a) and
            keep = np.uint32(~((1 << 12) | (1 << 13)) & 0xffffffff)   # differ between the flavours by design
            assert np.array_equal(a & keep, b & keep), name
        else:
            assert np.array_equal(a, b), name
