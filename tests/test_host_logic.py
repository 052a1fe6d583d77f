"""Host-side mirror of the reference interface: configuration semantics, constants, observation layout."""
import json

import numpy as np
import pytest

from quad_swarm_rl_amd import airframe, config as qcfg
from tests import golden_util as gu


def test_constants_match_reference_G1():
    g, _ = gu.load("c2_n8_random")
    const = json.loads(str(g["const"]))
    af = airframe.crazyflie()
    assert af["mass"] == const["mass"] and af["arm"] == const["arm"]
    np.testing.assert_array_equal(af["inertia"], const["inertia"])
    np.testing.assert_array_equal(af["prop_pos"], const["prop_pos"])
    np.testing.assert_array_equal(af["prop_cross"], const["prop_crossproducts"])
    np.testing.assert_array_equal(af["thrust_max"], const["thrust_max"])
    np.testing.assert_array_equal(af["torque_max"], const["torque_max"])
    assert af["motor_tau_up"] == const["motor_tau_up"] and af["motor_tau_down"] == const["motor_tau_down"]
    c = gu.config_from_golden(json.loads(str(g["cfg"])))
    assert c.collision_threshold == const["collision_threshold"]
    assert c.collision_falloff_threshold == const["collision_falloff_threshold"]
    assert c.ep_len == const["ep_len"] and qcfg.config_obs_dim(c) == const["obs_dim"]


@pytest.mark.parametrize("name", ["c1_single_numpy", "c2_n8_random", "c2_n8_kall", "c3_n8_obst", "c4_n32_svs"])
def test_observation_space_bounds(name):
    g, cfgd = gu.load(name)
    const = json.loads(str(g["const"]))
    low, high = qcfg.obs_bounds(gu.config_from_golden(cfgd))
    np.testing.assert_array_equal(low, np.array(const["obs_low"], dtype=np.float32))
    np.testing.assert_array_equal(high, np.array(const["obs_high"], dtype=np.float32))


def test_obs_dims():
    # C2 54, C3 40, C1 18
    assert qcfg.obs_dim("xyz_vxyz_R_omega", 6, False) == 54
    assert qcfg.obs_dim("xyz_vxyz_R_omega_floor", 2, True) == 40
    assert qcfg.obs_dim("xyz_vxyz_R_omega", 0, False) == 18
    assert qcfg.obs_dim("xyz_vxyz_R_omega_wall", 7, False) == 24 + 42


def test_reference_error_behaviour():
    with pytest.raises(RuntimeError, match="Incorrect number of neigbors"):       # quadrotor_multi.py:274
        qcfg.make_config(num_agents=4, neighbor_visible_num=5, neighbor_obs_type="pos_vel")
    with pytest.raises(NotImplementedError):                                      # unknown scenario (mix.py:31 eval fails)
        qcfg.make_config(quads_mode="no_such_scenario")
    with pytest.raises(AssertionError):                                           # quadrotor_multi.py:99
        qcfg.make_config(rew_coeff=dict(not_a_coeff=1.0))
    with pytest.raises(ValueError):
        qcfg.make_config(num_agents=65)


def test_flag_semantics():
    c = qcfg.make_config(num_agents=8, neighbor_visible_num=-1, neighbor_obs_type="pos_vel")
    assert c.num_neighbors == 7                                                   # -1 = all (quadrotor_multi.py:47-50)
    c = qcfg.make_config(num_agents=8, neighbor_visible_num=6, neighbor_obs_type="none")
    assert c.num_neighbors == 0
    c = qcfg.make_config(use_numba=True)
    assert c.floor_mode == 0
    c = qcfg.make_config(use_numba=False)
    assert c.floor_mode == 1
    c = qcfg.make_config(use_obstacles=True, quads_mode="o_static_same_goal", obst_spawn_area=(8.0, 8.0), obst_density=0.2)
    assert c.num_obstacles == 12 and c.spawn_box == 0.1 and c.approach_goal_metric == 1.0
    assert qcfg.svd_period(0.005) == 100     # fl(sum of 100 x 0.005) = 0.5000000000000003 > 0.5
    assert qcfg.make_config(ep_time=15.0).ep_len == 1500


def test_scenario_table_matches_header():
    """Every quads_mode of the reference's training configs is accepted, and the ids are the header's enum."""
    import os
    import re
    hdr = open(os.path.join(os.path.dirname(__file__), "..", "include", "quadswarm.h")).read()
    ids = {m.group(1).lower(): int(m.group(2)) for m in re.finditer(r"QS_SCENARIO_([A-Z0-9_]+) = (\d+)", hdr)}
    count = ids.pop("count")
    assert count == len(qcfg.SCENARIOS) == 16
    for name, sid in qcfg.SCENARIOS.items():
        assert ids[name.lower()] == sid
        obst = name.startswith("o_")
        c = qcfg.make_config(num_agents=8, neighbor_visible_num=2, neighbor_obs_type="pos_vel", quads_mode=name,
                             use_obstacles=obst, obst_spawn_area=(8.0, 8.0), obst_density=0.2)
        assert c.scenario == sid
        assert qcfg.SCENARIO_CLASS_NAMES[sid] == "Scenario_" + name
    # obstacle scenarios need obstacles and vice versa (quadrotor_multi.py:118-125, scenarios/mix.py:16-29)
    with pytest.raises((AssertionError, ValueError, NotImplementedError)):
        qcfg.make_config(quads_mode="o_random", use_obstacles=False)


def test_bench_workloads_follow_the_baseline_and_the_cpu_leg_runs():
    """bench.py: the headline workload is BASELINE.json's (8 drones x 1024 envs), the algorithmic bytes per drone-control-step are
    SURVEY 8(d)'s, and the cpu_baseline leg (the only part of bench.py that may touch the oracle) produces a sane record."""
    import json
    import os
    import bench
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    base = json.load(open(os.path.join(repo, "BASELINE.json")))
    assert "8 drones" in base["metric"] and "1024" in base["metric"]
    c2 = bench.WORKLOADS["c2"]
    assert c2["num_envs"] == 1024 and c2["kw"]["num_agents"] == 8 and c2["kw"]["neighbor_visible_num"] == 6
    assert bench.ALGO_BYTES_PER_DRONE_STEP == {"c2": 500, "c3": 456, "c4": 500, "c1": 356}
    assert bench.HBM_PEAK_GBS == 8000.0
    rec = bench.cpu_baseline("c1", 0.2)
    assert rec["kind"] == "port" and rec["unit"] == "env-steps/s" and rec["value"] > 0 and rec["cores"] >= 1 and "C oracle" in rec["sample"]


def test_shard_spec_of_a_torchrun_process():
    """--quads_num_gpus: contiguous env ranges per rank, GPU = LOCAL_RANK, and a clear error when the launch does not match"""
    from quad_swarm_rl_amd import sf_env
    assert sf_env.shard_spec(1024, 1, {}) == (1024, 0, 0, None)
    env = dict(WORLD_SIZE="8", RANK="3", LOCAL_RANK="3")
    assert sf_env.shard_spec(4096, 8, env) == (512, 1536, 3, 3)
    covered = []
    for r in range(8):
        per, off, rank, lr = sf_env.shard_spec(4096, 8, dict(WORLD_SIZE="8", RANK=str(r), LOCAL_RANK=str(r)))
        covered += list(range(off, off + per))
    assert covered == list(range(4096))
    with pytest.raises(ValueError, match="torch.distributed.run"):
        sf_env.shard_spec(4096, 8, {})
    with pytest.raises(ValueError, match="divisible"):
        sf_env.shard_spec(1001, 8, env)


def test_backend_flag_has_no_cpu_fallback():
    import argparse
    from quad_swarm_rl_amd import sf_env
    p = argparse.ArgumentParser()
    sf_env.add_quadrotors_env_args(None, p)
    cfg = p.parse_args(["--quads_backend=cpu"])
    with pytest.raises(NotImplementedError, match="no CPU simulator"):
        sf_env.make_quadrotor_env("quadrotor_multi", cfg)
    cfg = p.parse_args(["--quads_num_gpus=2"])
    with pytest.raises(ValueError, match="quads_num_envs"):
        sf_env.make_quadrotor_env("quadrotor_multi", cfg)


def test_q8_wire_reference_layout_and_error_bound():
    """QS_WIRE_Q8 (include/quadswarm_exchange.h) by its plain-torch specification on the CPU: row bytes of the BASELINE shapes (<= 80, what
    the >= 6x of north_star needs: DESIGN.md 7), section layout, round-half-even, clamping, and the clip / 254 error bound after dequantisation."""
    import torch
    from quad_swarm_rl_amd import config as qcfg, native, parallel
    cfg = qcfg.make_config(num_envs=1, num_agents=8, neighbor_visible_num=6, neighbor_obs_type="pos_vel")
    q8 = native.wire_q8_layout(cfg, 54)
    assert (q8.q0, q8.q1) == (18, 54) and list(q8.clip) == [10.0, 10.0, 10.0, 6.0, 6.0, 6.0]
    g = torch.Generator().manual_seed(0)
    x = (torch.rand((257, 54), generator=g) * 2 - 1) * 11.0
    x[0, 18] = 10.0 * 0.5 / 127          # exactly half a step: rounds to the even neighbour 0
    x[0, 19] = 10.0 * 1.5 / 127          # 1.5 steps -> 2
    x[0, 21] = 7.0                       # beyond the velocity clip of 6 -> 127
    w = parallel.quantize_rows_reference(x, "q8", q8)
    assert w.shape == (257, 72) and w.dtype == torch.uint8
    q = w[:, 36:].view(torch.int8)
    assert (q[0, 0].item(), q[0, 1].item(), q[0, 3].item()) == (0, 2, 127)
    assert torch.equal(w[:, :36].contiguous().view(torch.bfloat16).reshape(257, 18), x[:, :18].to(torch.bfloat16))
    clip = torch.tensor([q8.clip[a % 6] for a in range(36)])
    deq = q.float() * clip / 127.0
    assert ((deq - x[:, 18:].clamp(-clip, clip)).abs() <= clip / 254 + 1e-6).all()
    # C3: 19 self + 12 neighbour + 9 SDF columns -> 28 bf16 + 12 int8 = 68 bytes; an odd bf16 count is padded to a whole word
    cfg3 = qcfg.make_config(num_envs=1, num_agents=8, neighbor_visible_num=2, neighbor_obs_type="pos_vel", use_obstacles=True, obst_density=0.2, obst_size=0.6,
                            quads_mode="o_static_same_goal", obs_repr="xyz_vxyz_R_omega_floor")
    q3 = native.wire_q8_layout(cfg3, 40)
    w3 = parallel.quantize_rows_reference(torch.zeros((3, 40)), "q8", q3)
    assert w3.shape == (3, 68)
    q1 = native.WireQ8(); q1.q0, q1.q1 = 19, 25
    for a in range(6):
        q1.clip[a] = 1.0
    assert parallel.quantize_rows_reference(torch.ones((2, 25)), "q8", q1).shape == (2, 2 * 20 + 8)   # 19 -> 20 bf16 slots, 6 -> 8 int8 slots
    assert parallel.wire_row_bytes(54, "q8", q8) == 72 and parallel.wire_row_bytes(40, "q8", q3) == 68 and parallel.wire_row_bytes(25, "q8", q1) == 48
