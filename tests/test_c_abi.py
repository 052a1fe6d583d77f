"""CPU-side checks of the drop-in boundary: the C-ABI library loads without a GPU, exports every symbol that
include/quadswarm.h declares, agrees with the ctypes mirror of qs_config, and fails loudly (no CPU fallback)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from quad_swarm_rl_amd import config as qcfg, native

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_text():
    return open(os.path.join(REPO, "include", "quadswarm.h")).read()


def test_library_exports_every_declared_symbol():
    native.build()
    lib = C.CDLL(native.LIB_PATH)
    declared = sorted(set(re.findall(r"^(?:int|size_t|const char \*)\s*\*?(qs_\w+)\(", header_text(), flags=re.M)))
    assert len(declared) >= 20, declared
    for sym in declared:
        assert hasattr(lib, sym), f"{sym} declared in include/quadswarm.h but not exported"
    assert sorted(native.EXPORTED_SYMBOLS) == declared


def test_library_exports_every_symbol_of_the_exchange_header():
    native.build()
    lib = C.CDLL(native.LIB_PATH)
    text = open(os.path.join(REPO, "include", "quadswarm_exchange.h")).read()
    declared = sorted(set(re.findall(r"^(?:int|int64_t|void \*|const char \*)\s*\*?(qs_\w+)\(", text, flags=re.M)))
    assert len(declared) >= 12, declared
    for sym in declared:
        assert hasattr(lib, sym), f"{sym} declared in include/quadswarm_exchange.h but not exported"
    assert sorted(native.EXCHANGE_SYMBOLS) == declared
    from quad_swarm_rl_amd import parallel
    m = re.search(r"#define QS_XCHG_EXPORT_BYTES \(2 \* QS_XCHG_HANDLE_BYTES \+ (\d+)\)", text)
    assert parallel.EXPORT_BYTES == 2 * int(re.search(r"#define QS_XCHG_HANDLE_BYTES (\d+)", text).group(1)) + int(m.group(1))


def test_config_struct_layout_matches():
    L = native.lib()
    assert L.qs_sizeof_config() == C.sizeof(qcfg.QsConfig)
    assert L.qs_version() == int(re.search(r"#define QS_VERSION (\d+)", header_text()).group(1))


def test_enums_match_header():
    h = header_text()
    assert int(re.search(r"#define QS_STATE_STRIDE (\d+)", h).group(1)) == qcfg.QS_STATE_STRIDE
    assert int(re.search(r"#define QS_MAX_AGENTS (\d+)", h).group(1)) == qcfg.QS_MAX_AGENTS
    ri = re.search(r"enum \{ QS_RI_REW_MAIN = 0,(.*?)QS_RI_COUNT \};", h, flags=re.S).group(1)
    assert ri.count(",") + 1 == len(qcfg.REW_INFO_KEYS)
    cnt = re.search(r"enum \{ QS_CNT_COLLISIONS = 0,(.*?)QS_CNT_COUNT \};", h, flags=re.S).group(1)
    assert cnt.count(",") + 1 == len(qcfg.COUNTER_KEYS)


def test_default_config_equals_host_derivation():
    """qs_default_config (C, Appendix-C constants) == make_config (python, derived from the link geometry)."""
    L = native.lib()
    c = qcfg.QsConfig()
    assert L.qs_default_config(C.byref(c), 4, 8) == 0
    ref = qcfg.make_config(num_envs=4, num_agents=8, neighbor_visible_num=6, neighbor_obs_type="pos_vel", use_numba=True,
                           collision_falloff_radius=4.0, quads_mode="static_same_goal")
    for name in ("mass", "arm", "motor_tau_up", "motor_tau_down", "omega_max", "dt", "thrust_noise_sigma", "ou_theta",
                 "collision_threshold", "collision_falloff_threshold", "spawn_box", "approach_goal_metric"):
        assert getattr(c, name) == pytest.approx(getattr(ref, name), rel=1e-15), name
    for name in ("ep_len", "sim_steps", "svd_period", "num_neighbors", "floor_mode", "obs_repr", "sense_noise"):
        assert getattr(c, name) == getattr(ref, name), name
    np.testing.assert_allclose(list(c.inertia), list(ref.inertia), rtol=1e-14)
    np.testing.assert_allclose(np.array(c.prop_cross), np.array(ref.prop_cross), atol=1e-17)
    np.testing.assert_allclose(list(c.thrust_max), list(ref.thrust_max), rtol=1e-14)
    np.testing.assert_allclose(list(c.torque_max), list(ref.torque_max), rtol=1e-14)
    assert L.qs_obs_dim(C.byref(c)) == 54


@pytest.mark.skipif(os.path.exists("/dev/kfd"), reason="GPU box: covered by the gpu tests")
def test_no_cpu_fallback_without_gpu():
    """The product path must fail loudly when no GPU is visible."""
    cfg = qcfg.make_config(num_envs=2, num_agents=2)
    with pytest.raises(native.QsError):
        native.Stepper(cfg)
    from quad_swarm_rl_amd import env
    with pytest.raises(native.QsError):
        env.QuadSwarmVecEnv(2, num_agents=2)


def test_invalid_arguments_rejected_before_touching_the_gpu():
    L = native.lib()
    h = C.c_void_p()
    cfg = qcfg.make_config(num_envs=2, num_agents=4)
    cfg.num_neighbors = 7
    assert L.qs_create(C.byref(cfg), 0, C.byref(h)) == -1
    assert b"neigbors" in L.qs_last_error()      # the reference's RuntimeError text (quadrotor_multi.py:274)
    cfg = qcfg.make_config(num_envs=2, num_agents=4)
    cfg.scenario = 16            # QS_SCENARIO_COUNT: anything unknown is unsupported
    assert L.qs_create(C.byref(cfg), 0, C.byref(h)) == -4
    assert L.qs_step(None, None, None) == -1


def test_spec_build_without_gpu(tmp_path, monkeypatch):
    """qs_spec_build compiles the config-specialised code object on a machine without a GPU; the cache key follows the
    configuration constants, not the run-time arguments (reward coefficients, seed, env offset, env count)."""
    monkeypatch.setenv("QS_SPEC_CACHE", str(tmp_path))
    kw = dict(num_agents=4, neighbor_visible_num=2, neighbor_obs_type="pos_vel", use_numba=True)
    p1 = native.spec_build(qcfg.make_config(num_envs=16, seed=1, **kw))
    assert p1.startswith(str(tmp_path)) and p1.endswith(".hsaco") and os.path.getsize(p1) > 10000
    t1 = os.path.getmtime(p1)
    p2 = native.spec_build(qcfg.make_config(num_envs=64, seed=99, env_id_offset=5, rew_coeff=dict(quadcol_bin=3.0), **kw))
    assert p2 == p1 and os.path.getmtime(p2) == t1                    # cache hit: nothing rebuilt
    p3 = native.spec_build(qcfg.make_config(num_envs=16, seed=1, **dict(kw, neighbor_visible_num=3)))
    assert p3 != p1                                                   # another configuration, another object
    assert native.spec_build(qcfg.make_config(num_envs=16, seed=1, **kw), team=0) != p1   # kernel flavour is part of the key
    blob = open(p1, "rb").read()
    for sym in (b"qs_spec_step", b"qs_spec_rollout", b"qs_spec_reset"):
        assert sym in blob


def test_encoder_library_exports_and_layout():
    """include/quadswarm_encoder.h: every declared symbol is exported by libquadswarm_encoder.so, qs_enc_params has the layout
    of the ctypes mirror, bad arguments are rejected before anything touches a GPU, and there is no CPU fallback."""
    from quad_swarm_rl_amd import policy
    policy.build()
    lib = C.CDLL(policy.ENC_LIB_PATH)
    h = open(os.path.join(REPO, "include", "quadswarm_encoder.h")).read()
    declared = sorted(set(re.findall(r"^(?:int|size_t|const char \*)\s*\*?(qs_enc_\w+)\(", h, flags=re.M)))
    assert declared == ["qs_enc_benchmark", "qs_enc_forward", "qs_enc_last_error", "qs_enc_lds_bytes", "qs_enc_lds_bytes_of", "qs_enc_lds_bytes_split",
                        "qs_enc_sizeof_params"]
    for sym in declared:
        assert hasattr(lib, sym), sym
    lib.qs_enc_sizeof_params.restype = C.c_size_t
    lib.qs_enc_lds_bytes.restype = C.c_size_t
    lib.qs_enc_last_error.restype = C.c_char_p
    assert lib.qs_enc_sizeof_params() == C.sizeof(policy.EncParams)
    assert lib.qs_enc_lds_bytes() <= 80 * 1024            # two workgroups per CU
    # the multi-head / Sim2Real kernel body never shares a CU: its LDS request is more than half of the 160 KiB
    lib.qs_enc_lds_bytes_of.restype = C.c_size_t
    lib.qs_enc_lds_bytes_of.argtypes = [C.c_int32]
    for model in (4, 5):
        assert 80 * 1024 < lib.qs_enc_lds_bytes_of(model) <= 160 * 1024
    for model in (0, 1, 2, 3):
        assert lib.qs_enc_lds_bytes_of(model) <= 80 * 1024
    # reference precision (fp16 pairs): two LDS planes, one workgroup per CU
    lib.qs_enc_lds_bytes_split.restype = C.c_size_t
    lib.qs_enc_lds_bytes_split.argtypes = [C.c_int32]
    assert 2 * lib.qs_enc_lds_bytes_of(0) <= lib.qs_enc_lds_bytes_split(0) <= lib.qs_enc_lds_bytes_split(1) <= 160 * 1024
    P = policy.EncParams()
    P.precision = 2
    lib.qs_enc_last_error.restype = C.c_char_p
    lib.qs_enc_forward.argtypes = [C.c_void_p, C.c_int32, C.POINTER(policy.EncParams), C.c_void_p, C.c_void_p]
    enum = re.search(r"enum \{ QS_ENC_NBR_MEAN_EMBED = 0,(.*?)\};", h, flags=re.S).group(0)
    for i, name in enumerate(("MEAN_EMBED", "ATTENTION", "MLP", "NONE")):
        assert f"QS_ENC_NBR_{name} = {i}" in enum and policy.MODELS[i] == policy.NBR_ENCODERS[i]
    assert "QS_ENC_MODEL_MHA = 4" in enum and policy.MODELS[4] == "multi_head_attention"
    assert "QS_ENC_MODEL_S2R = 5" in enum and policy.MODELS[5] == "single_head_sim2real"
    lib.qs_enc_forward.argtypes = [C.c_void_p, C.c_int32, C.POINTER(policy.EncParams), C.c_void_p, C.c_void_p]
    P = policy.EncParams()
    assert lib.qs_enc_forward(None, 4, C.byref(P), None, None) == -1 and b"bad argument" in lib.qs_enc_last_error()
    P.num_nbr = 9                                          # more neighbours than a workgroup's row tiles
    assert lib.qs_enc_forward(C.c_void_p(16), 4, C.byref(P), C.c_void_p(16), None) == -4
    P.num_nbr, P.nbr_encoder = 2, 7
    assert lib.qs_enc_forward(C.c_void_p(16), 4, C.byref(P), C.c_void_p(16), None) == -4
    import torch
    if not torch.cuda.is_available():
        with pytest.raises(native.QsError):
            policy.FusedQuadEncoder(policy.make_reference_encoder(num_nbr=2))
        from quad_swarm_rl_amd import rollout
        with pytest.raises(native.QsError):
            rollout.GraphedRollout(None, None, None, 4)


def test_reference_encoder_modules_cover_the_flag_choices():
    """--quads_neighbor_encoder_type choices (quadrotor_params.py:38-40) and both encoder classes build and run on the CPU in torch."""
    import torch
    from quad_swarm_rl_amd import policy
    for enc in policy.NBR_ENCODERS:
        m = policy.make_reference_encoder(num_nbr=2, obst_dim=9, self_dim=19, nbr_encoder=enc)
        assert m(torch.zeros(3, 40)).shape == (3, 512)
    with pytest.raises(NotImplementedError):
        policy.make_reference_encoder(nbr_encoder="transformer")
    assert policy.make_reference_mha_encoder()(torch.zeros(3, 40)).shape == (3, 512)
    assert policy.make_reference_sim2real_encoder()(torch.zeros(3, 40)).shape == (3, 256)


def test_weight_packing_follows_the_header_formula():
    """policy.pack_linear == the fragment order include/quadswarm_encoder.h documents:
    w[((mt * (K/32) + ks) * 64 + lane) * 8 + j] = bf16(W[mt*16 + (lane & 15)][ks*32 + 8*(lane >> 4) + j]), zero padded to M % 16 == 0, K % 32 == 0."""
    import torch
    from quad_swarm_rl_amd import policy
    for m_real, k_real in ((256, 18), (256, 256), (1, 256), (512, 768), (256, 48)):
        lin = torch.nn.Linear(k_real, m_real)
        w, b, M, K = policy.pack_linear(lin, "cpu")
        assert M == -(-m_real // 16) * 16 and K == -(-k_real // 32) * 32 and w.dtype == torch.bfloat16 and w.numel() == M * K
        W = torch.zeros(M, K)
        W[:m_real, :k_real] = lin.weight.detach()
        flat = w.float().reshape(-1)
        rng = np.random.RandomState(m_real + k_real)
        for _ in range(200):
            mt, ks, lane, j = rng.randint(M // 16), rng.randint(K // 32), rng.randint(64), rng.randint(8)
            want = W[mt * 16 + (lane & 15), ks * 32 + 8 * (lane >> 4) + j].to(torch.bfloat16).float()
            assert flat[((mt * (K // 32) + ks) * 64 + lane) * 8 + j] == want
        assert torch.equal(b[:m_real], lin.bias.detach()) and (b[m_real:] == 0).all()
    half = policy.pack_linear(torch.nn.Linear(512, 256), "cpu", cols=(256, 512))   # the e_mean half of the attention score layer
    assert half[3] == 256


def test_spec_cache_key_covers_every_source_the_specialised_object_includes():
    """The config-specialised code objects are cached under a hash of their sources (spec_key, quadswarm_hip.hip).  A file that
    qs_spec_kernels.hip pulls in but the key does not hash would let an edit of that file run stale objects (qs_step_sem.h was such a
    file for part of round 3)."""
    import re
    csrc = os.path.join(REPO, "quad-swarm-rl_amd", "csrc")
    host = open(os.path.join(csrc, "quadswarm_hip.hip")).read()
    listed = set(re.findall(r'"([^"]+)"', re.search(r"kSpecSources\[\] = \{([^}]*)\}", host).group(1)))
    seen, todo = set(), ["qs_spec_kernels.hip"]
    while todo:
        name = todo.pop()
        if name in seen:
            continue
        seen.add(name)
        for inc in re.findall(r'#include "([^"]+)"', open(os.path.join(csrc, name)).read()):
            if os.path.exists(os.path.join(csrc, inc)):
                todo.append(inc)
    listed |= {"../../include/" + n for n in re.findall(r'"(quadswarm[a-z_]*\.h)"', host[host.index("static bool spec_key("):host.index("static std::string spec_cache_dir()")])}
    missing = {n for n in seen if not n.startswith("spec_cache")} - listed
    assert not missing, f"not part of the spec cache key: {sorted(missing)}"
