"""ctypes binding of the HIP stepper's C ABI (include/quadswarm.h -> csrc/libquadswarm_hip.so).

The product path: there is NO CPU fallback here.  If the HIP extension is missing, cannot be built, or no
GPU is visible, the constructor raises.
"""
import ctypes as C
import os
import sys
import subprocess

import numpy as np

from . import config as qcfg

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB_PATH = os.environ.get("QS_LIB", os.path.join(CSRC, "libquadswarm_hip.so"))   # QS_LIB: A/B builds (tools only)
SOURCES = [os.path.join(CSRC, "quadswarm_hip.hip"), os.path.join(CSRC, "qs_tape_kernels.hip"), os.path.join(CSRC, "qs_exchange.hip"),
           os.path.join(CSRC, "qs_kernels.h"), os.path.join(CSRC, "qs_step_kernel.inc"), os.path.join(CSRC, "qs_step_team.inc"),
           os.path.join(CSRC, "qs_device.h"),
           os.path.join(CSRC, "qs_scenarios.h"),
           os.path.join(os.path.dirname(HERE), "include", "quadswarm.h"), os.path.join(os.path.dirname(HERE), "include", "quadswarm_exchange.h")]

QS_OK = 0
QS_ERR_NAN_REWARD = -3


class QsBuffers(C.Structure):
    _fields_ = [(name, C.c_void_p) for name in (
        "obs", "reward", "done", "rew_info", "actions", "pos", "vel", "omega", "rot", "thrust_rot_damp",
        "thrust_cmds_damp", "ou_state", "goal", "flags", "obst_hit_idx", "col_pair_mask", "new_pair_mask",
        "unique_col_mask", "obst_new_mask", "room_new_mask", "counters", "tick", "obst_pos", "ep_stats",
        "ep_counters", "error_flag", "scenario_id", "ep_scenario", "run_sums", "ep_sums", "obst_count", "obst_size_env", "obst_density_env")] + [
        ("obs_dim", C.c_int32), ("real_size", C.c_int32), ("state_block_bytes", C.c_int32), ("envs_per_block", C.c_int32), ("state_lane_major", C.c_int32)]


def build(force=False, verbose=False):
    """Compile the HIP extension in-tree for gfx950 (hipcc cross-compiles without a GPU)."""
    newest = max(os.path.getmtime(s) for s in SOURCES)
    if not force and os.path.exists(LIB_PATH) and os.path.getmtime(LIB_PATH) >= newest:
        return LIB_PATH
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    base = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC"]
    objs = [os.path.join(CSRC, "quadswarm_hip.o"), os.path.join(CSRC, "qs_tape_kernels.o"), os.path.join(CSRC, "qs_exchange.o")]
    # the noise-tape flavour replays the reference's float64 arithmetic: no FMA contraction there (NumPy has none)
    cmds = [base + ["-c", SOURCES[0], "-o", objs[0]], base + ["-ffp-contract=off", "-c", SOURCES[1], "-o", objs[1]],
            base + ["-c", SOURCES[2], "-o", objs[2]],
            [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB_PATH] + objs]
    procs = []
    for cmd in cmds[:-1]:
        if verbose:
            print(" ".join(cmd))
        procs.append(subprocess.Popen(cmd))
    for pr, cmd in zip(procs, cmds[:-1]):
        if pr.wait() != 0:
            raise subprocess.CalledProcessError(pr.returncode, cmd)
    if verbose:
        print(" ".join(cmds[-1]))
    subprocess.check_call(cmds[-1])
    for o in objs:
        os.remove(o)
    finish_library(LIB_PATH, verbose)
    return LIB_PATH


def finish_library(path, verbose=False):
    """Every freshly built library goes through the code-object check of include/quadswarm.h: exec restores that the compiler
    placed behind a VGPR spill / copy at a join block are moved in front of it (qs_spec_repair), and a library that still shows the pattern
    afterwards is not kept.  (The checker lives in libquadswarm_hip.so itself - host code; when that library is the one being built, it is
    loaded from the file just written.)"""
    checker = C.CDLL(LIB_PATH)
    buf = C.create_string_buffer(1 << 16)
    checker.qs_spec_repair.argtypes = checker.qs_spec_verify.argtypes = [C.c_char_p, C.c_char_p, C.c_int]
    fixed = checker.qs_spec_repair(path.encode(), buf, len(buf))
    left = buf.value.decode()
    if verbose or fixed or left:
        print(f"{os.path.basename(path)}: {fixed} misplaced exec restore(s) repaired" + (f"; left:\n{left}" if left else ""))
    rc = checker.qs_spec_verify(path.encode(), buf, len(buf))
    if rc != 0:
        os.replace(path, path + ".rejected")
        raise RuntimeError(f"{path}: VGPR spill / copy in front of an exec restore that could not be repaired, or the check could not run "
                           f"(kept as .rejected):\n{buf.value.decode()}\n{left}")


_lib = None


class GateInfo(C.Structure):
    """qs_gate_info_t (include/quadswarm.h): the action ring and the sequence words of resident-state stepping"""
    _fields_ = [("action_ring", C.c_void_p), ("action_stride_bytes", C.c_int64), ("ring_len", C.c_int32),
                ("groups", C.c_int32), ("wg_per_group", C.c_int32), ("workgroups", C.c_int32), ("envs_per_workgroup", C.c_int32),
                ("act_flag", C.c_void_p), ("done_flag", C.c_void_p), ("steps_launched", C.c_int64), ("steps_fed", C.c_int64)]


class WireQ8(C.Structure):
    """qs_wire_q8 (include/quadswarm_exchange.h): the 8-bit fixed-point block [q0, q1) of an observation row and its clip ranges"""
    _fields_ = [("q0", C.c_int32), ("q1", C.c_int32), ("clip", C.c_float * 6)]


def wire_q8_layout(cfg, obs_dim):
    """The QS_WIRE_Q8 layout of configuration `cfg`: the neighbour block (6 columns per visible neighbour) behind the self observation,
    clipped by the environment to +-nbr_clip_pos / +-nbr_clip_vel (quadrotor_single.py:294-295)."""
    self_dim = 18 if cfg.obs_repr == 0 else (19 if cfg.obs_repr == 1 else 24)
    q = WireQ8()
    q.q0, q.q1 = self_dim, self_dim + 6 * cfg.num_neighbors
    assert q.q1 <= obs_dim
    for a in range(3):
        q.clip[a], q.clip[3 + a] = cfg.nbr_clip_pos[a], cfg.nbr_clip_vel[a]
    return q


def _preload_torch_hip_runtime():
    """PyTorch-ROCm wheels bundle their own libamdhip64 / libhsa-runtime64 (same SONAMEs as /opt/rocm's).  One process
    must run ONE HIP runtime: if this library pulled in the system copy first, a later `import torch` would come up with
    "No HIP GPUs are available".  So, when torch is installed, its copy is loaded first (without importing torch) and
    libquadswarm_hip.so binds to it - the same arrangement as when the caller imported torch before us."""
    import importlib.util
    if "torch" in sys.modules:
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.submodule_search_locations:
        return
    libdir = os.path.join(list(spec.submodule_search_locations)[0], "lib")
    for name in ("libhsa-runtime64.so", "libamdhip64.so"):
        path = os.path.join(libdir, name)
        if os.path.exists(path):
            try:
                C.CDLL(path, mode=C.RTLD_GLOBAL)
            except OSError:
                return


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build()
        _preload_torch_hip_runtime()
        L = C.CDLL(LIB_PATH)
        vp = C.c_void_p
        L.qs_version.restype = C.c_int
        L.qs_sizeof_config.restype = C.c_size_t
        L.qs_last_error.restype = C.c_char_p
        L.qs_default_config.argtypes = [C.POINTER(qcfg.QsConfig), C.c_int32, C.c_int32]
        L.qs_obs_dim.argtypes = [C.POINTER(qcfg.QsConfig)]
        L.qs_create.argtypes = [C.POINTER(qcfg.QsConfig), C.c_int, C.POINTER(vp)]
        L.qs_destroy.argtypes = [vp]
        L.qs_reset.argtypes = [vp, C.POINTER(C.c_uint8), vp]
        L.qs_step.argtypes = [vp, vp, vp]
        L.qs_step_many.argtypes = [vp, vp, C.c_int32, vp]
        L.qs_sync.argtypes = [vp, vp]
        L.qs_get_buffers.argtypes = [vp, C.POINTER(QsBuffers)]
        L.qs_set_reward_coeffs.argtypes = [vp, C.POINTER(C.c_double)]
        L.qs_get_state.argtypes = [vp, C.c_int32, C.POINTER(C.c_double), C.POINTER(C.c_int32)]
        L.qs_set_state.argtypes = [vp, C.c_int32, C.POINTER(C.c_double), C.c_int32]
        L.qs_memcpy_d2h.argtypes = [vp, vp, vp, C.c_size_t]
        L.qs_memcpy_h2d.argtypes = [vp, vp, vp, C.c_size_t]
        L.qs_state_array_copy.argtypes = [vp, vp, vp, C.c_int32, C.c_int32, C.c_int32]
        L.qs_check_errors.argtypes = [vp]
        L.qs_set_profiling.argtypes = [vp, C.c_int32]
        L.qs_get_kernel_time.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(C.c_int64)]
        L.qs_spec_build.argtypes = [C.POINTER(qcfg.QsConfig), C.c_int, C.c_char_p, C.c_int]
        L.qs_spec_verify.argtypes = [C.c_char_p, C.c_char_p, C.c_int]
        L.qs_spec_repair.argtypes = [C.c_char_p, C.c_char_p, C.c_int]
        L.qs_is_specialized.argtypes = [vp]
        L.qs_spec_status.argtypes = [vp, C.c_char_p, C.c_int]
        L.qs_kernel_flavor.argtypes = [vp]
        L.qs_snapshot_pool.argtypes = [vp, C.c_int32]
        L.qs_snapshot_save.argtypes = [vp, C.c_int32, C.c_int32, vp]
        L.qs_snapshot_load.argtypes = [vp, C.c_int32, C.c_int32, vp]
        L.qs_snapshot_copy.argtypes = [vp, C.c_int32, C.c_int32, vp]
        L.qs_replay_enable.argtypes = [vp, C.c_double]
        L.qs_replay_stats.argtypes = [vp, C.POINTER(C.c_int32)]
        L.qs_replay_set_active.argtypes = [vp, C.POINTER(C.c_uint8)]
        L.qs_set_noise_tape.argtypes = [vp, C.POINTER(C.c_double), C.c_int64]
        L.qs_get_tape_pos.argtypes = [vp, C.POINTER(C.c_int32)]
        L.qs_set_tape_pos.argtypes = [vp, C.POINTER(C.c_int32)]
        L.qs_gate_create.argtypes = [vp, C.c_int32, C.c_int32]
        L.qs_gate_info.argtypes = [vp, C.POINTER(GateInfo)]
        L.qs_step_gated.argtypes = [vp, C.c_int32, vp]
        L.qs_gate_wait.argtypes = [vp, vp]
        L.qs_gate_produce.argtypes = [vp, vp, C.c_int32, C.c_int32, C.c_int32, vp]
        L.qs_gate_produce_verify.argtypes = [vp, vp, C.c_int32, C.c_int32, vp, vp]
        L.qs_gate_status.argtypes = [vp, C.POINTER(C.c_int64)]
        L.qs_set_obs_target.argtypes = [vp, vp]
        L.qs_set_obs_exchange.argtypes = [vp, vp, C.c_int32]
        L.qs_xchg_fused_desc.argtypes = [vp, C.c_int32, C.c_int32, C.POINTER(C.c_int64)]
        L.qs_xchg_fused_desc.restype = vp
        # observation exchange between env shards (include/quadswarm_exchange.h)
        L.qs_xchg_last_error.restype = C.c_char_p
        L.qs_xchg_create.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int64, C.c_int32, C.c_int, C.POINTER(vp)]
        L.qs_xchg_destroy.argtypes = [vp]
        L.qs_xchg_export.argtypes = [vp, vp]
        L.qs_xchg_attach.argtypes = [vp, vp]
        L.qs_xchg_attach_local.argtypes = [vp, C.c_int, vp]
        L.qs_xchg_staging.argtypes = [vp, C.c_int]
        L.qs_xchg_staging.restype = vp
        L.qs_xchg_gathered.argtypes = [vp, C.c_int]
        L.qs_xchg_gathered.restype = vp
        L.qs_xchg_push.argtypes = [vp, vp, vp]
        L.qs_xchg_wait.argtypes = [vp, vp]
        L.qs_xchg_release.argtypes = [vp, vp]
        L.qs_xchg_wait_release.argtypes = [vp, vp]
        L.qs_xchg_status.argtypes = [vp, C.POINTER(C.c_int64)]
        L.qs_obs_pack.argtypes = [vp, vp, C.c_int64, C.c_int, vp]
        L.qs_xchg_create_q8.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int64, C.c_int32, C.POINTER(WireQ8), C.POINTER(vp)]
        L.qs_wire_row_bytes.argtypes = [C.c_int32, C.c_int, C.POINTER(WireQ8)]
        L.qs_wire_row_bytes.restype = C.c_int64
        L.qs_obs_pack_rows.argtypes = [vp, vp, C.c_int64, C.c_int32, C.c_int, C.POINTER(WireQ8), vp]
        L.qs_obs_unpack_rows.argtypes = [vp, vp, C.c_int64, C.c_int32, C.c_int, C.POINTER(WireQ8), vp]
        if L.qs_sizeof_config() != C.sizeof(qcfg.QsConfig):
            raise RuntimeError("qs_config layout mismatch between config.py and libquadswarm_hip.so")
        _lib = L
    return _lib


EXPORTED_SYMBOLS = ["qs_spec_verify", "qs_spec_repair", "qs_version", "qs_sizeof_config", "qs_last_error", "qs_default_config", "qs_obs_dim", "qs_create",
                    "qs_destroy", "qs_reset", "qs_step", "qs_step_many", "qs_sync", "qs_get_buffers", "qs_set_reward_coeffs",
                    "qs_get_state", "qs_set_state", "qs_memcpy_d2h", "qs_memcpy_h2d", "qs_state_array_copy", "qs_check_errors", "qs_set_profiling",
                    "qs_get_kernel_time", "qs_spec_build", "qs_is_specialized", "qs_spec_status", "qs_kernel_flavor",
                    "qs_snapshot_pool", "qs_snapshot_save", "qs_snapshot_load", "qs_snapshot_copy",
                    "qs_set_noise_tape", "qs_get_tape_pos", "qs_set_tape_pos", "qs_gate_create", "qs_gate_info", "qs_step_gated", "qs_gate_wait", "qs_gate_produce", "qs_gate_produce_verify", "qs_gate_status", "qs_replay_enable", "qs_replay_stats", "qs_replay_set_active", "qs_set_obs_target", "qs_set_obs_exchange"]
# include/quadswarm_exchange.h
EXCHANGE_SYMBOLS = ["qs_xchg_create", "qs_xchg_destroy", "qs_xchg_export", "qs_xchg_attach", "qs_xchg_attach_local", "qs_xchg_staging",
                    "qs_xchg_gathered", "qs_xchg_push", "qs_xchg_wait", "qs_xchg_release", "qs_xchg_wait_release", "qs_xchg_fused_desc", "qs_xchg_status", "qs_obs_pack", "qs_xchg_last_error",
                    "qs_xchg_create_q8", "qs_wire_row_bytes", "qs_obs_pack_rows", "qs_obs_unpack_rows", "qs_xchg_set_fenced", "qs_xchg_get_fenced"]


class QsError(RuntimeError):
    pass


def spec_verify(path):
    """(0 | 1, report): include/quadswarm.h qs_spec_verify - 1 = the code object / library has a VGPR spill or copy in front of an exec restore"""
    buf = C.create_string_buffer(1 << 16)
    rc = lib().qs_spec_verify(str(path).encode(), buf, len(buf))
    if rc < 0:
        raise QsError(lib().qs_last_error().decode())
    return rc, buf.value.decode()


def spec_repair(path):
    """(places repaired, report of the ones left): include/quadswarm.h qs_spec_repair"""
    buf = C.create_string_buffer(1 << 16)
    rc = lib().qs_spec_repair(str(path).encode(), buf, len(buf))
    if rc < 0:
        raise QsError(lib().qs_last_error().decode())
    return rc, buf.value.decode()


def spec_build(cfg, team=-1):
    """Ahead-of-time build of the config-specialised code object of `cfg` (hipcc --genco; no GPU needed).
    Returns the path of the cached .hsaco.  team: 1 = 4-wave kernels, 0 = single-wave, -1 = qs_create's default."""
    buf = C.create_string_buffer(4096)
    rc = lib().qs_spec_build(C.byref(cfg), team, buf, len(buf))
    if rc != QS_OK:
        raise QsError(f"qs_spec_build failed ({rc}): {lib().qs_last_error().decode()}")
    return buf.value.decode()


def _check(rc):
    if rc != QS_OK:
        msg = lib().qs_last_error().decode()
        if rc == QS_ERR_NAN_REWARD:
            raise ValueError("QuadEnv: reward is Nan")   # quadrotor_single.py:87-90
        raise QsError(f"quadswarm error {rc}: {msg}")


class _DevArray:
    """Zero-copy view of a library-owned device buffer for torch.as_tensor (CUDA array interface)."""

    def __init__(self, ptr, shape, typestr):
        self.__cuda_array_interface__ = dict(shape=tuple(shape), typestr=typestr, data=(int(ptr), False), version=2,
                                             strides=None)


class Stepper:
    """E envs x N drones stepped by the HIP kernels.  Thin, allocation-free per step."""

    def __init__(self, cfg, device=0):
        self.cfg = cfg
        self.device = device
        self._h = C.c_void_p()
        _check(lib().qs_create(C.byref(cfg), device, C.byref(self._h)))
        self.bufs = QsBuffers()
        _check(lib().qs_get_buffers(self._h, C.byref(self.bufs)))
        self.E, self.N = cfg.num_envs, cfg.num_agents
        self.T = self.E * self.N
        self.obs_dim = self.bufs.obs_dim
        self.real_size = self.bufs.real_size
        self.np_real = np.float64 if self.real_size == 8 else np.float32
        self._shapes = dict(
            obs=((self.T, self.obs_dim), "real"), reward=((self.T,), "real"), done=((self.T,), "u1"),
            rew_info=((17, self.T), "real"), actions=((self.T, 4), "real"),
            pos=((3, self.T), "real"), vel=((3, self.T), "real"), omega=((3, self.T), "real"), rot=((9, self.T), "real"),
            thrust_rot_damp=((4, self.T), "real"), thrust_cmds_damp=((4, self.T), "real"), ou_state=((4, self.T), "real"),
            goal=((3, self.T), "real"), flags=((self.T,), "u4"), obst_hit_idx=((self.T,), "i4"),
            col_pair_mask=((self.T,), "u8"), new_pair_mask=((self.T,), "u8"), unique_col_mask=((self.E,), "u8"),
            obst_new_mask=((self.E,), "u8"), room_new_mask=((self.E,), "u8"), counters=((11, self.E), "i4"),
            tick=((self.E,), "i4"), obst_pos=((2, self.E * max(cfg.num_obstacles, 1)), "real"),
            ep_stats=((6, self.T), "real"), ep_counters=((11, self.E), "i4"), error_flag=((1,), "u4"),
            scenario_id=((self.E,), "i4"), ep_scenario=((self.E,), "i4"),
            run_sums=((25, self.T), "real"), ep_sums=((25, self.T), "real"),
            obst_count=((self.E,), "i4"), obst_size_env=((self.E,), "real"), obst_density_env=((self.E,), "real"))
        self._torch_cache = {}

    # ---- lifecycle -------------------------------------------------------------------------
    def close(self):
        if self._h:
            lib().qs_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- stepping ----------------------------------------------------------------------------
    @staticmethod
    def _stream_ptr(stream):
        if stream is None:
            return None
        return C.c_void_p(int(getattr(stream, "cuda_stream", stream)))

    def reset(self, env_mask=None, stream=None):
        m = None
        if env_mask is not None:
            arr = np.ascontiguousarray(env_mask, dtype=np.uint8)
            assert arr.size == self.E
            m = arr.ctypes.data_as(C.POINTER(C.c_uint8))
        _check(lib().qs_reset(self._h, m, self._stream_ptr(stream)))

    def step(self, actions_ptr=None, stream=None):
        """actions_ptr: device address of real[T,4] (int) or None to use the staging buffer `actions`."""
        _check(lib().qs_step(self._h, C.c_void_p(actions_ptr) if actions_ptr else None, self._stream_ptr(stream)))

    def step_many(self, actions_ptr, k, stream=None):
        _check(lib().qs_step_many(self._h, C.c_void_p(actions_ptr), k, self._stream_ptr(stream)))

    def sync(self, stream=None):
        _check(lib().qs_sync(self._h, self._stream_ptr(stream)))

    def set_obs_exchange(self, xchg_handle, auto_ack=True):
        """Fused exchange: every step launch also stores its observation rows into all ranks' windows (None switches it off)."""
        _check(lib().qs_set_obs_exchange(self._h, xchg_handle, 1 if auto_ack else 0))

    def set_obs_target(self, ptr):
        """Observation rows of the following reset / step launches go to device address `ptr` (None = the library's `obs` buffer)."""
        _check(lib().qs_set_obs_target(self._h, C.c_void_p(int(ptr)) if ptr else None))

    def check_errors(self):
        _check(lib().qs_check_errors(self._h))

    def set_reward_coeffs(self, coeffs):
        arr = (C.c_double * 8)(*[float(x) for x in coeffs])
        _check(lib().qs_set_reward_coeffs(self._h, arr))

    def set_profiling(self, enable):
        _check(lib().qs_set_profiling(self._h, int(enable)))

    # ---- environment snapshots (include/quadswarm.h) ------------------------------------------------
    def snapshot_pool(self, slots):
        _check(lib().qs_snapshot_pool(self._h, slots))

    def snapshot_save(self, env, slot, stream=None):
        _check(lib().qs_snapshot_save(self._h, env, slot, self._stream_ptr(stream)))

    def snapshot_load(self, slot, env, stream=None):
        _check(lib().qs_snapshot_load(self._h, slot, env, self._stream_ptr(stream)))

    def snapshot_copy(self, src_slot, dst_slot, stream=None):
        _check(lib().qs_snapshot_copy(self._h, src_slot, dst_slot, self._stream_ptr(stream)))

    # ---- batched experience replay on the device (include/quadswarm.h) --------------------------------------
    def replay_enable(self, sample_prob):
        _check(lib().qs_replay_enable(self._h, float(sample_prob)))
        self.replay_on = True

    def replay_stats(self):
        """dict of per-env int arrays: episodes, replayed, buffer_len, replayed_sum, active, checkpoints, errors, ep_was_replay, ep_steps"""
        out = np.zeros((9, self.E), dtype=np.int32)
        _check(lib().qs_replay_stats(self._h, out.ctypes.data_as(C.POINTER(C.c_int32))))
        return dict(zip(("episodes", "replayed", "buffer_len", "replayed_sum", "active", "checkpoints", "errors", "ep_was_replay", "ep_steps"), out))

    def replay_set_active(self, active=None):
        """override the activation rule of the replay buffers (None = all on)"""
        a = None if active is None else np.ascontiguousarray(active, dtype=np.uint8).ctypes.data_as(C.POINTER(C.c_uint8))
        _check(lib().qs_replay_set_active(self._h, a))

    # ---- noise tape (test instrument, include/quadswarm.h) ---------------------------------------------
    def set_noise_tape(self, tape):
        """tape: [E, len] float64 reference draws per environment (None = back to the Philox stream)."""
        if tape is None:
            _check(lib().qs_set_noise_tape(self._h, None, 0))
            return
        a = np.ascontiguousarray(tape, dtype=np.float64).reshape(self.E, -1)
        _check(lib().qs_set_noise_tape(self._h, a.ctypes.data_as(C.POINTER(C.c_double)), a.shape[1]))

    def tape_pos(self):
        out = np.zeros(self.E, dtype=np.int32)
        _check(lib().qs_get_tape_pos(self._h, out.ctypes.data_as(C.POINTER(C.c_int32))))
        return out

    # ---- resident-state stepping (include/quadswarm.h: qs_gate_*) -------------------------------------
    def gate_create(self, ring_len=64, wg_per_group=8):
        _check(lib().qs_gate_create(self._h, ring_len, wg_per_group))

    def gate_info(self):
        out = GateInfo()
        _check(lib().qs_gate_info(self._h, C.byref(out)))
        return out

    def step_gated(self, k, stream=None):
        """ONE launch for k control steps: per step it waits for the step's actions in the gate's ring and publishes its outputs"""
        _check(lib().qs_step_gated(self._h, int(k), self._stream_ptr(stream)))

    def gate_wait(self, stream=None):
        """order `stream` behind the last gated launch (call it after the producer's work has been issued)"""
        _check(lib().qs_gate_wait(self._h, self._stream_ptr(stream)))

    def gate_produce(self, src_ptr, n_src, k, closed_loop=False, stream=None):
        """the trivial producer: k steps, action batches round-robin from a table of n_src batches at device address src_ptr"""
        _check(lib().qs_gate_produce(self._h, C.c_void_p(int(src_ptr)), int(n_src), int(k), 1 if closed_loop else 0, self._stream_ptr(stream)))

    def gate_produce_verify(self, src_ptr, n_src, k, sums_ptr, stream=None):
        """closed-loop producer that also consumes the stepper's outputs concurrently: sums[t * groups + g] = sum of the 32-bit words of the
        observation rows and rewards of step t as the producer saw them behind done_flag + an agent-scope acquire (test instrument)"""
        _check(lib().qs_gate_produce_verify(self._h, C.c_void_p(int(src_ptr)), int(n_src), int(k), C.c_void_p(int(sums_ptr)), self._stream_ptr(stream)))

    def gate_status(self):
        out = (C.c_int64 * 4)()
        _check(lib().qs_gate_status(self._h, out))
        return dict(error=int(out[0]), steps_launched=int(out[1]), min_act_flag=int(out[2]), min_done_flag=int(out[3]))

    def set_tape_pos(self, pos):
        """move the per-environment tape cursors (teacher forcing from a fixture's recorded tape positions)"""
        a = np.ascontiguousarray(np.broadcast_to(np.asarray(pos, dtype=np.int32), (self.E,)))
        _check(lib().qs_set_tape_pos(self._h, a.ctypes.data_as(C.POINTER(C.c_int32))))

    @property
    def specialized(self):
        """True when the handle runs a config-specialised code object (QS_SPEC, include/quadswarm.h)."""
        return bool(lib().qs_is_specialized(self._h))

    @property
    def spec_note(self):
        """why the handle runs the generic kernels ('' when it runs the config-specialised object)"""
        buf = C.create_string_buffer(1024)
        lib().qs_spec_status(self._h, buf, len(buf))
        return buf.value.decode()

    @property
    def team(self):
        """True when the handle launches the 4-wave team kernels (small batches), False for the single-wave ones."""
        return bool(lib().qs_kernel_flavor(self._h) & 2)

    @property
    def waves_per_workgroup(self):
        return (lib().qs_kernel_flavor(self._h) >> 8) & 0xff

    @property
    def kernel_name(self):
        """Name of the step kernel this handle launches (as it appears in a rocprofv3 kernel trace)."""
        fl = lib().qs_kernel_flavor(self._h)
        if fl & 1:
            return "qs_spec_step"
        real = "float" if self.real_size == 4 else "double"
        return ("qs_step_team" if fl & 2 else "qs_step_kernel") + ("_full" if fl & 4 else "") + f"<{real}>"

    def kernel_time(self):
        ms, n = C.c_double(0), C.c_int64(0)
        _check(lib().qs_get_kernel_time(self._h, C.byref(ms), C.byref(n)))
        return ms.value, n.value

    # ---- data access ---------------------------------------------------------------------------
    def _dtype(self, kind):
        return {"real": self.np_real, "u1": np.uint8, "u4": np.uint32, "i4": np.int32, "u8": np.uint64}[kind]

    def ptr(self, name):
        return getattr(self.bufs, name)

    # state arrays the library keeps wave-blocked (include/quadswarm.h, qs_buffers): host code sees them component-major
    BLOCKED = frozenset(("pos", "vel", "omega", "rot", "thrust_rot_damp", "thrust_cmds_damp", "ou_state", "goal", "flags", "col_pair_mask"))

    def to_host(self, name):
        shape, kind = self._shapes[name]
        out = np.empty(shape, dtype=self._dtype(kind))
        if name in self.BLOCKED:
            _check(lib().qs_state_array_copy(self._h, out.ctypes.data_as(C.c_void_p), C.c_void_p(self.ptr(name)), out.dtype.itemsize,
                                             shape[0] if len(shape) == 2 else 1, 0))
        else:
            _check(lib().qs_memcpy_d2h(self._h, out.ctypes.data_as(C.c_void_p), C.c_void_p(self.ptr(name)), out.nbytes))
        return out

    def from_host(self, name, arr):
        shape, kind = self._shapes[name]
        a = np.ascontiguousarray(arr, dtype=self._dtype(kind)).reshape(shape)
        if name in self.BLOCKED:
            _check(lib().qs_state_array_copy(self._h, a.ctypes.data_as(C.c_void_p), C.c_void_p(self.ptr(name)), a.dtype.itemsize,
                                             shape[0] if len(shape) == 2 else 1, 1))
        else:
            _check(lib().qs_memcpy_h2d(self._h, C.c_void_p(self.ptr(name)), a.ctypes.data_as(C.c_void_p), a.nbytes))

    def tensor(self, name):
        """Zero-copy torch view (device tensor) of a library buffer."""
        import torch
        if name in self.BLOCKED:
            raise QsError(f"'{name}' is a wave-blocked state array: use to_host / from_host (or qs_get_state)")
        if name not in self._torch_cache:
            shape, kind = self._shapes[name]
            typestr = {"real": "<f8" if self.real_size == 8 else "<f4", "u1": "|u1", "u4": "<u4", "i4": "<i4", "u8": "<u8"}[kind]
            if kind in ("u4", "u8"):   # torch has no unsigned 32/64 views everywhere: expose as signed
                typestr = typestr.replace("u", "i")
            self._torch_cache[name] = torch.as_tensor(_DevArray(self.ptr(name), shape, typestr), device=f"cuda:{self.device}")
        return self._torch_cache[name]

    def get_state(self, env):
        s = np.zeros((self.N, qcfg.QS_STATE_STRIDE))
        tick = C.c_int32(0)
        _check(lib().qs_get_state(self._h, env, s.ctypes.data_as(C.POINTER(C.c_double)), C.byref(tick)))
        return s, tick.value

    def set_state(self, env, s, tick=-1):
        a = np.ascontiguousarray(s, dtype=np.float64).reshape(self.N, qcfg.QS_STATE_STRIDE)
        _check(lib().qs_set_state(self._h, env, a.ctypes.data_as(C.POINTER(C.c_double)), tick))
