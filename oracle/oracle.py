"""ctypes wrapper of the CPU oracle (oracle/quadswarm_oracle.c).   *** TEST INFRASTRUCTURE ***

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libquadswarm_oracle.so")

QS_MAX_AGENTS, QS_MAX_OBSTACLES, QS_CNT_COUNT, QS_EPS_COUNT, QS_RI_COUNT, QS_STATE_STRIDE = 64, 64, 11, 6, 17, 35


class QsoInfo(C.Structure):
    _fields_ = [
        ("unique_col_mask", C.c_uint64), ("obst_new_mask", C.c_uint64), ("obst_hit_mask", C.c_uint64),
        ("room_new_mask", C.c_uint64),
        ("col_pair_mask", C.c_uint64 * QS_MAX_AGENTS), ("new_pair_mask", C.c_uint64 * QS_MAX_AGENTS),
        ("counters", C.c_int32 * QS_CNT_COUNT), ("ep_counters", C.c_int32 * QS_CNT_COUNT),
        ("obst_hit_idx", C.c_int32 * QS_MAX_AGENTS), ("flags", C.c_uint32 * QS_MAX_AGENTS),
        ("tick", C.c_int32), ("num_resets", C.c_int32),
        ("ep_stats", (C.c_double * QS_EPS_COUNT) * QS_MAX_AGENTS),
        ("obst_pos", (C.c_double * 2) * QS_MAX_OBSTACLES),
        ("acc", (C.c_double * 3) * QS_MAX_AGENTS),
        ("nan_reward", C.c_int32), ("tape_underrun", C.c_int32), ("scenario", C.c_int32), ("ep_scenario", C.c_int32),
    ]


def build(force=False):
    src = os.path.join(HERE, "quadswarm_oracle.c")
    if force or not os.path.exists(LIB_PATH) or os.path.getmtime(LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", HERE, "-s"])
    return LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(LIB_PATH)
        dp, u8p = C.POINTER(C.c_double), C.POINTER(C.c_uint8)
        _lib.qso_create.restype = C.c_void_p
        _lib.qso_create.argtypes = [C.c_void_p, C.c_int32]
        _lib.qso_destroy.argtypes = [C.c_void_p]
        _lib.qso_set_tape.argtypes = [C.c_void_p, dp, C.c_int64]
        _lib.qso_tape_pos.restype = C.c_int64
        _lib.qso_tape_pos.argtypes = [C.c_void_p]
        _lib.qso_reset.argtypes = [C.c_void_p, dp]
        _lib.qso_step.argtypes = [C.c_void_p, dp, dp, dp, u8p, dp]
        _lib.qso_get_state.argtypes = [C.c_void_p, dp, C.POINTER(C.c_int32)]
        _lib.qso_set_state.argtypes = [C.c_void_p, dp, C.c_int32]
        _lib.qso_get_info.argtypes = [C.c_void_p, C.POINTER(QsoInfo)]
        _lib.qso_set_reward_coeffs.argtypes = [C.c_void_p, dp]
        _lib.qso_set_numpy126_quirk.argtypes = [C.c_void_p, C.c_int32]
        _lib.qso_step_batch.argtypes = [C.POINTER(C.c_void_p), C.c_int32, dp, dp, dp, u8p]
        _lib.qso_rollout_batch.argtypes = [C.POINTER(C.c_void_p), C.c_int32, dp, C.c_int32, C.c_int32, dp, dp, u8p]
        _lib.qso_rollout_batch_threads.argtypes = [C.POINTER(C.c_void_p), C.c_int32, dp, C.c_int32, C.c_int32, dp, dp, u8p, C.c_int32]
        _lib.qso_sizeof_config.restype = C.c_size_t
        _lib.qso_sizeof_info.restype = C.c_size_t
        _lib.qso_obs_dim.argtypes = [C.c_void_p]
        _lib.qso_polar_rotation.argtypes = [dp, dp]
        _lib.qso_cell_centers.argtypes = [C.c_int32, C.c_int32, dp]
        _lib.qso_surround_sdf.argtypes = [dp, dp, C.c_int32, C.c_double, C.c_double, dp]
        _lib.qso_obst_first_hit.argtypes = [dp, dp, C.c_int32, C.c_double]
        _lib.qso_collision_obstacle_kat.argtypes = [dp, dp, dp, dp, dp]
        _lib.qso_collision_matrix.argtypes = [dp, C.c_int32, C.c_double, C.POINTER(C.c_int32), C.POINTER(C.c_uint64)]
        _lib.qso_philox4x32.argtypes = [C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
        assert _lib.qso_sizeof_info() == C.sizeof(QsoInfo), (_lib.qso_sizeof_info(), C.sizeof(QsoInfo))
    return _lib


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


class OracleEnv:
    """One environment (N drones) of the oracle.  Mirrors QuadrotorEnvMulti.reset()/step()."""

    def __init__(self, cfg, env_global_id=0, tape=None):
        L = lib()
        assert L.qso_sizeof_config() == C.sizeof(cfg), "qs_config layout mismatch"
        self.cfg = cfg
        self.n = cfg.num_agents
        self.obs_dim = L.qso_obs_dim(C.byref(cfg))
        self._h = L.qso_create(C.byref(cfg), env_global_id)
        if not self._h:
            raise ValueError("qso_create failed (bad config)")
        self._tape = None
        if tape is not None:
            self.set_tape(tape)

    def set_tape(self, tape):
        self._tape = np.ascontiguousarray(tape, dtype=np.float64)
        lib().qso_set_tape(self._h, _dp(self._tape), self._tape.size)

    @property
    def tape_pos(self):
        return lib().qso_tape_pos(self._h)

    def reset(self):
        obs = np.zeros((self.n, self.obs_dim))
        lib().qso_reset(self._h, _dp(obs))
        return obs

    def step(self, actions):
        a = np.ascontiguousarray(actions, dtype=np.float64).reshape(self.n, 4)
        obs = np.zeros((self.n, self.obs_dim))
        rew = np.zeros(self.n)
        done = np.zeros(self.n, dtype=np.uint8)
        ri = np.zeros((self.n, QS_RI_COUNT))
        lib().qso_step(self._h, _dp(a), _dp(obs), _dp(rew), done.ctypes.data_as(C.POINTER(C.c_uint8)), _dp(ri))
        return obs, rew, done, ri

    def get_state(self):
        s = np.zeros((self.n, QS_STATE_STRIDE))
        tick = C.c_int32(0)
        lib().qso_get_state(self._h, _dp(s), C.byref(tick))
        return s, tick.value

    def set_state(self, s, tick=-1):
        s = np.ascontiguousarray(s, dtype=np.float64).reshape(self.n, QS_STATE_STRIDE)
        lib().qso_set_state(self._h, _dp(s), tick)

    def info(self):
        out = QsoInfo()
        lib().qso_get_info(self._h, C.byref(out))
        return out

    def set_reward_coeffs(self, coeffs):
        c = np.ascontiguousarray(coeffs, dtype=np.float64)
        lib().qso_set_reward_coeffs(self._h, _dp(c))

    def set_numpy126_quirk(self, on=True):
        """NumPy-1.26 value-based casting of the omega damping factor after a float32 omega (numpy floor mode)"""
        lib().qso_set_numpy126_quirk(self._h, 1 if on else 0)

    def close(self):
        if self._h:
            lib().qso_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class OracleBatch:
    """E independent oracle envs stepped together (OpenMP over envs): the CPU baseline."""

    def __init__(self, cfg, num_envs=None, env_id_offset=0):
        self.envs = [OracleEnv(cfg, env_global_id=env_id_offset + k) for k in range(num_envs or cfg.num_envs)]
        self.n, self.obs_dim, self.e = cfg.num_agents, self.envs[0].obs_dim, len(self.envs)
        self._handles = (C.c_void_p * self.e)(*[e._h for e in self.envs])
        self._out = None

    def reset(self):
        return np.stack([e.reset() for e in self.envs])

    def rollout(self, action_ring, steps, threads=0):
        """`steps` control steps of every env inside one OpenMP region of `threads` threads (0 = the OpenMP default; no
        per-step fork/join): the CPU baseline."""
        a = np.ascontiguousarray(action_ring, dtype=np.float64).reshape(-1, self.e, self.n, 4)
        if self._out is None:
            self._out = (np.zeros((self.e, self.n, self.obs_dim)), np.zeros((self.e, self.n)), np.zeros((self.e, self.n), dtype=np.uint8))
        obs, rew, done = self._out
        lib().qso_rollout_batch_threads(self._handles, self.e, _dp(a), a.shape[0], steps, _dp(obs), _dp(rew),
                                        done.ctypes.data_as(C.POINTER(C.c_uint8)), int(threads))
        return obs, rew, done

    def step(self, actions):
        a = np.ascontiguousarray(actions, dtype=np.float64).reshape(self.e, self.n, 4)
        obs = np.zeros((self.e, self.n, self.obs_dim))
        rew = np.zeros((self.e, self.n))
        done = np.zeros((self.e, self.n), dtype=np.uint8)
        lib().qso_step_batch(self._handles, self.e, _dp(a), _dp(obs), _dp(rew), done.ctypes.data_as(C.POINTER(C.c_uint8)))
        return obs, rew, done
