// quadswarm_hip.hip - MI355X (gfx950) QuadSwarm environment stepper: kernels + C ABI (include/quadswarm.h).
//
// Mapping (wave64): one lane = one drone, one workgroup = one wavefront = floor(64/N) whole environments,
// so every cross-drone exchange of an environment (pair scan, neighbour selection, downwash, collision
// responses) goes through LDS inside one wave and needs no inter-workgroup traffic.  State is
// struct-of-arrays in HBM (component-major), so lane l of a wave reads word l of each component row:
// fully coalesced.  Observations are staged in LDS and copied out as one contiguous block per workgroup.
//
// Reference path (gym_art/quadrotor_multi/): quadrotor_multi.py:413-724 (step), :339-411 (reset); the
// per-piece citations are in qs_device.h and next to each phase below.  SURVEY.md Appendix A gives the
// order of operations that the step kernel follows.
#include <hip/hip_runtime.h>

#include <cctype>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <dlfcn.h>
#include <sys/stat.h>
#include <unistd.h>

#include <string>
#include <vector>

#include "qs_kernels.h"

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
extern "C" int qs_obs_dim(const qs_config *c);
extern "C" int qs_destroy(struct qs_handle *h);
// noise-tape flavour of the kernels (qs_tape_kernels.hip, compiled with QS_TAPE)
extern "C" int qs_tape_lds_bytes(const qs_config *cfg, int obs_dim, int full, int real_size);
extern "C" int qs_tape_launch(int which, const qs_config *cfg, int obs_dim, int full, int real_size, const void *consts,
    const void *ptrs, const void *actions, void *stream);
static thread_local std::string g_last_error;
static int fail(int code, const std::string &msg) { g_last_error = msg; return code; }
#define HIP_TRY(expr) do { hipError_t _e = (expr); if (_e != hipSuccess) return fail(QS_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e)); } while (0)

struct qs_handle {
    qs_config cfg;
    int device = 0;
    int real_size = 4;
    int obs_dim = 0, epb = 1, blocks = 0;
    LdsLayout lds;
    bool full = false;     // scenario outside the fast set => kernels compiled with QS_SCEN_FULL
    int cus = 256;
    // waves per workgroup of the team kernels (qs_step_team.inc): 4 generic, 8 (or 4) specialised; 0 = single-wave kernels
    int team = 0;
    // config-specialised code object (qs_spec_kernels.hip), when one is cached / could be built
    // environment snapshots (qs_snapshot_*): `snap_slots` packed copies of one environment's complete state
    // strides / counts in elements; kind: 1 = obs, 2 = episode sums; group > 0: wave-blocked (envs per block, bytes between blocks)
    struct SnapArray { char *base; size_t elem, comps, comp_stride, per_env; int kind; size_t group, group_stride; };
    std::vector<SnapArray> snap_arrays;
    size_t snap_bytes = 0;
    char *snap_pool = nullptr;
    int32_t snap_slots = 0;
    hipModule_t spec_mod = nullptr;
    // (spec_gated: team objects only)
    hipFunction_t spec_step = nullptr, spec_rollout = nullptr, spec_reset = nullptr, spec_gated = nullptr;
    std::string spec_note;   // why the handle runs the generic kernels (empty when it runs a config-specialised code object)
    Consts<float> kf;    // kernel constants, passed by value in the kernarg segment
    Consts<double> kd;
    Ptrs<float> pf;     // same field layout for float/double: only the pointee type differs
    std::vector<void *> allocs;
    qs_buffers bufs;
    void *d_actions = nullptr;
    void *obs_target = nullptr;   // qs_set_obs_target: where the next launches write their observation rows (nullptr = bufs.obs)
    double *d_state_buf = nullptr;
    int32_t *d_tick_io = nullptr;
    // cached hipGraph of a K-step rollout (qs_step_many): K identical step-kernel nodes
    hipStream_t cap_stream = nullptr;
    hipGraphExec_t graph_exec = nullptr;
    const void *graph_actions = nullptr;
    int32_t graph_k = 0;
    uint8_t *h_mask = nullptr;   // pinned staging for qs_reset masks
    // batched experience replay (qs_replay_enable)
    bool replay_on = false;
    // a step was taken since qs_replay_enable: an explicit reset from now on is recorded by the wrapper state
    bool replay_stepped = false;
    ReplayParams rp;
    // noise tape (qs_set_noise_tape): device copy [E][tape_len] + per-env cursor; while set, reset / step run the tape kernels
    double *d_tape = nullptr;
    int32_t *d_tape_pos = nullptr;
    int64_t tape_len = 0;
    // resident-state stepping (qs_gate_create / qs_step_gated): device descriptor + action ring + sequence flags, one allocation
    qsx::Gate *d_gate = nullptr;
    qsx::Gate gate_host = {};
    // control steps launched so far by qs_step_gated / fed so far by qs_gate_produce
    unsigned long long gate_step_seq = 0, gate_prod_seq = 0;
    // The gated launch must be RESIDENT while its producer runs: two streams of one process may share a hardware queue (the runtime maps
    // streams onto a few HSA queues per priority level), and a queue runs its kernels one after the other - the producer behind a stepper
    // that waits for it would be a bounded deadlock (seen: 500 ms per launch in one of two otherwise identical bench runs).  The gated
    // kernel therefore runs on a stream of the library's own with the HIGHEST priority - a different queue pool than the caller's normal-
    // priority streams - stream-ordered with the caller's stream through two events.
    bool gate_pending = false;   // a gated launch was issued and has not been joined on the host since
    hipStream_t gate_stream = nullptr;
    hipEvent_t gate_ev_in = nullptr, gate_ev_out = nullptr;
    // profiling of the step kernel
    bool profiling = false;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> events;
    size_t events_used = 0;
};

template <typename real> static void fill_consts(const qs_config &c, Consts<real> &k) {
    memset(&k, 0, sizeof k);
    for (int q = 0; q < 3; ++q) { k.inertia[q] = (real)c.inertia[q]; k.inv_inertia[q] = (real)(1.0 / c.inertia[q]);
        k.room_lo[q] = (real)c.room_lo[q]; k.room_hi[q] = (real)c.room_hi[q];
                                  k.nbr_clip_pos[q] = (real)c.nbr_clip_pos[q]; k.nbr_clip_vel[q] = (real)c.nbr_clip_vel[q]; }
    k.arm = (real)c.arm; k.mass = (real)c.mass; k.inv_mass = (real)(1.0 / c.mass);
    for (int m = 0; m < 4; ++m) { for (int q = 0; q < 3; ++q) k.prop_cross[m][q] = (real)c.prop_cross[m][q];
                                  k.prop_ccw[m] = (real)c.prop_ccw[m]; k.thrust_max[m] = (real)c.thrust_max[m];
                                  k.torque_max[m] = (real)c.torque_max[m]; }
    k.motor_tau_up = (real)c.motor_tau_up; k.motor_tau_down = (real)c.motor_tau_down; k.motor_linearity = (real)c.motor_linearity;
    k.vel_damp = (real)c.vel_damp; k.damp_omega_quadratic = (real)c.damp_omega_quadratic; k.omega_max = (real)c.omega_max;
    k.thrust_noise_sigma = (real)c.thrust_noise_sigma; k.ou_theta = (real)c.ou_theta;
    k.dt = (real)c.dt; k.control_dt = (real)(c.dt * c.sim_steps);
    k.floor_threshold = (real)(c.floor_mode == QS_FLOOR_NUMPY ? 0.05 : c.arm);   // quadrotor_dynamics.py:75 / :378
    k.pos_norm_std = (real)c.pos_norm_std; k.pos_unif_range = (real)c.pos_unif_range; k.vel_norm_std = (real)c.vel_norm_std;
    k.vel_unif_range = (real)c.vel_unif_range; k.quat_norm_std = (real)c.quat_norm_std; k.quat_unif_range = (real)c.quat_unif_range;
    k.gyro_noise_density = (real)c.gyro_noise_density;
    k.collision_threshold = (real)c.collision_threshold; k.collision_falloff_threshold = (real)c.collision_falloff_threshold;
    for (int q = 0; q < QS_REW_COUNT; ++q) k.rew_coeff[q] = (real)c.rew_coeff[q];
    k.spawn_box = (real)c.spawn_box; k.approach_goal_metric = (real)c.approach_goal_metric;
    k.obst_radius = (real)(c.obst_size / 2.0); k.obst_hit_threshold = (real)(c.arm + c.obst_size / 2.0); k.obst_size = (real)c.obst_size;
    k.room_mid_z = (real)((c.room_hi[2] - c.room_lo[2]) / 2.0);
    k.sim_steps = c.sim_steps; k.ep_len = c.ep_len; k.floor_mode = c.floor_mode; k.svd_period = c.svd_period; k.sense_noise = c.sense_noise;
    k.obs_repr = c.obs_repr; k.self_dim = c.obs_repr == 0 ? 18 : (c.obs_repr == 1 ? 19 : 24); k.obs_dim = qs_obs_dim(&c);
    k.num_neighbors = c.num_neighbors; k.use_downwash = c.use_downwash; k.use_obstacles = c.use_obstacles; k.scenario = c.scenario;
    k.num_obstacles = c.num_obstacles; k.obst_area[0] = c.obst_area[0]; k.obst_area[1] = c.obst_area[1];
    const double control_freq = 1.0 / (c.dt * c.sim_steps);
    k.control_freq = (int)(control_freq + 0.5);
    k.grace_steps = (int)std::ceil(1.5 * control_freq - 1e-9);    // tick >= 1.5*control_freq (quadrotor_multi.py:146,:451)
    k.final_steps = (int)std::floor(5.0 * control_freq + 1e-9);   // time_remain <= 5*control_freq (:150,:455)
    k.cube_fd_all = (int)pow((double)c.num_agents, 1.0 / 3);
    k.cube_fd[0] = (int)pow((double)(c.num_agents / 2), 1.0 / 3);
    k.cube_fd[1] = (int)pow((double)(c.num_agents - c.num_agents / 2), 1.0 / 3);
    k.seed_lo = (uint32_t)(c.seed & 0xffffffffu); k.seed_hi = (uint32_t)(c.seed >> 32);
    k.env_id_offset = c.env_id_offset; k.num_envs = c.num_envs; k.num_agents = c.num_agents;
    k.write_rew_info = c.write_rew_info;
    k.episode_sums = c.episode_sums;
    k.dr_num_density = c.use_obstacles ? c.dr_num_density : 0; k.dr_num_size = c.use_obstacles ? c.dr_num_size : 0;
    k.dr_on = (k.dr_num_density > 0 || k.dr_num_size > 0) ? 1 : 0;
    k.arm_r = (real)c.arm;
    k.inv_dt = (real)(1.0 / c.dt);
    k.prox_ratio = (real)(-c.rew_coeff[QS_REW_QUADCOL_SMOOTH_MAX] / c.collision_falloff_threshold);
    for (int w = 0; w < 3; ++w) {
        const int win = (w == 0 ? 1 : (w == 1 ? 3 : 5)) * k.control_freq, total = c.ep_len + 1;
        k.inv_win[w] = (real)(1.0 / (double)(total < win ? total : win));
    }
}

extern "C" int qs_obs_dim(const qs_config *c);
static int validate(const qs_config *c);

// ------------------------------------------------------------------------------------------------
// Config-specialised code objects: header text -> key -> <cache>/qs_<key>.hsaco (built with hipcc --genco)
// ------------------------------------------------------------------------------------------------
static bool scenario_is_full(int scenario) {
    return !(scenario == QS_SCENARIO_STATIC_SAME_GOAL || scenario == QS_SCENARIO_O_STATIC_SAME_GOAL
        || scenario == QS_SCENARIO_SWARM_VS_SWARM);
}
static int spec_team_waves(int num_agents);
// Team kernels pay off while the whole batch still fits at <= 8 waves per CU (measured on MI355X, specialised fp32 kernels, us per
// step team / single-wave: C2 shape 1024 envs 8.3 / 9.1, 2048 envs 9.1 / 9.8, 3072 envs 13.8 / 10.8; C4 shape 1024 envs 23.1 / 26.4)
static bool team_default(int blocks, int cus, int num_agents) { return (long)blocks * spec_team_waves(num_agents) <= 8L * cus; }
// Specialised team kernels: 8 waves (2 per SIMD) halve the striped phases once more for N <= 8 (C2 8.65 -> 8.15 us); with
// N > 8 the merge of 8 sorted lists outweighs that (C4 24.6 -> 28.3 us), so those keep 4 waves.
static int spec_team_waves(int num_agents) { return num_agents <= 8 ? 8 : 4; }

// Rows per pass of the observation output of a config-specialised single-wave kernel (qs_kernels.h, lds_layout): 16 rows keep a
// workgroup under 10 KB of LDS (16 workgroups per CU); the register-held row values need D - S <= QS_NV_MAX, i.e. K <= 8.
// QS_OBS_RP=16|32|64 overrides (64 = complete rows at once, the generic kernels' layout).
static int spec_rows_per_pass(const qs_config *cfg, int team) {
    const int self = cfg->obs_repr == 0 ? 18 : (cfg->obs_repr == 1 ? 19 : 24);
    if (team || qs_obs_dim(cfg) - self > QS_NV_MAX) return QS_WAVE;
    int rp = 16;
    if (const char *ev = getenv("QS_OBS_RP")) { const int v = atoi(ev); if (v == 16 || v == 32 || v == 64) rp = v; }
    return rp;
}

static std::string spec_header_text(const qs_config *cfg, int team) {
    const int rs = cfg->precision == QS_PRECISION_F64 ? 8 : 4, epb = QS_WAVE / cfg->num_agents;
    LdsLayout L = lds_layout(rs, QS_WAVE, cfg->num_agents, epb, qs_obs_dim(cfg), cfg->num_obstacles, cfg->num_neighbors, team,
        scenario_is_full(cfg->scenario), cfg->scenario,
                             spec_rows_per_pass(cfg, team));
    std::vector<uint32_t> w;
    if (rs == 8) { Consts<double> k; fill_consts<double>(*cfg, k); memset(k.rew_coeff, 0, sizeof k.rew_coeff); k.prox_ratio = 0;
        k.seed_lo = k.seed_hi = 0; k.env_id_offset = 0; k.num_envs = 0;
                   w.resize(sizeof k / 4); memcpy(w.data(), &k, sizeof k); }
    else { Consts<float> k; fill_consts<float>(*cfg, k); memset(k.rew_coeff, 0, sizeof k.rew_coeff); k.prox_ratio = 0;
        k.seed_lo = k.seed_hi = 0; k.env_id_offset = 0; k.num_envs = 0;
           w.resize(sizeof k / 4); memcpy(w.data(), &k, sizeof k); }
    std::string o = "// generated by quadswarm_hip (spec_header_text): configuration constants as literals\n";
    char t[256];
    snprintf(t, sizeof t, "#define QS_SPEC_PRECISION %d\n#define QS_SPEC_TEAM %d\n#define QS_SPEC_FULL %d\n#define QS_SPEC_EPB %d\n#define QS_SPEC_N %d\n", rs, team,
             scenario_is_full(cfg->scenario) ? 1 : 0, epb, cfg->num_agents);
    o += t;
    snprintf(t, sizeof t, "struct QsSpecCW { uint32_t w[%zu]; };\n", w.size()); o += t;
    o += "static constexpr QsSpecCW qs_spec_cw = {{";
    for (size_t k = 0; k < w.size(); ++k) { snprintf(t, sizeof t, "%s0x%08xu", k ? "," : "", w[k]); o += t; }
    o += "}};\n";
    const int *li = (const int *)&L;
    snprintf(t, sizeof t, "struct QsSpecLW { int w[%zu]; };\n", sizeof L / 4); o += t;
    o += "static constexpr QsSpecLW qs_spec_lw = {{";
    for (size_t k = 0; k < sizeof L / 4; ++k) { snprintf(t, sizeof t, "%s%d", k ? "," : "", li[k]); o += t; }
    o += "}};\n";
    return o;
}

static std::string lib_dir() {
    Dl_info info;
    if (dladdr((const void *)&qs_obs_dim, &info) && info.dli_fname) {
        std::string f = info.dli_fname;
        size_t k = f.rfind('/');
        return k == std::string::npos ? std::string(".") : f.substr(0, k);
    }
    return ".";
}
static uint64_t fnv1a(uint64_t h, const std::string &s) { for (unsigned char ch : s) { h ^= ch; h *= 1099511628211ull; } return h; }
static bool read_file(const std::string &path, std::string &out) {
    FILE *f = fopen(path.c_str(), "rb");
    if (!f) return false;
    char buf[65536]; size_t n;
    out.clear();
    while ((n = fread(buf, 1, sizeof buf, f)) > 0) out.append(buf, n);
    fclose(f);
    return true;
}
static const char *const kSpecSources[] = {"qs_spec_kernels.hip", "qs_kernels.h", "qs_device.h", "qs_scenarios.h", "qs_step_sem.h",
    "qs_xchg_dev.h", "qs_step_kernel.inc", "qs_step_team.inc"};
static const char *const kSpecFlags = "--genco --offload-arch=gfx950 -O3 -std=c++17";
// fp32 objects only (the production precision, specified to 1e-5): reassociation / finite-math simplifications are worth ~8 %
// of the step; the SLP vectoriser's v_pk_* pairs cost more register shuffling than they save on this code.  The f64 parity
// instantiation and the generic library keep strict IEEE semantics.
static const char *const kSpecFlagsF32 = "-ffast-math -fno-slp-vectorize";
// team objects only (one wave per SIMD, a budget of 256 - 512 VGPRs it does not need for occupancy): the machine scheduler's max-ILP
// strategy instead of the occupancy-first default.  Same box, us per step: C4 13.58 -> 12.98, C3 8.39 -> 8.17, C2 7.86 -> 7.84, mix 11.95
// -> 11.76; the single-wave throughput kernels lose 1 % with it and keep the default (profiles/r03_sched_max_ilp_ab.txt).  Instruction
// order
// only: results are bit-identical.  If the compiler fails on an object with it, the object is built without.
static const char *const kSpecFlagsTeam = "-mllvm -amdgpu-sched-strategy=max-ilp";
// 8-wave team objects (N <= 8) only: without the post-RA scheduler on top.  Round 5's scheduler sweeps (tools/sched_sweep.py,
// profiles/r05s_ / r05t_sched_sweep.txt; same box, us per step): C2 7.65 -> 7.43-7.46, C3 8.01 -> 8.01, while the 4-wave C4 object loses
// with it (12.65 -> 13.3) and keeps the flag above; every other knob of the sweep (clustering, reschedule stages, RP trackers, scheduling
// direction) stayed inside +- 0.03 of these.  Instruction order only: results are bit-identical.
static const char *const kSpecFlagsTeam8 = "-mllvm -amdgpu-sched-strategy=max-ilp -mllvm -enable-post-misched=0";
// (QS_SPEC_TEAM_FLAGS in the environment replaces both - part of the cache key like QS_SPEC_EXTRA_FLAGS: the sweeps of
// tools/sched_sweep.py) single-wave float32 objects (the throughput kernels, capped at 128 registers): the scheduler's AMDGPU-specific
// register-pressure trackers. Round 5's sweeps (profiles/r05z_sched_sweep.txt, r05z5_sched_sweep.txt; 2^20 drones, same box, us per step):
// the C3 shape 118.6 -> 109.2, 119.7 -> 114.0, 112.6 -> 108.2 (44 -> 12 bytes of scratch per lane), the C2 and C4 shapes inside their
// run-to-run noise.  Round 5 did not ship it because one parity case failed with it (e_n17_kall_obst); round 6 found why - a spill the
// register allocator put in front of an exec restore, a compiler defect that any object can carry, with or without this flag
// (spec_verify_file above) - and every object is now checked for it whatever its flags.  Instruction order only: results are
// bit-identical (tests/test_object_identity_gpu.py).
static const char *const kSpecFlagsSingleF32 = "-mllvm -amdgpu-use-amdgpu-trackers";
static const char *spec_team_flags(int team) { const char *ev = getenv("QS_SPEC_TEAM_FLAGS"); return ev
    ? ev : (team == 8 ? kSpecFlagsTeam8 : kSpecFlagsTeam); }
static const char *spec_sched_flags(int team, int precision) {
    if (team > 0) return spec_team_flags(team);
    if (const char *ev = getenv("QS_SPEC_SINGLE_FLAGS")) return ev;
    return precision == QS_PRECISION_F64 ? "" : kSpecFlagsSingleF32;
}

// key = hash(header text, kernel sources, flags); false if the sources are not next to the library
static bool spec_key(const std::string &header, std::string &key) {
    uint64_t h = fnv1a(14695981039346656037ull, header);
    h = fnv1a(h, kSpecFlags);
    h = fnv1a(h, kSpecFlagsF32);
    // (the header carries team width and precision: all strings, whichever applies)
    h = fnv1a(h, spec_team_flags(4)); h = fnv1a(h, spec_team_flags(8)); h = fnv1a(h, spec_sched_flags(0, QS_PRECISION_F32));
    if (const char *xf = getenv("QS_SPEC_EXTRA_FLAGS")) h = fnv1a(h, xf);   // e.g. -DQS_TIMING for tools/phase_timing.py
    const std::string dir = lib_dir();
    for (const char *src : kSpecSources) {
        std::string text;
        if (!read_file(dir + "/" + src, text)) return false;
        h = fnv1a(h, text);
    }
    for (const char *pub_name : {"quadswarm.h", "quadswarm_exchange.h"}) {   // the public headers the device code includes
        std::string pub;
        if (!read_file(dir + "/../../include/" + pub_name, pub)) return false;
        h = fnv1a(h, pub);
    }
    char t[32];
    snprintf(t, sizeof t, "%016llx", (unsigned long long)h);
    key = t;
    return true;
}
static std::string spec_cache_dir() {
    const char *ev = getenv("QS_SPEC_CACHE");
    return (ev && ev[0]) ? std::string(ev) : lib_dir() + "/spec_cache";
}
static bool file_exists(const std::string &path) { struct stat st; return stat(path.c_str(), &st) == 0 && st.st_size > 0; }

// ---- code-object verification ------------------------------------------------------------------------------------------
// ROCm 7.2's register allocator can place a VGPR spill, reload, copy or rematerialised constant at the top of a control-flow JOIN block in
// front of the instruction that restores exec there (`s_or_b64 exec, exec, s[a:b]`): the scalar allocator, which runs first, puts its own
// spills / copies at the very top of the block (they do not depend on exec), and the vector allocator's "skip the block prologue" stops at
// the first of those that is not a spill.  A wave that reaches the join through the branch that skipped the `then` side arrives with
// exec == 0: the spill stores nothing and the later reload returns stale scratch memory.  That is what round 5's unexplained parity
// failure was (single-wave N = 17 object with the RP-tracker flag: the environment index came back as a float, the counter loads faulted),
// and the default objects only differed from it by luck: 18 of 269 cached objects carried the pattern somewhere.  Every specialised object
// is therefore disassembled and scanned before it is used; an object with the pattern is rebuilt with other scheduler settings (instruction
// order and register assignment change, results do not) and, if none is clean, not used at all (generic kernels, loudly).
static std::string llvm_bin() { const char *ev = getenv("QS_LLVM_BIN"); return (ev && ev[0]) ? ev : "/opt/rocm/lib/llvm/bin"; }
static bool starts_with(const std::string &t, const char *p) { return t.compare(0, strlen(p), p) == 0; }
// instructions of a block prologue that do not depend on exec (SGPR spills to VGPR lanes, scalar moves / adds, waits)
static bool hz_silent(const std::string &t) {
    static const char *const k[] = {"v_writelane_b32", "v_readlane_b32", "s_nop", "s_waitcnt", "s_mov_b32", "s_mov_b64", "s_add_i32",
        "s_add_u32", "s_addk_i32"};
    for (const char *q : k) if (starts_with(t, q)) return t.find("exec") == std::string::npos;
    return false;
}
// ... and those that do: what the vector register allocator inserts (spill, reload, copy, rematerialised constant, AGPR copy)
static bool hz_exec_dependent(const std::string &t) {
    return starts_with(t, "scratch_load_") || starts_with(t, "scratch_store_") || starts_with(t, "v_mov_b32")
        || starts_with(t, "v_mov_b64") || starts_with(t, "v_accvgpr_");
}
// the exec restore of a join / else block: s_or_b64 exec, exec, s[..] | s_xor_b64 exec, exec, s[..] | s_or_saveexec_b64 s[..], s[..]  (NOT
// `, -1`: whole-wave mode)
static bool hz_restore(const std::string &t) {
    if (starts_with(t, "s_or_b64 exec, exec, s[") || starts_with(t, "s_xor_b64 exec, exec, s[")) return true;
    return starts_with(t, "s_or_saveexec_b64 s[") && t.find("], s[") != std::string::npos;
}
// One hazard: the block prologue (instructions + encodings) in front of a misplaced exec restore, the restore itself, what follows it.
struct SpecHazard { std::string kernel, label; std::vector<std::string> ins, raw; std::string restore, restore_raw;
    std::vector<std::string> after; };
static std::string hz_describe(const SpecHazard &h) {
    std::string o = h.kernel + " <" + h.label + ">:";
    for (const std::string &t : h.ins) o += " " + t + " ;";
    return o + " " + h.restore;
}
// Scan `llvm-objdump -d --symbolize-operands` text (instruction, then `// address: encoding dwords`).
static void spec_scan_disassembly(FILE *f, std::vector<SpecHazard> &out) {
    char line[1024];
    std::string kernel = "?";
    SpecHazard cur;
    bool scanning = false, dependent = false;
    int follow = 0;   // instructions still to record behind the last hazard's restore
    while (fgets(line, sizeof line, f)) {
        std::string t(line);
        while (!t.empty() && (t.back() == '\n' || t.back() == '\r' || t.back() == ' ' || t.back() == '\t')) t.pop_back();
        const size_t lt = t.find(" <"), gt = t.rfind(">:");
        if (!t.empty() && isxdigit((unsigned char)t[0]) && lt != std::string::npos && gt == t.size() - 2) {   // "0000000000002b60 <L14>:"
            const std::string label = t.substr(lt + 2, gt - lt - 2);
            if (!(label.size() > 1 && label[0] == 'L' && isdigit((unsigned char)label[1]))) kernel = label;
            cur = SpecHazard(); cur.kernel = kernel; cur.label = label;
            scanning = true; dependent = false; follow = 0;
            continue;
        }
        const size_t cm = t.find("//");
        if (cm == std::string::npos) continue;
        std::string raw;   // the encoding as bytes (little-endian dwords)
        {
            const size_t colon = t.find(':', cm);
            if (colon != std::string::npos) {
                const char *q = t.c_str() + colon + 1;
                while (*q) {
                    while (*q == ' ') ++q;
                    if (!isxdigit((unsigned char)*q)) break;
                    char *e = nullptr;
                    const unsigned long w = strtoul(q, &e, 16);
                    if (e - q != 8) break;
                    for (int b = 0; b < 4; ++b) raw.push_back((char)((w >> (8 * b)) & 0xff));
                    q = e;
                }
            }
        }
        t.resize(cm);
        size_t b = 0;
        while (b < t.size() && (t[b] == ' ' || t[b] == '\t')) ++b;
        t = t.substr(b);
        while (!t.empty() && (t.back() == ' ' || t.back() == '\t')) t.pop_back();
        if (t.empty()) continue;
        if (follow > 0) { out.back().after.push_back(t); --follow; }
        if (!scanning) continue;
        if (hz_restore(t)) {
            if (dependent) { cur.restore = t; cur.restore_raw = raw; out.push_back(cur); follow = 6; }
            scanning = false;
        } else if (hz_exec_dependent(t)) { dependent = true; cur.ins.push_back(t); cur.raw.push_back(raw); }
        else if (hz_silent(t)) { cur.ins.push_back(t); cur.raw.push_back(raw); }
        else scanning = false;
    }
}
// the plain gfx950 code objects inside `path` - a bundle (hipcc --genco), a plain object, or a shared library whose .hip_fatbin section
// holds one bundle per translation unit - written to temporary files (the caller unlinks them)
static int spec_extract_elfs(const std::string &path, std::vector<std::string> &elfs, std::string &why) {
    static int serial = 0;
    char tag[96];
    snprintf(tag, sizeof tag, "/tmp/qs_verify_%ld_%d", (long)getpid(), serial++);
    const std::string base = tag, bin = llvm_bin();
    std::string blob;
    if (path.size() > 3 && path.compare(path.size() - 3, 3, ".so") == 0) {
        const std::string fat = base + ".fatbin";
        const std::string cmd = "'" + bin + "/llvm-objcopy' --dump-section '.hip_fatbin=" + fat + "' '" + path + "' '" + base + ".copy' > /dev/null 2>&1";
        const int rc = system(cmd.c_str());
        unlink((base + ".copy").c_str());
        const bool ok = rc == 0 && read_file(fat, blob);
        unlink(fat.c_str());
        if (!ok) { why = "cannot extract .hip_fatbin from " + path + " (QS_LLVM_BIN=" + bin + ")"; return -1; }
    } else if (!read_file(path, blob)) { why = "cannot read " + path; return -1; }
    const std::string magic = "__CLANG_OFFLOAD_BUNDLE__";
    int k = 0;
    for (size_t at = blob.find(magic); at != std::string::npos; ++k) {
        const size_t next = blob.find(magic, at + magic.size());
        const std::string part = base + "." + std::to_string(k) + ".bundle", elf = base + "." + std::to_string(k) + ".elf";
        FILE *f = fopen(part.c_str(), "wb");
        if (!f) { why = "cannot write " + part; return -1; }
        fwrite(blob.data() + at, 1, (next == std::string::npos ? blob.size() : next) - at, f);
        fclose(f);
        const std::string unb = "'" + bin + "/clang-offload-bundler' --unbundle --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 '--input=" + part + "' '--output=" + elf + "' > /dev/null 2>&1";
        const bool ok = system(unb.c_str()) == 0 && file_exists(elf);
        unlink(part.c_str());
        if (ok) elfs.push_back(elf); else unlink(elf.c_str());
        at = next;
    }
    if (k == 0) {   // not a bundle: a plain code object (copied, so that the caller can unlink uniformly)
        const std::string elf = base + ".plain.elf";
        FILE *f = fopen(elf.c_str(), "wb");
        if (!f) { why = "cannot write " + elf; return -1; }
        fwrite(blob.data(), 1, blob.size(), f);
        fclose(f);
        elfs.push_back(elf);
    }
    if (elfs.empty()) { why = "no gfx950 code object in " + path; return -1; }
    return 0;
}
static int spec_find_hazards(const std::string &path, std::vector<SpecHazard> &hz, std::string &why) {
    std::vector<std::string> elfs;
    if (spec_extract_elfs(path, elfs, why) != 0) { for (const std::string &e : elfs) unlink(e.c_str()); return -1; }
    int rc = 0;
    for (const std::string &elf : elfs) {
        const std::string cmd = "'" + llvm_bin() + "/llvm-objdump' -d --symbolize-operands '" + elf + "' 2> /dev/null";
        FILE *f = rc == 0 ? popen(cmd.c_str(), "r") : nullptr;
        if (f) { spec_scan_disassembly(f, hz);
            if (pclose(f) != 0) { rc = -1; why = "llvm-objdump failed on " + path + " (QS_LLVM_BIN=" + llvm_bin() + ")"; } }
        else if (rc == 0) { rc = -1; why = "cannot run " + llvm_bin() + "/llvm-objdump"; }
        unlink(elf.c_str());
    }
    return rc;
}
// 0 = clean, 1 = the pattern is there (report: one line per place), < 0 = could not be checked (tools missing, not a code object)
static int spec_verify_file(const std::string &path, std::string &report) {
    std::vector<SpecHazard> hz;
    if (spec_find_hazards(path, hz, report) != 0) return -1;
    for (const SpecHazard &h : hz) report += hz_describe(h) + "\n";
    return hz.empty() ? 0 : 1;
}
// scalar registers named in an operand text: "s5" -> {5}, "s[4:7]" -> {4..7}
static void hz_sregs(const std::string &t, std::vector<int> &regs) {
    for (size_t k = 0; k < t.size(); ++k) {
        if (t[k] != 's' || (k > 0 && (isalnum((unsigned char)t[k - 1]) || t[k - 1] == '_'))) continue;
        if (k + 1 < t.size() && t[k + 1] == '[') {
            int lo = 0, hi = 0;
            if (sscanf(t.c_str() + k, "s[%d:%d]", &lo, &hi) == 2) for (int r = lo; r <= hi; ++r) regs.push_back(r);
        } else if (k + 1 < t.size() && isdigit((unsigned char)t[k + 1])) regs.push_back(atoi(t.c_str() + k + 1));
    }
}
// REPAIR: the exec restore moves to the front of its block prologue (as far forward as the definition of the mask it reads allows) -
// straight-line code that nobody jumps into; every other instruction keeps its place relative to the others.  Made only where it provably
// changes nothing else:
//   * no exec-dependent instruction sits in front of a prologue instruction that writes the SGPR pair the restore reads;
// * none of the last five prologue instructions is a v_readlane (VALU-writes-SGPR -> VMEM-reads-it needs 5 wait states, and the restore was
//     one of them); the first two instructions behind the restore are neither DPP nor lane operations (VALU-write -> DPP wait states);
//   * the byte sequence occurs in the file exactly as often as the scan reports it.
// Returns the number of places repaired (file rewritten in place), `left` = the ones that were not, with the reason; < 0 on errors.
static int spec_repair_file(const std::string &path, std::string &left) {
    std::vector<SpecHazard> hz;
    std::string why;
    if (spec_find_hazards(path, hz, why) != 0) { left = why; return -1; }
    if (hz.empty()) return 0;
    std::string blob;
    if (!read_file(path, blob)) { left = "cannot read " + path; return -1; }
    int repaired = 0;
    std::vector<bool> done(hz.size(), false);
    for (size_t a = 0; a < hz.size(); ++a) {
        if (done[a]) continue;
        const SpecHazard &h = hz[a];
        std::string old_bytes;
        for (const std::string &r : h.raw) old_bytes += r;
        old_bytes += h.restore_raw;
        int same = 0;   // hazards with the identical byte sequence (the same code in two kernels)
        for (size_t b = a; b < hz.size(); ++b) {
            std::string ob;
            for (const std::string &r : hz[b].raw) ob += r;
            ob += hz[b].restore_raw;
            if (ob == old_bytes) { done[b] = true; ++same; }
        }
        // what the restore reads (and, for s_or_saveexec, also writes)
        std::vector<int> need;
        const size_t c1 = h.restore.find(',');
        hz_sregs(starts_with(h.restore, "s_or_saveexec")
            ? h.restore.substr(h.restore.find(' ')) : h.restore.substr(h.restore.find(',', c1 + 1)), need);
        size_t pos = 0;
        for (size_t k = 0; k < h.ins.size(); ++k) {
            const std::string &t = h.ins[k];
            if (!hz_silent(t) || starts_with(t, "s_nop") || starts_with(t, "s_waitcnt") || starts_with(t, "v_writelane")) continue;
            std::vector<int> wr;
            const size_t sp = t.find(' ');
            hz_sregs(t.substr(sp == std::string::npos ? 0 : sp, t.find(',') == std::string::npos ? std::string::npos : t.find(',') - sp),
                wr);
            for (int w : wr) for (int n : need) if (w == n) pos = k + 1;
        }
        std::string reason;
        for (size_t k = 0; k < pos; ++k) if (hz_exec_dependent(h.ins[k])) reason = "an exec-dependent instruction sits in front of the definition of the saved mask";
        // VALU writes an SGPR (v_readlane) -> a VMEM instruction / a lane select reads it: 5 wait states, and the restore was one of them
        for (size_t k = h.ins.size() >= 5 ? h.ins.size() - 5 : 0; k < h.ins.size(); ++k) {
            if (!starts_with(h.ins[k], "v_readlane")) continue;
            std::vector<int> wr;
            hz_sregs(h.ins[k].substr(0, h.ins[k].find(',')), wr);
            for (size_t a2 = 0; a2 < h.after.size() && a2 < 5; ++a2) {
                const std::string &u = h.after[a2];
                const bool vmem = starts_with(u, "buffer_") || starts_with(u, "global_") || starts_with(u, "flat_")
                    || starts_with(u, "scratch_") || starts_with(u, "v_readlane") || starts_with(u, "v_writelane");
                if (!vmem) continue;
                std::vector<int> rd;
                hz_sregs(u, rd);
                for (int w : wr) for (int r : rd) if (w == r) reason = "a v_readlane near the end of the prologue feeds a memory / lane instruction right behind the restore";
            }
        }
        for (size_t k = 0; k < h.after.size() && k < 2; ++k)
            if (h.after[k].find("dpp") != std::string::npos || starts_with(h.after[k], "v_readlane")
                || starts_with(h.after[k], "v_writelane")
                || starts_with(h.after[k], "v_readfirstlane")) reason = "DPP / lane operation right behind the restore";
        if (h.restore_raw.empty() || old_bytes.size() < 8) reason = "no encoding in the disassembly";
        if (reason.empty()) {
            size_t count = 0;
            for (size_t at = blob.find(old_bytes); at != std::string::npos; at = blob.find(old_bytes, at + 1)) ++count;
            if ((int)count != same) reason = "byte sequence found " + std::to_string(count) + " times in the file, " + std::to_string(same) + " expected";
        }
        if (!reason.empty()) { left += hz_describe(h) + "   [" + reason + "]\n"; continue; }
        std::string new_bytes;
        for (size_t k = 0; k < pos; ++k) new_bytes += h.raw[k];
        new_bytes += h.restore_raw;
        for (size_t k = pos; k < h.ins.size(); ++k) new_bytes += h.raw[k];
        for (size_t at = blob.find(old_bytes); at != std::string::npos;
            at = blob.find(old_bytes, at + new_bytes.size())) blob.replace(at, old_bytes.size(), new_bytes);
        repaired += same;
    }
    if (repaired > 0) {
        struct stat st;
        const std::string tmp = path + ".repair.tmp";
        FILE *f = fopen(tmp.c_str(), "wb");
        if (!f) { left = "cannot write " + tmp; return -1; }
        fwrite(blob.data(), 1, blob.size(), f);
        fclose(f);
        if (stat(path.c_str(), &st) == 0) chmod(tmp.c_str(), st.st_mode);
        if (rename(tmp.c_str(), path.c_str()) != 0) { unlink(tmp.c_str()); left = "cannot replace " + path; return -1; }
    }
    return repaired;
}
extern "C" int qs_spec_repair(const char *path, char *left_out, int cap) {
    if (!path) return fail(QS_ERR_INVALID, "null argument");
    std::string left;
    const int rc = spec_repair_file(path, left);
    if (left_out && cap > 0) { const size_t n = left.size() < (size_t)cap - 1 ? left.size() : (size_t)cap - 1;
        memcpy(left_out, left.data(), n); left_out[n] = 0; }
    if (rc < 0) return fail(QS_ERR_UNSUPPORTED, left);
    return rc;
}
extern "C" int qs_spec_verify(const char *path, char *report_out, int cap) {
    if (!path) return fail(QS_ERR_INVALID, "null argument");
    std::string report;
    const int rc = spec_verify_file(path, report);
    if (report_out && cap > 0) { const size_t n = report.size() < (size_t)cap - 1 ? report.size() : (size_t)cap - 1;
        memcpy(report_out, report.data(), n); report_out[n] = 0; }
    if (rc < 0) return fail(QS_ERR_UNSUPPORTED, report);
    return rc;
}

// build <cache>/qs_<key>.hsaco if it is missing; returns its path or "" (reason in g_last_error)
static std::string spec_ensure(const qs_config *cfg, int team, bool build) {
    const std::string header = spec_header_text(cfg, team);
    std::string key;
    if (!spec_key(header, key)) { g_last_error = "kernel sources not found next to the library"; return ""; }
    const std::string dir = spec_cache_dir(), out = dir + "/qs_" + key + ".hsaco", stamp = dir + "/qs_" + key + ".ok";
    // QS_SPEC_VERIFY=0: tools that WANT a flagged object (tools/flag_diff.py)
    const bool verify = !(getenv("QS_SPEC_VERIFY") && atoi(getenv("QS_SPEC_VERIFY")) == 0);
    if (file_exists(out)) {
        if (!verify || file_exists(stamp)) return out;
        std::string report, left;   // an object without its stamp (an older cache): checked now, repaired if that is all it needs
        int v = spec_verify_file(out, report);
        if (v == 1 && spec_repair_file(out, left) > 0) { report.clear(); v = spec_verify_file(out, report); }
        if (v == 0) { FILE *f = fopen(stamp.c_str(), "wb");
            if (f) { fputs("verified: no VGPR spill / copy in front of an exec restore\n", f); fclose(f); } return out; }
        if (v < 0) { g_last_error = "cached code object cannot be verified: " + report; return ""; }
        unlink(out.c_str());   // the pattern is there and cannot be repaired: rebuilt below with other settings
    }
    if (!build) { g_last_error = "no cached code object for this configuration"; return ""; }
    mkdir(dir.c_str(), 0755);
    char tag[64];
    snprintf(tag, sizeof tag, ".%ld.tmp", (long)getpid());
    const std::string hdr = dir + "/qs_" + key + ".h", tmp = out + tag, log = dir + "/qs_" + key + ".log";
    {
        FILE *f = fopen((hdr + tag).c_str(), "wb");
        if (!f) { g_last_error = "cannot write " + hdr; return ""; }
        fwrite(header.data(), 1, header.size(), f);
        fclose(f);
        rename((hdr + tag).c_str(), hdr.c_str());
    }
    const char *cc = getenv("HIPCC");
    const std::string src = lib_dir();
    auto command = [&](const std::string &sched) {
        return std::string(cc && cc[0] ? cc : "/opt/rocm/bin/hipcc") + " " + kSpecFlags + " " + (cfg->precision == QS_PRECISION_F64
            ? "" : kSpecFlagsF32) + " " + sched + " " +
               (getenv("QS_SPEC_EXTRA_FLAGS") ? getenv("QS_SPEC_EXTRA_FLAGS") : "") + " -DQS_SPEC_FILE='\"" + hdr + "\"' '" + src + "/qs_spec_kernels.hip' -o '" + tmp + "' > '" + log + "' 2>&1";
    };
    // The configured scheduler settings first.  An object that carries a spill in front of an exec restore (spec_verify_file) is repaired
    // in place
    // (spec_repair_file) and checked again; if the compiler fails, or a place cannot be repaired, the alternatives follow - each only moves
    // instructions and registers around, the arithmetic is the same (tested: tests/test_object_identity_gpu.py).  Single-wave objects end
    // with the register cap lifted (no spills at all: 3 instead of 4 waves per SIMD).
    std::vector<std::string> tries = {spec_sched_flags(team, cfg->precision)};
    const char *const alt_team[] = {"", "-mllvm -amdgpu-sched-strategy=max-ilp",
        "-mllvm -amdgpu-sched-strategy=max-ilp -mllvm -enable-post-misched=0", "-mllvm -amdgpu-use-amdgpu-trackers"};
    const char *const alt_single[] = {"", "-mllvm -amdgpu-use-amdgpu-trackers", "-mllvm -enable-post-misched=0", "-DQS_WAVES_PER_EU=0"};
    for (const char *a : team > 0 ? alt_team : alt_single) { bool seen = false; for (const std::string &t : tries) seen |= t == a;
        if (!seen) tries.push_back(a); }
    std::string why = "specialised kernel build failed, see " + log, used;
    bool ok = false;
    for (const std::string &sched : tries) {
        unlink(tmp.c_str());
        const int rc = system(command(sched).c_str());
        if (rc != 0 || !file_exists(tmp)) continue;
        if (!verify) { ok = true; used = sched; break; }
        std::string report, left;
        int v = spec_verify_file(tmp, report);
        // the exec restores moved in front of the spills (spec_repair_file): checked again
        if (v == 1 && spec_repair_file(tmp, left) > 0) {
            report.clear();
            v = spec_verify_file(tmp, report);
            if (v == 0) { ok = true; used = sched + "' + exec restores moved to the front of their block prologues '"; break; }
        }
        if (v == 0) { ok = true; used = sched; break; }
        why = v < 0 ? "specialised object cannot be verified: " + report
                    : "every build of this configuration's object has a VGPR spill / copy in front of an exec restore (DESIGN.md 5.3); last one: " + report;
        if (v < 0) break;
    }
    if (!ok) { unlink(tmp.c_str()); g_last_error = why; return ""; }
    if (rename(tmp.c_str(), out.c_str()) != 0) { unlink(tmp.c_str()); g_last_error = "cannot move code object into the cache"; return ""; }
    if (verify) {
        FILE *f = fopen(stamp.c_str(), "wb");
        if (f) { fprintf(f, "verified: no VGPR spill / copy in front of an exec restore; scheduler flags: '%s'%s\n", used.c_str(),
            used.compare(0, tries[0].size() + 1, tries[0] + "'") == 0 || used == tries[0] ? "" : " (the configured ones were rejected)");
            fclose(f); }
    }
    return out;
}

// Ahead-of-time build of the code object qs_create() would look for (no GPU needed).  team: 0 / 1, or -1 = the default
// rule for a 256-CU device.  Writes the path to path_out; returns 0, or < 0 with qs_last_error() set.
extern "C" int qs_spec_build(const qs_config *cfg, int team, char *path_out, int cap) {
    if (!cfg) return fail(QS_ERR_INVALID, "null argument");
    int rc = validate(cfg);
    if (rc != QS_OK) return rc;
    const int epb = QS_WAVE / cfg->num_agents, blocks = (cfg->num_envs + epb - 1) / epb;
    if (team < 0) team = team_default(blocks, 256, cfg->num_agents) ? 1 : 0;
    if (team == 1) team = spec_team_waves(cfg->num_agents);
    if (team != 0 && team != 4 && team != 8) return fail(QS_ERR_INVALID, "team must be -1, 0, 1, 4 or 8");
    std::string path = spec_ensure(cfg, team, true);
    if (path.empty()) return QS_ERR_UNSUPPORTED;
    if (path_out) { if ((int)path.size() + 1 > cap) return fail(QS_ERR_INVALID, "buffer too small");
        memcpy(path_out, path.c_str(), path.size() + 1); }
    return QS_OK;
}

static int validate(const qs_config *c) {
    if (c->num_envs < 1) return fail(QS_ERR_INVALID, "num_envs must be >= 1");
    if (c->num_agents < 1 || c->num_agents > QS_MAX_AGENTS) return fail(QS_ERR_INVALID, "num_agents must be in [1, 64]");
    if (c->num_neighbors < 0 || c->num_neighbors > c->num_agents - 1) return fail(QS_ERR_INVALID, "Incorrect number of neigbors");
    if (c->precision != QS_PRECISION_F32 && c->precision != QS_PRECISION_F64) return fail(QS_ERR_INVALID, "bad precision");
    if (c->scenario < 0 || c->scenario >= QS_SCENARIO_COUNT) return fail(QS_ERR_UNSUPPORTED, "unsupported scenario");
    if (c->scenario == QS_SCENARIO_SWARM_VS_SWARM && c->num_agents < 2) return fail(QS_ERR_INVALID, "swarm_vs_swarm needs >= 2 drones");
    if (c->scenario == QS_SCENARIO_RUN_AWAY && c->num_agents < 2) return fail(QS_ERR_INVALID, "run_away needs >= 2 drones");
    {
        const bool o_scen = c->scenario == QS_SCENARIO_O_STATIC_SAME_GOAL || c->scenario == QS_SCENARIO_O_RANDOM ||
                            c->scenario == QS_SCENARIO_O_DYNAMIC_SAME_GOAL || c->scenario == QS_SCENARIO_O_SWAP_GOALS ||
                            c->scenario == QS_SCENARIO_O_EP_RAND_BEZIER;
        if (c->scenario != QS_SCENARIO_MIX && o_scen != (c->use_obstacles != 0)) return fail(QS_ERR_INVALID,
            "obstacle scenario <=> use_obstacles");
    }
    if (c->use_obstacles) {
        if (c->obst_area[0] < 1 || c->obst_area[1] < 1 || c->obst_area[0] > 16 || c->obst_area[1] > 16) return fail(QS_ERR_UNSUPPORTED,
            "obst_area must be within [1,16]x[1,16]");
        if (c->num_obstacles < 1 || c->num_obstacles > QS_MAX_OBSTACLES
            || c->num_obstacles > c->obst_area[0] * c->obst_area[1]) return fail(QS_ERR_INVALID, "bad num_obstacles");
        if (c->obst_area[0] * c->obst_area[1] - c->num_obstacles < c->num_agents) return fail(QS_ERR_INVALID,
            "not enough free cells to spawn the drones");
    }
    {
        LdsLayout L = lds_layout(c->precision == QS_PRECISION_F64 ? 8 : 4, QS_WAVE, c->num_agents, QS_WAVE / c->num_agents,
            qs_obs_dim(c), c->num_obstacles, c->num_neighbors,
                                 spec_team_waves(c->num_agents) /* the largest layout qs_create may pick */, scenario_is_full(c->scenario), c->scenario);
        if (L.total > 160 * 1024) return fail(QS_ERR_UNSUPPORTED, "observation staging does not fit the 160 KiB LDS of a CU");
    }
    if (c->dr_num_density < 0 || c->dr_num_density > QS_MAX_DR_CHOICES || c->dr_num_size < 0 || c->dr_num_size > QS_MAX_DR_CHOICES)
        return fail(QS_ERR_INVALID, "bad number of domain-randomisation choices");
    if (c->use_obstacles)
        for (int q = 0; q < c->dr_num_density; ++q)
            if (c->dr_obst_count[q] < 1 || c->dr_obst_count[q] > c->num_obstacles) return fail(QS_ERR_INVALID,
                "domain randomisation: obstacle counts must be in [1, num_obstacles]");
    if (c->sim_steps < 1 || c->ep_len < 1 || c->svd_period < 1 || c->svd_period > 255) return fail(QS_ERR_INVALID,
        "bad sim_steps/ep_len/svd_period");
    return QS_OK;
}

template <typename T> static int dalloc(qs_handle *h, T **ptr, size_t count) {
    void *q = nullptr;
    size_t bytes = count * sizeof(T);
    if (bytes == 0) bytes = sizeof(T);
    HIP_TRY(hipMalloc(&q, bytes));
    HIP_TRY(hipMemset(q, 0, bytes));
    h->allocs.push_back(q);
    *ptr = (T *)q;
    return QS_OK;
}

template <typename real> static int create_typed(qs_handle *h) {
    const qs_config &c = h->cfg;
    const size_t E = c.num_envs, N = c.num_agents, T = E * N, D = h->obs_dim, M_ = c.num_obstacles;
    Ptrs<real> p;
    memset(&p, 0, sizeof p);
    int rc;
#define DA(field, count) if ((rc = dalloc(h, &p.field, (count))) != QS_OK) return rc
    const size_t EPB = QS_WAVE / N, NBLK = (E + EPB - 1) / EPB;   // environments per wave block, blocks
    {   // the state allocation (StateBlk, qs_kernels.h): wave-blocked state rows, then the flat per-step outputs; 32-bit offsets
        const size_t R = sizeof(real);
        size_t row = 0;   // inside a block: array after array, each 64 lanes x comps elements (component rows or lane-major: qs_kernels.h)
        auto rows = [&](size_t comps, size_t elem) { size_t o = row; row += comps * 64 * elem; return o; };
        const size_t o_pos = rows(3, R), o_vel = rows(3, R), o_rot = rows(9, R), o_omega = rows(3, R), o_rd = rows(4, R),
            o_cd = rows(4, R), o_ou = rows(4, R),
                     o_goal = rows(3, R), o_ring = rows(4, R), o_sums = rows(3, R), o_flags = rows(1, 4), o_pair = rows(1, 8);
        const size_t block_bytes = row;
        typedef BlkOff<real> BO;
        if ((int)block_bytes != qs_block_bytes((int)R) || block_bytes != BO::bytes || o_pos != BO::pos || o_vel != BO::vel
            || o_rot != BO::rot || o_omega != BO::omega || o_rd != BO::rot_damp ||
            o_cd != BO::cmds_damp || o_ou != BO::ou || o_goal != BO::goal || o_ring != BO::ring || o_sums != BO::sums
                || o_flags != BO::flags || o_pair != BO::pair)
            return fail(QS_ERR_INVALID, "state block layout out of step with BlkOff / qs_block_bytes()");
        size_t off = NBLK * block_bytes;
        auto carve = [&](size_t bytes) { size_t o = off; off = (off + bytes + 255) & ~(size_t)255; return o; };
        const size_t o_newpair = carve(T * 8), o_reward = carve(T * R), o_done = carve(T), o_ohit = carve(T * 4);
        if (off >= ((size_t)1 << 32)) return fail(QS_ERR_UNSUPPORTED,
            "per-drone state exceeds the 4 GiB a buffer resource addresses: use fewer envs per handle");
        char *blk = nullptr;
        if ((rc = dalloc(h, &blk, off)) != QS_OK) return rc;
        p.blk = {blk, (uint32_t)off, (uint32_t)block_bytes, (uint32_t)EPB, (uint32_t)o_pos, (uint32_t)o_vel, (uint32_t)o_rot,
            (uint32_t)o_omega, (uint32_t)o_rd, (uint32_t)o_cd,
                 (uint32_t)o_ou, (uint32_t)o_goal, (uint32_t)o_ring, (uint32_t)o_sums, (uint32_t)o_flags, (uint32_t)o_pair,
                     (uint32_t)o_newpair, (uint32_t)o_reward,
                 // lane-major <=> the specialised 8-wave team kernels step this handle
                 (uint32_t)o_done, (uint32_t)o_ohit, (uint32_t)(h->team == 8 ? 1 : 0)};
        // block 0's first row of each blocked array (what qs_buffers hands out; layout in include/quadswarm.h)
        p.pos = (real *)(blk + o_pos); p.vel = (real *)(blk + o_vel); p.rot = (real *)(blk + o_rot); p.omega = (real *)(blk + o_omega);
        p.rot_damp = (real *)(blk + o_rd); p.cmds_damp = (real *)(blk + o_cd); p.ou = (real *)(blk + o_ou); p.goal = (real *)(blk + o_goal);
        p.dist_ring = (real *)(blk + o_ring); p.dist_sums = (real *)(blk + o_sums); p.flags = (uint32_t *)(blk + o_flags);
        p.pair_mask = (uint64_t *)(blk + o_pair); p.new_pair_mask = (uint64_t *)(blk + o_newpair); p.reward = (real *)(blk + o_reward);
        p.done = (uint8_t *)(blk + o_done); p.obst_hit_idx = (int32_t *)(blk + o_ohit);
    }
    DA(obs, T * D); DA(rew_info, QS_RI_COUNT * T);
    DA(unique_col, E); DA(obst_new, E); DA(room_new, E); DA(counters, QS_CNT_COUNT * E); DA(tick, E); DA(step_ctr, E);
    DA(obst_pos, 2 * E * (M_ ? M_ : 1)); DA(ep_stats, QS_EPS_COUNT * T); DA(ep_counters, QS_CNT_COUNT * E);
    DA(run_sums, QS_SUM_COUNT * T); DA(ep_sums, QS_SUM_COUNT * T);
    DA(obst_count, E); DA(obst_size_env, E); DA(obst_density_env, E);
    {   // --quads_domain_random choice tables
        int32_t *dc = nullptr; real *dd = nullptr, *ds = nullptr;
        if ((rc = dalloc(h, &dc, QS_MAX_DR_CHOICES)) != QS_OK || (rc = dalloc(h, &dd, QS_MAX_DR_CHOICES)) != QS_OK
            || (rc = dalloc(h, &ds, QS_MAX_DR_CHOICES)) != QS_OK) return rc;
        real hd[QS_MAX_DR_CHOICES], hs[QS_MAX_DR_CHOICES];
        for (int q = 0; q < QS_MAX_DR_CHOICES; ++q) { hd[q] = (real)c.dr_density[q]; hs[q] = (real)c.dr_size[q]; }
        HIP_TRY(hipMemcpy(dc, c.dr_obst_count, sizeof c.dr_obst_count, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(dd, hd, sizeof hd, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(ds, hs, sizeof hs, hipMemcpyHostToDevice));
        p.dr_count = dc; p.dr_density = dd; p.dr_size = ds;
    }
    {   // until the first reset draws: the configured density / size
        std::vector<int32_t> cnt(E, c.num_obstacles);
        std::vector<real> sz(E, (real)c.obst_size), dn(E, (real)c.obst_density);
        HIP_TRY(hipMemcpy(p.obst_count, cnt.data(), E * sizeof(int32_t), hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(p.obst_size_env, sz.data(), E * sizeof(real), hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(p.obst_density_env, dn.data(), E * sizeof(real), hipMemcpyHostToDevice));
    }
    DA(scen_real, SR_COUNT * E); DA(scen_int, SI_COUNT * E); DA(scen_omap, 4 * E); DA(scenario_id, E); DA(ep_scenario, E);
    // the running episode's scenario: the resets of the full scenario set (mix: a new one per episode) keep it up to date; the three scenarios of
    {
        // the fast set never change it, so it is filled here (it stayed 0 = static_same_goal for o_static_same_goal / swarm_vs_swarm until round 5:
        // the per-scenario reward keys of the Sample Factory env carried the wrong name - found by tests/test_sf_env_vs_reference_gpu.py)
        std::vector<int32_t> sid(E, c.scenario);
        HIP_TRY(hipMemcpy(p.scenario_id, sid.data(), E * sizeof(int32_t), hipMemcpyHostToDevice));
    }
    // QS_TIMING builds: phase stamps of workgroup 0, then {start, end, HW_ID, XCC_ID, wall start, wall end} of every workgroup
    DA(error_flag, 1); DA(reset_mask, E); DA(timing, 128 + 16 * NBLK);
#undef DA
    {   // run-time reward coefficients (+ proximity slope), read by every launch
        real *rw = nullptr;
        if ((rc = dalloc(h, &rw, QS_REW_COUNT + 1)) != QS_OK) return rc;
        p.rew_rt = rw;
        Consts<real> k;
        fill_consts<real>(c, k);
        real host[QS_REW_COUNT + 1];
        for (int q = 0; q < QS_REW_COUNT; ++q) host[q] = k.rew_coeff[q];
        host[QS_REW_COUNT] = k.prox_ratio;
        HIP_TRY(hipMemcpy(rw, host, sizeof host, hipMemcpyHostToDevice));
    }
    real *act = nullptr;
    if ((rc = dalloc(h, &act, 4 * T)) != QS_OK) return rc;
    h->d_actions = act;
    static_assert(sizeof(Ptrs<float>) == sizeof(Ptrs<double>), "layout");
    memcpy(&h->pf, &p, sizeof p);
    fill_consts<float>(c, h->kf);
    fill_consts<double>(c, h->kd);
    qs_buffers &b = h->bufs;
    memset(&b, 0, sizeof b);
    b.obs = p.obs; b.reward = p.reward; b.done = p.done; b.rew_info = p.rew_info; b.actions = act;
    b.pos = p.pos; b.vel = p.vel; b.omega = p.omega; b.rot = p.rot; b.thrust_rot_damp = p.rot_damp; b.thrust_cmds_damp = p.cmds_damp;
    b.ou_state = p.ou; b.goal = p.goal; b.flags = p.flags; b.obst_hit_idx = p.obst_hit_idx; b.col_pair_mask = p.pair_mask;
    b.new_pair_mask = p.new_pair_mask; b.unique_col_mask = p.unique_col; b.obst_new_mask = p.obst_new; b.room_new_mask = p.room_new;
    b.counters = p.counters; b.tick = p.tick; b.obst_pos = p.obst_pos; b.ep_stats = p.ep_stats; b.ep_counters = p.ep_counters;
    b.run_sums = p.run_sums; b.ep_sums = p.ep_sums;
    b.obst_count = p.obst_count; b.obst_size_env = p.obst_size_env; b.obst_density_env = p.obst_density_env;
    b.error_flag = p.error_flag; b.scenario_id = p.scenario_id; b.ep_scenario = p.ep_scenario; b.obs_dim = h->obs_dim;
    b.real_size = sizeof(real);
    b.state_block_bytes = (int32_t)p.blk.block_bytes; b.envs_per_block = (int32_t)EPB; b.state_lane_major = (int32_t)p.blk.lane_major;
    // what a deep copy of one reference env carries (quad_experience_replay.py:99-104 deep-copies the whole env): every
    // per-drone and per-env array except the noise-stream position (step_ctr: a restored env draws fresh noise, as the
    // reference's does from the global numpy stream) and the per-step outputs
    auto &sa = h->snap_arrays;
    sa.clear();
#define SNAP_T(field, comps) sa.push_back({(char *)p.field, sizeof(*p.field), (size_t)(comps), T, N, 0, 0, 0})
    // wave-blocked state array: `comps` rows of 64 elements, or (lane-major) the N drones of an env as ONE contiguous piece of N * comps
    // elements
#define SNAP_B(field, comps) sa.push_back(p.blk.lane_major ? qs_handle::SnapArray{(char *)p.field, sizeof(*p.field), 1, 64 * (size_t)(comps), N * (size_t)(comps), 0, EPB, (size_t)p.blk.block_bytes} \
                                                           : qs_handle::SnapArray{(char *)p.field, sizeof(*p.field), (size_t)(comps), 64, N, 0, EPB, (size_t)p.blk.block_bytes})
#define SNAP_E(field, comps) sa.push_back({(char *)p.field, sizeof(*p.field), (size_t)(comps), E, 1, 0, 0, 0})
    SNAP_B(pos, 3); SNAP_B(vel, 3); SNAP_B(rot, 9); SNAP_B(omega, 3); SNAP_B(rot_damp, 4); SNAP_B(cmds_damp, 4); SNAP_B(ou, 4);
    SNAP_B(goal, 3);
    SNAP_B(flags, 1); SNAP_B(pair_mask, 1); SNAP_T(new_pair_mask, 1); SNAP_T(obst_hit_idx, 1); SNAP_B(dist_ring, 4); SNAP_B(dist_sums, 3);
    SNAP_T(run_sums, QS_SUM_COUNT); sa.back().kind = 2;
    sa.push_back({(char *)p.obs, sizeof(real), 1, T * D, N * D, 1, 0, 0});                     // the observation that goes with the state
    SNAP_E(unique_col, 1); SNAP_E(obst_new, 1); SNAP_E(room_new, 1); SNAP_E(counters, QS_CNT_COUNT); SNAP_E(tick, 1);
    SNAP_E(scen_real, SR_COUNT); SNAP_E(scen_int, SI_COUNT); SNAP_E(scen_omap, 4); SNAP_E(scenario_id, 1);
    SNAP_E(obst_count, 1); SNAP_E(obst_size_env, 1); SNAP_E(obst_density_env, 1);
    sa.push_back({(char *)p.obst_pos, sizeof(real), 2, E * (M_ ? M_ : 1), (M_ ? M_ : 1), 0, 0, 0});
#undef SNAP_T
#undef SNAP_B
#undef SNAP_E
    h->snap_bytes = 0;
    for (const auto &a : sa) h->snap_bytes += (a.elem * a.comps * a.per_env + 15) & ~(size_t)15;
    return QS_OK;
}

extern "C" {

int qs_version(void) { return QS_VERSION; }
size_t qs_sizeof_config(void) { return sizeof(qs_config); }
const char *qs_last_error(void) { return g_last_error.c_str(); }

int qs_obs_dim(const qs_config *c) {
    int self = c->obs_repr == 0 ? 18 : (c->obs_repr == 1 ? 19 : 24);
    return self + 6 * c->num_neighbors + (c->use_obstacles ? 9 : 0);
}

int qs_default_config(qs_config *c, int32_t num_envs, int32_t num_agents) {
    // Crazyflie constants as derived by the reference at construction time; the python
    // host layer (quad-swarm-rl_amd/airframe.py) re-derives them from the link geometry and overrides these.
    memset(c, 0, sizeof *c);
    c->num_envs = num_envs; c->num_agents = num_agents; c->precision = QS_PRECISION_F32; c->seed = 0;
    c->mass = 0.028000000000000008; c->arm = 0.04596194077712559;
    c->inertia[0] = 1.3669232142857143e-05; c->inertia[1] = 1.4356732142857143e-05; c->inertia[2] = 2.656158333333334e-05;
    const double pc[4][3] = {{-0.0325, -0.0325, 0}, {-0.0325, 0.0325, 0}, {0.0325, 0.0325, 0}, {0.0325, -0.0325, 0}};
    const double ccw[4] = {-1, 1, -1, 1};
    for (int m = 0; m < 4; ++m) {
        for (int q = 0; q < 3; ++q) c->prop_cross[m][q] = pc[m][q];
        c->prop_ccw[m] = ccw[m];
        c->thrust_max[m] = 9.81 * c->mass * 1.9 / 4.0;
        c->torque_max[m] = 0.006 * c->thrust_max[m];
    }
    c->dt = 1.0 / 200.0; c->sim_steps = 2;
    c->motor_tau_up = c->motor_tau_down = 4 * c->dt / (0.15 + 1e-6);
    c->motor_linearity = 1.0; c->vel_damp = 0; c->damp_omega_quadratic = 0; c->omega_max = 40.0; c->gravity = 9.81;
    // the numba path (floor_mode below): OUNoiseNumba keeps theta / sigma in float32 members (numba_utils.py:67-74) - the values the jitted
    // reference computes with are the float32 roundings widened back to double (config.make_config: numba_float32_ou)
    c->thrust_noise_sigma = (double)(float)(0.2 * 0.05); c->ou_theta = (double)(float)0.15;
    c->ep_len = (int)(15.0 / (c->dt * c->sim_steps));
    c->room_lo[0] = -5; c->room_lo[1] = -5; c->room_lo[2] = 0; c->room_hi[0] = 5; c->room_hi[1] = 5; c->room_hi[2] = 10;
    c->floor_mode = QS_FLOOR_NUMBA;
    { double s = 0; int n = 0; do { s += c->dt; ++n; } while (!(s > 0.5)); c->svd_period = n; }
    c->sense_noise = 1; c->obs_repr = QS_OBS_XYZ_VXYZ_R_OMEGA;
    c->pos_norm_std = 0.005; c->vel_norm_std = 0.01; c->gyro_noise_density = 0.000175;
    c->num_neighbors = num_agents > 6 ? 6 : num_agents - 1;
    c->use_downwash = 0; c->use_obstacles = 0; c->scenario = QS_SCENARIO_STATIC_SAME_GOAL;
    c->collision_threshold = 2.0 * c->arm; c->collision_falloff_threshold = 4.0 * c->arm;
    const double rc[QS_REW_COUNT] = {1.0, 0.05, 1.0, 1.0, 0.1, 5.0, 4.0, 5.0};
    for (int q = 0; q < QS_REW_COUNT; ++q) c->rew_coeff[q] = rc[q];
    c->spawn_box = 2.0; c->approach_goal_metric = 0.5;
    for (int q = 0; q < 3; ++q) { c->nbr_clip_pos[q] = 10.0; c->nbr_clip_vel[q] = 6.0; }
    c->obst_size = 1.0; c->obst_density = 0.2; c->obst_area[0] = 6; c->obst_area[1] = 6; c->num_obstacles = 0;
    c->write_rew_info = 1;
    c->episode_sums = 0;
    return QS_OK;
}

int qs_create(const qs_config *cfg, int device, qs_handle **out) {
    if (!cfg || !out) return fail(QS_ERR_INVALID, "null argument");
    int rc = validate(cfg);
    if (rc != QS_OK) return rc;
    int ndev = 0;
    HIP_TRY(hipGetDeviceCount(&ndev));
    if (device < 0 || device >= ndev) return fail(QS_ERR_HIP, "no such HIP device");
    HIP_TRY(hipSetDevice(device));
    qs_handle *h = new qs_handle();
    h->cfg = *cfg;
    h->device = device;
    h->real_size = cfg->precision == QS_PRECISION_F64 ? 8 : 4;
    h->obs_dim = qs_obs_dim(cfg);
    h->epb = QS_WAVE / cfg->num_agents;
    h->blocks = (cfg->num_envs + h->epb - 1) / h->epb;
    // Kernel flavour: a team of 4 waves per workgroup shortens the per-step critical path when the batch cannot fill the
    // chip anyway (<= 8 waves per CU, team_default); the single-wave kernels do less total work per drone and win on throughput.
    // QS_TEAM=0/1 in the environment overrides the choice (both flavours produce the same results).
    {
        hipDeviceProp_t prop;
        int cus = 256;
        if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0) cus = prop.multiProcessorCount;
        h->team = team_default(h->blocks, cus, cfg->num_agents) ? QS_TEAM_WAVES : 0;
        h->cus = cus;
        const char *ev = getenv("QS_TEAM");
        if (ev && ev[0] == '0') h->team = 0;
        else if (ev && (ev[0] == '1' || ev[0] == '4' || ev[0] == '8')) h->team = QS_TEAM_WAVES;
    }
    // Config-specialised kernels.  QS_SPEC = "jit" (default): use the cached code object of this configuration, building it
    // first if needed (one hipcc run, ~5 s, cached next to the library); "cache": use it only if it is already there;
    // "off": always the generic kernels.  Any failure falls back to the generic kernels (same results, slower).
    {
        const char *ev = getenv("QS_SPEC"), *tv = getenv("QS_TEAM");
        const std::string mode = (ev && ev[0]) ? ev : "jit";
        const int spec_team = h->team ? ((tv && tv[0] == '4') ? 4 : ((tv && tv[0] == '8') ? 8 : spec_team_waves(cfg->num_agents))) : 0;
        const LdsLayout sl = lds_layout(h->real_size, QS_WAVE, cfg->num_agents, h->epb, h->obs_dim, cfg->num_obstacles,
            cfg->num_neighbors, spec_team, scenario_is_full(cfg->scenario), cfg->scenario,
                                           spec_rows_per_pass(cfg, spec_team));
        const bool require = mode == "require";
        if (mode == "off" || mode == "0") h->spec_note = "QS_SPEC=off";
        else if (sl.total > 64 * 1024) h->spec_note = "LDS layout above the 64 KiB a module-loaded kernel gets";
        else {
            const std::string path = spec_ensure(cfg, spec_team, mode == "jit" || require);
            if (!path.empty()) {
                if (hipModuleLoad(&h->spec_mod, path.c_str()) == hipSuccess &&
                    hipModuleGetFunction(&h->spec_step, h->spec_mod, "qs_spec_step") == hipSuccess &&
                    hipModuleGetFunction(&h->spec_rollout, h->spec_mod, "qs_spec_rollout") == hipSuccess &&
                    hipModuleGetFunction(&h->spec_reset, h->spec_mod, "qs_spec_reset") == hipSuccess) {
                    h->team = spec_team;   // specialised kernels in use
                    if (spec_team > 0 && hipModuleGetFunction(&h->spec_gated, h->spec_mod,
                        "qs_spec_gated") != hipSuccess) { (void)hipGetLastError(); h->spec_gated = nullptr; }
                } else {
                    (void)hipGetLastError();
                    if (h->spec_mod) { (void)hipModuleUnload(h->spec_mod); h->spec_mod = nullptr; }
                    h->spec_step = h->spec_rollout = h->spec_reset = h->spec_gated = nullptr;
                    h->spec_note = "cannot load " + path;
                }
            } else {
                h->spec_note = g_last_error;
            }
        }
        // The fallback is LOUD: one line on stderr, the reason kept for qs_spec_status(), and QS_SPEC=require turns it into an error
        // (the generic kernels give the same results - tests/test_fp32_parity_gpu.py - at ~1.3x the step time).
        if (!h->spec_step && mode != "off" && mode != "0") {
            if (require && sl.total <= 64 * 1024) { const std::string why = h->spec_note; delete h;
                return fail(QS_ERR_UNSUPPORTED, "QS_SPEC=require: no config-specialised kernels: " + why); }
            fprintf(stderr, "quadswarm_hip: WARNING: running the GENERIC step kernels (%s)\n", h->spec_note.c_str());
        }
    }
    h->lds = lds_layout(h->real_size, QS_WAVE, cfg->num_agents, h->epb, h->obs_dim, cfg->num_obstacles, cfg->num_neighbors, h->team,
        scenario_is_full(cfg->scenario), cfg->scenario,
                        h->spec_step ? spec_rows_per_pass(cfg, h->team) : QS_WAVE);
    h->full = scenario_is_full(cfg->scenario);
    rc = (h->real_size == 8) ? create_typed<double>(h) : create_typed<float>(h);
    if (rc == QS_OK) {
        if (hipMalloc((void **)&h->d_state_buf, sizeof(double) * QS_MAX_AGENTS * QS_STATE_STRIDE) != hipSuccess ||
            hipMalloc((void **)&h->d_tick_io, sizeof(int32_t)) != hipSuccess ||
            hipHostMalloc((void **)&h->h_mask, (size_t)cfg->num_envs) != hipSuccess)
            rc = fail(QS_ERR_HIP, "allocation failed");
    }
    if (rc != QS_OK) { qs_destroy(h); return rc; }
    if (h->lds.total > 64 * 1024) {
        const void *fns[] = {(const void *)qs_step_kernel<float>, (const void *)qs_step_kernel<double>,
            (const void *)qs_rollout_kernel<float>,
                             (const void *)qs_rollout_kernel<double>, (const void *)qs_step_kernel_full<float>,
                                 (const void *)qs_step_kernel_full<double>,
                             (const void *)qs_rollout_kernel_full<float>, (const void *)qs_rollout_kernel_full<double>,
                             (const void *)qs_step_team<float>, (const void *)qs_step_team<double>, (const void *)qs_rollout_team<float>,
                             (const void *)qs_rollout_team<double>, (const void *)qs_step_team_full<float>,
                                 (const void *)qs_step_team_full<double>,
                             (const void *)qs_rollout_team_full<float>, (const void *)qs_rollout_team_full<double>,
                             (const void *)qs_reset_kernel<float, false>, (const void *)qs_reset_kernel<double, false>,
                             (const void *)qs_reset_kernel<float, true>, (const void *)qs_reset_kernel<double, true>};
        for (const void *fn : fns)
            if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, h->lds.total) != hipSuccess) {
                qs_destroy(h);
                return fail(QS_ERR_HIP, "cannot raise dynamic LDS limit");
            }
    }
    *out = h;
    return QS_OK;
}

// 1 if the handle runs config-specialised kernels, 0 if the generic ones
int qs_is_specialized(qs_handle *h) { return (h && h->spec_step) ? 1 : 0; }
int qs_spec_status(qs_handle *h, char *why_out, int cap) {
    if (!h) return fail(QS_ERR_INVALID, "null handle");
    if (why_out && cap > 0) { snprintf(why_out, (size_t)cap, "%s", h->spec_note.c_str()); }
    return h->spec_step ? 1 : 0;
}
// bit 0: config-specialised code object, bit 1: team kernels, bit 2: full scenario set, bits 8..15: waves per workgroup
int qs_kernel_flavor(qs_handle *h) { return h
    ? ((h->spec_step ? 1 : 0) | (h->team ? 2 : 0) | (h->full ? 4 : 0) | ((h->team ? h->team : 1) << 8)) : 0; }

int qs_destroy(qs_handle *h) {
    if (!h) return QS_OK;
    (void)hipSetDevice(h->device);
    (void)hipDeviceSynchronize();
    if (h->spec_mod) { (void)hipModuleUnload(h->spec_mod); h->spec_mod = nullptr; }
    if (h->snap_pool) (void)hipFree(h->snap_pool);
    for (void *q : h->allocs) (void)hipFree(q);
    if (h->d_state_buf) (void)hipFree(h->d_state_buf);
    if (h->d_tick_io) (void)hipFree(h->d_tick_io);
    if (h->h_mask) (void)hipHostFree(h->h_mask);
    if (h->d_tape) (void)hipFree(h->d_tape);
    if (h->d_tape_pos) (void)hipFree(h->d_tape_pos);
    if (h->d_gate) (void)hipFree(h->d_gate);
    if (h->gate_stream) (void)hipStreamDestroy(h->gate_stream);
    if (h->gate_ev_in) (void)hipEventDestroy(h->gate_ev_in);
    if (h->gate_ev_out) (void)hipEventDestroy(h->gate_ev_out);
    for (auto &ev : h->events) { (void)hipEventDestroy(ev.first); (void)hipEventDestroy(ev.second); }
    if (h->graph_exec) (void)hipGraphExecDestroy(h->graph_exec);
    if (h->cap_stream) (void)hipStreamDestroy(h->cap_stream);
    delete h;
    return QS_OK;
}

static int launch_reset(qs_handle *h, hipStream_t s) {
    if (h->d_tape) {
        hipError_t e = (hipError_t)qs_tape_launch(0, &h->cfg, h->obs_dim, h->full ? 1 : 0, h->real_size, h->real_size == 8
            ? (const void *)&h->kd : (const void *)&h->kf, &h->pf, nullptr, s);
        if (e != hipSuccess) return fail(QS_ERR_HIP, std::string("tape reset kernel: ") + hipGetErrorString(e));
        return QS_OK;
    }
    Ptrs<float> pf = h->pf;   // this launch's pointers: the observation rows go to the target of qs_set_obs_target, if one is set
    if (h->obs_target) pf.obs = (float *)h->obs_target;
    if (h->spec_reset) {
        Ptrs<double> pd; memcpy(&pd, &pf, sizeof pd);
        void *args[] = {h->real_size == 8 ? (void *)&h->kd : (void *)&h->kf, h->real_size == 8 ? (void *)&pd : (void *)&pf, &h->lds,
            &h->epb};
        HIP_TRY(hipModuleLaunchKernel(h->spec_reset, h->blocks, 1, 1, QS_WAVE, 1, 1, h->lds.total, s, args, nullptr));
        return QS_OK;
    }
    if (h->real_size == 8) {
        Ptrs<double> p; memcpy(&p, &pf, sizeof p);
        if (h->full) hipLaunchKernelGGL((qs_reset_kernel<double, true>), dim3(h->blocks), dim3(QS_WAVE), h->lds.total, s, h->kd, p,
            h->lds, h->epb);
        else hipLaunchKernelGGL((qs_reset_kernel<double, false>), dim3(h->blocks), dim3(QS_WAVE), h->lds.total, s, h->kd, p, h->lds,
            h->epb);
    } else {
        if (h->full) hipLaunchKernelGGL((qs_reset_kernel<float, true>), dim3(h->blocks), dim3(QS_WAVE), h->lds.total, s, h->kf, pf,
            h->lds, h->epb);
        else hipLaunchKernelGGL((qs_reset_kernel<float, false>), dim3(h->blocks), dim3(QS_WAVE), h->lds.total, s, h->kf, pf, h->lds,
            h->epb);
    }
    HIP_TRY(hipGetLastError());
    return QS_OK;
}

// A gated launch (qs_step_gated) runs on the library's own stream and nothing waits for it by itself: every other entry point that touches
// the handle's device state first joins it - stream-ordered where the call takes a stream, on the host where it copies synchronously.
static int gate_join_stream(qs_handle *h, hipStream_t s) {
    if (h->gate_pending) { HIP_TRY(hipStreamWaitEvent(s, h->gate_ev_out, 0)); }
    return QS_OK;
}
static int gate_join_host(qs_handle *h) {
    if (h->gate_pending) { HIP_TRY(hipStreamSynchronize(h->gate_stream)); h->gate_pending = false; }
    return QS_OK;
}

int qs_reset(qs_handle *h, const uint8_t *env_mask_host, void *stream) {
    if (!h) return fail(QS_ERR_INVALID, "null handle");
    HIP_TRY(hipSetDevice(h->device));
    if (int jr = gate_join_stream(h, (hipStream_t)stream)) return jr;
    hipStream_t s = (hipStream_t)stream;
    const int E = h->cfg.num_envs;
    for (int e = 0; e < E; ++e) h->h_mask[e] = env_mask_host ? (env_mask_host[e] ? 1 : 0) : 1;
    HIP_TRY(hipMemcpyAsync(h->pf.reset_mask, h->h_mask, (size_t)E, hipMemcpyHostToDevice, s));
    // the replay wrapper's bookkeeping of an explicit reset (before the reset kernel zeroes the running
    if (h->replay_on && h->replay_stepped) {
        // sums); the reset that starts the very first episode is already in the history (qs_replay_enable)
        hipLaunchKernelGGL(qs_replay_reset_kernel, dim3((E + QS_WAVE - 1) / QS_WAVE), dim3(QS_WAVE), 0, s, h->rp,
            (const uint8_t *)h->pf.reset_mask);
        HIP_TRY(hipGetLastError());
    }
    int rc = launch_reset(h, s);
    if (rc != QS_OK) return rc;
    HIP_TRY(hipStreamSynchronize(s));   // h_mask is reused by the next call
    return QS_OK;
}

static int launch_step(qs_handle *h, const void *actions, hipStream_t s, int ksteps = 1, bool gated = false) {
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (h->profiling) {
        if (h->events_used == h->events.size()) {
            hipEvent_t a, b;
            HIP_TRY(hipEventCreate(&a)); HIP_TRY(hipEventCreate(&b));
            h->events.emplace_back(a, b);
        }
        e0 = h->events[h->events_used].first; e1 = h->events[h->events_used].second;
        ++h->events_used;
        HIP_TRY(hipEventRecord(e0, s));
    }
    if (h->d_tape) {   // noise-tape flavour: one launch per control step
        const size_t stride = (size_t)h->cfg.num_envs * h->cfg.num_agents * 4 * (size_t)h->real_size;
        for (int t = 0; t < ksteps; ++t) {
            hipError_t e = (hipError_t)qs_tape_launch(1, &h->cfg, h->obs_dim, h->full ? 1 : 0, h->real_size, h->real_size == 8
                ? (const void *)&h->kd : (const void *)&h->kf, &h->pf,
                                                     (const char *)actions + stride * t, s);
            if (e != hipSuccess) return fail(QS_ERR_HIP, std::string("tape step kernel: ") + hipGetErrorString(e));
        }
        if (h->profiling) HIP_TRY(hipEventRecord(e1, s));
        return QS_OK;
    }
    Ptrs<float> pf = h->pf;   // this launch's pointers (qs_set_obs_target)
    if (h->obs_target) pf.obs = (float *)h->obs_target;
    // the fused exchange epilogue exists in the one-step kernels only: a multi-step launch would advance the environments without sending
    // their rows and desynchronise push / wait sequence numbers (qs_step_many splits into single steps while an exchange is set)
    if (pf.xchg != nullptr && ksteps > 1) return fail(QS_ERR_UNSUPPORTED,
        "multi-step launches do not exchange observation rows: qs_set_obs_exchange is active");
    // the multi-step team kernels read the exchange slot as their gate (qs_step_team.inc)
    if (gated) pf.xchg = (const qsx::XchgDev *)h->d_gate;
    if (h->spec_step) {
        Ptrs<double> pd; memcpy(&pd, &pf, sizeof pd);
        void *args[] = {h->real_size == 8 ? (void *)&h->kd : (void *)&h->kf, h->real_size == 8 ? (void *)&pd : (void *)&pf,
            (void *)&actions, &h->lds, &h->epb, &ksteps};
        const int grid = h->blocks;
        if (gated && !h->spec_gated) return fail(QS_ERR_UNSUPPORTED,
            "the specialised code object of this handle has no resident-state kernel");
        HIP_TRY(hipModuleLaunchKernel(gated ? h->spec_gated : (ksteps == 1 ? h->spec_step : h->spec_rollout), grid, 1, 1, h->team
            ? QS_WAVE * h->team : QS_WAVE, 1, 1, h->lds.total, s, args, nullptr));
        if (h->profiling) HIP_TRY(hipEventRecord(e1, s));
        return QS_OK;
    }
#define QS_LAUNCH(KERNEL, THREADS, CONSTS, PTRS, TYPE, ...) hipLaunchKernelGGL(KERNEL<TYPE>, dim3(h->blocks), dim3(THREADS), h->lds.total, s, CONSTS, PTRS, \
                                                                             (const TYPE *)actions, h->lds, h->epb, ##__VA_ARGS__)
#define QS_LAUNCH_ALL(CONSTS, PTRS, TYPE) do { \
        if (h->team) { \
            if (gated) { if (h->full) QS_LAUNCH(qs_gated_team_full, QS_TEAM_THREADS, CONSTS, PTRS, TYPE, ksteps); else QS_LAUNCH(qs_gated_team, QS_TEAM_THREADS, CONSTS, PTRS, TYPE, ksteps); } \
            else if (ksteps == 1) { if (h->full) QS_LAUNCH(qs_step_team_full, QS_TEAM_THREADS, CONSTS, PTRS, TYPE); else QS_LAUNCH(qs_step_team, QS_TEAM_THREADS, CONSTS, PTRS, TYPE); } \
            else { if (h->full) QS_LAUNCH(qs_rollout_team_full, QS_TEAM_THREADS, CONSTS, PTRS, TYPE, ksteps); else QS_LAUNCH(qs_rollout_team, QS_TEAM_THREADS, CONSTS, PTRS, TYPE, ksteps); } \
        } else { \
            if (ksteps == 1) { if (h->full) QS_LAUNCH(qs_step_kernel_full, QS_WAVE, CONSTS, PTRS, TYPE); else QS_LAUNCH(qs_step_kernel, QS_WAVE, CONSTS, PTRS, TYPE); } \
            else { if (h->full) QS_LAUNCH(qs_rollout_kernel_full, QS_WAVE, CONSTS, PTRS, TYPE, ksteps); else QS_LAUNCH(qs_rollout_kernel, QS_WAVE, CONSTS, PTRS, TYPE, ksteps); } \
        } } while (0)
    if (h->real_size == 8) {
        Ptrs<double> p; memcpy(&p, &pf, sizeof p);
        QS_LAUNCH_ALL(h->kd, p, double);
    } else {
        QS_LAUNCH_ALL(h->kf, pf, float);
    }
#undef QS_LAUNCH_ALL
#undef QS_LAUNCH
    HIP_TRY(hipGetLastError());
    if (h->profiling) HIP_TRY(hipEventRecord(e1, s));
    return QS_OK;   // the auto-reset is the tail of the step kernel itself
}

// the replay wrapper's step() / new_episode() for every environment, behind each control step (qs_replay_enable)
static int launch_replay(qs_handle *h, hipStream_t s) {
    hipLaunchKernelGGL(qs_replay_kernel, dim3(h->cfg.num_envs), dim3(QS_WAVE), 0, s, h->rp);
    HIP_TRY(hipGetLastError());
    return QS_OK;
}

int qs_step(qs_handle *h, const void *actions_dev, void *stream) {
    if (!h) return fail(QS_ERR_INVALID, "null handle");
    HIP_TRY(hipSetDevice(h->device));
    if (h->gate_pending) { if (int jr = gate_join_stream(h, (hipStream_t)stream)) return jr; }
    int rc = launch_step(h, actions_dev ? actions_dev : h->d_actions, (hipStream_t)stream);
    if (rc == QS_OK && h->replay_on) { rc = launch_replay(h, (hipStream_t)stream); h->replay_stepped = true; }
    return rc;
}

int qs_step_many(qs_handle *h, const void *actions_dev, int32_t k, void *stream) {
    if (!h || !actions_dev || k < 0) return fail(QS_ERR_INVALID, "bad argument");
    HIP_TRY(hipSetDevice(h->device));
    if (h->gate_pending) { if (int jr = gate_join_stream(h, (hipStream_t)stream)) return jr; }
    const size_t stride = (size_t)h->cfg.num_envs * h->cfg.num_agents * 4 * h->real_size;
    // per-step HIP events / the replay kernel behind every step / the fused exchange epilogue: one launch per control step
    if (h->profiling || h->replay_on || h->pf.xchg) {
        for (int32_t t = 0; t < k; ++t) {
            int rc = launch_step(h, (const char *)actions_dev + stride * t, (hipStream_t)stream, 1);
            if (rc == QS_OK && h->replay_on) { rc = launch_replay(h, (hipStream_t)stream); h->replay_stepped = true; }
            if (rc != QS_OK) return rc;
        }
        return QS_OK;
    }
    // open-loop rollout: the step kernel keeps the state in registers across up to QS_STEPS_PER_LAUNCH control steps
    const int32_t per = 64;
    for (int32_t t = 0; t < k; t += per) {
        int rc = launch_step(h, (const char *)actions_dev + stride * t, (hipStream_t)stream, (k - t) < per ? (k - t) : per);
        if (rc != QS_OK) return rc;
    }
    return QS_OK;
}

// ------------------------------------------------------------------------------------------------
// Resident-state stepping (include/quadswarm.h)
// ------------------------------------------------------------------------------------------------
// the benchmark's / the tests' producer: per control step it (closed_loop: waits until the outputs of the previous step of ITS workgroups
// are published, else: only until the ring slot is free), copies the group's share of the next action batch from a table resident in HBM
// into the ring - written through the L2 - and raises the group's sequence word
// `sums` (qs_gate_produce_verify, closed loop only): the kernel is also a CONSUMER of the stepper's outputs the way the protocol describes
// one - having seen done_flag >= s for its workgroups it executes an agent-scope acquire and reads the observation rows and rewards of step
// s with plain loads - and records a checksum (the sum of their 32-bit words) per step and group, WHILE the gated launch is resident and
// working on step s + 1.  tests/test_gated_gpu.py compares the sums with those of a one-launch-per-step twin.
__global__ void __launch_bounds__(256) qs_gate_producer_kernel(qsx::Gate *G, const char *src, unsigned int n_src,
    unsigned long long seq0, int k, int closed_loop,
                                                                unsigned long long wg_bytes, unsigned long long batch_bytes,
                                                                unsigned long long *sums, const unsigned int *obs_words,
                                                                    const unsigned int *rew_words,
                                                                unsigned long long obs_words_per_wg, unsigned long long rew_words_per_wg,
                                                                unsigned long long obs_words_total, unsigned long long rew_words_total) {
    const unsigned int grp = blockIdx.x, w0 = grp * G->wg_per_group, w1 = (w0 + G->wg_per_group < G->blocks)
        ? w0 + G->wg_per_group : G->blocks;
    const unsigned long long lo = (unsigned long long)w0 * wg_bytes, hi0 = (unsigned long long)w1 * wg_bytes, hi = hi0 < batch_bytes
        ? hi0 : batch_bytes;
    __shared__ int dead;
    __shared__ unsigned long long acc;
    if (threadIdx.x == 0) dead = 0;
    __syncthreads();
    for (int t = 0; t <= k; ++t) {
        const unsigned long long seq = seq0 + (unsigned long long)t + 1;
        if (t == k && sums == nullptr) break;   // (the extra round only reads the last step's outputs)
        const unsigned long long need = (closed_loop || t == k) ? seq - 1 : (seq > G->ring_len ? seq - G->ring_len : 0);
        if (need > 0 && !dead) {   // every workgroup of the group, 256 at a time (a group may hold all ~512 workgroups of a team handle)
            for (unsigned int w = w0 + threadIdx.x; w < w1; w += 256)
                if (!qsx::poll_ge_agent(&G->done_flag[w], need, G->timeout_ticks)) { dead = 1; atomicOr(&G->status, 2u); break; }
        }
        __syncthreads();
        // the outputs of sequence number seq - 1 (a step of THIS call), read the way a policy would read them
        if (sums != nullptr && t >= 1) {
            if (threadIdx.x == 0) acc = 0;
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");   // buffer_inv sc1: this XCD's L2 may hold the rows of the step before
            __syncthreads();
            unsigned long long part = 0;
            const unsigned long long o0 = (unsigned long long)w0 * obs_words_per_wg, o1u = (unsigned long long)w1 * obs_words_per_wg,
                o1 = o1u < obs_words_total ? o1u : obs_words_total;
            for (unsigned long long j = o0 + threadIdx.x; j < o1; j += 256) part += obs_words[j];
            const unsigned long long r0 = (unsigned long long)w0 * rew_words_per_wg, r1u = (unsigned long long)w1 * rew_words_per_wg,
                r1 = r1u < rew_words_total ? r1u : rew_words_total;
            for (unsigned long long j = r0 + threadIdx.x; j < r1; j += 256) part += rew_words[j];
            atomicAdd(&acc, part);
            __syncthreads();
            if (threadIdx.x == 0) sums[(unsigned long long)(t - 1) * G->groups + grp] = acc;
            __syncthreads();
        }
        if (t == k) break;
        const char *from = src + ((seq - 1) % n_src) * batch_bytes;
        char *to = G->act_ring + ((seq - 1) % G->ring_len) * G->act_stride;
        // system-scope write-through: the flag below must not become visible before the batch (`sc1` alone was seen to let it: one run in
        // three of the run-ahead parity test read a stale batch)
        for (unsigned long long off = lo + 16ull * threadIdx.x; off < hi; off += 16ull * 256) qsx::st16_wt(to + off,
            *(const qsx::u32x4_t *)(from + off));
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) qsx::st_agent(&G->act_flag[grp], seq);
    }
}

int qs_gate_create(qs_handle *h, int32_t ring_len, int32_t wg_per_group) {
    if (!h || ring_len < 1 || ring_len > 65536 || wg_per_group < 1) return fail(QS_ERR_INVALID, "qs_gate_create: bad argument");
    if (h->d_gate) return fail(QS_ERR_INVALID, "qs_gate_create: the handle has a gate already");
    if (!h->team) return fail(QS_ERR_UNSUPPORTED, "resident-state stepping lives in the team kernels (batches up to ~8 waves per CU, see qs_kernel_flavor): larger batches are bandwidth-bound, not launch-bound");
    if (h->replay_on || h->d_tape) return fail(QS_ERR_UNSUPPORTED,
        "resident-state stepping is not available with the device-side replay wrapper or a noise tape");
    HIP_TRY(hipSetDevice(h->device));
    const size_t T = (size_t)h->cfg.num_envs * h->cfg.num_agents, stride = (T * 4 * (size_t)h->real_size + 255) & ~(size_t)255;
    const unsigned int groups = (unsigned int)((h->blocks + wg_per_group - 1) / wg_per_group);
    const size_t o_ring = 256, o_act = o_ring + stride * (size_t)ring_len, o_done = o_act + (((size_t)groups * 8 + 255) & ~(size_t)255),
        total = o_done + (((size_t)h->blocks * 8 + 255) & ~(size_t)255);
    // Fine-grained (uncached) device memory: the ring and the sequence words are handed between kernels that run CONCURRENTLY, mostly on
    // different XCDs, whose L2s are not coherent with each other - an `sc1` load that hits a stale line of its own XCD's L2 is how a first
    // version of this took 7.8 ms per closed-loop step (profiles/r04c_bench_c2_default.json); uncached memory has no such line
    char *base = nullptr;
    if (hipExtMallocWithFlags((void **)&base, total, hipDeviceMallocUncached) != hipSuccess) {
        (void)hipGetLastError();
        return fail(QS_ERR_HIP, "qs_gate_create: fine-grained (uncached) device memory is not available: resident-state stepping needs it for its action ring and sequence words");
    }
    HIP_TRY(hipMemset(base, 0, total));
    qsx::Gate g;
    memset(&g, 0, sizeof g);
    int khz = 100000;
    if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, h->device) != hipSuccess || khz <= 0) { (void)hipGetLastError();
        khz = 100000; }
    long ms = 500;
    if (const char *ev = getenv("QS_GATE_TIMEOUT_MS")) { const long v = atol(ev); if (v > 0) ms = v; }
    g.timeout_ticks = (unsigned long long)khz * (unsigned long long)ms;
    g.act_ring = base + o_ring; g.act_stride = stride; g.ring_len = (unsigned int)ring_len; g.groups = groups;
    g.wg_per_group = (unsigned int)wg_per_group; g.blocks = (unsigned int)h->blocks;
    g.act_flag = (unsigned long long *)(base + o_act); g.done_flag = (unsigned long long *)(base + o_done);
    HIP_TRY(hipMemcpy(base, &g, sizeof g, hipMemcpyHostToDevice));
    HIP_TRY(hipDeviceSynchronize());
    {
        int least = 0, greatest = 0;
        hipError_t er = hipDeviceGetStreamPriorityRange(&least, &greatest);
        if (er == hipSuccess && !h->gate_stream) er = hipStreamCreateWithPriority(&h->gate_stream, hipStreamNonBlocking, greatest);
        if (er == hipSuccess && !h->gate_ev_in) er = hipEventCreateWithFlags(&h->gate_ev_in, hipEventDisableTiming);
        if (er == hipSuccess && !h->gate_ev_out) er = hipEventCreateWithFlags(&h->gate_ev_out, hipEventDisableTiming);
        if (er != hipSuccess) { (void)hipFree(base); return fail(QS_ERR_HIP, std::string("qs_gate_create: ") + hipGetErrorString(er)); }
    }
    h->d_gate = (qsx::Gate *)base; h->gate_host = g; h->gate_step_seq = 0; h->gate_prod_seq = 0;
    return QS_OK;
}

int qs_gate_info(qs_handle *h, qs_gate_info_t *out) {
    if (!h || !out) return fail(QS_ERR_INVALID, "null argument");
    if (!h->d_gate) return fail(QS_ERR_INVALID, "no gate: call qs_gate_create first");
    const qsx::Gate &g = h->gate_host;
    out->action_ring = g.act_ring; out->action_stride_bytes = (int64_t)g.act_stride; out->ring_len = (int32_t)g.ring_len;
    out->act_flag = g.act_flag; out->done_flag = g.done_flag; out->groups = (int32_t)g.groups;
    out->wg_per_group = (int32_t)g.wg_per_group; out->workgroups = (int32_t)g.blocks;
    out->envs_per_workgroup = h->epb; out->steps_launched = (int64_t)h->gate_step_seq; out->steps_fed = (int64_t)h->gate_prod_seq;
    return QS_OK;
}

int qs_step_gated(qs_handle *h, int32_t k, void *stream) {
    if (!h || k < 1) return fail(QS_ERR_INVALID, "bad argument");
    if (!h->d_gate) return fail(QS_ERR_INVALID, "no gate: call qs_gate_create first");
    if (h->profiling || h->replay_on || h->d_tape || h->pf.xchg) return fail(QS_ERR_UNSUPPORTED,
        "qs_step_gated: not available with per-launch profiling, the replay wrapper, a noise tape or the fused exchange");
    HIP_TRY(hipSetDevice(h->device));
    // stream-ordered behind everything on the caller's stream, and the caller's stream behind the launch - but the kernel itself sits in
    // the library's high-priority queue (see qs_handle::gate_stream)
    HIP_TRY(hipEventRecord(h->gate_ev_in, (hipStream_t)stream));
    HIP_TRY(hipStreamWaitEvent(h->gate_stream, h->gate_ev_in, 0));
    // the kernel's action-pointer argument carries the sequence base of this launch (qs_step_team.inc)
    int rc = launch_step(h, (const void *)(uintptr_t)h->gate_step_seq, h->gate_stream, k, true);
    if (rc != QS_OK) return rc;
    h->gate_step_seq += (unsigned long long)k;
    HIP_TRY(hipEventRecord(h->gate_ev_out, h->gate_stream));
    h->gate_pending = true;
    // NOT waited for on `stream` here: a wait packet in the caller's hardware queue would hold back whatever shares that queue - possibly
    // the producer this launch is waiting for.  qs_gate_wait orders a stream behind the launch when the caller asks for it.
    return QS_OK;
}

int qs_gate_wait(qs_handle *h, void *stream) {
    if (!h) return fail(QS_ERR_INVALID, "null handle");
    if (!h->d_gate) return fail(QS_ERR_INVALID, "no gate: call qs_gate_create first");
    HIP_TRY(hipSetDevice(h->device));
    if (h->gate_step_seq > 0) HIP_TRY(hipStreamWaitEvent((hipStream_t)stream, h->gate_ev_out, 0));
    return QS_OK;
}

int qs_gate_produce(qs_handle *h, const void *src_actions_dev, int32_t n_src, int32_t k, int32_t closed_loop, void *stream) {
    if (!h || !src_actions_dev || n_src < 1 || k < 1) return fail(QS_ERR_INVALID, "bad argument");
    if (!h->d_gate) return fail(QS_ERR_INVALID, "no gate: call qs_gate_create first");
    if (((uintptr_t)src_actions_dev) & 15) return fail(QS_ERR_INVALID, "qs_gate_produce: the action table must be 16-byte aligned");
    HIP_TRY(hipSetDevice(h->device));
    const unsigned long long T = (unsigned long long)h->cfg.num_envs * h->cfg.num_agents;
    const unsigned long long wg_bytes = (unsigned long long)h->epb * h->cfg.num_agents * 4 * h->real_size,
        batch_bytes = T * 4 * h->real_size;
    hipLaunchKernelGGL(qs_gate_producer_kernel, dim3(h->gate_host.groups), dim3(256), 0, (hipStream_t)stream, h->d_gate,
        (const char *)src_actions_dev, (unsigned int)n_src,
                       h->gate_prod_seq, (int)k, (int)(closed_loop != 0), wg_bytes, batch_bytes, (unsigned long long *)nullptr,
                           (const unsigned int *)nullptr,
                       (const unsigned int *)nullptr, 0ull, 0ull, 0ull, 0ull);
    HIP_TRY(hipGetLastError());
    h->gate_prod_seq += (unsigned long long)k;
    return QS_OK;
}

int qs_gate_produce_verify(qs_handle *h, const void *src_actions_dev, int32_t n_src, int32_t k, unsigned long long *sums_dev,
    void *stream) {
    if (!h || !src_actions_dev || !sums_dev || n_src < 1 || k < 1) return fail(QS_ERR_INVALID, "bad argument");
    if (!h->d_gate) return fail(QS_ERR_INVALID, "no gate: call qs_gate_create first");
    if (((uintptr_t)src_actions_dev) & 15) return fail(QS_ERR_INVALID, "qs_gate_produce_verify: the action table must be 16-byte aligned");
    HIP_TRY(hipSetDevice(h->device));
    const unsigned long long T = (unsigned long long)h->cfg.num_envs * h->cfg.num_agents,
        rows_wg = (unsigned long long)h->epb * h->cfg.num_agents, wpr = h->real_size / 4;
    const unsigned long long wg_bytes = rows_wg * 4 * h->real_size, batch_bytes = T * 4 * h->real_size;
    hipLaunchKernelGGL(qs_gate_producer_kernel, dim3(h->gate_host.groups), dim3(256), 0, (hipStream_t)stream, h->d_gate,
        (const char *)src_actions_dev, (unsigned int)n_src,
                       h->gate_prod_seq, (int)k, 1, wg_bytes, batch_bytes, sums_dev, (const unsigned int *)h->pf.obs,
                           (const unsigned int *)h->pf.reward,
                       rows_wg * h->obs_dim * wpr, rows_wg * wpr, T * h->obs_dim * wpr, T * wpr);
    HIP_TRY(hipGetLastError());
    h->gate_prod_seq += (unsigned long long)k;
    return QS_OK;
}

int qs_gate_status(qs_handle *h, int64_t out[4]) {
    if (!h || !out) return fail(QS_ERR_INVALID, "null argument");
    if (!h->d_gate) return fail(QS_ERR_INVALID, "no gate: call qs_gate_create first");
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipDeviceSynchronize());
    qsx::Gate g;
    HIP_TRY(hipMemcpy(&g, h->d_gate, sizeof g, hipMemcpyDeviceToHost));
    std::vector<unsigned long long> a(g.groups), d(g.blocks);
    HIP_TRY(hipMemcpy(a.data(), g.act_flag, a.size() * 8, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(d.data(), g.done_flag, d.size() * 8, hipMemcpyDeviceToHost));
    unsigned long long amin = ~0ull, dmin = ~0ull;
    for (auto v : a) amin = v < amin ? v : amin;
    for (auto v : d) dmin = v < dmin ? v : dmin;
    out[0] = g.status; out[1] = (int64_t)h->gate_step_seq; out[2] = (int64_t)amin; out[3] = (int64_t)dmin;
    return QS_OK;
}

int qs_sync(qs_handle *h, void *stream) {
    if (!h) return fail(QS_ERR_INVALID, "null handle");
    HIP_TRY(hipSetDevice(h->device));
    if (int jr = gate_join_host(h)) return jr;
    HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    return QS_OK;
}

int qs_get_buffers(qs_handle *h, qs_buffers *out) {
    if (!h || !out) return fail(QS_ERR_INVALID, "null argument");
    *out = h->bufs;
    return QS_OK;
}

int qs_set_obs_target(qs_handle *h, void *obs_dev) {
    if (!h) return fail(QS_ERR_INVALID, "null handle");
    if (obs_dev && h->replay_on) return fail(QS_ERR_UNSUPPORTED,
        "qs_set_obs_target: the device-side replay wrapper restores observations into qs_buffers.obs");
    if (obs_dev && h->d_tape) return fail(QS_ERR_UNSUPPORTED, "qs_set_obs_target: not available while a noise tape is set");
    h->obs_target = obs_dev;
    return QS_OK;
}

extern "C" void *qs_xchg_fused_desc(struct qs_xchg *x, int32_t blocks, int32_t auto_ack, int64_t *n_out);
extern "C" const char *qs_xchg_last_error(void);
extern "C" int qs_xchg_row_layout_is(struct qs_xchg *x, int32_t cols, int32_t q0, int32_t q1);
int qs_set_obs_exchange(qs_handle *h, struct qs_xchg *xchg, int32_t auto_ack) {
    if (!h) return fail(QS_ERR_INVALID, "null handle");
    if (!xchg) { h->pf.xchg = nullptr; return QS_OK; }
    if (!h->team || h->real_size != 4) return fail(QS_ERR_UNSUPPORTED,
        "qs_set_obs_exchange: the fused exchange lives in the float32 team kernels (small batches); use qs_xchg_push for this handle");
    if (h->d_tape) return fail(QS_ERR_UNSUPPORTED, "qs_set_obs_exchange: not available while a noise tape is set");
    HIP_TRY(hipSetDevice(h->device));
    int64_t n = 0;
    void *desc = qs_xchg_fused_desc(xchg, h->blocks, auto_ack, &n);
    if (!desc) return fail(QS_ERR_INVALID, std::string("qs_set_obs_exchange: ") + qs_xchg_last_error());
    if (n != (int64_t)h->cfg.num_envs * h->cfg.num_agents * h->obs_dim) return fail(QS_ERR_INVALID,
        "qs_set_obs_exchange: the endpoint's rows * cols must be E*N * obs_dim");
    {   // the kernels rebuild a QS_WIRE_Q8 layout from their own constants: the neighbour block behind the self observation
        const int self_dim = h->cfg.obs_repr == 0 ? 18 : (h->cfg.obs_repr == 1 ? 19 : 24);
        if (!qs_xchg_row_layout_is(xchg, h->obs_dim, self_dim, self_dim + 6 * h->cfg.num_neighbors))
            return fail(QS_ERR_INVALID, "qs_set_obs_exchange: the endpoint's row layout is not this configuration's (QS_WIRE_Q8: q0 = self columns, q1 = q0 + 6 * visible neighbours)");
    }
    h->pf.xchg = (const qsx::XchgDev *)desc;
    return QS_OK;
}

int qs_set_reward_coeffs(qs_handle *h, const double *coeffs) {
    if (!h || !coeffs) return fail(QS_ERR_INVALID, "null argument");
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipDeviceSynchronize());
    for (int q = 0; q < QS_REW_COUNT; ++q) h->cfg.rew_coeff[q] = coeffs[q];
    fill_consts<float>(h->cfg, h->kf);
    fill_consts<double>(h->cfg, h->kd);
    // the kernels read the coefficients from this buffer on every launch - also launches replayed from a captured HIP graph
    if (h->real_size == 8) {
        double host[QS_REW_COUNT + 1];
        for (int q = 0; q < QS_REW_COUNT; ++q) host[q] = h->kd.rew_coeff[q];
        host[QS_REW_COUNT] = h->kd.prox_ratio;
        HIP_TRY(hipMemcpy((void *)h->pf.rew_rt, host, sizeof host, hipMemcpyHostToDevice));
    } else {
        float host[QS_REW_COUNT + 1];
        for (int q = 0; q < QS_REW_COUNT; ++q) host[q] = h->kf.rew_coeff[q];
        host[QS_REW_COUNT] = h->kf.prox_ratio;
        HIP_TRY(hipMemcpy((void *)h->pf.rew_rt, host, sizeof host, hipMemcpyHostToDevice));
    }
    return QS_OK;
}

static int state_io(qs_handle *h, int32_t env, double *host, int32_t *tick, int set) {
    if (!h || !host) return fail(QS_ERR_INVALID, "null argument");
    if (env < 0 || env >= h->cfg.num_envs) return fail(QS_ERR_INVALID, "env out of range");
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipDeviceSynchronize());
    const int N = h->cfg.num_agents;
    const size_t bytes = sizeof(double) * N * QS_STATE_STRIDE;
    int32_t t = tick ? *tick : -1;
    if (set) {
        HIP_TRY(hipMemcpy(h->d_state_buf, host, bytes, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(h->d_tick_io, &t, sizeof t, hipMemcpyHostToDevice));
    }
    if (h->real_size == 8) { Ptrs<double> p; memcpy(&p, &h->pf, sizeof p);
        hipLaunchKernelGGL(qs_state_kernel<double>, dim3(1), dim3(QS_WAVE), 0, 0, p, h->cfg.num_envs, N, env, h->d_state_buf,
            h->d_tick_io, set); }
    else hipLaunchKernelGGL(qs_state_kernel<float>, dim3(1), dim3(QS_WAVE), 0, 0, h->pf, h->cfg.num_envs, N, env, h->d_state_buf,
        h->d_tick_io, set);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipDeviceSynchronize());
    if (!set) {
        HIP_TRY(hipMemcpy(host, h->d_state_buf, bytes, hipMemcpyDeviceToHost));
        HIP_TRY(hipMemcpy(&t, h->d_tick_io, sizeof t, hipMemcpyDeviceToHost));
        if (tick) *tick = t;
    }
    return QS_OK;
}

int qs_get_state(qs_handle *h, int32_t env, double *state_host, int32_t *tick) { return state_io(h, env, state_host, tick, 0); }
int qs_set_state(qs_handle *h, int32_t env, const double *state_host, int32_t tick) { int32_t t = tick;
    return state_io(h, env, (double *)state_host, &t, 1); }

int qs_memcpy_d2h(qs_handle *h, void *host_dst, const void *dev_src, size_t bytes) {
    if (!h || !host_dst || !dev_src) return fail(QS_ERR_INVALID, "null argument");
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(host_dst, dev_src, bytes, hipMemcpyDeviceToHost));
    return QS_OK;
}

int qs_memcpy_h2d(qs_handle *h, void *dev_dst, const void *host_src, size_t bytes) {
    if (!h || !dev_dst || !host_src) return fail(QS_ERR_INVALID, "null argument");
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(dev_dst, host_src, bytes, hipMemcpyHostToDevice));
    return QS_OK;
}

int qs_state_array_copy(qs_handle *h, void *host, void *dev_array, int32_t elem, int32_t comps, int32_t to_device) {
    if (!h || !host || !dev_array || elem < 1 || comps < 1) return fail(QS_ERR_INVALID, "bad argument");
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipDeviceSynchronize());
    const size_t E = h->cfg.num_envs, N = h->cfg.num_agents, T = E * N, epb = h->bufs.envs_per_block, pitch = h->bufs.state_block_bytes;
    if (!h->bufs.state_lane_major) {   // rows of 64 elements per component: one strided copy per component
        const size_t full = E / epb, rem = E - full * epb;   // whole blocks, environments of the last partial one
        for (int32_t c = 0; c < comps; ++c) {
            char *dev = (char *)dev_array + (size_t)c * 64 * elem, *hst = (char *)host + (size_t)c * T * elem;
            const size_t width = epb * N * elem;
            if (full) {
                if (to_device) HIP_TRY(hipMemcpy2D(dev, pitch, hst, width, width, full, hipMemcpyHostToDevice));
                else HIP_TRY(hipMemcpy2D(hst, width, dev, pitch, width, full, hipMemcpyDeviceToHost));
            }
            if (rem) {
                if (to_device) HIP_TRY(hipMemcpy(dev + full * pitch, hst + full * width, rem * N * elem, hipMemcpyHostToDevice));
                else HIP_TRY(hipMemcpy(hst + full * width, dev + full * pitch, rem * N * elem, hipMemcpyDeviceToHost));
            }
        }
        return QS_OK;
    }
    // lane-major: per block 64 lanes x comps adjacent components (lane = local env * N + drone); host: [comps][E * N].  Through a staging
    // copy of the blocks' pieces of this array (a debugging / test path: one strided copy and a transposition on the host)
    const size_t nblk = (E + epb - 1) / epb, lanes = epb * N, width = lanes * comps * elem;
    std::vector<char> stage(nblk * width);
    if (!to_device || E % epb)   // (a partial last block: keep what its idle lanes hold)
        HIP_TRY(hipMemcpy2D(stage.data(), width, dev_array, pitch, width, nblk, hipMemcpyDeviceToHost));
    for (size_t b = 0; b < nblk; ++b)
        for (size_t l = 0; l < lanes && b * lanes + l < T; ++l)
            for (int32_t c = 0; c < comps; ++c) {
                char *st = stage.data() + b * width + (l * comps + c) * elem, *hs = (char *)host + ((size_t)c * T + b * lanes + l) * elem;
                if (to_device) memcpy(st, hs, elem); else memcpy(hs, st, elem);
            }
    if (to_device) HIP_TRY(hipMemcpy2D(dev_array, pitch, stage.data(), width, width, nblk, hipMemcpyHostToDevice));
    return QS_OK;
}

/* debug: phase time stamps (shader clock) of workgroup 0; all zero unless built with -DQS_TIMING */
// ---- environment snapshots: device-side deep copies of single environments (replay wrapper) ----
int qs_snapshot_pool(qs_handle *h, int32_t slots) {
    if (!h || slots < 0) return fail(QS_ERR_INVALID, "bad argument");
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipDeviceSynchronize());
    if (h->snap_pool) { (void)hipFree(h->snap_pool); h->snap_pool = nullptr; h->snap_slots = 0; }
    if (slots == 0) return QS_OK;
    HIP_TRY(hipMalloc((void **)&h->snap_pool, h->snap_bytes * (size_t)slots));
    h->snap_slots = slots;
    return QS_OK;
}

static int snapshot_io(qs_handle *h, int32_t env, int32_t slot, bool save, hipStream_t s) {
    if (!h) return fail(QS_ERR_INVALID, "null handle");
    if (env < 0 || env >= h->cfg.num_envs) return fail(QS_ERR_INVALID, "env out of range");
    if (slot < 0 || slot >= h->snap_slots) return fail(QS_ERR_INVALID, "snapshot slot out of range (qs_snapshot_pool first)");
    HIP_TRY(hipSetDevice(h->device));
    if (h->gate_pending) { if (int jr = gate_join_stream(h, s)) return jr; }
    char *dst = h->snap_pool + h->snap_bytes * (size_t)slot;
    for (const auto &a : h->snap_arrays) {
        const size_t gb = a.group ? (size_t)env / a.group : 0, ge = a.group ? (size_t)env - gb * a.group : (size_t)env;
        char *src = a.base + gb * a.group_stride + a.elem * a.per_env * ge;
        const size_t width = a.elem * a.per_env, spitch = a.elem * a.comp_stride;
        if (save) HIP_TRY(hipMemcpy2DAsync(dst, width, src, spitch, width, a.comps, hipMemcpyDeviceToDevice, s));
        else HIP_TRY(hipMemcpy2DAsync(src, spitch, dst, width, width, a.comps, hipMemcpyDeviceToDevice, s));
        dst += (a.elem * a.comps * a.per_env + 15) & ~(size_t)15;
    }
    return QS_OK;
}
int qs_snapshot_save(qs_handle *h, int32_t env, int32_t slot, void *stream) { return snapshot_io(h, env, slot, true, (hipStream_t)stream); }
int qs_snapshot_load(qs_handle *h, int32_t slot, int32_t env,
    void *stream) { return snapshot_io(h, env, slot, false, (hipStream_t)stream); }
int qs_snapshot_copy(qs_handle *h, int32_t src_slot, int32_t dst_slot, void *stream) {
    if (!h || src_slot < 0 || dst_slot < 0 || src_slot >= h->snap_slots || dst_slot >= h->snap_slots) return fail(QS_ERR_INVALID,
        "snapshot slot out of range");
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipMemcpyAsync(h->snap_pool + h->snap_bytes * (size_t)dst_slot, h->snap_pool + h->snap_bytes * (size_t)src_slot, h->snap_bytes,
                           hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return QS_OK;
}

/* Batched experience replay on the device: see include/quadswarm.h. */
int qs_replay_enable(qs_handle *h, double sample_prob) {
    if (!h) return fail(QS_ERR_INVALID, "null handle");
    if (h->replay_on) return fail(QS_ERR_INVALID, "replay is already enabled on this handle");
    if (h->obs_target) return fail(QS_ERR_UNSUPPORTED,
        "qs_replay_enable: the replay wrapper restores observations into qs_buffers.obs (reset qs_set_obs_target first)");
    if (!h->cfg.episode_sums) return fail(QS_ERR_INVALID,
        "qs_replay_enable needs a handle created with episode_sums = 1 (per-episode crash reward)");
    if (!(sample_prob >= 0.0 && sample_prob <= 1.0)) return fail(QS_ERR_INVALID, "sample_prob must be in [0, 1]");
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipDeviceSynchronize());
    ReplayParams &P = h->rp;
    memset(&P, 0, sizeof P);
    const size_t E = h->cfg.num_envs;
    uint32_t off = 0;
    for (const auto &a : h->snap_arrays) {
        if (a.kind == 2) continue;   // the reward-shaping wrapper sits outside the replay wrapper: its sums restart with the episode
        if (P.narr == QS_REPLAY_MAX_ARR) return fail(QS_ERR_UNSUPPORTED, "too many snapshot arrays");
        if (a.kind == 1) P.obs_arr = P.narr;
        if (a.base == (char *)h->pf.tick) P.tick_arr = P.narr;
        P.arr[P.narr++] = {a.base, (uint32_t)a.elem, (uint32_t)a.comps, (uint32_t)a.per_env, off, (uint64_t)a.comp_stride,
            (uint32_t)a.group, (uint32_t)a.group_stride};
        off += (uint32_t)((a.elem * a.comps * a.per_env + 15) & ~(size_t)15);
    }
    P.snap_bytes = off;
    const double control_freq = 1.0 / (h->cfg.dt * h->cfg.sim_steps);
    P.N = h->cfg.num_agents; P.E = h->cfg.num_envs; P.use_obstacles = h->cfg.use_obstacles;
    P.ep_len = h->cfg.ep_len;
    P.cp_every = (int)(0.5 * control_freq + 0.5);        // cp_step_size_freq (:18-19)
    P.grace_ticks = (int)(1.5 * control_freq + 0.5);     // collisions_grace_period_seconds * control_freq (:150)
    P.min_gap = (int)(5.0 * control_freq + 0.5);         // :152
    P.seed_lo = (uint32_t)(h->cfg.seed & 0xffffffffu); P.seed_hi = (uint32_t)(h->cfg.seed >> 32);
    P.env_id_offset = h->cfg.env_id_offset;
    P.sample_prob = (float)sample_prob;
    P.done = h->pf.done; P.tick = h->pf.tick; P.step_ctr = h->pf.step_ctr; P.unique_col = h->pf.unique_col; P.obst_new = h->pf.obst_new;
    P.counters = h->pf.counters; P.ep_sums = h->pf.ep_sums; P.run_sums = h->pf.run_sums; P.real_size = h->real_size;
    P.T = (int32_t)(E * h->cfg.num_agents);
    int rc;
    if ((rc = dalloc(h, &P.pool, (size_t)P.snap_bytes * (QS_REPLAY_RING + QS_REPLAY_EVENTS) * E)) != QS_OK) return rc;
    if ((rc = dalloc(h, &P.active, E)) != QS_OK || (rc = dalloc(h, &P.saved, E)) != QS_OK || (rc = dalloc(h, &P.ep_saved, E)) != QS_OK
        || (rc = dalloc(h, &P.crash_hist, 100 * E)) != QS_OK ||
        (rc = dalloc(h, &P.crash_n, E)) != QS_OK || (rc = dalloc(h, &P.crash_pos, E)) != QS_OK
            || (rc = dalloc(h, &P.ck_count, E)) != QS_OK ||
        (rc = dalloc(h, &P.ck_head, E)) != QS_OK || (rc = dalloc(h, &P.last_added, E)) != QS_OK
            || (rc = dalloc(h, &P.ev_len, E)) != QS_OK ||
        (rc = dalloc(h, &P.ev_idx, E)) != QS_OK || (rc = dalloc(h, &P.ev_replayed, QS_REPLAY_EVENTS * E)) != QS_OK ||
        (rc = dalloc(h, &P.ev_slot, QS_REPLAY_EVENTS * E)) != QS_OK || (rc = dalloc(h, &P.episodes, E)) != QS_OK ||
        (rc = dalloc(h, &P.replayed, E)) != QS_OK || (rc = dalloc(h, &P.errors, E)) != QS_OK
            || (rc = dalloc(h, &P.start_tick, E)) != QS_OK ||
        (rc = dalloc(h, &P.last_steps, E)) != QS_OK) return rc;
    {   // the reset() that starts the first episode records crashes_last_episode = 0 (quadrotor_multi.py:356-359); last_added = -1e9
        std::vector<int32_t> ones(E, 1), neg(E, -1000000000);
        HIP_TRY(hipMemcpy(P.crash_n, ones.data(), E * sizeof(int32_t), hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(P.crash_pos, ones.data(), E * sizeof(int32_t), hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(P.last_added, neg.data(), E * sizeof(int32_t), hipMemcpyHostToDevice));
    }
    h->replay_on = true;
    return QS_OK;
}

int qs_replay_stats(qs_handle *h, int32_t *out) {
    if (!h || !out) return fail(QS_ERR_INVALID, "null argument");
    if (!h->replay_on) return fail(QS_ERR_INVALID, "replay is not enabled");
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipDeviceSynchronize());
    const size_t E = h->cfg.num_envs;
    const ReplayParams &P = h->rp;
    std::vector<int32_t> len(E), rep(QS_REPLAY_EVENTS * E);
    std::vector<uint8_t> act(E), eps(E);
    HIP_TRY(hipMemcpy(eps.data(), P.ep_saved, E, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(out + 0 * E, P.episodes, E * 4, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(out + 1 * E, P.replayed, E * 4, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(len.data(), P.ev_len, E * 4, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(rep.data(), P.ev_replayed, QS_REPLAY_EVENTS * E * 4, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(act.data(), P.active, E, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(out + 5 * E, P.ck_count, E * 4, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(out + 6 * E, P.errors, E * 4, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(out + 8 * E, P.last_steps, E * 4, hipMemcpyDeviceToHost));
    for (size_t e = 0; e < E; ++e) {
        int32_t sum = 0;
        for (int q = 0; q < len[e]; ++q) sum += rep[(size_t)q * E + e];
        out[2 * E + e] = len[e]; out[3 * E + e] = sum; out[4 * E + e] = act[e]; out[7 * E + e] = eps[e];
    }
    return QS_OK;
}

int qs_replay_set_active(qs_handle *h, const uint8_t *active_host) {
    if (!h) return fail(QS_ERR_INVALID, "null handle");
    if (!h->replay_on) return fail(QS_ERR_INVALID, "replay is not enabled");
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipDeviceSynchronize());
    const size_t E = h->cfg.num_envs;
    std::vector<uint8_t> v(E, 1);
    if (active_host) for (size_t e = 0; e < E; ++e) v[e] = active_host[e] ? 1 : 0;
    HIP_TRY(hipMemcpy(h->rp.active, v.data(), E, hipMemcpyHostToDevice));
    return QS_OK;
}

/* Noise tape (test instrument): see include/quadswarm.h. */
int qs_set_noise_tape(qs_handle *h, const double *tape_host, int64_t len_per_env) {
    if (!h) return fail(QS_ERR_INVALID, "null handle");
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipDeviceSynchronize());
    if (h->d_tape) { (void)hipFree(h->d_tape); h->d_tape = nullptr; }
    if (h->d_tape_pos) { (void)hipFree(h->d_tape_pos); h->d_tape_pos = nullptr; }
    h->tape_len = 0;
    h->pf.tape = nullptr; h->pf.tape_pos = nullptr; h->pf.tape_len = 0;
    if (!tape_host || len_per_env <= 0) return QS_OK;   // back to the counter-based stream
    if (len_per_env > 0x7fffff00ll) return fail(QS_ERR_INVALID, "tape too long");
    if (qs_tape_lds_bytes(&h->cfg, h->obs_dim, h->full ? 1 : 0, h->real_size) > 160 * 1024) return fail(QS_ERR_UNSUPPORTED,
        "noise tape: the single-wave layout does not fit the LDS");
    const size_t E = h->cfg.num_envs, bytes = E * (size_t)len_per_env * sizeof(double);
    HIP_TRY(hipMalloc((void **)&h->d_tape, bytes));
    HIP_TRY(hipMalloc((void **)&h->d_tape_pos, E * sizeof(int32_t)));
    HIP_TRY(hipMemcpy(h->d_tape, tape_host, bytes, hipMemcpyHostToDevice));
    HIP_TRY(hipMemset(h->d_tape_pos, 0, E * sizeof(int32_t)));
    h->tape_len = len_per_env;
    h->pf.tape = h->d_tape; h->pf.tape_pos = h->d_tape_pos; h->pf.tape_len = len_per_env;
    return QS_OK;
}

int qs_set_tape_pos(qs_handle *h, const int32_t *pos_host) {
    if (!h || !pos_host) return fail(QS_ERR_INVALID, "null argument");
    if (!h->d_tape) return fail(QS_ERR_INVALID, "no noise tape set");
    for (int e = 0; e < h->cfg.num_envs; ++e) if (pos_host[e] < 0 || pos_host[e] > h->tape_len) return fail(QS_ERR_INVALID,
        "tape position out of range");
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(h->d_tape_pos, pos_host, (size_t)h->cfg.num_envs * sizeof(int32_t), hipMemcpyHostToDevice));
    return QS_OK;
}

int qs_get_tape_pos(qs_handle *h, int32_t *pos_host) {
    if (!h || !pos_host) return fail(QS_ERR_INVALID, "null argument");
    if (!h->d_tape) return fail(QS_ERR_INVALID, "no noise tape set");
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(pos_host, h->d_tape_pos, (size_t)h->cfg.num_envs * sizeof(int32_t), hipMemcpyDeviceToHost));
    return QS_OK;
}

/* debug / tools: dynamic LDS bytes per workgroup of the layout qs_create would use (team: waves per workgroup, 0 = single-wave;
 * spec: 1 = config-specialised kernels) */
int qs_debug_lds_bytes(const qs_config *cfg, int team, int spec) {
    if (!cfg) return -1;
    const int rs = cfg->precision == QS_PRECISION_F64 ? 8 : 4;
    return lds_layout(rs, QS_WAVE, cfg->num_agents, QS_WAVE / cfg->num_agents, qs_obs_dim(cfg), cfg->num_obstacles, cfg->num_neighbors,
        team,
                      scenario_is_full(cfg->scenario), cfg->scenario, spec ? spec_rows_per_pass(cfg, team) : QS_WAVE).total;
}

// [blocks][16]: start, end (s_memtime), HW_ID, XCC_ID, start, end (100 MHz wall clock), then s_memtime at 10 phase boundaries of wave 0 of
// every workgroup
int qs_debug_wg_times(qs_handle *h, unsigned long long *out, int32_t max_blocks) {
    if (!h || !out) return fail(QS_ERR_INVALID, "null argument");
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipDeviceSynchronize());
    const int n = h->blocks < max_blocks ? h->blocks : max_blocks;
    HIP_TRY(hipMemcpy(out, h->pf.timing + 128, (size_t)n * 16 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    return n;
}
int qs_debug_timing(qs_handle *h, unsigned long long *out128) {   // [4 waves][32 stamps] of workgroup 0 (QS_TIMING builds)
    if (!h || !out128) return fail(QS_ERR_INVALID, "null argument");
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(out128, h->pf.timing, 128 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    return QS_OK;
}

int qs_check_errors(qs_handle *h) {
    if (!h) return fail(QS_ERR_INVALID, "null handle");
    HIP_TRY(hipSetDevice(h->device));
    // the gated kernel's stream is non-blocking: a null-stream copy does not wait for it by itself
    if (int jr = gate_join_host(h)) return jr;
    uint32_t f = 0;
    HIP_TRY(hipMemcpy(&f, h->pf.error_flag, sizeof f, hipMemcpyDeviceToHost));
    if (f) return fail(QS_ERR_NAN_REWARD, "QuadEnv: reward is Nan");
    return QS_OK;
}

int qs_set_profiling(qs_handle *h, int32_t enable) {
    if (!h) return fail(QS_ERR_INVALID, "null handle");
    h->profiling = enable != 0;
    h->events_used = 0;
    return QS_OK;
}

int qs_get_kernel_time(qs_handle *h, double *avg_ms, int64_t *launches) {
    if (!h || !avg_ms || !launches) return fail(QS_ERR_INVALID, "null argument");
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipDeviceSynchronize());
    double total = 0;
    for (size_t k = 0; k < h->events_used; ++k) {
        float ms = 0;
        HIP_TRY(hipEventElapsedTime(&ms, h->events[k].first, h->events[k].second));
        total += ms;
    }
    *launches = (int64_t)h->events_used;
    *avg_ms = h->events_used ? total / (double)h->events_used : 0.0;
    h->events_used = 0;
    return QS_OK;
}

}  // extern "C"
