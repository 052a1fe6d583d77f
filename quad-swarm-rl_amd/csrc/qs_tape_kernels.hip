// qs_tape_kernels.hip - the "noise tape" flavour of the stepper behind qs_set_noise_tape (include/quadswarm.h).  TEST INSTRUMENT.
//
// Same device code as quadswarm_hip.hip (qs_device.h, qs_scenarios.h, qs_kernels.h, qs_step_kernel.inc) compiled with QS_TAPE:
// every random draw pops a sequential tape of the reference's own draws (values in final units, in the reference's call order,
// SURVEY.md Appendix B) instead of running Philox, and the lanes of an environment take turns wherever the reference's draw
// order is data-dependent (per-drone step, collision responses, reset).  This is how the golden fixtures captured from the
// reference (tests/golden/*.npz: actions, tape, outputs) are replayed straight through the HIP arithmetic:
// tests/test_hip_vs_reference.py.  Generic single-wave kernels, one launch per control step; float64 (the reference's arithmetic:
// free-running replay of whole fixtures to 1e-9) and float32 (the production precision: the same fixtures teacher-forced from the
// reference's recorded states, one step at a time, to north_star's 1e-5 - tests/test_hip_vs_reference_f32.py).  Nothing of this is on
// the production path, which never sees QS_TAPE.
#define QS_TAPE 1
#include "qs_kernels.h"

static LdsLayout tape_layout(const qs_config *cfg, int obs_dim, bool full, int real_size) {
    return lds_layout(real_size, QS_WAVE, cfg->num_agents, QS_WAVE / cfg->num_agents, obs_dim, cfg->num_obstacles, cfg->num_neighbors, 0,
        full, cfg->scenario, QS_WAVE);
}

extern "C" int qs_tape_lds_bytes(const qs_config *cfg, int obs_dim, int full,
    int real_size) { return tape_layout(cfg, obs_dim, full != 0, real_size).total; }

template <typename real>
static int tape_launch(int which, const qs_config *cfg, int obs_dim, int full, const void *consts, const void *ptrs, const void *actions,
    void *stream) {
    const LdsLayout L = tape_layout(cfg, obs_dim, full != 0, (int)sizeof(real));
    const int epb = QS_WAVE / cfg->num_agents, blocks = (cfg->num_envs + epb - 1) / epb;
    Consts<real> c;
    Ptrs<real> p;
    memcpy(&c, consts, sizeof c);
    memcpy(&p, ptrs, sizeof p);
    hipStream_t s = (hipStream_t)stream;
    if (L.total > 64 * 1024) {
        const void *fns[] = {(const void *)qs_tape_step_kernel<real>, (const void *)qs_tape_step_kernel_full<real>,
                             (const void *)qs_tape_reset_kernel<real, false>, (const void *)qs_tape_reset_kernel<real, true>};
        for (const void *fn : fns) {
            hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, L.total);
            if (e != hipSuccess) return (int)e;
        }
    }
    if (which == 0) {
        if (full) hipLaunchKernelGGL((qs_tape_reset_kernel<real, true>), dim3(blocks), dim3(QS_WAVE), L.total, s, c, p, L, epb);
        else hipLaunchKernelGGL((qs_tape_reset_kernel<real, false>), dim3(blocks), dim3(QS_WAVE), L.total, s, c, p, L, epb);
    } else {
        if (full) hipLaunchKernelGGL(qs_tape_step_kernel_full<real>, dim3(blocks), dim3(QS_WAVE), L.total, s, c, p,
            (const real *)actions, L, epb);
        else hipLaunchKernelGGL(qs_tape_step_kernel<real>, dim3(blocks), dim3(QS_WAVE), L.total, s, c, p, (const real *)actions, L, epb);
    }
    return (int)hipGetLastError();
}

// which: 0 = reset kernel, 1 = step kernel; consts = Consts<float> / Consts<double> by real_size (4 / 8).  Returns a hipError_t.
extern "C" int qs_tape_launch(int which, const qs_config *cfg, int obs_dim, int full, int real_size, const void *consts,
    const void *ptrs, const void *actions, void *stream) {
    return real_size == 8 ? tape_launch<double>(which, cfg, obs_dim, full, consts, ptrs, actions, stream)
                          : tape_launch<float>(which, cfg, obs_dim, full, consts, ptrs, actions, stream);
}
