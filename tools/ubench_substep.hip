// micro-benchmark: cycles of the per-drone sub-step (qs_device.h substep<float>) for a lone wave
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <cmath>
#include "../quad-swarm-rl_amd/csrc/qs_device.h"
using namespace qs;

__global__ void k(Consts<float> c, float *io, unsigned long long *cyc, int iters) {
    Drone<float> d;
    int t = threadIdx.x;
    for (int q = 0; q < 3; ++q) { d.pos[q] = io[q * 64 + t]; d.vel[q] = io[(3 + q) * 64 + t]; d.omega[q] = io[(6 + q) * 64 + t]; }
    for (int q = 0; q < 9; ++q) d.rot[q] = (q % 4 == 0) ? 1.f : 0.f;
    for (int q = 0; q < 4; ++q) { d.rot_damp[q] = 0.5f; d.cmds_damp[q] = 0.25f; d.ou[q] = 0.001f * t; }
    d.flags = 0;
    RngKey key = {1, 2, 3, 4};
    float cmds[4] = {0.4f + 0.001f * t, 0.5f, 0.6f, 0.55f}, acc[3];
    unsigned long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
        substep<float>(c, key, t, 0, d, cmds, acc);
        substep<float>(c, key, t, 1, d, cmds, acc);
    }
    unsigned long long t1 = clock64();
    for (int q = 0; q < 3; ++q) io[q * 64 + t] = d.pos[q] + d.vel[q] + d.omega[q] + d.rot[q] + acc[q];
    if (t == 0) cyc[0] = t1 - t0;
}

int main() {
    Consts<float> c; memset(&c, 0, sizeof c);
    for (int q = 0; q < 3; ++q) { c.inertia[q] = 1.4e-5f; c.inv_inertia[q] = 1.f / 1.4e-5f; c.room_lo[q] = -5; c.room_hi[q] = 5; }
    c.room_lo[2] = 0; c.room_hi[2] = 10; c.arm = 0.046f; c.mass = 0.028f; c.inv_mass = 1 / 0.028f;
    float pc[4][3] = {{-.0325f, -.0325f, 0}, {-.0325f, .0325f, 0}, {.0325f, .0325f, 0}, {.0325f, -.0325f, 0}}, ccw[4] = {-1, 1, -1, 1};
    for (int m = 0; m < 4; ++m) { for (int q = 0; q < 3; ++q) c.prop_cross[m][q] = pc[m][q]; c.prop_ccw[m] = ccw[m]; c.thrust_max[m] = 0.13f; c.torque_max[m] = 7.8e-4f; }
    c.motor_tau_up = c.motor_tau_down = 0.1333f; c.motor_linearity = 1; c.omega_max = 40; c.dt = 0.005f; c.control_dt = 0.01f;
    c.floor_threshold = 0.046f; c.sim_steps = 2; c.svd_period = 100; c.floor_mode = 0;
    float h[9 * 64];
    for (int t = 0; t < 64; ++t) { h[0 * 64 + t] = 0.1f * t - 3; h[64 + t] = 0.05f * t - 1; h[128 + t] = 2.0f + 0.01f * t;
        for (int q = 3; q < 9; ++q) h[q * 64 + t] = 0.01f * (q + t % 7); }
    float *io; unsigned long long *cyc, hc;
    hipMalloc(&io, sizeof h); hipMalloc(&cyc, 8);
    for (int rep = 0; rep < 2; ++rep) {
        hipMemcpy(io, h, sizeof h, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, c, io, cyc, 200);
        hipDeviceSynchronize();
    }
    hipMemcpy(&hc, cyc, 8, hipMemcpyDeviceToHost);
    printf("2 sub-steps: %.1f ticks (memtime) per control step, lone wave\n", (double)hc / 200.0);
    return 0;
}
