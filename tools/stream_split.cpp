// Experiment: the C2 batch (8 drones x E envs) stepped as S independent sub-batches on S HIP streams, launches issued from one
// host thread through the C ABI (no Python in the loop).  Prints whole-batch us per control step for S = 1, 2, 4.
// build: hipcc -O2 -Iinclude tools/stream_split.cpp -o tools/stream_split -Lquad-swarm-rl_amd/csrc -lquadswarm_hip -Wl,-rpath,'$ORIGIN/../quad-swarm-rl_amd/csrc'
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "quadswarm.h"

int main(int argc, char **argv) {
    const int E = argc > 1 ? atoi(argv[1]) : 1024, K = argc > 2 ? atoi(argv[2]) : 3000, N = 8;
    for (int S : {1, 2, 4}) {
        const int sub = E / S;
        std::vector<qs_handle *> h(S);
        std::vector<hipStream_t> sm(S);
        std::vector<float *> act(S);
        for (int i = 0; i < S; ++i) {
            qs_config c;
            qs_default_config(&c, sub, N);
            c.env_id_offset = i * sub; c.use_downwash = 1; c.write_rew_info = 0; c.precision = QS_PRECISION_F32;
            if (qs_create(&c, 0, &h[i]) != 0) { printf("create failed: %s\n", qs_last_error()); return 1; }
            (void)hipStreamCreateWithFlags(&sm[i], hipStreamNonBlocking);
            (void)hipMalloc(&act[i], sizeof(float) * sub * N * 4);
            std::vector<float> a(sub * N * 4);
            for (auto &x : a) x = 2.0f * rand() / RAND_MAX - 1.0f;
            (void)hipMemcpy(act[i], a.data(), a.size() * 4, hipMemcpyHostToDevice);
            qs_reset(h[i], nullptr, sm[i]);
        }
        auto run = [&](int k) { for (int t = 0; t < k; ++t) for (int i = 0; i < S; ++i) qs_step(h[i], act[i], sm[i]); };
        run(200);
        (void)hipDeviceSynchronize();
        auto t0 = std::chrono::steady_clock::now();
        run(K);
        (void)hipDeviceSynchronize();
        double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        printf("{\"streams\": %d, \"envs_per_stream\": %d, \"us_per_control_step\": %.3f, \"env_steps_per_s\": %.4g, \"flavor\": %d}\n", S, sub, 1e6 * dt / K,
               (double)E * N * 2 * K / dt, qs_kernel_flavor(h[0]));
        for (int i = 0; i < S; ++i) { qs_destroy(h[i]); (void)hipStreamDestroy(sm[i]); (void)hipFree(act[i]); }
    }
    return 0;
}
