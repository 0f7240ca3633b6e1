// c_abi_demo.cpp -- a caller of include/crl.h with no Python and no torch in sight: what a
// host written in another language would do through its FFI.  Plain HIP runtime calls for the
// device buffers, everything else through the C ABI.
//
//   hipcc --offload-arch=gfx950 -I include examples/c_abi_demo.cpp -L competitive_rl_amd -lcrl_hip \
//         -Wl,-rpath,'$ORIGIN/../competitive_rl_amd' -o examples/c_abi_demo
//   examples/c_abi_demo [num_envs] [steps]
//
// Steps cPongDouble-v0 (raw frames) with a fixed action pattern, prints throughput and
// order-independent checksums of rewards / dones / the last frames; exit code 0 on success.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <chrono>
#include <vector>

#include "crl.h"

#define HIP_OK(x)                                                                   \
    do {                                                                            \
        hipError_t e_ = (x);                                                        \
        if (e_ != hipSuccess) {                                                     \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                 \
            return 2;                                                               \
        }                                                                           \
    } while (0)
#define CRL_OK_(x)                                                                  \
    do {                                                                            \
        int r_ = (x);                                                               \
        if (r_ != CRL_OK) {                                                         \
            fprintf(stderr, "%s -> %d: %s\n", #x, r_, crl_last_error());            \
            return 3;                                                               \
        }                                                                           \
    } while (0)

int main(int argc, char **argv) {
    const int64_t n = argc > 1 ? atoll(argv[1]) : 1024;
    const int steps = argc > 2 ? atoi(argv[2]) : 200;
    // score band: the library takes whatever gray glyphs the host baked; plain white here
    std::vector<uint8_t> atlas(CRL_PONG_ATLAS_BYTES, 255);
    crl_opts o = {};
    o.env_kind = CRL_ENV_PONG_DOUBLE, o.obs_mode = CRL_OBS_RAW_RGB, o.frame_stack = 1;
    o.num_envs = n, o.env_id_base = 0, o.seed = 7, o.device = 0;
    crl_ctx *ctx = nullptr;
    CRL_OK_(crl_create(&o, atlas.data(), &ctx));
    const int64_t obs_bytes = crl_obs_bytes_per_env(ctx) * n;
    uint8_t *obs = nullptr, *done = nullptr;
    float *rew = nullptr;
    int32_t *act = nullptr;
    HIP_OK(hipMalloc(&obs, obs_bytes));
    HIP_OK(hipMalloc(&done, n));
    HIP_OK(hipMalloc(&rew, n * 2 * sizeof(float)));
    HIP_OK(hipMalloc(&act, n * 2 * sizeof(int32_t)));
    std::vector<int32_t> a(n * 2);
    for (int64_t i = 0; i < n * 2; i++) a[i] = (int32_t)((i * 2654435761u >> 7) % 3);
    HIP_OK(hipMemcpy(act, a.data(), a.size() * sizeof(int32_t), hipMemcpyHostToDevice));
    hipStream_t st;
    HIP_OK(hipStreamCreate(&st));
    CRL_OK_(crl_reset(ctx, obs, st));
    std::vector<float> r(n * 2);
    std::vector<uint8_t> d(n);
    double rsum = 0;
    int64_t dsum = 0;
    HIP_OK(hipStreamSynchronize(st));
    const auto t0 = std::chrono::steady_clock::now();
    for (int t = 0; t < steps; t++) CRL_OK_(crl_step(ctx, act, obs, rew, done, st));
    HIP_OK(hipStreamSynchronize(st));
    const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    HIP_OK(hipMemcpy(r.data(), rew, r.size() * sizeof(float), hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(d.data(), done, d.size(), hipMemcpyDeviceToHost));
    for (float v : r) rsum += v;
    for (uint8_t v : d) dsum += v;
    // the raw frame of env 0, agent 0: white band rows, black court, white ball/bats
    std::vector<uint8_t> f0(CRL_PONG_FRAME_BYTES);
    HIP_OK(hipMemcpy(f0.data(), obs, f0.size(), hipMemcpyDeviceToHost));
    int64_t white = 0;
    for (uint8_t v : f0) white += v == 255;
    std::vector<crl_pong_env_state> state(n);
    CRL_OK_(crl_get_state(ctx, state.data(), 0, n, st));
    int64_t frames = 0;
    for (const auto &s : state) frames += s.num_steps;
    printf("%s: %lld envs x %d steps in %.3f s = %.2f M env-steps/s; last-step reward sum %.0f, dones %lld, "
           "white bytes in frame 0: %lld, sum of num_steps %lld\n",
           crl_version(), (long long)n, steps, dt, n * (double)steps / dt / 1e6, rsum, (long long)dsum, (long long)white,
           (long long)frames);
    // zero-sum rewards, a frame that is mostly the white borders + black court
    const bool ok = rsum == 0.0 && white > 3 * CRL_PONG_W * (CRL_PONG_TOP + CRL_PONG_H - CRL_PONG_BOTTOM) && white < CRL_PONG_FRAME_BYTES / 2;
    crl_destroy(ctx);
    (void)hipFree(obs), (void)hipFree(done), (void)hipFree(rew), (void)hipFree(act);
    (void)hipStreamDestroy(st);
    return ok ? 0 : 1;
}
