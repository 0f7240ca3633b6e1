// c_abi_demo.cpp -- a caller of include/crl.h with no Python and no torch in sight: what a
// host written in another language would do through its FFI.  Plain HIP runtime calls for the
// device buffers, everything else through the C ABI.
//
//   hipcc --offload-arch=gfx950 -I include examples/c_abi_demo.cpp -L competitive_rl_amd -lcrl_hip \
//         -Wl,-rpath,'$ORIGIN/../competitive_rl_amd' -o examples/c_abi_demo
//   examples/c_abi_demo [num_envs] [steps]
//
// Steps cPongDouble-v0 (raw frames) with a fixed action pattern, prints throughput and
// order-independent checksums of rewards / dones / the last frames; then the trainer's float32 frame stack, once drawn by
// the step (crl_step_stack) and once rolled and appended (crl_frame_stack_update_to), compared; exit code 0 on success.
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
    bool ok = rsum == 0.0 && white > 3 * CRL_PONG_W * (CRL_PONG_TOP + CRL_PONG_H - CRL_PONG_BOTTOM) && white < CRL_PONG_FRAME_BYTES / 2;
    crl_destroy(ctx);
    (void)hipFree(obs), (void)hipFree(done), (void)hipFree(rew);

    // ---- the trainer's frame stack, twice (reference utils/utils.py:23-60, 145-173): context A has the step DRAW agent 0's float32
    // (4, 84, 84) stack (crl_step_stack: two buffers, alternating); context B steps and then rolls-and-appends it with the generic kernel
    // (crl_frame_stack_update_to, mask = 1 - done built on the host from flags that crl_set_flags_event hands over ahead of the draw).
    // Same seed, same actions: the two stacks must be the same bytes.
    {
        const int64_t m = n < 512 ? n : 512;
        const int R = 84, K = 4, steps2 = steps < 300 ? steps : 300;
        crl_opts g = {};
        g.env_kind = CRL_ENV_PONG_DOUBLE, g.obs_mode = CRL_OBS_GRAY_RESIZED, g.resized_dim = R, g.frame_stack = 1;
        g.num_envs = m, g.seed = 11, g.device = 0;
        crl_ctx *A = nullptr, *B = nullptr;
        CRL_OK_(crl_create(&g, atlas.data(), &A));
        CRL_OK_(crl_create(&g, atlas.data(), &B));
        const size_t plane = (size_t)R * R, stack_elems = (size_t)m * K * plane;
        uint8_t *obsA, *obsB, *doneA, *doneB, *done_pinned;
        float *rewA, *rewB, *sa[2], *sb[2], *mask;
        HIP_OK(hipMalloc(&obsA, m * 2 * plane));
        HIP_OK(hipMalloc(&obsB, m * 2 * plane));
        HIP_OK(hipMalloc(&doneA, m));
        HIP_OK(hipMalloc(&doneB, m));
        HIP_OK(hipMalloc(&rewA, m * 2 * sizeof(float)));
        HIP_OK(hipMalloc(&rewB, m * 2 * sizeof(float)));
        HIP_OK(hipMalloc(&mask, m * sizeof(float)));
        HIP_OK(hipHostMalloc((void **)&done_pinned, m, hipHostMallocDefault));
        for (int k = 0; k < 2; k++) {
            HIP_OK(hipMalloc(&sa[k], stack_elems * sizeof(float)));
            HIP_OK(hipMalloc(&sb[k], stack_elems * sizeof(float)));
            HIP_OK(hipMemset(sa[k], 0, stack_elems * sizeof(float)));
            HIP_OK(hipMemset(sb[k], 0, stack_elems * sizeof(float)));
        }
        hipStream_t side;
        hipEvent_t flags_ready;
        HIP_OK(hipStreamCreate(&side));
        HIP_OK(hipEventCreateWithFlags(&flags_ready, hipEventDisableTiming));
        CRL_OK_(crl_set_flags_event(B, flags_ready));
        crl_stack_desc sd = {};
        sd.planes = K, sd.dtype = CRL_OBS_F32, sd.agent = 0;
        // frame_stack_tensor.update(envs.reset()[0]): A draws it, B appends it
        CRL_OK_(crl_reset(A, nullptr, st));
        sd.stack_dev = sa[0], sd.valid_planes = 1;
        CRL_OK_(crl_draw_stack(A, obsA, &sd, st));
        CRL_OK_(crl_reset(B, obsB, st));
        CRL_OK_(crl_frame_stack_update_to(sb[0], sb[1], obsB, CRL_OBS_U8, (int64_t)2 * plane, nullptr, m, 1, K, (int64_t)plane, st));
        {   // a third of the envs one round before the end of their episode (21 rounds, pong/register.py:20-22): the erase path runs
            std::vector<crl_pong_env_state> es(m);
            for (crl_ctx *c : {A, B}) {
                CRL_OK_(crl_get_state(c, es.data(), 0, m, st));
                for (int64_t i = 0; i < m; i += 3) es[i].num_rounds = 20;
                CRL_OK_(crl_set_state(c, es.data(), 0, m, st));
            }
        }
        int cur = 0, ends = 0;
        std::vector<float> hm(m);
        for (int t = 0; t < steps2; t++) {
            const int nxt = cur ^ 1;
            sd.stack_dev = sa[nxt], sd.valid_planes = t + 2 < K ? t + 2 : K;
            CRL_OK_(crl_step_stack(A, act, obsA, rewA, doneA, &sd, st));   // envs.step + frame_stack_tensor.update in one launch
            CRL_OK_(crl_step(B, act, obsB, rewB, doneB, st));
            HIP_OK(hipStreamWaitEvent(side, flags_ready, 0));             // the flags, behind the dynamics kernel only
            HIP_OK(hipMemcpyAsync(done_pinned, doneB, m, hipMemcpyDeviceToHost, side));
            HIP_OK(hipStreamSynchronize(side));
            for (int64_t i = 0; i < m; i++) hm[i] = done_pinned[i] ? 0.0f : 1.0f, ends += done_pinned[i];
            HIP_OK(hipMemcpyAsync(mask, hm.data(), m * sizeof(float), hipMemcpyHostToDevice, st));
            CRL_OK_(crl_frame_stack_update_to(sb[nxt], sb[cur], obsB, CRL_OBS_U8, (int64_t)2 * plane, mask, m, 1, K, (int64_t)plane, st));
            HIP_OK(hipStreamSynchronize(st));  // (hm is reused by the next step)
            cur = nxt;
        }
        std::vector<float> ha(stack_elems), hb(stack_elems);
        HIP_OK(hipMemcpy(ha.data(), sa[cur], stack_elems * sizeof(float), hipMemcpyDeviceToHost));
        HIP_OK(hipMemcpy(hb.data(), sb[cur], stack_elems * sizeof(float), hipMemcpyDeviceToHost));
        size_t diff = 0, nonzero = 0;
        for (size_t i = 0; i < stack_elems; i++) diff += ha[i] != hb[i], nonzero += ha[i] != 0.0f;
        printf("frame stack drawn by the step vs rolled and appended: %lld envs x %d steps, %d episode ends, %zu of %zu elements differ (%zu non-zero)\n",
               (long long)m, steps2, ends, diff, stack_elems, nonzero);
        ok = ok && diff == 0 && nonzero > 0 && ends > 0;
        // what the fused draw refuses is an error code with a text, not a crash
        sd.planes = 5;
        if (crl_draw_stack(A, nullptr, &sd, st) != CRL_EINVAL || !crl_ctx_last_error(A)[0]) ok = false;
        crl_destroy(A), crl_destroy(B);
        (void)hipEventDestroy(flags_ready), (void)hipStreamDestroy(side), (void)hipHostFree(done_pinned);
        (void)hipFree(obsA), (void)hipFree(obsB), (void)hipFree(doneA), (void)hipFree(doneB), (void)hipFree(rewA), (void)hipFree(rewB), (void)hipFree(mask);
        for (int k = 0; k < 2; k++) (void)hipFree(sa[k]), (void)hipFree(sb[k]);
    }
    (void)hipFree(act);
    (void)hipStreamDestroy(st);
    return ok ? 0 : 1;
}
