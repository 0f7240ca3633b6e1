// pong_dynamics.hip -- per-env game update for cPongDouble, one lane per env.
//
// State is SoA in HBM so the 64 lanes of a wavefront load/store 64 consecutive
// int32/f64 words per array (fully coalesced); the update itself is ~100 scalar ops
// and runs entirely in registers.  The kernel's only job besides the update is to
// emit the 8-byte frame descriptors the raster kernels draw from -- no pixels here.
//
// Restates (reference, relative to competitive_rl/):
//   raw mode      DummyVecEnv.step_wait over PongDoublePlayerEnv._step
//                 utils/dummy_vec_env.py:51-63, pong/base_pong_env.py:113-142
//   wrapped mode  MaxAndSkipEnv.step + ClipRewardEnv.step + auto-reset
//                 utils/atari_wrappers.py:118-160,175-181
#include "pong_device.h"

namespace crl {

__global__ __launch_bounds__(256) void pong_reset_kernel(PongSoA s, ServeSrc src, int64_t n, int replicate) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    PongEnv e;
    e.serve_ctr = s.serve_ctr[i];
    e.rounds = 0, e.steps = 0;
    game_reset(e, src, i);
    store_env(s, i, e);
    s.wrap_steps[i] = 0;
    const uint64_t f = frame_of(e);
    s.obs_frames[i] = f, s.obs_frames[n + i] = f;
    // stack after reset = [0, 0, 0, reset_obs]  (FrameStackTensor.reset + update)
    for (int h = 0; h < 6; h++) s.ring[h * n + i] = replicate ? f : kBlankFrame;
    s.ring[6 * n + i] = f, s.ring[7 * n + i] = f;
    s.real_reward[2 * i] = 0.f, s.real_reward[2 * i + 1] = 0.f;
    s.num_steps[i] = 0;
}

template <bool WRAPPED>
__global__ __launch_bounds__(256) void pong_dynamics_kernel(PongSoA s, ServeSrc src, const int32_t *__restrict__ actions,
                                                            int64_t n, float *__restrict__ rew,
                                                            uint8_t *__restrict__ done_out, int single, int replicate) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    PongEnv e = load_env(s, i);
    // cPong-v0: one action per env, the right bat is the AutoBat = CHEAT_CODES on that side
    const int2 a = single ? make_int2(actions[i], CRL_PONG_CHEAT) : reinterpret_cast<const int2 *>(actions)[i];
    if (!action_ok(a.x) || !action_ok(a.y)) *s.bad_action = (action_ok(a.x) ? a.y : a.x) + 1;  // rare; any writer wins
    int r_l, r_r;
    bool done;
    if (!WRAPPED) {
        done = frame_step(e, a.x, a.y, src, i, r_l, r_r);
        const float fl = (float)r_l, fr = (float)r_r;
        if (rew) {
            if (single) rew[i] = fl;
            else reinterpret_cast<float2 *>(rew)[i] = make_float2(fl, fr);
        }
        reinterpret_cast<float2 *>(s.real_reward)[i] = make_float2(fl, fr);
        if (done) {
            const uint64_t t = frame_of(e);
            s.term_frames[i] = t, s.term_frames[n + i] = t;
            game_reset(e, src, i);
        }
        if (done_out) done_out[i] = done ? 1 : 0;
        s.obs_frames[i] = frame_of(e);
        store_env(s, i, e);
        return;
    }
    // ---- MaxAndSkipEnv.step, skip = 4: keep frames 2 and 3, break on done
    uint64_t k0 = s.keep[i], k1 = s.keep[n + i];
    double tot_l = 0.0, tot_r = 0.0;
    done = false;
#pragma unroll 1
    for (int k = 0; k < 4; k++) {
        done = frame_step(e, a.x, a.y, src, i, r_l, r_r);
        if (k == 2) k0 = frame_of(e);
        if (k == 3) k1 = frame_of(e);
        tot_l += (double)r_l, tot_r += (double)r_r;
        if (done) break;
    }
    s.keep[i] = k0, s.keep[n + i] = k1;
    // ---- ClipRewardEnv.step
    int ws = s.wrap_steps[i] + 1;
    reinterpret_cast<float2 *>(s.real_reward)[i] = make_float2((float)tot_l, (float)tot_r);
    s.num_steps[i] = ws;
    if (rew) {
        const float sl = (float)((tot_l > 0) - (tot_l < 0)), sr = (float)((tot_r > 0) - (tot_r < 0));
        if (single) rew[i] = sl;
        else reinterpret_cast<float2 *>(rew)[i] = make_float2(sl, sr);
    }
    if (done_out) done_out[i] = done ? 1 : 0;
    uint64_t f0 = k0, f1 = k1;
    if (done) {
        // terminal_observation = max over the (possibly stale) buffers; then reset():
        // the returned obs is WarpFrame(single reset frame)
        s.term_frames[i] = k0, s.term_frames[n + i] = k1;
        game_reset(e, src, i);
        ws = 0;
        f0 = f1 = frame_of(e);
    }
    s.wrap_steps[i] = ws;
    s.obs_frames[i] = f0, s.obs_frames[n + i] = f1;
    // ---- stack ring, FrameStackTensor.update (utils/utils.py:158-170): planes are kept as
    // the frame pairs that produce them, so "roll" moves 16 B per plane instead of R*R
    // pixels and the raster kernel re-draws the whole stack without reading old planes.
    // ring plane p (0 oldest .. 3 newest), slot sl: ring[(2p + sl) * n + i].
#pragma unroll
    for (int p = 0; p < 3; p++) {
        // done: FrameStackTensor zeroes the history; the FrameStack wrapper refills it with the reset frame
        uint64_t a0 = done ? (replicate ? f0 : kBlankFrame) : s.ring[(2 * (p + 1) + 0) * n + i];
        uint64_t a1 = done ? (replicate ? f1 : kBlankFrame) : s.ring[(2 * (p + 1) + 1) * n + i];
        s.ring[(2 * p + 0) * n + i] = a0, s.ring[(2 * p + 1) * n + i] = a1;
    }
    s.ring[6 * n + i] = f0, s.ring[7 * n + i] = f1;
    store_env(s, i, e);
}

__global__ __launch_bounds__(256) void pong_gather_frames_kernel(const uint64_t *__restrict__ frames, const int64_t *__restrict__ idx,
                                                                 int64_t count, int64_t n, uint64_t *__restrict__ ring_out) {
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= count) return;
    const int64_t i = idx[k];
    const bool ok = i >= 0 && i < n;
    ring_out[6 * count + k] = ok ? frames[i] : kBlankFrame;
    ring_out[7 * count + k] = ok ? frames[n + i] : kBlankFrame;
}

void launch_pong_gather_frames(const uint64_t *frames, const int64_t *idx_dev, int64_t count, int64_t n, uint64_t *ring_out,
                               hipStream_t st) {
    hipLaunchKernelGGL(pong_gather_frames_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, st, frames, idx_dev, count, n,
                       ring_out);
}

void launch_pong_reset(const PongSoA &s, const ServeSrc &src, int64_t n, PongMode mode, hipStream_t st) {
    hipLaunchKernelGGL(pong_reset_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, s, src, n, mode.replicate ? 1 : 0);
}

void launch_pong_dynamics(const PongSoA &s, const ServeSrc &src, const int32_t *actions, int64_t n, PongMode mode,
                          float *rew, uint8_t *done, hipStream_t st) {
    dim3 grid((unsigned)((n + 255) / 256)), block(256);
    const int single = mode.single ? 1 : 0, rep = mode.replicate ? 1 : 0;
    if (mode.wrapped)
        hipLaunchKernelGGL(pong_dynamics_kernel<true>, grid, block, 0, st, s, src, actions, n, rew, done, single, rep);
    else
        hipLaunchKernelGGL(pong_dynamics_kernel<false>, grid, block, 0, st, s, src, actions, n, rew, done, single, rep);
}

}  // namespace crl
