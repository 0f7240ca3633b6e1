// crl_internal.h -- shared between the per-env-kind host files of libcrl_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <vector>

#include "../../include/crl.h"

int crl_fail(int code, const char *fmt, ...);
void crl_fail_no_ctx(void);  // first line of every entry point without a crl_ctx argument

// Timing ablations that give WRONG results (skip a phase to size it) exist only in a profiling build (-DCRL_ABLATION):
// in the shipped library their switches read as zero and the compiler drops the branches.
#ifdef CRL_ABLATION
#define CRL_ABL(x) (x)
#else
#define CRL_ABL(x) 0
#endif

struct crl_event_pair {
    hipEvent_t a, b;
};

// hipEvent brackets around the kernels of a step, on the launch stream (crl_kernel_timing)
// slots: 0 = dynamics (CarRacing: everything in front of the frame launch on the bulk stream), 1 = the observation's draw, 2 = CarRacing's
// touching solve (the kernel that ends a steady-state step: crl_kernel_time_stats reports its mean and its longest launch)
static constexpr int kTimerSlots = 3;
struct crl_timer {
    bool on = false;
    std::vector<crl_event_pair> ev[kTimerSlots];
    std::vector<crl_event_pair> pool;  // recycled events
    double ms[kTimerSlots] = {0, 0, 0}, max_ms[kTimerSlots] = {0, 0, 0};
    int64_t cnt[kTimerSlots] = {0, 0, 0};
};
void crl_timer_begin(crl_timer *t, int which, hipStream_t st);
void crl_timer_end(crl_timer *t, int which, hipStream_t st);

struct crl_car_ctx;
int crl_car_create(const crl_opts *opts, const uint32_t *text_bits_host, crl_car_ctx **out);
void crl_car_destroy(crl_car_ctx *c);
void crl_car_seed(crl_car_ctx *c, uint64_t seed);
int64_t crl_car_obs_bytes(const crl_car_ctx *c);
int crl_car_reset(crl_car_ctx *c, uint8_t *obs_dev, hipStream_t st);
int crl_car_render(crl_car_ctx *c, uint8_t *obs_dev, hipStream_t st);
const uint8_t *crl_car_terminal_frames(const crl_car_ctx *c);
const uint8_t *crl_car_done_flags(const crl_car_ctx *c);
const int32_t *crl_car_info_steps(const crl_car_ctx *c);
const int32_t *crl_car_info_elapsed(const crl_car_ctx *c);
int crl_car_players(const crl_car_ctx *c);
int crl_car_step(crl_car_ctx *c, const float *actions_dev, uint8_t *obs_dev, float *rew_dev, uint8_t *done_dev, hipStream_t st,
                 crl_timer *tm);
int crl_car_get_state_impl(crl_car_ctx *c, crl_car_env_state *out, int64_t first, int64_t count, hipStream_t st);
int crl_car_set_state_impl(crl_car_ctx *c, const crl_car_env_state *in, int64_t first, int64_t count, hipStream_t st);
int crl_car_get_track_impl(crl_car_ctx *c, int64_t env, int32_t *n_out, float *tile_poly, float *border_poly, uint8_t *border,
                           float *start_pose, hipStream_t st);
int crl_car_set_track_impl(crl_car_ctx *c, int64_t env, int32_t nt, const double *tile_poly, const double *border_poly,
                           const uint8_t *border, const float *start_pose, hipStream_t st);
int crl_car_cap_hits_impl(crl_car_ctx *c, int32_t *out4, hipStream_t st);
int crl_car_get_map_impl(crl_car_ctx *c, int64_t env, uint8_t *palette_host, int32_t *overflow, hipStream_t st);
int crl_car_set_replay_impl(crl_car_ctx *c, const double *u, const uint8_t *swap, int64_t attempts);

namespace crl {
// out[k] = frames[idx[k]] for `count` device-resident env indices (tile = bytes per env)
void launch_car_gather_frames(const uint8_t *frames, const int64_t *idx_dev, int64_t count, int64_t n, int64_t tile, uint8_t *out,
                              hipStream_t st);
}  // namespace crl
