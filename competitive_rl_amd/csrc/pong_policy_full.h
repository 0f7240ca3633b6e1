// pong_policy_full.h -- the full-size ActorCritic opponent (pong_policy_full.hip) as seen from pong_policy.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace crl {

static constexpr int kRingPlanePad = 1776;  // bytes per 42 x 42 plane of the frame ring (111 16-byte chunks), four planes per env

struct PolicyFull;  // device weights + activation scratch of one crl_policy

// weights: host pointers, torch layouts (conv1 [16][4][4][4], conv2 [32][16][4][4], conv3 [256][32][11][11], actor [3][256])
hipError_t policy_full_create(PolicyFull **out, int64_t num_envs, const float *conv1_w, const float *conv1_b, const float *conv2_w,
                              const float *conv2_b, const float *conv3_w, const float *conv3_b, const float *actor_w,
                              const float *actor_b);
void policy_full_destroy(PolicyFull *f);
// ring: the policy's frame ring (pong_policy.hip: kRingBytes per env, plane pitch kPlanePad), head: the plane to overwrite
hipError_t policy_full_act(PolicyFull *f, uint8_t *ring, int head, int64_t n, const uint8_t *frame_dev, int64_t frame_stride,
                           int32_t *actions_dev, int64_t action_stride, float *logits_dev, hipStream_t st);

}  // namespace crl
