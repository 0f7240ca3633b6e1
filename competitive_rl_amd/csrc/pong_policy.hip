// pong_policy.hip -- the built-in CNN opponents of cPongTournament-v0 (WEAK / MEDIUM) on device.
//
// Restates Policy.__call__ (reference utils/policy_serving.py:46-66) with use_light_model=True:
// the policy keeps its own stack of the last four 42x42 frames it was shown
// (FrameStackTensor.update without a mask, utils/utils.py:159-170), runs LightActorCritic
// (utils/network.py:73-93) on it and plays the argmax of the three logits.
//
// One kernel per call: u8 frames in, int32 actions out; nothing else touches HBM but the 7 KB
// stack per env.  The two convolutions fuse exactly because conv2 is 2x2 with stride 2: each of the
// 10x10 conv2 positions owns its 2x2 block of conv1 outputs (16 channels) and its 6x6x4 input
// patch.  One lane per conv2 position, five envs per 512-thread workgroup (500 lanes busy):
//   conv1: 4 positions x 16 channels x 64 taps = 4 096 FMAs per lane   (weights: scalar loads)
//   conv2: 16 channels x 64 taps                = 1 024 FMAs per lane   (weights: scalar loads)
//   actor: 3 x 16 per lane, then a fixed-order sum over the 100 lanes of an env in LDS
// = 516 800 FMAs per env, fp32 VALU (bf16/fp8 MFMA would change which action wins in close
// calls; the reference is fp32).  Roofline: 65 536 envs x 1.03 MFLOP = 67.7 GFLOP per call against
// the 78.6 TFLOP/s a CDNA4 chip issues with plain v_fma_f32 (157 with packed fp32).
// The stack is a ring of four planes: the new frame overwrites the oldest plane in place, so a call
// reads 3 planes + the frame and writes 1 plane per env (8.8 KB) instead of rolling the stack.
#include <string.h>

#include <vector>

#include "crl_internal.h"

namespace crl {

static constexpr int kDim = CRL_POLICY_DIM;            // 42
static constexpr int kPlane = kDim * kDim;             // 1764 bytes
static constexpr int kPlaneWords = kPlane / 4;         // 441
static constexpr int kEnvsPerWg = 5;
static constexpr int kPos = 100;                       // 10 x 10 conv2 positions
static constexpr int kPolicyThreads = 512;

struct PolicyWeights {
    const float *w1;  // [16][64]       conv1.weight [oc][ic][ky][kx]
    const float *b1;  // [16]
    const float *w2;  // [16][16][4]    conv2 weights regrouped [ic][oc][ky][kx]
    const float *b2;  // [16]
    const float *wa;  // [3][1600]      actor_linear.weight
    const float *ba;  // [3]
};

__global__ __launch_bounds__(kPolicyThreads) void pong_policy_light_kernel(PolicyWeights W, uint8_t *__restrict__ ring, int head,
                                                                           const uint8_t *__restrict__ frame, int64_t frame_stride,
                                                                           int32_t *__restrict__ actions, int64_t action_stride,
                                                                           float *__restrict__ logits_out, int64_t n) {
    __shared__ __attribute__((aligned(16))) uint8_t sh_in[kEnvsPerWg][CRL_POLICY_STACK][kPlane];  // logical order: oldest first
    __shared__ float sh_div[256];                                                              // b / 255.0f, correctly rounded
    __shared__ float sh_part[kEnvsPerWg * kPos][3];
    __shared__ float sh_logit[kEnvsPerWg][3];
    const int tid = threadIdx.x;
    const int64_t env0 = (int64_t)blockIdx.x * kEnvsPerWg;

    if (tid < 256) sh_div[tid] = (float)tid / 255.0f;
    // stage the four planes (dwords); the new frame also replaces the oldest ring plane
    for (int i = tid; i < kEnvsPerWg * CRL_POLICY_STACK * kPlaneWords; i += kPolicyThreads) {
        const int e = i / (CRL_POLICY_STACK * kPlaneWords);
        const int r = i - e * (CRL_POLICY_STACK * kPlaneWords);
        const int j = r / kPlaneWords, d = r - j * kPlaneWords;
        const int64_t env = env0 + e;
        uint32_t v = 0;
        if (env < n) {
            uint32_t *rp = reinterpret_cast<uint32_t *>(ring + env * (int64_t)(CRL_POLICY_STACK * kPlane));
            if (j < 3) {
                v = rp[((head + 1 + j) & 3) * kPlaneWords + d];
            } else {
                v = reinterpret_cast<const uint32_t *>(frame + env * frame_stride)[d];
                rp[head * kPlaneWords + d] = v;
            }
        }
        reinterpret_cast<uint32_t *>(&sh_in[e][0][0])[r] = v;
    }
    __syncthreads();

    const int e = tid / kPos, pos = tid - e * kPos;
    const bool live = tid < kEnvsPerWg * kPos && env0 + e < n;
    if (live) {
        const int y2 = pos / 10, x2 = pos - y2 * 10;
        float in[4][6][6];
#pragma unroll
        for (int ic = 0; ic < 4; ic++)
#pragma unroll
            for (int r = 0; r < 6; r++) {
                const uint8_t *row = &sh_in[e][ic][(4 * y2 + r) * kDim + 4 * x2];  // even offset
#pragma unroll
                for (int k = 0; k < 3; k++) {
                    const uint32_t two = *reinterpret_cast<const uint16_t *>(row + 2 * k);
                    in[ic][r][2 * k] = sh_div[two & 255u];
                    in[ic][r][2 * k + 1] = sh_div[two >> 8];
                }
            }
        float acc[16];
#pragma unroll
        for (int oc = 0; oc < 16; oc++) acc[oc] = W.b2[oc];
        for (int c = 0; c < 16; c++) {  // conv1 output channel == conv2 input channel
            const float *__restrict__ w1 = W.w1 + c * 64;
            const float bias = W.b1[c];
            float h00 = bias, h01 = bias, h10 = bias, h11 = bias;
#pragma unroll
            for (int ic = 0; ic < 4; ic++)
#pragma unroll
                for (int ky = 0; ky < 4; ky++)
#pragma unroll
                    for (int kx = 0; kx < 4; kx++) {
                        const float w = w1[ic * 16 + ky * 4 + kx];
                        h00 = __builtin_fmaf(w, in[ic][ky][kx], h00);
                        h01 = __builtin_fmaf(w, in[ic][ky][kx + 2], h01);
                        h10 = __builtin_fmaf(w, in[ic][ky + 2][kx], h10);
                        h11 = __builtin_fmaf(w, in[ic][ky + 2][kx + 2], h11);
                    }
            h00 = fmaxf(h00, 0.f), h01 = fmaxf(h01, 0.f), h10 = fmaxf(h10, 0.f), h11 = fmaxf(h11, 0.f);
            const float *__restrict__ w2 = W.w2 + c * 64;
#pragma unroll
            for (int oc = 0; oc < 16; oc++) {
                float a = acc[oc];
                a = __builtin_fmaf(w2[oc * 4 + 0], h00, a);
                a = __builtin_fmaf(w2[oc * 4 + 1], h01, a);
                a = __builtin_fmaf(w2[oc * 4 + 2], h10, a);
                a = __builtin_fmaf(w2[oc * 4 + 3], h11, a);
                acc[oc] = a;
            }
        }
        float l0 = 0.f, l1 = 0.f, l2 = 0.f;
#pragma unroll
        for (int oc = 0; oc < 16; oc++) {
            const float f = fmaxf(acc[oc], 0.f);
            l0 = __builtin_fmaf(W.wa[0 * 1600 + oc * kPos + pos], f, l0);
            l1 = __builtin_fmaf(W.wa[1 * 1600 + oc * kPos + pos], f, l1);
            l2 = __builtin_fmaf(W.wa[2 * 1600 + oc * kPos + pos], f, l2);
        }
        sh_part[tid][0] = l0, sh_part[tid][1] = l1, sh_part[tid][2] = l2;
    }
    __syncthreads();
    if (tid < kEnvsPerWg * 3) {  // fixed-order sum: results do not depend on scheduling
        const int pe = tid / 3, a = tid - pe * 3;
        float s = W.ba[a];
        for (int p = 0; p < kPos; p++) s += sh_part[pe * kPos + p][a];
        sh_logit[pe][a] = s;
    }
    __syncthreads();
    if (tid < kEnvsPerWg && env0 + tid < n) {
        const float a0 = sh_logit[tid][0], a1 = sh_logit[tid][1], a2 = sh_logit[tid][2];
        int best = 0;  // argmax, first index wins ties (torch.argmax)
        float bv = a0;
        if (a1 > bv) best = 1, bv = a1;
        if (a2 > bv) best = 2;
        actions[(env0 + tid) * action_stride] = best;
        if (logits_out) {
            float *lo = logits_out + (env0 + tid) * 3;
            lo[0] = a0, lo[1] = a1, lo[2] = a2;
        }
    }
}

// ring <-> logical order (tests, checkpoints): plane j of the model's stack is ring plane (head + j) & 3
__global__ void pong_policy_copy_stack_kernel(uint8_t *__restrict__ ring, uint8_t *__restrict__ ext, int head, int64_t words, int to_ring) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= words) return;
    const int64_t env = i / (CRL_POLICY_STACK * kPlaneWords);
    const int r = (int)(i - env * (CRL_POLICY_STACK * kPlaneWords));
    const int j = r / kPlaneWords, d = r - j * kPlaneWords;
    uint32_t *rp = reinterpret_cast<uint32_t *>(ring) + env * (CRL_POLICY_STACK * kPlaneWords) + ((head + j) & 3) * kPlaneWords + d;
    uint32_t *ep = reinterpret_cast<uint32_t *>(ext) + i;
    if (to_ring) *rp = *ep;
    else *ep = *rp;
}

}  // namespace crl

using namespace crl;

#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess) return crl_fail(CRL_EHIP, "%s: %s", #expr, hipGetErrorString(e_));   \
    } while (0)

struct crl_policy {
    int device = 0;
    int64_t n = 0;
    int head = 0;  // ring plane holding the OLDEST frame (the next one to be replaced)
    float *weights = nullptr;
    uint8_t *ring = nullptr;
    PolicyWeights W{};
};

extern "C" {

int crl_policy_create(int32_t device, int64_t num_envs, const float *conv1_w, const float *conv1_b, const float *conv2_w,
                      const float *conv2_b, const float *actor_w, const float *actor_b, crl_policy **out) {
    if (!out || num_envs <= 0 || !conv1_w || !conv1_b || !conv2_w || !conv2_b || !actor_w || !actor_b)
        return crl_fail(CRL_EINVAL, "crl_policy_create: bad arguments");
    HIP_TRY(hipSetDevice(device));
    crl_policy *p = new crl_policy();
    p->device = device, p->n = num_envs;
    // one blob: w1 1024 | b1 16 | w2 1024 | b2 16 | wa 4800 | ba 3 (+ pad)
    std::vector<float> blob(1024 + 16 + 1024 + 16 + 4800 + 4, 0.f);
    float *w1 = blob.data(), *b1 = w1 + 1024, *w2 = b1 + 16, *b2 = w2 + 1024, *wa = b2 + 16, *ba = wa + 4800;
    memcpy(w1, conv1_w, 1024 * sizeof(float));
    memcpy(b1, conv1_b, 16 * sizeof(float));
    for (int oc = 0; oc < 16; oc++)  // torch [oc][ic][ky][kx] -> [ic][oc][ky][kx]
        for (int ic = 0; ic < 16; ic++)
            for (int k = 0; k < 4; k++) w2[(ic * 16 + oc) * 4 + k] = conv2_w[(oc * 16 + ic) * 4 + k];
    memcpy(b2, conv2_b, 16 * sizeof(float));
    memcpy(wa, actor_w, 4800 * sizeof(float));
    memcpy(ba, actor_b, 3 * sizeof(float));
    hipError_t e = hipMalloc(&p->weights, blob.size() * sizeof(float));
    if (e == hipSuccess) e = hipMemcpy(p->weights, blob.data(), blob.size() * sizeof(float), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMalloc(&p->ring, (size_t)num_envs * CRL_POLICY_STACK * kPlane);
    if (e == hipSuccess) e = hipMemset(p->ring, 0, (size_t)num_envs * CRL_POLICY_STACK * kPlane);
    if (e != hipSuccess) {
        crl_policy_destroy(p);
        return crl_fail(e == hipErrorOutOfMemory ? CRL_ENOMEM : CRL_EHIP, "crl_policy_create: %s", hipGetErrorString(e));
    }
    p->W.w1 = p->weights, p->W.b1 = p->W.w1 + 1024, p->W.w2 = p->W.b1 + 16, p->W.b2 = p->W.w2 + 1024;
    p->W.wa = p->W.b2 + 16, p->W.ba = p->W.wa + 4800;
    *out = p;
    return CRL_OK;
}

void crl_policy_destroy(crl_policy *p) {
    if (!p) return;
    (void)hipSetDevice(p->device);
    if (p->weights) (void)hipFree(p->weights);
    if (p->ring) (void)hipFree(p->ring);
    delete p;
}

int crl_policy_reset(crl_policy *p, void *stream) {
    if (!p) return crl_fail(CRL_EINVAL, "crl_policy_reset: null policy");
    HIP_TRY(hipMemsetAsync(p->ring, 0, (size_t)p->n * CRL_POLICY_STACK * kPlane, (hipStream_t)stream));
    p->head = 0;
    return CRL_OK;
}

int crl_policy_act(crl_policy *p, const uint8_t *frame_dev, int64_t frame_stride, int32_t *actions_dev, int64_t action_stride,
                   float *logits_dev, void *stream) {
    if (!p || !frame_dev || !actions_dev) return crl_fail(CRL_EINVAL, "crl_policy_act: null argument");
    if (frame_stride < kPlane || (frame_stride & 3) || ((uintptr_t)frame_dev & 3) || action_stride < 1)
        return crl_fail(CRL_EINVAL, "crl_policy_act: frame_stride must be a multiple of 4 and >= 1764, frames 4-byte aligned");
    const unsigned grid = (unsigned)((p->n + kEnvsPerWg - 1) / kEnvsPerWg);
    hipLaunchKernelGGL(pong_policy_light_kernel, dim3(grid), dim3(kPolicyThreads), 0, (hipStream_t)stream, p->W, p->ring, p->head, frame_dev,
                       frame_stride, actions_dev, action_stride, logits_dev, p->n);
    HIP_TRY(hipGetLastError());
    p->head = (p->head + 1) & 3;
    return CRL_OK;
}

static int copy_stack(crl_policy *p, uint8_t *ext, int to_ring, void *stream) {
    if (!p || !ext) return crl_fail(CRL_EINVAL, "crl_policy stack copy: null argument");
    const int64_t words = p->n * CRL_POLICY_STACK * kPlaneWords;
    hipLaunchKernelGGL(pong_policy_copy_stack_kernel, dim3((unsigned)((words + 255) / 256)), dim3(256), 0, (hipStream_t)stream, p->ring, ext,
                       p->head, words, to_ring);
    HIP_TRY(hipGetLastError());
    return CRL_OK;
}

int crl_policy_get_stack(crl_policy *p, uint8_t *stack_out_dev, void *stream) { return copy_stack(p, stack_out_dev, 0, stream); }
int crl_policy_set_stack(crl_policy *p, const uint8_t *stack_in_dev, void *stream) {
    return copy_stack(p, const_cast<uint8_t *>(stack_in_dev), 1, stream);
}

}  // extern "C"
