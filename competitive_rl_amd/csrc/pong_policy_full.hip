// pong_policy_full.hip -- the full-size ActorCritic opponent of cPongTournament-v0 (STRONG / ALPHA_PONG's model) on device.
//
// Restates Policy.__call__ (reference utils/policy_serving.py:46-66) with use_light_model=False: the same four-frame
// stack as the light opponents (pong_policy.hip owns the ring), then ActorCritic.forward (utils/network.py:14-50):
//   x / 255 -> conv1 4->16 k4 s2 (20x20) -> ReLU -> conv2 16->32 k4 s2 pad 2 (11x11) -> ReLU -> conv3 32->256 k11 (1x1) -> ReLU
//   -> actor_linear 256->3, argmax.        4.79 MFLOP per env: conv1 0.82, conv2 1.98, conv3 1.98.
//
// All three convolutions are GEMMs on the fp32 matrix instruction v_mfma_f32_16x16x4_f32 (exact f32 products, f32
// accumulation: the reference is fp32 and close calls between two logits decide the action, so no bf16 / fp8 here):
//   conv1  C[(env, pos) x 16]  = im2col(stack)[.. x 64]   x W1^T   weights in 16 VGPRs, bytes gathered from the ring
//   conv2  C[(env, pos) x 32]  = im2col(act1)[.. x 256]   x W2^T   weights in LDS (33 KB), act1 gathered as float2
//   conv3  C[env x 256]        = act2[env x 3872]         x W3^T   128 x 128 x 16 LDS tiles, double buffered
// The k index inside a group of 16 is dealt kq-major (lane kq of the instruction's four k lanes takes k = 16 g + 4 kq + j in
// step j): a sum does not care about the order of its terms, and this way every operand fetch is one 8- or 16-byte load of
// CONSECUTIVE k (a row of the 4x4 window / four consecutive columns of a K-contiguous matrix) with no transposed copy of
// anything.  Activations go through HBM scratch (25.6 + 15.5 + 1 KB per env, sized for one chunk of envs): 4.8 MFLOP against
// 82 KB of traffic per env is 58 FLOP / byte, compute bound on the 157 TFLOP/s fp32 matrix pipe (HBM share: 0.7 of 2 ms at
// 65 536 envs).
#include "pong_policy_full.h"

#include <string.h>

#include <algorithm>
#include <vector>

#include "crl_internal.h"

namespace crl {

typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));

static constexpr int kFDim = 42, kFPlanePad = kRingPlanePad, kFRingBytes = 4 * kFPlanePad;  // the ring of pong_policy.hip
static constexpr int kC1 = 16, kP1 = 400;        // conv1: 16 channels x 20 x 20
static constexpr int kC2 = 32, kP2 = 121;        // conv2: 32 channels x 11 x 11
static constexpr int kK3 = kC2 * kP2;            // 3872 = conv3's receptive field: the whole of act2
static constexpr int kC3 = 256;
static constexpr int kAct1 = kC1 * kP1;          // 6400 floats per env
static constexpr int64_t kChunk = 65536;         // envs per pass (scratch: 42 KB per env)

static constexpr int kOffW1 = 0, kOffB1 = kOffW1 + 1024, kOffW2 = kOffB1 + 16, kOffB2 = kOffW2 + kC2 * 256, kOffW3 = kOffB2 + 32;
static constexpr int kOffB3 = kOffW3 + kC3 * kK3, kOffWa = kOffB3 + kC3, kOffBa = kOffWa + 3 * kC3, kBlob = kOffBa + 4;

struct PolicyFull {
    int64_t n = 0, chunk = 0;
    int cus = 256;
    float *w = nullptr;     // w1 | b1 | w2 | b2 | w3 | b3 | wa | ba, torch layouts
    float *act1 = nullptr;  // [chunk][16][400] after ReLU
    float *act2 = nullptr;  // [chunk][32][121] after ReLU = [chunk][3872]
    float *feat = nullptr;  // [chunk][256] after ReLU
};

#define MFMA4(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

// the new frame replaces the oldest plane of the ring (FrameStackTensor.update without a mask, utils/utils.py:159-170)
__global__ __launch_bounds__(256) void policy_full_push_kernel(uint8_t *__restrict__ ring, int head, const uint8_t *__restrict__ frame,
                                                               int64_t frame_stride, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n * (kFDim * kFDim / 4)) return;
    const int64_t env = i / (kFDim * kFDim / 4);
    const int w = (int)(i - env * (kFDim * kFDim / 4));
    reinterpret_cast<uint32_t *>(ring + env * kFRingBytes + head * kFPlanePad)[w] =
        reinterpret_cast<const uint32_t *>(frame + env * frame_stride)[w];
}

// conv1 + ReLU.  One wavefront per tile of 16 positions of one env (25 tiles); A = pixels / 255 (lane: position li, window row
// lk, the row's four pixels in the four steps), B = weights (lane: channel li), D: positions 4 lk + r x channel li.
__global__ __launch_bounds__(256) void policy_full_conv1_kernel(const uint8_t *__restrict__ ring, int head, const float *__restrict__ w1,
                                                                const float *__restrict__ b1, float *__restrict__ act1, int64_t n) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 15, lk = lane >> 4;
    f4 wr[4];
#pragma unroll
    for (int g = 0; g < 4; g++) wr[g] = *reinterpret_cast<const f4 *>(w1 + li * 64 + g * 16 + lk * 4);
    const float bias = b1[li];
    const int64_t tiles = n * (kP1 / 16);
    for (int64_t t = (int64_t)blockIdx.x * 4 + wave; t < tiles; t += (int64_t)gridDim.x * 4) {
        const int64_t env = t / (kP1 / 16);
        const int tl = (int)(t - env * (kP1 / 16));
        const int p = tl * 16 + li, y = p / 20, x = p - y * 20;
        const uint8_t *base = ring + env * kFRingBytes + (2 * y + lk) * kFDim + 2 * x;
        f4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int g = 0; g < 4; g++) {  // plane g of the stack (oldest first) is ring plane (head + 1 + g) & 3 once the frame is in
            const uint8_t *q = base + ((head + 1 + g) & 3) * kFPlanePad;
            const uint32_t lo = *reinterpret_cast<const uint16_t *>(q), hi = *reinterpret_cast<const uint16_t *>(q + 2);
            acc = MFMA4((float)(lo & 255u) / 255.0f, wr[g][0], acc);
            acc = MFMA4((float)(lo >> 8) / 255.0f, wr[g][1], acc);
            acc = MFMA4((float)(hi & 255u) / 255.0f, wr[g][2], acc);
            acc = MFMA4((float)(hi >> 8) / 255.0f, wr[g][3], acc);
        }
        f4 o;
#pragma unroll
        for (int r = 0; r < 4; r++) o[r] = fmaxf(acc[r] + bias, 0.f);
        *reinterpret_cast<f4 *>(act1 + env * kAct1 + li * kP1 + tl * 16 + 4 * lk) = o;
    }
}

// conv2 (k4 s2 pad 2) + ReLU.  Tiles of 16 positions q = env * 121 + pos; A = act1 (lane: position li, window row lk: two
// float2 loads per input channel, zero outside the 20 x 20 plane -- the window's column pairs are either inside or outside as
// a whole), B = weights from LDS (one 16-byte read per channel block and input channel).
static constexpr int kW2Pitch = 260;  // floats per output channel in LDS: 65 x 16 bytes, odd -> the 16 lanes of a read phase hit 16 bank groups
__global__ __launch_bounds__(256) void policy_full_conv2_kernel(const float *__restrict__ act1, const float *__restrict__ w2,
                                                                const float *__restrict__ b2, float *__restrict__ act2, int64_t n) {
    __shared__ __attribute__((aligned(16))) float sw[kC2 * kW2Pitch];
    for (int i = threadIdx.x; i < kC2 * 64; i += 256) {
        const int r = i >> 6, c = i & 63;
        *reinterpret_cast<f4 *>(sw + r * kW2Pitch + 4 * c) = *reinterpret_cast<const f4 *>(w2 + r * 256 + 4 * c);
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 15, lk = lane >> 4;
    const float bias0 = b2[li], bias1 = b2[16 + li];
    const float *swa = sw + li * kW2Pitch + 4 * lk, *swb = swa + 16 * kW2Pitch;
    const int64_t total = n * kP2, tiles = (total + 15) / 16;
    for (int64_t t = (int64_t)blockIdx.x * 4 + wave; t < tiles; t += (int64_t)gridDim.x * 4) {
        int64_t q = t * 16 + li;
        if (q >= total) q = total - 1;  // the last tile's spare rows repeat a valid position; their results are not stored
        const int64_t env = q / kP2;
        const int pos = (int)(q - env * kP2), y = pos / 11, x = pos - y * 11;
        const int row = 2 * y - 2 + lk;
        const bool rok = row >= 0 && row < 20, ok01 = rok && x > 0, ok23 = rok && x < 10;
        const float *base = act1 + env * kAct1 + row * 20 + 2 * x - 2;
        f4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 8
        for (int g = 0; g < kC1; g++) {
            f2 v01 = {0.f, 0.f}, v23 = {0.f, 0.f};
            if (ok01) v01 = *reinterpret_cast<const f2 *>(base + g * kP1);
            if (ok23) v23 = *reinterpret_cast<const f2 *>(base + g * kP1 + 2);
            const f4 wa = *reinterpret_cast<const f4 *>(swa + 16 * g), wb = *reinterpret_cast<const f4 *>(swb + 16 * g);
            acc0 = MFMA4(v01[0], wa[0], acc0), acc1 = MFMA4(v01[0], wb[0], acc1);
            acc0 = MFMA4(v01[1], wa[1], acc0), acc1 = MFMA4(v01[1], wb[1], acc1);
            acc0 = MFMA4(v23[0], wa[2], acc0), acc1 = MFMA4(v23[0], wb[2], acc1);
            acc0 = MFMA4(v23[1], wa[3], acc0), acc1 = MFMA4(v23[1], wb[3], acc1);
        }
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int64_t q2 = t * 16 + 4 * lk + r;
            if (q2 < total) {
                const int64_t e2 = q2 / kP2;
                float *o = act2 + e2 * kK3 + (q2 - e2 * kP2);
                o[li * kP2] = fmaxf(acc0[r] + bias0, 0.f);
                o[(16 + li) * kP2] = fmaxf(acc1[r] + bias1, 0.f);
            }
        }
    }
}

// conv3 + ReLU = feat[env][oc] = relu(b3[oc] + sum_k act2[env][k] * w3[oc][k]): both operands K-contiguous.  Workgroup tile 128 envs
// x 128 channels, 2 x 2 wavefronts of 64 x 64 (16 accumulators), K in slabs of 16 through two LDS buffers; a wavefront reads
// each operand of a slab with one 16-byte LDS read per 16-row block (4 + 4 reads for 64 matrix instructions).
static constexpr int kGM = 128, kGN = 128, kGPitch = 20;  // pitch 20 floats = 5 x 16 bytes, odd: conflict-free 16-lane read phases
__global__ __launch_bounds__(256) void policy_full_conv3_kernel(const float *__restrict__ act2, const float *__restrict__ w3,
                                                                const float *__restrict__ b3, float *__restrict__ feat, int64_t n) {
    __shared__ __attribute__((aligned(16))) float sA[2][kGM * kGPitch];
    __shared__ __attribute__((aligned(16))) float sB[2][kGN * kGPitch];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, lk = lane >> 4, wm = wave >> 1, wn = wave & 1;
    const int64_t m0 = (int64_t)blockIdx.x * kGM;
    const int n0 = blockIdx.y * kGN;
    const int lr = tid >> 2, lc = tid & 3;  // loader: rows lr and lr + 64, 16-byte column lc of the slab
    const float *pa0 = act2 + std::min<int64_t>(m0 + lr, n - 1) * kK3 + 4 * lc;  // rows past the last env repeat it (not stored)
    const float *pa1 = act2 + std::min<int64_t>(m0 + lr + 64, n - 1) * kK3 + 4 * lc;
    const float *pb0 = w3 + (int64_t)(n0 + lr) * kK3 + 4 * lc, *pb1 = pb0 + (int64_t)64 * kK3;
    const int so0 = lr * kGPitch + 4 * lc, so1 = so0 + 64 * kGPitch;
    f4 ga0 = *reinterpret_cast<const f4 *>(pa0), ga1 = *reinterpret_cast<const f4 *>(pa1);
    f4 gb0 = *reinterpret_cast<const f4 *>(pb0), gb1 = *reinterpret_cast<const f4 *>(pb1);
    *reinterpret_cast<f4 *>(&sA[0][so0]) = ga0, *reinterpret_cast<f4 *>(&sA[0][so1]) = ga1;
    *reinterpret_cast<f4 *>(&sB[0][so0]) = gb0, *reinterpret_cast<f4 *>(&sB[0][so1]) = gb1;
    __syncthreads();
    f4 acc[4][4];
#pragma unroll
    for (int mb = 0; mb < 4; mb++)
#pragma unroll
        for (int nb = 0; nb < 4; nb++) acc[mb][nb] = f4{0.f, 0.f, 0.f, 0.f};
    const int ra = (wm * 64 + li) * kGPitch + 4 * lk, rb = (wn * 64 + li) * kGPitch + 4 * lk;
    constexpr int KT = kK3 / 16;  // 242 slabs
    for (int kt = 0; kt < KT; kt++) {
        const int cur = kt & 1;
        if (kt + 1 < KT) {
            const int o = 16 * (kt + 1);
            ga0 = *reinterpret_cast<const f4 *>(pa0 + o), ga1 = *reinterpret_cast<const f4 *>(pa1 + o);
            gb0 = *reinterpret_cast<const f4 *>(pb0 + o), gb1 = *reinterpret_cast<const f4 *>(pb1 + o);
        }
        f4 a[4], b[4];
#pragma unroll
        for (int i = 0; i < 4; i++) {
            a[i] = *reinterpret_cast<const f4 *>(&sA[cur][ra + i * 16 * kGPitch]);
            b[i] = *reinterpret_cast<const f4 *>(&sB[cur][rb + i * 16 * kGPitch]);
        }
#pragma unroll
        for (int j = 0; j < 4; j++)
#pragma unroll
            for (int mb = 0; mb < 4; mb++)
#pragma unroll
                for (int nb = 0; nb < 4; nb++) acc[mb][nb] = MFMA4(a[mb][j], b[nb][j], acc[mb][nb]);
        if (kt + 1 < KT) {
            *reinterpret_cast<f4 *>(&sA[cur ^ 1][so0]) = ga0, *reinterpret_cast<f4 *>(&sA[cur ^ 1][so1]) = ga1;
            *reinterpret_cast<f4 *>(&sB[cur ^ 1][so0]) = gb0, *reinterpret_cast<f4 *>(&sB[cur ^ 1][so1]) = gb1;
        }
        __syncthreads();
    }
#pragma unroll
    for (int nb = 0; nb < 4; nb++) {
        const int oc = n0 + wn * 64 + nb * 16 + li;
        const float bias = b3[oc];
#pragma unroll
        for (int mb = 0; mb < 4; mb++)
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int64_t env = m0 + wm * 64 + mb * 16 + 4 * lk + r;
                if (env < n) feat[env * kC3 + oc] = fmaxf(acc[mb][nb][r] + bias, 0.f);
            }
    }
}

// actor_linear + argmax (first maximum, like torch.argmax / numpy): one wavefront per env, a fixed-shape butterfly sum
__global__ __launch_bounds__(256) void policy_full_actor_kernel(const float *__restrict__ feat, const float *__restrict__ wa,
                                                                const float *__restrict__ ba, int32_t *__restrict__ actions,
                                                                int64_t action_stride, float *__restrict__ logits, int64_t n) {
    const int lane = threadIdx.x & 63;
    const int64_t env = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (env >= n) return;
    const f4 f = *reinterpret_cast<const f4 *>(feat + env * kC3 + 4 * lane);
    float s[3];
#pragma unroll
    for (int c = 0; c < 3; c++) {
        const f4 w = *reinterpret_cast<const f4 *>(wa + c * kC3 + 4 * lane);
        s[c] = ((f[0] * w[0] + f[1] * w[1]) + f[2] * w[2]) + f[3] * w[3];
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) s[c] += __shfl_xor(s[c], off);
        s[c] += ba[c];
    }
    if (lane == 0) {
        int a = 0;
        if (s[1] > s[0]) a = 1;
        if (s[2] > fmaxf(s[0], s[1])) a = 2;
        actions[env * action_stride] = a;
        if (logits) logits[env * 3] = s[0], logits[env * 3 + 1] = s[1], logits[env * 3 + 2] = s[2];
    }
}

hipError_t policy_full_create(PolicyFull **out, int64_t num_envs, const float *conv1_w, const float *conv1_b, const float *conv2_w,
                              const float *conv2_b, const float *conv3_w, const float *conv3_b, const float *actor_w,
                              const float *actor_b) {
    PolicyFull *f = new PolicyFull();
    f->n = num_envs, f->chunk = std::min<int64_t>(num_envs, kChunk);
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&f->cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess ||
        f->cus <= 0)
        f->cus = 256;
    std::vector<float> blob(kBlob, 0.f);
    memcpy(blob.data() + kOffW1, conv1_w, 1024 * sizeof(float)), memcpy(blob.data() + kOffB1, conv1_b, 16 * sizeof(float));
    memcpy(blob.data() + kOffW2, conv2_w, (size_t)kC2 * 256 * sizeof(float)), memcpy(blob.data() + kOffB2, conv2_b, kC2 * sizeof(float));
    memcpy(blob.data() + kOffW3, conv3_w, (size_t)kC3 * kK3 * sizeof(float)), memcpy(blob.data() + kOffB3, conv3_b, kC3 * sizeof(float));
    memcpy(blob.data() + kOffWa, actor_w, (size_t)3 * kC3 * sizeof(float)), memcpy(blob.data() + kOffBa, actor_b, 3 * sizeof(float));
    hipError_t e = hipMalloc(&f->w, blob.size() * sizeof(float));
    if (e == hipSuccess) e = hipMemcpy(f->w, blob.data(), blob.size() * sizeof(float), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMalloc(&f->act1, (size_t)f->chunk * kAct1 * sizeof(float));
    if (e == hipSuccess) e = hipMalloc(&f->act2, (size_t)f->chunk * kK3 * sizeof(float));
    if (e == hipSuccess) e = hipMalloc(&f->feat, (size_t)f->chunk * kC3 * sizeof(float));
    if (e != hipSuccess) {
        policy_full_destroy(f);
        return e;
    }
    *out = f;
    return hipSuccess;
}

void policy_full_destroy(PolicyFull *f) {
    if (!f) return;
    if (f->w) (void)hipFree(f->w);
    if (f->act1) (void)hipFree(f->act1);
    if (f->act2) (void)hipFree(f->act2);
    if (f->feat) (void)hipFree(f->feat);
    delete f;
}

hipError_t policy_full_act(PolicyFull *f, uint8_t *ring, int head, int64_t n, const uint8_t *frame_dev, int64_t frame_stride,
                           int32_t *actions_dev, int64_t action_stride, float *logits_dev, hipStream_t st) {
    const int64_t words = n * (kFDim * kFDim / 4);
    hipLaunchKernelGGL(policy_full_push_kernel, dim3((unsigned)((words + 255) / 256)), dim3(256), 0, st, ring, head, frame_dev, frame_stride, n);
    for (int64_t e0 = 0; e0 < n; e0 += f->chunk) {
        const int64_t c = std::min<int64_t>(f->chunk, n - e0);
        const int64_t t1 = (c * (kP1 / 16) + 3) / 4, t2 = ((c * kP2 + 15) / 16 + 3) / 4;
        hipLaunchKernelGGL(policy_full_conv1_kernel, dim3((unsigned)std::min<int64_t>(t1, (int64_t)f->cus * 8)), dim3(256), 0, st,
                           ring + e0 * kFRingBytes, head, f->w + kOffW1, f->w + kOffB1, f->act1, c);
        hipLaunchKernelGGL(policy_full_conv2_kernel, dim3((unsigned)std::min<int64_t>(t2, (int64_t)f->cus * 4)), dim3(256), 0, st, f->act1,
                           f->w + kOffW2, f->w + kOffB2, f->act2, c);
        hipLaunchKernelGGL(policy_full_conv3_kernel, dim3((unsigned)((c + kGM - 1) / kGM), kC3 / kGN), dim3(256), 0, st, f->act2,
                           f->w + kOffW3, f->w + kOffB3, f->feat, c);
        hipLaunchKernelGGL(policy_full_actor_kernel, dim3((unsigned)((c + 3) / 4)), dim3(256), 0, st, f->feat, f->w + kOffWa, f->w + kOffBa,
                           actions_dev + e0 * action_stride, action_stride, logits_dev ? logits_dev + e0 * 3 : nullptr, c);
    }
    return hipGetLastError();
}

}  // namespace crl
