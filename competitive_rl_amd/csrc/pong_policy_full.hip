// pong_policy_full.hip -- the full-size ActorCritic opponent of cPongTournament-v0 (STRONG / ALPHA_PONG's model) on device.
//
// Restates Policy.__call__ (reference utils/policy_serving.py:46-66) with use_light_model=False: the same four-frame
// stack as the light opponents (pong_policy.hip owns the ring), then ActorCritic.forward (utils/network.py:14-50):
//   x / 255 -> conv1 4->16 k4 s2 (20x20) -> ReLU -> conv2 16->32 k4 s2 pad 2 (11x11) -> ReLU -> conv3 32->256 k11 (1x1) -> ReLU
//   -> actor_linear 256->3, argmax.        4.79 MFLOP per env: conv1 0.82, conv2 1.98, conv3 1.98.
//
// All three convolutions are GEMMs on the matrix pipe with fp32 results (the reference is fp32 and close calls between two
// logits decide the action, so no bf16 / fp8 activations):
//   front kernel, one env per workgroup pass, act1 stays in LDS:
//     conv1  C[16 x (env, pos)]  = W1 x im2col(stack)[64 x ..]    v_mfma_f32_16x16x32_bf16 on exact operands (bytes; weights as
//                                                                  three bf16 terms), bytes prefetched one env ahead
//     conv2  C[32 x (env, pos)]  = W2 x im2col(act1)[256 x ..]    v_mfma_f32_16x16x4_f32, weights (33 KB) and act1 (37 KB) in LDS
//   conv3    C[env x 256]        = act2[env x 3872] x W3^T        v_mfma_f32_16x16x4_f32, 128 x 128 x 16 LDS tiles, double buffered
//   actor    one wavefront per env: 3 dot products of 256, argmax
// The k index inside a group of 16 is dealt kq-major (lane kq of the instruction's four k lanes takes k = 16 g + 4 kq + j in
// step j): a sum does not care about the order of its terms, and this way every operand fetch is one 8- or 16-byte read of
// CONSECUTIVE k (a row of the 4x4 window / four consecutive columns of a K-contiguous matrix) with no transposed copy of
// anything.  HBM traffic per env: 7 KB stack in, 15.5 KB act2 out and in again, 1 KB features: 4.8 MFLOP against 40 KB is
// compute bound on the 157 TFLOP/s fp32 matrix pipe.  Measured at 65 536 envs (profiles/r04_policy_full_stats.csv): front
// 1.34 ms (ideal 0.94), conv3 1.00 ms (ideal 0.83), actor 0.017 ms = 2.36 ms per call; the first version (conv1 and conv2
// as separate kernels through HBM, byte / 255 divisions on the vector pipe) took 4.4.
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
static constexpr int64_t kChunk = 65536;         // envs per pass (scratch: 42 KB per env)

static constexpr int kOffW1 = 0, kOffB1 = kOffW1 + 1024, kOffW2 = kOffB1 + 16, kOffB2 = kOffW2 + kC2 * 256, kOffW3 = kOffB2 + 32;
static constexpr int kOffB3 = kOffW3 + kC3 * kK3, kOffWa = kOffB3 + kC3, kOffBa = kOffWa + 3 * kC3, kBlob = kOffBa + 4;

struct PolicyFull {
    int64_t n = 0, chunk = 0;
    int cus = 256;
    float *w = nullptr;     // w1 | b1 | w2 | b2 | w3 | b3 | wa | ba, torch layouts
    float *act2 = nullptr;  // [chunk][32][121] after ReLU = [chunk][3872]
    float *feat = nullptr;  // [chunk][256] after ReLU
};

#define MFMA4(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

// conv1 + ReLU + conv2 + ReLU of one env per workgroup pass, act1 never leaves the CU.
//   conv1 on the bf16 matrix instruction at fp32 accuracy, like the light opponents' kernel: the inputs are integers 0..255 --
//   exact in bf16 -- and each weight (pre-divided by 255) is the sum of three bf16 terms, so all products are exact and
//   v_mfma_f32_16x16x32_bf16 accumulates them in fp32: 6 instructions x 16 cycles per tile of 16 positions instead of 16 x 32.
//   A = pixels (lane: position li; k block lk = plane pair member lk >> 1, window rows 2 (lk & 1) + {0, 1}, four columns), B =
//   weights in k order 32 h + 8 lk + j = torch's own.  The 20 x 20 x 16 result goes to LDS inside a zero border of two (the
//   padding of conv2), so conv2's window reads need no bounds tests.
//   conv2: a wavefront owns two tiles of 16 positions (121 = 8 tiles, the last one ragged): per input channel four 8-byte LDS
//   reads of the window rows (A) and two 16-byte reads of the weights (B) feed 16 fp32 matrix instructions on four accumulators.
// The new frame is read straight from the caller's buffer (plane 3 of the stack) and copied over the oldest ring plane here.
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
__device__ inline uint32_t pk_bytes_bf16(uint32_t two) {  // the two low bytes of `two` as a pair of bf16
    const bf2 v = {(__bf16)(float)(two & 255u), (__bf16)(float)((two >> 8) & 255u)};
    return __builtin_bit_cast(uint32_t, v);
}
static constexpr int kS1Rows = 24, kS1Pitch = 580;  // padded plane 24 x 24 = 576 floats + 4: channels start 4 banks apart
static constexpr int kW2Pitch = 260;  // floats per output channel in LDS: 65 x 16 bytes, odd -> the 16 lanes of a read phase hit 16 bank groups
__global__ __launch_bounds__(256) void policy_full_front_kernel(uint8_t *__restrict__ ring, int head, const uint8_t *__restrict__ frame,
                                                                int64_t frame_stride, const float *__restrict__ w1,
                                                                const float *__restrict__ b1, const float *__restrict__ w2,
                                                                const float *__restrict__ b2, float *__restrict__ act2, int64_t n) {
    __shared__ __attribute__((aligned(16))) float sw[kC2 * kW2Pitch];
    __shared__ __attribute__((aligned(16))) float s1[kC1 * kS1Pitch];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, lk = lane >> 4;
    for (int i = tid; i < kC2 * 64; i += 256) {
        const int r = i >> 6, c = i & 63;
        *reinterpret_cast<f4 *>(sw + r * kW2Pitch + 4 * c) = *reinterpret_cast<const f4 *>(w2 + r * 256 + 4 * c);
    }
    for (int i = tid; i < kC1 * kS1Pitch; i += 256) s1[i] = 0.f;  // the border stays zero
    bf8 wB[3][2];
#pragma unroll
    for (int h = 0; h < 2; h++)
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const float w = w1[li * 64 + 32 * h + 8 * lk + j] / 255.0f;
            const __bf16 t0 = (__bf16)w;
            const float r1 = w - (float)t0;  // exact
            const __bf16 t1 = (__bf16)r1;
            const float r2 = r1 - (float)t1;  // exact
            wB[0][h][j] = t0, wB[1][h][j] = t1, wB[2][h][j] = (__bf16)r2;
        }
    f4 bias1, bias2[2];  // D rows = channels 4 lk + r
#pragma unroll
    for (int r = 0; r < 4; r++) bias1[r] = b1[4 * lk + r], bias2[0][r] = b2[4 * lk + r], bias2[1][r] = b2[16 + 4 * lk + r];
    const float *swa = sw + li * kW2Pitch + 4 * lk, *swb = swa + 16 * kW2Pitch;
    int ao[2];  // conv2: the window row lk of position 32 wave + 16 u + li in the padded plane
#pragma unroll
    for (int u = 0; u < 2; u++) {
        const int pos = min(32 * wave + 16 * u + li, kP2 - 1), y = pos / 11, x = pos - 11 * y;
        ao[u] = (2 * y + lk) * kS1Rows + 2 * x;
    }
    // conv1's input of one env: per tile two planes x two window rows x four bytes (2-byte aligned), fetched one env AHEAD -- the
    // loads are in flight while conv2 of the current env runs (a wavefront's seven tiles one after the other, each waiting for
    // its own loads, cost 0.86 ms of the kernel's 1.63 at 65 536 envs)
    uint32_t raw[7][2][2];
    auto prefetch = [&](int64_t env) {
        const uint8_t *fr = frame + env * frame_stride, *rg = ring + env * kFRingBytes;
        const int win = (2 * (lk & 1)) * kFDim;
        const uint8_t *pl[2] = {rg + ((head + 1 + (lk >> 1)) & 3) * kFPlanePad + win,                 // plane lk >> 1
                                ((lk >> 1) ? fr : rg + ((head + 3) & 3) * kFPlanePad) + win};          // plane 2 + (lk >> 1); 3 = the new frame
#pragma unroll
        for (int it = 0; it < 7; it++) {  // 25 tiles over 4 wavefronts: the spare turns repeat tile 24 (same values, same addresses)
            const int tl = min(wave + 4 * it, kP1 / 16 - 1);
            const int p = tl * 16 + li, y = p / 20, x = p - y * 20, off = 2 * y * kFDim + 2 * x;
#pragma unroll
            for (int h = 0; h < 2; h++) {
                __builtin_memcpy(&raw[it][h][0], pl[h] + off, 4);
                __builtin_memcpy(&raw[it][h][1], pl[h] + off + kFDim, 4);
            }
        }
    };
    __syncthreads();
    if ((int64_t)blockIdx.x < n) prefetch(blockIdx.x);
    for (int64_t env = blockIdx.x; env < n; env += gridDim.x) {
#pragma unroll
        for (int it = 0; it < 7; it++) {
            const int tl = min(wave + 4 * it, kP1 / 16 - 1);
            bf8 ax[2];
#pragma unroll
            for (int h = 0; h < 2; h++) {
                const uint32_t a = raw[it][h][0], c = raw[it][h][1];
                const u32x4 v = {pk_bytes_bf16(a), pk_bytes_bf16(a >> 16), pk_bytes_bf16(c), pk_bytes_bf16(c >> 16)};
                ax[h] = __builtin_bit_cast(bf8, v);
            }
            f4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int tm = 2; tm >= 0; tm--)  // smallest weight term first
#pragma unroll
                for (int h = 0; h < 2; h++) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wB[tm][h], ax[h], acc, 0, 0, 0);
            const int p = tl * 16 + li, y = p / 20, x = p - y * 20;  // D: channels 4 lk + r x position li
            float *o = s1 + 4 * lk * kS1Pitch + (y + 2) * kS1Rows + x + 2;
#pragma unroll
            for (int r = 0; r < 4; r++) o[r * kS1Pitch] = fmaxf(acc[r] + bias1[r], 0.f);
        }
        __syncthreads();
        if (env + gridDim.x < n) prefetch(env + gridDim.x);
        f4 acc[2][2];
#pragma unroll
        for (int u = 0; u < 2; u++) acc[u][0] = f4{0.f, 0.f, 0.f, 0.f}, acc[u][1] = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
        for (int g = 0; g < kC1; g++) {
            const f4 wa = *reinterpret_cast<const f4 *>(swa + 16 * g), wb = *reinterpret_cast<const f4 *>(swb + 16 * g);
#pragma unroll
            for (int u = 0; u < 2; u++) {
                const f2 v01 = *reinterpret_cast<const f2 *>(s1 + g * kS1Pitch + ao[u]);
                const f2 v23 = *reinterpret_cast<const f2 *>(s1 + g * kS1Pitch + ao[u] + 2);
                acc[u][0] = MFMA4(wa[0], v01[0], acc[u][0]), acc[u][1] = MFMA4(wb[0], v01[0], acc[u][1]);
                acc[u][0] = MFMA4(wa[1], v01[1], acc[u][0]), acc[u][1] = MFMA4(wb[1], v01[1], acc[u][1]);
                acc[u][0] = MFMA4(wa[2], v23[0], acc[u][0]), acc[u][1] = MFMA4(wb[2], v23[0], acc[u][1]);
                acc[u][0] = MFMA4(wa[3], v23[1], acc[u][0]), acc[u][1] = MFMA4(wb[3], v23[1], acc[u][1]);
            }
        }
#pragma unroll
        for (int u = 0; u < 2; u++) {  // D: channels 16 nb + 4 lk + r x position li: the 16 lanes of a row store 64 consecutive bytes
            const int pos = 32 * wave + 16 * u + li;
            if (pos < kP2) {
                float *o = act2 + env * kK3 + 4 * lk * kP2 + pos;
#pragma unroll
                for (int nb = 0; nb < 2; nb++)
#pragma unroll
                    for (int r = 0; r < 4; r++) o[(16 * nb + r) * kP2] = fmaxf(acc[u][nb][r] + bias2[nb][r], 0.f);
            }
        }
        {  // the new frame replaces the oldest plane of the ring (FrameStackTensor.update without a mask, utils/utils.py:159-170)
            const uint32_t *src = reinterpret_cast<const uint32_t *>(frame + env * frame_stride);
            uint32_t *dst = reinterpret_cast<uint32_t *>(ring + env * kFRingBytes + head * kFPlanePad);
            for (int i = tid; i < kFDim * kFDim / 4; i += 256) dst[i] = src[i];
        }
        __syncthreads();  // s1 is rewritten by the next env's conv1
    }
}

// conv3 + ReLU = feat[env][oc] = relu(b3[oc] + sum_k act2[env][k] * w3[oc][k]): both operands K-contiguous.  Workgroup tile 128 envs
// x 128 channels, 2 x 2 wavefronts of 64 x 64 (16 accumulators), K in slabs of 16 through two LDS buffers; a wavefront reads
// each operand of a slab with one 16-byte LDS read per 16-row block (4 + 4 reads for 64 matrix instructions).
static constexpr int kGM = 128, kGN = 128, kGPitch = 20;  // pitch 20 floats = 5 x 16 bytes, odd: conflict-free 16-lane read phases
__global__ __launch_bounds__(256, 4) void policy_full_conv3_kernel(const float *__restrict__ act2, const float *__restrict__ w3,
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
    if (f->act2) (void)hipFree(f->act2);
    if (f->feat) (void)hipFree(f->feat);
    delete f;
}

hipError_t policy_full_act(PolicyFull *f, uint8_t *ring, int head, int64_t n, const uint8_t *frame_dev, int64_t frame_stride,
                           int32_t *actions_dev, int64_t action_stride, float *logits_dev, hipStream_t st) {
    for (int64_t e0 = 0; e0 < n; e0 += f->chunk) {
        const int64_t c = std::min<int64_t>(f->chunk, n - e0);
        hipLaunchKernelGGL(policy_full_front_kernel, dim3((unsigned)std::min<int64_t>(c, (int64_t)f->cus * 2)), dim3(256), 0, st,
                           ring + e0 * kFRingBytes, head, frame_dev + e0 * frame_stride, frame_stride, f->w + kOffW1, f->w + kOffB1,
                           f->w + kOffW2, f->w + kOffB2, f->act2, c);
        hipLaunchKernelGGL(policy_full_conv3_kernel, dim3((unsigned)((c + kGM - 1) / kGM), kC3 / kGN), dim3(256), 0, st, f->act2,
                           f->w + kOffW3, f->w + kOffB3, f->feat, c);
        hipLaunchKernelGGL(policy_full_actor_kernel, dim3((unsigned)((c + 3) / 4)), dim3(256), 0, st, f->feat, f->w + kOffWa, f->w + kOffBa,
                           actions_dev + e0 * action_stride, action_stride, logits_dev ? logits_dev + e0 * 3 : nullptr, c);
    }
    return hipGetLastError();
}

}  // namespace crl
