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
// patch.  One lane per conv2 position; a workgroup (256 threads) takes five envs at a time = 500
// positions in two passes:
//   conv1: 4 positions x 16 channels x 64 taps = 4 096 FMAs per lane
//   conv2: 16 channels x 64 taps                = 1 024 FMAs per lane
//   actor: 3 x 16 per lane, then a fixed-shape sum over the 100 lanes of an env in LDS
// = 516 800 FMAs per env, fp32 on the vector pipes as v_pk_fma_f32 (output channels in pairs, weights
// uniform in SGPRs): bf16/fp8 MFMA would change which action wins in close calls, the reference is
// fp32.  Roofline: 65 536 envs x 1.03 MFLOP = 67.7 GFLOP per call against 157.3 TFLOP/s packed fp32
// (measured issue rate on this chip 134-142; plain v_fma_f32 76.6).  DESIGN.md 4c has the history.
// The stack is a ring of four planes (padded to 111 16-byte chunks): the new frame overwrites the
// oldest plane in place, so a call reads 3 planes + the frame and writes 1 plane per env instead of
// rolling the stack; both go global -> LDS by LDS-DMA.
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "crl_internal.h"
#include "pong_policy_full.h"

namespace crl {

static constexpr int kDim = CRL_POLICY_DIM;            // 42
static constexpr int kPlane = kDim * kDim;             // 1764 bytes
static constexpr int kPlaneWords = kPlane / 4;         // 441
static constexpr int kPlanePad = 1776;                 // a plane in the ring / in LDS: 111 16-byte chunks (12 bytes of padding)
static constexpr int kPlaneChunks = kPlanePad / 16;    // 111
static_assert(kPlanePad == kRingPlanePad, "pong_policy_full.hip reads the same ring");
static constexpr int kRingBytes = CRL_POLICY_STACK * kPlanePad;  // 7104 per env
static constexpr int kEnvsPerWg = 5;
static constexpr int kPos = 100;                       // 10 x 10 conv2 positions
static constexpr int kPolicyThreads = 256;
static constexpr int kPasses = 2;                      // 500 positions per group over 256 lanes

typedef float f2 __attribute__((ext_vector_type(2)));

// Output channels are processed in PAIRS (2p, 2p + 1) so that the multiply-adds are v_pk_fma_f32
// (two fp32 FMAs per lane per issue): weights are stored as (w[2p], w[2p + 1]) pairs, uniform per
// wavefront (scalar loads), the activation is broadcast to both halves.
struct PolicyWeights {
    const float *stream;  // conv weights in consumption order, 16 batches of 16 pairs per channel pair:
                          //   [cp 8][ conv1 [ic 4][ky 4][kx 4] | conv2 [oc pair 8][ic half 2][k 4] ] pairs, + one batch of padding
    const float *b1;      // [16]
    const f2 *b2;         // [8]
    const float *wa;      // [3][1600]      actor_linear.weight
    const float *ba;      // [3]
};

// A batch of 16 weight pairs in 32 SGPRs.  The compiler puts s_load + s_waitcnt lgkmcnt(0) right in front of
// every use (scalar loads return out of order, so it can only wait for all of them): ~200 cycles exposed per 16
// FMAs.  Here the NEXT batch is requested before the current one is consumed, and the wait sits one batch later.
typedef float v16 __attribute__((ext_vector_type(16)));
struct WBatch {
    v16 a, b;
};
__device__ inline void wbatch_request(WBatch &w, const float *p) {
    asm volatile("s_load_dwordx16 %0, %2, 0x0\n\ts_load_dwordx16 %1, %2, 0x40" : "=&s"(w.a), "=&s"(w.b) : "s"(p) : "memory");
}
__device__ inline void wbatch_wait(WBatch &w) { asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(w.a), "+s"(w.b)); }
// Pins a batch's FMAs between the volatile request / wait statements around it (plain asm statements with no
// dependence on them may otherwise be scheduled across, which puts every wait right behind its own request).
__device__ inline void fence4(f2 &a, f2 &b, f2 &c, f2 &d) { asm volatile("" : "+v"(a), "+v"(b), "+v"(c), "+v"(d)); }
__device__ inline void fence2(f2 &a, f2 &b) { asm volatile("" : "+v"(a), "+v"(b)); }
__device__ inline f2 wbatch_get(const WBatch &w, int i) {  // i: compile-time constant
    return i < 8 ? f2{w.a[2 * i], w.a[2 * i + 1]} : f2{w.b[2 * (i - 8)], w.b[2 * (i - 8) + 1]};
}

// acc += w * broadcast(x.lo) / broadcast(x.hi): the compiler materialises a broadcast operand as a second
// register pair (doubling the 144 input registers), the instruction can select the half itself (op_sel).
__device__ inline void pk_fma_lo(f2 &acc, f2 w, f2 x) {
    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(acc) : "s"(w), "v"(x));
}
__device__ inline void pk_fma_hi(f2 &acc, f2 w, f2 x) {
    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "+v"(acc) : "s"(w), "v"(x));
}
__device__ inline void pk_fma_sel(f2 &acc, f2 w, f2 x, int half) {  // `half` is a compile-time constant after unrolling
    if (half) pk_fma_hi(acc, w, x);
    else pk_fma_lo(acc, w, x);
}
__device__ inline f2 relu2(f2 v) { return f2{fmaxf(v.x, 0.f), fmaxf(v.y, 0.f)}; }

// Request a group's data straight into LDS (global_load_lds: no registers are held while the loads are in flight):
// the three ring planes that stay (16-byte chunks, planes are padded to 111 chunks for this) and the new frame
// (dwords: frames are only 4-byte aligned) into the plane it replaces.  One wavefront-instruction fills a
// contiguous run of LDS (M0 = run base, lane l lands at base + l * size), so the work is cut into (env, plane,
// half) and (env, 64-dword run) pieces dealt to the eight wavefronts.
// Issued as inline asm: with the builtin the compiler assumes that any later LDS read may alias the transfer
// and waits for vmcnt(0) in front of the convolutions, which is exactly the overlap this is for.  The kernel
// waits itself (vmcnt(0) + barrier before the buffer is read).  M0 = LDS base of the run (one wait state
// between writing M0 and the LDS-DMA instruction); the compiler reserves M0 and sets it itself right before any
// instruction of its own that reads it.
typedef __attribute__((address_space(3))) void *lptr_t;
__device__ inline uint32_t lds_addr(const void *p) { return (uint32_t)(uintptr_t)(lptr_t)p; }
__device__ inline void lds_dma_b128(const void *src, uint32_t lds_base) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(src), "s"(__builtin_amdgcn_readfirstlane(lds_base)) : "memory");
}
__device__ inline void lds_dma_b32(const void *src, uint32_t lds_base) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dword %0, off" ::"v"(src), "s"(__builtin_amdgcn_readfirstlane(lds_base)) : "memory");
}
__device__ inline void group_request(uint8_t *shbuf, const uint8_t *__restrict__ ring, int head, const uint8_t *__restrict__ frame,
                                     int64_t frame_stride, int64_t env0, int envs_here, int wave, int lane) {
    for (int s = wave; s < kEnvsPerWg * 3; s += kPolicyThreads / 64) {
        const int fe = s / 3, j = s - fe * 3;
        if (fe >= envs_here) continue;
        const int pp = (head + 1 + j) & 3;
        const uint8_t *src = ring + (env0 + fe) * (int64_t)kRingBytes + pp * kPlanePad;
        uint8_t *dst = shbuf + (fe * CRL_POLICY_STACK + pp) * kPlanePad;
#pragma unroll
        for (int half = 0; half < 2; half++) {
            const int c = half * 64 + lane;
            if (c < kPlaneChunks) lds_dma_b128(src + c * 16, lds_addr(dst + half * 1024));
        }
    }
    for (int s = wave; s < kEnvsPerWg * 7; s += kPolicyThreads / 64) {
        const int fe = s / 7, q = s - fe * 7;
        if (fe >= envs_here) continue;
        const int d = q * 64 + lane;
        const uint8_t *src = frame + (env0 + fe) * frame_stride;
        uint8_t *dst = shbuf + (fe * CRL_POLICY_STACK + head) * kPlanePad + q * 256;
        if (d < kPlaneWords) lds_dma_b32(src + d * 4, lds_addr(dst));
    }
}

// after the group's loads have landed (vmcnt(0) + barrier): the new frame also replaces plane `head` of the ring
__device__ inline void group_write_back(const uint8_t *shbuf, uint8_t *__restrict__ ring, int head, int64_t env0, int envs_here, int tid) {
    for (int i = tid; i < envs_here * kPlaneChunks; i += kPolicyThreads) {
        const int fe = i / kPlaneChunks, c = i - fe * kPlaneChunks;
        const uint4 v = reinterpret_cast<const uint4 *>(shbuf + (fe * CRL_POLICY_STACK + head) * kPlanePad)[c];
        reinterpret_cast<uint4 *>(ring + (env0 + fe) * (int64_t)kRingBytes + head * kPlanePad)[c] = v;
    }
}

// Persistent workgroups, TWO per CU, four wavefronts each (173 VGPRs leave two wavefronts per SIMD: one of each
// workgroup).  A workgroup takes groups b, b + gridDim.x, ... of five envs; per group it (1) pulls the rings and
// frames into LDS, (2) runs the 500 conv2 positions in two passes of 256 lanes, (3) reduces the logits.  Steps
// (1) and (3) and the patch gather of (2) keep the FMA pipes idle; the two workgroups of a CU drift apart, so
// one's idle phases run under the other's convolutions.  Tables that do not depend on the group (actor weights,
// biases) are staged once.
template <int DBG>  // 0 production, 1 ablation switches (CRL_POLICY_DEBUG bits 1, 2), 2 production code + phase cycle counters (4)
__global__ __launch_bounds__(kPolicyThreads) void pong_policy_light_kernel(PolicyWeights W, uint8_t *__restrict__ ring, int head,
                                                                           const uint8_t *__restrict__ frame, int64_t frame_stride,
                                                                           int32_t *__restrict__ actions, int64_t action_stride,
                                                                           float *__restrict__ logits_out, int64_t n, int dbg_arg, int phase_sleeps,
                                                                           unsigned *__restrict__ ticket) {
    const int dbg = DBG == 1 ? dbg_arg : 0;  // CRL_POLICY_DEBUG (profiling only): 1 skip the convolutions, 2 skip the patch gather
    const bool timed = DBG != 0 && (dbg_arg & 4) && n >= 8192;  // the counters go into logits_out (needs n * 12 >= 66 560 bytes)
    __shared__ __attribute__((aligned(16))) uint8_t sh_in[kEnvsPerWg][CRL_POLICY_STACK][kPlanePad];
    __shared__ float sh_wa[3 * 1600];
    __shared__ __attribute__((aligned(8))) float sh_b2[16];  // conv2.bias; actor bias: no VMEM loads inside the loop,
    __shared__ float sh_ba[4];                               // a wait on one would also wait on the group in flight
    __shared__ float sh_part[kEnvsPerWg * kPos][3];
    __shared__ float sh_grp[kEnvsPerWg][3][4];
    __shared__ float sh_logit[kEnvsPerWg][3];
    const int tid = threadIdx.x;
    const int64_t ngroups = (n + kEnvsPerWg - 1) / kEnvsPerWg;

    if (tid < 16) sh_b2[tid] = reinterpret_cast<const float *>(W.b2)[tid];
    if (tid < 3) sh_ba[tid] = W.ba[tid];
    for (int i = tid; i < 3 * 1600; i += kPolicyThreads) sh_wa[i] = W.wa[i];
    // The conv1 bias is fetched with v_readlane from lanes 0..15 of the wavefront, so there is NO divergent
    // branch around the convolutions: every wavefront that runs the loop must hold it in those lanes (a
    // wavefront whose live lanes stop before lane 15 would read registers that were never written).  Idle
    // lanes (the last 12 of the workgroup, envs past the end) redo a valid position and drop the result.
    const float b1i = W.b1[tid & 15];  // lane l of every wavefront holds conv1.bias[l & 15]
    const int b1lane = __float_as_int(b1i);
    const int wave = tid >> 6, lane = tid & 63;
    __syncthreads();
    // The two workgroups of a CU start together and have the same period, so left alone they stay IN phase: both in
    // the convolutions (sharing the FMA pipes), then both in staging / reduction (pipes idle).  The second half of the
    // grid (the workgroups that land in the CUs' second slots) starts half a period late.
    {
        const int mode = phase_sleeps >> 8, reps = phase_sleeps & 255;
        const bool late = mode == 0 ? blockIdx.x >= (gridDim.x + 1) / 2 : mode == 1 ? (blockIdx.x & 1) : mode == 2 ? ((blockIdx.x >> 3) & 1) : ((blockIdx.x >> 8) & 1);
        if (late)
            for (int i = 0; i < reps; i++) __builtin_amdgcn_s_sleep(127);
    }

    long long tacc[6] = {0, 0, 0, 0, 0, 0}, tprev = 0;
    int gcount = 0;
#define CRL_TICK(K)                                        \
    if (timed) {                                           \
        const long long now_ = __builtin_readcyclecounter(); \
        tacc[K] += now_ - tprev;                           \
        tprev = now_;                                      \
    }
    if (timed) tprev = __builtin_readcyclecounter();
    // Groups are handed out by a ticket counter, not b, b + grid, ...: the SIMDs favour their OLDEST wavefront, so the
    // workgroup that reached a CU first runs about twice as fast as its co-resident (cycle-counter timelines: 48 k vs
    // 96 k cycles per group) and a static split leaves the slow half to finish alone.
    __shared__ unsigned sh_ticket;
    for (int64_t g = blockIdx.x; g < ngroups;) {
        const int64_t env0 = g * kEnvsPerWg;
        const int envs_here = (int)((n - env0) < kEnvsPerWg ? (n - env0) : kEnvsPerWg);
        if (tid == 0) sh_ticket = atomicAdd(ticket, 1u);
        group_request(&sh_in[0][0][0], ring, head, frame, frame_stride, env0, envs_here, wave, lane);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wavefront's share has landed
        __syncthreads();                                    // ... everybody's has
        const int64_t g_next = (int64_t)gridDim.x + sh_ticket;  // rewritten only after this iteration's last barrier
        CRL_TICK(0)
        group_write_back(&sh_in[0][0][0], ring, head, env0, envs_here, tid);
        CRL_TICK(1)
      for (int pass = 0; pass < kPasses; pass++) {
        const int task = pass * kPolicyThreads + tid;
        const int e = task < kEnvsPerWg * kPos ? task / kPos : kEnvsPerWg - 1;
        const int pos = task < kEnvsPerWg * kPos ? task - e * kPos : 0;
        const int y2 = pos / 10, x2 = pos - y2 * 10;
        const bool live = task < kEnvsPerWg * kPos && env0 + e < n;
        float l0 = 0.f, l1 = 0.f, l2 = 0.f;
        {
            // The 6x6x4 patch as floats, columns (2k, 2k + 1) in one register pair.  Measured with the cycle counter:
            // this gather, not the FMAs next to it, was a quarter of the kernel when it was 72 ds_read_u16 + 144
            // look-ups in a b/255 table per lane -- LDS-pipe bound (eight wavefronts of a CU share it), not latency
            // bound.  Now two ALIGNED dwords per 6-byte row piece in one ds_read2_b32 (the piece starts on a multiple
            // of 4 for even r and 2 bytes after one for odd r -- known at compile time; an unaligned ds_read_b64 is no
            // faster than the 216 small reads) and the division on the vector pipes: q = b * fl(1/255) + one
            // FMA-corrected Newton step = correctly rounded b / 255.0f for every byte (Markstein), i.e. the
            // reference's x / 255 bit for bit.
            f2 in[4][6][3];
            const f2 rcp = f2{1.0f / 255.0f, 1.0f / 255.0f}, m255 = f2{-255.0f, -255.0f};
#pragma unroll
            for (int ic = 0; ic < 4; ic++)
#pragma unroll
                for (int r = 0; r < 6; r++) {
                    // logical plane ic (oldest first) is ring plane (head + 1 + ic) & 3; bytes read past the piece stay
                    // inside the padded plane
                    const uint8_t *row = &sh_in[e][(head + 1 + ic) & 3][(4 * y2 + r) * kDim + 4 * x2];
                    const uint32_t *p32 = reinterpret_cast<const uint32_t *>(row - 2 * (r & 1));
                    uint32_t w0 = (dbg & 2) ? 0x01020304u : p32[0], w1 = (dbg & 2) ? 0x0506u : p32[1];
                    if (r & 1) w0 = (w0 >> 16) | (w1 << 16), w1 >>= 16;
                    const f2 b[3] = {f2{(float)(w0 & 255u), (float)((w0 >> 8) & 255u)}, f2{(float)((w0 >> 16) & 255u), (float)(w0 >> 24)},
                                     f2{(float)(w1 & 255u), (float)((w1 >> 8) & 255u)}};
#pragma unroll
                    for (int k = 0; k < 3; k++) {
                        const f2 q = b[k] * rcp;
                        const f2 rem = __builtin_elementwise_fma(q, m255, b[k]);
                        in[ic][r][k] = __builtin_elementwise_fma(rem, rcp, q);
                    }
                }
            CRL_TICK(2)
            if (timed && logits_out && tid == 0 && (blockIdx.x == 48 || blockIdx.x == 304) && gcount < 12)
                reinterpret_cast<long long *>(logits_out)[8192 + (blockIdx.x == 304) * 64 + gcount * 4 + pass * 2] = __builtin_readcyclecounter();
            f2 acc[8];  // conv2 accumulators, output channels (2p, 2p + 1)
#pragma unroll
            for (int p = 0; p < 8; p++) acc[p] = reinterpret_cast<const f2 *>(sh_b2)[p];
            const float *wp = W.stream;
            WBatch wa_, wb_;
            wbatch_request(wa_, wp);
#define CRL_CONV1_BATCH(WB, IC)                                                  \
    _Pragma("unroll") for (int ky = 0; ky < 4; ky++)                             \
        _Pragma("unroll") for (int kx = 0; kx < 4; kx++) {                       \
        const f2 w = wbatch_get(WB, ky * 4 + kx);                                \
        pk_fma_sel(h00, w, in[IC][ky][kx >> 1], kx & 1);                         \
        pk_fma_sel(h01, w, in[IC][ky][(kx >> 1) + 1], kx & 1);                   \
        pk_fma_sel(h10, w, in[IC][ky + 2][kx >> 1], kx & 1);                     \
        pk_fma_sel(h11, w, in[IC][ky + 2][(kx >> 1) + 1], kx & 1);               \
    }                                                                            \
    fence4(h00, h01, h10, h11);
#define CRL_CONV2_BATCH(WB, J)                                                   \
    {                                                                            \
        f2 a0 = acc[2 * (J)], a1 = acc[2 * (J) + 1];                             \
        pk_fma_lo(a0, wbatch_get(WB, 0), h00);                                   \
        pk_fma_lo(a1, wbatch_get(WB, 8), h00);                                   \
        pk_fma_lo(a0, wbatch_get(WB, 1), h01);                                   \
        pk_fma_lo(a1, wbatch_get(WB, 9), h01);                                   \
        pk_fma_lo(a0, wbatch_get(WB, 2), h10);                                   \
        pk_fma_lo(a1, wbatch_get(WB, 10), h10);                                  \
        pk_fma_lo(a0, wbatch_get(WB, 3), h11);                                   \
        pk_fma_lo(a1, wbatch_get(WB, 11), h11);                                  \
        pk_fma_hi(a0, wbatch_get(WB, 4), h00);                                   \
        pk_fma_hi(a1, wbatch_get(WB, 12), h00);                                  \
        pk_fma_hi(a0, wbatch_get(WB, 5), h01);                                   \
        pk_fma_hi(a1, wbatch_get(WB, 13), h01);                                  \
        pk_fma_hi(a0, wbatch_get(WB, 6), h10);                                   \
        pk_fma_hi(a1, wbatch_get(WB, 14), h10);                                  \
        pk_fma_hi(a0, wbatch_get(WB, 7), h11);                                   \
        pk_fma_hi(a1, wbatch_get(WB, 15), h11);                                  \
        fence2(a0, a1);                                                          \
        acc[2 * (J)] = a0, acc[2 * (J) + 1] = a1;                                \
    }
#define CRL_STEP(CUR, NXT, OFS, WORK)   \
    wbatch_wait(CUR);                   \
    wbatch_request(NXT, wp + (OFS));    \
    WORK
            for (int cp = 0; cp < ((dbg & 1) ? 0 : 8); cp++) {  // conv1 output channels (2cp, 2cp + 1) == conv2 input channels
                const f2 bias = f2{__int_as_float(__builtin_amdgcn_readlane(b1lane, 2 * cp)),
                                   __int_as_float(__builtin_amdgcn_readlane(b1lane, 2 * cp + 1))};
                f2 h00 = bias, h01 = bias, h10 = bias, h11 = bias;
                CRL_STEP(wa_, wb_, 32, CRL_CONV1_BATCH(wa_, 0))
                CRL_STEP(wb_, wa_, 64, CRL_CONV1_BATCH(wb_, 1))
                CRL_STEP(wa_, wb_, 96, CRL_CONV1_BATCH(wa_, 2))
                CRL_STEP(wb_, wa_, 128, CRL_CONV1_BATCH(wb_, 3))
                h00 = relu2(h00), h01 = relu2(h01), h10 = relu2(h10), h11 = relu2(h11);
                CRL_STEP(wa_, wb_, 160, CRL_CONV2_BATCH(wa_, 0))
                CRL_STEP(wb_, wa_, 192, CRL_CONV2_BATCH(wb_, 1))
                CRL_STEP(wa_, wb_, 224, CRL_CONV2_BATCH(wa_, 2))
                CRL_STEP(wb_, wa_, 256, CRL_CONV2_BATCH(wb_, 3))  // the next channel pair's first batch (padding after the last)
                wp += 256;
            }
            wbatch_wait(wa_);  // drain the padding request
            CRL_TICK(3)
            if (timed && logits_out && tid == 0 && (blockIdx.x == 48 || blockIdx.x == 304) && gcount < 12)
                reinterpret_cast<long long *>(logits_out)[8192 + (blockIdx.x == 304) * 64 + gcount * 4 + pass * 2 + 1] = __builtin_readcyclecounter();
#undef CRL_STEP
#undef CRL_CONV1_BATCH
#undef CRL_CONV2_BATCH
#pragma unroll
            for (int oc = 0; oc < 16; oc++) {
                const float f = fmaxf((oc & 1) ? acc[oc >> 1].y : acc[oc >> 1].x, 0.f);
                l0 = __builtin_fmaf(sh_wa[0 * 1600 + oc * kPos + pos], f, l0);
                l1 = __builtin_fmaf(sh_wa[1 * 1600 + oc * kPos + pos], f, l1);
                l2 = __builtin_fmaf(sh_wa[2 * 1600 + oc * kPos + pos], f, l2);
            }
        }
        if (live) sh_part[task][0] = l0, sh_part[task][1] = l1, sh_part[task][2] = l2;
        CRL_TICK(4)
      }
        __syncthreads();
        // fixed-shape sum over the 100 positions of an env (4 groups of 25, then the 4 groups): the result
        // does not depend on scheduling
        if (tid < kEnvsPerWg * 12) {
            const int pe = tid / 12, r = tid - pe * 12, a = r >> 2, grp = r & 3;
            float s = 0.f;
#pragma unroll
            for (int p = 0; p < 25; p++) s += sh_part[pe * kPos + grp * 25 + p][a];
            sh_grp[pe][a][grp] = s;
        }
        __syncthreads();
        if (tid < kEnvsPerWg * 3) {
            const int pe = tid / 3, a = tid - pe * 3;
            sh_logit[pe][a] = sh_ba[a] + ((sh_grp[pe][a][0] + sh_grp[pe][a][1]) + (sh_grp[pe][a][2] + sh_grp[pe][a][3]));
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
        __syncthreads();  // sh_logit / sh_in are rewritten by the next group
        CRL_TICK(5)
        gcount++;
        g = g_next;
    }
    if (timed && logits_out && lane == 0 && blockIdx.x < 512) {  // where the hardware put this wavefront (HW_ID: simd, cu, sh, se, ...)
        unsigned hwid;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        reinterpret_cast<unsigned *>(logits_out)[4096 + (blockIdx.x * 4 + wave) * 2] = hwid;
        reinterpret_cast<unsigned *>(logits_out)[4096 + (blockIdx.x * 4 + wave) * 2 + 1] = xcc;
    }
    if (timed && logits_out && tid == 0)  // profiling: cycles per phase of this workgroup's first wavefront
        for (int k = 0; k < 6; k++) logits_out[blockIdx.x * 6 + k] = (float)tacc[k];
#undef CRL_TICK
}

// ---------------------------------------------------------------------------------------------------------------
// fp32 MFMA version (round 2; CRL_POLICY_MFMA=0 selects the packed-FMA kernel above for A/B).
//
// v_mfma_f32_16x16x4_f32 multiplies exact f32 products at the vector peak (64 FLOP/clk/SIMD, MI355X_MICROARCH.md) but
// on the matrix pipe: the VALU stays free for the patch conversion, and the weights sit in VGPRs once per wavefront --
// no scalar weight stream, whose s_waitcnt in front of every 16 FMAs cost the packed-FMA kernel a third of its time.
//
// Both convolutions are computed TRANSPOSED, D^T[oc][pos] = W[oc][tap] x im2col^T[tap][pos], on tiles of 16 conv2
// positions q = env * 100 + pos2 (8 envs per group = 50 tiles, no padding):
//   conv1: the 2x2 conv1 outputs under a conv2 position form four parity CLASSES c = (ky0, kx0); per class one
//          accumulator tile D1[c] (rows = 16 output channels, cols = the 16 positions), 16 MFMAs each: step (ky, kx)
//          takes A = W1[oc = l & 15][ic = l >> 4][ky][kx] and B = x[ic = l >> 4][4 y2 + 2 ky0 + ky][4 x2 + 2 kx0 + kx] / 255
//          -- lane (position l & 15, plane l >> 4) needs the 6 x 6 patch of ONE plane: six aligned ds_read2_b32 per tile;
//   conv2: the accumulator layout (lane = position, registers r = channels 4 (l >> 4) + r) IS the B operand of the
//          next product: step (c, r) takes B = relu(D1[c][r]) as it stands and A = W2[oc2 = l & 15][ic = 4 (l >> 4) + r][c]
//          -- no transpose, no LDS round trip between the layers;
//   actor: each lane folds its four conv2 channels into three partial logits, two cross-lane adds finish the
//          position, and the 100 positions of an env are summed in a fixed shape (4 x 25, then 4) as before.
// 80 MFMAs per 16 positions = 500 per env: 65 536 envs x 500 x 32 cycles / 1 024 SIMDs = 1.02 M cycles (427 us at 2.4 GHz).
// One persistent 512-thread workgroup per CU -- two wavefronts per SIMD: a lone wavefront issues a vector instruction
// every 4 cycles, and the ~300 non-matrix instructions of a tile (gather, x / 255, relu, actor) next to its 80 MFMAs made
// the one-wavefront version 2.4x slower than the matrix pipe allows (1 018 us); with a partner, one's vector work runs
// under the other's MFMAs.  Groups are handed out by the ticket counter, the NEXT group's rings and frames land in the
// other half of a double buffer by LDS-DMA while this one is computed.
typedef float f4 __attribute__((ext_vector_type(4)));
#ifndef CRL_MFMA_WAVES
#define CRL_MFMA_WAVES 8
#endif
#ifndef CRL_MFMA_PREFETCH
#define CRL_MFMA_PREFETCH 0
#endif
static constexpr int kME = 8;                       // envs per group
static constexpr int kMTiles = kME * kPos / 16;     // 50 tiles of 16 conv2 positions
static constexpr int kMWaves = CRL_MFMA_WAVES;                    // wavefronts per workgroup: two per SIMD (one gathers / converts while the other's MFMAs run)
static constexpr int kMThreads = 64 * kMWaves;
static constexpr int kMBuf = kME * CRL_POLICY_STACK * kPlanePad;  // 56 832 bytes per staging buffer
static constexpr int kMLdsRest = (3 * 1600 + 2 * kME * kPos * 3 + 4) * 4;
static constexpr int kMLds = 2 * kMBuf + kMLdsRest;

__device__ inline void group_request_m(uint8_t *shbuf, const uint8_t *__restrict__ ring, int head, const uint8_t *__restrict__ frame,
                                       int64_t frame_stride, int64_t env0, int envs_here, int wave, int lane) {
    for (int s = wave; s < kME * 3; s += kMWaves) {
        const int fe = s / 3, j = s - fe * 3;
        if (fe >= envs_here) continue;
        const int pp = (head + 1 + j) & 3;
        const uint8_t *src = ring + (env0 + fe) * (int64_t)kRingBytes + pp * kPlanePad;
        uint8_t *dst = shbuf + (fe * CRL_POLICY_STACK + pp) * kPlanePad;
#pragma unroll
        for (int half = 0; half < 2; half++) {
            const int c = half * 64 + lane;
            if (c < kPlaneChunks) lds_dma_b128(src + c * 16, lds_addr(dst + half * 1024));
        }
    }
    for (int s = wave; s < kME * 7; s += kMWaves) {
        const int fe = s / 7, q = s - fe * 7;
        if (fe >= envs_here) continue;
        const int d = q * 64 + lane;
        const uint8_t *src = frame + (env0 + fe) * frame_stride;
        uint8_t *dst = shbuf + (fe * CRL_POLICY_STACK + head) * kPlanePad + q * 256;
        if (d < kPlaneWords) lds_dma_b32(src + d * 4, lds_addr(dst));
    }
}

struct PolicyWeightsM {
    const float *w1, *b1, *w2, *b2, *wa, *ba;  // torch layouts: conv1 [16][4][4][4], conv2 [16][16][2][2], actor [3][1600]
};

// BF = true: conv1 on the bf16 matrix instruction at fp32 accuracy.  The inputs are integers 0..255 -- exact in bf16 -- and
// each weight (pre-divided by 255) is the sum of three bf16 terms (8 + 8 + 8 mantissa bits), so the three products per tap
// are exact and v_mfma_f32_16x16x32_bf16 accumulates them in fp32: 6 instructions x 16 cycles per parity class instead
// of 16 x 32, and the patch conversion shrinks from 36 correctly rounded x / 255 to 48 byte -> bf16 conversions.
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
__device__ inline uint32_t pk_bf16(float a, float b) {
    const bf2 v = {(__bf16)a, (__bf16)b};
    return __builtin_bit_cast(uint32_t, v);
}

template <bool BF>
__global__ __launch_bounds__(kMThreads, 1) void pong_policy_mfma_kernel(PolicyWeightsM W, uint8_t *__restrict__ ring, int head,
                                                                  const uint8_t *__restrict__ frame, int64_t frame_stride,
                                                                  int32_t *__restrict__ actions, int64_t action_stride,
                                                                  float *__restrict__ logits_out, int64_t n, unsigned *__restrict__ ticket, int dbg_arg) {
    const int dbg = CRL_ABL(dbg_arg);  // timing ablations / phase stamps: profiling build only
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    uint8_t *sh_buf = smem;                                             // [2][kME][4][kPlanePad]
    float *sh_wa = reinterpret_cast<float *>(smem + 2 * kMBuf);         // [3][1600]
    float *sh_part = sh_wa + 3 * 1600;                                  // [2][kME * 100][3]: a group's partial logits, double-buffered
    unsigned *sh_ticket = reinterpret_cast<unsigned *>(sh_part + 2 * kME * kPos * 3);  // [2]
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int lj = lane & 15, lk = lane >> 4;
    const int64_t ngroups = (n + kME - 1) / kME;

    for (int i = tid; i < 3 * 1600; i += kMThreads) sh_wa[i] = W.wa[i];
    // the wavefront's weights, once: A operands of every MFMA step
    float w1[16], w2[4][4];
    bf8 wA[3][2];  // BF: A[oc = lj][k = 32 i + 8 lk + j], k = ic * 16 + ky * 4 + kx (torch's own order), as three bf16 terms
    if constexpr (BF) {
#pragma unroll
        for (int i = 0; i < 2; i++)
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const float w = W.w1[lj * 64 + 32 * i + 8 * lk + j] / 255.0f;
                const __bf16 h1 = (__bf16)w;
                const float r1 = w - (float)h1;  // exact
                const __bf16 h2 = (__bf16)r1;
                const float r2 = r1 - (float)h2;  // exact
                wA[0][i][j] = h1, wA[1][i][j] = h2, wA[2][i][j] = (__bf16)r2;
            }
#pragma unroll
        for (int s = 0; s < 16; s++) w1[s] = 0.f;
    } else {
#pragma unroll
        for (int s = 0; s < 16; s++) w1[s] = W.w1[lj * 64 + lk * 16 + s];                  // W1[oc = lj][ic = lk][ky = s / 4][kx = s % 4]
    }
#pragma unroll
    for (int c = 0; c < 4; c++)
#pragma unroll
        for (int r = 0; r < 4; r++) w2[c][r] = W.w2[(lj * 16 + 4 * lk + r) * 4 + c];      // W2[oc2 = lj][ic = 4 lk + r][ky0, kx0 = c]
    f4 bias1, bias2;
#pragma unroll
    for (int r = 0; r < 4; r++) bias1[r] = W.b1[4 * lk + r], bias2[r] = W.b2[4 * lk + r];  // D rows = channels 4 lk + r
    float ba0 = W.ba[0], ba1 = W.ba[1], ba2 = W.ba[2];
    // Every load above must have RETURNED before the first LDS-DMA request is issued: the compiler waits for a load at its
    // first use with a vmcnt(N) that counts only the loads it knows of -- inside the tile loop that wait would also drain
    // the next group's LDS-DMA transfers (inline asm, invisible to it) and serialise staging with the convolutions.
    if constexpr (BF) {
#pragma unroll
        for (int t = 0; t < 3; t++)
#pragma unroll
            for (int i = 0; i < 2; i++) {
                u32x4 bits = __builtin_bit_cast(u32x4, wA[t][i]);
                asm volatile("" : "+v"(bits));
                wA[t][i] = __builtin_bit_cast(bf8, bits);
            }
    } else {
#pragma unroll
        for (int s = 0; s < 16; s++) asm volatile("" : "+v"(w1[s]));
    }
#pragma unroll
    for (int c = 0; c < 4; c++)
#pragma unroll
        for (int r = 0; r < 4; r++) asm volatile("" : "+v"(w2[c][r]));
    asm volatile("" : "+v"(bias1), "+v"(bias2), "+v"(ba0), "+v"(ba1), "+v"(ba2));
    __syncthreads();  // sh_wa is staged

    // Three groups are in play: g (being computed), g1 (streaming into the other buffer) and the ticket for the one after
    // (a returning global atomic takes microseconds: it is requested at the top of an iteration and read at its end).
    int64_t g = blockIdx.x, g1 = ngroups, gprev = -1;
    int cur = 0;  // parity of the group being computed: staging buffer, partial buffer; the ticket lands in slot cur ^ 1
    if (g < ngroups) {
        const int64_t env0 = g * kME;
        if (tid == 0) sh_ticket[0] = atomicAdd(ticket, 1u);
        group_request_m(sh_buf, ring, head, frame, frame_stride, env0, (int)((n - env0) < kME ? (n - env0) : kME), wave, lane);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        g1 = (int64_t)gridDim.x + sh_ticket[0];
    }
    // A finished group's 100 partial logits per env and action are summed in a fixed shape (lane j: positions j, j + 32, j + 64,
    // j + 96; then a 32-lane butterfly) by ONE wavefront per env, at the top of the NEXT group's iteration -- beside the other
    // wavefronts' matrix work instead of between two workgroup barriers.
    auto finish_group = [&](int64_t gp, int par) {
        const int64_t e0 = gp * kME;
        const int envs = (int)((n - e0) < kME ? (n - e0) : kME);
        const float *part = sh_part + par * (kME * kPos * 3);
        for (int pe = wave; pe < envs; pe += kMWaves) {
            const int j = lane & 31;
            float s0 = 0.f, s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int m = 0; m < 4; m++) {
                const int pos = j + 32 * m;
                if (pos < kPos) {
                    const float *pp = part + (pe * kPos + pos) * 3;
                    s0 += pp[0], s1 += pp[1], s2 += pp[2];
                }
            }
#pragma unroll
            for (int d = 16; d >= 1; d >>= 1) s0 += __shfl_xor(s0, d), s1 += __shfl_xor(s1, d), s2 += __shfl_xor(s2, d);
            if (lane == 0) {
                const float a0 = ba0 + s0, a1 = ba1 + s1, a2 = ba2 + s2;
                int best = 0;  // argmax, first index wins ties (torch.argmax)
                float bv = a0;
                if (a1 > bv) best = 1, bv = a1;
                if (a2 > bv) best = 2;
                actions[(e0 + pe) * action_stride] = best;
                if (logits_out) {
                    float *lo = logits_out + (e0 + pe) * 3;
                    lo[0] = a0, lo[1] = a1, lo[2] = a2;
                }
            }
        }
    };
    long long tk[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tprev = __builtin_readcyclecounter();
    const long long tstart = tprev;
#define MTICK(K)                                                \
    if (dbg & 4) {                                              \
        const long long now_ = __builtin_readcyclecounter();    \
        tk[K] += now_ - tprev;                                  \
        tprev = now_;                                           \
    }
    while (g < ngroups) {
        const int64_t env0 = g * kME;
        const int envs_here = (int)((n - env0) < kME ? (n - env0) : kME);
        uint8_t *buf = sh_buf + cur * kMBuf;
        float *part_out = sh_part + cur * (kME * kPos * 3);
        if (tid == 0) sh_ticket[cur ^ 1] = atomicAdd(ticket, 1u);
        MTICK(3)
        if (gprev >= 0) finish_group(gprev, cur ^ 1);
        MTICK(4)
        if (g1 < ngroups) {  // the next group streams into the other buffer during the convolutions
            const int64_t e1 = g1 * kME;
            group_request_m(sh_buf + (cur ^ 1) * kMBuf, ring, head, frame, frame_stride, e1, (int)((n - e1) < kME ? (n - e1) : kME), wave, lane);
        }
        MTICK(5)
        // the new frame also replaces plane `head` of the ring in HBM
        for (int i = tid; i < envs_here * kPlaneChunks; i += kMThreads) {
            const int fe = i / kPlaneChunks, c = i - fe * kPlaneChunks;
            reinterpret_cast<uint4 *>(ring + (env0 + fe) * (int64_t)kRingBytes + head * kPlanePad)[c] =
                reinterpret_cast<const uint4 *>(buf + (fe * CRL_POLICY_STACK + head) * kPlanePad)[c];
        }
        MTICK(0)
        // The patch of tile t + kMWaves is gathered and converted WHILE tile t's MFMAs run: the two are independent, so the
        // scheduler threads the vector work between the matrix instructions (a wavefront that converts first and multiplies
        // afterwards leaves the matrix pipe idle 44 % of the time even with a partner on the SIMD).
        auto gather = [&](int t, float (&xo)[6][6]) {
            const int q = 16 * t + lj, e = q / kPos, pos = q - e * kPos, y2 = pos / 10, x2 = pos - y2 * 10;
            // this lane's 6 x 6 patch of plane ic = lk, as x / 255 (correctly rounded: Markstein).  Two ALIGNED dwords per 6-byte
            // row piece: (4 y2 + r) * 42 + 4 x2 is a multiple of 4 for even r and 2 past one for odd r -- known at compile time.
            const float rcp = 1.0f / 255.0f;
            const uint8_t *base = buf + (e * CRL_POLICY_STACK + ((head + 1 + lk) & 3)) * kPlanePad + (4 * y2) * kDim + 4 * x2;
#pragma unroll
            for (int r = 0; r < 6; r++) {
                const uint32_t *p32 = reinterpret_cast<const uint32_t *>(base + r * kDim - 2 * (r & 1));
                uint32_t w0 = p32[0], w1_ = p32[1];
                if (r & 1) w0 = (w0 >> 16) | (w1_ << 16), w1_ >>= 16;
                const float b[6] = {(float)(w0 & 255u), (float)((w0 >> 8) & 255u), (float)((w0 >> 16) & 255u), (float)(w0 >> 24),
                                    (float)(w1_ & 255u), (float)((w1_ >> 8) & 255u)};
#pragma unroll
                for (int k = 0; k < 6; k++) {
                    const float qv = b[k] * rcp;
                    const float rem = __builtin_fmaf(qv, -255.0f, b[k]);
                    xo[r][k] = __builtin_fmaf(rem, rcp, qv);
                }
            }
        };
        // BF: lane (position lj, k-block lk) needs, of planes lk >> 1 and 2 + (lk >> 1), rows 4 y2 + 2 (lk & 1) + 0..3 and the six
        // columns from 4 x2 on: the B operand of (class (ky0, kx0), K half i) is rows 2 ky0, 2 ky0 + 1 of that window, columns
        // 2 kx0 .. 2 kx0 + 3, of plane 2 i + (lk >> 1) -- eight bf16 in k order.  Same aligned-dword trick as above.
        auto gather_bf = [&](int t, bf8 (&bx)[4][2]) {
            const int q = 16 * t + lj, e = q / kPos, pos = q - e * kPos, y2 = pos / 10, x2 = pos - y2 * 10;
#pragma unroll
            for (int i = 0; i < 2; i++) {
                const int plane = 2 * i + (lk >> 1);
                const uint8_t *base = buf + (e * CRL_POLICY_STACK + ((head + 1 + plane) & 3)) * kPlanePad + (4 * y2 + 2 * (lk & 1)) * kDim + 4 * x2;
                uint32_t pk[4][3];  // per window row: columns (0,1) (2,3) (4,5) as bf16 pairs
#pragma unroll
                for (int rr = 0; rr < 4; rr++) {
                    const uint32_t *p32 = reinterpret_cast<const uint32_t *>(base + rr * kDim - 2 * (rr & 1));
                    uint32_t w0 = p32[0], w1_ = p32[1];
                    if (rr & 1) w0 = (w0 >> 16) | (w1_ << 16), w1_ >>= 16;
                    pk[rr][0] = pk_bf16((float)(w0 & 255u), (float)((w0 >> 8) & 255u));
                    pk[rr][1] = pk_bf16((float)((w0 >> 16) & 255u), (float)(w0 >> 24));
                    pk[rr][2] = pk_bf16((float)(w1_ & 255u), (float)((w1_ >> 8) & 255u));
                }
#pragma unroll
                for (int c = 0; c < 4; c++) {
                    const int ky0 = c >> 1, kx0 = c & 1;
                    const u32x4 v = {pk[2 * ky0][kx0], pk[2 * ky0][kx0 + 1], pk[2 * ky0 + 1][kx0], pk[2 * ky0 + 1][kx0 + 1]};
                    bx[c][i] = __builtin_bit_cast(bf8, v);
                }
            }
        };
        float xnext[6][6];
        if (!CRL_MFMA_PREFETCH || BF) {
        } else if (dbg & 1) {  // ablation: no gather
#pragma unroll
            for (int r = 0; r < 6; r++)
#pragma unroll
                for (int k = 0; k < 6; k++) xnext[r][k] = 0.25f * (float)(r + k + lk);
        } else if (wave < kMTiles) gather(wave, xnext);
        for (int t = wave; t < kMTiles; t += kMWaves) {
            const int q = 16 * t + lj, e = q / kPos, pos = q - e * kPos;
            float xin[6][6];
            bf8 bx[4][2];
            if constexpr (BF) {
                if (dbg & 1) {  // ablation: no gather
#pragma unroll
                    for (int c = 0; c < 4; c++)
#pragma unroll
                        for (int i = 0; i < 2; i++) {
                            const u32x4 v = {0x3f803f80u + (unsigned)lk, 0x40004000u, 0x3f803f80u, 0x40004000u + (unsigned)c};
                            bx[c][i] = __builtin_bit_cast(bf8, v);
                        }
                } else {
                    gather_bf(t, bx);
                }
            } else if (CRL_MFMA_PREFETCH) {
#pragma unroll
                for (int r = 0; r < 6; r++)
#pragma unroll
                    for (int k = 0; k < 6; k++) xin[r][k] = xnext[r][k];
                if (t + kMWaves < kMTiles && !(dbg & 1)) gather(t + kMWaves, xnext);
            } else {
                gather(t, xin);
            }
            // ---- conv1: four parity classes, 16 MFMAs each (independent accumulator chains)
            f4 d1[4] = {bias1, bias1, bias1, bias1};
            if constexpr (BF) {
#pragma unroll
                for (int tm = 2; tm >= 0; tm--)  // smallest weight term first
#pragma unroll
                    for (int i = 0; i < 2; i++)
#pragma unroll
                        for (int c = 0; c < 4; c++) d1[c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wA[tm][i], bx[c][i], d1[c], 0, 0, 0);
            } else {
#pragma unroll
                for (int s = 0; s < 16; s++)
#pragma unroll
                    for (int c = 0; c < 4; c++)
                        d1[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(w1[s], xin[2 * (c >> 1) + (s >> 2)][2 * (c & 1) + (s & 3)], d1[c], 0, 0, 0);
            }
            // ---- conv2: the accumulators are the B operands as they stand
            f4 d2a = bias2, d2b = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int c = 0; c < 4; c++)
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const float hval = fmaxf(d1[c][r], 0.f);
                    if ((c * 4 + r) & 1) d2b = __builtin_amdgcn_mfma_f32_16x16x4f32(w2[c][r], hval, d2b, 0, 0, 0);
                    else d2a = __builtin_amdgcn_mfma_f32_16x16x4f32(w2[c][r], hval, d2a, 0, 0, 0);
                }
            if (dbg & 2) {  // ablation: no actor / cross-lane sums
                if (lk == 0) part_out[q * 3] = d2a[0] + d2b[1], part_out[q * 3 + 1] = 0.f, part_out[q * 3 + 2] = 0.f;
                continue;
            }
            // ---- actor: lane (position lj, channels 4 lk + r)
            float l0 = 0.f, l1 = 0.f, l2 = 0.f;
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const float f = fmaxf(d2a[r] + d2b[r], 0.f);
                const int wi = (4 * lk + r) * kPos + pos;
                l0 = __builtin_fmaf(sh_wa[wi], f, l0);
                l1 = __builtin_fmaf(sh_wa[1600 + wi], f, l1);
                l2 = __builtin_fmaf(sh_wa[3200 + wi], f, l2);
            }
            // the four channel groups of a position sit 16 lanes apart: (g0 + g1) + (g2 + g3), a fixed order
            l0 += __shfl_xor(l0, 16), l1 += __shfl_xor(l1, 16), l2 += __shfl_xor(l2, 16);
            l0 += __shfl_xor(l0, 32), l1 += __shfl_xor(l1, 32), l2 += __shfl_xor(l2, 32);
            if (lk == 0) part_out[q * 3 + 0] = l0, part_out[q * 3 + 1] = l1, part_out[q * 3 + 2] = l2;
        }
        MTICK(1)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wavefront's share of the next group has landed (and the ticket is back)
        __syncthreads();                                    // ... everybody's has, and every tile of this group is in part_out
        MTICK(2)
        gprev = g;
        g = g1;
        g1 = (int64_t)gridDim.x + sh_ticket[cur ^ 1];
        cur ^= 1;
    }
    if (gprev >= 0) finish_group(gprev, cur ^ 1);
    if ((dbg & 4) && logits_out && lane == 0 && blockIdx.x < 64) {  // profiling: cycles per phase of every wavefront of the first workgroups
        float *o = logits_out + (blockIdx.x * kMWaves + wave) * 8;
        for (int k = 0; k < 7; k++) o[k] = (float)tk[k];
        o[7] = (float)(__builtin_readcyclecounter() - tstart);
    }
#undef MTICK
}

// ring <-> logical order (tests, checkpoints): plane j of the model's stack is ring plane (head + j) & 3
__global__ void pong_policy_copy_stack_kernel(uint8_t *__restrict__ ring, uint8_t *__restrict__ ext, int head, int64_t words, int to_ring) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= words) return;
    const int64_t env = i / (CRL_POLICY_STACK * kPlaneWords);
    const int r = (int)(i - env * (CRL_POLICY_STACK * kPlaneWords));
    const int j = r / kPlaneWords, d = r - j * kPlaneWords;
    uint32_t *rp = reinterpret_cast<uint32_t *>(ring) + env * (kRingBytes / 4) + ((head + j) & 3) * (kPlanePad / 4) + d;
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
    int cus = 256;
    int head = 0;  // ring plane holding the OLDEST frame (the next one to be replaced)
    float *weights = nullptr;
    uint8_t *ring = nullptr;
    unsigned *ticket = nullptr;  // next group to hand out (reset before every launch)
    PolicyWeights W{};
    float *raw = nullptr;        // the checkpoint tensors in torch layout (MFMA kernel): w1 1024 | b1 16 | w2 1024 | b2 16 | wa 4800 | ba 3
    PolicyWeightsM WM{};
    PolicyFull *full = nullptr;  // crl_policy_create_full: ActorCritic instead of LightActorCritic (pong_policy_full.hip)
};

extern "C" {

int crl_policy_create(int32_t device, int64_t num_envs, const float *conv1_w, const float *conv1_b, const float *conv2_w,
                      const float *conv2_b, const float *actor_w, const float *actor_b, crl_policy **out) {
    crl_fail_no_ctx();
    if (!out || num_envs <= 0 || !conv1_w || !conv1_b || !conv2_w || !conv2_b || !actor_w || !actor_b)
        return crl_fail(CRL_EINVAL, "crl_policy_create: bad arguments");
    HIP_TRY(hipSetDevice(device));
    crl_policy *p = new crl_policy();
    p->device = device, p->n = num_envs;
    if (hipDeviceGetAttribute(&p->cus, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess || p->cus <= 0) p->cus = 256;
    // one blob: stream 2048 + 32 pad | b1 16 | b2 16 | wa 4800 | ba 3 (+ pad)
    std::vector<float> blob(2080 + 16 + 16 + 4800 + 4, 0.f);
    float *st = blob.data(), *b1 = st + 2080, *b2 = b1 + 16, *wa = b2 + 16, *ba = wa + 4800;
    for (int oc = 0; oc < 16; oc++)  // torch conv1 [oc][tap] -> [oc / 2][tap][oc & 1] at the head of block oc / 2
        for (int k = 0; k < 64; k++) st[(oc >> 1) * 256 + k * 2 + (oc & 1)] = conv1_w[oc * 64 + k];
    for (int oc = 0; oc < 16; oc++)  // torch conv2 [oc][ic][ky][kx] -> block ic / 2: [oc / 2][ic & 1][k][oc & 1]
        for (int ic = 0; ic < 16; ic++)
            for (int k = 0; k < 4; k++)
                st[(ic >> 1) * 256 + 128 + (((oc >> 1) * 2 + (ic & 1)) * 4 + k) * 2 + (oc & 1)] = conv2_w[(oc * 16 + ic) * 4 + k];
    memcpy(b1, conv1_b, 16 * sizeof(float));
    memcpy(b2, conv2_b, 16 * sizeof(float));
    memcpy(wa, actor_w, 4800 * sizeof(float));
    memcpy(ba, actor_b, 3 * sizeof(float));
    hipError_t e = hipMalloc(&p->weights, blob.size() * sizeof(float));
    if (e == hipSuccess) e = hipMemcpy(p->weights, blob.data(), blob.size() * sizeof(float), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMalloc(&p->ticket, 128);  // [0]: packed-FMA kernel's counter, [16]: MFMA kernel's (hybrid mode)
    if (e == hipSuccess) e = hipMalloc(&p->ring, (size_t)num_envs * kRingBytes);
    if (e == hipSuccess) e = hipMemset(p->ring, 0, (size_t)num_envs * kRingBytes);
    if (e != hipSuccess) {
        crl_policy_destroy(p);
        return crl_fail(e == hipErrorOutOfMemory ? CRL_ENOMEM : CRL_EHIP, "crl_policy_create: %s", hipGetErrorString(e));
    }
    {
        std::vector<float> raw(1024 + 16 + 1024 + 16 + 4800 + 4, 0.f);
        memcpy(raw.data(), conv1_w, 1024 * 4), memcpy(raw.data() + 1024, conv1_b, 16 * 4), memcpy(raw.data() + 1040, conv2_w, 1024 * 4);
        memcpy(raw.data() + 2064, conv2_b, 16 * 4), memcpy(raw.data() + 2080, actor_w, 4800 * 4), memcpy(raw.data() + 6880, actor_b, 3 * 4);
        e = hipMalloc(&p->raw, raw.size() * sizeof(float));
        if (e == hipSuccess) e = hipMemcpy(p->raw, raw.data(), raw.size() * sizeof(float), hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void *>(pong_policy_mfma_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, kMLds);
        if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void *>(pong_policy_mfma_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, kMLds);
        if (e != hipSuccess) {
            crl_policy_destroy(p);
            return crl_fail(CRL_EHIP, "crl_policy_create (mfma weights): %s", hipGetErrorString(e));
        }
        p->WM.w1 = p->raw, p->WM.b1 = p->raw + 1024, p->WM.w2 = p->raw + 1040, p->WM.b2 = p->raw + 2064, p->WM.wa = p->raw + 2080, p->WM.ba = p->raw + 6880;
    }
    const float *base = p->weights;  // hipMalloc: 256-byte aligned, so every 64-byte batch is aligned
    p->W.stream = base, p->W.b1 = base + 2080, p->W.b2 = reinterpret_cast<const f2 *>(base + 2096);
    p->W.wa = base + 2112, p->W.ba = base + 6912;
    *out = p;
    return CRL_OK;
}

int crl_policy_create_full(int32_t device, int64_t num_envs, const float *conv1_w, const float *conv1_b, const float *conv2_w,
                           const float *conv2_b, const float *conv3_w, const float *conv3_b, const float *actor_w, const float *actor_b,
                           crl_policy **out) {
    crl_fail_no_ctx();
    if (!out || num_envs <= 0 || !conv1_w || !conv1_b || !conv2_w || !conv2_b || !conv3_w || !conv3_b || !actor_w || !actor_b)
        return crl_fail(CRL_EINVAL, "crl_policy_create_full: bad arguments");
    HIP_TRY(hipSetDevice(device));
    crl_policy *p = new crl_policy();
    p->device = device, p->n = num_envs;
    hipError_t e = hipMalloc(&p->ring, (size_t)num_envs * kRingBytes);
    if (e == hipSuccess) e = hipMemset(p->ring, 0, (size_t)num_envs * kRingBytes);
    if (e == hipSuccess) e = policy_full_create(&p->full, num_envs, conv1_w, conv1_b, conv2_w, conv2_b, conv3_w, conv3_b, actor_w, actor_b);
    if (e != hipSuccess) {
        crl_policy_destroy(p);
        return crl_fail(e == hipErrorOutOfMemory ? CRL_ENOMEM : CRL_EHIP, "crl_policy_create_full: %s", hipGetErrorString(e));
    }
    *out = p;
    return CRL_OK;
}

void crl_policy_destroy(crl_policy *p) {
    if (!p) return;
    (void)hipSetDevice(p->device);
    policy_full_destroy(p->full);
    if (p->weights) (void)hipFree(p->weights);
    if (p->raw) (void)hipFree(p->raw);
    if (p->ring) (void)hipFree(p->ring);
    if (p->ticket) (void)hipFree(p->ticket);
    delete p;
}

int crl_policy_reset(crl_policy *p, void *stream) {
    crl_fail_no_ctx();
    if (!p) return crl_fail(CRL_EINVAL, "crl_policy_reset: null policy");
    HIP_TRY(hipMemsetAsync(p->ring, 0, (size_t)p->n * kRingBytes, (hipStream_t)stream));
    p->head = 0;
    return CRL_OK;
}

int crl_policy_act(crl_policy *p, const uint8_t *frame_dev, int64_t frame_stride, int32_t *actions_dev, int64_t action_stride,
                   float *logits_dev, void *stream) {
    crl_fail_no_ctx();
    if (!p || !frame_dev || !actions_dev) return crl_fail(CRL_EINVAL, "crl_policy_act: null argument");
    if (frame_stride < kPlane || (frame_stride & 3) || ((uintptr_t)frame_dev & 3) || action_stride < 1)
        return crl_fail(CRL_EINVAL, "crl_policy_act: frame_stride must be a multiple of 4 and >= 1764, frames 4-byte aligned");
    hipStream_t main_st = (hipStream_t)stream;
    if (p->full) {
        HIP_TRY(policy_full_act(p->full, p->ring, p->head, p->n, frame_dev, frame_stride, actions_dev, action_stride, logits_dev, main_st));
        p->head = (p->head + 1) & 3;
        return CRL_OK;
    }
    HIP_TRY(hipMemsetAsync(p->ticket, 0, sizeof(unsigned), main_st));
    // The matrix-pipe kernel, conv1 as three exact bf16 products per tap (430 us at 65 536 envs).  Profiling build only
    // (CRL_POLICY_MFMA): 1 = the same kernel with conv1 on the fp32 matrix instruction (757 us), 0 = the packed-FMA kernel of
    // round 1 (725-805 us); CRL_POLICY_DEBUG / CRL_POLICY_MFMA_DEBUG skip phases (wrong outputs).
    static const int use_mfma = CRL_ABL(getenv("CRL_POLICY_MFMA") != nullptr) ? atoi(getenv("CRL_POLICY_MFMA")) : 3;
    static const int dbg = CRL_ABL(getenv("CRL_POLICY_DEBUG") ? atoi(getenv("CRL_POLICY_DEBUG")) : 0);
    static const int mdbg = CRL_ABL(getenv("CRL_POLICY_MFMA_DEBUG") ? atoi(getenv("CRL_POLICY_MFMA_DEBUG")) : 0);
    if ((use_mfma == 1 || use_mfma == 3) && !dbg) {
        const int64_t mgroups = (p->n + kME - 1) / kME;
        const unsigned mgrid = (unsigned)(mgroups < p->cus ? mgroups : p->cus);  // persistent: one workgroup per CU
#ifdef CRL_ABLATION
        if (use_mfma != 3)
            hipLaunchKernelGGL(pong_policy_mfma_kernel<false>, dim3(mgrid), dim3(kMThreads), kMLds, main_st, p->WM, p->ring, p->head, frame_dev,
                               frame_stride, actions_dev, action_stride, logits_dev, p->n, p->ticket, mdbg);
        else
#endif
            hipLaunchKernelGGL(pong_policy_mfma_kernel<true>, dim3(mgrid), dim3(kMThreads), kMLds, main_st, p->WM, p->ring, p->head, frame_dev,
                               frame_stride, actions_dev, action_stride, logits_dev, p->n, p->ticket, mdbg);
        HIP_TRY(hipGetLastError());
        p->head = (p->head + 1) & 3;
        return CRL_OK;
    }
#ifdef CRL_ABLATION
    static const int phase = getenv("CRL_POLICY_PHASE") ? atoi(getenv("CRL_POLICY_PHASE")) : 0;  // x 8 128 cycles
    const int64_t groups = (p->n + kEnvsPerWg - 1) / kEnvsPerWg;
    static const int per_cu = getenv("CRL_POLICY_WGS") ? atoi(getenv("CRL_POLICY_WGS")) : 2;  // tuning experiments only
    const unsigned grid = (unsigned)(groups < per_cu * p->cus ? groups : per_cu * p->cus);  // persistent: two workgroups per CU
    if (dbg == 4)
        hipLaunchKernelGGL(pong_policy_light_kernel<2>, dim3(grid), dim3(kPolicyThreads), 0, (hipStream_t)stream, p->W, p->ring, p->head,
                           frame_dev, frame_stride, actions_dev, action_stride, logits_dev, p->n, dbg, phase, p->ticket);
    else if (dbg)
        hipLaunchKernelGGL(pong_policy_light_kernel<1>, dim3(grid), dim3(kPolicyThreads), 0, (hipStream_t)stream, p->W, p->ring, p->head,
                           frame_dev, frame_stride, actions_dev, action_stride, logits_dev, p->n, dbg, phase, p->ticket);
    else
        hipLaunchKernelGGL(pong_policy_light_kernel<0>, dim3(grid), dim3(kPolicyThreads), 0, (hipStream_t)stream, p->W, p->ring, p->head,
                           frame_dev, frame_stride, actions_dev, action_stride, logits_dev, p->n, 0, phase, p->ticket);
    HIP_TRY(hipGetLastError());
    p->head = (p->head + 1) & 3;
    return CRL_OK;
#else
    return crl_fail(CRL_ESTATE, "crl_policy_act: no kernel selected");
#endif
}

static int copy_stack(crl_policy *p, uint8_t *ext, int to_ring, void *stream) {
    crl_fail_no_ctx();
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
