// pong_raster_raw.hip -- raw cPongDouble observation writer: (N, 2, 210, 160, 3) uint8.
//
// Restates PongGame.draw + Scoreboard.draw + _surface_to_img + the second agent's
// mirrored view (reference pong/base_pong_env.py:259-266, 72-74, 149-155) as an
// ANALYTIC pixel function of the 8-byte frame descriptor -- nothing is read back,
// nothing is copied between views.
//
// Roofline: pure HBM store stream, 201 600 B per env-step, no reuse.  The env's two frames
// are 12 600 16-byte chunks; a chunk's 16 bytes are built in registers:
//   rows <34  : white, or (ink rows only) a 16-byte load from the RGB-expanded score
//               band of this (score_l, score_r) -- L2/MALL resident, ~8 MB total;
//   rows 34-193: 16-bit coverage mask of ball/bat rectangles -> 4 dwords of 0x00/0xFF;
//   rows >=194: white.
// Agent 1's view is rows >= 25 mirrored; since every pixel is achromatic (R=G=B) a
// mirrored chunk is the byte-reversed chunk (29 - c) of the unmirrored row.
// Three work splits: pong_raster_raw_sweep_kernel (production: address-linear, a thread's four chunks a whole grid
// apart), pong_raster_raw_linear_kernel (address-linear, a workgroup's chunks contiguous; CRL_RAW_SWEEP=0) and the
// original one workgroup per env (thread t writes chunks t, t+256, ...; CRL_RAW_SWEEP=0 CRL_RAW_LINEAR=0), kept for A/B.
#include <stdlib.h>

#include "crl_internal.h"
#include "pong_device.h"

namespace crl {

static constexpr int kRowBytes = CRL_PONG_W * 3;            // 480
static constexpr int kRowChunks = kRowBytes / 16;           // 30
static constexpr int kFrameChunks = CRL_PONG_H * kRowChunks;  // 6300

__device__ inline uint32_t nibble_to_bytes(uint32_t nib) {
    // bit k of nib -> byte k = 0xFF
    return ((nib * 0x00204081u) & 0x01010101u) * 0xFFu;
}

__device__ inline uint32_t span_bits(int a, int b) {
    // bits [a, b) of a 16-bit chunk mask, a/b in chunk-local byte coordinates (any int)
    a = max(a, 0), b = min(b, 16);
    return a < b ? (((1u << b) - 1u) & ~((1u << a) - 1u)) : 0u;
}

__device__ inline uint32_t bswap32(uint32_t v) { return __builtin_bswap32(v); }

// The 16 bytes of chunk q (0 .. views * kFrameChunks - 1) of one env's frames: THE pixel function of this file, shared by
// the three work splits below.  dbg: 1 = constants only, 2 = no score-band loads (profiling).
__device__ __forceinline__ uint4 raw_chunk(const Frame &f, int q, const uint4 *__restrict__ atlas_rgb, int ink_row0, int ink_row1, int dbg) {
    const bool blank = f.sl == 255;
    const uint32_t bg = blank ? 0u : 0xFFFFFFFFu;
    const int view = q >= kFrameChunks;
    const int c = q - view * kFrameChunks;
    const int row = c / kRowChunks;
    const int cc = c - row * kRowChunks;
    const bool mirror = view && row >= CRL_PONG_MIRROR_ROW;
    const int sc = mirror ? (kRowChunks - 1 - cc) : cc;  // source chunk in the unmirrored row
    uint4 v = make_uint4(bg, bg, bg, bg);
    if (!blank && !(dbg & 1)) {
        if (row < CRL_PONG_TOP) {
            if (row >= ink_row0 && row < ink_row1 && !(dbg & 2)) v = atlas_rgb[(int64_t)((f.sl * 22 + f.sr) * CRL_PONG_TOP + row) * kRowChunks + sc];
        } else if (row < CRL_PONG_BOTTOM) {
            const int lo = sc * 16;
            uint32_t m = 0;
            if (row >= f.y && row < f.y + CRL_PONG_BALL) m |= span_bits(3 * f.x - lo, 3 * (f.x + CRL_PONG_BALL) - lo);
            if (row >= f.bl && row < f.bl + CRL_PONG_BAT_H)
                m |= span_bits(3 * CRL_PONG_BATL_X - lo, 3 * (CRL_PONG_BATL_X + CRL_PONG_BAT_W) - lo);
            if (row >= f.br && row < f.br + CRL_PONG_BAT_H)
                m |= span_bits(3 * CRL_PONG_BATR_X - lo, 3 * (CRL_PONG_BATR_X + CRL_PONG_BAT_W) - lo);
            v = make_uint4(nibble_to_bytes(m & 15u), nibble_to_bytes((m >> 4) & 15u), nibble_to_bytes((m >> 8) & 15u),
                           nibble_to_bytes((m >> 12) & 15u));
        }
        if (mirror) v = make_uint4(bswap32(v.w), bswap32(v.z), bswap32(v.y), bswap32(v.x));
    }
    return v;
}

// of two neighbouring envs' descriptors, the one chunk index q (relative to the first) falls into
__device__ __forceinline__ Frame pick_frame(const Frame &f0, const Frame &f1, bool second) {
    Frame f;
    f.x = second ? f1.x : f0.x, f.y = second ? f1.y : f0.y, f.bl = second ? f1.bl : f0.bl, f.br = second ? f1.br : f0.br;
    f.sl = second ? f1.sl : f0.sl, f.sr = second ? f1.sr : f0.sr;
    return f;
}

#ifdef CRL_ABLATION  // superseded writers (one workgroup per env; workgroup-contiguous): profiling build only, CRL_RAW_SWEEP=0
__global__ __launch_bounds__(256) void pong_raster_raw_kernel(const uint64_t *__restrict__ frames,
                                                              const uint4 *__restrict__ atlas_rgb, int ink_row0,
                                                              int ink_row1, uint4 *__restrict__ obs, int views, int dbg) {
    const int64_t env = blockIdx.x;
    const uint64_t packed = frames[env];  // wave-uniform -> scalar load
    const Frame f = unpack_frame(packed);
    uint4 *__restrict__ out = obs + env * (int64_t)(views * kFrameChunks);
    for (int q = threadIdx.x; q < views * kFrameChunks; q += 256) out[q] = raw_chunk(f, q, atlas_rgb, ink_row0, ink_row1, dbg);
}

// Address-linear kernel, workgroup-contiguous (production until the sweep variant below): workgroup b writes chunks [b * 512, (b + 1) * 512) of
// the WHOLE output tensor -- two 16-byte chunks per thread, 8 KiB per workgroup -- whichever envs
// they belong to (at most two).  Measured on MI355X at 65 536 envs (13.2 GB per launch):
//   one workgroup per env, 49 chunks per thread            2 250-2 300 us  (5.8 TB/s)
//   address-linear, 1 / 2 / 4 / 8 chunks per thread    2 550 / 2 030 / 2 120 / 2 180 us
//   torch.Tensor.fill_ of the same bytes (ceiling)           1 915 us      (6.9 TB/s)
// With one workgroup per env the ~2 000 resident workgroups write 2 000 separate streams 201 600 B
// apart, which DRAM sees as that many open pages; here the resident workgroups cover one moving
// 16 MB window of the tensor, like a plain fill.  One chunk per thread is bound by wave launch
// plus the frame-descriptor load in front of every wave; more than two widen the window again.
// The pixel arithmetic itself is off the critical path (removing it changes nothing).
template <int ITERS, int THREADS, int VIEWS>
__global__ __launch_bounds__(THREADS) void pong_raster_raw_linear_kernel(const uint64_t *__restrict__ frames,
                                                                         const uint4 *__restrict__ atlas_rgb, int ink_row0,
                                                                         int ink_row1, uint4 *__restrict__ obs, int views, int64_t n,
                                                                         int dbg) {
    constexpr int per_env = VIEWS * kFrameChunks;
    const int64_t g0 = (int64_t)blockIdx.x * (THREADS * ITERS);
    const int64_t total = n * per_env;
    const int64_t e0 = g0 / per_env;  // first env of this span; the span is shorter than one env
    const int64_t e1 = e0 + 1 < n ? e0 + 1 : e0;
    const Frame f0 = unpack_frame(frames[e0]), f1 = unpack_frame(frames[e1]);  // uniform -> scalar loads
    const int q0 = (int)(g0 - e0 * per_env);
#pragma unroll
    for (int i = 0; i < ITERS; i++) {
        const int64_t g = g0 + i * THREADS + threadIdx.x;
        if (g >= total) break;
        int q = q0 + i * THREADS + (int)threadIdx.x;
        const bool second = q >= per_env;
        q -= second ? per_env : 0;
        obs[g] = raw_chunk(pick_frame(f0, f1, second), q, atlas_rgb, ink_row0, ink_row1, dbg);
    }
}

#endif  // CRL_ABLATION

// Sweep variant of the address-linear kernel: thread t of workgroup b writes chunks b * 256 + t + i * (gridDim.x * 256),
// i = 0 .. ITERS-1 -- every "round" i of the whole chip is one dense linear sweep over 1/ITERS of the tensor.  Pure-store
// probes (tools/store_order_probe.hip): a wavefront that walks through a private contiguous span makes the chip write a
// comb (one tooth per resident wavefront) and loses 8-20 % of the fill rate; stores a whole grid apart do not, however
// many a wavefront issues.  Per round a workgroup's 4 KB block touches at most two envs (two scalar descriptor loads).
template <int ITERS, int VIEWS>
__global__ __launch_bounds__(256) void pong_raster_raw_sweep_kernel(const uint64_t *__restrict__ frames, const uint4 *__restrict__ atlas_rgb,
                                                                    int ink_row0, int ink_row1, uint4 *__restrict__ obs, int64_t n, int dbg) {
    constexpr int per_env = VIEWS * kFrameChunks;
    const int64_t total = n * per_env;
    const int64_t stride = (int64_t)gridDim.x * 256;
    Frame f0[ITERS], f1[ITERS];
    int q0[ITERS];
#pragma unroll
    for (int i = 0; i < ITERS; i++) {
        int64_t g0 = (int64_t)blockIdx.x * 256 + i * stride;
        g0 = g0 < total ? g0 : total - 1;
        const int64_t e0 = g0 / per_env;
        const int64_t e1 = e0 + 1 < n ? e0 + 1 : e0;
        f0[i] = unpack_frame(frames[e0]), f1[i] = unpack_frame(frames[e1]);  // uniform -> scalar loads, all issued up front
        q0[i] = (int)(g0 - e0 * per_env);
    }
#pragma unroll
    for (int i = 0; i < ITERS; i++) {
        const int64_t g = (int64_t)blockIdx.x * 256 + i * stride + threadIdx.x;
        if (g >= total) break;
        int q = q0[i] + (int)threadIdx.x;
        const bool second = q >= per_env;
        q -= second ? per_env : 0;
        obs[g] = raw_chunk(pick_frame(f0[i], f1[i], second), q, atlas_rgb, ink_row0, ink_row1, dbg);
    }
}

void launch_pong_raster_raw(const uint64_t *frames, int64_t n, const uint8_t *atlas_rgb, int ink_row0, int ink_row1,
                            uint8_t *obs, int views, hipStream_t st) {
    if (n <= 0) return;
    // (profiling build only: CRL_RAW_DEBUG bit 1 = constant chunks, i.e. WRONG pixels, to size the pixel arithmetic; CRL_RAW_SWEEP /
    // CRL_RAW_LINEAR select the superseded writers)
    static const int dbg = CRL_ABL(getenv("CRL_RAW_DEBUG") ? atoi(getenv("CRL_RAW_DEBUG")) : 0);
    static const int sweep = CRL_ABL(getenv("CRL_RAW_SWEEP") != nullptr) ? atoi(getenv("CRL_RAW_SWEEP")) : 4;
    if (views != 1 && views != 2) return;  // (crl_create admits nothing else)
    if (sweep > 0) {
        const int64_t total = n * views * kFrameChunks;
        const uint4 *at = reinterpret_cast<const uint4 *>(atlas_rgb);
        uint4 *ob = reinterpret_cast<uint4 *>(obs);
#define CRL_LAUNCH_SWEEP(I, V)                                                                                              \
    hipLaunchKernelGGL((pong_raster_raw_sweep_kernel<I, V>), dim3((unsigned)((total + 256 * I - 1) / (256 * I))), dim3(256), 0, st, frames, \
                       at, ink_row0, ink_row1, ob, n, dbg)
        if (views == 2) {
#ifdef CRL_ABLATION
            if (sweep <= 2) CRL_LAUNCH_SWEEP(2, 2);
            else if (sweep > 4) CRL_LAUNCH_SWEEP(8, 2);
            else
#endif
                CRL_LAUNCH_SWEEP(4, 2);
        } else {
            CRL_LAUNCH_SWEEP(4, 1);
        }
#undef CRL_LAUNCH_SWEEP
        return;
    }
#ifdef CRL_ABLATION
    static const int lin = getenv("CRL_RAW_LINEAR") ? atoi(getenv("CRL_RAW_LINEAR")) : 2;
    if (lin > 0) {
        const int64_t total = n * views * kFrameChunks;
        const uint4 *at = reinterpret_cast<const uint4 *>(atlas_rgb);
        uint4 *ob = reinterpret_cast<uint4 *>(obs);
#define CRL_LAUNCH_LIN(I, V)                                                                                         \
    hipLaunchKernelGGL((pong_raster_raw_linear_kernel<I, 256, V>), dim3((unsigned)((total + 256 * I - 1) / (256 * I))), dim3(256), 0, \
                       st, frames, at, ink_row0, ink_row1, ob, views, n, dbg)
        if (views == 2) {
            if (lin <= 1) CRL_LAUNCH_LIN(1, 2);
            else if (lin <= 2) CRL_LAUNCH_LIN(2, 2);
            else if (lin <= 4) CRL_LAUNCH_LIN(4, 2);
            else CRL_LAUNCH_LIN(8, 2);
        } else {
            CRL_LAUNCH_LIN(2, 1);
        }
#undef CRL_LAUNCH_LIN
        return;
    }
    hipLaunchKernelGGL(pong_raster_raw_kernel, dim3((unsigned)n), dim3(256), 0, st, frames,
                       reinterpret_cast<const uint4 *>(atlas_rgb), ink_row0, ink_row1, reinterpret_cast<uint4 *>(obs), views, dbg);
#endif
}

}  // namespace crl
