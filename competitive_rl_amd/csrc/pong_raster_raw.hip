// pong_raster_raw.hip -- raw cPongDouble observation writer: (N, 2, 210, 160, 3) uint8.
//
// Restates PongGame.draw + Scoreboard.draw + _surface_to_img + the second agent's
// mirrored view (reference pong/base_pong_env.py:259-266, 72-74, 149-155) as an
// ANALYTIC pixel function of the 8-byte frame descriptor -- nothing is read back,
// nothing is copied between views.
//
// Roofline: pure HBM store stream, 201 600 B per env-step, no reuse.  Work split:
// one 256-thread workgroup per env; the env's two frames are 12 600 16-byte chunks;
// thread t writes chunks t, t+256, ... so every wave-level store is one contiguous
// 1 KiB global_store_dwordx4.  A chunk's 16 bytes are built in registers:
//   rows <34  : white, or (ink rows only) a 16-byte load from the RGB-expanded score
//               band of this (score_l, score_r) -- L2/MALL resident, ~8 MB total;
//   rows 34-193: 16-bit coverage mask of ball/bat rectangles -> 4 dwords of 0x00/0xFF;
//   rows >=194: white.
// Agent 1's view is rows >= 25 mirrored; since every pixel is achromatic (R=G=B) a
// mirrored chunk is the byte-reversed chunk (29 - c) of the unmirrored row.
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

__global__ __launch_bounds__(256) void pong_raster_raw_kernel(const uint64_t *__restrict__ frames,
                                                              const uint4 *__restrict__ atlas_rgb, int ink_row0,
                                                              int ink_row1, uint4 *__restrict__ obs, int views) {
    const int64_t env = blockIdx.x;
    const uint64_t packed = frames[env];  // wave-uniform -> scalar load
    const Frame f = unpack_frame(packed);
    uint4 *__restrict__ out = obs + env * (int64_t)(views * kFrameChunks);
    const bool blank = f.sl == 255;
    const uint4 *__restrict__ band = atlas_rgb + (int64_t)((blank ? 0 : f.sl * 22 + f.sr) * CRL_PONG_TOP) * kRowChunks;
    const uint32_t bg = blank ? 0u : 0xFFFFFFFFu;

    for (int q = threadIdx.x; q < views * kFrameChunks; q += 256) {
        const int view = q >= kFrameChunks;
        const int c = q - view * kFrameChunks;
        const int row = c / kRowChunks;
        const int cc = c - row * kRowChunks;
        const bool mirror = view && row >= CRL_PONG_MIRROR_ROW;
        const int sc = mirror ? (kRowChunks - 1 - cc) : cc;  // source chunk in the unmirrored row
        uint4 v = make_uint4(bg, bg, bg, bg);
        if (!blank) {
            if (row < CRL_PONG_TOP) {
                if (row >= ink_row0 && row < ink_row1) v = band[row * kRowChunks + sc];
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
        out[q] = v;
    }
}

void launch_pong_raster_raw(const uint64_t *frames, int64_t n, const uint8_t *atlas_rgb, int ink_row0, int ink_row1,
                            uint8_t *obs, int views, hipStream_t st) {
    if (n <= 0) return;
    hipLaunchKernelGGL(pong_raster_raw_kernel, dim3((unsigned)n), dim3(256), 0, st, frames,
                       reinterpret_cast<const uint4 *>(atlas_rgb), ink_row0, ink_row1, reinterpret_cast<uint4 *>(obs), views);
}

}  // namespace crl
