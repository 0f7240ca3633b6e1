// pong_raster_gray.hip -- fused Atari preprocessing for cPongDouble:
//   skip-4 / max-of-last-2 / RGB2GRAY / cv2.resize(INTER_AREA, RxR) / K-frame stack
// written straight to (N, 2, K, R, R) uint8 without ever materialising an RGB frame.
//
// Restates (reference, relative to competitive_rl/): MaxAndSkipEnv.step max over the two
// kept frames utils/atari_wrappers.py:155-158; WarpFrame.parse_single_frame :215-219;
// FrameStackTensor.update utils/utils.py:158-170; the pixel function of the raw frame
// pong/base_pong_env.py:259-266,149-155.  cv2's INTER_AREA is third-party: the f32
// accumulation order of its ResizeArea_Invoker is restated (parity unpinned).
//
// Mapping: ONE WAVEFRONT PER (env, view, plane) TILE.  A tile is R*R bytes (7 056 at
// R=84) staged in LDS:
//   1. fill   -- the "empty court" template: score-dependent top rows from a pre-resized
//                band table (L2 resident), zeros for the court, white bottom rows;
//   2. patch  -- only output pixels whose INTER_AREA footprint touches a ball/bat
//                rectangle of either kept frame (<= ~200 px) are evaluated exactly,
//                one lane per pixel, and byte-stored into the LDS tile;
//   3. stream -- ds_read_b128 -> global_store_dwordx4, 1 KiB contiguous per wave store.
// The whole stack is re-drawn from the 16-byte frame pairs kept in the ring, so older
// planes are never read back from HBM: traffic = K*R*R*2 bytes of stores per env-step.
// LDS ordering inside a tile needs no workgroup barrier: one wave owns the tile and LDS
// operations of a wave complete in issue order.
#include "pong_device.h"

namespace crl {

struct GrayCtx {
    const uint8_t *atlas_gray;
    const int32_t *xofs, *yofs, *xsi, *ysi;
    const float *xalpha, *yalpha;
};

// gray value of source pixel (r, c) of `view` for one frame
__device__ inline int px_view(const Frame &f, const uint8_t *__restrict__ atlas, int view, int r, int c) {
    if (f.sl == 255) return 0;  // BLANK plane
    if (view == 1 && r >= CRL_PONG_MIRROR_ROW) c = CRL_PONG_W - 1 - c;
    if (r < CRL_PONG_TOP) return atlas[((f.sl * 22 + f.sr) * CRL_PONG_TOP + r) * CRL_PONG_W + c];
    if (r >= CRL_PONG_BOTTOM) return 255;
    const bool ball = (unsigned)(c - f.x) < (unsigned)CRL_PONG_BALL && (unsigned)(r - f.y) < (unsigned)CRL_PONG_BALL;
    const bool bl = (unsigned)(c - CRL_PONG_BATL_X) < (unsigned)CRL_PONG_BAT_W && (unsigned)(r - f.bl) < (unsigned)CRL_PONG_BAT_H;
    const bool br = (unsigned)(c - CRL_PONG_BATR_X) < (unsigned)CRL_PONG_BAT_W && (unsigned)(r - f.br) < (unsigned)CRL_PONG_BAT_H;
    return (ball || bl || br) ? 255 : 0;
}

// One output pixel, OpenCV accumulation order: per source row buf = sum_k S*alpha_k (k
// ascending, f32), then sum = beta_0*buf_0 (+= beta_j*buf_j), saturate_cast = rint.
__device__ inline uint8_t eval_pixel(const GrayCtx &g, const Frame &fa, const Frame &fb, int view, int dy, int dx) {
    const int j0 = g.yofs[dy], j1 = g.yofs[dy + 1];
    const int k0 = g.xofs[dx], k1 = g.xofs[dx + 1];
    float sum = 0.f;
    for (int j = j0; j < j1; j++) {
        const int r = g.ysi[j];
        float buf = 0.f;
        for (int k = k0; k < k1; k++) {
            const int c = g.xsi[k];
            const int s = max(px_view(fa, g.atlas_gray, view, r, c), px_view(fb, g.atlas_gray, view, r, c));
            buf = buf + (float)s * g.xalpha[k];
        }
        const float t = g.yalpha[j] * buf;
        sum = (j == j0) ? t : sum + t;
    }
    const int v = (int)rintf(sum);
    return (uint8_t)min(max(v, 0), 255);
}

struct GrayGeom {
    int R, K, band_rows, band_chunks;  // band table holds band_chunks*16 bytes per (score pair, view)
    const uint8_t *band;               // [484][2][band_chunks*16]
    const uint8_t *rest;               // [R*R] score-independent template (used for bytes >= band_chunks*16)
    int zero_row0, zero_row1;          // output rows [zero_row0, zero_row1) of the template are all 0
    const uint8_t *x_first, *x_last;   // [160] first/last output col fed by a source col
    const uint8_t *y_first, *y_last;   // [210]
};

// Builds the templates with the exact evaluator (ball and bats moved off-screen), so the
// fast path is consistent with the per-pixel definition by construction.
__global__ __launch_bounds__(256) void pong_gray_template_kernel(GrayCtx g, GrayGeom q, uint8_t *band, uint8_t *rest) {
    const int R = q.R, bb = q.band_chunks * 16;
    const int total_band = 484 * 2 * bb;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    Frame e;
    e.x = -100, e.y = -100, e.bl = 250, e.br = 250;
    if (i < total_band) {
        const int sp = i / (2 * bb), rem = i - sp * 2 * bb, view = rem / bb, idx = rem - view * bb;
        e.sl = sp / 22, e.sr = sp - e.sl * 22;
        band[i] = idx < R * R ? eval_pixel(g, e, e, view, idx / R, idx % R) : 0;
    } else if (i < total_band + R * R) {
        const int idx = i - total_band;
        e.sl = 0, e.sr = 0;
        rest[idx] = eval_pixel(g, e, e, 0, idx / R, idx % R);
    }
}

struct Box {
    int x0, y0, w, h;  // output-pixel box
};

__device__ inline Box rect_box(const GrayGeom &q, int c0, int c1, int r0, int r1) {
    // source rect [c0,c1) x [r0,r1) (already in view coordinates) -> affected output box
    c0 = max(c0, 0), c1 = min(c1, CRL_PONG_W), r0 = max(r0, 0), r1 = min(r1, CRL_PONG_H);
    Box b = {0, 0, 0, 0};
    if (c0 >= c1 || r0 >= r1) return b;
    b.x0 = q.x_first[c0], b.y0 = q.y_first[r0];
    b.w = q.x_last[c1 - 1] - b.x0 + 1, b.h = q.y_last[r1 - 1] - b.y0 + 1;
    return b;
}

static constexpr int kTileLds = 7168;  // >= 84*84, multiple of 16

__global__ __launch_bounds__(256) void pong_raster_gray_kernel(const uint64_t *__restrict__ ring, int64_t n, GrayCtx g,
                                                               GrayGeom q, uint8_t *__restrict__ obs) {
    __shared__ __attribute__((aligned(16))) uint8_t lds[4][kTileLds];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t tile = (int64_t)blockIdx.x * 4 + wave;
    const int tiles_per_env = 2 * q.K;
    if (tile >= n * tiles_per_env) return;
    const int64_t env = tile / tiles_per_env;
    const int t = (int)(tile - env * tiles_per_env);
    const int view = t / q.K, plane = t - view * q.K;
    const int rp = 4 - q.K + plane;  // ring plane
    const uint64_t pa = ring[(int64_t)(2 * rp + 0) * n + env], pb = ring[(int64_t)(2 * rp + 1) * n + env];
    const Frame fa = unpack_frame(pa), fb = unpack_frame(pb);
    const int R = q.R, RR = R * R, chunks = (RR + 15) >> 4;
    // R*R % 16 == 0 (R = 84): 16-byte stores; otherwise (R = 42: 1764 B tiles) dword stores
    const bool vec16 = (RR & 15) == 0;
    uint4 *__restrict__ out = reinterpret_cast<uint4 *>(obs + tile * (int64_t)RR);
    uint32_t *__restrict__ out32 = reinterpret_cast<uint32_t *>(obs + tile * (int64_t)RR);
    uint8_t *tl = lds[wave];
    uint4 *tl4 = reinterpret_cast<uint4 *>(tl);

    const bool blank_a = fa.sl == 255, blank_b = fb.sl == 255;
    if (blank_a && blank_b) {  // plane erased by a done (FrameStackTensor mask)
        if (vec16)
            for (int c = lane; c < chunks; c += 64) out[c] = make_uint4(0, 0, 0, 0);
        else
            for (int w = lane; w < (RR >> 2); w += 64) out32[w] = 0u;
        return;
    }
    // fast path needs one score pair for the whole plane (a point scored between the two
    // kept frames puts two different texts under the max)
    const bool slow = blank_a || blank_b || fa.sl != fb.sl || fa.sr != fb.sr;

    // ---- 1. fill
    const int bb = q.band_chunks;
    const uint4 *__restrict__ band4 =
        reinterpret_cast<const uint4 *>(q.band) + (int64_t)((slow ? 0 : (fa.sl * 22 + fa.sr)) * 2 + view) * bb;
    const uint4 *__restrict__ rest4 = reinterpret_cast<const uint4 *>(q.rest);
    const int zc0 = (q.zero_row0 * R + 15) >> 4, zc1 = (q.zero_row1 * R) >> 4;  // chunks fully inside zero rows
    for (int c = lane; c < chunks; c += 64) {
        uint4 v = make_uint4(0, 0, 0, 0);
        if (c < bb) v = band4[c];
        else if (c < zc0 || c >= zc1) v = rest4[c];
        tl4[c] = v;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");

    // ---- 2. patch: boxes of the six rectangles in view coordinates
    Box bx[6];
    {
        const bool m = view == 1;
        const int bax0 = m ? CRL_PONG_W - fa.x - CRL_PONG_BALL : fa.x, bbx0 = m ? CRL_PONG_W - fb.x - CRL_PONG_BALL : fb.x;
        const int lx = m ? CRL_PONG_BATR_X : CRL_PONG_BATL_X, rx = m ? CRL_PONG_BATL_X : CRL_PONG_BATR_X;
        const Box none = {0, 0, 0, 0};
        bx[0] = blank_a ? none : rect_box(q, bax0, bax0 + CRL_PONG_BALL, max(fa.y, CRL_PONG_TOP), min(fa.y + CRL_PONG_BALL, CRL_PONG_BOTTOM));
        bx[1] = blank_a ? none : rect_box(q, lx, lx + CRL_PONG_BAT_W, fa.bl, fa.bl + CRL_PONG_BAT_H);
        bx[2] = blank_a ? none : rect_box(q, rx, rx + CRL_PONG_BAT_W, fa.br, fa.br + CRL_PONG_BAT_H);
        const bool same_ball = !blank_a && fa.x == fb.x && fa.y == fb.y;
        bx[3] = (blank_b || same_ball) ? none : rect_box(q, bbx0, bbx0 + CRL_PONG_BALL, max(fb.y, CRL_PONG_TOP), min(fb.y + CRL_PONG_BALL, CRL_PONG_BOTTOM));
        bx[4] = (blank_b || (!blank_a && fa.bl == fb.bl)) ? none : rect_box(q, lx, lx + CRL_PONG_BAT_W, fb.bl, fb.bl + CRL_PONG_BAT_H);
        bx[5] = (blank_b || (!blank_a && fa.br == fb.br)) ? none : rect_box(q, rx, rx + CRL_PONG_BAT_W, fb.br, fb.br + CRL_PONG_BAT_H);
    }
    int pre[7];
    pre[0] = slow ? q.band_rows * R : 0;  // slow path: evaluate every pixel of the score rows
#pragma unroll
    for (int i = 0; i < 6; i++) pre[i + 1] = pre[i] + bx[i].w * bx[i].h;
    const int total = pre[6];
    for (int p = lane; p < total; p += 64) {
        int dy, dx;
        if (p < pre[0]) {
            dy = p / R, dx = p - dy * R;
        } else {
            int i = 0;
#pragma unroll
            for (int k = 1; k < 6; k++) i += (p >= pre[k]) ? 1 : 0;
            // select box i without dynamic register indexing
            int x0 = bx[0].x0, y0 = bx[0].y0, w = bx[0].w, base = pre[0];
#pragma unroll
            for (int k = 1; k < 6; k++)
                if (i == k) x0 = bx[k].x0, y0 = bx[k].y0, w = bx[k].w, base = pre[k];
            const int o = p - base;
            const int yy = o / w;
            dy = y0 + yy, dx = x0 + (o - yy * w);
        }
        tl[dy * R + dx] = eval_pixel(g, fa, fb, view, dy, dx);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");

    // ---- 3. stream the tile out
    if (vec16) {
        for (int c = lane; c < chunks; c += 64) out[c] = tl4[c];
    } else {
        const uint32_t *tl32 = reinterpret_cast<const uint32_t *>(tl);
        for (int w = lane; w < (RR >> 2); w += 64) out32[w] = tl32[w];
    }
}

void launch_pong_gray_templates(const GrayParams &p, const uint8_t *x_first, const uint8_t *x_last, const uint8_t *y_first,
                                const uint8_t *y_last, int band_rows, int band_chunks, uint8_t *band, uint8_t *rest,
                                hipStream_t st) {
    GrayCtx g = {p.atlas_gray, p.xofs, p.yofs, p.xsi, p.ysi, p.xalpha, p.yalpha};
    GrayGeom q = {};
    q.R = p.R, q.K = p.K, q.band_rows = band_rows, q.band_chunks = band_chunks;
    const int total = 484 * 2 * band_chunks * 16 + p.R * p.R;
    hipLaunchKernelGGL(pong_gray_template_kernel, dim3((total + 255) / 256), dim3(256), 0, st, g, q, band, rest);
}

void launch_pong_raster_gray_ex(const GrayParams &p, const uint8_t *rest, int zero_row0, int zero_row1,
                                const uint8_t *x_first, const uint8_t *x_last, const uint8_t *y_first,
                                const uint8_t *y_last, int band_chunks, hipStream_t st) {
    if (p.n <= 0) return;
    GrayCtx g = {p.atlas_gray, p.xofs, p.yofs, p.xsi, p.ysi, p.xalpha, p.yalpha};
    GrayGeom q;
    q.R = p.R, q.K = p.K, q.band_rows = p.band_rows, q.band_chunks = band_chunks;
    q.band = p.band, q.rest = rest, q.zero_row0 = zero_row0, q.zero_row1 = zero_row1;
    q.x_first = x_first, q.x_last = x_last, q.y_first = y_first, q.y_last = y_last;
    const int64_t tiles = p.n * 2 * p.K;
    hipLaunchKernelGGL(pong_raster_gray_kernel, dim3((unsigned)((tiles + 3) / 4)), dim3(256), 0, st, p.ring, p.n, g, q,
                       p.obs);
}

}  // namespace crl
