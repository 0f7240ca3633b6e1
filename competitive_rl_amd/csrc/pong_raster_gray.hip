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
#include <stdio.h>
#include <stdlib.h>

#include "pong_device.h"
#include "crl_internal.h"

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
    int R, K, views, band_rows, band_chunks;  // band table holds band_chunks*16 bytes per (score pair, view)
    const uint8_t *band;               // [484][2][band_chunks*16]
    const uint8_t *rest;               // [R*R] score-independent template (used for bytes >= band_chunks*16)
    int zero_row0, zero_row1;          // output rows [zero_row0, zero_row1) of the template are all 0
    const uint8_t *x_first, *x_last;   // [160] first/last output col fed by a source col
    const uint8_t *y_first, *y_last;   // [210]
    const uint8_t *tab_blob;           // dense tap tables + first/last maps, staged into LDS per workgroup
    GrayTabOfs t;
    int debug;                         // ablation switches for profiling builds (0 in production)
};

// Builds the templates with the exact evaluator (ball and bats moved off-screen), so the
// fast path is consistent with the per-pixel definition by construction.
__global__ __launch_bounds__(256) void pong_gray_template_kernel(GrayCtx g, GrayGeom q, uint8_t *band, uint8_t *rest) {
    // band[((variant * 484 + sp) * 2 + view) * bb + idx]: top rows of the empty court for
    // variant 0: both kept frames show score pair sp; 1: the second shows (sl+1, sr);
    // 2: the second shows (sl, sr+1) -- a point scored between the two max-pooled frames.
    const int R = q.R, bb = q.band_chunks * 16;
    const int per_variant = 484 * 2 * bb, total_band = 3 * per_variant;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    Frame e;
    e.x = -100, e.y = -100, e.bl = 250, e.br = 250;
    if (i < total_band) {
        const int variant = i / per_variant, i2 = i - variant * per_variant;
        const int sp = i2 / (2 * bb), rem = i2 - sp * 2 * bb, view = rem / bb, idx = rem - view * bb;
        e.sl = sp / 22, e.sr = sp - e.sl * 22;
        Frame e2 = e;
        if (variant == 1) e2.sl += 1;
        if (variant == 2) e2.sr += 1;
        const bool valid = e2.sl < 22 && e2.sr < 22 && idx < R * R;
        band[i] = valid ? eval_pixel(g, e, e2, view, idx / R, idx % R) : 0;
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

[[maybe_unused]] static constexpr int kTileLds = 7168;  // >= 84*84, multiple of 16
static constexpr int kTabLds = 6144;   // dense tap tables + first/last maps (4.6 KB at R = 84)
static constexpr int kMaxR = 96;       // largest resized_dim the LDS tile holds (84) rounded up

// Rectangles of the two kept frames in VIEW coordinates (agent 1 sees the court mirrored,
// so its left-hand bat is the physical right bat).  A blank frame has its rows at -1000.
struct Rects {
    int ax, ay, bx, by;      // balls of frame a / b
    int la, lb, ra, rb;      // y of the view-left bat in a / b, view-right bat in a / b
};

// Exact INTER_AREA value of one output pixel whose footprint lies in source rows that hold
// no score ink (rows >= ink_row1): those rows are white bands or court rectangles, so the
// source pixel is a few interval tests instead of a table lookup.  Same f32 operation
// order as eval_pixel.
template <int MAXT>
__device__ inline uint8_t eval_fast(const uint8_t *__restrict__ tabs, const GrayTabOfs &o, int R, const Rects &q, int dy,
                                    int dx) {
    const float *xa = reinterpret_cast<const float *>(tabs + o.xa), *ya = reinterpret_cast<const float *>(tabs + o.ya);
    const int sx0 = tabs[o.xs0 + dx], nx = tabs[o.xn + dx], sy0 = tabs[o.ys0 + dy], ny = tabs[o.yn + dy];
    float p[MAXT];
    unsigned xb[MAXT];
#pragma unroll
    for (int k = 0; k < MAXT; k++) {
        const int c = sx0 + k;
        p[k] = xa[k * R + dx];  // table holds 255 * alpha
        const unsigned m = ((unsigned)(c - q.ax) < (unsigned)CRL_PONG_BALL ? 1u : 0u) |
                           ((unsigned)(c - q.bx) < (unsigned)CRL_PONG_BALL ? 2u : 0u) |
                           ((unsigned)(c - CRL_PONG_BATL_X) < (unsigned)CRL_PONG_BAT_W ? 4u : 0u) |
                           ((unsigned)(c - CRL_PONG_BATR_X) < (unsigned)CRL_PONG_BAT_W ? 8u : 0u);
        xb[k] = k < nx ? m : 0u;
    }
    float sum = 0.f;
#pragma unroll
    for (int j = 0; j < MAXT; j++) {
        if (j < ny) {
            const int r = sy0 + j;
            const bool white = r < CRL_PONG_TOP || r >= CRL_PONG_BOTTOM;
            unsigned yb = ((unsigned)(r - q.ay) < (unsigned)CRL_PONG_BALL ? 1u : 0u) |
                          ((unsigned)(r - q.by) < (unsigned)CRL_PONG_BALL ? 2u : 0u) |
                          (((unsigned)(r - q.la) < (unsigned)CRL_PONG_BAT_H || (unsigned)(r - q.lb) < (unsigned)CRL_PONG_BAT_H) ? 4u : 0u) |
                          (((unsigned)(r - q.ra) < (unsigned)CRL_PONG_BAT_H || (unsigned)(r - q.rb) < (unsigned)CRL_PONG_BAT_H) ? 8u : 0u);
            float buf = 0.f;
#pragma unroll
            for (int k = 0; k < MAXT; k++) {
                const bool on = k < nx && (white || (xb[k] & yb) != 0u);
                buf = on ? buf + p[k] : buf;
            }
            const float t = ya[j * R + dy] * buf;
            sum = (j == 0) ? t : sum + t;
        }
    }
    const int v = (int)rintf(sum);
    return (uint8_t)min(max(v, 0), 255);
}

__device__ inline Box rect_box_lds(const uint8_t *__restrict__ tabs, const GrayTabOfs &o, int c0, int c1, int r0, int r1) {
    c0 = max(c0, 0), c1 = min(c1, CRL_PONG_W), r0 = max(r0, 0), r1 = min(r1, CRL_PONG_H);
    Box b = {0, 0, 0, 0};
    if (c0 >= c1 || r0 >= r1) return b;
    b.x0 = tabs[o.xf + c0], b.y0 = tabs[o.yf + r0];
    b.w = tabs[o.xl + c1 - 1] - b.x0 + 1, b.h = tabs[o.yl + r1 - 1] - b.y0 + 1;
    return b;
}

// The same box without control flow (round 6): the four table reads are issued whatever the rectangle is (at clamped, always valid
// indices) and an empty or absent rectangle selects zeros afterwards -- so the 24 reads of a tile's six boxes are independent loads the
// compiler can issue together (one LDS latency) instead of six branchy chains.  Same values as rect_box_lds wherever that reads at all.
__device__ inline Box rect_box_lds_nb(const uint8_t *__restrict__ tabs, const GrayTabOfs &o, bool none, int c0, int c1, int r0, int r1) {
    c0 = max(c0, 0), c1 = min(c1, CRL_PONG_W), r0 = max(r0, 0), r1 = min(r1, CRL_PONG_H);
    const bool empty = none || c0 >= c1 || r0 >= r1;
    const int ic0 = min(c0, CRL_PONG_W - 1), ic1 = min(max(c1 - 1, 0), CRL_PONG_W - 1);
    const int ir0 = min(r0, CRL_PONG_H - 1), ir1 = min(max(r1 - 1, 0), CRL_PONG_H - 1);
    const int x0 = tabs[o.xf + ic0], y0 = tabs[o.yf + ir0], x1 = tabs[o.xl + ic1], y1 = tabs[o.yl + ir1];
    Box b;
    b.x0 = empty ? 0 : x0, b.y0 = empty ? 0 : y0, b.w = empty ? 0 : x1 - x0 + 1, b.h = empty ? 0 : y1 - y0 + 1;
    return b;
}

#ifdef CRL_ABLATION  // the first tile kernel (four tiles per workgroup, per-pixel evaluator): profiling build only, CRL_GRAY_DEBUG=8
__global__ __launch_bounds__(256) void pong_raster_gray_kernel(const uint64_t *__restrict__ ring, int64_t n, GrayCtx g,
                                                               GrayGeom q, uint8_t *__restrict__ obs) {
    __shared__ __attribute__((aligned(16))) uint8_t lds[4][kTileLds];
    __shared__ __attribute__((aligned(16))) uint8_t tabs[kTabLds];
    // stage the constant tap tables once per workgroup (L2-resident source)
    {
        const uint4 *src = reinterpret_cast<const uint4 *>(q.tab_blob);
        uint4 *dst = reinterpret_cast<uint4 *>(tabs);
        for (int i = threadIdx.x; i < (q.t.total >> 4); i += 256) dst[i] = src[i];
    }
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t tile = (int64_t)blockIdx.x * 4 + wave;
    const int tiles_per_env = q.views * q.K;
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
    // The score rows come pre-resized from the band table when the two kept frames show
    // the same score pair, or pairs one point apart (a point scored between the two
    // max-pooled frames puts two texts under the max).  Anything else (only reachable
    // through set_state or the never-written initial buffers) is evaluated per pixel.
    bool slow = blank_a || blank_b;
    int variant = 0, sp = fa.sl * 22 + fa.sr;
    if (!slow && (fa.sl != fb.sl || fa.sr != fb.sr)) {
        const int spb = fb.sl * 22 + fb.sr;
        if (fb.sl == fa.sl + 1 && fb.sr == fa.sr) variant = 1;
        else if (fb.sl == fa.sl && fb.sr == fa.sr + 1) variant = 2;
        else if (fa.sl == fb.sl + 1 && fa.sr == fb.sr) variant = 1, sp = spb;
        else if (fa.sl == fb.sl && fa.sr == fb.sr + 1) variant = 2, sp = spb;
        else slow = true;
    }
    if (q.debug & 4) slow = false;

    // ---- 1. fill
    const int bb = q.band_chunks;
    const uint4 *__restrict__ band4 =
        reinterpret_cast<const uint4 *>(q.band) + (int64_t)((slow ? 0 : (variant * 484 + sp)) * 2 + view) * bb;
    const uint4 *__restrict__ rest4 = reinterpret_cast<const uint4 *>(q.rest);
    const int zc0 = (q.zero_row0 * R + 15) >> 4, zc1 = (q.zero_row1 * R) >> 4;  // chunks fully inside zero rows
    for (int c = lane; c < chunks; c += 64) {
        uint4 v = make_uint4(0, 0, 0, 0);
        if (!(q.debug & 2)) {
            if (c < bb) v = band4[c];
            else if (c < zc0 || c >= zc1) v = rest4[c];
        }
        tl4[c] = v;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");

    // ---- 2. patch: boxes of the six rectangles in view coordinates
    const bool m = view == 1;
    Rects rc;
    rc.ax = blank_a ? -1000 : (m ? CRL_PONG_W - fa.x - CRL_PONG_BALL : fa.x), rc.ay = blank_a ? -1000 : fa.y;
    rc.bx = blank_b ? -1000 : (m ? CRL_PONG_W - fb.x - CRL_PONG_BALL : fb.x), rc.by = blank_b ? -1000 : fb.y;
    rc.la = blank_a ? -1000 : (m ? fa.br : fa.bl), rc.ra = blank_a ? -1000 : (m ? fa.bl : fa.br);
    rc.lb = blank_b ? -1000 : (m ? fb.br : fb.bl), rc.rb = blank_b ? -1000 : (m ? fb.bl : fb.br);
    Box bx[6];
    {
        const Box none = {0, 0, 0, 0};
        bx[0] = blank_a ? none : rect_box_lds(tabs, q.t, rc.ax, rc.ax + CRL_PONG_BALL, max(rc.ay, CRL_PONG_TOP), min(rc.ay + CRL_PONG_BALL, CRL_PONG_BOTTOM));
        bx[1] = blank_a ? none : rect_box_lds(tabs, q.t, CRL_PONG_BATL_X, CRL_PONG_BATL_X + CRL_PONG_BAT_W, rc.la, rc.la + CRL_PONG_BAT_H);
        bx[2] = blank_a ? none : rect_box_lds(tabs, q.t, CRL_PONG_BATR_X, CRL_PONG_BATR_X + CRL_PONG_BAT_W, rc.ra, rc.ra + CRL_PONG_BAT_H);
        const bool same_ball = rc.ax == rc.bx && rc.ay == rc.by;
        bx[3] = (blank_b || same_ball) ? none : rect_box_lds(tabs, q.t, rc.bx, rc.bx + CRL_PONG_BALL, max(rc.by, CRL_PONG_TOP), min(rc.by + CRL_PONG_BALL, CRL_PONG_BOTTOM));
        bx[4] = (blank_b || rc.la == rc.lb) ? none : rect_box_lds(tabs, q.t, CRL_PONG_BATL_X, CRL_PONG_BATL_X + CRL_PONG_BAT_W, rc.lb, rc.lb + CRL_PONG_BAT_H);
        bx[5] = (blank_b || rc.ra == rc.rb) ? none : rect_box_lds(tabs, q.t, CRL_PONG_BATR_X, CRL_PONG_BATR_X + CRL_PONG_BAT_W, rc.rb, rc.rb + CRL_PONG_BAT_H);
    }
    int pre[7];
    pre[0] = slow ? q.band_rows * R : 0;  // slow path: evaluate every pixel of the score rows
#pragma unroll
    for (int i = 0; i < 6; i++) pre[i + 1] = pre[i] + bx[i].w * bx[i].h;
    const int total = (q.debug & 1) ? 0 : pre[6];
    const bool fast_ok = q.t.fast_ok != 0;
    for (int p = lane; p < total; p += 64) {
        int dy, dx;
        const bool band_px = p < pre[0];
        if (band_px) {
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
        uint8_t v;
        if (band_px || !fast_ok) v = eval_pixel(g, fa, fb, view, dy, dx);
        else v = q.t.max_taps <= 3 ? eval_fast<3>(tabs, q.t, R, rc, dy, dx) : eval_fast<5>(tabs, q.t, R, rc, dy, dx);
        tl[dy * R + dx] = v;
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

#endif  // CRL_ABLATION

// ---------------------------------------------------------------------------------------
// Production kernel: ONE WAVEFRONT PER (env, plane group), looping over the group's planes
// and both views.  Compared with the tile-per-wave kernel above (kept for ablation,
// CRL_GRAY_DEBUG=8) this
//   * pays the HBM latency of the ring read once per env instead of once per tile (lanes
//     0..7 fetch the eight frame words, parked in LDS) and stages the tap tables once per
//     32 tiles instead of once per 4;
//   * keeps everything that is uniform over the wavefront -- frame decode, score-pair
//     classification, the six output boxes -- in scalar registers (the wave index goes
//     through readfirstlane, the first/last maps are int32 so they come in by s_load);
//   * issues the template loads of a tile before its box arithmetic so their L2 latency is
//     covered, and decodes patch positions with a reciprocal instead of an integer divide.
// Separable form of eval_fast.  For the pixel (dy, dx) the source pixel under tap (j, k) is
// lit iff some rectangle (ball a, ball b, view-left bat, view-right bat, white band) holds
// both source row sy0+j and source col sx0+k.  Row membership depends only on dy and
// column membership only on dx, so each is computed once per tile into a 5-bit-per-tap
// word (rowpack[dy], colpack[dx]) and a pixel costs one AND per tap pair:
//   lit(j, k) = ((rowpack >> 5j) & (colpack >> 5k) & 31) != 0.
// Bit 0 ball a, 1 ball b, 2 left bat, 3 right bat, 4 white (rows: the row is white; cols:
// the tap exists).  The static bits (bats' columns, white rows, tap validity) come from
// tables built on the host.  Same f32 operation order as eval_pixel.
template <int MAXT>
__device__ inline uint32_t row_pack(const uint8_t *__restrict__ tabs, const GrayTabOfs &o, const Rects &q, int dy) {
    const int sy0 = tabs[o.ys0 + dy], ny = tabs[o.yn + dy];
    uint32_t pack = 0;
#pragma unroll
    for (int j = 0; j < MAXT; j++) {
        const int r = sy0 + j;
        const uint32_t b = ((unsigned)(r - q.ay) < (unsigned)CRL_PONG_BALL ? 1u : 0u) |
                           ((unsigned)(r - q.by) < (unsigned)CRL_PONG_BALL ? 2u : 0u) |
                           (((unsigned)(r - q.la) < (unsigned)CRL_PONG_BAT_H || (unsigned)(r - q.lb) < (unsigned)CRL_PONG_BAT_H) ? 4u : 0u) |
                           (((unsigned)(r - q.ra) < (unsigned)CRL_PONG_BAT_H || (unsigned)(r - q.rb) < (unsigned)CRL_PONG_BAT_H) ? 8u : 0u);
        pack |= b << (5 * j);
    }
    const uint32_t valid = (1u << (5 * ny)) - 1u;
    return (pack & valid) | reinterpret_cast<const uint32_t *>(tabs + o.rowstatic)[dy];
}

template <int MAXT>
__device__ inline uint32_t col_pack(const uint8_t *__restrict__ tabs, const GrayTabOfs &o, const Rects &q, int dx) {
    const int sx0 = tabs[o.xs0 + dx], nx = tabs[o.xn + dx];
    uint32_t pack = 0;
#pragma unroll
    for (int k = 0; k < MAXT; k++) {
        const int c = sx0 + k;
        const uint32_t b = ((unsigned)(c - q.ax) < (unsigned)CRL_PONG_BALL ? 1u : 0u) | ((unsigned)(c - q.bx) < (unsigned)CRL_PONG_BALL ? 2u : 0u);
        pack |= b << (5 * k);
    }
    const uint32_t valid = (1u << (5 * nx)) - 1u;
    return (pack & valid) | reinterpret_cast<const uint32_t *>(tabs + o.colstatic)[dx];
}

template <int MAXT>
__device__ inline uint8_t eval_sep(const uint8_t *__restrict__ tabs, const GrayTabOfs &o, int R, uint32_t rp, uint32_t cp, int dy,
                                   int dx) {
    const float *xa = reinterpret_cast<const float *>(tabs + o.xa255), *ya = reinterpret_cast<const float *>(tabs + o.ya);
    float p[MAXT];
    uint32_t cm[MAXT];
#pragma unroll
    for (int k = 0; k < MAXT; k++) p[k] = xa[k * R + dx], cm[k] = (cp >> (5 * k)) & 31u;
    float sum = 0.f;
#pragma unroll
    for (int j = 0; j < MAXT; j++) {
        const uint32_t rn = (rp >> (5 * j)) & 31u;
        float buf = 0.f;
#pragma unroll
        for (int k = 0; k < MAXT; k++) buf = buf + ((rn & cm[k]) != 0u ? p[k] : 0.f);
        const float t = ya[j * R + dy] * buf;
        sum = (j == 0) ? t : sum + t;
    }
    const int v = (int)rintf(sum);
    return (uint8_t)min(max(v, 0), 255);
}

// CRL_GRAY_DEBUG & 128 (instrumented instance): s_memtime stamps around the phases of a tile, summed over all wavefronts
__device__ unsigned long long g_gray_ticks[1024][8];  // (1 024 rows, one per workgroup index mod 1 024: every wavefront adding to ONE row took 4.7 ms of atomics per launch)
#define GRAY_TICK(Kk)                                        \
    if (DBG && (dbg & 128)) {                                \
        const long long now_ = __builtin_readcyclecounter(); \
        gtick[Kk] += (unsigned)(now_ - gprev);               \
        gprev = now_;                                        \
    }
// TI = 16-byte chunks per lane per tile: 7 holds R <= 84, 2 holds R <= 45 (the reference's default resized_dim = 42:
// a quarter of the LDS and 40 fewer registers, so more wavefronts cover the per-tile latency chain)
// F32: the observation tensor is float32 (DummyVecEnv's buffers, utils/dummy_vec_env.py:37-44): the tile is built in
// LDS as bytes exactly as for uint8 and widened in the store epilogue -- every store instruction still writes 1 KiB
// contiguous (lane l reads the dword of pixels 4l..4l+3 of a 256-pixel group and stores them as one float4).
// STACK (round 6): FrameStackTensor fused into the draw (GrayStack, pong_device.h).  The wavefront's tiles are then JOBS of its env:
// first the k planes of the bound stack (one agent's view, planes oldest to newest = ring planes 4 - k .. 3; the planes older than the
// stack's last reset() are zeros), then the tiles of the observation tensor (all but the one the stack's newest plane stands in for
// when `alias` is set).  Same tile code, same values; only where a tile goes and its element type (SF32 for the stack) differ.
#ifndef CRL_F32_EPI
#define CRL_F32_EPI 28  // float32 store epilogue: 256-pixel groups per batch (28 = the whole tile's LDS reads before its first store)
#endif
#ifndef CRL_F32_LB
#define CRL_F32_LB 1    // float32 instances: workgroups per CU the register allocation must allow
#endif
// EPWV (round 6): envs per wavefront of the K = 1 launch (make_envs("cPongDouble-v0")'s own observation: one plane per agent).  A wavefront that
// draws ONE env's two tiles pays the workgroup's table staging, its ring read and its launch for 14 KB of output; EPWV consecutive envs per
// wavefront amortise them like the four-plane stack does.
template <int MAXT, bool DBG, int TI, bool F32, bool STACK = false, bool SF32 = false, int EPWV = 1>
#ifndef CRL_GRAY_SMALL_LB
#define CRL_GRAY_SMALL_LB 6  // R <= 45 instances: workgroups per CU the register allocation must allow
#endif
__global__ __launch_bounds__(256, TI == 2 ? CRL_GRAY_SMALL_LB : ((STACK || EPWV > 1) && !F32 && !SF32) ? 4 : (F32 || SF32) ? CRL_F32_LB : 1) void pong_raster_gray_env_kernel(const uint64_t *__restrict__ ring, int64_t n, GrayCtx g,
                                                                   GrayGeom q, uint8_t *__restrict__ obs, int ppw, GrayStack sk) {
    __shared__ __attribute__((aligned(16))) uint8_t lds[4][TI * 1024];
    __shared__ __attribute__((aligned(16))) uint8_t tabs[kTabLds];
    __shared__ uint64_t words_[4][8];
    __shared__ uint32_t rowpack_[4][kMaxR], colpack_[4][kMaxR];
    {
        const uint4 *src = reinterpret_cast<const uint4 *>(q.tab_blob);
        uint4 *dst = reinterpret_cast<uint4 *>(tabs);
        for (int i = threadIdx.x; i < (q.t.total >> 4); i += 256) dst[i] = src[i];
    }
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int K = q.K;
    // STACK: `ppw` = jobs per wavefront out of the env's ks + (views K - alias) jobs
    const int ks = STACK ? sk.k : 0, obs_jobs = STACK ? (obs ? q.views * K - (sk.alias ? 1 : 0) : 0) : 0, jobs = ks + obs_jobs;
    const int alias_tile = STACK && sk.alias ? sk.view * K + K - 1 : 1 << 20;
    const int groups = STACK ? (jobs + ppw - 1) / ppw : K / ppw;
    const int64_t wid = (int64_t)blockIdx.x * 4 + wave;
    const bool live = EPWV > 1 ? wid * EPWV < n : wid < n * groups;   // (EPWV > 1: whole envs, groups = 1)
    const int64_t env = live ? (EPWV > 1 ? wid * EPWV : wid / groups) : 0;
    const int p0 = EPWV > 1 ? 0 : (int)(wid - env * groups) * ppw;  // first plane (first job) of this wavefront
    if (live && lane < 8) words_[wave][lane] = ring[(int64_t)lane * n + env];
    __syncthreads();
    if (!live) return;

    const int R = q.R, RR = R * R, chunks = (RR + 15) >> 4;
    const bool vec16 = (RR & 15) == 0;
    uint8_t *tl = lds[wave];
    uint4 *tl4 = reinterpret_cast<uint4 *>(tl);
    const int bb = q.band_chunks;
    const uint4 *__restrict__ rest4 = reinterpret_cast<const uint4 *>(q.rest);
    const int zc0 = (q.zero_row0 * R + 15) >> 4, zc1 = (q.zero_row1 * R) >> 4;  // chunks fully inside zero rows
    const bool fast_ok = q.t.fast_ok != 0;
    const int dbg = DBG ? q.debug : 0;  // ablation switches (profiling instance only)
    unsigned gtick[6] = {0, 0, 0, 0, 0, 0};
    long long gprev = (DBG && (dbg & 128)) ? __builtin_readcyclecounter() : 0;

    if constexpr (STACK) {
        const int ntiles = min(ppw, jobs - p0);
#pragma unroll 1
        for (int it = 0; it < ntiles; it++) {
            const int j = p0 + it;
            int view, plane, rp;
            bool to_stack = false, erased = false;
            if (j < ks) {
                to_stack = true, view = sk.view, plane = j, rp = 4 - ks + j, erased = j < ks - sk.valid;
            } else {
                int t = j - ks;
                t += t >= alias_tile ? 1 : 0;
                view = t >= K ? 1 : 0, plane = t - view * K, rp = 4 - K + plane;
            }
            const bool f32 = to_stack ? SF32 : F32;
#include "pong_gray_tile.inc"
        }
    } else if constexpr (EPWV > 1) {
        const int64_t env_first = env;
#pragma unroll 1
        for (int e = 0; e < EPWV; e++) {
            const int64_t env = env_first + e;   // (the tile code's `env`)
            if (env >= n) break;
            if (e > 0) {  // the next env's ring words (this wavefront's own LDS row: a wavefront's LDS operations complete in order)
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                if (lane < 8) words_[wave][lane] = ring[(int64_t)lane * n + env];
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            }
#pragma unroll 1
            for (int view = 0; view < q.views; view++)
#pragma unroll 1
            for (int plane = 0; plane < K; plane++) {
                const int rp = 4 - K + plane;  // ring plane
                constexpr bool to_stack = false, erased = false, f32 = F32;
#include "pong_gray_tile.inc"
            }
        }
    } else {
        // view-major: a wave writes its env's planes in ascending address order (the four planes of agent
        // 0's view, then agent 1's) -- 4 % faster than alternating between the two views per plane
#pragma unroll 1
        for (int view = 0; view < q.views; view++)
#pragma unroll 1
        for (int plane = p0; plane < p0 + ppw; plane++) {
            const int rp = 4 - K + plane;  // ring plane
            constexpr bool to_stack = false, erased = false, f32 = F32;
#include "pong_gray_tile.inc"
        }
    }
    if (DBG && (dbg & 128) && lane == 0) {
        unsigned long long *row = g_gray_ticks[blockIdx.x & 1023];
        for (int i = 0; i < 6; i++) atomicAdd(&row[i], (unsigned long long)gtick[i]);
        atomicAdd(&row[7], 1ull);
    }
}

#ifdef CRL_ABLATION  // bit-exact and slower than the env kernel (lab notes 4.3a): profiling build only, CRL_GRAY_SWEEP=1
// ---------------------------------------------------------------------------------------
// Address-linear writer (round 2; CRL_GRAY_SWEEP=0 falls back to the env kernel above).
//
// Why: only a chip-wide dense sweep of stores reaches the fill rate; a wavefront streaming its own 56 KB env writes a
// comb.  Here the output tensor is cut into 1-KiB blocks (64 chunks of 16 B: 1/7 of a 7 056-byte tile at
// R = 84) and block b is written by wavefront b -- workgroups are dispatched in index order, so at any instant the chip
// writes one dense window of the tensor, like a fill.  What makes that affordable:
//   * a tiny HEADER kernel first computes, one lane per tile, everything that is uniform over a tile: blank / fast / slow,
//     the band-table row of its score pair, the rectangles in view coordinates, the six output boxes and which of the
//     tile's blocks a box touches (64 B per tile, 2 % of the step's traffic);
//   * a wavefront reads its tile's header with scalar loads; blocks no box touches (more than half) are the template
//     chunk -- score band rows from the band table, zeros for the court, the white bottom rows -- stored straight away;
//   * a touched block is composed in 1 KiB of LDS exactly as the env kernel does for a whole tile (row / column
//     membership words, one lane per affected pixel, same f32 operation order), restricted to the block's rows.
struct __attribute__((aligned(64))) GrayTileHdr {
    // first 16 bytes = all an untouched segment needs (one scalar load)
    uint32_t band_off;     // first chunk of this tile's score rows in the band table
    uint8_t kind;          // 0 = blank plane (zeros), 1 = fast, 2 = per-pixel evaluator (unrelated scores / one blank frame)
    uint8_t pad[3];
    uint64_t chunkmask;    // bit j: chunks [8 j, 8 j + 8) of the tile hold pixels of a box
    uint8_t box[6][4];     // x0, y0, w, h of the six rectangles' output boxes (w*h = 0: none)
    int16_t rc[8];         // Rects: ax, ay, bx, by, la, lb, ra, rb (view coordinates)
    uint8_t pad2[8];
};
static_assert(sizeof(GrayTileHdr) == 64, "header layout");

__global__ __launch_bounds__(256) void pong_gray_header_kernel(const uint64_t *__restrict__ ring, int64_t n, GrayGeom q,
                                                               GrayTileHdr *__restrict__ hdr) {
    const int64_t tile = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int K = q.K, tiles_per_env = q.views * K;
    if (tile >= n * tiles_per_env) return;
    const int64_t env = tile / tiles_per_env;
    const int t = (int)(tile - env * tiles_per_env), view = t / K, plane = t - view * K, rp = 4 - K + plane;
    const uint64_t pa = ring[(int64_t)(2 * rp) * n + env], pb = ring[(int64_t)(2 * rp + 1) * n + env];
    Frame fa = unpack_frame(pa), fb = unpack_frame(pb);
    if (fa.sl == 255 && fb.sl != 255) fa = fb;  // the reset observation: a single frame, max(x, 0) = x
    else if (fb.sl == 255 && fa.sl != 255) fb = fa;
    GrayTileHdr h;
    memset(&h, 0, sizeof(h));
    if (fa.sl == 255) {  // both blank: plane erased by a done
        hdr[tile] = h;
        reinterpret_cast<uint2 *>(hdr + n * tiles_per_env)[tile] = make_uint2(h.band_off, h.kind);
        reinterpret_cast<uint16_t *>(reinterpret_cast<uint2 *>(hdr + n * tiles_per_env) + n * tiles_per_env)[tile] = 0;
        return;
    }
    bool slow = false;
    int variant = 0, sp = fa.sl * 22 + fa.sr;
    if (fa.sl != fb.sl || fa.sr != fb.sr) {
        const int spb = fb.sl * 22 + fb.sr;
        if (fb.sl == fa.sl + 1 && fb.sr == fa.sr) variant = 1;
        else if (fb.sl == fa.sl && fb.sr == fa.sr + 1) variant = 2;
        else if (fa.sl == fb.sl + 1 && fa.sr == fb.sr) variant = 1, sp = spb;
        else if (fa.sl == fb.sl && fa.sr == fb.sr + 1) variant = 2, sp = spb;
        else slow = true;
    }
    if (!q.t.fast_ok) slow = true;
    h.kind = slow ? 2 : 1;
    h.band_off = (uint32_t)(((slow ? 0 : (variant * 484 + sp)) * 2 + view) * q.band_chunks);
    const bool m = view == 1;
    Rects rc;
    rc.ax = m ? CRL_PONG_W - fa.x - CRL_PONG_BALL : fa.x, rc.ay = fa.y;
    rc.bx = m ? CRL_PONG_W - fb.x - CRL_PONG_BALL : fb.x, rc.by = fb.y;
    rc.la = m ? fa.br : fa.bl, rc.ra = m ? fa.bl : fa.br;
    rc.lb = m ? fb.br : fb.bl, rc.rb = m ? fb.bl : fb.br;
    h.rc[0] = (int16_t)rc.ax, h.rc[1] = (int16_t)rc.ay, h.rc[2] = (int16_t)rc.bx, h.rc[3] = (int16_t)rc.by;
    h.rc[4] = (int16_t)rc.la, h.rc[5] = (int16_t)rc.lb, h.rc[6] = (int16_t)rc.ra, h.rc[7] = (int16_t)rc.rb;
    const uint8_t *tabs = q.tab_blob;
    const Box none = {0, 0, 0, 0};
    Box bx[6];
    const bool same_ball = rc.ax == rc.bx && rc.ay == rc.by;
    bx[0] = rect_box_lds(tabs, q.t, rc.ax, rc.ax + CRL_PONG_BALL, max(rc.ay, CRL_PONG_TOP), min(rc.ay + CRL_PONG_BALL, CRL_PONG_BOTTOM));
    bx[1] = rect_box_lds(tabs, q.t, CRL_PONG_BATL_X, CRL_PONG_BATL_X + CRL_PONG_BAT_W, rc.la, rc.la + CRL_PONG_BAT_H);
    bx[2] = rect_box_lds(tabs, q.t, CRL_PONG_BATR_X, CRL_PONG_BATR_X + CRL_PONG_BAT_W, rc.ra, rc.ra + CRL_PONG_BAT_H);
    bx[3] = same_ball ? none : rect_box_lds(tabs, q.t, rc.bx, rc.bx + CRL_PONG_BALL, max(rc.by, CRL_PONG_TOP), min(rc.by + CRL_PONG_BALL, CRL_PONG_BOTTOM));
    bx[4] = rc.la == rc.lb ? none : rect_box_lds(tabs, q.t, CRL_PONG_BATL_X, CRL_PONG_BATL_X + CRL_PONG_BAT_W, rc.lb, rc.lb + CRL_PONG_BAT_H);
    bx[5] = rc.ra == rc.rb ? none : rect_box_lds(tabs, q.t, CRL_PONG_BATR_X, CRL_PONG_BATR_X + CRL_PONG_BAT_W, rc.rb, rc.rb + CRL_PONG_BAT_H);
    uint64_t mask = 0;
    const int R = q.R;
    for (int i = 0; i < 6; i++) {
        if (bx[i].w * bx[i].h <= 0) bx[i] = none;
        h.box[i][0] = (uint8_t)bx[i].x0, h.box[i][1] = (uint8_t)bx[i].y0, h.box[i][2] = (uint8_t)bx[i].w, h.box[i][3] = (uint8_t)bx[i].h;
        for (int r = bx[i].y0; r < bx[i].y0 + bx[i].h; r++) {
            const int lo = r * R + bx[i].x0, hi = lo + bx[i].w - 1;
            mask |= 1ull << (lo >> 7);  // pixel -> chunk (16 px) -> group of 8 chunks
            mask |= 1ull << (hi >> 7);
        }
    }
    h.chunkmask = mask;
    hdr[tile] = h;
    reinterpret_cast<uint2 *>(hdr + n * tiles_per_env)[tile] = make_uint2(h.band_off, h.kind);  // dense copy (skeleton experiment)
    reinterpret_cast<uint16_t *>(reinterpret_cast<uint2 *>(hdr + n * tiles_per_env) + n * tiles_per_env)[tile] =
        (uint16_t)((h.band_off / (uint32_t)q.band_chunks) | ((uint32_t)h.kind << 12));  // 2-byte record
}

// One wavefront per 1-KiB-ALIGNED block of the output tensor (chunks [64 b, 64 b + 64) of the whole tensor), NB blocks per
// wavefront a whole grid apart.  What the probes and the first versions of this kernel showed:
//   * every wavefront store must be a full, aligned KiB: tiles cut into 6 x 64 + 57 chunks write at 4.2 TB/s, aligned
//     blocks at 6.8 (tools/sweep_overhead_probe.hip).  A 7 056-byte tile is not a multiple of 1 KiB, so a block holds the
//     tail of one tile and the head of the next: up to two SEGMENTS;
//   * the CU's four SIMDs share ONE scalar unit (one scalar instruction per cycle): at the fill rate a CU has ~95 cycles
//     per KiB, and a version whose per-block bookkeeping (tile index, header decode, mask tests: ~150 instructions) ran
//     on the scalar unit was bound by exactly that (900 us with no loads and no raster work at all).  So the bookkeeping
//     is LANE-PARALLEL here: every lane derives its own tile and chunk, reads its tile's header with a vector load
//     (one or two distinct addresses per wavefront: a broadcast), and only the branch decisions are wave-level votes.
__device__ inline int opaque_zero() {  // a zero the compiler must treat as per-lane: keeps uniform arithmetic off the scalar unit
    int z;
    asm volatile("v_mov_b32 %0, 0" : "=v"(z));
    return z;
}

// Measurement only (CRL_GRAY_SWEEP=2, wrong pixels wherever a ball or bat is): the SKELETON of an address-linear writer with
// real per-tile metadata -- tile index, one 16-byte header read, the template chunk, an aligned 1-KiB store -- and none of the
// box path, i.e. what a writer that got its box pixels for free could at best cost.  NB blocks per wavefront, a grid apart.
// F32OUT (round 6): the same blocks widened to float32 on the way out -- a block is 1 024 pixels either way (one 16-byte chunk of bytes per
// lane), so the float32 writer pays the per-block bookkeeping once per 4 KiB instead of once per KiB: the block goes through 1 KiB of LDS
// and leaves as four 1-KiB wave stores (store k, lane l: the float4 of pixels 256 k + 4 l .. + 3).
__device__ inline void store_block_f32(const uint8_t *tl, float4 *__restrict__ outf, int64_t first_chunk, int lane, int64_t total_chunks) {
    const uint32_t *tl32 = reinterpret_cast<const uint32_t *>(tl);
    uint32_t px[4];
#pragma unroll
    for (int k = 0; k < 4; k++) px[k] = tl32[64 * k + lane];
#pragma unroll
    for (int k = 0; k < 4; k++)
        if (first_chunk + 16 * k + (lane >> 2) < total_chunks)
            outf[first_chunk * 4 + 64 * k + lane] = make_float4((float)(px[k] & 255u), (float)((px[k] >> 8) & 255u), (float)((px[k] >> 16) & 255u), (float)(px[k] >> 24));
}

template <int NB, bool F32OUT = false>
__global__ __launch_bounds__(256) void pong_gray_sweep_skeleton_kernel(const GrayTileHdr *__restrict__ hdrs, int n_tiles, GrayGeom q,
                                                                       uint8_t *__restrict__ obs, int stride, int dense) {
    __shared__ __attribute__((aligned(16))) uint8_t lds[F32OUT ? 4 : 1][F32OUT ? 1024 : 16];
    constexpr int R = 84, chunks = R * R >> 4;
    const int total = n_tiles * chunks;
    const int bb = q.band_chunks;
    const int zc0 = (q.zero_row0 * R + 15) >> 4, zc1 = (q.zero_row1 * R) >> 4;
    const uint4 *__restrict__ band4 = reinterpret_cast<const uint4 *>(q.band);
    const uint4 *__restrict__ rest4 = reinterpret_cast<const uint4 *>(q.rest);
    int gl[NB], c[NB];
    uint2 h[NB];
#pragma unroll
    for (int i = 0; i < NB; i++) {
        gl[i] = ((int)blockIdx.x * 256 + (int)threadIdx.x) + i * stride * 64;
        const int gg = min(gl[i], total - 1), tile = gg / chunks;
        c[i] = gg - tile * chunks;
        if (dense == 2) {  // 2 bytes per tile (1 MB for 524 288 tiles: resident in every L2): band-table row | kind << 12
            const uint32_t m16 = reinterpret_cast<const uint16_t *>(reinterpret_cast<const uint2 *>(hdrs + n_tiles) + n_tiles)[tile];
            h[i] = make_uint2((m16 & 4095u) * (uint32_t)bb, m16 >> 12);
        } else {
            h[i] = dense ? reinterpret_cast<const uint2 *>(hdrs + n_tiles)[tile] : *reinterpret_cast<const uint2 *>(hdrs + tile);  // band offset, kind
        }
    }
#pragma unroll
    for (int i = 0; i < NB; i++) {
        uint4 v = make_uint4(0, 0, 0, 0);
        if ((h[i].y & 255u) != 0) {
            if (c[i] < bb) v = band4[h[i].x + c[i]];
            else if (c[i] < zc0 || c[i] >= zc1) v = rest4[c[i]];
        }
        if (F32OUT) {
            const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
            reinterpret_cast<uint4 *>(lds[wave])[lane] = v;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            store_block_f32(lds[wave], reinterpret_cast<float4 *>(obs), (int64_t)gl[i] - lane, lane, total);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        } else if (gl[i] < total) {
            reinterpret_cast<uint4 *>(obs)[gl[i]] = v;
        }
    }
}

template <int RT, int NB, bool F32OUT = false>
__global__ __launch_bounds__(256) void pong_raster_gray_sweep_kernel(const GrayTileHdr *__restrict__ hdrs, int n_tiles, GrayCtx g, GrayGeom q,
                                                                     uint8_t *__restrict__ obs, int dbg, const uint64_t *__restrict__ ring,
                                                                     int64_t n, int stride) {
    __shared__ __attribute__((aligned(16))) uint8_t lds[4][1024];
    __shared__ uint32_t rowpack_[4][16], colpack_[4][kMaxR];
    constexpr int MAXT = 3;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int R = RT ? RT : q.R, RR = R * R, chunks = RR >> 4;   // RT = resized_dim at compile time (84), 0 = run-time value
    const int total = n_tiles * chunks;
    const int bb = q.band_chunks;
    const int zc0 = (q.zero_row0 * R + 15) >> 4, zc1 = (q.zero_row1 * R) >> 4;
    const uint4 *__restrict__ band4 = reinterpret_cast<const uint4 *>(q.band);
    const uint4 *__restrict__ rest4 = reinterpret_cast<const uint4 *>(q.rest);
    int gl[NB], tile[NB], c[NB];
    uint4 h[NB], v[NB];
#pragma unroll
    for (int i = 0; i < NB; i++) {  // ---- every block's header words (16 B per lane, a broadcast), all in flight together
        gl[i] = ((int)blockIdx.x * 256 + (int)threadIdx.x) + i * stride * 64;   // this lane's chunk, whole-tensor index (< 2^31)
        const int gg = min(gl[i], total - 1);
        tile[i] = gg / chunks, c[i] = gg - tile[i] * chunks;
        h[i] = *reinterpret_cast<const uint4 *>(hdrs + tile[i]);
    }
#pragma unroll
    for (int i = 0; i < NB; i++) {  // ---- every block's template chunk: score rows from the band table, zeros for the court, ...
        const uint32_t kind = h[i].y & 255u;
        v[i] = make_uint4(0, 0, 0, 0);
        if (kind != 0 && !(dbg & 4)) {
            if (c[i] < bb) v[i] = band4[h[i].x + c[i]];
            else if (c[i] < zc0 || c[i] >= zc1) v[i] = rest4[c[i]];
        }
    }
#pragma unroll
    for (int i = 0; i < NB; i++) {
        const bool active = gl[i] < total;
        uint4 *__restrict__ out = reinterpret_cast<uint4 *>(obs) + gl[i];
        const uint32_t kind = h[i].y & 255u;
        // bit j of the tile's 64-bit mask: chunks [8 j, 8 j + 8) hold pixels of a box
        const uint32_t mword = (c[i] >> 3) < 32 ? h[i].z : h[i].w;
        const bool mine = active && (kind == 2 || (kind == 1 && ((mword >> ((c[i] >> 3) & 31)) & 1u) && !(dbg & 1)));
        if (!__any(mine)) {  // most blocks: the template, stored straight away
            if (F32OUT) {
                reinterpret_cast<uint4 *>(lds[wave])[lane] = v[i];
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                store_block_f32(lds[wave], reinterpret_cast<float4 *>(obs), (int64_t)gl[i] - lane, lane, total);
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            } else if (active) {
                *out = v[i];
            }
            continue;
        }
        // ---- a block with box pixels: compose it in LDS, segment by segment (wave-uniform tile each)
        uint8_t *tl = lds[wave];
        reinterpret_cast<uint4 *>(tl)[lane] = v[i];
        const int tileA = __builtin_amdgcn_readfirstlane(tile[i]), tileZ = __builtin_amdgcn_readlane(tile[i], 63);
        const int cA = __builtin_amdgcn_readfirstlane(c[i]);
        const int split = tileZ != tileA ? chunks - cA : 64;   // lanes [0, split) are segment A, the rest segment B (the next tile)
        const uint8_t *tabs = q.tab_blob;  // dense tap tables, read through L1 (a few hundred bytes per touched segment)
        uint32_t *rowpack = rowpack_[wave], *colpack = colpack_[wave];
        const int vz = opaque_zero();
#pragma unroll 1
        for (int seg = 0; seg < 2; seg++) {
            if (seg == 1 && tileZ == tileA) break;
            const bool in_seg = (lane < split) == (seg == 0);
            if (!__any(mine && in_seg)) continue;
            const int tl_id = seg ? tileZ : tileA;
            const int c0 = seg ? 0 : cA, c1 = seg ? 64 - split : cA + split;   // the segment's chunks of its tile
            const int l0 = seg ? split : 0;                                    // first lane / LDS chunk of the segment
            const int P0 = c0 * 16, P1 = c1 * 16;
            // the header again, per lane (vector loads + vector arithmetic: every lane needs all six boxes)
            const GrayTileHdr *hp = hdrs + (tl_id + vz);
            const uint4 hw = *reinterpret_cast<const uint4 *>(hp);
            if ((hw.y & 255u) == 2u) {  // rare: unrelated scores under the max, or a blank buffer (set_state / the never-written buffers)
                const int tiles_per_env = q.views * q.K, env = tl_id / tiles_per_env, tt = tl_id - env * tiles_per_env;
                const int view = tt / q.K, rp = 4 - q.K + (tt - view * q.K);
                Frame fa = unpack_frame(ring[(int64_t)(2 * rp) * n + env]), fb = unpack_frame(ring[(int64_t)(2 * rp + 1) * n + env]);
                if (fa.sl == 255 && fb.sl != 255) fa = fb;
                else if (fb.sl == 255 && fa.sl != 255) fb = fa;
                for (int p = P0 + lane; p < P1; p += 64) {
                    const int dy = p / R, dx = p - dy * R;
                    tl[p - P0 + l0 * 16] = eval_pixel(g, fa, fb, view, dy, dx);
                }
                continue;
            }
            const uint4 hb = reinterpret_cast<const uint4 *>(hp)[1];   // boxes 0..3
            const uint4 hc = reinterpret_cast<const uint4 *>(hp)[2];   // boxes 4..5, rects ax ay bx by
            const uint2 hd = reinterpret_cast<const uint2 *>(hp)[6];   // rects la lb ra rb
            const int r0 = P0 / R, r1 = (P1 - 1) / R;   // first / last output row with pixels in this segment
            Rects rc;
            rc.ax = (int16_t)(hc.z & 0xFFFFu), rc.ay = (int16_t)(hc.z >> 16), rc.bx = (int16_t)(hc.w & 0xFFFFu), rc.by = (int16_t)(hc.w >> 16);
            rc.la = (int16_t)(hd.x & 0xFFFFu), rc.lb = (int16_t)(hd.x >> 16), rc.ra = (int16_t)(hd.y & 0xFFFFu), rc.rb = (int16_t)(hd.y >> 16);
            const uint32_t bw32[6] = {hb.x, hb.y, hb.z, hb.w, hc.x, hc.y};
            int bx0[6], by0[6], bw[6], pre[7];
            pre[0] = 0;
            int xmin = R, xmax = 0;
#pragma unroll
            for (int k = 0; k < 6; k++) {
                const int x0 = bw32[k] & 255u, y0 = (bw32[k] >> 8) & 255u, w = (bw32[k] >> 16) & 255u, hh = bw32[k] >> 24;
                const int ylo = max(y0, r0), yhi = min(y0 + hh, r1 + 1), cnt = w * max(yhi - ylo, 0);
                bx0[k] = x0, by0[k] = ylo, bw[k] = w;
                pre[k + 1] = pre[k] + cnt;
                if (cnt > 0) xmin = min(xmin, x0), xmax = max(xmax, x0 + w);
            }
            if (lane <= r1 - r0) rowpack[lane] = row_pack<MAXT>(tabs, q.t, rc, r0 + lane);
            for (int dx = xmin + lane; dx < xmax; dx += 64) colpack[dx] = col_pack<MAXT>(tabs, q.t, rc, dx);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            const int npx = __builtin_amdgcn_readfirstlane(pre[6]);
            for (int p = lane; p < npx; p += 64) {
                int x0 = bx0[0], y0 = by0[0], w = bw[0], base = 0;
#pragma unroll
                for (int k = 1; k < 6; k++)
                    if (p >= pre[k]) x0 = bx0[k], y0 = by0[k], w = bw[k], base = pre[k];
                const int o = p - base;
                const int yy = (int)(((float)o + 0.5f) * (1.0f / (float)max(w, 1)));  // o / w, exact for these sizes
                const int dy = y0 + yy, dx = x0 + (o - yy * w);
                const int idx = dy * R + dx - P0;
                if ((unsigned)idx < (unsigned)(P1 - P0)) tl[idx + l0 * 16] = eval_sep<MAXT>(tabs, q.t, R, rowpack[dy - r0], colpack[dx], dy, dx);
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        if (F32OUT) store_block_f32(tl, reinterpret_cast<float4 *>(obs), (int64_t)gl[i] - lane, lane, total);
        else if (active) *out = reinterpret_cast<const uint4 *>(tl)[lane];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");  // the LDS block is reused by the wave's next block
    }
}

void pong_gray_print_ticks() {
    static unsigned long long rows[1024][8];
    if (hipMemcpyFromSymbol(rows, HIP_SYMBOL(g_gray_ticks), sizeof(rows)) != hipSuccess) return;
    unsigned long long t[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int r = 0; r < 1024; r++)
        for (int i = 0; i < 8; i++) t[i] += rows[r][i];
    if (!t[7]) return;
    fprintf(stderr, "gray env kernel, mean cycles per wavefront (all its tiles) over %llu wavefronts: loop top/ring words %llu | template issue + boxes %llu | "
            "row/col words %llu | LDS fill %llu | patch %llu | stream-out %llu\n", t[7], t[0] / t[7], t[1] / t[7], t[2] / t[7], t[3] / t[7], t[4] / t[7], t[5] / t[7]);
}

#endif  // CRL_ABLATION

// ---------------------------------------------------------------------------------------
// CRL_OBS_F32_REF: the reference's own float32 observation (include/crl.h).  During step() WarpFrame.parse_single_frame
// (utils/atari_wrappers.py:215-219) receives MaxAndSkipEnv's FLOAT32 max frame (the buffers take the Box dtype, :104-116), so
// cv2.cvtColor computes gray = R * 0.299f + G * 0.587f + B * 0.114f in float32 and cv2.resize(INTER_AREA) returns the UNROUNDED
// area average of that; reset() and the auto-reset of a finished env go through the uint8 image (rounded).  A plane whose two
// kept frames are the same frame is such a reset observation (consecutive frames of a running game differ in the ball's x).
// Every tap is evaluated from the frame descriptors in OpenCV's accumulation order (eval_pixel's).  This is the exact mode, not
// the fast one (round 4: 7.6 ms per step at 65 536 envs against 2.7 ms for the widened uint8 values, CRL_OBS_F32; round 5: DESIGN.md 7).
__device__ inline float gray_of_f32(int v) {
    const float f = (float)v;
    return f * 0.299f + f * 0.587f + f * 0.114f;  // (one rounding per operation: -ffp-contract=off)
}

// one output pixel of that path, OpenCV's accumulation order (eval_pixel's); rounded: a reset observation (the uint8 image)
__device__ inline float f32ref_pixel(const GrayCtx &g, const Frame &fa, const Frame &fb, int view, int dy, int dx, bool rounded) {
    const int j0 = g.yofs[dy], j1 = g.yofs[dy + 1];
    const int k0 = g.xofs[dx], k1 = g.xofs[dx + 1];
    float sum = 0.f;
    for (int j = j0; j < j1; j++) {
        const int r = g.ysi[j];
        float buf = 0.f;
        for (int k = k0; k < k1; k++) {
            const int c = g.xsi[k];
            const int sv = max(px_view(fa, g.atlas_gray, view, r, c), px_view(fb, g.atlas_gray, view, r, c));
            buf = buf + (rounded ? (float)sv : gray_of_f32(sv)) * g.xalpha[k];
        }
        const float tj = g.yalpha[j] * buf;
        sum = (j == j0) ? tj : sum + tj;
    }
    if (rounded) {
        const int v = (int)rintf(sum);
        sum = (float)min(max(v, 0), 255);
    }
    return sum;
}

// a frame with the given scores whose ball and bats are parked in the middle of the court: what the top / bottom band tables are
// drawn from (an output pixel of a real plane differs from them only where its taps touch a ball or a bat: those are re-drawn)
__device__ inline Frame parked_frame(int sl, int sr) {
    Frame f;
    f.sl = sl, f.sr = sr, f.x = 78, f.y = 112, f.bl = 100, f.br = 100;
    return f;
}

__global__ __launch_bounds__(256) void pong_gray_f32ref_table_kernel(GrayCtx g, int R, int band_rows, int bot0, float *__restrict__ top,
                                                                     float *__restrict__ bot) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    // top[pair][kind][view][rounded][band_rows][R]; kind 0: both kept frames show the pair; 1 / 2: one of them shows the left / right score
    // one higher (a point was scored between the two frames: the text under the max is the union of both -- round 4 evaluated every
    // pixel of such a plane, 2.5 % of the tiles of a random-action batch at 40-100 x the cost of the others: 6.4 of the kernel's 7.6 ms)
    const int64_t ntop = (int64_t)484 * 3 * 2 * 2 * band_rows * R, nbot = (int64_t)2 * (R - bot0) * R;
    if (i < ntop) {
        const int dx = (int)(i % R), dy = (int)(i / R % band_rows), rnd = (int)(i / ((int64_t)R * band_rows) % 2);
        const int view = (int)(i / ((int64_t)R * band_rows * 2) % 2), kind = (int)(i / ((int64_t)R * band_rows * 4) % 3);
        const int pair = (int)(i / ((int64_t)R * band_rows * 12)), sl = pair / 22, sr = pair % 22;
        const Frame f = parked_frame(sl, sr);
        const int sl2 = sl + (kind == 1), sr2 = sr + (kind == 2);
        if (sl2 > 21 || sr2 > 21) return;  // (no such score; the slot is never read)
        const Frame f2 = parked_frame(sl2, sr2);
        top[i] = f32ref_pixel(g, f, f2, view, dy, dx, rnd != 0);
    } else if (i < ntop + nbot) {
        const int64_t q = i - ntop;
        const int dx = (int)(q % R), dy = bot0 + (int)(q / R % (R - bot0)), rnd = (int)(q / ((int64_t)R * (R - bot0)));
        const Frame f = parked_frame(0, 0);
        bot[q] = f32ref_pixel(g, f, f, 0, dy, dx, rnd != 0);
    }
}

void launch_pong_gray_f32ref_tables(const GrayParams &p, float *top, float *bot, hipStream_t st) {
    GrayCtx g = {p.atlas_gray, p.xofs, p.yofs, p.xsi, p.ysi, p.xalpha, p.yalpha};
    const int64_t total = (int64_t)484 * 12 * p.band_rows * p.R + (int64_t)2 * (p.R - p.f32_bot0) * p.R;
    hipLaunchKernelGGL(pong_gray_f32ref_table_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, g, p.R, p.band_rows, p.f32_bot0, top, bot);
}

// One wavefront per (env, view, plane) tile.  Usual case (both kept frames show the same scores): the plane is the score pair's
// band rows + an empty court + the white band, copied, and the few dozen output pixels whose taps touch the ball or a bat of
// either frame, re-drawn exactly -- every other pixel sees the same source values as in the table.  A point scored between the
// two frames puts two score texts under the max: the tables hold that union too (kind 1 / 2: left / right score one higher in one
// frame).  Only unrelated score pairs (a set_state can produce them) evaluate every pixel.  The tap tables are staged in LDS once per
// workgroup (an output pixel's evaluation is a chain of ~25 dependent table reads).
struct F32RefGeom {
    const float *top, *bot;
    int band_rows, bot0;
    const uint8_t *x_first, *x_last, *y_first, *y_last;
    int xtaps, ytaps;  // entries of the x / y tap tables
    int debug;         // profiling build: 16 = no re-draw, 32 = no table copy (WRONG pixels: what each phase costs)
    int map_row0, map_rows;  // output rows fed by the court's source rows [CRL_PONG_TOP, CRL_PONG_BOTTOM)
};
static constexpr int kF32MaxR = 84, kF32MaxTaps = 3 * kF32MaxR + 8;
static constexpr int kF32MapBytes = 66 * 84;  // the output rows a court rectangle can feed (R = 84: rows 13..77), one byte per pixel
#ifndef CRL_F32REF_LB
#define CRL_F32REF_LB 5  // workgroups per CU the register allocation must allow (round 6: 102 -> 96 VGPRs, 4 -> 5 wavefronts per SIMD: 2 795 -> 2 698 us; 30 KB of LDS per workgroup allow no more)
#endif
__global__ __launch_bounds__(256, CRL_F32REF_LB) void pong_gray_f32ref_kernel(const uint64_t *__restrict__ ring, int64_t n, GrayCtx gg, F32RefGeom q, int R, int K, int views,
                                                               float *__restrict__ obs, GrayStack sk) {
    __shared__ int32_t s_xofs[kF32MaxR + 1], s_yofs[kF32MaxR + 1], s_xsi[kF32MaxTaps], s_ysi[kF32MaxTaps];
    __shared__ float s_xalpha[kF32MaxTaps], s_yalpha[kF32MaxTaps];
    __shared__ __attribute__((aligned(16))) uint8_t s_map[4][kF32MapBytes];  // per wavefront: court pixel -> index of its re-drawn value (255: none)
    __shared__ float s_pval[4][192];
    if (q.debug & 64) return;
    GrayCtx g = gg;
    if (R <= kF32MaxR && q.xtaps <= kF32MaxTaps && q.ytaps <= kF32MaxTaps) {  // (uniform)
        for (int i = threadIdx.x; i <= R; i += 256) s_xofs[i] = gg.xofs[i], s_yofs[i] = gg.yofs[i];
        for (int i = threadIdx.x; i < q.xtaps; i += 256) s_xsi[i] = gg.xsi[i], s_xalpha[i] = gg.xalpha[i];
        for (int i = threadIdx.x; i < q.ytaps; i += 256) s_ysi[i] = gg.ysi[i], s_yalpha[i] = gg.yalpha[i];
        g.xofs = s_xofs, g.yofs = s_yofs, g.xsi = s_xsi, g.ysi = s_ysi, g.xalpha = s_xalpha, g.yalpha = s_yalpha;
    }
    __syncthreads();
    if (q.debug & 128) return;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t tile = (int64_t)blockIdx.x * 4 + wave;
    // jobs of an env: the planes of a bound FrameStackTensor first (GrayStack, pong_device.h), then the observation's tiles
    const int ks = sk.out ? sk.k : 0, tiles_per_env = ks + (obs ? views * K - (ks && sk.alias ? 1 : 0) : 0);
    if (tile >= n * tiles_per_env) return;
    const int64_t env = tile / tiles_per_env;
    int t = (int)(tile - env * tiles_per_env), view, plane, rp;
    float *out;
    bool erased = false;
    if (t < ks) {
        view = sk.view, plane = t, rp = 4 - ks + t, erased = t < ks - sk.valid;
        out = reinterpret_cast<float *>(sk.out) + (env * ks + t) * (int64_t)(R * R);
    } else {
        t -= ks;
        if (ks && sk.alias && t >= sk.view * K + K - 1) t++;
        view = t / K, plane = t - view * K, rp = 4 - K + plane;
        out = obs + ((env * views + view) * K + plane) * (int64_t)(R * R);
    }
    const uint64_t pa = ring[(int64_t)(2 * rp) * n + env], pb = ring[(int64_t)(2 * rp + 1) * n + env];
    Frame fa = unpack_frame(pa), fb = unpack_frame(pb);
    if ((fa.sl == 255 && fb.sl == 255) || erased) {  // a plane that was never written, erased by a done, or older than the bound stack's reset()
        for (int i = lane; i < R * R; i += 64) out[i] = 0.0f;
        return;
    }
    const bool rounded = pa == pb || fa.sl == 255 || fb.sl == 255;  // a reset observation: the uint8 path
    if (fa.sl == 255) fa = fb;
    else if (fb.sl == 255) fb = fa;
    // which table: the two frames' score pair, or a pair and its successor (the frames in either order)
    const int slo = min(fa.sl, fb.sl), sro = min(fa.sr, fb.sr), dl = abs(fa.sl - fb.sl), dr = abs(fa.sr - fb.sr);
    const int kind = (dl == 0 && dr == 0) ? 0 : (dl == 1 && dr == 0) ? 1 : (dl == 0 && dr == 1) ? 2 : -1;
    if (kind < 0 || max(fa.sl, fb.sl) > 21 || max(fa.sr, fb.sr) > 21 || q.band_rows > q.bot0) {  // unrelated score texts (only a set_state produces them) or a tiny R: every pixel
        for (int i = lane; i < R * R; i += 64) out[i] = f32ref_pixel(g, fa, fb, view, i / R, i - (i / R) * R, rounded);
        return;
    }
    // ball, left bat, right bat of both frames: source rectangle -> the output pixels it feeds; one list over the six rectangles
    // (where two of them overlap a pixel is drawn twice, to the same value)
    int rdx0[6], rwx[6], rdy0[6], rend[6], total = 0;
#pragma unroll
    for (int o = 0; o < 6; o++) {
        const Frame &f = o < 3 ? fa : fb;
        const int k = o % 3;
        int x0 = k == 0 ? f.x : k == 1 ? CRL_PONG_BATL_X : CRL_PONG_BATR_X, w = k == 0 ? CRL_PONG_BALL : CRL_PONG_BAT_W;
        int y0 = k == 0 ? f.y : k == 1 ? f.bl : f.br, h = k == 0 ? CRL_PONG_BALL : CRL_PONG_BAT_H;
        int x1 = min(x0 + w - 1, CRL_PONG_W - 1), y1 = min(y0 + h - 1, CRL_PONG_BOTTOM - 1);
        x0 = max(x0, 0), y0 = max(y0, CRL_PONG_TOP);  // (px_view draws the court's rows only)
        int npx = 0;
        rdx0[o] = rwx[o] = rdy0[o] = 0;
        const bool twice = o >= 3 && (k == 0 ? (fb.x == fa.x && fb.y == fa.y) : k == 1 ? fb.bl == fa.bl : fb.br == fa.br);  // the same rectangle as in frame a
        if (x0 <= x1 && y0 <= y1 && !twice) {
            if (view == 1) {  // court rows are mirrored in the second view
                const int m0 = CRL_PONG_W - 1 - x1, m1 = CRL_PONG_W - 1 - x0;
                x0 = m0, x1 = m1;
            }
            const int dx0 = q.x_first[x0], dx1 = q.x_last[x1], dy0 = q.y_first[y0], dy1 = q.y_last[y1];
            rdx0[o] = dx0, rwx[o] = dx1 - dx0 + 1, rdy0[o] = dy0, npx = rwx[o] * (dy1 - dy0 + 1);
        }
        total += npx;
        rend[o] = total;
    }
    // The re-drawn pixels are evaluated FIRST, into registers (round 5: an output pixel is a chain of ~25 dependent LDS reads, 2-3 us for a
    // lone wavefront; the table loads and the stores of the rows no rectangle touches run beside it), and -- round 6 -- PATCHED INTO THE
    // PIECES before these are stored: every lane drops its pixels into a byte map of the court's rows in LDS (pixel -> index of its value),
    // and the copy of the rows the rectangles touch looks its pieces up there.  Round 5 stored the table copy and then over-stored the
    // ~150 pixels as 4-byte writes behind a fence: 18.1 GB written for 14.8 GB of tensor (1.36 x with the table reads); now every byte of
    // the tile is written exactly once, in 16-byte pieces.
    // A lane holds at most kRedraw pixels (the six rectangles cover < 64 kRedraw output pixels at every R up to 84); what does not fit is
    // drawn behind the stores as before.
    constexpr int kRedraw = 3;
    int rpos[kRedraw];
    float rval[kRedraw];
    int ymin = R, ymax = 0;  // output rows the rectangles touch (uniform)
#pragma unroll
    for (int o = 0; o < 6; o++)
        if (rwx[o] > 0) ymin = min(ymin, rdy0[o]), ymax = max(ymax, rdy0[o] + (rend[o] - (o ? rend[o - 1] : 0)) / rwx[o]);
#pragma unroll
    for (int k = 0; k < kRedraw; k++) rpos[k] = -1, rval[k] = 0.0f;
#pragma unroll
    for (int k = 0; k < kRedraw; k++) {
        const int i = lane + 64 * k;
        if (i >= total || (q.debug & 16)) continue;
        int o = 0, base = 0;
#pragma unroll
        for (int m = 0; m < 5; m++)
            if (i >= rend[m]) o = m + 1, base = rend[m];
        int dx0 = rdx0[0], wx = rwx[0], dy0 = rdy0[0];
#pragma unroll
        for (int m = 1; m < 6; m++)
            if (o == m) dx0 = rdx0[m], wx = rwx[m], dy0 = rdy0[m];
        const int j = i - base, dy = dy0 + j / wx, dx = dx0 + j % wx;
        rpos[k] = dy * R + dx, rval[k] = f32ref_pixel(g, fa, fb, view, dy, dx, rounded);
    }
    const float *top = q.top + (((((int64_t)(slo * 22 + sro) * 3 + kind) * 2 + view) * 2 + (rounded ? 1 : 0)) * q.band_rows) * R;
    const float *bot = q.bot + (int64_t)(rounded ? 1 : 0) * (R - q.bot0) * R;
    const int ntop = q.band_rows * R, nbot0 = q.bot0 * R;
    // the map covers output rows [map_row0, map_row0 + map_rows): every row a court rectangle can feed
    const bool pieces = ((ntop | nbot0 | (R * R) | R) & 3) == 0 && q.map_rows * R <= kF32MapBytes && !(q.debug & (16 | 32));  // (uniform; R = 84: always)
    if (pieces) {
        uint8_t *map = s_map[wave];
        float *pval = s_pval[wave];
        const int map0 = q.map_row0 * R;  // first pixel of the map
        ymin = max(ymin, q.map_row0), ymax = min(max(ymax, ymin), q.map_row0 + q.map_rows);
        // clear the rows the rectangles touch (0xFF = not re-drawn), then drop the re-drawn pixels in
        const int c0 = (ymin * R - map0) >> 2, c1 = (ymax * R - map0) >> 2;
        for (int c = c0 + lane; c < c1; c += 64) reinterpret_cast<uint32_t *>(map)[c] = 0xFFFFFFFFu;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
#pragma unroll
        for (int k = 0; k < kRedraw; k++)
            if (rpos[k] >= 0) {
                pval[lane + 64 * k] = rval[k];
                map[rpos[k] - map0] = (uint8_t)(lane + 64 * k);  // (two rectangles on one pixel: the same value twice, either index serves)
            }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        const float4 *top4 = reinterpret_cast<const float4 *>(top), *bot4 = reinterpret_cast<const float4 *>(bot);
        float4 *out4 = reinterpret_cast<float4 *>(out);
        const int nt4 = ntop >> 2, nb4 = nbot0 >> 2, nn4 = (R * R) >> 2;
        const int p0 = (ymin * R) >> 2, p1 = (ymax * R) >> 2;  // pieces of the touched rows
        auto piece = [&](int i) -> float4 {  // piece i of the plane without ball and bats
            if (i < nt4) return top4[i];
            if (i >= nb4) return bot4[i - nb4];
            return make_float4(0.f, 0.f, 0.f, 0.f);
        };
        constexpr int kU = 7;  // pieces in flight per lane: all loads of a batch, then its stores
        // 1. the rows no rectangle touches: table (or zero) pieces straight out -- these stores run beside the chain above
        for (int i0 = lane; i0 < nn4; i0 += 64 * kU) {
            float4 v[kU];
#pragma unroll
            for (int k = 0; k < kU; k++) {
                const int i = i0 + 64 * k;
                if (i < nn4 && (i < p0 || i >= p1)) v[k] = piece(i);
            }
#pragma unroll
            for (int k = 0; k < kU; k++) {
                const int i = i0 + 64 * k;
                if (i < nn4 && (i < p0 || i >= p1)) out4[i] = v[k];
            }
        }
        // 2. the touched rows: the same pieces with the re-drawn pixels put in
        for (int i = p0 + lane; i < p1; i += 64) {
            float4 v = piece(i);
            const uint32_t mw = reinterpret_cast<const uint32_t *>(map)[i - (map0 >> 2)];
            if (mw != 0xFFFFFFFFu) {
                if ((mw & 255u) != 255u) v.x = pval[mw & 255u];
                if (((mw >> 8) & 255u) != 255u) v.y = pval[(mw >> 8) & 255u];
                if (((mw >> 16) & 255u) != 255u) v.z = pval[(mw >> 16) & 255u];
                if ((mw >> 24) != 255u) v.w = pval[mw >> 24];
            }
            out4[i] = v;
        }
    } else {
        if (!(q.debug & 32)) {
            for (int i = lane; i < ntop; i += 64) out[i] = top[i];
            for (int i = ntop + lane; i < nbot0; i += 64) out[i] = 0.0f;
            for (int i = nbot0 + lane; i < R * R; i += 64) out[i] = bot[i - nbot0];
        }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");  // the re-drawn pixels below overwrite what other lanes have just stored: those stores first
        if (q.debug & 16) return;
#pragma unroll
        for (int k = 0; k < kRedraw; k++)
            if (rpos[k] >= 0) out[rpos[k]] = rval[k];
    }
    if (total > 64 * kRedraw) {  // (more output pixels than the registers hold: not at the sizes the wrappers use)
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
        for (int i = lane + 64 * kRedraw; i < total; i += 64) {
            int o = 0, base = 0;
#pragma unroll
            for (int k = 0; k < 5; k++)
                if (i >= rend[k]) o = k + 1, base = rend[k];
            int dx0 = rdx0[0], wx = rwx[0], dy0 = rdy0[0];
#pragma unroll
            for (int k = 1; k < 6; k++)
                if (o == k) dx0 = rdx0[k], wx = rwx[k], dy0 = rdy0[k];
            const int j = i - base, dy = dy0 + j / wx, dx = dx0 + j % wx;
            out[dy * R + dx] = f32ref_pixel(g, fa, fb, view, dy, dx, rounded);
        }
    }
}

void launch_pong_gray_templates(const GrayParams &p, const uint8_t *x_first, const uint8_t *x_last, const uint8_t *y_first,
                                const uint8_t *y_last, int band_rows, int band_chunks, uint8_t *band, uint8_t *rest,
                                hipStream_t st) {
    GrayCtx g = {p.atlas_gray, p.xofs, p.yofs, p.xsi, p.ysi, p.xalpha, p.yalpha};
    GrayGeom q = {};
    q.R = p.R, q.K = p.K, q.band_rows = band_rows, q.band_chunks = band_chunks;
    const int total = 3 * 484 * 2 * band_chunks * 16 + p.R * p.R;
    hipLaunchKernelGGL(pong_gray_template_kernel, dim3((total + 255) / 256), dim3(256), 0, st, g, q, band, rest);
}

void launch_pong_raster_gray_ex(const GrayParams &p, const uint8_t *rest, int zero_row0, int zero_row1,
                                const uint8_t *x_first, const uint8_t *x_last, const uint8_t *y_first,
                                const uint8_t *y_last, int band_chunks, const uint8_t *tab_blob, const GrayTabOfs &tofs,
                                hipStream_t st) {
    if (p.n <= 0) return;
    GrayCtx g = {p.atlas_gray, p.xofs, p.yofs, p.xsi, p.ysi, p.xalpha, p.yalpha};
    if (p.obs_f32 == 2) {  // CRL_OBS_F32_REF
        const int views = p.views > 0 ? p.views : 2;
        const int64_t tiles = p.n * ((p.stack.out ? p.stack.k : 0) + (p.obs ? views * p.K - (p.stack.out && p.stack.alias ? 1 : 0) : 0));
        static const int dbg = CRL_ABL(getenv("CRL_GRAY_DEBUG") != nullptr) ? atoi(getenv("CRL_GRAY_DEBUG")) : 0;
        const F32RefGeom fq = {p.f32_top, p.f32_bot, p.band_rows, p.f32_bot0, x_first, x_last, y_first, y_last, p.f32_xtaps, p.f32_ytaps, dbg,
                               p.f32_map_row0, p.f32_map_rows};
        hipLaunchKernelGGL(pong_gray_f32ref_kernel, dim3((unsigned)((tiles + 3) / 4)), dim3(256), 0, st, p.ring, p.n, g, fq, p.R, p.K, views,
                           reinterpret_cast<float *>(p.obs), p.stack);
        return;
    }
    GrayGeom q;
    q.R = p.R, q.K = p.K, q.views = p.views > 0 ? p.views : 2, q.band_rows = p.band_rows, q.band_chunks = band_chunks;
    q.band = p.band, q.rest = rest, q.zero_row0 = zero_row0, q.zero_row1 = zero_row1;
    q.x_first = x_first, q.x_last = x_last, q.y_first = y_first, q.y_last = y_last;
    q.tab_blob = tab_blob, q.t = tofs;
    q.debug = 0;
#ifdef CRL_ABLATION
    const int64_t tiles = p.n * q.views * p.K;
    // profiling build only: CRL_GRAY_DEBUG bits skip phases of the env kernel (WRONG pixels) or select the first tile kernel (8);
    // CRL_GRAY_SWEEP=1 the address-linear writer (bit-exact, slower), 2 | 3 its skeletons (wrong pixels)
    {
        static const int dbg = getenv("CRL_GRAY_DEBUG") ? atoi(getenv("CRL_GRAY_DEBUG")) : 0;
        q.debug = dbg;
    }
    if ((q.debug & 8) && !p.obs_f32 && !p.stack.out) {
        hipLaunchKernelGGL(pong_raster_gray_kernel, dim3((unsigned)((tiles + 3) / 4)), dim3(256), 0, st, p.ring, p.n, g, q,
                           p.obs);
        return;
    }
    // address-linear writer (uint8 output, 16-byte-aligned tiles, three-tap tables: R = 84 and similar sizes):
    // bit-exact, but 1.0-2.0 ms against the env kernel's 0.75 ms at 65 536 envs -- every block has to read per-tile metadata
    // first, and a dependent read in a store-saturated memory system takes microseconds.
    static const int sweep_env = getenv("CRL_GRAY_SWEEP") ? atoi(getenv("CRL_GRAY_SWEEP")) : 0;
    if (sweep_env && p.obs_f32 != 2 && (!p.obs_f32 || p.R == 84) && !p.stack.out && !q.debug && p.hdr && (p.R * p.R) % 16 == 0 && tofs.max_taps <= 3 && tofs.fast_ok && tiles * (p.R * p.R >> 4) < (1ll << 31) && (p.R * p.R >> 4) <= 512) {
        GrayTileHdr *hdr = reinterpret_cast<GrayTileHdr *>(p.hdr);
        hipLaunchKernelGGL(pong_gray_header_kernel, dim3((unsigned)((tiles + 255) / 256)), dim3(256), 0, st, p.ring, p.n, q, hdr);
        static const int sdbg = getenv("CRL_GRAY_SWEEP_DEBUG") ? atoi(getenv("CRL_GRAY_SWEEP_DEBUG")) : 0;
        static const int nb_env = getenv("CRL_GRAY_SWEEP_NB") ? atoi(getenv("CRL_GRAY_SWEEP_NB")) : 2;
        const int64_t wblocks = (tiles * (p.R * p.R >> 4) + 63) / 64;   // 1-KiB blocks of the whole tensor
        const int nb = wblocks < 8192 ? 1 : nb_env;
        const int stride = (int)(((wblocks + nb - 1) / nb + 3) / 4 * 4);  // blocks per round, a multiple of the 4 waves of a workgroup
        const dim3 grid((unsigned)(stride / 4));
        const int sweep_mode = sweep_env;
        if ((sweep_mode == 2 || sweep_mode == 3 || sweep_mode == 4) && p.R == 84) {  // skeletons: 64-byte headers | dense 8-byte records | dense 2-byte records
            const int dense = sweep_mode - 2;
#define CRL_SKEL(NBv) do { if (p.obs_f32) hipLaunchKernelGGL((pong_gray_sweep_skeleton_kernel<NBv, true>), grid, dim3(256), 0, st, hdr, (int)tiles, q, p.obs, stride, dense); \
                           else hipLaunchKernelGGL((pong_gray_sweep_skeleton_kernel<NBv, false>), grid, dim3(256), 0, st, hdr, (int)tiles, q, p.obs, stride, dense); } while (0)
            if (nb == 1) CRL_SKEL(1);
            else if (nb == 2) CRL_SKEL(2);
            else CRL_SKEL(4);
#undef CRL_SKEL
            return;
        }
        if (p.obs_f32) {  // float32 output (round 6): R = 84 only
            if (nb == 1) hipLaunchKernelGGL((pong_raster_gray_sweep_kernel<84, 1, true>), grid, dim3(256), 0, st, hdr, (int)tiles, g, q, p.obs, sdbg, p.ring, p.n, stride);
            else if (nb == 2) hipLaunchKernelGGL((pong_raster_gray_sweep_kernel<84, 2, true>), grid, dim3(256), 0, st, hdr, (int)tiles, g, q, p.obs, sdbg, p.ring, p.n, stride);
            else hipLaunchKernelGGL((pong_raster_gray_sweep_kernel<84, 4, true>), grid, dim3(256), 0, st, hdr, (int)tiles, g, q, p.obs, sdbg, p.ring, p.n, stride);
            return;
        }
#define CRL_SWEEP(RTv, NBv) hipLaunchKernelGGL((pong_raster_gray_sweep_kernel<RTv, NBv>), grid, dim3(256), 0, st, hdr, (int)tiles, g, q, p.obs, sdbg, p.ring, p.n, stride)
        if (p.R == 84) {
            if (nb == 1) CRL_SWEEP(84, 1);
            else if (nb == 2) CRL_SWEEP(84, 2);
            else if (nb == 8) CRL_SWEEP(84, 8);
            else CRL_SWEEP(84, 4);
        } else {
            if (nb == 1) CRL_SWEEP(0, 1);
            else CRL_SWEEP(0, 4);
        }
#undef CRL_SWEEP
        return;
    }
#endif  // CRL_ABLATION
    // planes per wave: the whole stack of an env per wave once there are enough envs to fill
    // the chip (256 CUs x 16 waves), one plane per wave below that
    static const int ppw_env = CRL_ABL(getenv("CRL_GRAY_PPW") ? atoi(getenv("CRL_GRAY_PPW")) : 0);  // tuning experiments (profiling build)
    int ppw = (p.n >= 8192 && !(q.debug & 16)) ? p.K : 1;
    if (ppw_env > 0 && p.K % ppw_env == 0) ppw = ppw_env;
    static const bool small_off = CRL_ABL(getenv("CRL_GRAY_SMALL_OFF") != nullptr);  // A/B: the R <= 45 instance off
    const GrayStack &sk = p.stack;
    if (sk.out) {  // FrameStackTensor fused into the draw: the env's jobs = the stack's planes + the observation's tiles
        const int jobs = sk.k + (p.obs ? q.views * p.K - (sk.alias ? 1 : 0) : 0);
        static const int jpw_env = CRL_ABL(getenv("CRL_GRAY_JPW") != nullptr) ? atoi(getenv("CRL_GRAY_JPW")) : 0;
        // jobs per wavefront: a uint8 stack is fastest with the whole env in one wavefront (497 us; two / three jobs: 655 / 576); a float32
        // stack with two (five alternating pairs, tools/ab/r06_stack_jpw2.sh: - 1.2 ... - 7.5 % against the whole env, mean - 3 %)
        int jpw = p.n >= 8192 ? (sk.f32 ? 2 : jobs) : 1;
        if (jpw_env > 0) jpw = jpw_env;
        if (jpw_env < 0) jpw = jobs;  // (profiling build: CRL_GRAY_JPW=-1 = all of the env's jobs)
        const int64_t waves = p.n * ((jobs + jpw - 1) / jpw);
        const dim3 grid((unsigned)((waves + 3) / 4));
        const int ti = tofs.max_taps <= 3 ? 0 : (p.R * p.R <= 2048 && !small_off) ? 1 : 2;
#define CRL_STACK_LAUNCH(MT, TIv, OF, SF) \
    hipLaunchKernelGGL((pong_raster_gray_env_kernel<MT, false, TIv, OF, true, SF>), grid, dim3(256), 0, st, p.ring, p.n, g, q, p.obs, jpw, sk)
#define CRL_STACK_TI(OF, SF)                           \
    do {                                               \
        if (ti == 0) CRL_STACK_LAUNCH(3, 7, OF, SF);   \
        else if (ti == 1) CRL_STACK_LAUNCH(5, 2, OF, SF); \
        else CRL_STACK_LAUNCH(5, 7, OF, SF);           \
    } while (0)
        if (p.obs_f32) CRL_STACK_TI(true, true);  // (a float32 context's stack is float32: launch_pong_raster_gray's caller checks)
        else if (sk.f32) CRL_STACK_TI(false, true);
        else CRL_STACK_TI(false, false);
#undef CRL_STACK_TI
#undef CRL_STACK_LAUNCH
        return;
    }
    const GrayStack nosk{};
    const int64_t waves = p.n * (p.K / ppw);
    const dim3 grid((unsigned)((waves + 3) / 4));
    if (p.obs_f32) {
        if (tofs.max_taps <= 3)
            hipLaunchKernelGGL((pong_raster_gray_env_kernel<3, false, 7, true>), grid, dim3(256), 0, st, p.ring, p.n, g, q, p.obs, ppw, nosk);
        else if (p.R * p.R <= 2048 && !small_off)
            hipLaunchKernelGGL((pong_raster_gray_env_kernel<5, false, 2, true>), grid, dim3(256), 0, st, p.ring, p.n, g, q, p.obs, ppw, nosk);
        else
            hipLaunchKernelGGL((pong_raster_gray_env_kernel<5, false, 7, true>), grid, dim3(256), 0, st, p.ring, p.n, g, q, p.obs, ppw, nosk);
        return;
    }
#ifdef CRL_ABLATION
    if (q.debug & ~16) {  // any ablation switch: the instrumented instance
        if (tofs.max_taps <= 3) hipLaunchKernelGGL((pong_raster_gray_env_kernel<3, true, 7, false>), grid, dim3(256), 0, st, p.ring, p.n, g, q, p.obs, ppw, nosk);
        else hipLaunchKernelGGL((pong_raster_gray_env_kernel<5, true, 2, false>), grid, dim3(256), 0, st, p.ring, p.n, g, q, p.obs, ppw, nosk);
        return;
    }
#endif
    // one plane per agent (the observation make_envs("cPongDouble-v0") itself returns) in a batch that fills the chip several times over: four
    // envs per wavefront (CRL_GRAY_EPWV=1 in the profiling build: one)
    static const int epwv_env = CRL_ABL(getenv("CRL_GRAY_EPWV") ? atoi(getenv("CRL_GRAY_EPWV")) : 0);
#ifndef CRL_GRAY_EPWV_N
#define CRL_GRAY_EPWV_N 4  // (measured at 65 536 envs, 84 x 84: 1 / 4 envs per wavefront 237 / 221 us)
#endif
    if (p.K == 1 && ppw == 1 && p.n >= 8192 * CRL_GRAY_EPWV_N && epwv_env != 1 && (tofs.max_taps <= 3 || (p.R * p.R <= 2048 && !small_off))) {
        constexpr int E = CRL_GRAY_EPWV_N;
        const dim3 grid4((unsigned)(((p.n + E - 1) / E + 3) / 4));
        if (tofs.max_taps <= 3)
            hipLaunchKernelGGL((pong_raster_gray_env_kernel<3, false, 7, false, false, false, E>), grid4, dim3(256), 0, st, p.ring, p.n, g, q, p.obs, ppw, nosk);
        else
            hipLaunchKernelGGL((pong_raster_gray_env_kernel<5, false, 2, false, false, false, E>), grid4, dim3(256), 0, st, p.ring, p.n, g, q, p.obs, ppw, nosk);
        return;
    }
    if (tofs.max_taps <= 3)
        hipLaunchKernelGGL((pong_raster_gray_env_kernel<3, false, 7, false>), grid, dim3(256), 0, st, p.ring, p.n, g, q, p.obs, ppw, nosk);
    else if (p.R * p.R <= 2048 && !small_off)
        hipLaunchKernelGGL((pong_raster_gray_env_kernel<5, false, 2, false>), grid, dim3(256), 0, st, p.ring, p.n, g, q, p.obs, ppw, nosk);
    else
        hipLaunchKernelGGL((pong_raster_gray_env_kernel<5, false, 7, false>), grid, dim3(256), 0, st, p.ring, p.n, g, q, p.obs, ppw, nosk);
}

}  // namespace crl
