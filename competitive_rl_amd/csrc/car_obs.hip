// car_obs.hip -- cCarRacingDouble observations the way the reference computes them: (N, players, 96, 96) uint8.
//
// Restates CarRacing.get_observation (reference car_racing/car_racing_multi_players.py:622-634):
//   reset  render_road_for_observation_map (:732-755, called at :519): grass, the range(-20, 20, 2) squares and every
//          road / border polygon are rastered ONCE per episode with pygame.draw.polygon (integer scanline fill) into
//          the env's map -- car_map_build_kernel, into a 4-bit palette map of the window of the 10000^2 surface that
//          can hold anything but grass (car_device.h: 16 x 16-pixel blocks of 128 bytes, 739 328 bytes per env);
//   step   camera_update("rgb_array") :791-804, camera_view :764-789 (192 x 192 crop at the int-truncated camera
//          pixel, pygame.transform.rotate = nearest neighbour in 16.16 fixed point, blit centred on (48, 48)),
//          Car.draw_for_pygame (car_dynamics.py:284-298), render_indicators_for_pygame :645-670, luma truncation:
//          car_camera_kernel (one lane per (env, viewer) tile: the double-precision camera, crop and rotation constants),
//          car_poly_kernel (16 lanes per tile: one lane per car polygon -> its scanline spans; one per indicator rectangle) and
//          car_obs_kernel (ONE wavefront per tile, no workgroup barrier: every pixel gathers its map nibble, then the
//          spans / rectangles / read-out are written over the tile in LDS in draw order and the tile is streamed out).
// pygame 1.9.6 and Box2D are third-party: their rules are restated from the published sources (the CPU checker of tests/
// holds the same restatement and is pinned to frames recorded from the reference's own Python, tests/golden/car_obs.npz).
// sin / cos / atan2 come from include/crl_f64.h and include/crl_rot.h, which that checker evaluates too.
#include <stdio.h>
#include <stdlib.h>

#include "car_device.h"

namespace crl {

#define G_GRASS 161
#define G_LIGHT 176
#define G_WHITE 255
#define G_RED 76
#define G_OWN 60
#define G_OTHER 29
#define G_BLUE 29
#define G_ABS_REAR 44
#define G_GREEN 149
// luma of the palette entries (car_device.h kPal*): grass, light, road 102 / 104 / 107, white, red
static constexpr uint32_t kLutLo = G_GRASS | (G_LIGHT << 8) | (101u << 16) | (103u << 24);
static constexpr uint32_t kLutHi = 107u | ((uint32_t)G_WHITE << 8) | ((uint32_t)G_RED << 16);

// ------------------------------------------------------------------------------------------------ map build
// pygame draw.c draw_fillpoly for ONE scanline y of a polygon with nv <= 5 integer vertices: the sorted crossings.
// Returns their number (0, 2 or 4); the single-scanline polygon (miny == maxy) gives [minx, maxx].
__device__ inline int fillpoly_row(const int *vx, const int *vy, int nv, int miny, int maxy, int minx, int maxx, int y, int xs[4]) {
    if (miny == maxy) {
        xs[0] = minx, xs[1] = maxx;
        return 2;
    }
    int k = 0;
    int t[6];
#pragma unroll
    for (int i = 0; i < 5; i++) {
        if (i < nv) {
            const int ip = i ? i - 1 : nv - 1;
            int y1 = vy[ip], y2 = vy[i], x1 = vx[ip], x2 = vx[i];
            if (y1 > y2) {
                const int ty = y1, tx = x1;
                y1 = y2, x1 = x2, y2 = ty, x2 = tx;
            }
            if (y1 != y2 && ((y >= y1 && y < y2) || (y == maxy && y > y1 && y <= y2))) t[k++] = (y - y1) * (x2 - x1) / (y2 - y1) + x1;
        }
    }
    for (int i = 1; i < k; i++)  // insertion sort (k <= 5)
        for (int j = i; j > 0 && t[j - 1] > t[j]; j--) {
            const int q = t[j];
            t[j] = t[j - 1], t[j - 1] = q;
        }
    k &= ~1;  // pygame draws pairs
    if (k > 4) k = 4;
    for (int i = 0; i < k; i++) xs[i] = t[i];
    return k;
}

// R map rows (16 = a whole block row, or a quarter of one: rows 4 sub .. 4 sub + 3 of block row `band`) of one env.
// rank[R][kMapW]: (draw order << 3 | palette) of the last polygon drawn over each pixel.
// (The quarter-band form is what the step pipeline uses for the few envs it resets per step: a 78 KB workgroup only fits on a CU
// that the frame kernel's 10 KB wavefronts have left half empty, and waited for one -- 300 us for a dozen maps.)
template <int R>
__device__ void car_map_band(const CarSoA &s, int64_t env, int band, int sub, uint32_t (*rank)[kMapW]) {
    const int tid = threadIdx.x;
    const int Y0 = band * 16 + sub * R;
    uint4 *z = reinterpret_cast<uint4 *>(&rank[0][0]);
    for (int i = tid; i < R * kMapW / 4; i += 256) z[i] = make_uint4(0, 0, 0, 0);
    __syncthreads();
    const int nt = s.ntiles[env];
    for (int i = tid; i < nt; i += 256) {
        const uint32_t yr = s.map_yr[env * kCarMaxTiles + i];
        const int ylo = (int)(int16_t)(yr & 0xFFFFu), yhi = (int)(int16_t)(yr >> 16);
        if (yhi < Y0 || ylo >= Y0 + R) continue;
        const uint32_t *v = s.map_vtx + (env * kCarMaxTiles + i) * 9;
        const int border = s.border_em[env * kCarMaxTiles + i];
        for (int poly = 0; poly < 2; poly++) {  // the tile, then its border (drawn right after it)
            if (poly == 1 && !border) break;
            const int nv = poly ? 4 : 5;
            int vx[5], vy[5];
            int miny = 1 << 30, maxy = -(1 << 30), minx = 1 << 30, maxx = -(1 << 30);
#pragma unroll
            for (int j = 0; j < 5; j++) {
                if (j < nv) {
                    const uint32_t w = v[poly * 5 + j];
                    vx[j] = (int)(int16_t)(w & 0xFFFFu), vy[j] = (int)(int16_t)(w >> 16);
                    miny = min(miny, vy[j]), maxy = max(maxy, vy[j]), minx = min(minx, vx[j]), maxx = max(maxx, vx[j]);
                } else {
                    vx[j] = vy[j] = 0;
                }
            }
            // road_poly order (crmp:400-441): i = n-1 .. 0, tile then border: later entries are drawn over earlier ones
            const uint32_t order = (uint32_t)(2 * (nt - 1 - i) + 1 + poly);
            const uint32_t pal = poly ? (border == 1 ? kPalWhite : kPalRed) : (uint32_t)(kPalRoad0 + i % 3);
            const uint32_t key = (order << 3) | pal;
            for (int y = max(miny, Y0); y <= min(maxy, Y0 + R - 1); y++) {
                int xs[4];
                const int k = fillpoly_row(vx, vy, nv, miny, maxy, minx, maxx, y, xs);
                for (int q = 0; q < k; q += 2)
                    for (int x = max(xs[q], 0); x <= min(xs[q + 1], kMapW - 1); x++) atomicMax(&rank[y - Y0][x], key);
            }
        }
    }
    __syncthreads();
    // compose: 16-byte pieces = rows 2 pr, 2 pr + 1 of block bx; the band's 76 blocks are contiguous in the map
    uint4 *out = reinterpret_cast<uint4 *>(env_map(s, env) + (int64_t)band * kMapBlocks * 128);
    for (int q0 = tid; q0 < kMapBlocks * (R / 2); q0 += 256) {
        const int bx = q0 / (R / 2), pr = q0 % (R / 2);  // pr: row pair inside this band
        const int q = bx * 8 + sub * (R / 2) + pr;       // ... and inside the block
        const uint32_t lx = (s.map_lightx[(bx * 16) >> 5] >> ((bx * 16) & 31)) & 0xFFFFu;
        uint32_t w[4];
#pragma unroll
        for (int r = 0; r < 2; r++) {
            const int row = 2 * pr + r, Y = Y0 + row;
            const uint32_t ly = (s.map_lighty[Y >> 5] >> (Y & 31)) & 1u;
            const uint4 *src = reinterpret_cast<const uint4 *>(&rank[row][bx * 16]);
            uint32_t kk[16];
#pragma unroll
            for (int c = 0; c < 4; c++) {
                const uint4 t = src[c];
                kk[4 * c] = t.x, kk[4 * c + 1] = t.y, kk[4 * c + 2] = t.z, kk[4 * c + 3] = t.w;
            }
            uint32_t lo = 0, hi = 0;
#pragma unroll
            for (int x = 0; x < 16; x++) {
                const uint32_t p = kk[x] ? (kk[x] & 7u) : (ly & (lx >> x) & 1u);  // grass 0 / light 1 where no polygon was drawn
                if (x < 8) lo |= p << (4 * x);
                else hi |= p << (4 * (x - 8));
            }
            w[2 * r] = lo, w[2 * r + 1] = hi;
        }
        out[q] = make_uint4(w[0], w[1], w[2], w[3]);
    }
}

__global__ __launch_bounds__(256) void car_map_build_kernel(CarSoA s, const uint8_t *__restrict__ only_env, int64_t env0) {
    __shared__ __attribute__((aligned(16))) uint32_t rank[4][kMapW];  // (quarter bands here too: 8 workgroups per CU instead of 2)
    const int64_t env = env0 + blockIdx.y;
    if (only_env && !only_env[env]) return;
    car_map_band<4>(s, env, (int)blockIdx.x >> 2, (int)blockIdx.x & 3, rank);
}

__global__ __launch_bounds__(256) void car_map_build_list_kernel(CarSoA s, const int32_t *__restrict__ list, const int32_t *__restrict__ list_count) {
    __shared__ __attribute__((aligned(16))) uint32_t rank[4][kMapW];
    const int count = *list_count;
    for (int i = blockIdx.y; i < count; i += gridDim.y) {
        car_map_band<4>(s, (int64_t)list[i], (int)blockIdx.x >> 2, (int)blockIdx.x & 3, rank);
        __syncthreads();
    }
}

void launch_car_map_build(const CarSoA &s, hipStream_t st, const uint8_t *only_env, int64_t first, int64_t count) {
    if (count < 0) count = s.n - first;
    for (int64_t e0 = first; e0 < first + count; e0 += 32768) {  // gridDim.y <= 65 535
        const int64_t m = first + count - e0 < 32768 ? first + count - e0 : 32768;
        hipLaunchKernelGGL(car_map_build_kernel, dim3(kMapBlocks * 4, (unsigned)m), dim3(256), 0, st, s, only_env, e0);
    }
}

void launch_car_map_build_list(const CarSoA &s, hipStream_t st, const int32_t *list, const int32_t *list_count, int64_t expected) {
    int64_t want = expected + expected / 4 + 8;
    want = want > s.n ? s.n : want;
    want = want > 4096 ? 4096 : want;
    hipLaunchKernelGGL(car_map_build_list_kernel, dim3(kMapBlocks * 4, (unsigned)want), dim3(256), 0, st, s, list, list_count);
}

int car_map_coord(double v) { return (int)(CRL_CAR_OBS_SCALE * -v + kMapSurface / 2.0); }

// the squares of render_road_for_observation_map (crmp:735-747): x, y in range(-20, 20, 2), each the polygon
// (k x + k, k y), (k x, k y), (k x, k y + k), (k x + k, k y + k) with k = PLAYFIELD / 20 -- an axis-aligned rectangle whose
// integer fill covers [min, max] of the truncated vertices in both directions, so the lighter pixels are a product set
void car_map_light_masks(uint32_t *lightx, uint32_t *lighty) {
    for (int i = 0; i < kMapW / 32; i++) lightx[i] = lighty[i] = 0u;
    const double k = CAR_PLAYFIELD / 20.0;
    for (int x = -20; x < 20; x += 2) {
        const int a = car_map_coord(k * x + k), b = car_map_coord(k * x + 0);
        for (int p = (a < b ? a : b); p <= (a < b ? b : a); p++) {
            const int w = p - kMapOrg;
            if (w >= 0 && w < kMapW) lightx[w >> 5] |= 1u << (w & 31), lighty[w >> 5] |= 1u << (w & 31);  // same intervals on both axes
        }
    }
}

// ------------------------------------------------------------------------------------------------ per-tile view
__device__ inline uint32_t pack_rect(double x, double y, double w, double h) {
    // pygame.draw.rect(surface, color, (x, y, w, h)) with float arguments: int-truncated, then filled as the polygon
    // (l,t),(r,t),(r,b),(l,b) with r = x + w - 1, b = y + h - 1 (negative sizes fill "backwards"); clipped to the screen
    const int l = (int)x, t = (int)y, r = (int)x + (int)w - 1, b = (int)y + (int)h - 1;
    const int x0 = max(min(l, r), 0), x1 = min(max(l, r), 95), y0 = max(min(t, b), 0), y1 = min(max(t, b), 95);
    if (x0 > x1 || y0 > y1) return 1u;  // x0 = 1 > x1 = 0: empty
    return (uint32_t)x0 | ((uint32_t)x1 << 8) | ((uint32_t)y0 << 16) | ((uint32_t)y1 << 24);
}

// One lane per tile: camera_update("rgb_array") and camera_view's crop / rotation constants -- the double-precision part
// (atan2, sin, cos in double-double), 64 tiles per wavefront.
__device__ __forceinline__ void camera_compute(const CarSoA &s, const CarConsts &K, int64_t env, int viewer, ViewParams &vp, float4 &cam) {
    const int64_t n = s.n, M = (int64_t)s.players * n;
    const int64_t me = viewer * n + env;
    const float h_cx = s.body[0 * M + me], h_cy = s.body[1 * M + me], h_a = s.body[2 * M + me];
    const float h_vx = s.body[3 * M + me], h_vy = s.body[4 * M + me];
    double angle = (double)h_a;
    const double vx = (double)h_vx, vy = (double)h_vy;
    if (vx * vx + vy * vy > 0.5 * 0.5) angle = crl_atan2(-vx, vy);
    float sn, cs, hs, hc;
    crl_sincosf((float)angle, &sn, &cs), crl_sincosf(h_a, &hs, &hc);
    const V2 hp = mk(h_cx, h_cy) - rotv(hs, hc, mk(K.hull_lc[0], K.hull_lc[1]));
    const V2 off = hp + mk(cs * 0.0f - sn * 16.0f, sn * 0.0f + cs * 16.0f);
    // ---- camera_view(mode="rgb_array"): crop rectangle, then surf_rotate's constants (pygame 1.9.6 transform.c)
    const int W = 96, H = 96, SW = 192, SH = 192;
    const double pos0 = CRL_CAR_OBS_SCALE * -(double)off.x + kMapSurface / 2.0, pos1 = CRL_CAR_OBS_SCALE * -(double)off.y + kMapSurface / 2.0;
    const double rxd = pos0 - W, ryd = pos1 - H;
    // crop rectangles that cannot meet the window (cars that left the playfield long ago) show grass; the reference
    // raises once a rectangle leaves its 10000^2 surface
    const bool far = !(rxd > kMapOrg - 256.0 && rxd < kMapOrg + kMapW + 64.0 && ryd > kMapOrg - 256.0 && ryd < kMapOrg + kMapW + 64.0);
    const int rx = far ? 0 : (int)rxd - kMapOrg, ry = far ? 0 : (int)ryd - kMapOrg;
    const float deg = (float)(57.295779513 * angle);  // PyArg_ParseTuple "f"
    int dx00, dy00, isin, icos;
    if (fmod((double)deg, 90.0) == 0.0) {  // rotate90(surf, (int)angle): exact quarter turns, as the same affine map
        int turns = ((int)deg / 90) % 4;
        if (turns < 0) turns += 4;
        // blit offset: the (rotated) 192 x 192 surface is centred, screen (X, Y) shows its pixel (X + 48, Y + 48)
        isin = turns == 1 ? 65536 : turns == 3 ? -65536 : 0;
        icos = turns == 0 ? 65536 : turns == 2 ? -65536 : 0;
        dx00 = ((turns == 0 || turns == 3) ? 48 : 143) << 16;
        dy00 = ((turns == 0 || turns == 1) ? 48 : 143) << 16;
    } else {
        const double radangle = deg * .01745329251994329;
        double sangle, cangle;
        crl_sincos(radangle, &sangle, &cangle);
        const double x = SW, y = SH, cxd = cangle * x, cyd = cangle * y, sxd = sangle * x, syd = sangle * y;
        const int nxmax = (int)fmax(fmax(fmax(fabs(cxd + syd), fabs(cxd - syd)), fabs(-cxd + syd)), fabs(-cxd - syd));
        const int nymax = (int)fmax(fmax(fmax(fabs(sxd + cyd), fabs(sxd - cyd)), fabs(-sxd + cyd)), fabs(-sxd - cyd));
        const int dcy = nymax / 2;
        const int xd = (SW - nxmax) * 32768, yd = (SH - nymax) * 32768;
        isin = (int)(sangle * 65536), icos = (int)(cangle * 65536);
        const int ax = (nxmax << 15) - (int)(cangle * ((nxmax - 1) << 15));
        const int ay = (nymax << 15) - (int)(sangle * ((nxmax - 1) << 15));
        const int bx = -(nxmax >> 1) + W / 2, by = -(nymax >> 1) + H / 2;  // where the rotated surface is blitted
        // rotated-surface pixel (X - bx, Y - by): dx = ax + isin * (dcy - (Y - by)) + xd + icos * (X - bx)
        dx00 = ax + isin * (dcy + by) + xd - icos * bx;
        dy00 = ay - icos * (dcy + by) + yd - isin * bx;
    }
    // extremes of the affine maps over the screen are at its corners
    int flags = 3;
    for (int c = 0; c < 4; c++) {
        const int X = (c & 1) ? 95 : 0, Y = (c & 2) ? 95 : 0;
        const int dx = dx00 + icos * X - isin * Y, dy = dy00 + isin * X + icos * Y;
        if (dx < 0 || dy < 0 || dx > (SW << 16) - 1 || dy > (SH << 16) - 1) flags &= ~2;
        const int mx = rx + (dx >> 16), my = ry + (dy >> 16);
        if (mx < 0 || my < 0 || mx >= kMapW || my >= kMapW) flags &= ~1;
    }
    if (far) flags = 4;
    vp.dx00 = dx00 + rx * 65536, vp.dy00 = dy00 + ry * 65536, vp.isin = isin, vp.icos = icos, vp.rx = rx, vp.ry = ry, vp.flags = flags;
    vp.text_idx = -1;
    if (s.text_bits) {
        const double r = s.reward[me];
        const double rr = rint(r);  // "%.0f" rounds half to even
        int idx = (int)rr - CRL_CAR_TEXT_RMIN;
        if (rr == 0.0 && (r < 0.0 || (r == 0.0 && signbit(r)))) idx = CRL_CAR_TEXT_STRINGS - 1;  // "-0000"
        vp.text_idx = min(max(idx, 0), CRL_CAR_TEXT_STRINGS - 1);
    }
    cam = make_float4(sn, cs, off.x, off.y);  // the float32 camera for the car polygons (Car.draw_for_pygame's tmp transform and offset)
}
__device__ void car_camera_tile(const CarSoA &s, const CarConsts &K, int64_t env, int viewer) {
    ViewParams vp;
    float4 cam;
    camera_compute(s, K, env, viewer, vp, cam);
    int32_t *dst = s.view + (env * s.players + viewer) * kViewWords;
    const int32_t *src = reinterpret_cast<const int32_t *>(&vp);
    for (int i = 0; i < 8; i++) dst[i] = src[i];
    reinterpret_cast<float4 *>(dst)[4] = cam;
}

// 16 lanes per tile: lane q = car polygon q of the draw order -> its scanline spans; lanes 0-7 also one indicator rectangle each.
// rect_out: the eight indicator rectangles (lane q < 8 writes [q]); rec: the polygon's span slots; returns the span count
__device__ __forceinline__ int poly_compute(const CarSoA &s, const CarConsts &K, int64_t env, int viewer, int q, const float4 cam, uint32_t *rect_out,
                                            uint32_t *rec) {
    const int64_t n = s.n, M = (int64_t)s.players * n;
    const int64_t me = viewer * n + env;
    const float sn = cam.x, cs = cam.y;
    const V2 off = mk(cam.z, cam.w);
    const float scale_f = (float)CRL_CAR_OBS_SCALE;
    if (q < 8) {  // render_indicators_for_pygame(width = height = 96): s = h = 2.4
        const double S = 96 / 40.0, Hh = 96 / 40.0;
        const float h_a = s.body[2 * M + me];
        uint32_t r;
        if (q == 0) r = pack_rect(0, 96 - 4 * Hh, 96, 4 * Hh * 1000);
        else if (q == 1) {
            const double vx = (double)s.body[3 * M + me], vy = (double)s.body[4 * M + me];
            r = pack_rect(5 * S, 96 - Hh, S, Hh * (-0.02 * sqrt(vx * vx + vy * vy)));
        } else if (q < 6) r = pack_rect((7 + (q - 2)) * S, 96 - Hh, S, Hh * (-0.01 * s.womega[(q - 2) * M + me]));
        else if (q == 6) r = pack_rect(20 * S, 96 - 2 * Hh, S * (10.0 * (double)(s.body[(6 + 2) * M + me] - h_a - 0.0f)), 2 * Hh);
        else r = pack_rect(30 * S, 96 - 2 * Hh, S * (0.8 * (double)s.body[5 * M + me]), 2 * Hh);
        rect_out[q] = r;
    }
    // ---- Car.draw_for_pygame: lane q = polygon q of the draw order (car 0: wheels 0-3, hull fixtures 0-3; then car 1)
    int cnt = 0;
    const int k = q >> 3, part = q & 7;
    if (k < s.players) {
        const int64_t ci = k * n + env;
        const int o = part < 4 ? 6 + 6 * part : 0;
        const float bx = s.body[(o + 0) * M + ci], by = s.body[(o + 1) * M + ci], ba = s.body[(o + 2) * M + ci];
        float bs, bc;
        crl_sincosf(ba, &bs, &bc);
        const V2 lc = part < 4 ? mk(0.f, 0.f) : mk(K.hull_lc[0], K.hull_lc[1]);
        const V2 bp = mk(bx, by) - rotv(bs, bc, lc);
        const int nv = part < 4 ? 4 : K.hull_n[part - 4];
        int px[8], py[8];
        int x0 = 1 << 30, y0 = 1 << 30, x1 = -(1 << 30), y1 = -(1 << 30);
#pragma unroll
        for (int i = 0; i < 8; i++) {
            if (i < nv) {
                const V2 v = part < 4 ? mk(K.wheel_poly[i & 3][0], K.wheel_poly[i & 3][1]) : mk(K.hull_poly[part - 4][i][0], K.hull_poly[part - 4][i][1]);
                const V2 wv = rotv(bs, bc, v) + bp;
                const V2 t = rotv(-sn, cs, wv - off);  // tmp.angle = -angle
                const float X = (-scale_f) * t.x + 48.0f, Y = (-scale_f) * t.y + 48.0f;
                // (a car far outside the view: any value that keeps the polygon off the screen)
                px[i] = (int)fminf(fmaxf(X, -30000.0f), 30000.0f), py[i] = (int)fminf(fmaxf(Y, -30000.0f), 30000.0f);
                x0 = min(x0, px[i]), x1 = max(x1, px[i]), y0 = min(y0, py[i]), y1 = max(y1, py[i]);
            } else {
                px[i] = py[i] = 0;
            }
        }
        if (x1 >= 0 && x0 <= 95 && y1 >= 0 && y0 <= 95) {
            int lx = px[0], ly = py[0];  // last vertex = predecessor of vertex 0
#pragma unroll
            for (int i = 1; i < 8; i++)
                if (i == nv - 1) lx = px[i], ly = py[i];
            for (int y = max(y0, 0); y <= min(y1, 95); y++) {
                int xs[8];
#pragma unroll
                for (int i = 0; i < 8; i++) xs[i] = 0x7FFFFFFF;
                if (y0 == y1) {
                    xs[0] = x0, xs[1] = x1;
                } else {
#pragma unroll
                    for (int i = 0; i < 8; i++) {
                        if (i < nv) {
                            const int xp = i ? px[i - 1] : lx, yp = i ? py[i - 1] : ly;
                            int ya = yp, yb = py[i], xa = xp, xb = px[i];
                            if (ya > yb) yb = yp, ya = py[i], xb = xp, xa = px[i];
                            if (ya != yb && ((y >= ya && y < yb) || (y == y1 && y > ya && y <= yb))) xs[i] = (y - ya) * (xb - xa) / (yb - ya) + xa;
                        }
                    }
#define CRL_CE(a, b)                                                  \
    {                                                                 \
        const int lo_ = min(xs[a], xs[b]), hi_ = max(xs[a], xs[b]);   \
        xs[a] = lo_, xs[b] = hi_;                                     \
    }
                    CRL_CE(0, 1) CRL_CE(2, 3) CRL_CE(4, 5) CRL_CE(6, 7) CRL_CE(0, 2) CRL_CE(1, 3) CRL_CE(4, 6) CRL_CE(5, 7) CRL_CE(1, 2) CRL_CE(5, 6)
                    CRL_CE(0, 4) CRL_CE(3, 7) CRL_CE(1, 5) CRL_CE(2, 6) CRL_CE(1, 4) CRL_CE(3, 6) CRL_CE(2, 4) CRL_CE(3, 5) CRL_CE(3, 4)
#undef CRL_CE
                }
#pragma unroll
                for (int i = 0; i < 8; i += 2) {
                    if (xs[i + 1] == 0x7FFFFFFF) continue;
                    const int xl = max(xs[i], 0), xr = min(xs[i + 1], 95);
                    if (xl > xr) continue;
                    if (cnt < kSpanSlots) rec[cnt] = (uint32_t)y | ((uint32_t)xl << 8) | ((uint32_t)xr << 16);
                    cnt++;
                }
            }
        }
    }
    // (a polygon is at most 5.3 px across: <= 8 spans; a count above the slots would be a bug and shows up as a missing span)
    return min(cnt, kSpanSlots);
}
__device__ void car_poly_tile(const CarSoA &s, const CarConsts &K, int64_t env, int viewer, int q) {
    const int64_t tile = env * s.players + viewer;
    const float4 cam = reinterpret_cast<const float4 *>(s.view + tile * kViewWords)[4];
    s.view_cnt[tile * 16 + q] = (uint8_t)poly_compute(s, K, env, viewer, q, cam, reinterpret_cast<uint32_t *>(s.view + tile * kViewWords) + 8,
                                                      s.view_rec + (tile * 16 + q) * kSpanSlots);
}

// `filter` (optional): the env is handled only if filter[env] == want
__global__ __launch_bounds__(64) void car_camera_kernel(CarSoA s, CarConsts K, const uint8_t *__restrict__ filter, int want) {
    const int64_t t = (int64_t)blockIdx.x * 64 + threadIdx.x;
    if (t >= s.n * s.players) return;
    const int64_t env = t / s.players;
    if (filter && filter[env] != want) return;
    car_camera_tile(s, K, env, (int)(t % s.players));
}
__global__ __launch_bounds__(64) void car_poly_kernel(CarSoA s, CarConsts K, const uint8_t *__restrict__ filter, int want) {
    const int64_t t = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 4);
    if (t >= s.n * s.players) return;
    const int64_t env = t / s.players;
    if (filter && filter[env] != want) return;
    car_poly_tile(s, K, env, (int)(t % s.players), threadIdx.x & 15);
}
// ------------------------------------------------------------------------------------------------ the tile
static constexpr int kPitch = 28;  // dwords per tile row in LDS (96 B of pixels + 16 B: the 4-row patch stores spread over the banks)

// CHECK = false: every source pixel is inside the window and inside the crop (ViewParams.flags == 3)
// it0, it_step: which of the nine 32 x 32-pixel regions this wavefront draws (0, 1: all of them; w, W: every W-th from w on)
template <bool CHECK>
__device__ __forceinline__ void obs_background(const uint8_t *__restrict__ map, const int dx00, const int dy00, const int isin, const int icos,
                                               const int rx, const int ry, uint32_t *__restrict__ tile, const int lane, const int it0 = 0,
                                               const int it_step = 1) {
    uint32_t bgpal = 0;
    if (CHECK) {  // rotate()'s background colour = the crop's first pixel
        if (rx >= 0 && ry >= 0 && rx < kMapW && ry < kMapW) {
            const uint32_t b = map[((ry >> 4) * kMapBlocks + (rx >> 4)) * 128 + (ry & 15) * 8 + ((rx & 15) >> 1)];
            bgpal = (b >> ((rx & 1) * 4)) & 15u;
        }
    }
#pragma unroll 1
    for (int it = it0; it < 9; it += it_step) {
        // a wavefront iteration covers a 32 x 32-pixel region: lane = a 4 x 4 patch, so one load instruction reads 64 pixels
        // of ONE region (a handful of 128-byte blocks) and a lane's 16 loads stay within one or two blocks
        const int X0 = 32 * (it % 3) + 4 * (lane & 7), Y0 = 32 * (it / 3) + 4 * (lane >> 3);
        int dxr = dx00 + icos * X0 - isin * Y0, dyr = dy00 + isin * X0 + icos * Y0;
        uint32_t nib[16];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            int dx = dxr, dy = dyr;
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int sx = dx >> 16, sy = dy >> 16;
                uint32_t p;
                if (!CHECK) {
                    // straight from the 16.16 coordinates (all non-negative here): block row dy >> 20, block column dx >> 20, row in
                    // block bits 16-19 of dy, byte in row bits 17-19 of dx, nibble bit 16 of dx; a 32-bit unsigned offset from
                    // the env's map base (uniform), so the load takes base + offset without 64-bit address arithmetic per pixel
                    const uint32_t udx = (uint32_t)dx, udy = (uint32_t)dy;
                    const uint32_t blk = (udy >> 20) * (uint32_t)kMapBlocks + (udx >> 20);
                    const uint32_t off = (blk << 7) | ((udy >> 13) & 0x78u) | __builtin_amdgcn_ubfe(udx, 17, 3);
                    const uint32_t b = map[off];
                    p = __builtin_amdgcn_ubfe(b, (udx >> 14) & 4u, 4);
                } else {
                    const int ux = dx - rx * 65536, uy = dy - ry * 65536;  // position inside the 192 x 192 crop
                    const bool in_crop = !(ux < 0 || uy < 0 || ux > (192 << 16) - 1 || uy > (192 << 16) - 1);
                    const bool in_win = sx >= 0 && sy >= 0 && sx < kMapW && sy < kMapW;
                    uint32_t b = 0;
                    if (in_crop && in_win) b = map[((sy >> 4) * kMapBlocks + (sx >> 4)) * 128 + (sy & 15) * 8 + ((sx & 15) >> 1)];
                    p = !in_crop ? bgpal : (in_win ? ((b >> ((sx & 1) * 4)) & 15u) : (uint32_t)kPalGrass);
                }
                nib[4 * j + i] = p;
                dx += icos, dy += isin;
            }
            dxr -= isin, dyr += icos;
        }
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const uint32_t sel = nib[4 * j] | (nib[4 * j + 1] << 8) | (nib[4 * j + 2] << 16) | (nib[4 * j + 3] << 24);
            tile[(Y0 + j) * kPitch + (X0 >> 2)] = __builtin_amdgcn_perm(kLutHi, kLutLo, sel);
        }
    }
}

// vp: ViewParams words; rec / cnt: the car polygons' spans (global memory from the camera / polygon kernels, or LDS in the fused kernel)
// WAVES wavefronts per tile: 1 (the big launch: throughput), or 4 (the list launches at the end of a step's chains: latency -- the
// nine background regions are shared out, the overlays stay with wavefront 0 -- LDS operations of ONE wavefront execute in
// program order, which is what makes a later layer overwrite an earlier one --, the stream-out is shared again)
template <int WAVES = 1>
__device__ __forceinline__ void car_obs_tile(const CarSoA &s, uint8_t *__restrict__ obs, const int64_t env, const int viewer, uint32_t *tile,
                                             const int32_t *vp, const uint32_t *rec, const uint8_t *cnt) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t t = env * s.players + viewer;
    const int dx00 = vp[0], dy00 = vp[1], isin = vp[2], icos = vp[3], rx = vp[4], ry = vp[5], flags = vp[6], text_idx = vp[7];
    const uint8_t *map = env_map(s, env);
    uint8_t *tile8 = reinterpret_cast<uint8_t *>(tile);
    // ---- background
    if (flags == 3) {
        obs_background<false>(map, dx00, dy00, isin, icos, rx, ry, tile, lane, wave, WAVES);
    } else if (flags & 4) {
        for (int i = threadIdx.x; i < 96 * kPitch; i += 64 * WAVES) tile[i] = G_GRASS * 0x01010101u;
    } else {
        obs_background<true>(map, dx00, dy00, isin, icos, rx, ry, tile, lane, wave, WAVES);
    }
    if (WAVES > 1) __syncthreads();
    if (wave == 0) {
    // ---- cars.  Draw order: car 0 wheels (black), car 0 hull, car 1 wheels, car 1 hull; within a layer every span has the
    // same colour, and LDS operations of ONE wavefront execute in program order, so a later layer simply overwrites
    const uint32_t *cnt32 = reinterpret_cast<const uint32_t *>(cnt);  // one byte per polygon, one word per layer
    const uint32_t cnt_all[4] = {cnt32[0], cnt32[1], cnt32[2], cnt32[3]};
#pragma unroll
    for (int layer = 0; layer < 4; layer++) {
        if (layer >= 2 * s.players) break;
        if (cnt_all[layer] == 0u) continue;  // uniform: no span in the four polygons of this layer
        const int poly = lane >> 4, slot = lane & 15;  // kSpanSlots == 16
        const int c = (int)((cnt_all[layer] >> (8 * poly)) & 0xFFu);
        const uint32_t r = rec[(layer * 4 + poly) * kSpanSlots + slot];
        const int gray = (layer & 1) ? ((layer >> 1) == viewer ? G_OWN : G_OTHER) : 0;
        if (slot < c) {
            const int y = (int)(r & 0xFFu), xl = (int)((r >> 8) & 0xFFu), xr = (int)((r >> 16) & 0xFFu);
            for (int x = xl; x <= xr; x++) tile8[y * (kPitch * 4) + x] = (uint8_t)gray;
        }
    }
    // ---- indicator bars, in order (later rectangles win); the black bar under them first
    const uint32_t *rects = reinterpret_cast<const uint32_t *>(vp) + 8;
    const int rgray[8] = {0, G_BLUE, G_BLUE, G_BLUE, G_ABS_REAR, G_ABS_REAR, G_GREEN, G_RED};
#pragma unroll
    for (int r = 0; r < 8; r++) {
        const uint32_t q = rects[r];
        const int x0 = (int)(q & 0xFFu), x1 = (int)((q >> 8) & 0xFFu), y0 = (int)((q >> 16) & 0xFFu), y1 = (int)(q >> 24);
        const int w = x1 - x0 + 1, h = y1 - y0 + 1;
        if (w <= 0) continue;
        for (int p = lane; p < w * h; p += 64) {
            const int yy = p / w, xx = p - yy * w;
            tile8[(y0 + yy) * (kPitch * 4) + x0 + xx] = (uint8_t)rgray[r];
        }
    }
    // ---- reward read-out "%05.0f" (white 1-bit glyphs) blitted at (0, 91): rows 91..95 of the 10
    if (text_idx >= 0 && lane < 32) {
        const uint32_t *rows = s.text_bits + (int64_t)text_idx * CRL_CAR_TEXT_ROWS;
#pragma unroll
        for (int row = 0; row < 5; row++)
            if ((rows[row] >> lane) & 1u) tile8[(91 + row) * (kPitch * 4) + lane] = 255;
    }
    }  // wave 0: overlays
    if (WAVES > 1) __syncthreads();
    // ---- stream the tile out: 16 B per lane, 1 KiB contiguous per wave store
    uint4 *__restrict__ out = reinterpret_cast<uint4 *>(obs + t * (96 * 96));
#pragma unroll
    for (int i = 0; i < (9 + WAVES - 1) / WAVES; i++) {
        const int c = (i * WAVES + wave) * 64 + lane, row = c / 6, col = c - row * 6;
        if (WAVES == 1 || c < 576) out[c] = *reinterpret_cast<const uint4 *>(&tile[row * kPitch + col * 4]);
    }
}

// Tile slot b -> (position i = 8 (b / 16) + b % 8, viewer (b % 16) / 8) when there are two views: workgroups b and b + 8 run on
// the same XCD, so the two views of an env (which look at neighbouring parts of the same map) share that XCD's L2.
__global__ __launch_bounds__(64) void car_obs_kernel(CarSoA s, uint8_t *__restrict__ obs, const uint8_t *__restrict__ only_env, int want) {
    __shared__ __attribute__((aligned(16))) uint32_t tile[96 * kPitch];
    int64_t env = blockIdx.x;
    int viewer = 0;
    if (s.players == 2) {
        const int r = (int)(blockIdx.x & 15);
        env = (int64_t)(blockIdx.x >> 4) * 8 + (r & 7), viewer = r >> 3;
    }
    if (env >= s.n) return;
    if (only_env && only_env[env] != want) return;
    const int64_t t = env * s.players + viewer;
    car_obs_tile(s, obs, env, viewer, tile, s.view + t * kViewWords, s.view_rec + t * kViewRecWords, s.view_cnt + t * 16);
}

// The envs of a compacted list (the small env classes of a step): camera, polygons and tile in ONE launch, one wavefront per
// tile -- every lane computes the (uniform) camera, lanes 0-15 the polygons, everything handed over through LDS.  Three
// dependent launches of a few hundred wavefronts each cost three launch latencies at the end of a step.
__global__ __launch_bounds__(64) void car_obs_list_kernel(CarSoA s, CarConsts K, uint8_t *__restrict__ obs, const int32_t *__restrict__ list,
                                                          const int32_t *__restrict__ list_count, int32_t *__restrict__ count_to_host,
                                                          const uint8_t *__restrict__ filter, int want) {
    __shared__ __attribute__((aligned(16))) uint32_t tile[96 * kPitch];
    __shared__ __attribute__((aligned(16))) int32_t vp_s[16];
    __shared__ __attribute__((aligned(16))) uint32_t rec_s[kViewRecWords];
    __shared__ __attribute__((aligned(16))) uint8_t cnt_s[16];
    const int lane = threadIdx.x;
    const int64_t positions = *list_count;
    if (count_to_host && blockIdx.x == 0 && lane == 0) *count_to_host = (int32_t)positions;
    const int64_t tiles = positions * s.players;
    for (int64_t b = blockIdx.x; b < tiles; b += gridDim.x) {
        const int64_t env = list[b / s.players];
        const int viewer = (int)(b % s.players);
        if (!filter || filter[env] == want) {
            ViewParams vp;
            float4 cam;
            camera_compute(s, K, env, viewer, vp, cam);
            if (lane == 0) {
                const int32_t *src = reinterpret_cast<const int32_t *>(&vp);
                for (int i = 0; i < 8; i++) vp_s[i] = src[i];
            }
            if (lane < 16) cnt_s[lane] = (uint8_t)poly_compute(s, K, env, viewer, lane, cam, reinterpret_cast<uint32_t *>(vp_s) + 8, rec_s + lane * kSpanSlots);
            __syncthreads();
            car_obs_tile(s, obs, env, viewer, tile, vp_s, rec_s, cnt_s);  // (<4>, 256 threads: 135 us instead of 70 for a thousand tiles -- four times the camera work, and four wavefronts to place per tile)
        }
        __syncthreads();  // the next tile reuses the LDS
    }
}

// camera + car polygons of every env (or of the envs with only_env[e] == want): what launch_car_obs reads
void launch_car_view(const CarSoA &s, const CarConsts &k, hipStream_t st, const uint8_t *only_env, int want) {
    const int64_t tiles = s.n * s.players;
    hipLaunchKernelGGL(car_camera_kernel, dim3((unsigned)((tiles + 63) / 64)), dim3(64), 0, st, s, k, only_env, want);
    hipLaunchKernelGGL(car_poly_kernel, dim3((unsigned)((tiles + 3) / 4)), dim3(64), 0, st, s, k, only_env, want);
}
void launch_car_obs(const CarSoA &s, const CarConsts &k, uint8_t *obs, hipStream_t st, const uint8_t *only_env, int want) {
    const unsigned grid = s.players == 2 ? (unsigned)((s.n + 7) / 8 * 16) : (unsigned)s.n;
    hipLaunchKernelGGL(car_obs_kernel, dim3(grid), dim3(64), 0, st, s, obs, only_env, want);
}

// the envs of a compacted list (its length in device memory; `expected` = the caller's guess of it, only for the grid size),
// optionally only those with filter[env] == want
void launch_car_obs_list(const CarSoA &s, const CarConsts &k, uint8_t *obs, hipStream_t st, const int32_t *list, const int32_t *list_count,
                         int32_t *count_to_host, int64_t expected, const uint8_t *filter, int want_cls) {
    int64_t want = expected + expected / 4 + 32;  // slack: a launch that falls short loops, it does not miss tiles
    want = want > s.n ? s.n : want;
    hipLaunchKernelGGL(car_obs_list_kernel, dim3((unsigned)(want * s.players)), dim3(64), 0, st, s, k, obs, list, list_count, count_to_host, filter, want_cls);
}

// MultipleFrameStack + FlattenMultiAgentObservation + WrapPyTorch (reference
// utils/atari_wrappers.py:262-334, 12-37): per agent the last K frames, oldest first, agents
// concatenated on the channel axis -> (N, 2K, 96, 96).  reset() fills all K slots with the first
// frame (:284-290), which is also what an auto-reset does.  Pure HBM copy: reads K-1 planes,
// writes K planes into the context's own stack and the caller's obs tensor.
__global__ __launch_bounds__(256) void car_stack_kernel(const uint4 *__restrict__ frame, uint4 *__restrict__ stack,
                                                        uint4 *__restrict__ obs, const uint8_t *__restrict__ fill_env,
                                                        int fill_all, int K, int64_t n, int players) {
    const int64_t tile = blockIdx.x;  // (env, agent)
    const int64_t env = tile / players;
    const bool fill = fill_all || fill_env[env];
    const int chunks = 96 * 96 / 16;
    const uint4 *f = frame + tile * chunks;
    uint4 *st = stack + tile * K * chunks;
    uint4 *ob = obs + tile * K * chunks;
    for (int c = threadIdx.x; c < chunks; c += 256) {
        const uint4 newest = f[c];
        for (int k = 0; k < K - 1; k++) {
            const uint4 v = fill ? newest : st[(int64_t)(k + 1) * chunks + c];
            st[(int64_t)k * chunks + c] = v, ob[(int64_t)k * chunks + c] = v;
        }
        st[(int64_t)(K - 1) * chunks + c] = newest, ob[(int64_t)(K - 1) * chunks + c] = newest;
    }
}

void launch_car_stack(const uint8_t *frame, uint8_t *stack, uint8_t *obs, const uint8_t *fill_env, bool fill_all, int K, int64_t n,
                      int players, hipStream_t st) {
    hipLaunchKernelGGL(car_stack_kernel, dim3((unsigned)(players * n)), dim3(256), 0, st, reinterpret_cast<const uint4 *>(frame),
                       reinterpret_cast<uint4 *>(stack), reinterpret_cast<uint4 *>(obs), fill_env, fill_all ? 1 : 0, K, n, players);
}

}  // namespace crl
