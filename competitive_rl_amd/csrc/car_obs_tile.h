// car_obs_tile.h -- the per-frame device code of the CarRacing observation (camera, car polygons, 96 x 96 tile) of car_obs.hip's
// frame kernels, in a header so that another kernel can draw a frame itself (the library is built without relocatable device code).
// What is restated, and from where: see the head of car_obs.hip.
#pragma once
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
    vp.text_idx = -1;  // (the reward read-out is looked up by the tile itself: the camera does not depend on the step's wheel sensors)
    cam = make_float4(sn, cs, off.x, off.y);  // the float32 camera for the car polygons (Car.draw_for_pygame's tmp transform and offset)
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
// ------------------------------------------------------------------------------------------------ the tile
static constexpr int kPitch = 28;  // dwords per tile row in LDS (96 B of pixels + 16 B: the 4-row patch stores spread over the banks)

// CHECK = false: every source pixel is inside the window and inside the crop (ViewParams.flags == 3)
// it0, it_step: which of the nine 32 x 32-pixel regions this wavefront draws (0, 1: all of them; w, W: every W-th from w on)
// it_end / row0 (round 5): the iterations stop in front of it_end, and tile row r of the LDS buffer holds screen row row0 + r -- a
// wavefront that owns one region row (iterations 3 t .. 3 t + 2, screen rows 32 t .. 32 t + 31) needs a 32-row buffer only.
template <bool CHECK>
__device__ __forceinline__ void obs_background(const uint8_t *__restrict__ map, const int dx00, const int dy00, const int isin, const int icos,
                                               const int rx, const int ry, uint32_t *__restrict__ tile_, const int lane, const int it0 = 0,
                                               const int it_step = 1, const int it_end = 9, const int row0 = 0) {
    uint32_t *__restrict__ tile = tile_ - row0 * kPitch;
    uint32_t bgpal = 0;
    if (CHECK) {  // rotate()'s background colour = the crop's first pixel
        if (rx >= 0 && ry >= 0 && rx < kMapW && ry < kMapW) {
            const uint32_t b = map[((ry >> 4) * kMapBlocks + (rx >> 4)) * 128 + (ry & 15) * 8 + ((rx & 15) >> 1)];
            bgpal = (b >> ((rx & 1) * 4)) & 15u;
        }
    }
    if (!CHECK) {
        // Every source pixel is inside the window and the crop: the byte's address comes straight from the 16.16 coordinates (all
        // non-negative): block row dy >> 20, block column dx >> 20, row in block bits 16-19 of dy, byte in row bits 17-19 of dx,
        // nibble bit 16 of dx; a 32-bit unsigned offset from the env's map base (uniform), so the load takes base + offset without
        // 64-bit address arithmetic per pixel.  Two regions in flight: the next region's 16 loads are issued before this region's
        // bytes are unpacked -- a tile is a chain of dependent load rounds (30 cycles per vector instruction), not issue slots.
        auto fetch = [&](int it, uint32_t (&raw)[16]) {
            const int X0 = 32 * (it % 3) + 4 * (lane & 7), Y0 = 32 * (it / 3) + 4 * (lane >> 3);
            int dxr = dx00 + icos * X0 - isin * Y0, dyr = dy00 + isin * X0 + icos * Y0;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                int dx = dxr, dy = dyr;
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    const uint32_t udx = (uint32_t)dx, udy = (uint32_t)dy;
                    const uint32_t blk = (udy >> 20) * (uint32_t)kMapBlocks + (udx >> 20);
                    const uint32_t off = (blk << 7) | ((udy >> 13) & 0x78u) | __builtin_amdgcn_ubfe(udx, 17, 3);
                    raw[4 * j + i] = map[off];
                    dx += icos, dy += isin;
                }
                dxr -= isin, dyr += icos;
            }
        };
        auto unpack = [&](int it, const uint32_t (&raw)[16]) {
            const int X0 = 32 * (it % 3) + 4 * (lane & 7), Y0 = 32 * (it / 3) + 4 * (lane >> 3);
            int dxr = dx00 + icos * X0 - isin * Y0;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                int dx = dxr;
                uint32_t sel = 0;
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    sel |= __builtin_amdgcn_ubfe(raw[4 * j + i], ((uint32_t)dx >> 14) & 4u, 4) << (8 * i);
                    dx += icos;
                }
                tile[(Y0 + j) * kPitch + (X0 >> 2)] = __builtin_amdgcn_perm(kLutHi, kLutLo, sel);
                dxr -= isin;
            }
        };
        uint32_t cur[16], nxt[16];
        if (it0 < it_end) fetch(it0, cur);
#pragma unroll 1
        for (int it = it0; it < it_end; it += it_step) {
            const bool more = it + it_step < it_end;
            if (more) fetch(it + it_step, nxt);
            unpack(it, cur);
            if (more) {
#pragma unroll
                for (int k = 0; k < 16; k++) cur[k] = nxt[k];
            }
        }
        return;
    }
#pragma unroll 1
    for (int it = it0; it < it_end; it += it_step) {
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
                const int ux = dx - rx * 65536, uy = dy - ry * 65536;  // position inside the 192 x 192 crop
                const bool in_crop = !(ux < 0 || uy < 0 || ux > (192 << 16) - 1 || uy > (192 << 16) - 1);
                const bool in_win = sx >= 0 && sy >= 0 && sx < kMapW && sy < kMapW;
                uint32_t b = 0;
                if (in_crop && in_win) b = map[((sy >> 4) * kMapBlocks + (sx >> 4)) * 128 + (sy & 15) * 8 + ((sx & 15) >> 1)];
                nib[4 * j + i] = !in_crop ? bgpal : (in_win ? ((b >> ((sx & 1) * 4)) & 15u) : (uint32_t)kPalGrass);
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
    const int dx00 = vp[0], dy00 = vp[1], isin = vp[2], icos = vp[3], rx = vp[4], ry = vp[5], flags = vp[6];
    int text_idx = -1;  // which of the pre-rendered "%05.0f" strings shows the viewer's reward (crmp:666-670)
    if (s.text_bits) {
        const double r = s.reward[(int64_t)viewer * s.n + env];
        const double rr = rint(r);  // "%.0f" rounds half to even
        int idx = (int)rr - CRL_CAR_TEXT_RMIN;
        if (rr == 0.0 && (r < 0.0 || (r == 0.0 && signbit(r)))) idx = CRL_CAR_TEXT_STRINGS - 1;  // "-0000"
        text_idx = min(max(idx, 0), CRL_CAR_TEXT_STRINGS - 1);
    }
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


// ---- one THIRD of a frame per wavefront (round 5, the big frame launch): screen rows 32 t .. 32 t + 31.  A whole tile needs 10.5 KB of
// LDS, which holds the frame kernel at four wavefronts per SIMD; its time is the latency of nine dependent gather rounds, not issue slots,
// so what it lacks is wavefronts in flight.  A third needs 3.5 KB (seven wavefronts per SIMD at 73 registers), gathers three regions,
// and paints the overlay spans / rectangles / read-out rows that fall into its rows -- in the same order, so every pixel ends up with the
// colour of the last layer drawn over it, as in the whole-tile form.
__device__ __forceinline__ void car_obs_third(const CarSoA &s, uint8_t *__restrict__ obs, const int64_t env, const int viewer, const int third, uint32_t *tile,
                                              const int32_t *vp, const uint32_t *rec, const uint8_t *cnt) {
    const int lane = threadIdx.x & 63;
    const int64_t t = env * s.players + viewer;
    const int dx00 = vp[0], dy00 = vp[1], isin = vp[2], icos = vp[3], rx = vp[4], ry = vp[5], flags = vp[6];
    const int r0 = 32 * third;
    const uint8_t *map = env_map(s, env);
    uint8_t *tile8 = reinterpret_cast<uint8_t *>(tile);
    if (flags == 3) {
        obs_background<false>(map, dx00, dy00, isin, icos, rx, ry, tile, lane, 3 * third, 1, 3 * third + 3, r0);
    } else if (flags & 4) {
        for (int i = lane; i < 32 * kPitch; i += 64) tile[i] = G_GRASS * 0x01010101u;
    } else {
        obs_background<true>(map, dx00, dy00, isin, icos, rx, ry, tile, lane, 3 * third, 1, 3 * third + 3, r0);
    }
    // ---- cars (draw order: car 0 wheels, car 0 hull, car 1 wheels, car 1 hull; LDS operations of one wavefront execute in program order)
    const uint32_t *cnt32 = reinterpret_cast<const uint32_t *>(cnt);
    const uint32_t cnt_all[4] = {cnt32[0], cnt32[1], cnt32[2], cnt32[3]};
#pragma unroll
    for (int layer = 0; layer < 4; layer++) {
        if (layer >= 2 * s.players) break;
        if (cnt_all[layer] == 0u) continue;
        const int poly = lane >> 4, slot = lane & 15;
        const int c = (int)((cnt_all[layer] >> (8 * poly)) & 0xFFu);
        const uint32_t r = rec[(layer * 4 + poly) * kSpanSlots + slot];
        const int gray = (layer & 1) ? ((layer >> 1) == viewer ? G_OWN : G_OTHER) : 0;
        const int y = (int)(r & 0xFFu) - r0;
        if (slot < c && (unsigned)y < 32u) {
            const int xl = (int)((r >> 8) & 0xFFu), xr = (int)((r >> 16) & 0xFFu);
            for (int x = xl; x <= xr; x++) tile8[y * (kPitch * 4) + x] = (uint8_t)gray;
        }
    }
    if (third == 2) {  // (the indicator strip and the read-out live in rows 86 .. 95)
        const uint32_t *rects = reinterpret_cast<const uint32_t *>(vp) + 8;
        const int rgray[8] = {0, G_BLUE, G_BLUE, G_BLUE, G_ABS_REAR, G_ABS_REAR, G_GREEN, G_RED};
#pragma unroll
        for (int r = 0; r < 8; r++) {
            const uint32_t q = rects[r];
            const int x0 = (int)(q & 0xFFu), x1 = (int)((q >> 8) & 0xFFu), y0 = max((int)((q >> 16) & 0xFFu), r0), y1 = (int)(q >> 24);
            const int w = x1 - x0 + 1, h = y1 - y0 + 1;
            if (w <= 0 || h <= 0) continue;
            for (int p = lane; p < w * h; p += 64) {
                const int yy = p / w, xx = p - yy * w;
                tile8[(y0 - r0 + yy) * (kPitch * 4) + x0 + xx] = (uint8_t)rgray[r];
            }
        }
        if (s.text_bits && lane < 32) {
            const double rw = s.reward[(int64_t)viewer * s.n + env];
            const double rr = rint(rw);  // "%.0f" rounds half to even
            int idx = (int)rr - CRL_CAR_TEXT_RMIN;
            if (rr == 0.0 && (rw < 0.0 || (rw == 0.0 && signbit(rw)))) idx = CRL_CAR_TEXT_STRINGS - 1;  // "-0000"
            const uint32_t *rows = s.text_bits + (int64_t)min(max(idx, 0), CRL_CAR_TEXT_STRINGS - 1) * CRL_CAR_TEXT_ROWS;
#pragma unroll
            for (int row = 0; row < 5; row++)
                if ((rows[row] >> lane) & 1u) tile8[(91 - r0 + row) * (kPitch * 4) + lane] = 255;
        }
    }
    // ---- stream the 32 rows out: 16 B per lane, 1 KiB contiguous per wave store
    uint4 *__restrict__ out = reinterpret_cast<uint4 *>(obs + t * (96 * 96) + r0 * 96);
#pragma unroll
    for (int i = 0; i < 3; i++) {
        const int c = i * 64 + lane, row = c / 6, col = c - row * 6;
        out[c] = *reinterpret_cast<const uint4 *>(&tile[row * kPitch + col * 4]);
    }
}

}  // namespace crl
