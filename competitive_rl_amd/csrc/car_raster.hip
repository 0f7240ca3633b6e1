// car_raster.hip -- cCarRacingDouble observations: (N, 2, 96, 96) uint8, one workgroup per
// (env, viewer) tile of 9 216 pixels.
//
// Restates CarRacing.get_observation (reference car_racing/car_racing_multi_players.py:622-634):
// camera_update("rgb_array") :791-804, the rotated crop of the pre-rendered map :764-789,
// Car.draw_for_pygame (car_dynamics.py:284-298), render_indicators_for_pygame :645-670 and the
// luma truncation.  The reference's 10000x10000 pygame map (400 MB per env) is replaced by an
// analytic classification of each pixel centre against the track polygons in world space;
// cars and indicator bars follow pygame's integer polygon fill rule.  Parity with real
// pygame is unpinned; the CPU checker under tests/ uses the same definition.
//
// Workgroup b draws (env, viewer) = the two views of env 8 (b / 16) + b % 8 in blocks 16 k + j and 16 k + 8 + j: same XCD
// (b % 8), dispatched together, so the env's track is fetched from HBM once.
// Per workgroup: (0) the camera (one wavefront, f64 atan2) goes through LDS, (1) the tiles whose AABB meets the camera's view
// box are compacted IN ORDER into LDS as 80-byte vertex records (draw order matters: lower tile index is drawn later and
// wins), (2) the 16 car
// polygons and 8 indicator rectangles are projected once into LDS, and every 8x8-pixel cell
// gets bit masks of the candidates that can cover it (separating-edge test in screen space),
// (3) the background is resolved 4 pixels per lane into an LDS tile (a wavefront per 16x16
// block), car polygons are filled as pygame scanline spans, indicator bars and the reward
// read-out are patched over, and the tile is streamed out 16 B per lane.
// Rule of the house: whatever a loop iteration needs from LDS is read with back-to-back loads
// into registers and evaluated branch-free -- a chain of data-dependent ~100-cycle LDS reads
// costs more than the arithmetic an early-out saves.
// Superseded in round 3 by car_obs.hip (the reference's own map + rotate algorithm); this analytic raster differs from the
// reference on ~6 % of the pixels and exists only in the profiling build (-DCRL_ABLATION, CRL_CAR_OBS_ANALYTIC=1) for A/B.
#ifdef CRL_ABLATION
#include <stdio.h>
#include <stdlib.h>

#include "car_device.h"

namespace crl {

#ifndef CRL_CAR_RASTER_WAVES
#define CRL_CAR_RASTER_WAVES 5  // wavefronts per SIMD the register allocation aims at (= workgroups per CU; LDS allows 5)
#endif
static constexpr int kMaxCand = 128;
static constexpr int kRankCap = 1024;  // pixels of one car's screen box handled by the patch pass (a car spans ~10 x 10)

#define G_GRASS 161
#define G_LIGHT 176
#define G_WHITE 255
#define G_RED 76
#define G_OWN 60
#define G_OTHER 29
#define G_BLUE 29
#define G_ABS_REAR 44
#define G_GREEN 149

// 96 bytes per candidate: VERTICES, not edges -- an edge vector is the difference of the same two floats wherever it is
// formed, so expanding the record in registers gives the values the edge form would have stored (LDS is what limits
// the workgroups per CU here: 12 KB of candidates instead of 23.5)
struct CandTile {
    float4 v01, v23, v4b0;  // tile polygon p0..p4 (counter-clockwise), then border quad b0
    float4 b12, b3m;        // b1, b2, b3; m.z = tile index | border kind << 16 (bits), m.w unused
    float4 bb_all;          // world AABB of tile + border
};

// edges as (ax, ay, bx-ax, by-ay)
__device__ inline void cand_expand(const CandTile &ct, float4 (&te)[5], float4 (&be)[4], float4 &ba, int &idx, int &border) {
    const float4 a = ct.v01, b = ct.v23, c = ct.v4b0, d = ct.b12, e = ct.b3m;
    ba = ct.bb_all;
    te[0] = make_float4(a.x, a.y, a.z - a.x, a.w - a.y);
    te[1] = make_float4(a.z, a.w, b.x - a.z, b.y - a.w);
    te[2] = make_float4(b.x, b.y, b.z - b.x, b.w - b.y);
    te[3] = make_float4(b.z, b.w, c.x - b.z, c.y - b.w);
    te[4] = make_float4(c.x, c.y, a.x - c.x, a.y - c.y);
    be[0] = make_float4(c.z, c.w, d.x - c.z, d.y - c.w);
    be[1] = make_float4(d.x, d.y, d.z - d.x, d.w - d.y);
    be[2] = make_float4(d.z, d.w, e.x - d.z, e.y - d.w);
    be[3] = make_float4(e.x, e.y, c.z - e.x, c.w - e.y);
    const int m = __float_as_int(e.z);
    idx = m & 0xFFFF, border = m >> 16;
}

// inside test against precomputed edges: (bx-ax)*(y-ay) - (by-ay)*(x-ax) >= 0 for every edge
__device__ inline bool in_edges(const float4 *e, int nv, float x, float y) {
    for (int i = 0; i < nv; i++) {
        const float4 q = e[i];
        if ((q.z * (y - q.y) - q.w * (x - q.x)) < 0) return false;
    }
    return true;
}

static constexpr int kCell = 8, kCellShift = 3, kCellsPerRow = 96 / kCell, kCells = kCellsPerRow * kCellsPerRow;

struct CarPoly {
    int px[8], py[8];
    int n, x0, x1, y0, y1, gray;  // 88 bytes
};

struct IndRect {
    int x0, x1, y0, y1, gray;
};

__device__ inline bool in_convex(const float *poly, int nv, float x, float y) {
    for (int i = 0; i < nv; i++) {
        const int j = i + 1 < nv ? i + 1 : 0;
        const float ax = poly[2 * i], ay = poly[2 * i + 1], bx = poly[2 * j], by = poly[2 * j + 1];
        if (((bx - ax) * (y - ay) - (by - ay) * (x - ax)) < 0) return false;
    }
    return true;
}

__device__ inline IndRect make_rect(double x, double y, double w, double h, int gray) {
    const int l = (int)x, t = (int)y, r = (int)x + (int)w - 1, b = (int)y + (int)h - 1;
    IndRect q;
    q.x0 = min(l, r), q.x1 = max(l, r), q.y0 = min(t, b), q.y1 = max(t, b), q.gray = gray;
    return q;
}

// CRL_CAR_DEBUG & 64: cycle counter (s_memtime) at every workgroup barrier, summed per phase over all workgroups
__device__ unsigned long long g_car_ticks[24];
#define CAR_TICK(Kk)                                                             \
    if (TICKS && (dbg & 64) && tid == 0) {                                       \
        const long long now_ = __builtin_readcyclecounter();                     \
        tick_acc[Kk] += (unsigned)(now_ - tick_prev);                            \
        tick_prev = now_;                                                        \
    }
// TICKS: the instance with the cycle stamps (their accumulators cost a dozen registers -- in the production instance they were spills)
template <bool TICKS>
__device__ __forceinline__ void car_raster_tile(const CarSoA &s, const CarConsts &K, uint8_t *__restrict__ obs, int dbg, const int64_t env,
                                                const int viewer) {
    __shared__ CandTile cand[kMaxCand];
    __shared__ __attribute__((aligned(8))) CarPoly cars[16];
    __shared__ __attribute__((aligned(16))) IndRect ind[8];
    __shared__ int wave_cnt[2][4];
    __shared__ int16_t cand_tile[kMaxCand];       // tile index of each candidate, ascending
    __shared__ uint16_t cand_cells[kMaxCand];     // culling: cell range cx0 | cx1 << 4 | cy0 << 8 | cy1 << 12
    __shared__ uint16_t cand_ofs[kMaxCand + 2];   // exclusive prefix of the cell counts (<= 128 * 144)
    __shared__ __attribute__((aligned(16))) uint32_t cell_tmask[kCells][kMaxCand / 32];  // candidates whose tile polygon may cover a pixel of the cell
    __shared__ __attribute__((aligned(16))) uint32_t cell_bmask[kCells][kMaxCand / 32];  // ... whose border quad may
    __shared__ int car_box[2][4];                  // per car: screen box of all its polygons
    __shared__ __attribute__((aligned(16))) int poly_row0[20];  // first scanline work item of each car polygon (+ total)
    __shared__ int ind_y0;
    __shared__ float cam[8];  // sin, cos, centre of the view; hull angle, velocity, spin (for the indicators)
    __shared__ __attribute__((aligned(16))) uint32_t tile32[96 * 96 / 4];
    const int64_t n = s.n, M = (int64_t)s.players * n;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int64_t me = viewer * n + env;
    long long tick_prev = (TICKS && (dbg & 64)) ? __builtin_readcyclecounter() : 0;
    unsigned tick_acc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};

    // Loads that do not depend on the camera go out first (tile AABBs: unconditional, the slots past the
    // track hold zeros; the reward for the read-out), so the double-precision camera math covers them.
    const int ntiles = (dbg & 8) ? 0 : s.ntiles[env];
    const float4 bb_pre0 = s.tile_aabb_em[env * kCarMaxTiles + tid], bb_pre1 = s.tile_aabb_em[env * kCarMaxTiles + 256 + tid];
    const double reward_pre = s.reward[me];
    // the glyph row of the reward read-out is a load that depends on that reward: requested here (the reward is needed
    // for nothing else), used by the last phase -- at its place of use it cost a whole exposed round trip per workgroup
    uint32_t text_row_pre = 0;
    if (s.text_bits && tid < 32 * CRL_CAR_TEXT_ROWS) {
        const double r = reward_pre;
        const double rr = rint(r);  // "%.0f" rounds half to even
        int idx = (int)rr - CRL_CAR_TEXT_RMIN;
        if (rr == 0.0 && (r < 0.0 || (r == 0.0 && signbit(r)))) idx = CRL_CAR_TEXT_STRINGS - 1;  // "-0000"
        idx = min(max(idx, 0), CRL_CAR_TEXT_STRINGS - 1);
        text_row_pre = s.text_bits[idx * CRL_CAR_TEXT_ROWS + (tid >> 5)];
    }

    // ---- camera_update("rgb_array") for the viewer: uniform across the workgroup, so ONE wavefront does the
    // double-precision part (atan2 of the velocity) and hands the result over through LDS; the other three only
    // wait for their tile boxes meanwhile
    if (wave == 0) {
        const float h_cx = s.body[0 * M + me], h_cy = s.body[1 * M + me], h_a = s.body[2 * M + me];
        const float h_vx = s.body[3 * M + me], h_vy = s.body[4 * M + me], h_w = s.body[5 * M + me];
        double angle = (double)h_a;
        const double vx = (double)h_vx, vy = (double)h_vy;
        if (vx * vx + vy * vy > 0.5 * 0.5) angle = atan2(-vx, vy);
        const float af = (float)angle;
        float sn_, cs_, hs, hc;
        crl_sincosf(af, &sn_, &cs_), crl_sincosf(h_a, &hs, &hc);
        const V2 hp = mk(h_cx, h_cy) - rotv(hs, hc, mk(K.hull_lc[0], K.hull_lc[1]));
        const V2 off_ = hp + mk(cs_ * 0.0f - sn_ * 16.0f, sn_ * 0.0f + cs_ * 16.0f);
        if (lane == 0) cam[0] = sn_, cam[1] = cs_, cam[2] = off_.x, cam[3] = off_.y, cam[4] = h_a, cam[5] = h_vx, cam[6] = h_vy, cam[7] = h_w;
    }
    __syncthreads();
    const float sn = cam[0], cs = cam[1];
    const V2 off = mk(cam[2], cam[3]);
    const float h_a = cam[4], h_w = cam[7];
    const double vx = (double)cam[5], vy = (double)cam[6];
    const double obs_scale = (10 / (100 / sqrt(96.0))) * 1.8;
    const float inv_scale = (float)(1.0 / obs_scale), scale_f = (float)obs_scale;
    const float kf = (float)(CAR_PLAYFIELD / 20.0), inv_kf = 1.0f / kf;

    // ---- (1) ordered compaction of the tiles near the view (half-diagonal 48*sqrt(2)/scale < 39):
    // both halves of the tile range are tested at once (kCarMaxTiles <= 512), then the kept tiles'
    // polygons are fetched by the first n_cand threads
    static_assert(kCarMaxTiles == 512, "compaction covers two tiles per thread");
    const float vr = 39.0f + 1.5f;  // + border width
    bool keep[2];
    unsigned long long km[2];
    for (int h = 0; h < 2; h++) {
        const int t = h * 256 + tid;
        keep[h] = false;
        if (t < ntiles) {
            const float4 bb = h ? bb_pre1 : bb_pre0;
            keep[h] = !(bb.x > off.x + vr || bb.z < off.x - vr || bb.y > off.y + vr || bb.w < off.y - vr);
        }
    }
    for (int h = 0; h < 2; h++) {
        km[h] = __ballot(keep[h]);
        if (lane == 0) wave_cnt[h][wave] = __popcll(km[h]);
    }
    __syncthreads();
    CAR_TICK(0)
    int nc;
    {
        int before[2] = {0, 0}, tot = 0;
        for (int h = 0; h < 2; h++)
            for (int w = 0; w < 4; w++) {
                if (w == wave) before[h] = tot;
                tot += wave_cnt[h][w];
            }
        nc = min(tot, kMaxCand);
        for (int h = 0; h < 2; h++) {
            const int slot = before[h] + __popcll(km[h] & ((1ull << lane) - 1ull));
            if (keep[h] && slot < kMaxCand) cand_tile[slot] = (int16_t)(h * 256 + tid);
        }
    }
    __syncthreads();
    CAR_TICK(1)
    if (tid < nc) {
        const int t = cand_tile[tid];
        CandTile &c = cand[tid];
        float pv[10], bv[8];
        const float4 bb = s.tile_aabb_em[env * kCarMaxTiles + t];
        for (int k = 0; k < 10; k++) pv[k] = s.tile_poly_em[(env * kCarMaxTiles + t) * 10 + k];
        const int border = s.border_em[env * kCarMaxTiles + t];
        // unconditional (the slots of tiles without a border hold zeros): a load that waits for the flag is a second
        // exposed round trip
        for (int k = 0; k < 8; k++) bv[k] = s.border_poly_em[(env * kCarMaxTiles + t) * 8 + k];
        c.v01 = make_float4(pv[0], pv[1], pv[2], pv[3]), c.v23 = make_float4(pv[4], pv[5], pv[6], pv[7]);
        c.v4b0 = make_float4(pv[8], pv[9], bv[0], bv[1]), c.b12 = make_float4(bv[2], bv[3], bv[4], bv[5]);
        c.b3m = make_float4(bv[6], bv[7], __int_as_float(t | (border << 16)), 0.0f);
        float4 bb_all = bb;
        float sx0 = 1e30f, sy0 = 1e30f, sx1 = -1e30f, sy1 = -1e30f;  // conservative screen box of tile + border
        auto grow = [&](float wx, float wy) {
            const V2 tt = rotv(-sn, cs, mk(wx, wy) - off);
            const float X = 48.0f - scale_f * tt.x, Y = 48.0f - scale_f * tt.y;
            sx0 = fminf(sx0, X), sy0 = fminf(sy0, Y), sx1 = fmaxf(sx1, X), sy1 = fmaxf(sy1, Y);
        };
        for (int i = 0; i < 5; i++) grow(pv[2 * i], pv[2 * i + 1]);
        if (border) {
            float x0 = bb.x, y0 = bb.y, x1 = bb.z, y1 = bb.w;
            for (int i = 0; i < 4; i++) {
                x0 = fminf(x0, bv[2 * i]), y0 = fminf(y0, bv[2 * i + 1]), x1 = fmaxf(x1, bv[2 * i]), y1 = fmaxf(y1, bv[2 * i + 1]);
                grow(bv[2 * i], bv[2 * i + 1]);
            }
            bb_all = make_float4(x0, y0, x1, y1);
        }
        c.bb_all = bb_all;
        // culling cells the screen box (+1.5 px) meets
        const int cx0 = max((int)floorf(sx0 - 1.5f), 0) >> kCellShift, cx1 = min((int)ceilf(sx1 + 1.5f), 95) >> kCellShift;
        const int cy0 = max((int)floorf(sy0 - 1.5f), 0) >> kCellShift, cy1 = min((int)ceilf(sy1 + 1.5f), 95) >> kCellShift;
        const bool any = cx1 >= cx0 && cy1 >= cy0 && sx1 + 1.5f >= 0.f && sy1 + 1.5f >= 0.f && sx0 - 1.5f <= 95.f && sy0 - 1.5f <= 95.f;
        cand_cells[tid] = any ? (uint16_t)(cx0 | (cx1 << 4) | (cy0 << 8) | (cy1 << 12)) : (uint16_t)0xFFFF;
    }

    // ---- (2) car polygons (wavefront 2) and indicator rectangles (wavefront 3), next to the candidate records (0, 1)
    if (tid >= 128 && tid < 128 + 8 * s.players) {
        const int ct_ = tid - 128;
        const int k = ct_ >> 3, part = ct_ & 7;  // car k; parts 0..3 wheels, 4..7 hull fixtures
        const int64_t ci = k * n + env;
        const int o = part < 4 ? 6 + 6 * part : 0;
        const float bx = s.body[(o + 0) * M + ci], by = s.body[(o + 1) * M + ci], ba = s.body[(o + 2) * M + ci];
        float bs, bc;
        crl_sincosf(ba, &bs, &bc);
        const V2 lc = part < 4 ? mk(0.f, 0.f) : mk(K.hull_lc[0], K.hull_lc[1]);
        const V2 bp = mk(bx, by) - rotv(bs, bc, lc);
        CarPoly q;
        q.n = part < 4 ? 4 : K.hull_n[part - 4];
        q.x0 = 1 << 30, q.y0 = 1 << 30, q.x1 = -(1 << 30), q.y1 = -(1 << 30);
        for (int i = 0; i < 8; i++) {
            if (i < q.n) {
                const V2 v = part < 4 ? mk(K.wheel_poly[i][0], K.wheel_poly[i][1]) : mk(K.hull_poly[part - 4][i][0], K.hull_poly[part - 4][i][1]);
                const V2 wv = rotv(bs, bc, v) + bp;
                const V2 d = wv - off;
                const V2 t = rotv(-sn, cs, d);
                const float X = (-scale_f) * t.x + 48.0f, Y = (-scale_f) * t.y + 48.0f;
                q.px[i] = (int)X, q.py[i] = (int)Y;
                q.x0 = min(q.x0, q.px[i]), q.x1 = max(q.x1, q.px[i]), q.y0 = min(q.y0, q.py[i]), q.y1 = max(q.y1, q.py[i]);
            } else {
                q.px[i] = q.py[i] = 0;
            }
        }
        q.gray = part < 4 ? 0 : (k == viewer ? G_OWN : G_OTHER);
        cars[ct_] = q;
    } else if (tid >= 192 && tid < 200) {
        const int r = tid - 192;
        const double S = 96 / 40.0, Hh = 96 / 40.0;
        IndRect q;
        if (r == 0) q = make_rect(0, 96 - 4 * Hh, 96, 4 * Hh * 1000, 0);
        else if (r == 1) q = make_rect(5 * S, 96 - Hh, S, Hh * (-0.02 * sqrt(vx * vx + vy * vy)), G_BLUE);
        else if (r < 6) {
            const int w = r - 2;
            q = make_rect((7 + w) * S, 96 - Hh, S, Hh * (-0.01 * s.womega[w * M + me]), w < 2 ? G_BLUE : G_ABS_REAR);
        } else if (r == 6) {
            const double ja = (double)(s.body[(6 + 2) * M + me] - h_a - 0.0f);
            q = make_rect(20 * S, 96 - 2 * Hh, S * (10.0 * ja), 2 * Hh, G_GREEN);
        } else {
            q = make_rect(30 * S, 96 - 2 * Hh, S * (0.8 * (double)h_w), 2 * Hh, G_RED);
        }
        ind[r] = q;
    }
    __syncthreads();
    CAR_TICK(2)

    // ---- (2b) screen-space culling: per 8x8-pixel cell the candidates (as bit masks, i.e. in draw
    // order) whose tile polygon / border quad may cover one of its pixel centres.  Membership
    // itself is still decided in world space, so culling never changes a pixel.
    if (tid < s.players) {
        int x0 = 1 << 30, y0 = 1 << 30, x1 = -(1 << 30), y1 = -(1 << 30);
        for (int p = 0; p < 8; p++) {
            const CarPoly &q = cars[tid * 8 + p];
            x0 = min(x0, q.x0), y0 = min(y0, q.y0), x1 = max(x1, q.x1), y1 = max(y1, q.y1);
        }
        car_box[tid][0] = x0, car_box[tid][1] = x1, car_box[tid][2] = y0, car_box[tid][3] = y1;
    }
    if (tid == 32) {
        int y0 = 1 << 30;
        for (int r = 0; r < 8; r++) y0 = min(y0, ind[r].y0);
        ind_y0 = y0;
    }
    if (wave == 1) {  // scanline work items of the car polygons: rows [max(y0,0), min(y1,95)] each, prefix over 16
        const int k = lane & 15;
        int rows = 0;
        if (k < 8 * s.players) rows = max(min(cars[k].y1, 95) - max(cars[k].y0, 0) + 1, 0);
        int inc = rows;
#pragma unroll
        for (int d = 1; d < 16; d <<= 1) {
            const int o = __shfl_up(inc, d, 16);
            if (k >= d) inc += o;
        }
        if (lane < 16) poly_row0[lane] = inc - rows;
        if (lane == 15) poly_row0[16] = inc;
    }
    if (wave == 1) {  // exclusive prefix of the candidates' cell counts (kMaxCand = 2 per lane)
        int cnt[2];
        for (int h = 0; h < 2; h++) {
            const int c = 2 * lane + h;
            const unsigned r = c < nc ? cand_cells[c] : 0xFFFFu;
            cnt[h] = r == 0xFFFFu ? 0 : (int)(((r >> 4) & 15u) - (r & 15u) + 1u) * (int)(((r >> 12) & 15u) - ((r >> 8) & 15u) + 1u);
        }
        int inc = cnt[0] + cnt[1];
        for (int d = 1; d < 64; d <<= 1) {
            const int o = __shfl_up(inc, d);
            if (lane >= d) inc += o;
        }
        const int ex = inc - cnt[0] - cnt[1];
        cand_ofs[2 * lane] = (uint16_t)ex, cand_ofs[2 * lane + 1] = (uint16_t)(ex + cnt[0]);
        if (lane == 63) cand_ofs[kMaxCand] = (uint16_t)inc;
    }
    for (int i = tid; i < kCells * (kMaxCand / 32); i += 256) (&cell_tmask[0][0])[i] = 0u, (&cell_bmask[0][0])[i] = 0u;
    __syncthreads();
    CAR_TICK(3)
    // one work item per (candidate, cell of its screen box): a separating-edge test of the cell's
    // pixel centres against the tile polygon and the border quad in screen space (the world ->
    // screen map is a similarity with positive determinant, so "inside" stays cross >= 0; a
    // quarter-pixel margin covers rounding).
    const int n_items = (dbg & 16) ? 0 : cand_ofs[kMaxCand];
    for (int item = tid; item < n_items; item += 256) {
        int lo = 0, hi = kMaxCand - 1;  // largest c with cand_ofs[c] <= item
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if ((int)cand_ofs[mid] <= item) lo = mid;
            else hi = mid - 1;
        }
        const int c = lo, j = item - (int)cand_ofs[c];
        const unsigned r = cand_cells[c];
        const int bx0 = r & 15u, cw = (int)((r >> 4) & 15u) - bx0 + 1, by0 = (r >> 8) & 15u;
        const int jy = (int)(((float)j + 0.5f) * (1.0f / (float)cw));
        const int cxi = bx0 + (j - jy * cw), cyi = by0 + jy, cell = cyi * kCellsPerRow + cxi;
        const int cx0 = cxi * kCell, cy0 = cyi * kCell;
        float4 te[5], be[4], ba_;
        int idx_, border_;
        cand_expand(cand[c], te, be, ba_, idx_, border_);  // back-to-back LDS reads, no early-out chain
        (void)ba_, (void)idx_;
        const float px0 = (float)cx0 + 0.25f, px1 = (float)(cx0 + kCell) - 0.25f, py0 = (float)cy0 + 0.25f, py1 = (float)(cy0 + kCell) - 0.25f;
        auto edge_rejects = [&](const float4 &q) {  // every pixel centre of the cell is outside this edge
            const V2 ta = rotv(-sn, cs, mk(q.x, q.y) - off), tb = rotv(-sn, cs, mk(q.x + q.z, q.y + q.w) - off);
            const float ax = 48.0f - scale_f * ta.x, ay = 48.0f - scale_f * ta.y;
            const float dx = (48.0f - scale_f * tb.x) - ax, dy = (48.0f - scale_f * tb.y) - ay;
            const float tol = -0.25f * (fabsf(dx) + fabsf(dy));
            const float c00 = dx * (py0 - ay) - dy * (px0 - ax), c10 = dx * (py0 - ay) - dy * (px1 - ax);
            const float c01 = dx * (py1 - ay) - dy * (px0 - ax), c11 = dx * (py1 - ay) - dy * (px1 - ax);
            return fmaxf(fmaxf(c00, c10), fmaxf(c01, c11)) < tol;
        };
        bool may_t = true, may_b = true;
#pragma unroll
        for (int i = 0; i < 5; i++) may_t = may_t && !edge_rejects(te[i]);
#pragma unroll
        for (int i = 0; i < 4; i++) may_b = may_b && !edge_rejects(be[i]);
        if (may_t) atomicOr(&cell_tmask[cell][c >> 5], 1u << (c & 31));
        if (border_ && may_b) atomicOr(&cell_bmask[cell][c >> 5], 1u << (c & 31));
    }
    __syncthreads();
    CAR_TICK(4)

    // ---- (3a) background into the LDS tile: 4 consecutive pixels per thread-iteration share one
    // culling cell, so every candidate's edges are read from LDS once per 4 pixels
    // A wavefront takes one 16x16-pixel block per iteration (2x2 culling cells), so its lanes walk
    // nearly the same candidates.
    for (int blk = wave; blk < 36 && !(dbg & 32); blk += 4) {
        const int sy = (blk / 6) * 16 + (lane >> 2), sx0 = (blk % 6) * 16 + (lane & 3) * 4;
        const int q = sy * 24 + (sx0 >> 2);
        float wx[4], wy[4];
        int g[4];
        float ax0 = 3.4e38f, ay0 = 3.4e38f, ax1 = -3.4e38f, ay1 = -3.4e38f;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const float dx = ((float)(sx0 + k) + 0.5f) - 48.0f, dy = ((float)sy + 0.5f) - 48.0f;
            const float rx = cs * dx - sn * dy, ry = sn * dx + cs * dy;
            wx[k] = off.x - rx * inv_scale, wy[k] = off.y - ry * inv_scale;
            g[k] = G_GRASS;  // the checker term is resolved after the polygons, for the pixels still open
            ax0 = fminf(ax0, wx[k]), ay0 = fminf(ay0, wy[k]), ax1 = fmaxf(ax1, wx[k]), ay1 = fmaxf(ay1, wy[k]);
        }
        unsigned open = 0xFu;  // pixels not yet covered by a (later-drawn) polygon
        const int cell = (sy >> kCellShift) * kCellsPerRow + (sx0 >> kCellShift);
        // (both masks of the cell in two 16-byte reads, not word by word)
        const uint4 tm4 = *reinterpret_cast<const uint4 *>(cell_tmask[cell]), bm4 = *reinterpret_cast<const uint4 *>(cell_bmask[cell]);
        static_assert(kMaxCand == 128, "one uint4 per cell mask");
#pragma unroll
        for (int wd = 0; wd < kMaxCand / 32; wd++) {
          if (!open || (dbg & 2)) break;
          const uint32_t tm = wd == 0 ? tm4.x : wd == 1 ? tm4.y : wd == 2 ? tm4.z : tm4.w;
          const uint32_t bmk = wd == 0 ? bm4.x : wd == 1 ? bm4.y : wd == 2 ? bm4.z : bm4.w;
          uint32_t todo = tm | bmk;
          while (todo && open) {
            const int bit = __ffs(todo) - 1;
            todo &= todo - 1u;
            const int c = wd * 32 + bit;
            const bool do_tile = (tm >> bit) & 1u, do_border = (bmk >> bit) & 1u;
            // everything the candidate needs comes in with back-to-back LDS reads (no data-dependent
            // early-out between them: a chain of dependent ~100-cycle reads costs more than the
            // arithmetic it would save)
            float4 te[5], be[4], ba;
            int border, tidx;
            cand_expand(cand[c], te, be, ba, tidx, border);
            if (ax0 > ba.z || ax1 < ba.x || ay0 > ba.w || ay1 < ba.y) continue;
            if (do_border) {  // the border quad is drawn right after its tile, so it is tested first
                unsigned in = open;
#pragma unroll
                for (int i = 0; i < 4; i++) {
#pragma unroll
                    for (int k = 0; k < 4; k++)
                        if ((be[i].z * (wy[k] - be[i].y) - be[i].w * (wx[k] - be[i].x)) < 0) in &= ~(1u << k);
                }
                const int bg = border == 1 ? G_WHITE : G_RED;
#pragma unroll
                for (int k = 0; k < 4; k++)
                    if (in >> k & 1u) g[k] = bg;
                open &= ~in;
            }
            if (!do_tile) continue;
            unsigned in = open;  // (a point outside the polygon's box fails one of the five edge tests anyway)
#pragma unroll
            for (int i = 0; i < 5; i++) {
#pragma unroll
                for (int k = 0; k < 4; k++)
                    if ((te[i].z * (wy[k] - te[i].y) - te[i].w * (wx[k] - te[i].x)) < 0) in &= ~(1u << k);
            }
            const int rg = tidx % 3 == 0 ? 101 : (tidx % 3 == 1 ? 103 : 107);
#pragma unroll
            for (int k = 0; k < 4; k++)
                if (in >> k & 1u) g[k] = rg;
            open &= ~in;
          }
        }
        // grass checker of the open pixels: floor(w / kf) with kf = PLAYFIELD / 20.  The quotient is
        // first taken as w * (1 / kf) (error < 4e-6 for |w / kf| < 32); only when that lands within
        // 1e-4 of an integer -- where the rounding of the true f32 division could matter -- is the
        // division itself evaluated, so the result is always floorf(w / kf).
#pragma unroll
        for (int k = 0; k < 4; k++) {
            if (!(open >> k & 1u)) continue;
            const float qx = wx[k] * inv_kf, qy = wy[k] * inv_kf;
            float fx = floorf(qx), fy = floorf(qy);
            const float rx = qx - fx, ry = qy - fy;
            if (!(fabsf(qx) < 32.0f) || rx < 1e-4f || rx > 1.0f - 1e-4f) fx = floorf(wx[k] / kf);
            if (!(fabsf(qy) < 32.0f) || ry < 1e-4f || ry > 1.0f - 1e-4f) fy = floorf(wy[k] / kf);
            const int ix = (int)fx, iy = (int)fy;
            const bool light = ix >= -20 && ix <= 18 && iy >= -20 && iy <= 18 && (ix & 1) == 0 && (iy & 1) == 0;
            if (light) g[k] = G_LIGHT;
        }
        tile32[q] = (uint32_t)g[0] | ((uint32_t)g[1] << 8) | ((uint32_t)g[2] << 16) | ((uint32_t)g[3] << 24);
    }
    __syncthreads();
    CAR_TICK(5)
    uint8_t *tile8 = reinterpret_cast<uint8_t *>(tile32);

    // ---- (3b) cars: one work item per (car polygon, scanline): the crossings of pygame's
    // draw_fillpoly are computed once per row and the spans written with an LDS atomicMax on
    // (draw rank << 8 | gray) -- draw order is car 0 wheels, hull, then car 1 -- into a scratch
    // image of each car's screen box, which is then copied over the background.
    if (!(dbg & 1)) {
        static_assert(sizeof(CandTile) * kMaxCand >= 2 * kRankCap * sizeof(uint32_t), "rank scratch must fit the candidate array");
        uint32_t *rank = reinterpret_cast<uint32_t *>(cand);  // candidates are dead after (3a): [2][kRankCap] scratch
        int bx0[2], by0[2], bw[2], bh[2];
        for (int c = 0; c < 2; c++) {
            bx0[c] = by0[c] = bw[c] = bh[c] = 0;
            if (c < s.players) {
                const int x0 = max(car_box[c][0], 0), x1 = min(car_box[c][1], 95), y0 = max(car_box[c][2], 0), y1 = min(car_box[c][3], 95);
                const int w = x1 - x0 + 1, h = y1 - y0 + 1;
                if (w > 0 && h > 0 && w * h <= kRankCap) bx0[c] = x0, by0[c] = y0, bw[c] = w, bh[c] = h;  // (a car never spans more than ~20x20 px)
            }
        }
        for (int c = 0; c < 2; c++)
            for (int p = tid; p < bw[c] * bh[c]; p += 256) rank[c * kRankCap + p] = 0u;
        __syncthreads();
        CAR_TICK(6)
        // one work item per (polygon, scanline); the item -> polygon map is a 16-entry prefix table
        const int4 pr0 = *reinterpret_cast<const int4 *>(&poly_row0[0]), pr1 = *reinterpret_cast<const int4 *>(&poly_row0[4]);
        const int4 pr2 = *reinterpret_cast<const int4 *>(&poly_row0[8]), pr3 = *reinterpret_cast<const int4 *>(&poly_row0[12]);
        const int total = poly_row0[16];
        for (int item = tid; item < total; item += 256) {
            const int k = (item >= pr0.y) + (item >= pr0.z) + (item >= pr0.w) + (item >= pr1.x) + (item >= pr1.y) + (item >= pr1.z) +
                          (item >= pr1.w) + (item >= pr2.x) + (item >= pr2.y) + (item >= pr2.z) + (item >= pr2.w) + (item >= pr3.x) +
                          (item >= pr3.y) + (item >= pr3.z) + (item >= pr3.w);
            const int r = item - poly_row0[k];
            // the polygon record comes in with eleven back-to-back 8-byte LDS reads; everything below
            // indexes registers statically (fully unrolled over the 8 possible vertices)
            int px[8], py[8], pn, px0, px1, py0, py1, pgray;
            {
                const int2 *src = reinterpret_cast<const int2 *>(&cars[k]);
                int2 a[11];
#pragma unroll
                for (int i = 0; i < 11; i++) a[i] = src[i];
#pragma unroll
                for (int i = 0; i < 4; i++) px[2 * i] = a[i].x, px[2 * i + 1] = a[i].y, py[2 * i] = a[4 + i].x, py[2 * i + 1] = a[4 + i].y;
                pn = a[8].x, px0 = a[8].y, px1 = a[9].x, py0 = a[9].y, py1 = a[10].x, pgray = a[10].y;
            }
            const int c = k >> 3, part = k & 7;
            if (bw[c] == 0) continue;
            const int y = max(py0, 0) + r;
            int xs[8];  // crossings of scanline y with the polygon's edges (pygame draw_fillpoly), INT_MAX = none
#pragma unroll
            for (int i = 0; i < 8; i++) xs[i] = 0x7FFFFFFF;
            if (py0 == py1) {
                xs[0] = px0, xs[1] = px1;
            } else {
                int lx = px[0], ly = py[0];  // last vertex = predecessor of vertex 0
#pragma unroll
                for (int i = 1; i < 8; i++)
                    if (i == pn - 1) lx = px[i], ly = py[i];
#pragma unroll
                for (int i = 0; i < 8; i++) {
                    if (i < pn) {
                        const int xp = i ? px[i - 1] : lx, yp = i ? py[i - 1] : ly;
                        int y1 = yp, y2 = py[i], x1 = xp, x2 = px[i];
                        if (y1 > y2) y2 = yp, y1 = py[i], x2 = xp, x1 = px[i];
                        if (y1 != y2 && ((y >= y1 && y < y2) || (y == py1 && y > y1 && y <= y2))) xs[i] = (y - y1) * (x2 - x1) / (y2 - y1) + x1;
                    }
                }
                // sort ascending (8-input odd-even merge network; INT_MAX entries end up last)
#define CRL_CE(a, b)                                      \
    {                                                     \
        const int lo_ = min(xs[a], xs[b]), hi_ = max(xs[a], xs[b]); \
        xs[a] = lo_, xs[b] = hi_;                         \
    }
                CRL_CE(0, 1) CRL_CE(2, 3) CRL_CE(4, 5) CRL_CE(6, 7) CRL_CE(0, 2) CRL_CE(1, 3) CRL_CE(4, 6) CRL_CE(5, 7) CRL_CE(1, 2) CRL_CE(5, 6)
                CRL_CE(0, 4) CRL_CE(3, 7) CRL_CE(1, 5) CRL_CE(2, 6) CRL_CE(1, 4) CRL_CE(3, 6) CRL_CE(2, 4) CRL_CE(3, 5) CRL_CE(3, 4)
#undef CRL_CE
            }
            const uint32_t key = ((uint32_t)(part + 1) << 8) | (uint32_t)pgray;
            uint32_t *row = rank + c * kRankCap + (y - by0[c]) * bw[c] - bx0[c];
#pragma unroll
            for (int i = 0; i < 8; i += 2)
                if (xs[i + 1] != 0x7FFFFFFF)
                    for (int x = max(xs[i], max(px0, 0)); x <= min(xs[i + 1], min(px1, 95)); x++) atomicMax(&row[x], key);
        }
        __syncthreads();
        CAR_TICK(7)
        for (int c = 0; c < 2; c++) {  // car 1 is drawn over car 0
            for (int p = tid; p < bw[c] * bh[c]; p += 256) {
                const uint32_t r = rank[c * kRankCap + p];
                if (r) {
                    const int yy = p / bw[c];
                    tile8[(by0[c] + yy) * 96 + bx0[c] + (p - yy * bw[c])] = (uint8_t)(r & 255u);
                }
            }
            __syncthreads();
            CAR_TICK(8)
        }
    }

    // ---- (3c) indicator strip: drawn last; later rectangles win
    {
        const int y0 = max(ind_y0, 0);
        const int npx = (96 - y0) * 96;
        // the eight rectangles in registers (ten 16-byte LDS reads), then branch-free: later rectangles win
        int rx0[8], rx1[8], ry0[8], ry1[8], rg[8];
        {
            const int4 *src = reinterpret_cast<const int4 *>(&ind[0]);
            int v[40];
#pragma unroll
            for (int i = 0; i < 10; i++) {
                const int4 q4 = src[i];
                v[4 * i] = q4.x, v[4 * i + 1] = q4.y, v[4 * i + 2] = q4.z, v[4 * i + 3] = q4.w;
            }
#pragma unroll
            for (int r = 0; r < 8; r++) rx0[r] = v[5 * r], rx1[r] = v[5 * r + 1], ry0[r] = v[5 * r + 2], ry1[r] = v[5 * r + 3], rg[r] = v[5 * r + 4];
        }
        for (int p = tid; p < npx && !(dbg & 4); p += 256) {
            const int yy = p / 96, sx = p - yy * 96, sy = y0 + yy;
            int gray = -1;
#pragma unroll
            for (int r = 0; r < 8; r++)
                if (sx >= rx0[r] && sx <= rx1[r] && sy >= ry0[r] && sy <= ry1[r]) gray = rg[r];
            if (gray >= 0) tile8[sy * 96 + sx] = (uint8_t)gray;
        }
    }
    __syncthreads();
    CAR_TICK(9)

    // ---- (3c') reward read-out "%05.0f" (white, 1-bit glyphs) blitted last at (0, 91)
    if (s.text_bits && tid < 32 * CRL_CAR_TEXT_ROWS) {
        const int row = tid >> 5, col = tid & 31, sy = 91 + row;
        if (sy < 96 && ((text_row_pre >> col) & 1u)) tile8[sy * 96 + col] = 255;
    }
    __syncthreads();
    CAR_TICK(10)

    // ---- (3d) stream the tile out: 16 B per lane, 1 KiB contiguous per wave store
    uint4 *__restrict__ out = reinterpret_cast<uint4 *>(obs + ((int64_t)env * s.players + viewer) * (96 * 96));
    const uint4 *tile4 = reinterpret_cast<const uint4 *>(tile32);
    {
        uint4 ov[3];  // 576 chunks = 2.25 per thread: LDS reads first, then the stores
#pragma unroll
        for (int i = 0; i < 3; i++) ov[i] = tile4[min(tid + 256 * i, 96 * 96 / 16 - 1)];
#pragma unroll
        for (int i = 0; i < 3; i++)
            if (tid + 256 * i < 96 * 96 / 16) out[tid + 256 * i] = ov[i];
    }
    CAR_TICK(11)
    if (TICKS && (dbg & 64) && tid == 0) {
        for (int i = 0; i < 12; i++) atomicAdd(&g_car_ticks[i], (unsigned long long)tick_acc[i]);
        atomicAdd(&g_car_ticks[23], 1ull);
    }
}

// The kernels proper.  Tile slot b -> (position i = 8 (b / 16) + b % 8, viewer (b % 16) / 8) when there are two views: workgroup b
// runs on XCD b % 8, so the two views of an env are slots b and b + 8 -- same XCD, dispatched together: the second view finds
// the env's track (tile boxes, polygons: most of what this kernel fetches) in that XCD's L2.
//   car_raster_kernel       position i = env i (all envs, or the subset only_env[env] == want: the big launch of a step);
//   car_raster_list_kernel  position i = entry i of a compacted list whose length lives in device memory (the small env
//                           classes of a step): a launch sized from the previous step's count that LOOPS if it falls short,
//                           instead of 32 768 workgroups that each wait for a CU slot only to exit.  The tile is a real function
//                           call there, the context read through pointers to its copy in device memory: inlined into the
//                           loop, the ~100 scalar registers of context pointers stay live across it and spill (108 VGPRs of
//                           spills), and by-value structs would be copied to every lane's private memory at the call -- a
//                           20 % slower tile, which the big launch must not pay.
template <bool TICKS>
__global__ __launch_bounds__(256, CRL_CAR_RASTER_WAVES) void car_raster_kernel(CarSoA s, CarConsts K, uint8_t *__restrict__ obs, int dbg,
                                                         const uint8_t *__restrict__ only_env, int want) {
    int64_t env = blockIdx.x;
    int viewer = 0;
    if (s.players == 2) {
        const int r = (int)(blockIdx.x & 15);
        env = (int64_t)(blockIdx.x >> 4) * 8 + (r & 7), viewer = r >> 3;
    }
    if (env >= s.n) return;
    if (only_env && only_env[env] != want) return;  // env subset: finished envs / one class of the step pipeline
    car_raster_tile<TICKS>(s, K, obs, dbg, env, viewer);
}

__device__ __noinline__ void car_raster_tile_call(const CarSoA *__restrict__ sp, const CarConsts *__restrict__ Kp, uint8_t *__restrict__ obs,
                                                  int dbg, int64_t env, int viewer) {
    car_raster_tile<false>(*sp, *Kp, obs, dbg, env, viewer);
}

__global__ __launch_bounds__(256, CRL_CAR_RASTER_WAVES) void car_raster_list_kernel(const CarSoA *__restrict__ sp, const CarConsts *__restrict__ Kp,
                                                              int players, uint8_t *__restrict__ obs, int dbg, const int32_t *__restrict__ list,
                                                              const int32_t *__restrict__ list_count, int32_t *__restrict__ count_to_host) {
    const int64_t positions = *list_count;
    if (count_to_host && blockIdx.x == 0 && threadIdx.x == 0) *count_to_host = (int32_t)positions;
    const int64_t slots = players == 2 ? (positions + 7) / 8 * 16 : positions;
    for (int64_t b = blockIdx.x; b < slots; b += gridDim.x) {  // (gridDim.x is a multiple of 16)
        int64_t i = b;
        int viewer = 0;
        if (players == 2) {
            const int r = (int)(b & 15);
            i = (b >> 4) * 8 + (r & 7), viewer = r >> 3;
        }
        if (i < positions) car_raster_tile_call(sp, Kp, obs, dbg, (int64_t)list[i], viewer);
        if (b + gridDim.x < slots) __syncthreads();  // the next tile reuses the LDS
    }
}

void car_raster_print_ticks() {
    unsigned long long t[24];
    if (hipMemcpyFromSymbol(t, HIP_SYMBOL(g_car_ticks), sizeof(t)) != hipSuccess || !t[23]) return;
    fprintf(stderr, "car_raster phases, mean cycles per workgroup over %llu workgroups:", t[23]);
    for (int i = 0; i < 23; i++)
        if (t[i]) fprintf(stderr, " [%d] %llu", i, t[i] / t[23]);
    fprintf(stderr, "\n");
}

void launch_car_raster(const CarSoA &s, const CarConsts &k, uint8_t *obs, hipStream_t st, const uint8_t *only_env, int want) {
    static const int dbg = getenv("CRL_CAR_DEBUG") ? atoi(getenv("CRL_CAR_DEBUG")) : 0;
    const unsigned grid = s.players == 2 ? (unsigned)((s.n + 7) / 8 * 16) : (unsigned)s.n;  // two views x groups of 8 envs (see the kernel)
    if (dbg & 64) hipLaunchKernelGGL(car_raster_kernel<true>, dim3(grid), dim3(256), 0, st, s, k, obs, dbg, only_env, want);
    else hipLaunchKernelGGL(car_raster_kernel<false>, dim3(grid), dim3(256), 0, st, s, k, obs, dbg, only_env, want);
}

// the envs of a compacted list (its length in device memory); `expected` = the caller's guess of that length, only for the grid size
void launch_car_raster_list(const CarSoA &s, const CarConsts &k, uint8_t *obs, hipStream_t st, const int32_t *list, const int32_t *list_count,
                            int32_t *count_to_host, int64_t expected) {
    static const int dbg = getenv("CRL_CAR_DEBUG") ? atoi(getenv("CRL_CAR_DEBUG")) : 0;
    int64_t want = expected + expected / 4 + 32;  // slack: a launch that falls short loops, it does not miss tiles
    want = want > s.n ? s.n : want;
    const unsigned grid = (unsigned)((want + 7) / 8 * 16);
    hipLaunchKernelGGL(car_raster_list_kernel, dim3(grid), dim3(256), 0, st, s.self_dev, s.consts_dev, s.players, obs, dbg & ~64, list, list_count,
                       count_to_host);
}

}  // namespace crl
#endif  // CRL_ABLATION
