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
#include "car_obs_tile.h"
#include "crl_internal.h"

namespace crl {

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

__device__ void car_camera_tile(const CarSoA &s, const CarConsts &K, int64_t env, int viewer) {
    ViewParams vp;
    float4 cam;
    camera_compute(s, K, env, viewer, vp, cam);
    int32_t *dst = s.view + (env * s.players + viewer) * kViewWords;
    const int32_t *src = reinterpret_cast<const int32_t *>(&vp);
    for (int i = 0; i < 8; i++) dst[i] = src[i];
    reinterpret_cast<float4 *>(dst)[4] = cam;
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
// Camera AND polygons of a tile in one launch (round 5, profiling build only: measured, not kept): 16 lanes per tile, every lane
// evaluates the tile's (uniform) camera itself, lane 0 stores the view, then lane q its polygon -- the same two device functions, one
// launch boundary and one trip through memory less on the bulk stream's chain per-car solve -> view -> frames.
#ifdef CRL_ABLATION
__global__ __launch_bounds__(64) void car_view_kernel(CarSoA s, CarConsts K, const uint8_t *__restrict__ filter, int want) {
    const int64_t t = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 4);
    if (t >= s.n * s.players) return;
    const int64_t env = t / s.players;
    if (filter && filter[env] != want) return;
    const int viewer = (int)(t % s.players), q = threadIdx.x & 15;
    ViewParams vp;
    float4 cam;
    camera_compute(s, K, env, viewer, vp, cam);
    int32_t *dst = s.view + t * kViewWords;
    if (q == 0) {
        const int32_t *src = reinterpret_cast<const int32_t *>(&vp);
        for (int i = 0; i < 8; i++) dst[i] = src[i];
        reinterpret_cast<float4 *>(dst)[4] = cam;
    }
    s.view_cnt[t * 16 + q] = (uint8_t)poly_compute(s, K, env, viewer, q, cam, reinterpret_cast<uint32_t *>(dst) + 8, s.view_rec + (t * 16 + q) * kSpanSlots);
}
#endif
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

// The big launch, one wavefront per THIRD of a tile (car_obs_third): workgroup slot b -> env 8 (b / 48) + b % 8 and item (b % 48) / 8 =
// 3 viewer + third, so that the six wavefronts of an env (b, b + 8, ...) run on one XCD and share its L2's copy of the map blocks.
__global__ __launch_bounds__(64) void car_obs_third_kernel(CarSoA s, uint8_t *__restrict__ obs, const uint8_t *__restrict__ only_env, int want) {
    __shared__ __attribute__((aligned(16))) uint32_t tile[32 * kPitch];
    const int per = 3 * s.players;  // wavefronts per env
    const int64_t g = blockIdx.x / (8 * per);
    const int r = (int)(blockIdx.x - g * (8 * per));
    const int64_t env = g * 8 + (r & 7);
    const int item = r >> 3, viewer = item / 3, third = item - 3 * viewer;
    if (env >= s.n) return;
    if (only_env && only_env[env] != want) return;
    const int64_t t = env * s.players + viewer;
    car_obs_third(s, obs, env, viewer, third, tile, s.view + t * kViewWords, s.view_rec + t * kViewRecWords, s.view_cnt + t * 16);
}

// ---- a LONG list (round 5: the touching envs' frames, ~1 150 envs = 2 300 tiles behind the step's longest chain) in TWO light launches
// instead of the one-wavefront-per-tile list kernel below: (1) camera + polygons of the listed envs into s.view* (16 lanes per tile: the
// double-double camera, ~15 us, then the polygon spans, ~20 us -- 575 wavefronts); (2) the gather in THIRDS of a tile (63 registers,
// 3.5 KB of LDS: 6 900 wavefronts, all resident at once).  The list kernel walks camera -> polygons -> nine gather rounds in ONE
// wavefront at 168 registers (three per SIMD): 75 us per tile alone, 215-295 us for the 2 300 tiles beside the tail of the big frame launch.
__global__ __launch_bounds__(64) void car_view_list_kernel(CarSoA s, CarConsts K, const int32_t *__restrict__ list, const int32_t *__restrict__ list_count,
                                                           const uint8_t *__restrict__ filter, int want, int urgent) {
    if (urgent) __builtin_amdgcn_s_setprio(3);
    const int64_t tiles = (int64_t)*list_count * s.players;
    for (int64_t t0 = (int64_t)blockIdx.x * 4; t0 < tiles; t0 += (int64_t)gridDim.x * 4) {
        const int64_t b = t0 + (threadIdx.x >> 4);
        if (b >= tiles) continue;
        const int64_t env = list[b / s.players];
        if (filter && filter[env] != want) continue;
        const int viewer = (int)(b % s.players), q = threadIdx.x & 15;
        const int64_t t = env * s.players + viewer;
        ViewParams vp;
        float4 cam;
        camera_compute(s, K, env, viewer, vp, cam);
        int32_t *dst = s.view + t * kViewWords;
        if (q == 0) {
            const int32_t *src = reinterpret_cast<const int32_t *>(&vp);
            for (int i = 0; i < 8; i++) dst[i] = src[i];
            reinterpret_cast<float4 *>(dst)[4] = cam;
        }
        s.view_cnt[t * 16 + q] = (uint8_t)poly_compute(s, K, env, viewer, q, cam, reinterpret_cast<uint32_t *>(dst) + 8, s.view_rec + (t * 16 + q) * kSpanSlots);
    }
}
// slot b -> list position 8 (b / 8 per) + b % 8 and item (b % 8 per) / 8 = 3 viewer + third (per = 3 players): an env's wavefronts on one XCD
__global__ __launch_bounds__(64) void car_obs_third_list_kernel(CarSoA s, uint8_t *__restrict__ obs, const int32_t *__restrict__ list,
                                                                const int32_t *__restrict__ list_count, const uint8_t *__restrict__ filter, int want, int urgent) {
    if (urgent) __builtin_amdgcn_s_setprio(3);
    __shared__ __attribute__((aligned(16))) uint32_t tile[32 * kPitch];
    const int per = 3 * s.players;
    const int64_t count = *list_count, slots = (count + 7) / 8 * 8 * per;
    for (int64_t b = blockIdx.x; b < slots; b += gridDim.x) {
        const int64_t g = b / (8 * per);
        const int r = (int)(b - g * (8 * per));
        const int64_t pos = g * 8 + (r & 7);
        if (pos < count) {
            const int64_t env = list[pos];
            if (!filter || filter[env] == want) {
                const int item = r >> 3, viewer = item / 3, third = item - 3 * viewer;
                const int64_t t = env * s.players + viewer;
                car_obs_third(s, obs, env, viewer, third, tile, s.view + t * kViewWords, s.view_rec + t * kViewRecWords, s.view_cnt + t * 16);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");  // (a looping workgroup reuses the LDS rows)
    }
}
// view_list / view_count: the envs whose views are still to be computed (the whole list, or -- when the touching solve's one-manifold wavefronts
// have prepared theirs -- the envs with two manifolds or more)
void launch_car_obs_long_list(const CarSoA &s, const CarConsts &k, uint8_t *obs, hipStream_t st, const int32_t *list, const int32_t *list_count, int64_t expected,
                              const uint8_t *filter, int want_cls, bool urgent, const int32_t *view_list, const int32_t *view_count, int64_t view_expected) {
    int64_t want = expected + expected / 4 + 32;  // slack: a launch that falls short loops, it does not miss tiles
    want = want > s.n ? s.n : want;
    int64_t vwant = view_expected + view_expected / 4 + 32;
    vwant = vwant > s.n ? s.n : vwant;
    hipLaunchKernelGGL(car_view_list_kernel, dim3((unsigned)((vwant * s.players + 3) / 4)), dim3(64), 0, st, s, k, view_list, view_count, filter, want_cls, urgent ? 1 : 0);
    hipLaunchKernelGGL(car_obs_third_list_kernel, dim3((unsigned)((want + 7) / 8 * 8 * 3 * s.players)), dim3(64), 0, st, s, obs, list, list_count, filter, want_cls, urgent ? 1 : 0);
}

// The envs of a compacted list (the small env classes of a step): camera, polygons and tile in ONE launch, one wavefront per
// tile -- every lane computes the (uniform) camera, lanes 0-15 the polygons, everything handed over through LDS.  Three
// dependent launches of a few hundred wavefronts each cost three launch latencies at the end of a step.
// PREPARED (round 5, instantiated in the profiling build only): the envs' views are already in s.view / s.view_rec / s.view_cnt (the
// touching solve's epilogue put them there, car_contact.hip): gather and overlays only.
template <bool PREPARED>
__global__ __launch_bounds__(64, PREPARED ? 4 : 3) void car_obs_list_kernel(CarSoA s, CarConsts K, uint8_t *__restrict__ obs, const int32_t *__restrict__ list,
                                                                         const int32_t *__restrict__ list_count, int32_t *__restrict__ count_to_host,
                                                                         const uint8_t *__restrict__ filter, int want, int urgent) {
    if (urgent) __builtin_amdgcn_s_setprio(3);  // the launch at the end of the step's longest chain, beside the big frame launch's 32 768 wavefronts
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
        if (PREPARED) {
            if (!filter || filter[env] == want) {
                const int64_t t = env * s.players + viewer;
                car_obs_tile(s, obs, env, viewer, tile, s.view + t * kViewWords, s.view_rec + t * kViewRecWords, s.view_cnt + t * 16);
            }
        } else if (!filter || filter[env] == want) {
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
    // (profiling build, round 5: camera + polygons in ONE launch, 16 lanes per tile -- bit-exact, and 1 % slower per step in three A/B
    // pairs: 16 x the wavefronts walk through the double-double camera while the solves need the issue slots; docs/LAB_NOTES_r05.md)
    static const bool merged = CRL_ABL(getenv("CRL_CAR_VIEW_MERGED") != nullptr);
#ifdef CRL_ABLATION
    if (merged) {
        hipLaunchKernelGGL(car_view_kernel, dim3((unsigned)((tiles + 3) / 4)), dim3(64), 0, st, s, k, only_env, want);
        return;
    }
#endif
    (void)merged;
    hipLaunchKernelGGL(car_camera_kernel, dim3((unsigned)((tiles + 63) / 64)), dim3(64), 0, st, s, k, only_env, want);
    hipLaunchKernelGGL(car_poly_kernel, dim3((unsigned)((tiles + 3) / 4)), dim3(64), 0, st, s, k, only_env, want);
}
void launch_car_obs(const CarSoA &s, const CarConsts &k, uint8_t *obs, hipStream_t st, const uint8_t *only_env, int want) {
    // (profiling build, CRL_CAR_OBS_WHOLE_TILES=1: rounds 3-4's one wavefront per tile.  Same box, four alternating pairs of 200-step windows:
    // the frame kernel 267-271 us in thirds against 295-313 whole; the step 0.840-0.842 against 0.838-0.847 ms, fma 0.784-0.791 against 0.793-0.800)
    static const bool whole = CRL_ABL(getenv("CRL_CAR_OBS_WHOLE_TILES") != nullptr);
    if (!whole) {
        hipLaunchKernelGGL(car_obs_third_kernel, dim3((unsigned)((s.n + 7) / 8 * 8 * 3 * s.players)), dim3(64), 0, st, s, obs, only_env, want);
        return;
    }
    const unsigned grid = s.players == 2 ? (unsigned)((s.n + 7) / 8 * 16) : (unsigned)s.n;
    hipLaunchKernelGGL(car_obs_kernel, dim3(grid), dim3(64), 0, st, s, obs, only_env, want);
}

// the envs of a compacted list (its length in device memory; `expected` = the caller's guess of it, only for the grid size),
// optionally only those with filter[env] == want
void launch_car_obs_list(const CarSoA &s, const CarConsts &k, uint8_t *obs, hipStream_t st, const int32_t *list, const int32_t *list_count,
                         int32_t *count_to_host, int64_t expected, const uint8_t *filter, int want_cls, bool urgent, int prepared) {
    int64_t want = expected + expected / 4 + 32;  // slack: a launch that falls short loops, it does not miss tiles
    want = want > s.n ? s.n : want;
#ifdef CRL_ABLATION
    if (prepared) {  // (profiling build: the touching solve's epilogue has prepared the listed envs' views, car_contact.hip)
        hipLaunchKernelGGL(car_obs_list_kernel<true>, dim3((unsigned)(want * s.players)), dim3(64), 0, st, s, k, obs, list, list_count, count_to_host, filter, want_cls, urgent ? 1 : 0);
        return;
    }
#endif
    (void)prepared;
    hipLaunchKernelGGL(car_obs_list_kernel<false>, dim3((unsigned)(want * s.players)), dim3(64), 0, st, s, k, obs, list, list_count, count_to_host, filter, want_cls, urgent ? 1 : 0);
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
