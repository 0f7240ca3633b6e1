// car_device.h -- device-side types shared by the cCarRacingDouble kernels (gfx950).
//
// What is restated (reference paths relative to competitive_rl/car_racing/):
//   wheel model ................ Car.step / gas / brake / steer        car_dynamics.py:131-234
//   step bookkeeping ........... CarRacing.step, process_action       car_racing_multi_players.py:527-620
//   tile rule .................. FrictionDetector._contact                                    :111-153
//   track ...................... CarRacing._create_track, reset                      :262-452, :454-525
//   rigid bodies + joints ...... b2World.Step call site :600.  box2d-py ~=2.3.5 (setup.py:14) is a
//        third-party dependency, absent from the reference tree: Box2D 2.3's published
//        b2Island::Solve / b2RevoluteJoint / b2PolygonShape::ComputeMass are restated in f32.
//        Car-car contacts: car_contact.hip (DESIGN.md "CarRacing: deviations").
//
// Layout: every per-car quantity is an SoA array indexed by car instance ci = car * N + env,
// so lane <-> car instance and all state loads/stores are coalesced; per-env track data is
// [tile][env] so a wave walking tile t reads 64 consecutive envs.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/crl.h"
#include "../../include/crl_f64.h"
#include "../../include/crl_rot.h"

namespace crl {

static constexpr int kCarMaxTiles = CRL_CAR_MAX_TILES;  // 512
static constexpr int kWheelSlots = 6;                   // tiles one wheel can touch at once
static constexpr int kMaxContacts = 8, kContactWords = 20;

// ---- observation map (car_obs.hip): the window [kMapOrg, kMapOrg + kMapW)^2 of the reference's 10000 x 10000
// `observation_playground` (car_racing_multi_players.py:216-222, 732-755) as a 7-colour palette, 4 bits per pixel,
// in 16 x 16-pixel blocks of 128 bytes (one L2 line): block (by, bx) at ((by * kMapBlocks + bx) * 128), row r of a
// block at + 8 r, pixel x of that row in nibble (x & 1) of byte (x >> 1).  A rotated 96 x 96 view touches ~50-64 blocks.
static constexpr int kMapOrg = CRL_CAR_MAP_ORG, kMapW = CRL_CAR_MAP_W;      // 4392, 1216
static constexpr int kMapBlocks = kMapW / 16;                                // 76 per row
static constexpr int64_t kMapBytes = (int64_t)kMapBlocks * kMapBlocks * 128;  // 739 328 per env
static constexpr int kMapSurface = 10000;
#define CRL_CAR_OBS_SCALE 0x1.c37d6d52bdddbp+0  // (10 / (100 / sqrt(96))) * 1.8 (crmp:214-215)
enum { kPalGrass = 0, kPalLight = 1, kPalRoad0 = 2, kPalWhite = 5, kPalRed = 6 };
// per (env, viewer) tile, written by car_view_kernel and read by car_obs_kernel
static constexpr int kViewWords = 20;     // struct ViewParams (16 words) + the float32 camera (sin, cos, offset x, y)
static constexpr int kWalkSaveWords = 18;
static constexpr int kSpanSlots = 16;     // scanline spans one car polygon can produce (a polygon is <= 5.3 px across: <= 8)
static constexpr int kViewRecWords = 16 * kSpanSlots;

// constants of car_racing_multi_players.py:54-88 and car_dynamics.py:17-51
#define CAR_SCALE 6.0
#define CAR_TRACK_RAD (900 / CAR_SCALE)
#define CAR_PLAYFIELD (2000 / CAR_SCALE)
#define CAR_FPS 50
#define CAR_TRACK_DETAIL_STEP (21 / CAR_SCALE)
#define CAR_TRACK_TURN_RATE 0.31
#define CAR_TRACK_WIDTH (40 / CAR_SCALE)
#define CAR_BORDER (8 / CAR_SCALE)
#define CAR_SIZE 0.02
#define CAR_ENGINE_POWER (100000000 * CAR_SIZE * CAR_SIZE)
#define CAR_WHEEL_MOI (4000 * CAR_SIZE * CAR_SIZE)
#define CAR_FRICTION_LIMIT (1000000 * CAR_SIZE * CAR_SIZE)
#define CAR_WHEEL_R 27
#define CAR_WHEEL_W 14

struct CarConsts {  // body constants computed on the host the way b2Body::ResetMassData does
    float hull_poly[4][8][2];
    int hull_n[4];
    float wheel_poly[4][2];
    float hull_inv_mass, hull_inv_I, hull_lc[2];
    float wheel_inv_mass, wheel_inv_I;
    float anchor[4][2];
};

struct CarSoA {
    int64_t n;  // envs
    int players; // cars per env: 2 = cCarRacingDouble-v0, 1 = cCarRacing-v0; M = players * n car instances
    // ---- per car instance [..][M]
    float *body;        // [30][M]: hull (cx,cy,a,vx,vy,w), wheel 0..3 likewise
    float *jimp;        // [12][M]: revolute joint impulse (x,y,z) x 4
    float *jmotor;      // [4][M]  motor impulse
    float *jspeed;      // [4][M]  motor speed (kept: a done car's joints keep their last target)
    int32_t *jlimit;    // [4][M]  limit state
    double *wgas, *womega, *wphase;  // [4][M] wheel attributes (python floats)
    int16_t *wtiles;    // [4][kWheelSlots][M] tiles each wheel touches (-1 = empty)
    uint32_t *visited;  // [16][M] tile.road_visited bits
    double *reward, *prev_reward;  // [M]
    double *step_acc;   // [M] step reward accumulated over the action repeats of the current step
    int32_t *visited_count, *last_block, *done, *step_count, *first_step;  // [M]
    // ---- per env [..][n]
    int32_t *elapsed;   // gym TimeLimit._elapsed_steps
    uint32_t *episode;  // resets so far (RNG counter)
    int32_t *ntiles;
    float4 *tile_aabb;  // [512][n]
    float4 *tile_blk;   // [64][n] union of the boxes of tiles 8 b .. 8 b + 7: what the wheel sensors' broadphase looks at first
    float *tile_poly;   // [512][10][n]  CCW float32 (b2PolygonShape)
    float *border_poly; // [512][8][n]
    uint8_t *border;    // [512][n]  0 none, 1 white, 2 red
    float *start_pose;  // [3][n]    track[0] beta, x, y (birth place)
    // env-major copies of the same track for the raster (one workgroup walks ONE env's tiles)
    float4 *tile_aabb_em;  // [n][512]
    float *tile_poly_em;   // [n][512][10]
    float *border_poly_em; // [n][512][8]
    uint8_t *border_em;    // [n][512]
    double *track_scratch;  // [n][2500][4] points (alpha, beta, x, y; f64) of the walk generated AHEAD for the env's next episode
    double *track_scratch_b;  // same shape: where a reset walks inline when no finished walk-ahead is there
    uint32_t *walk_tag;     // [n] episode index the stored walk belongs to (0xFFFFFFFF = none); written last, device-scope release
    int32_t *walk_list, *walk_count;  // [n], [1]: the envs the current walk-ahead launch has to walk, compacted
    int32_t *walk_len, *walk_first, *walk_swap;  // [n] lap length, its first point, birth-place swap of the stored walk
    uint32_t *walk_save;    // [n][kWalkSaveWords] an unfinished walk-ahead's state between two bounded launches (car_track.hip WalkSave)
    const uint32_t *text_bits;  // reward read-out bitmaps [CRL_CAR_TEXT_STRINGS][CRL_CAR_TEXT_ROWS] or nullptr
    // ---- car-car contacts (players == 2)
    int contacts_enabled;
    int abl_keep_tag;       // profiling build only (CRL_CAR_ABL_KEEP_TAG): a walk-ahead does not void the stored walk's tag before overwriting it
    int abl_no_walk;        // profiling build only (CRL_CAR_ABL_NO_WALK): resets reuse the stored walk, no walk-ahead kernel is launched
    int touch_view;         // this launch of the touching solve also prepares its envs' views (camera, polygon spans): the step pipeline's last world.Step
    int fma;                // CRL_FLAG_CAR_FMA: the island solver's iterations in fused multiply-adds (car_solver.h: FM)
    float *wforce;          // [8][M] tyre forces of this step, handed to the coupled kernel
    float *wsnap;           // [12][M] wheel transforms (cx, cy, angle) the step starts from, for car_sensor_kernel
    uint8_t *sensor_ovf;    // [M] 1 = car_sensor_kernel's lists overflowed for this car: car_sensor_serial_kernel redoes it
    // this struct and the body constants once more in device memory (the raster reads the context through pointers)
    const struct CarSoA *self_dev;
    const struct CarConsts *consts_dev;
    float *sleep;           // [5][M] b2Body::m_sleepTime of hull, wheels 0-3
    int32_t *coupled;       // [n] 1 = the two cars are solved together this step
    int32_t *coupled_list;  // [n] the coupled envs of this step, compacted (any order), and
    int32_t *coupled_count; // [8] how many: [0] coupled (car_step_kernel; car_broad_kernel run ahead: those with final poses, the envs that touch now: [7], from the list's end), then the narrow phase's split: [1] near-only, [2..4] touching
                            //     with one / two / three-or-more manifolds
    int32_t *near_list, *touch_list;  // [n], [3][n]
    int32_t *touch_all;     // [n] every touching env (any manifold count), [5] of coupled_count: their frames
    int32_t *touch_multi;   // [n] the touching envs with two manifolds or more, [6] of coupled_count: their frames when the one-manifold envs have a stream of their own
    int32_t *coupled_to_host;  // host-mapped word: the step's coupled-env count, read by the host one step late (sizes the list launches)
    int32_t *cap_hits;      // [4] times a fixed capacity was hit since create: [0] a wheel touching more than kWheelSlots tiles (the
                            //     extra tile is not recorded), [1] more than kMaxContacts manifolds between two cars (the rest are dropped)
    unsigned long long *stamps;  // [64] phase cycle counters of the coupled kernels (profiling build, -DCRL_ABLATION, only)
    int32_t *zero_next;     // [16] the OTHER step parity's counter block (coupled_count[8] + class counts): car_step_kernel clears it for the next step
    int32_t *nc_new;        // [n] this step's manifold count (car_narrow_kernel)
    float *contact_new;     // [n][kMaxContacts][kContactWords] this step's manifolds with the carried-over impulses
    int32_t *n_contact;     // [n] touching car-car contacts carried to the next step (warm start)
    float *contact;         // [n][16][kContactWords] persisted manifolds + impulses
    // ---- observations as the reference computes them (car_obs.hip)
    uint8_t *obs_map;       // [2][n][kMapBytes] (one slot in the one-stream mode) pre-rastered palette map of the env's track (built at reset), two slots per env: the map of
                            //     a finished env's NEXT episode is built into the other slot while the terminal frame still needs this one
    uint8_t *map_par;       // [n] which slot is current
    int map_alt;            // 1 = this view of the state addresses the OTHER slot (the staged reset of the step pipeline)
    uint32_t *map_vtx;      // [n][512][9] map-space vertices (x | y << 16, int16 each, window coordinates) of tile i (5) and its border (4)
    uint32_t *map_yr;       // [n][512] first | last << 16 map row (int16 each) that tile i or its border touches
    int32_t *map_overflow;  // [n] polygon vertices that fell outside the window at the last reset (0 for every track)
    uint32_t *map_lightx, *map_lighty;  // [kMapW / 32] bit masks: columns / rows of the window covered by the lighter squares
    int32_t *view;          // [tiles][kViewWords] ViewParams
    uint8_t *view_cnt;      // [tiles][16] spans per car polygon
    uint32_t *view_rec;     // [tiles][16][kSpanSlots] spans: y | xl << 8 | xr << 16
};

__device__ __forceinline__ uint8_t *env_map(const CarSoA &s, int64_t env) {
    return s.obs_map + ((int64_t)((s.map_par[env] ^ s.map_alt) & 1) * s.n + env) * kMapBytes;
}

struct ViewParams {  // one (env, viewer) tile: where its 96 x 96 pixels come from, and what is drawn over them
    // source of screen pixel (X, Y) in 16.16 fixed point, in WINDOW coordinates of the map (the crop rectangle folded in):
    //   dx = dx00 + icos * X - isin * Y,  dy = dy00 + isin * X + icos * Y;  map pixel = (dx >> 16, dy >> 16)
    int32_t dx00, dy00, isin, icos;
    int32_t rx, ry;         // top-left corner of the 192 x 192 crop, window coordinates
    int32_t flags;          // bit 0: every source pixel lies inside the window; bit 1: no pixel takes rotate()'s background colour
    int32_t text_idx;       // row of the reward read-out bitmaps, -1 = none
    uint32_t rect[8];       // indicator rectangles, clipped: x0 | x1 << 8 | y0 << 16 | y1 << 24 (x0 > x1: empty)
};
static_assert(sizeof(ViewParams) == 16 * 4 && kViewWords >= 20, "ViewParams layout");

struct V2 {
    float x, y;
};
__host__ __device__ inline V2 mk(float x, float y) { V2 r = {x, y}; return r; }
__host__ __device__ inline V2 operator+(V2 a, V2 b) { return mk(a.x + b.x, a.y + b.y); }
__host__ __device__ inline V2 operator-(V2 a, V2 b) { return mk(a.x - b.x, a.y - b.y); }
__host__ __device__ inline V2 operator*(float s, V2 a) { return mk(s * a.x, s * a.y); }
__host__ __device__ inline float dot(V2 a, V2 b) { return a.x * b.x + a.y * b.y; }
__host__ __device__ inline float cross(V2 a, V2 b) { return a.x * b.y - a.y * b.x; }
__host__ __device__ inline V2 scross(float s, V2 a) { return mk(-s * a.y, s * a.x); }
__host__ __device__ inline V2 rotv(float s, float c, V2 v) { return mk(c * v.x - s * v.y, s * v.x + c * v.y); }

struct Body {
    float cx, cy, a, vx, vy, w;
};

// Conservative "could these two cars touch" test on the hull poses: oriented boxes that contain hull + steered wheels
// (half extents 1.75 x 2.75 about the hull origin, + margin), separating axis test.  Symmetric in its arguments' roles,
// evaluated identically by both lanes of an env.  This only ROUTES the env (coupled solve or per-car solve: the same
// arithmetic for cars that do not touch), so it may use the fast hardware sine / cosine (error ~1e-6, covered by the margin).
__device__ inline void fast_sincosf(float a, float *s, float *c) { *s = __sinf(a), *c = __cosf(a); }
__device__ inline bool cars_near(const CarConsts &K, float x0, float y0, float a0, float x1, float y1, float a1) {
    const float ex = 1.75f + 0.21f, ey = 2.75f + 0.21f;
    float s0, c0, s1, c1;
    fast_sincosf(a0, &s0, &c0), fast_sincosf(a1, &s1, &c1);
    // box centres = hull origins (body origin = centre of mass - R * localCenter)
    const float ox0 = x0 - (c0 * K.hull_lc[0] - s0 * K.hull_lc[1]), oy0 = y0 - (s0 * K.hull_lc[0] + c0 * K.hull_lc[1]);
    const float ox1 = x1 - (c1 * K.hull_lc[0] - s1 * K.hull_lc[1]), oy1 = y1 - (s1 * K.hull_lc[0] + c1 * K.hull_lc[1]);
    const float dx = ox1 - ox0, dy = oy1 - oy0;
    const float ax[4][2] = {{c0, s0}, {-s0, c0}, {c1, s1}, {-s1, c1}};
    for (int i = 0; i < 4; i++) {
        const float ux = ax[i][0], uy = ax[i][1];
        const float r0 = ex * fabsf(c0 * ux + s0 * uy) + ey * fabsf(-s0 * ux + c0 * uy);
        const float r1 = ex * fabsf(c1 * ux + s1 * uy) + ey * fabsf(-s1 * ux + c1 * uy);
        if (fabsf(dx * ux + dy * uy) > r0 + r1) return false;
    }
    return true;
}

struct CarTrackSrc {  // where reset draws come from
    uint64_t seed;
    int64_t env_id_base;
    const double *ru;        // replay: [n][attempts][24] uniforms, or nullptr (Philox)
    const uint8_t *rshuffle; // replay: [n][attempts] birth-place swap bit
    int64_t attempts;
};

void launch_car_reset(const CarSoA &s, const CarConsts &k, const CarTrackSrc &src, bool only_done, const uint8_t *done_env,
                      hipStream_t st);
// live <- stage for the car state of the listed envs, and the envs' map slots flipped (the staged reset of the step pipeline)
void launch_car_commit_list(const CarSoA &live, const CarSoA &stage, const int32_t *list, const int32_t *list_count, int64_t expected, hipStream_t st);
void launch_car_reset_list(const CarSoA &s, const CarConsts &k, const CarTrackSrc &src, const int32_t *list, const int32_t *list_count,
                           int64_t expected, hipStream_t st);
void launch_car_walk_ahead(const CarSoA &s, const CarTrackSrc &src, hipStream_t st, int budget = 0);  // budget: walk iterations per env (<= 0: to the end)
void launch_car_step(const CarSoA &s, const CarConsts &k, const float *actions, float *rew, uint8_t *done_car, int sub, int repeat,
                     hipStream_t st, bool do_broad = true);
void launch_car_broad(const CarSoA &s, const CarConsts &k, hipStream_t st, const float *fresh_body = nullptr, const uint8_t *cls = nullptr,
                      const int32_t *touching_now = nullptr, const int32_t *manifolds_now = nullptr);
void launch_car_narrow(const CarSoA &s, const CarConsts &k, hipStream_t st, bool urgent, const float *fresh_body = nullptr,
                       const uint8_t *cls = nullptr, int phase = 0);
void launch_car_solve(const CarSoA &s, const CarConsts &k, hipStream_t st);
void launch_car_sensors(const CarSoA &s, const CarConsts &k, hipStream_t st);
// near_st == nullptr: everything on st.  skip_narrow: the narrow phase of this step already ran (ahead, at the end of the previous
// step); the caller has ordered `st` and `near_st` behind it
void launch_car_coupled(const CarSoA &s, const CarConsts &k, hipStream_t st, hipStream_t near_st = nullptr, hipEvent_t ev_narrow = nullptr,
                        bool skip_narrow = false);
void launch_car_post(const CarSoA &s, const uint8_t *done_car, uint8_t *done_env, uint8_t *done_out, uint8_t *slow_env, int32_t *info_steps,
                     int32_t *info_elapsed, int max_episode_steps, bool car0_only, hipStream_t st, int32_t *class_list = nullptr, int32_t *class_count = nullptr);

// car_raster.hip: round 2's analytic raster, profiling build only (CRL_CAR_OBS_ANALYTIC=1)
#ifdef CRL_ABLATION
// only_env: draw env e iff only_env[e] == want; nullptr = every env
void launch_car_raster(const CarSoA &s, const CarConsts &k, uint8_t *obs, hipStream_t st, const uint8_t *only_env = nullptr, int want = 1);
void launch_car_raster_list(const CarSoA &s, const CarConsts &k, uint8_t *obs, hipStream_t st, const int32_t *list, const int32_t *list_count,
                            int32_t *count_to_host, int64_t expected);
void car_raster_print_ticks();  // CRL_CAR_DEBUG & 64
#else
inline void launch_car_raster(const CarSoA &, const CarConsts &, uint8_t *, hipStream_t, const uint8_t * = nullptr, int = 1) {}
inline void launch_car_raster_list(const CarSoA &, const CarConsts &, uint8_t *, hipStream_t, const int32_t *, const int32_t *, int32_t *, int64_t) {}
#endif
// car_obs.hip: the reference's observation pipeline
void launch_car_map_build(const CarSoA &s, hipStream_t st, const uint8_t *only_env = nullptr, int64_t first = 0, int64_t count = -1);  // envs [first, first + count), all or only_env[e] != 0
void launch_car_map_build_list(const CarSoA &s, hipStream_t st, const int32_t *list, const int32_t *list_count, int64_t expected);
void launch_car_view(const CarSoA &s, const CarConsts &k, hipStream_t st, const uint8_t *only_env = nullptr, int want = 1);
void launch_car_obs(const CarSoA &s, const CarConsts &k, uint8_t *obs, hipStream_t st, const uint8_t *only_env = nullptr, int want = 1);
void launch_car_obs_list(const CarSoA &s, const CarConsts &k, uint8_t *obs, hipStream_t st, const int32_t *list, const int32_t *list_count,
                         int32_t *count_to_host, int64_t expected, const uint8_t *filter = nullptr, int want_cls = 0, bool urgent = false,
                         int prepared = 0);
void launch_car_obs_long_list(const CarSoA &s, const CarConsts &k, uint8_t *obs, hipStream_t st, const int32_t *list, const int32_t *list_count, int64_t expected,
                              const uint8_t *filter, int want_cls, bool urgent, const int32_t *view_list, const int32_t *view_count, int64_t view_expected);
void car_map_light_masks(uint32_t *lightx, uint32_t *lighty);  // host: the squares' columns / rows (kMapW / 32 words each)
int car_map_coord(double v);                                     // host: (int)(obs_scale * -v + 5000), as the map polygons' vertices
void launch_car_stack(const uint8_t *frame, uint8_t *stack, uint8_t *obs, const uint8_t *fill_env, bool fill_all, int K, int64_t n,
                      int players, hipStream_t st);

}  // namespace crl
