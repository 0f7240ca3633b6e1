// crl_car_api.hip -- host side of the cCarRacingDouble context (C ABI of include/crl.h).
#include <math.h>
#include <string.h>

#include <algorithm>
#include <vector>

#include "car_device.h"
#include "crl_internal.h"

using namespace crl;

struct crl_car_ctx {
    crl_opts o;
    int64_t n;
    CarSoA s{};
    CarSoA stage{};  // only the per-car arrays a reset writes: their staged copies (car_commit_list_kernel)
    hipEvent_t ev_early3 = nullptr, ev_collide = nullptr, ev_c3 = nullptr;
    hipEvent_t ev_walk[2] = {nullptr, nullptr};  // behind the last two walk-ahead pieces: no third one is queued while both are pending
    uint64_t walk_turn = 0;
    bool collide_valid = false;  // the NEXT step's broadphase + narrow phase already ran, at the end of the last step (car_broad_kernel)
    bool collide_dirty = false;  // ... was enqueued and not consumed yet: its counter block has to be cleared if the results are not used
    hipStream_t collide_joined = nullptr;  // ... and the step's join already stood behind it: this caller's stream needs no barrier for it
    bool collide_is_joined = false;
    int32_t *coupled2 = nullptr, *lists2 = nullptr;  // [2][n] coupled flags, [2][6][n] near / touch lists: one block per step parity
    hipStream_t one = nullptr;  // HIGH priority (a queue class of its own): the wheel sensors, then the finished envs' early chain
    CarConsts K_{};
    CarTrackSrc src{};
    std::vector<void *> allocs;
    uint8_t *done_car = nullptr, *done_env = nullptr;
    float *rew_tmp = nullptr;
    int K = 1;                      // MultipleFrameStack depth (1 = no stack)
    int repeat = 1;                 // CarRacing(action_repeat=...)
    uint8_t *frame = nullptr, *stack = nullptr;  // K > 1: newest frames, and the context's own stack
    uint8_t *term = nullptr;  // [n][players][96][96] last frame of the episode an env just finished
    double *ru = nullptr;
    uint8_t *rshuffle = nullptr;
    // latency-bound part of a step (coupled solve, track generation for finished envs) runs on a
    // side stream next to the raster; slow_env = pipeline class per env (car_post_kernel)
    uint8_t *slow_env = nullptr;
    int32_t *class_list = nullptr;   // [4][n] envs of class 1 (coupled), 2 (finished, cars on their own), 3 (finished and coupled) and "4" = 2 and 3 together, of the current step, compacted
    int32_t *class_count = nullptr;  // [3] their lengths (inside `counters`)
    int32_t *counters = nullptr;     // [2][16] per step parity: coupled_count[8], class_count[2]; a step's first kernel clears the other block
    int parity = 0;
    hipEvent_t ev_post = nullptr;
    hipEvent_t ev_fin3 = nullptr;
    int32_t *class_count_host = nullptr, *class_count_hdev = nullptr;  // host-mapped copy (one step late): sizes the next step's launches
    hipStream_t side = nullptr;
    hipStream_t side2 = nullptr;  // the near-only coupled envs (plain island solves), beside the touching ones on `side`
    hipEvent_t ev_narrow = nullptr;
    hipEvent_t ev_sens = nullptr;
    hipStream_t gen = nullptr;  // walk-ahead of the next episode's track, beside the steps
    std::vector<hipStream_t> pads;  // idle streams that only occupy hardware-queue slots (crl_car_create)
    hipEvent_t ev_reset = nullptr;
    int32_t *info_steps = nullptr;  // [n] CarRacing.step_count after the step, before the auto-reset (info["num_steps"])
    int32_t *info_elapsed = nullptr;  // [n] gym TimeLimit._elapsed_steps after the step, before the auto-reset (info["TimeLimit.truncated"])
    bool car0_only = false;         // crl_opts.done_policy == CRL_CAR_DONE_CAR0
    hipEvent_t ev_fork = nullptr, ev_coupled = nullptr, ev_term = nullptr, ev_join = nullptr;
    bool overlap = true;
    bool collide_ahead = true;  // CRL_CAR_NO_COLLIDE_AHEAD=1 (read when the context is created): every step runs its own Collide (A/B, twin tests)
    bool analytic = false;  // CRL_CAR_OBS_ANALYTIC=1: rounds 1-2' analytic raster (car_raster.hip) instead of map + gather, for A/B
};

namespace crl {
// info["terminal_observation"] of the finished envs a caller lists: frames[idx[k]] -> out[k], 16 bytes per thread
__global__ __launch_bounds__(256) void car_gather_frames_kernel(const uint4 *__restrict__ frames, const int64_t *__restrict__ idx,
                                                                int64_t n, int64_t chunks, uint4 *__restrict__ out) {
    const int64_t k = blockIdx.y, i = idx[k];
    const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= chunks) return;
    out[k * chunks + c] = (i >= 0 && i < n) ? frames[i * chunks + c] : make_uint4(0, 0, 0, 0);
}
void launch_car_gather_frames(const uint8_t *frames, const int64_t *idx_dev, int64_t count, int64_t n, int64_t tile, uint8_t *out,
                              hipStream_t st) {
    const int64_t chunks = tile / 16;  // players * 96 * 96 is a multiple of 16
    for (int64_t k0 = 0; k0 < count; k0 += 65535) {
        const int64_t m = std::min<int64_t>(65535, count - k0);
        hipLaunchKernelGGL(car_gather_frames_kernel, dim3((unsigned)((chunks + 255) / 256), (unsigned)m), dim3(256), 0, st,
                           reinterpret_cast<const uint4 *>(frames), idx_dev + k0, n, chunks, reinterpret_cast<uint4 *>(out + k0 * tile));
    }
}
}  // namespace crl

// ---- body constants, the Box2D way (b2PolygonShape::ComputeMass, b2Body::ResetMassData), float32
static void poly_mass(const V2 *vs, int n, float density, float *mass, V2 *center, float *I) {
    V2 c = mk(0, 0), s = mk(0, 0);
    float area = 0, in = 0;
    for (int i = 0; i < n; i++) s = s + vs[i];
    s = (1.0f / n) * s;
    const float k_inv3 = 1.0f / 3.0f;
    for (int i = 0; i < n; i++) {
        const V2 e1 = vs[i] - s, e2 = vs[i + 1 < n ? i + 1 : 0] - s;
        const float D = cross(e1, e2), ta = 0.5f * D;
        area += ta;
        c = c + (ta * k_inv3) * (e1 + e2);
        const float intx2 = e1.x * e1.x + e2.x * e1.x + e2.x * e2.x, inty2 = e1.y * e1.y + e2.y * e1.y + e2.y * e2.y;
        in += (0.25f * k_inv3 * D) * (intx2 + inty2);
    }
    *mass = density * area;
    c = (1.0f / area) * c;
    *center = c + s;
    *I = density * in;
    *I += *mass * (dot(*center, *center) - dot(c, c));
}

static int ccw(const double (*src)[2], int n, double scale, float (*dst)[2]) {
    double area = 0;
    for (int i = 0; i < n; i++) {
        const int j = (i + 1) % n;
        area += src[i][0] * src[j][1] - src[j][0] * src[i][1];
    }
    for (int i = 0; i < n; i++) {
        const int k = area > 0 ? i : n - 1 - i;
        dst[i][0] = (float)(src[k][0] * scale), dst[i][1] = (float)(src[k][1] * scale);
    }
    return n;
}

static void make_consts(CarConsts &K) {
    static const double H1[4][2] = {{-60, 130}, {60, 130}, {60, 110}, {-60, 110}};
    static const double H2[4][2] = {{-15, 120}, {15, 120}, {20, 20}, {-20, 20}};
    static const double H3[8][2] = {{25, 20}, {50, -10}, {50, -40}, {20, -90}, {-20, -90}, {-50, -40}, {-50, -10}, {-25, 20}};
    static const double H4[4][2] = {{-50, -120}, {50, -120}, {50, -90}, {-50, -90}};
    static const double WP[4][2] = {{-CAR_WHEEL_W, +CAR_WHEEL_R}, {+CAR_WHEEL_W, +CAR_WHEEL_R}, {+CAR_WHEEL_W, -CAR_WHEEL_R}, {-CAR_WHEEL_W, -CAR_WHEEL_R}};
    static const double WPOS[4][2] = {{-55, +80}, {+55, +80}, {-55, -82}, {+55, -82}};
    const double(*polys[4])[2] = {H1, H2, H3, H4};
    const int cnt[4] = {4, 4, 8, 4};
    float mass = 0, I = 0;
    V2 lc = mk(0, 0);
    memset(&K, 0, sizeof(K));
    for (int f = 0; f < 4; f++) {
        K.hull_n[f] = ccw(polys[f], cnt[f], CAR_SIZE, K.hull_poly[f]);
        float m, i;
        V2 c;
        poly_mass(reinterpret_cast<const V2 *>(K.hull_poly[f]), cnt[f], 1.0f, &m, &c, &i);
        mass += m, lc = lc + m * c, I += i;
    }
    K.hull_inv_mass = 1.0f / mass;
    lc = K.hull_inv_mass * lc;
    K.hull_lc[0] = lc.x, K.hull_lc[1] = lc.y;
    I -= mass * dot(lc, lc);
    K.hull_inv_I = 1.0f / I;
    ccw(WP, 4, CAR_SIZE, K.wheel_poly);
    float m, i;
    V2 c;
    poly_mass(reinterpret_cast<const V2 *>(K.wheel_poly), 4, 0.1f, &m, &c, &i);
    i -= m * dot(c, c);
    K.wheel_inv_mass = 1.0f / m, K.wheel_inv_I = 1.0f / i;
    for (int w = 0; w < 4; w++) K.anchor[w][0] = (float)(WPOS[w][0] * CAR_SIZE), K.anchor[w][1] = (float)(WPOS[w][1] * CAR_SIZE);
}

template <class T>
static int calloc_dev(crl_car_ctx *c, T **p, size_t count) {
    void *q = nullptr;
    const size_t bytes = std::max<size_t>(count * sizeof(T), 16);
    if (hipMalloc(&q, bytes) != hipSuccess) return crl_fail(CRL_ENOMEM, "hipMalloc(%zu) failed", bytes);
    if (hipMemset(q, 0, bytes) != hipSuccess) return crl_fail(CRL_EHIP, "hipMemset failed");
    c->allocs.push_back(q);
    *p = (T *)q;
    return CRL_OK;
}

// coupled flags, near / touch lists and the counter block of step parity `par` (the step pipeline runs the NEXT step's broadphase
// and narrow phase at the end of a step, into the other block, while this step's frame launches still read this block's lists)
static void point_parity(const crl_car_ctx *c, CarSoA &v, int par) {
    const int64_t n = c->n;
    v.coupled = c->coupled2 + par * n;
    int32_t *l = c->lists2 + (int64_t)par * 6 * n;
    v.near_list = l, v.touch_list = l + n, v.touch_all = l + 4 * n, v.touch_multi = l + 5 * n;
    v.coupled_count = c->counters + 16 * par;
}

// The context's events only order its own streams on one device: no system-scope fence (an L2 writeback + invalidate per record --
// with 300 MB of fresh frames in flight that is tens of microseconds in front of whatever waits, a dozen times per step).
// CRL_EVENT_SYSTEM_FENCE=1 restores the default (A/B).
static const unsigned kEvFlags = hipEventDisableTiming | (getenv("CRL_EVENT_SYSTEM_FENCE") ? 0u : (unsigned)hipEventDisableSystemFence);

int crl_car_create(const crl_opts *opts, const uint32_t *text_bits_host, crl_car_ctx **out) {
    crl_car_ctx *c = new crl_car_ctx();
    c->o = *opts;
    const int players = opts->env_kind == CRL_ENV_CAR_SINGLE ? 1 : 2;
    const int64_t n = c->n = opts->num_envs, M = (int64_t)players * n;
    CarSoA &s = c->s;
    s.n = n, s.players = players;
    s.contacts_enabled = (players == 2 && !(opts->flags & CRL_FLAG_CAR_NO_CONTACTS)) ? 1 : 0;
    s.fma = (opts->flags & CRL_FLAG_CAR_FMA) ? 1 : 0;
    s.abl_no_walk = CRL_ABL(getenv("CRL_CAR_ABL_NO_WALK") != nullptr) ? 1 : 0;
    s.abl_keep_tag = CRL_ABL(getenv("CRL_CAR_ABL_KEEP_TAG") != nullptr) ? 1 : 0;
    int rc = 0;
    c->overlap = !getenv("CRL_CAR_NO_OVERLAP");
    c->collide_ahead = !getenv("CRL_CAR_NO_COLLIDE_AHEAD");
    c->analytic = CRL_ABL(getenv("CRL_CAR_OBS_ANALYTIC") && atoi(getenv("CRL_CAR_OBS_ANALYTIC")) != 0);
    {   // the pre-rastered maps, [slot][env]: 739 328 bytes per env and slot -- by far the largest part of the context.  The second
        // slot exists only for the step pipeline's staged reset (a finished env's next map is built while its terminal frame still
        // reads the current one); the one-stream mode builds in place
        const size_t slots = c->overlap && !c->analytic ? 2 : 1;
        rc = calloc_dev(c, &s.obs_map, (size_t)kMapBytes * slots * n);
        if (rc == CRL_ENOMEM)
            rc = crl_fail(CRL_ENOMEM, "car create: %zu map slot(s) of %d bytes for each of %lld envs (%.1f GB) do not fit; fewer envs per context, or CRL_CAR_NO_OVERLAP=1 (one slot)",
                          slots, (int)kMapBytes, (long long)n, (double)kMapBytes * slots * n / 1e9);
    }
#define A(f, cnt) if (!rc) rc = calloc_dev(c, &s.f, (size_t)(cnt))
    A(body, 30 * M); A(jimp, 12 * M); A(jmotor, 4 * M); A(jspeed, 4 * M); A(jlimit, 4 * M);
    A(wgas, 4 * M); A(womega, 4 * M); A(wphase, 4 * M); A(wtiles, 4 * kWheelSlots * M); A(visited, 16 * M);
    A(reward, M); A(prev_reward, M); A(step_acc, M); A(visited_count, M); A(last_block, M); A(done, M); A(step_count, M); A(first_step, M);
    A(elapsed, n); A(episode, n); A(ntiles, n); A(tile_aabb, (size_t)kCarMaxTiles * n); A(tile_blk, (size_t)(kCarMaxTiles / 8) * n); A(tile_poly, (size_t)kCarMaxTiles * 10 * n);
    A(border_poly, (size_t)kCarMaxTiles * 8 * n); A(border, (size_t)kCarMaxTiles * n); A(start_pose, 3 * n);
    A(track_scratch, (size_t)2500 * 4 * n);    // every point of a walk (car_track.hip: kWalkMax), f64: walk-ahead ...
    A(track_scratch_b, (size_t)2500 * 4 * n);  // ... and inline walks
    A(walk_tag, n); A(walk_save, (size_t)kWalkSaveWords * n); A(walk_list, n); A(walk_count, 4); A(walk_len, n); A(walk_first, n); A(walk_swap, n);
    A(wforce, 8 * M); A(wsnap, 12 * M); A(sensor_ovf, M); A(sleep, 5 * M); A(coupled_list, n); A(cap_hits, 4); A(stamps, 64); A(nc_new, n); A(contact_new, (size_t)n * kMaxContacts * kContactWords); A(n_contact, n); A(contact, (size_t)n * kMaxContacts * kContactWords);
    A(tile_aabb_em, (size_t)kCarMaxTiles * n); A(tile_poly_em, (size_t)kCarMaxTiles * 10 * n);
    A(border_poly_em, (size_t)kCarMaxTiles * 8 * n); A(border_em, (size_t)kCarMaxTiles * n);
    A(map_par, n); A(map_vtx, (size_t)kCarMaxTiles * 9 * n); A(map_yr, (size_t)kCarMaxTiles * n); A(map_overflow, n);
    A(map_lightx, kMapW / 32); A(map_lighty, kMapW / 32);
    A(view, (size_t)kViewWords * M); A(view_cnt, (size_t)16 * M); A(view_rec, (size_t)kViewRecWords * M);
#undef A
    if (!rc) rc = calloc_dev(c, &c->coupled2, 2 * n);
    if (!rc) rc = calloc_dev(c, &c->lists2, 2 * 6 * n);
    {  // staged copies of the per-car arrays a reset writes
        CarSoA &g = c->stage;
#define B(f, cnt) if (!rc) rc = calloc_dev(c, &g.f, (size_t)(cnt))
        B(body, 30 * M); B(jimp, 12 * M); B(jmotor, 4 * M); B(jspeed, 4 * M); B(jlimit, 4 * M); B(wgas, 4 * M); B(womega, 4 * M); B(wphase, 4 * M);
        B(wtiles, 4 * kWheelSlots * M); B(visited, 16 * M); B(reward, M); B(prev_reward, M); B(visited_count, M); B(last_block, M); B(done, M);
        B(step_count, M); B(first_step, M); B(sleep, 5 * M); B(elapsed, n); B(n_contact, n); B(coupled, n);
#undef B
    }
    if (!rc) rc = calloc_dev(c, &c->done_car, M);
    if (!rc) rc = calloc_dev(c, &c->done_env, n);
    if (!rc) rc = calloc_dev(c, &c->slow_env, n);
    if (!rc) rc = calloc_dev(c, &c->class_list, 4 * n);
    if (!rc) rc = calloc_dev(c, &c->counters, 2 * 16);
    if (!rc) rc = calloc_dev(c, &c->rew_tmp, M);
    if (!rc) rc = calloc_dev(c, &c->info_steps, n);
    if (!rc) rc = calloc_dev(c, &c->info_elapsed, n);
    if (!rc) rc = calloc_dev(c, &c->term, (size_t)M * 96 * 96);
    c->K = opts->frame_stack < 1 ? 1 : opts->frame_stack;
    c->repeat = opts->action_repeat >= 1 ? opts->action_repeat : 1;
    c->car0_only = opts->done_policy == CRL_CAR_DONE_CAR0;
    if (c->K > 1) {
        if (!rc) rc = calloc_dev(c, &c->frame, (size_t)M * 96 * 96);
        if (!rc) rc = calloc_dev(c, &c->stack, (size_t)M * c->K * 96 * 96);
    }
    if (rc) { crl_car_destroy(c); return rc; }
    make_consts(c->K_);
    if (text_bits_host) {
        uint32_t *tb = nullptr;
        rc = calloc_dev(c, &tb, (size_t)CRL_CAR_TEXT_STRINGS * CRL_CAR_TEXT_ROWS);
        if (rc) { crl_car_destroy(c); return rc; }
        if (hipMemcpy(tb, text_bits_host, (size_t)CRL_CAR_TEXT_STRINGS * CRL_CAR_TEXT_ROWS * 4, hipMemcpyHostToDevice) != hipSuccess) {
            crl_car_destroy(c);
            return crl_fail(CRL_EHIP, "car create: reward text upload");
        }
        c->s.text_bits = tb;
    }
    c->src.seed = opts->seed, c->src.env_id_base = opts->env_id_base;
    if (hipHostMalloc((void **)&c->class_count_host, 4 * sizeof(int32_t), hipHostMallocMapped) != hipSuccess ||
        hipHostGetDevicePointer((void **)&c->class_count_hdev, c->class_count_host, 0) != hipSuccess) {
        crl_car_destroy(c);
        return crl_fail(CRL_EHIP, "car create: host-mapped counters");
    }
    c->class_count_host[0] = c->class_count_host[1] = (int32_t)std::min<int64_t>(n, 64);
    c->s.coupled_to_host = c->class_count_hdev;
    point_parity(c, c->s, 0);
    c->class_count = c->counters + 8, c->s.zero_next = c->counters + 16;
    {
        uint32_t lx[kMapW / 32], ly[kMapW / 32];
        car_map_light_masks(lx, ly);
        if (hipMemcpy(c->s.map_lightx, lx, sizeof(lx), hipMemcpyHostToDevice) != hipSuccess ||
            hipMemcpy(c->s.map_lighty, ly, sizeof(ly), hipMemcpyHostToDevice) != hipSuccess) {
            crl_car_destroy(c);
            return crl_fail(CRL_EHIP, "car create: map constants");
        }
    }
    int prio_lo = 0, prio_hi = 0;
    hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);  // (numerically lower = higher priority)
    {
        // WHICH stream shares a hardware pipe with which is decided by the order in which their hardware queues are created: the
        // runtime gives every priority class up to four queues and the driver deals new queues round the chip's four pipes (one
        // micro-engine each, which serves its queues' barrier and dispatch packets one after the other).  Measured (round 4,
        // 16 384 envs, steady state, tools/car_quick.py; docs/LAB_NOTES_r04.md): with the walk-ahead's queue (one 15 ms kernel
        // after the other) or the high-priority queue on the pipe of the CALLER's stream -- which carries the step's longest chain --
        // a step takes 1.03-1.08 ms instead of 0.92; the walk-ahead beside `side2` (join, next step's Collide) 1.04; beside `side`
        // (the bulk, which has slack) 0.92.  In a process whose first streams are these (the legacy default stream's queue came
        // first: pipe 0) the order below puts side, side2 and `one` on pipes 1-3, an idle low-priority queue on the caller's pipe
        // and the walk-ahead beside `side`.  CRL_CAR_STREAM_ORDER (letters s 2 o g, d / D / H = an idle normal / low / high
        // priority stream) re-deals them for a process that created streams before this context.
        const char *order = getenv("CRL_CAR_STREAM_ORDER") ? getenv("CRL_CAR_STREAM_ORDER") : "s2oDg";
        bool ok = true;
        for (const char *q = order; *q && ok; q++) {
            hipStream_t pad = nullptr;
            if (*q == 's' && !c->side) ok = hipStreamCreateWithPriority(&c->side, hipStreamNonBlocking, 0) == hipSuccess;  // the bulk of a step
            else if (*q == 'S' && !c->side) ok = hipStreamCreateWithPriority(&c->side, hipStreamNonBlocking, prio_lo) == hipSuccess;  // (experiment: the bulk at low dispatch priority)
            else if (*q == '2' && !c->side2) ok = hipStreamCreateWithFlags(&c->side2, hipStreamNonBlocking) == hipSuccess;
            else if (*q == 'o' && !c->one) ok = hipStreamCreateWithPriority(&c->one, hipStreamNonBlocking, prio_hi) == hipSuccess;
            else if (*q == 'g' && !c->gen) ok = hipStreamCreateWithPriority(&c->gen, hipStreamNonBlocking, prio_lo) == hipSuccess;  // milliseconds-long walks: a priority class of its own
            else if (*q == 'd') ok = hipStreamCreateWithFlags(&pad, hipStreamNonBlocking) == hipSuccess;
            else if (*q == 'D') ok = hipStreamCreateWithPriority(&pad, hipStreamNonBlocking, prio_lo) == hipSuccess;
            else if (*q == 'H') ok = hipStreamCreateWithPriority(&pad, hipStreamNonBlocking, prio_hi) == hipSuccess;
            if (pad) c->pads.push_back(pad);
        }
        if (ok && !c->side) ok = hipStreamCreateWithPriority(&c->side, hipStreamNonBlocking, 0) == hipSuccess;  // (a letter the order string left out)
        if (ok && !c->side2) ok = hipStreamCreateWithFlags(&c->side2, hipStreamNonBlocking) == hipSuccess;
        if (ok && !c->one) ok = hipStreamCreateWithPriority(&c->one, hipStreamNonBlocking, prio_hi) == hipSuccess;
        if (ok && !c->gen) ok = hipStreamCreateWithPriority(&c->gen, hipStreamNonBlocking, prio_lo) == hipSuccess;
        if (!ok) {
            crl_car_destroy(c);
            return crl_fail(CRL_EHIP, "car create: streams");
        }
    }
    if (hipEventCreateWithFlags(&c->ev_narrow, kEvFlags) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_post, kEvFlags) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_early3, kEvFlags) != hipSuccess || hipEventCreateWithFlags(&c->ev_collide, kEvFlags) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_c3, kEvFlags) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_fin3, kEvFlags) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_sens, kEvFlags) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_reset, kEvFlags) != hipSuccess || hipEventCreateWithFlags(&c->ev_walk[0], kEvFlags) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_walk[1], kEvFlags) != hipSuccess ||
        hipMemset(c->s.walk_tag, 0xFF, (size_t)c->n * sizeof(uint32_t)) != hipSuccess ||
        hipMemset(c->s.walk_save, 0xFF, (size_t)c->n * kWalkSaveWords * sizeof(uint32_t)) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_fork, kEvFlags) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_coupled, kEvFlags) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_term, kEvFlags) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_join, kEvFlags) != hipSuccess) {
        crl_car_destroy(c);
        return crl_fail(CRL_EHIP, "car create: side stream");
    }
    {  // device copies of the context struct and the constants (everything in them is final here)
        CarSoA *sd = nullptr;
        CarConsts *kd = nullptr;
        if (calloc_dev(c, &sd, 1) || calloc_dev(c, &kd, 1)) {
            crl_car_destroy(c);
            return crl_fail(CRL_ENOMEM, "car create: context copy");
        }
        c->s.self_dev = sd, c->s.consts_dev = kd;
        if (hipMemcpy(sd, &c->s, sizeof(CarSoA), hipMemcpyHostToDevice) != hipSuccess ||
            hipMemcpy(kd, &c->K_, sizeof(CarConsts), hipMemcpyHostToDevice) != hipSuccess) {
            crl_car_destroy(c);
            return crl_fail(CRL_EHIP, "car create: context copy");
        }
    }
    *out = c;
    return CRL_OK;
}

void crl_car_destroy(crl_car_ctx *c) {
#ifdef CRL_ABLATION
    if (c && c->s.stamps && getenv("CRL_CAR_STAMPS")) {  // profiling build: mean cycles per wavefront and phase of car_touch_kernel / car_narrow_kernel
        unsigned long long h[64];
        hipDeviceSynchronize();
        hipMemcpy(h, c->s.stamps, sizeof(h), hipMemcpyDeviceToHost);
        for (int cls = 0; cls < 3; cls++) {
            const unsigned long long *q = h + 8 * cls;
            const double w = q[7] ? (double)q[7] : 1.0;
            fprintf(stderr, "touch class %d: %llu waves; mean cycles setup %.0f | velocity %.0f | position %.0f (%.1f iterations) | store %.0f (sleep scan %.0f) | max total %llu\n", cls + 1,
                    q[7], q[0] / w, q[1] / w, q[2] / w, q[4] / w, q[3] / w, q[6] / w, q[5]);
            fprintf(stderr, "              longest: setup %llu | velocity %llu | position %llu; wavefronts with all 60 position iterations %llu\n", h[40 + 4 * cls + 3],
                    h[40 + 4 * cls + 0], h[40 + 4 * cls + 1], h[40 + 4 * cls + 2]);
        }
        const unsigned long long *q = h + 32;
        const double w = q[7] ? (double)q[7] : 1.0;
        fprintf(stderr, "narrow: %llu wave-slots; mean cycles prologue %.0f | load+sincos+circle %.0f | collide %.0f | compaction+store %.0f\n", q[7], q[0] / w, q[1] / w, q[2] / w, q[3] / w);
    }
#endif
    if (!c) return;
    hipDeviceSynchronize();
#ifdef CRL_ABLATION
    if (getenv("CRL_CAR_DEBUG") && (atoi(getenv("CRL_CAR_DEBUG")) & 64)) crl::car_raster_print_ticks();
#endif
    if (c->side) hipStreamDestroy(c->side);
    if (c->side2) hipStreamDestroy(c->side2);
    if (c->ev_narrow) hipEventDestroy(c->ev_narrow);
    if (c->ev_post) hipEventDestroy(c->ev_post);
    if (c->ev_early3) hipEventDestroy(c->ev_early3);
    if (c->ev_c3) hipEventDestroy(c->ev_c3);
    for (hipEvent_t e : c->ev_walk)
        if (e) hipEventDestroy(e);
    if (c->ev_collide) hipEventDestroy(c->ev_collide);
    if (c->one) hipStreamDestroy(c->one);
    if (c->ev_fin3) hipEventDestroy(c->ev_fin3);
    if (c->ev_sens) hipEventDestroy(c->ev_sens);
    if (c->gen) hipStreamDestroy(c->gen);
    for (hipStream_t p : c->pads) hipStreamDestroy(p);
    if (c->ev_reset) hipEventDestroy(c->ev_reset);
    if (c->ev_fork) hipEventDestroy(c->ev_fork);
    if (c->ev_coupled) hipEventDestroy(c->ev_coupled);
    if (c->ev_term) hipEventDestroy(c->ev_term);
    if (c->ev_join) hipEventDestroy(c->ev_join);
    if (c->class_count_host) hipHostFree(c->class_count_host);
    for (void *p : c->allocs) hipFree(p);
    if (c->ru) hipFree(c->ru);
    if (c->rshuffle) hipFree(c->rshuffle);
    delete c;
}

// The stored walks depend on the seed / the replay stream: changing either invalidates them.
static void invalidate_walks(crl_car_ctx *c) {
    hipDeviceSynchronize();
    hipMemset(c->s.walk_tag, 0xFF, (size_t)c->n * sizeof(uint32_t));
    hipMemset(c->s.walk_save, 0xFF, (size_t)c->n * kWalkSaveWords * sizeof(uint32_t));  // (episode 0xFFFFFFFF: no unfinished walk)
}
// Queues a BOUNDED piece of the walk-ahead (every env whose stored walk is not the one its next reset needs advances by
// kWalkBudget iterations of its walk) on the context's own low-priority stream.  At most one piece per step and two in flight:
// ~0.3 ms of a few wavefronts (0.7 beside a step), against one lane's 5-17 ms when a launch walked to the end -- whatever
// synchronises the device (the end of a timed window, reset, set_state) waits for a millisecond, not for a walk (a 20-step window
// used to end with up to 17 ms of it: 0.85 ms per step).  The pieces are NOT ordered behind the step that asked for them: the
// kernels find their work through the episode / tag words (acquire / release), so a host that runs hundreds of steps ahead of
// the GPU does not starve the walks.  A walk is needed ~1 000 steps after it is asked for and takes 20-80 pieces; a reset that
// comes too early for its walk simply walks inline (results never depend on it).
static constexpr int kWalkBudget = 160;
// (behind_reset: after a FULL reset the piece is ordered behind it -- that reset walks every env itself, on the caller's stream and
// into the same scratch, and a piece that started another attempt of the same walk beside it would overwrite its points)
static void queue_walk_ahead(crl_car_ctx *c, hipStream_t after, bool behind_reset = false) {
    if (!c->overlap) return;
    if (c->s.abl_no_walk && !behind_reset) return;  // (profiling build: the step without a walk-ahead piece in flight)
    static const int budget = CRL_ABL(getenv("CRL_CAR_WALK_BUDGET") != nullptr) ? atoi(getenv("CRL_CAR_WALK_BUDGET")) : kWalkBudget;
    hipEvent_t &ev = c->ev_walk[c->walk_turn & 1];
#ifdef CRL_TEST_WALK_UNORDERED  // (a test build reproduces the race tests/test_hip_round2.py guards against)
    behind_reset = false;
#endif
    if (behind_reset) {
        hipEventRecord(c->ev_reset, after);
        hipStreamWaitEvent(c->gen, c->ev_reset, 0);
    } else if (c->walk_turn >= 2 && hipEventQuery(ev) == hipErrorNotReady) {
        return;  // two pieces are still queued
    }
    launch_car_walk_ahead(c->s, c->src, c->gen, budget);
    hipEventRecord(ev, c->gen);
    c->walk_turn++;
}
void crl_car_seed(crl_car_ctx *c, uint64_t seed) {
    invalidate_walks(c);
    c->src.seed = seed;
}
int64_t crl_car_obs_bytes(const crl_car_ctx *c) { return (int64_t)c->s.players * c->K * CRL_CAR_OBS * CRL_CAR_OBS; }

// The state as the staged reset of the step pipeline sees it: per-car arrays -> their staged copies, the env's OTHER map slot
static CarSoA stage_view(const crl_car_ctx *c) {
    CarSoA v = c->s;
    const CarSoA &g = c->stage;
    v.body = g.body, v.jimp = g.jimp, v.jmotor = g.jmotor, v.jspeed = g.jspeed, v.jlimit = g.jlimit, v.wgas = g.wgas, v.womega = g.womega, v.wphase = g.wphase;
    v.wtiles = g.wtiles, v.visited = g.visited, v.reward = g.reward, v.prev_reward = g.prev_reward, v.visited_count = g.visited_count;
    v.last_block = g.last_block, v.done = g.done, v.step_count = g.step_count, v.first_step = g.first_step, v.sleep = g.sleep;
    v.elapsed = g.elapsed;
    if (v.n_contact) v.n_contact = g.n_contact, v.coupled = g.coupled;
    v.map_alt = 1;
    return v;
}

// frames of every env, or of the envs with only_env[e] == want
// tm (optional): timer 1 brackets the frame kernel alone (car_obs_kernel / car_raster_kernel), for bench.py's roofline
static void frames(crl_car_ctx *c, uint8_t *dst, hipStream_t st, const uint8_t *only_env = nullptr, int want = 1, crl_timer *tm = nullptr) {
    if (c->analytic) {
        crl_timer_begin(tm, 1, st);
        launch_car_raster(c->s, c->K_, dst, st, only_env, want);
        crl_timer_end(tm, 1, st);
        return;
    }
    launch_car_view(c->s, c->K_, st, only_env, want);
    crl_timer_begin(tm, 1, st);
    launch_car_obs(c->s, c->K_, dst, st, only_env, want);
    crl_timer_end(tm, 1, st);
}
// frames of the envs of a compacted list
static void frames_list(crl_car_ctx *c, const CarSoA &s, uint8_t *dst, hipStream_t st, const int32_t *list, const int32_t *list_count,
                        int32_t *count_to_host, int64_t expected) {
    if (c->analytic) launch_car_raster_list(s, c->K_, dst, st, list, list_count, count_to_host, expected);
    else launch_car_obs_list(s, c->K_, dst, st, list, list_count, count_to_host, expected);
}

// newest frames -> obs_dev, through the frame stack when K > 1
static void draw(crl_car_ctx *c, uint8_t *obs_dev, bool fill_all, hipStream_t st) {
    if (c->K == 1) {
        frames(c, obs_dev, st);
        return;
    }
    frames(c, c->frame, st);
    launch_car_stack(c->frame, c->stack, obs_dev, c->done_env, fill_all, c->K, c->n, c->s.players, st);
}

// Before the state is changed from outside: the collide-ahead of the last step may still be reading it, and its results are void
static void void_collide_ahead(crl_car_ctx *c, hipStream_t st) {
    if (c->collide_dirty) hipStreamWaitEvent(st, c->ev_collide, 0);
    c->collide_valid = false;
}

int crl_car_reset(crl_car_ctx *c, uint8_t *obs_dev, hipStream_t st) {
    void_collide_ahead(c, st);
    // Full reset: every env needs a walk now.  The walk kernel does them one per LANE (64 envs per wavefront);
    // the reset kernel, one wavefront per env, then only builds the tiles.  (Any walk-ahead still running on the
    // context's own stream writes the same scratch: wait for it first; a full reset is rare.)
    if (c->overlap) {
        hipStreamSynchronize(c->gen);
        launch_car_walk_ahead(c->s, c->src, st);
    }
    launch_car_reset(c->s, c->K_, c->src, false, nullptr, st);
    launch_car_map_build(c->s, st);  // render_road_for_observation_map (crmp:519)
    queue_walk_ahead(c, st, true);
    if (obs_dev) draw(c, obs_dev, true, st);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return crl_fail(CRL_EHIP, "car reset: %s", hipGetErrorString(e));
    return CRL_OK;
}

const uint8_t *crl_car_terminal_frames(const crl_car_ctx *c) { return c->term; }
const uint8_t *crl_car_done_flags(const crl_car_ctx *c) { return c->done_car; }
const int32_t *crl_car_info_steps(const crl_car_ctx *c) { return c->info_steps; }
const int32_t *crl_car_info_elapsed(const crl_car_ctx *c) { return c->info_elapsed; }
int crl_car_players(const crl_car_ctx *c) { return c->s.players; }

int crl_car_render(crl_car_ctx *c, uint8_t *obs_dev, hipStream_t st) {
    if (c->K > 1) return crl_fail(CRL_ESTATE, "crl_render on a stacked CarRacing context would advance the stack");
    frames(c, obs_dev, st);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return crl_fail(CRL_EHIP, "car render: %s", hipGetErrorString(e));
    return CRL_OK;
}

// VecEnv.step over make_car_racing_double envs (car_racing/register.py:43-53): CarRacing.step,
// TimeLimit, FlattenMultiAgentObservation's any-done, then auto-reset of finished envs.
int crl_car_step(crl_car_ctx *c, const float *actions_dev, uint8_t *obs_dev, float *rew_dev, uint8_t *done_dev, hipStream_t st,
                 crl_timer *tm) {
    // Rewards and done flags are final after car_step_kernel (the reference evaluates them before
    // world.Step, crmp:576-603), so the env-level bookkeeping does not wait for the coupled solve.
    const bool fork = c->overlap && obs_dev != nullptr;
    const bool contacts = c->s.players == 2 && c->s.contacts_enabled;
    // The step's longest chain (touching solve -> those envs' frames) runs on the CALLER's stream, directly behind
    // car_step_kernel and directly in front of the next step's: every cross-stream hop costs tens of microseconds of command-processor
    // latency.  The bulk (per-car solve -> camera -> frames of the envs on their own) forks to `side` and has the slack for its hops.
    const hipStream_t crit = st, bulk = c->side;
    crl_timer_begin(tm, 0, st);
    auto next_counters = [&]() {  // this sub-step's counter block; car_step_kernel clears the other one for the next
        c->parity ^= 1;
        point_parity(c, c->s, c->parity);
        c->class_count = c->s.coupled_count + 8;
        c->s.zero_next = c->counters + 16 * (c->parity ^ 1);
    };
    // Collide ahead: when the last step ended with the broadphase + narrow phase of THIS one (same poses: Car.step moves nothing),
    // the step's longest chain starts with the touching solve.  Not with action repeats (every sub-step collides), not after the
    // state was changed from outside (reset, set_state), not in the one-stream mode.  CRL_CAR_NO_COLLIDE_AHEAD=1: A/B, twin tests
    const bool can_ahead = c->collide_ahead && fork && c->repeat == 1 && contacts;
    const bool ahead = can_ahead && c->collide_valid;
    const bool collide_joined = c->collide_is_joined && c->collide_joined == st;  // (the last step's join on this stream stood behind the Collide)
    c->collide_valid = c->collide_is_joined = false;
    if (!ahead && c->collide_dirty) {  // results of a collide-ahead that will not be used (the state was changed in between): its counters go
        hipStreamWaitEvent(st, c->ev_collide, 0);
        hipMemsetAsync(c->counters + 16 * (c->parity ^ 1), 0, 8 * sizeof(int32_t), st);  // (the block this step is about to use)
    }
    c->collide_dirty = false;
    for (int sub = 0; sub < c->repeat; sub++) {  // action repetition: Car.step + world.Step per repeat (crmp:576-603)
        next_counters();
        // Car.step, rewards, done flags; decides which cars are solved together
        launch_car_step(c->s, c->K_, actions_dev, rew_dev ? rew_dev : c->rew_tmp, c->done_car, sub, c->repeat, st, !ahead);
        if (fork && sub == c->repeat - 1) break;  // the last world.Step is forked below
        launch_car_sensors(c->s, c->K_, st);  // world.Step: Collide (wheel sensors), then Solve
        launch_car_solve(c->s, c->K_, st);
        launch_car_coupled(c->s, c->K_, st);
    }
    if (!fork) {
        launch_car_post(c->s, c->done_car, c->done_env, done_dev, c->slow_env, c->info_steps, c->info_elapsed, 1000, c->car0_only, st, c->class_list, c->class_count);
        // info["terminal_observation"] (dummy_vec_env.py:55-57): draw finished envs before they are reset
        if (obs_dev) frames(c, c->term, st, c->done_env);
        launch_car_reset(c->s, c->K_, c->src, true, c->done_env, st);
        launch_car_map_build(c->s, st, c->done_env);
        queue_walk_ahead(c, st);
        crl_timer_end(tm, 0, st);
        if (obs_dev) {
            crl_timer_begin(tm, 1, st);
            draw(c, obs_dev, false, st);
            crl_timer_end(tm, 1, st);
        }
    } else {
        // Streams over disjoint classes of envs (car_post_kernel's slow_env: 0 on its own, 1 coupled, 2 finished, 3 finished and coupled):
        //   crit   (the caller's) [narrow phase of the coupled envs ->] the touching ones' island solve -> their frames (the step's longest chain)
        //   bulk   per-car solve -> camera, polygons -> frames of class 0 (the big launch)
        //   side2  env bookkeeping; the coupled envs where nothing touches (two islands of their own) -> their frames; the terminal
        //          frames of the finished envs on their own as soon as the per-car solve is in, their commit once the new episode is
        //          staged, then the NEXT step's broadphase; behind the touching solve its narrow phase; the step's join
        //   one    (HIGH priority) the wheel sensors (tile rewards, road_visited: they read the transforms the step started from and
        //          feed nothing into its solve; every frame shows the reward, so every frame launch waits for them), then the finished
        //          envs' NEW episode, prepared early on the staged view: reset, map, first frame; behind the touching solve the
        //          terminal frames + commit of the few envs that are finished AND coupled.  Round 4: the sensors used to have a
        //          normal-priority stream of their own and finished at ~350 us; here they are done at ~135-250 us, everything that
        //          waits for them starts earlier, and the context needs one stream less (1.03 -> 0.93 ms per step, same box)
        // (streams of one priority share four hardware queues, and two streams that share one wait for each other's kernels;
        // the walk-ahead's pieces have a priority class of their own)
        uint8_t *target = c->K == 1 ? obs_dev : c->frame;
        hipEventRecord(c->ev_fork, st);
        hipStreamWaitEvent(c->side, c->ev_fork, 0);
        hipStreamWaitEvent(c->side2, c->ev_fork, 0);
        hipStreamWaitEvent(c->one, c->ev_fork, 0);
        launch_car_post(c->s, c->done_car, c->done_env, done_dev, c->slow_env, c->info_steps, c->info_elapsed, 1000, c->car0_only, c->side2, c->class_list,
                        c->class_count);
        hipEventRecord(c->ev_post, c->side2);  // classes, class lists, done flags: what the frame launches and the finished-env chains filter by
        if (ahead && !collide_joined) {  // (side2 ran the narrow phase itself, at the end of the last step)
            hipStreamWaitEvent(crit, c->ev_collide, 0);
            hipStreamWaitEvent(bulk, c->ev_collide, 0);
        }  // (else: two barriers less, one of them on the step's longest chain)
        if (contacts && !ahead) {  // narrow phase at the head of crit; the sensor kernel beside it doubles both: sensors behind it
            launch_car_narrow(c->s, c->K_, crit, true);
            hipEventRecord(c->ev_narrow, crit);
            hipStreamWaitEvent(c->side2, c->ev_narrow, 0);
            hipStreamWaitEvent(c->one, c->ev_narrow, 0);
        }
        hipStreamWaitEvent(c->one, c->ev_post, 0);  // (a 6 us kernel, long done: ev_sens then stands for the bookkeeping as well, and the frame launches pass ONE barrier each)
        launch_car_sensors(c->s, c->K_, c->one);
        hipEventRecord(c->ev_sens, c->one);
        // CRL_CAR_TOUCH_VIEW (profiling build; measured, not kept, docs/LAB_NOTES_r05.md): 1 = every wavefront of the touching solve prepares its envs'
        // views (+ 2-3 %), 2 = the one-manifold wavefronts do and only the multi-manifold envs' views are computed behind the solve (+- 0)
        static const int touch_view_mode = CRL_ABL(getenv("CRL_CAR_TOUCH_VIEW") != nullptr) ? atoi(getenv("CRL_CAR_TOUCH_VIEW")) : 0;
        const int touch_view = contacts && !c->analytic ? touch_view_mode : 0;
        {
            CarSoA sv2 = c->s;
            sv2.touch_view = touch_view;
            crl_timer_begin(tm, 2, crit);  // (crl_kernel_time_stats slot 2: on this stream nothing but the touching solve lies between the two records)
            launch_car_coupled(sv2, c->K_, crit, c->side2, nullptr, true);  // near-only solve on side2, touching solve on crit
            crl_timer_end(tm, 2, crit);
        }
        hipEventRecord(c->ev_coupled, crit);
        const int64_t exp_coupled = c->class_count_host[0], exp_done = c->class_count_host[1];
        hipStreamWaitEvent(c->side2, c->ev_sens, 0);  // (every frame shows the reward)
        if (contacts) {  // frames of the near-only envs
            if (c->analytic) launch_car_raster_list(c->s, c->K_, target, c->side2, c->s.near_list, c->s.coupled_count + 1, nullptr, exp_coupled);
            else {  // (round 5: as a view launch + a gather-in-thirds launch, like the touching envs' frames: - 0.5 %; CRL_CAR_NEAR_LIST=1 in the profiling build: the list kernel)
                static const bool near_list_kernel = CRL_ABL(getenv("CRL_CAR_NEAR_LIST") != nullptr);
                if (!near_list_kernel) launch_car_obs_long_list(c->s, c->K_, target, c->side2, c->s.near_list, c->s.coupled_count + 1, exp_coupled, c->slow_env, 1, false,
                                                                c->s.near_list, c->s.coupled_count + 1, exp_coupled);
                else launch_car_obs_list(c->s, c->K_, target, c->side2, c->s.near_list, c->s.coupled_count + 1, nullptr, exp_coupled, c->slow_env, 1);
            }
        }
        // bulk: the per-car solve, then the frames of every env that is neither coupled nor finished
        launch_car_solve(c->s, c->K_, bulk);
        hipEventRecord(c->ev_term, bulk);  // (bodies of the non-coupled cars are final)
        hipStreamWaitEvent(bulk, c->ev_sens, 0);
        crl_timer_end(tm, 0, bulk);
        frames(c, target, bulk, c->slow_env, 0, tm);
        // The touching envs' frames, behind their solve, on the caller's stream (wave priority 3: beside the big launch's 32 768 wavefronts).
        // (Round 5, profiling build, CRL_CAR_TOUCH_FRAMES_ON_ONE=1: on the HIGH-priority stream instead, in front of the class-3 chain that
        // waits for the same solve there, so that dispatch priority gets their 2 300 workgroups the CU slots the bulk frame kernel frees --
        // bit-exact and 5-6 % SLOWER in three A/B pairs (0.885-0.895 against 0.826-0.845 ms; fma 0.823-0.836 against 0.784-0.796): once
        // more, what ends a step has to sit on the caller's stream.)
        static const bool touch_frames_on_one = CRL_ABL(getenv("CRL_CAR_TOUCH_FRAMES_ON_ONE") != nullptr);
        auto touch_frames = [&](hipStream_t q) {
            if (c->analytic) launch_car_raster_list(c->s, c->K_, target, q, c->s.touch_all, c->s.coupled_count + 5, nullptr, exp_coupled);
            else if (!CRL_ABL(getenv("CRL_CAR_ABL_NO_TOUCH_FRAMES") != nullptr))  // (timing ablation, WRONG frames: what the step costs without them)
            {
                static const bool one_launch = CRL_ABL(getenv("CRL_CAR_TOUCH_FRAMES_LIST") != nullptr);  // (profiling build: rounds 3-4's one-wavefront-per-tile list kernel)
                if (touch_view == 1 || one_launch) launch_car_obs_list(c->s, c->K_, target, q, c->s.touch_all, c->s.coupled_count + 5, nullptr, exp_coupled, c->slow_env, 1, true, touch_view == 1 ? 1 : 0);
                else if (touch_view == 2)  // views: only the envs with two manifolds or more are left (a tenth of the touching envs); frames: all, in thirds
                    launch_car_obs_long_list(c->s, c->K_, target, q, c->s.touch_all, c->s.coupled_count + 5, exp_coupled, c->slow_env, 1, true, c->s.touch_multi,
                                             c->s.coupled_count + 6, exp_coupled / 4);
                else launch_car_obs_long_list(c->s, c->K_, target, q, c->s.touch_all, c->s.coupled_count + 5, exp_coupled, c->slow_env, 1, true, c->s.touch_all,
                                              c->s.coupled_count + 5, exp_coupled);
            }
        };
        if (contacts && !touch_frames_on_one) {  // crit again
            hipStreamWaitEvent(crit, c->ev_sens, 0);
            touch_frames(crit);
        }
        // The finished envs.  Their NEW episode (track arrays in place, map into the env's other slot, car state into the staged
        // arrays, first frame straight into the caller's tensor) only needs the step's sensor contacts to be in: it is prepared
        // beside the solves, on the sensors' stream.  What has to wait for a solve is small: the terminal frame
        // (info["terminal_observation"], drawn from the solved bodies over the OLD map) and -- behind the new episode's staging --
        // the commit that makes it current.  Class 2 (cars on their own: nearly all of them) waits for the per-car solve; class 3
        // (finished AND coupled) for the touching solve: two small kernels behind the step's longest chain.
        const CarSoA sv = stage_view(c);
        auto list_of = [&](int cls) { return c->class_list + (int64_t)(cls - 1) * c->n; };
        auto count_of = [&](int cls) { return c->class_count + (cls - 1); };
        auto terminal_frames = [&](hipStream_t q, int cls, int64_t expected, int32_t *count_to_host) {
            if (c->analytic) launch_car_raster_list(c->s, c->K_, c->term, q, list_of(cls), count_of(cls), count_to_host, expected);
            else launch_car_obs_list(c->s, c->K_, c->term, q, list_of(cls), count_of(cls), count_to_host, expected, c->slow_env, cls,
                                     cls == 3 && !CRL_ABL(getenv("CRL_CAR_C3_CALM") != nullptr));  // (class 3: behind the touching solve, beside 2 300 other tiles)
        };
        auto finish_chain = [&](hipStream_t q, int cls, int64_t expected, int32_t *count_to_host) {  // everything in place, in order
            frames_list(c, c->s, c->term, q, list_of(cls), count_of(cls), count_to_host, expected);
            launch_car_reset_list(c->s, c->K_, c->src, list_of(cls), count_of(cls), expected, q);
            launch_car_map_build_list(c->s, q, list_of(cls), count_of(cls), expected);
            frames_list(c, c->s, target, q, list_of(cls), count_of(cls), nullptr, expected);
        };
        bool split_narrow = false;
        const bool staged = !c->analytic;  // (the analytic raster reads the track arrays themselves: it needs them until the terminal frame is drawn)
        const bool early_broad = can_ahead && staged && !CRL_ABL(getenv("CRL_CAR_COLLIDE_LATE") != nullptr) && !CRL_ABL(getenv("CRL_CAR_BROAD_LATE") != nullptr);
        hipStreamWaitEvent(c->side2, c->ev_term, 0);
        if (staged) {
            // early chain, every finished env (class 2 and 3 alike), at high priority: the map build's small workgroups get CU slots
            // ahead of the frame kernel's 32 768 wavefronts instead of behind them (320 us for a dozen maps otherwise)
            launch_car_reset_list(sv, c->K_, c->src, list_of(4), count_of(4), exp_done + 8, c->one);
            launch_car_map_build_list(sv, c->one, list_of(4), count_of(4), exp_done + 8);
            frames_list(c, sv, target, c->one, list_of(4), count_of(4), nullptr, exp_done + 8);
            hipEventRecord(c->ev_early3, c->one);
            terminal_frames(c->side2, 2, exp_done, c->class_count_hdev + 1);  // (side2 idles between the near-only envs' frames and the touching solve)
            hipStreamWaitEvent(c->side2, c->ev_early3, 0);
            launch_car_commit_list(c->s, sv, list_of(2), count_of(2), exp_done, c->side2);
            if (early_broad) {
                // The NEXT step's broadphase already here: every pose but the touching islands' is final (per-car solve: ev_term; near-only
                // solve and class-2 commit: this stream; class 3: staged), and an env that touches now is filed as coupled untested --
                // behind the touching solve only the narrow phase is left.
                CarSoA nx = c->s;
                point_parity(c, nx, c->parity ^ 1);
                launch_car_broad(nx, c->K_, c->side2, c->stage.body, c->slow_env, c->s.coupled, c->s.nc_new);
                // ... and the narrow phase of the envs whose poses ARE final (round 5: everything but the envs that touch in this step --
                // half of the 2 300 coupled envs), beside the touching solve instead of behind it (CRL_CAR_NARROW_LATE=1, profiling build: all behind)
                static const bool narrow_late = CRL_ABL(getenv("CRL_CAR_NARROW_LATE") != nullptr);
                split_narrow = !narrow_late;
                if (split_narrow) launch_car_narrow(nx, c->K_, c->side2, false, c->stage.body, c->slow_env, 1);
            }
        } else {
            finish_chain(c->side2, 2, exp_done, c->class_count_hdev + 1);
        }
        // Behind the touching solve: (a) the envs that finished while coupled (class 3, a handful): terminal frames + commit; (b) the
        // NEXT step's Collide (every solve of this step is in); (c) the touching envs' frames on the caller's stream (above).
        // early_collide: (a) on the high-priority stream and (b) on side2 at once -- the Collide reads the class-3 envs' new episode
        // from the staging arrays, which (a)'s commit is copying from, and takes no manifolds over for them -- instead of (a), the
        // join, then (b) in a row in front of the next step's touching solve.
        static const bool late_collide = CRL_ABL(getenv("CRL_CAR_COLLIDE_LATE") != nullptr);
        const bool early_collide = can_ahead && staged && !late_collide;
        auto collide_next = [&](bool fresh) {
            CarSoA nx = c->s;
            point_parity(c, nx, c->parity ^ 1);
            if (!(fresh && early_broad)) launch_car_broad(nx, c->K_, c->side2, fresh ? c->stage.body : nullptr, fresh ? c->slow_env : nullptr);
            launch_car_narrow(nx, c->K_, c->side2, false, fresh ? c->stage.body : nullptr, fresh ? c->slow_env : nullptr, fresh && early_broad && split_narrow ? 2 : 0);
            hipEventRecord(c->ev_collide, c->side2);
            c->collide_valid = c->collide_dirty = true;
        };
        hipStreamWaitEvent(c->side2, c->ev_coupled, 0);  // the touching solve
        if (contacts && touch_frames_on_one && !early_collide) {  // (no class-3 chain on `one` in this mode: the frames go there all the same)
            hipStreamWaitEvent(c->one, c->ev_coupled, 0);
            touch_frames(c->one);
            hipEventRecord(c->ev_c3, c->one);
            hipStreamWaitEvent(c->side2, c->ev_c3, 0);
        }
        if (early_collide) {
            hipStreamWaitEvent(c->one, c->ev_coupled, 0);
            if (contacts && touch_frames_on_one) touch_frames(c->one);  // (`one` ran the sensors itself: no wait for ev_sens)
            if (!CRL_ABL(getenv("CRL_CAR_ABL_NO_C3") != nullptr)) terminal_frames(c->one, 3, 8, nullptr);  // (timing ablation: no terminal frames for class 3)
            launch_car_commit_list(c->s, sv, list_of(3), count_of(3), 8, c->one);
            hipEventRecord(c->ev_c3, c->one);
            collide_next(true);
            hipStreamWaitEvent(c->side2, c->ev_c3, 0);
        } else if (staged) {
            terminal_frames(c->side2, 3, 8, nullptr);
            launch_car_commit_list(c->s, sv, list_of(3), count_of(3), 8, c->side2);
        } else {
            finish_chain(c->side2, 3, 8, nullptr);
        }
        // join: side2 collects the bulk stream (and, through ev_early3 / ev_c3, the high-priority one) behind its own last kernel, so
        // that the caller's stream -- whose last kernel is usually the last of the step -- passes ONE barrier instead of three
        hipEventRecord(c->ev_join, c->side);
        hipStreamWaitEvent(c->side2, c->ev_join, 0);
        hipEventRecord(c->ev_fin3, c->side2);
        if (can_ahead && !early_collide) collide_next(false);  // (beside this step's last frames and the next car_step_kernel)
        queue_walk_ahead(c, c->side2);
        // The next step may skip its barriers on ev_collide ONLY because of this order on side2: collide_next(true) [records ev_collide] ...
        // record(ev_fin3) ... and the caller's stream waits for ev_fin3 here.  Anything added to side2 behind the ev_fin3 record that the next
        // car_step_kernel must see, or an early return between that record and this wait, breaks it silently: the flag is set
        // behind the wait, and only when the wait was accepted.
        const bool joined = hipStreamWaitEvent(st, c->ev_fin3, 0) == hipSuccess;
        if (joined && early_collide && !CRL_ABL(getenv("CRL_CAR_COLLIDE_WAIT") != nullptr)) c->collide_is_joined = true, c->collide_joined = st;
        if (c->K > 1) launch_car_stack(c->frame, c->stack, obs_dev, c->done_env, false, c->K, c->n, c->s.players, st);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return crl_fail(CRL_EHIP, "car step: %s", hipGetErrorString(e));
    return CRL_OK;
}

// ---- state exchange
template <class T>
static std::vector<T> pull(const T *dev, size_t count, hipStream_t st) {
    std::vector<T> v(count);
    hipMemcpyAsync(v.data(), dev, count * sizeof(T), hipMemcpyDeviceToHost, st);
    return v;
}

struct HostCopy {
    std::vector<float> body, jimp, jmotor, jspeed, sleep;
    std::vector<int32_t> jlimit, visited_count, last_block, done, step_count, first_step, elapsed;
    std::vector<double> wgas, womega, wphase, reward, prev_reward;
    std::vector<int16_t> wtiles;
    std::vector<uint32_t> visited, episode;
    std::vector<int32_t> n_contact, coupled;
    std::vector<float> contact;
};

static void pull_all(crl_car_ctx *c, HostCopy &h, hipStream_t st) {
    const int64_t n = c->n, M = (int64_t)c->s.players * n;
    const CarSoA &s = c->s;
    h.body = pull(s.body, 30 * M, st), h.jimp = pull(s.jimp, 12 * M, st), h.jmotor = pull(s.jmotor, 4 * M, st);
    h.jspeed = pull(s.jspeed, 4 * M, st), h.jlimit = pull(s.jlimit, 4 * M, st), h.sleep = pull(s.sleep, 5 * M, st);
    h.wgas = pull(s.wgas, 4 * M, st), h.womega = pull(s.womega, 4 * M, st), h.wphase = pull(s.wphase, 4 * M, st);
    h.wtiles = pull(s.wtiles, 4 * kWheelSlots * M, st), h.visited = pull(s.visited, 16 * M, st);
    h.reward = pull(s.reward, M, st), h.prev_reward = pull(s.prev_reward, M, st);
    h.visited_count = pull(s.visited_count, M, st), h.last_block = pull(s.last_block, M, st), h.done = pull(s.done, M, st);
    h.step_count = pull(s.step_count, M, st), h.first_step = pull(s.first_step, M, st);
    h.elapsed = pull(s.elapsed, n, st), h.episode = pull(s.episode, n, st);
    h.coupled = pull(s.coupled, n, st);
    h.n_contact = pull(s.n_contact, n, st), h.contact = pull(s.contact, (size_t)n * kMaxContacts * kContactWords, st);
    hipStreamSynchronize(st);
}

template <class T>
static void push(const std::vector<T> &v, T *dev, hipStream_t st) {
    hipMemcpyAsync(dev, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice, st);
}

int crl_car_get_state_impl(crl_car_ctx *c, crl_car_env_state *out, int64_t first, int64_t count, hipStream_t st) {
    if (first < 0 || count < 0 || first + count > c->n) return crl_fail(CRL_EINVAL, "bad range");
    HostCopy h;
    pull_all(c, h, st);
    const int64_t n = c->n, M = (int64_t)c->s.players * n;
    for (int64_t e = 0; e < count; e++) {
        crl_car_env_state &o = out[e];
        memset(&o, 0, sizeof(o));
        const int64_t env = first + e;
        for (int car = 0; car < c->s.players; car++) {
            const int64_t ci = car * n + env;
            crl_car_state &q = o.car[car];
            crl_car_body *bodies[5] = {&q.hull, &q.wheel[0], &q.wheel[1], &q.wheel[2], &q.wheel[3]};
            for (int b = 0; b < 5; b++) {
                float *f = &bodies[b]->cx;
                for (int k = 0; k < 6; k++) f[k] = h.body[(6 * b + k) * M + ci];
            }
            for (int w = 0; w < 4; w++) {
                for (int k = 0; k < 3; k++) q.imp[w][k] = h.jimp[(3 * w + k) * M + ci];
                q.motor_imp[w] = h.jmotor[w * M + ci], q.motor_speed[w] = h.jspeed[w * M + ci], q.limit_state[w] = h.jlimit[w * M + ci];
                q.gas[w] = h.wgas[w * M + ci], q.omega[w] = h.womega[w * M + ci], q.phase[w] = h.wphase[w * M + ci];
                for (int k = 0; k < kWheelSlots; k++) {
                    const int t = h.wtiles[(w * kWheelSlots + k) * M + ci];
                    if (t >= 0) q.wheel_tiles[w][t >> 5] |= 1u << (t & 31);
                }
            }
            for (int k = 0; k < 16; k++) q.visited[k] = h.visited[k * M + ci];
            q.reward = h.reward[ci], q.prev_reward = h.prev_reward[ci];
            q.tile_visited_count = h.visited_count[ci], q.last_block = h.last_block[ci], q.done = h.done[ci];
            q.step_count = h.step_count[ci], q.first_step = h.first_step[ci];
            for (int b = 0; b < 5; b++) q.sleep_time[b] = h.sleep[b * M + ci];
        }
        o.elapsed = h.elapsed[env], o.episode = h.episode[env];
        o.n_contact = h.n_contact[env], o.coupled = h.coupled[env];
        for (int k = 0; k < o.n_contact && k < kMaxContacts; k++)
            memcpy(&o.contact[k], &h.contact[((size_t)env * kMaxContacts + k) * kContactWords], sizeof(crl_car_contact));
    }
    return CRL_OK;
}

int crl_car_set_state_impl(crl_car_ctx *c, const crl_car_env_state *in, int64_t first, int64_t count, hipStream_t st) {
    void_collide_ahead(c, st);
    if (first < 0 || count < 0 || first + count > c->n) return crl_fail(CRL_EINVAL, "bad range");
    HostCopy h;
    pull_all(c, h, st);
    const int64_t n = c->n, M = (int64_t)c->s.players * n;
    for (int64_t e = 0; e < count; e++) {
        const crl_car_env_state &o = in[e];
        const int64_t env = first + e;
        for (int car = 0; car < c->s.players; car++) {
            const int64_t ci = car * n + env;
            const crl_car_state &q = o.car[car];
            const crl_car_body *bodies[5] = {&q.hull, &q.wheel[0], &q.wheel[1], &q.wheel[2], &q.wheel[3]};
            for (int b = 0; b < 5; b++) {
                const float *f = &bodies[b]->cx;
                for (int k = 0; k < 6; k++) h.body[(6 * b + k) * M + ci] = f[k];
            }
            for (int w = 0; w < 4; w++) {
                for (int k = 0; k < 3; k++) h.jimp[(3 * w + k) * M + ci] = q.imp[w][k];
                h.jmotor[w * M + ci] = q.motor_imp[w], h.jspeed[w * M + ci] = q.motor_speed[w], h.jlimit[w * M + ci] = q.limit_state[w];
                h.wgas[w * M + ci] = q.gas[w], h.womega[w * M + ci] = q.omega[w], h.wphase[w * M + ci] = q.phase[w];
                int slot = 0;
                for (int k = 0; k < kWheelSlots; k++) h.wtiles[(w * kWheelSlots + k) * M + ci] = -1;
                for (int t = 0; t < kCarMaxTiles; t++)
                    if ((q.wheel_tiles[w][t >> 5] >> (t & 31)) & 1) {
                        if (slot >= kWheelSlots) return crl_fail(CRL_EINVAL, "wheel touches more than %d tiles", kWheelSlots);
                        h.wtiles[(w * kWheelSlots + slot++) * M + ci] = (int16_t)t;
                    }
            }
            for (int k = 0; k < 16; k++) h.visited[k * M + ci] = q.visited[k];
            h.reward[ci] = q.reward, h.prev_reward[ci] = q.prev_reward;
            h.visited_count[ci] = q.tile_visited_count, h.last_block[ci] = q.last_block, h.done[ci] = q.done;
            h.step_count[ci] = q.step_count, h.first_step[ci] = q.first_step;
            for (int b = 0; b < 5; b++) h.sleep[b * M + ci] = q.sleep_time[b];
        }
        h.elapsed[env] = o.elapsed, h.episode[env] = o.episode;
        h.n_contact[env] = o.n_contact;
        for (int k = 0; k < o.n_contact && k < kMaxContacts; k++)
            memcpy(&h.contact[((size_t)env * kMaxContacts + k) * kContactWords], &o.contact[k], sizeof(crl_car_contact));
    }
    const CarSoA &s = c->s;
    push(h.body, s.body, st), push(h.jimp, s.jimp, st), push(h.jmotor, s.jmotor, st), push(h.jspeed, s.jspeed, st);
    push(h.jlimit, s.jlimit, st), push(h.wgas, s.wgas, st), push(h.womega, s.womega, st), push(h.wphase, s.wphase, st);
    push(h.wtiles, s.wtiles, st), push(h.visited, s.visited, st), push(h.reward, s.reward, st), push(h.prev_reward, s.prev_reward, st);
    push(h.visited_count, s.visited_count, st), push(h.last_block, s.last_block, st), push(h.done, s.done, st);
    push(h.step_count, s.step_count, st), push(h.first_step, s.first_step, st), push(h.elapsed, s.elapsed, st), push(h.episode, s.episode, st);
    push(h.n_contact, s.n_contact, st), push(h.contact, s.contact, st), push(h.sleep, s.sleep, st);
    hipStreamSynchronize(st);
    return CRL_OK;
}

int crl_car_get_track_impl(crl_car_ctx *c, int64_t env, int32_t *n_out, float *tile_poly, float *border_poly, uint8_t *border,
                           float *start_pose, hipStream_t st) {
    if (env < 0 || env >= c->n) return crl_fail(CRL_EINVAL, "env out of range");
    const int64_t n = c->n;
    int32_t nt = 0;
    hipMemcpyAsync(&nt, c->s.ntiles + env, 4, hipMemcpyDeviceToHost, st);
    hipStreamSynchronize(st);
    if (n_out) *n_out = nt;
    // strided gathers: one 2-D copy per array
    if (tile_poly) hipMemcpy2DAsync(tile_poly, 4, c->s.tile_poly + env, n * 4, 4, (size_t)nt * 10, hipMemcpyDeviceToHost, st);
    if (border_poly) hipMemcpy2DAsync(border_poly, 4, c->s.border_poly + env, n * 4, 4, (size_t)nt * 8, hipMemcpyDeviceToHost, st);
    if (border) hipMemcpy2DAsync(border, 1, c->s.border + env, n, 1, (size_t)nt, hipMemcpyDeviceToHost, st);
    if (start_pose) hipMemcpy2DAsync(start_pose, 4, c->s.start_pose + env, n * 4, 4, 3, hipMemcpyDeviceToHost, st);
    hipStreamSynchronize(st);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return crl_fail(CRL_EHIP, "get_track: %s", hipGetErrorString(e));
    return CRL_OK;
}

// Track of one env from the reference's float64 polygons (road_poly, crmp:400-441): tile i = (road1_l, road_m, road1_r,
// road2_r, road2_l), border quad (b1_l, b1_r, b2_r, b2_l).  Derives what the kernels keep: counter-clockwise float32 polygons
// (b2PolygonShape), their boxes, the integer map-space vertices of render_road_for_observation_map -- and rebuilds the map.
int crl_car_set_track_impl(crl_car_ctx *c, int64_t env, int32_t nt, const double *tile_poly, const double *border_poly,
                           const uint8_t *border, const float *start_pose, hipStream_t st) {
    if (env < 0 || env >= c->n || nt <= 0 || nt > kCarMaxTiles || !tile_poly) return crl_fail(CRL_EINVAL, "bad track");
    const int64_t n = c->n;
    std::vector<float4> aabb(nt);
    std::vector<float> tp((size_t)nt * 10), bp((size_t)nt * 8, 0.0f);
    std::vector<uint32_t> mv((size_t)nt * 9, 0u), yr(nt);
    std::vector<uint8_t> bflag(nt, 0);
    int32_t overflow = 0;
    auto map_vertex = [&](const double *v) -> uint32_t {
        const int mx = car_map_coord(v[0]) - kMapOrg, my = car_map_coord(v[1]) - kMapOrg;
        if (mx < 0 || my < 0 || mx >= kMapW || my >= kMapW) overflow++;
        const int cx = std::min(std::max(mx, -32768), 32767), cy = std::min(std::max(my, -32768), 32767);
        return (uint32_t)(uint16_t)(int16_t)cx | ((uint32_t)(uint16_t)(int16_t)cy << 16);
    };
    for (int t = 0; t < nt; t++) {
        float x0 = 3.4e38f, y0 = 3.4e38f, x1 = -3.4e38f, y1 = -3.4e38f;
        float tmp[5][2];
        ccw(reinterpret_cast<const double(*)[2]>(tile_poly + (size_t)t * 10), 5, 1.0, tmp);
        int ylo = 32767, yhi = -32768;
        for (int k = 0; k < 5; k++) {
            tp[t * 10 + 2 * k] = tmp[k][0], tp[t * 10 + 2 * k + 1] = tmp[k][1];
            x0 = fminf(x0, tmp[k][0]), y0 = fminf(y0, tmp[k][1]), x1 = fmaxf(x1, tmp[k][0]), y1 = fmaxf(y1, tmp[k][1]);
            const uint32_t w = mv[t * 9 + k] = map_vertex(tile_poly + (size_t)t * 10 + 2 * k);
            ylo = std::min(ylo, (int)(int16_t)(w >> 16)), yhi = std::max(yhi, (int)(int16_t)(w >> 16));
        }
        aabb[t] = make_float4(x0, y0, x1, y1);
        if (border && border[t] && border_poly) {
            float tb[4][2];
            ccw(reinterpret_cast<const double(*)[2]>(border_poly + (size_t)t * 8), 4, 1.0, tb);
            for (int k = 0; k < 4; k++) {
                bp[t * 8 + 2 * k] = tb[k][0], bp[t * 8 + 2 * k + 1] = tb[k][1];
                const uint32_t w = mv[t * 9 + 5 + k] = map_vertex(border_poly + (size_t)t * 8 + 2 * k);
                ylo = std::min(ylo, (int)(int16_t)(w >> 16)), yhi = std::max(yhi, (int)(int16_t)(w >> 16));
            }
            bflag[t] = border[t];
        }
        yr[t] = (uint32_t)(uint16_t)(int16_t)ylo | ((uint32_t)(uint16_t)(int16_t)yhi << 16);
    }
    hipMemcpyAsync(c->s.ntiles + env, &nt, 4, hipMemcpyHostToDevice, st);
    hipMemcpy2DAsync(c->s.tile_poly + env, n * 4, tp.data(), 4, 4, (size_t)nt * 10, hipMemcpyHostToDevice, st);
    hipMemcpy2DAsync(c->s.tile_aabb + env, n * 16, aabb.data(), 16, 16, (size_t)nt, hipMemcpyHostToDevice, st);
    std::vector<float4> blk((size_t)(nt + 7) / 8);
    for (int b = 0; b < (int)blk.size(); b++) {
        float4 u = aabb[8 * b];
        for (int t = 8 * b + 1; t < std::min(8 * b + 8, (int)nt); t++)
            u = make_float4(std::min(u.x, aabb[t].x), std::min(u.y, aabb[t].y), std::max(u.z, aabb[t].z), std::max(u.w, aabb[t].w));
        blk[b] = u;
    }
    hipMemcpy2DAsync(c->s.tile_blk + env, n * 16, blk.data(), 16, 16, blk.size(), hipMemcpyHostToDevice, st);
    hipMemcpy2DAsync(c->s.border_poly + env, n * 4, bp.data(), 4, 4, (size_t)nt * 8, hipMemcpyHostToDevice, st);
    hipMemcpy2DAsync(c->s.border + env, n, bflag.data(), 1, 1, (size_t)nt, hipMemcpyHostToDevice, st);
    if (start_pose) hipMemcpy2DAsync(c->s.start_pose + env, n * 4, start_pose, 4, 4, 3, hipMemcpyHostToDevice, st);
    // env-major copies (analytic raster, map build)
    hipMemcpyAsync(c->s.tile_poly_em + env * kCarMaxTiles * 10, tp.data(), (size_t)nt * 40, hipMemcpyHostToDevice, st);
    hipMemcpyAsync(c->s.tile_aabb_em + env * kCarMaxTiles, aabb.data(), (size_t)nt * 16, hipMemcpyHostToDevice, st);
    hipMemcpyAsync(c->s.border_poly_em + env * kCarMaxTiles * 8, bp.data(), (size_t)nt * 32, hipMemcpyHostToDevice, st);
    hipMemcpyAsync(c->s.border_em + env * kCarMaxTiles, bflag.data(), (size_t)nt, hipMemcpyHostToDevice, st);
    hipMemcpyAsync(c->s.map_vtx + env * kCarMaxTiles * 9, mv.data(), (size_t)nt * 36, hipMemcpyHostToDevice, st);
    hipMemcpyAsync(c->s.map_yr + env * kCarMaxTiles, yr.data(), (size_t)nt * 4, hipMemcpyHostToDevice, st);
    hipMemcpyAsync(c->s.map_overflow + env, &overflow, 4, hipMemcpyHostToDevice, st);
    launch_car_map_build(c->s, st, nullptr, env, 1);
    hipStreamSynchronize(st);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return crl_fail(CRL_EHIP, "set_track: %s", hipGetErrorString(e));
    return CRL_OK;
}

int crl_car_cap_hits_impl(crl_car_ctx *c, int32_t *out4, hipStream_t st) {
    hipMemcpyAsync(out4, c->s.cap_hits, 16, hipMemcpyDeviceToHost, st);
    if (hipStreamSynchronize(st) != hipSuccess) return crl_fail(CRL_EHIP, "cap_hits: copy failed");
    return CRL_OK;
}

// The env's pre-rastered map as one palette index per pixel, [CRL_CAR_MAP_W][CRL_CAR_MAP_W] (tests); *overflow = polygon
// vertices that fell outside the window at the last reset / set_track
int crl_car_get_map_impl(crl_car_ctx *c, int64_t env, uint8_t *palette_host, int32_t *overflow, hipStream_t st) {
    if (env < 0 || env >= c->n || !palette_host) return crl_fail(CRL_EINVAL, "bad argument");
    std::vector<uint8_t> raw((size_t)kMapBytes);
    uint8_t par = 0;
    hipMemcpyAsync(&par, c->s.map_par + env, 1, hipMemcpyDeviceToHost, st);
    if (hipStreamSynchronize(st) != hipSuccess) return crl_fail(CRL_EHIP, "get_map: copy failed");
    hipMemcpyAsync(raw.data(), c->s.obs_map + ((int64_t)(par & 1) * c->n + env) * kMapBytes, (size_t)kMapBytes, hipMemcpyDeviceToHost, st);
    int32_t ov = 0;
    hipMemcpyAsync(&ov, c->s.map_overflow + env, 4, hipMemcpyDeviceToHost, st);
    if (hipStreamSynchronize(st) != hipSuccess) return crl_fail(CRL_EHIP, "get_map: copy failed");
    if (overflow) *overflow = ov;
    for (int y = 0; y < kMapW; y++)
        for (int x = 0; x < kMapW; x++) {
            const uint8_t b = raw[(size_t)((y >> 4) * kMapBlocks + (x >> 4)) * 128 + (y & 15) * 8 + ((x & 15) >> 1)];
            palette_host[(size_t)y * kMapW + x] = (b >> ((x & 1) * 4)) & 15;
        }
    return CRL_OK;
}

int crl_car_set_replay_impl(crl_car_ctx *c, const double *u, const uint8_t *swap, int64_t attempts) {
    invalidate_walks(c);
    if (c->ru) hipFree(c->ru), c->ru = nullptr;
    if (c->rshuffle) hipFree(c->rshuffle), c->rshuffle = nullptr;
    c->src.ru = nullptr, c->src.rshuffle = nullptr, c->src.attempts = 0;
    if (attempts <= 0) return CRL_OK;
    if (!u || !swap) return crl_fail(CRL_EINVAL, "null replay arrays");
    const size_t m = (size_t)c->n * attempts;
    if (hipMalloc((void **)&c->ru, m * 24 * 8) != hipSuccess || hipMalloc((void **)&c->rshuffle, m) != hipSuccess)
        return crl_fail(CRL_ENOMEM, "replay alloc");
    hipMemcpy(c->ru, u, m * 24 * 8, hipMemcpyHostToDevice);
    hipMemcpy(c->rshuffle, swap, m, hipMemcpyHostToDevice);
    c->src.ru = c->ru, c->src.rshuffle = c->rshuffle, c->src.attempts = attempts;
    return CRL_OK;
}
