// car_track.hip -- CarRacing.reset on the GPU: procedural track + car placement, one lane per env.
//
// Restates CarRacing._create_track / reset (reference car_racing/car_racing_multi_players.py
// :262-452, :454-525) and Car.__init__ (car_dynamics.py:55-129).  The walk that lays the track
// is f64 with sin/cos/atan2; it keeps no 2500-point scratch list: pass 1 finds the two
// start-line crossings that delimit the lap, pass 2 replays the same deterministic walk and
// emits only the lap's points.  Runs at reset and for envs whose episode just ended, so its
// cost is amortised over ~1000 steps.
#include "car_device.h"

namespace crl {

__device__ inline void philox4x32_10c(uint32_t c[4], uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int r = 0; r < 10; r++) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c[0], p1 = (uint64_t)0xCD9E8D57u * c[2];
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0, n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1, n3 = (uint32_t)p0;
        c[0] = n0, c[1] = n1, c[2] = n2, c[3] = n3;
        k0 += 0x9E3779B9u, k1 += 0xBB67AE85u;
    }
}

struct Checkpoints {
    double a[12], x[12], y[12];
    double start_alpha;
};

struct Walk {  // state of the track-laying walk
    double x, y, beta;
    long dest_i;
    int laps, visited_other_side;
};

__device__ inline double sgnd(double v) { return (double)((v > 0) - (v < 0)); }

// one iteration of the walk; returns the appended point
__device__ inline void walk_step(const Checkpoints &cp, Walk &w, double out[4]) {
    const double PI = 3.141592653589793;
    double alpha = crl_atan2_fast(w.y, w.x);
    if (w.visited_other_side && alpha > 0) w.laps++, w.visited_other_side = 0;
    if (alpha < 0) w.visited_other_side = 1, alpha += 2 * PI;
    double dest_x = 0, dest_y = 0;
    for (;;) {
        bool failed = true;
        for (;;) {
            const int k = (int)(w.dest_i % 12);
            dest_x = cp.x[k], dest_y = cp.y[k];
            if (alpha <= cp.a[k]) { failed = false; break; }
            w.dest_i++;
            if (w.dest_i % 12 == 0) break;
        }
        if (!failed) break;
        alpha -= 2 * PI;
    }
    const double r1x = crl_cos_fast(w.beta), r1y = crl_sin_fast(w.beta), p1x = -r1y, p1y = r1x;
    const double dest_dx = dest_x - w.x, dest_dy = dest_y - w.y;
    double proj = r1x * dest_dx + r1y * dest_dy;
    while (w.beta - alpha > 1.5 * PI) w.beta -= 2 * PI;
    while (w.beta - alpha < -1.5 * PI) w.beta += 2 * PI;
    const double prev_beta = w.beta;
    proj *= CAR_SCALE;
    if (proj > 0.3) w.beta -= fmin(CAR_TRACK_TURN_RATE, fabs(0.001 * proj));
    if (proj < -0.3) w.beta += fmin(CAR_TRACK_TURN_RATE, fabs(0.001 * proj));
    w.x += p1x * CAR_TRACK_DETAIL_STEP;
    w.y += p1y * CAR_TRACK_DETAIL_STEP;
    out[0] = alpha, out[1] = prev_beta * 0.5 + w.beta * 0.5, out[2] = w.x, out[3] = w.y;
}

__device__ inline void walk_init(Walk &w) {
    w.x = 1.5 * CAR_TRACK_RAD, w.y = 0, w.beta = 0, w.dest_i = 0, w.laps = 0, w.visited_other_side = 0;
}

// Walks one track attempt, keeping EVERY point in `pts` (global scratch [kWalkMax][4] per env, env-major -- with the points strided by
// the env count every store of the milliseconds-long walk-ahead opened another page, and the TLB misses slowed whatever ran beside it:
// 80 KB per env buys not having to repeat the ~2 500-step f64 walk once the lap's end points are known).
// On success returns the lap length and, in *first, the index of its first point; 0 otherwise.
static constexpr int kWalkMax = 2500;
__device__ inline void make_checkpoints(const double u[24], Checkpoints &cp) {
    const double PI = 3.141592653589793;
    cp.start_alpha = 0;
    for (int c = 0; c < 12; c++) {
        const double noise = 0 + (2 * PI * 1 / 12 - 0) * u[2 * c];
        double alpha = 2 * PI * c / 12 + noise;
        double rad = CAR_TRACK_RAD / 3 + (CAR_TRACK_RAD - CAR_TRACK_RAD / 3) * u[2 * c + 1];
        if (c == 0) alpha = 0, rad = 1.5 * CAR_TRACK_RAD;
        if (c == 11) alpha = 2 * PI * c / 12, cp.start_alpha = 2 * PI * (-0.5) / 12, rad = 1.5 * CAR_TRACK_RAD;
        cp.a[c] = alpha, cp.x[c] = rad * crl_cos_fast(alpha), cp.y[c] = rad * crl_sin_fast(alpha);
    }
}

// one attempt's walk as a resumable run: the walk-ahead advances it a bounded number of iterations per launch
struct WalkRun {
    Walk w;
    int n, cross_last, cross_prev, no_freeze;
    double prev_alpha;
};
__device__ inline void walk_run_init(WalkRun &r) {
    walk_init(r.w);
    r.n = 0, r.cross_last = -1, r.cross_prev = -1, r.no_freeze = kWalkMax, r.prev_alpha = 0;
}
// at most `budget` iterations; true when the walk has ended (five laps, or kWalkMax points)
__device__ inline bool walk_run(const Checkpoints &cp, WalkRun &r, double *__restrict__ pts, int64_t stride, int budget) {
    for (; budget > 0; budget--) {
        double p[4];
        walk_step(cp, r.w, p);
#pragma unroll
        for (int q = 0; q < 4; q++) pts[((int64_t)r.n * 4 + q) * stride] = p[q];
        if (r.n > 0 && p[0] > cp.start_alpha && r.prev_alpha <= cp.start_alpha) r.cross_prev = r.cross_last, r.cross_last = r.n;
        r.prev_alpha = p[0];
        r.n++;
        if (r.w.laps > 4) return true;
        if (--r.no_freeze == 0) return true;
    }
    return false;
}
// the finished walk's lap: its length and, in *first, the index of its first point; 0 = the attempt failed
__device__ inline int walk_close(const WalkRun &r, const double *__restrict__ pts, int64_t stride, int *first) {
    // the reference scans i = n-1 .. 1 and fails at i == 0 before testing it
    const int i2 = r.cross_last, i1 = r.cross_prev;
    if (i2 < 1 || i1 < 1) return 0;
    const int len = (i2 - 1) - i1;  // points i1 .. i2-2
    if (len <= 0 || len > kCarMaxTiles) return 0;
    const double *trk = pts + (int64_t)i1 * 4 * stride;
    const double fb = trk[1 * stride], fpx = crl_cos_fast(fb), fpy = crl_sin_fast(fb);
    const double a = fpx * (trk[2 * stride] - trk[((int64_t)(len - 1) * 4 + 2) * stride]);
    const double b = fpy * (trk[3 * stride] - trk[((int64_t)(len - 1) * 4 + 3) * stride]);
    if (sqrt(a * a + b * b) > CAR_TRACK_DETAIL_STEP) return 0;
    *first = i1;
    return len;
}
__device__ int create_track(const double u[24], double *__restrict__ pts, int64_t stride, int *first) {
    Checkpoints cp;
    make_checkpoints(u, cp);
    WalkRun r;
    walk_run_init(r);
    walk_run(cp, r, pts, stride, 0x7fffffff);
    return walk_close(r, pts, stride, first);
}

__device__ inline void store_poly_ccw(const double (*v)[2], int nv, float *dst, int64_t stride, float *aabb) {
    double area = 0;
    for (int i = 0; i < nv; i++) {
        const int j = (i + 1) % nv;
        area += v[i][0] * v[j][1] - v[j][0] * v[i][1];
    }
    float x0 = 3.4e38f, y0 = 3.4e38f, x1 = -3.4e38f, y1 = -3.4e38f;
    for (int i = 0; i < nv; i++) {
        const int k = area > 0 ? i : nv - 1 - i;
        const float x = (float)v[k][0], y = (float)v[k][1];
        dst[(2 * i) * stride] = x, dst[(2 * i + 1) * stride] = y;
        x0 = fminf(x0, x), y0 = fminf(y0, y), x1 = fmaxf(x1, x), y1 = fmaxf(y1, y);
    }
    if (aabb) aabb[0] = x0, aabb[1] = y0, aabb[2] = x1, aabb[3] = y1;
}

// Builds tiles / borders from the lap in `trk` and places both cars.  swap = np.random.shuffle
// outcome for the two birth places (crmp:508-512).  Called by a whole wavefront for ONE env: the
// tiles are independent of each other and go one per lane (20 f64 sin/cos each); only the in-place
// dilation of the border flags (which chains through the wrap-around, crmp:384-396) is done by one
// lane, in LDS.
__device__ void finish_reset(CarSoA &s, const CarConsts &K, int64_t env, const double *trk, int len, int swap, uint8_t *flag /* LDS [512] */) {
    const int64_t n = s.n, M = (int64_t)s.players * n;
    const int lane = threadIdx.x & 63;
    auto T = [&](int i, int q) { return trk[(int64_t)i * 4 + q]; };  // (env-major scratch: a walk's points are contiguous)
    if (lane == 0) s.ntiles[env] = len, s.map_overflow[env] = 0;
    // one vertex of a polygon of render_road_for_observation_map (crmp:745-753): (obs_scale * -v + world_size / 2), truncated
    // by pygame; kept in window coordinates as int16 (the window holds every track: map_overflow counts what does not fit)
    int overflow = 0;
    auto map_vertex = [&](const double *v) -> uint32_t {
        const int mx = (int)(CRL_CAR_OBS_SCALE * -v[0] + kMapSurface / 2.0) - kMapOrg, my = (int)(CRL_CAR_OBS_SCALE * -v[1] + kMapSurface / 2.0) - kMapOrg;
        if (mx < 0 || my < 0 || mx >= kMapW || my >= kMapW) overflow++;
        const int cx = min(max(mx, -32768), 32767), cy = min(max(my, -32768), 32767);
        return (uint32_t)(uint16_t)(int16_t)cx | ((uint32_t)(uint16_t)(int16_t)cy << 16);
    };
    // red-white border on hard turns: 4 consecutive same-sign turns ...
    for (int i = lane; i < len; i += 64) {
        bool good = true;
        double oneside = 0;
        for (int neg = 0; neg < 4; neg++) {
            const double b1 = T(((i - neg) % len + len) % len, 1), b2 = T(((i - neg - 1) % len + len) % len, 1);
            good = good && fabs(b1 - b2) > CAR_TRACK_TURN_RATE * 0.2;
            oneside += sgnd(b1 - b2);
        }
        good = good && fabs(oneside) == 4;
        flag[i] = good ? 1 : 0;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    // ... then dilated backwards IN PLACE exactly like the reference loop (wrap-around entries chain)
    if (lane == 0)
        for (int i = 0; i < len; i++)
            if (flag[i])
                for (int neg = 0; neg < 4; neg++) flag[((i - neg) % len + len) % len] = 1;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    for (int i = lane; i < len; i += 64) {
        const int j = ((i - 1) % len + len) % len;
        const double b1 = T(i, 1), x1 = T(i, 2), y1 = T(i, 3), b2 = T(j, 1), x2 = T(j, 2), y2 = T(j, 3);
        const double PI = 3.141592653589793;
        // (each angle's sine and cosine once: the reference evaluates math.cos / math.sin of the same argument again and again)
        double s1, c1, s2, c2, sm, cm;
        crl_sincos_fast(b1, &s1, &c1), crl_sincos_fast(b2, &s2, &c2), crl_sincos_fast(b1 - PI / 2, &sm, &cm);
        const double v[5][2] = {
            {x1 - CAR_TRACK_WIDTH * c1, y1 - CAR_TRACK_WIDTH * s1},
            {x1 - CAR_TRACK_WIDTH / 2 * cm, y1 - CAR_TRACK_WIDTH / 2 * sm},
            {x1 + CAR_TRACK_WIDTH * c1, y1 + CAR_TRACK_WIDTH * s1},
            {x2 + CAR_TRACK_WIDTH * c2, y2 + CAR_TRACK_WIDTH * s2},
            {x2 - CAR_TRACK_WIDTH * c2, y2 - CAR_TRACK_WIDTH * s2},
        };
        float bb[4];
        uint32_t *mv = s.map_vtx + (env * kCarMaxTiles + i) * 9;
        int ylo = 32767, yhi = -32768;
        for (int q = 0; q < 5; q++) {
            const uint32_t w = map_vertex(v[q]);
            mv[q] = w;
            ylo = min(ylo, (int)(int16_t)(w >> 16)), yhi = max(yhi, (int)(int16_t)(w >> 16));
        }
        store_poly_ccw(v, 5, s.tile_poly + (int64_t)i * 10 * n + env, n, bb);
        s.tile_aabb[(int64_t)i * n + env] = make_float4(bb[0], bb[1], bb[2], bb[3]);
        s.tile_aabb_em[env * kCarMaxTiles + i] = make_float4(bb[0], bb[1], bb[2], bb[3]);
        for (int q = 0; q < 10; q++) s.tile_poly_em[(env * kCarMaxTiles + i) * 10 + q] = s.tile_poly[((int64_t)i * 10 + q) * n + env];
        uint8_t bflag = 0;
        if (flag[i]) {
            const double side = sgnd(b2 - b1);
            const double bp[4][2] = {
                {x1 + side * CAR_TRACK_WIDTH * c1, y1 + side * CAR_TRACK_WIDTH * s1},
                {x1 + side * (CAR_TRACK_WIDTH + CAR_BORDER) * c1, y1 + side * (CAR_TRACK_WIDTH + CAR_BORDER) * s1},
                {x2 + side * (CAR_TRACK_WIDTH + CAR_BORDER) * c2, y2 + side * (CAR_TRACK_WIDTH + CAR_BORDER) * s2},
                {x2 + side * CAR_TRACK_WIDTH * c2, y2 + side * CAR_TRACK_WIDTH * s2},
            };
            for (int q = 0; q < 4; q++) {
                const uint32_t w = map_vertex(bp[q]);
                mv[5 + q] = w;
                ylo = min(ylo, (int)(int16_t)(w >> 16)), yhi = max(yhi, (int)(int16_t)(w >> 16));
            }
            store_poly_ccw(bp, 4, s.border_poly + (int64_t)i * 8 * n + env, n, nullptr);
            bflag = (i % 2 == 0) ? 1 : 2;  // white / red
            for (int q = 0; q < 8; q++) s.border_poly_em[(env * kCarMaxTiles + i) * 8 + q] = s.border_poly[((int64_t)i * 8 + q) * n + env];
        }
        s.border[(int64_t)i * n + env] = bflag;
        s.border_em[env * kCarMaxTiles + i] = bflag;
        s.map_yr[env * kCarMaxTiles + i] = (uint32_t)(uint16_t)(int16_t)ylo | ((uint32_t)(uint16_t)(int16_t)yhi << 16);
    }
    if (overflow) atomicAdd(&s.map_overflow[env], overflow);
    // the boxes of 8 consecutive tiles, united: the sensors' broadphase tests these first (the tile boxes above were stored by
    // other lanes of this wavefront: agent-scope fence before they are read back)
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "agent");
    for (int b = lane; 8 * b < len; b += 64) {
        float4 u = s.tile_aabb_em[env * kCarMaxTiles + 8 * b];
        for (int t = 8 * b + 1; t < min(8 * b + 8, len); t++) {
            const float4 q = s.tile_aabb_em[env * kCarMaxTiles + t];
            u = make_float4(fminf(u.x, q.x), fminf(u.y, q.y), fmaxf(u.z, q.z), fmaxf(u.w, q.w));
        }
        s.tile_blk[(int64_t)b * n + env] = u;
    }
    const double ia = T(0, 1), ix = T(0, 2), iy = T(0, 3);
    if (lane == 0) s.start_pose[0 * n + env] = (float)ia, s.start_pose[1 * n + env] = (float)ix, s.start_pose[2 * n + env] = (float)iy;
    if (lane < s.players) {
        const int car = lane;
        const int64_t ci = car * n + env;
        // np.random.shuffle(arange(num_player)): one car always gets birth place 0
        const int birth = s.players == 1 ? 0 : (car == 0 ? (swap ? 1 : 0) : (swap ? 0 : 1));
        const double x0 = ix - (birth % 2) * 5, y0 = iy - floor(birth / 2.0) * 10;
        const float a = (float)ia;
        float sa, ca;
        crl_sincosf(a, &sa, &ca);
        const V2 com = mk((float)x0, (float)y0) + rotv(sa, ca, mk(K.hull_lc[0], K.hull_lc[1]));
        float *b = s.body + ci;
        for (int k = 0; k < 30; k++) b[k * M] = 0.f;
        b[0 * M] = com.x, b[1 * M] = com.y, b[2 * M] = a;
        const double wpos[4][2] = {{-55, +80}, {+55, +80}, {-55, -82}, {+55, -82}};
        for (int w = 0; w < 4; w++) {
            b[(6 + 6 * w + 0) * M] = (float)(x0 + wpos[w][0] * CAR_SIZE);
            b[(6 + 6 * w + 1) * M] = (float)(y0 + wpos[w][1] * CAR_SIZE);
            b[(6 + 6 * w + 2) * M] = a;
            s.jmotor[w * M + ci] = 0.f, s.jspeed[w * M + ci] = 0.f, s.jlimit[w * M + ci] = 0;
            for (int q = 0; q < 3; q++) s.jimp[(3 * w + q) * M + ci] = 0.f;
            s.wgas[w * M + ci] = 0.0, s.womega[w * M + ci] = 0.0, s.wphase[w * M + ci] = 0.0;
            for (int q = 0; q < kWheelSlots; q++) s.wtiles[(w * kWheelSlots + q) * M + ci] = -1;
        }
        for (int q = 0; q < 16; q++) s.visited[q * M + ci] = 0u;
        s.reward[ci] = 0.0, s.prev_reward[ci] = 0.0;
        s.visited_count[ci] = 0, s.last_block[ci] = -1, s.done[ci] = 0, s.step_count[ci] = 0, s.first_step[ci] = 1;
        for (int bq = 0; bq < 5; bq++) s.sleep[bq * M + ci] = 0.0f;
    }
    if (lane == 0) {
        s.elapsed[env] = 0;
        if (s.n_contact) s.n_contact[env] = 0, s.coupled[env] = 0;
    }
}

// The attempts loop of CarRacing.reset (crmp:454-525): fresh draws until a lap closes.  The draws of attempt a
// only depend on (seed, global env id, episode, a) -- or on the replay stream -- so the walk of an env's NEXT
// episode can be generated at any time before that reset.
// the 24 uniform draws + the birth-place swap of attempt `attempt` of episode `episode` of an env
__device__ inline void draw_attempt(const CarTrackSrc &src, int64_t env, uint32_t episode, int attempt, double u[24], int *swap) {
    if (src.attempts > 0) {
        const int64_t a = ((int64_t)episode * 16 + attempt) % src.attempts;
        for (int k = 0; k < 24; k++) u[k] = src.ru[(env * src.attempts + a) * 24 + k];
        *swap = src.rshuffle[env * src.attempts + a];
    } else {
        const uint64_t gid = (uint64_t)(src.env_id_base + env);
        for (int k = 0; k < 13; k++) {
            uint32_t c[4] = {(uint32_t)gid, (uint32_t)(gid >> 32), (episode << 12) | ((uint32_t)attempt << 4) | (uint32_t)k,
                             0x43415253u /* "CARS" */};
            philox4x32_10c(c, (uint32_t)src.seed, (uint32_t)(src.seed >> 32));
            if (k < 12) {
                u[2 * k] = (double)((((uint64_t)c[0] << 32) | c[1]) >> 11) * (1.0 / 9007199254740992.0);
                u[2 * k + 1] = (double)((((uint64_t)c[2] << 32) | c[3]) >> 11) * (1.0 / 9007199254740992.0);
            } else {
                *swap = c[0] & 1;
            }
        }
    }
}
static constexpr int kMaxAttempts = 256;
__device__ void gen_walk(const CarSoA &s, const CarTrackSrc &src, int64_t env, uint32_t episode, double *pts, int *len_out,
                         int *first_out, int *swap_out) {
    int len = 0, swap = 0, first = 0;
    for (int attempt = 0; attempt < kMaxAttempts && len == 0; attempt++) {
        double u[24];
        draw_attempt(src, env, episode, attempt, u, &swap);
        len = create_track(u, pts, 1, &first);
    }
    *len_out = len, *first_out = first, *swap_out = swap;
}

// Reset of the finished envs.  The ~4.6 ms walk is normally already there (car_walk_ahead_kernel, tagged with
// the episode it belongs to); then only the tiles are built and the cars placed.  Without a finished
// walk-ahead (first episode, a reset right after a reset) the walk is done here, into its own scratch.
__device__ inline void reset_one_env(CarSoA &s, const CarConsts &K, const CarTrackSrc &src, int64_t env, uint8_t *flag);

__global__ __launch_bounds__(64) void car_reset_kernel(CarSoA s, CarConsts K, CarTrackSrc src, int only_done,
                                                       const uint8_t *__restrict__ done_env) {
    // one WAVEFRONT per env: the tiles are built one per lane; a missing walk is done by lane 0
    __shared__ uint8_t flag[kCarMaxTiles];
    const int64_t env = blockIdx.x;
    if (only_done && !done_env[env]) return;
    reset_one_env(s, K, src, env, flag);
}

// the same over a compacted list of envs (the finished envs of a step: a handful of wavefronts instead of one per env that mostly exit)
__global__ __launch_bounds__(64) void car_reset_list_kernel(CarSoA s, CarConsts K, CarTrackSrc src, const int32_t *__restrict__ list,
                                                            const int32_t *__restrict__ list_count) {
    __shared__ uint8_t flag[kCarMaxTiles];
    const int count = *list_count;
    for (int i = blockIdx.x; i < count; i += gridDim.x) {
        reset_one_env(s, K, src, (int64_t)list[i], flag);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    }
}

__device__ inline void reset_one_env(CarSoA &s, const CarConsts &K, const CarTrackSrc &src, int64_t env, uint8_t *flag) {
    const int lane = threadIdx.x;
    const uint32_t episode = __hip_atomic_load(&s.episode[env], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    int len = 0, swap = 0, first = 0;
    const double *pts;
    if (__hip_atomic_load(&s.walk_tag[env], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) == episode) {
        len = s.walk_len[env], first = s.walk_first[env], swap = s.walk_swap[env];
        pts = s.track_scratch + env * (int64_t)(kWalkMax * 4);
    } else {
        if (lane == 0) gen_walk(s, src, env, episode, s.track_scratch_b + env * (int64_t)(kWalkMax * 4), &len, &first, &swap);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "agent");  // lane 0's points are read by the whole wave below
        len = __shfl(len, 0), first = __shfl(first, 0), swap = __shfl(swap, 0);
        pts = s.track_scratch_b + env * (int64_t)(kWalkMax * 4);
    }
    finish_reset(s, K, env, pts + (int64_t)first * 4, len, swap, flag);
    // The episode index is what the walk-ahead compares its tag with: it is published only now, with a release, after
    // every lane has consumed the stored walk -- a walk-ahead wavefront that starts while this reset is still reading
    // `track_scratch` sees the old index (tag == episode: nothing to do) and cannot overwrite the points under it.
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
#ifdef CRL_ABLATION
    // timing ablation (WRONG tracks: every episode of an env replays the walk it has): no walk-ahead kernel ever runs; what a step costs
    // without one in flight (CRL_CAR_ABL_NO_WALK, docs/LAB_NOTES_r05.md)
    if (s.abl_no_walk && lane == 0 && pts == s.track_scratch + env * (int64_t)(kWalkMax * 4))
        __hip_atomic_store(&s.walk_tag[env], episode + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
    if (lane == 0) __hip_atomic_store(&s.episode[env], episode + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}

// Walk-ahead: for every env whose stored walk is not the one its next reset needs, generate it.  Runs on its
// own stream beside the steps; a reset that comes before it has finished simply walks inline.
// Two kernels: the first lists the envs that need a walk (a few dozen of 16 384), the second walks them in DENSE wavefronts --
// one lane per env directly would keep a 165-register wavefront resident for milliseconds on every CU that holds one such env, and
// a frame workgroup needs a slot on all four SIMDs of its CU.
__global__ __launch_bounds__(256) void car_walk_mark_kernel(CarSoA s, int32_t *__restrict__ list, int32_t *__restrict__ count) {
    const int64_t env = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    bool need = false;
    if (env < s.n) {
        // the index the env's next reset will use (acquire: pairs with the release at the end of the reset)
        const uint32_t episode = __hip_atomic_load(&s.episode[env], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
        need = __hip_atomic_load(&s.walk_tag[env], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != episode;
    }
    const unsigned long long m = __ballot(need);
    if (!m) return;
    const int lane = threadIdx.x & 63;
    int base = 0;
    if (lane == 0) base = atomicAdd(count, (int)__popcll(m));
    base = __shfl(base, 0);
    if (need) list[base + (int)__popcll(m & ((1ull << lane) - 1ull))] = (int32_t)env;
}

// What a walk-ahead launch leaves behind for the next one: the attempt it is in and that attempt's walk (72 bytes per env).
struct WalkSave {
    uint32_t episode;  // the episode this state belongs to (anything else: start from attempt 0)
    int32_t attempt, n, cross_last, cross_prev, no_freeze, laps, visited_other_side;
    int64_t dest_i;
    double prev_alpha, x, y, beta;
};
static_assert(sizeof(WalkSave) == kWalkSaveWords * 4, "WalkSave words");

// `budget` walk iterations per lane and launch (<= 0: to the end).  The step pipeline queues one bounded launch per step on its
// low-priority stream: a walk of 2 500 iterations x a few attempts is 5-17 ms of ONE lane's latency, and a kernel of that length
// was what a device-wide synchronise at the end of a short timed window waited for (up to 0.85 ms per step of a 20-step window).
__global__ __launch_bounds__(64) void car_walk_ahead_kernel(CarSoA s, CarTrackSrc src, const int32_t *__restrict__ list,
                                                            const int32_t *__restrict__ count, int budget) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= *count) return;
    const int64_t env = list[i];
    const uint32_t episode = __hip_atomic_load(&s.episode[env], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
    if (__hip_atomic_load(&s.walk_tag[env], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == episode) return;
    WalkSave *save = reinterpret_cast<WalkSave *>(s.walk_save) + env;
    double *pts = s.track_scratch + env * (int64_t)(kWalkMax * 4);
    // The stored walk (of an EARLIER episode: the tag differs from the one the next reset asks for) is about to be overwritten, in pieces
    // over the next 20-80 steps: from here on it must not be taken for anybody's walk -- a set_state that puts `episode` back to the
    // stored walk's index would otherwise find tag == episode and build its track from points partly replaced by this one.
    // With the tag void such a reset walks inline into the other scratch, like any reset that comes before its walk-ahead has finished.
#ifdef CRL_ABLATION
    if (!s.abl_keep_tag)  // (profiling build, CRL_CAR_ABL_KEEP_TAG: round 4's behaviour, to show that the test of this hazard detects it)
#endif
    __hip_atomic_store(&s.walk_tag[env], 0xFFFFFFFFu, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    WalkRun r;
    int attempt = 0;
    if (budget > 0 && save->episode == episode) {
        attempt = save->attempt;
        r.n = save->n, r.cross_last = save->cross_last, r.cross_prev = save->cross_prev, r.no_freeze = save->no_freeze;
        r.w.laps = save->laps, r.w.visited_other_side = save->visited_other_side, r.w.dest_i = (long)save->dest_i;
        r.prev_alpha = save->prev_alpha, r.w.x = save->x, r.w.y = save->y, r.w.beta = save->beta;
    } else {
        walk_run_init(r);
    }
    int left = budget > 0 ? budget : 0x7fffffff;
    for (;;) {
        double u[24];
        int swap = 0;
        draw_attempt(src, env, episode, attempt, u, &swap);
        Checkpoints cp;
        make_checkpoints(u, cp);
        const int n0 = r.n;
        const bool ended = walk_run(cp, r, pts, 1, left);
        left -= r.n - n0;
        if (!ended) break;  // out of budget in the middle of an attempt
        int first = 0;
        const int len = walk_close(r, pts, 1, &first);
        if (len > 0 || attempt + 1 >= kMaxAttempts) {  // (gen_walk: the lap, or nothing after the last attempt)
            s.walk_len[env] = len, s.walk_first[env] = first, s.walk_swap[env] = swap;
            __hip_atomic_store(&s.walk_tag[env], episode, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            return;
        }
        attempt++;
        walk_run_init(r);
        if (left <= 0) break;
    }
    save->episode = episode, save->attempt = attempt;
    save->n = r.n, save->cross_last = r.cross_last, save->cross_prev = r.cross_prev, save->no_freeze = r.no_freeze;
    save->laps = r.w.laps, save->visited_other_side = r.w.visited_other_side, save->dest_i = (int64_t)r.w.dest_i;
    save->prev_alpha = r.prev_alpha, save->x = r.w.x, save->y = r.w.y, save->beta = r.w.beta;
}

void launch_car_reset(const CarSoA &s, const CarConsts &k, const CarTrackSrc &src, bool only_done, const uint8_t *done_env,
                      hipStream_t st) {
    hipLaunchKernelGGL(car_reset_kernel, dim3((unsigned)s.n), dim3(64), 0, st, s, k, src, only_done ? 1 : 0, done_env);
}

void launch_car_reset_list(const CarSoA &s, const CarConsts &k, const CarTrackSrc &src, const int32_t *list, const int32_t *list_count,
                           int64_t expected, hipStream_t st) {
    int64_t want = expected + expected / 4 + 32;
    want = want > s.n ? s.n : want;
    hipLaunchKernelGGL(car_reset_list_kernel, dim3((unsigned)want), dim3(64), 0, st, s, k, src, list, list_count);
}

// The step pipeline resets a finished env EARLY, beside the solves: track arrays in place (nothing reads them once the step's
// sensor contacts are in), the new map into the env's other slot, the cars' state into a staged copy of the per-car arrays
// (finish_reset called on a CarSoA whose car-state pointers address the staging arrays) -- the env's terminal frame still
// needs the old map and the solved bodies.  Once that frame is drawn this kernel makes the new episode current.
__global__ __launch_bounds__(64) void car_commit_list_kernel(CarSoA live, CarSoA stg, const int32_t *__restrict__ list, const int32_t *__restrict__ list_count) {
    const int count = *list_count, lane = threadIdx.x;
    const int64_t n = live.n, M = (int64_t)live.players * n;
    for (int i = blockIdx.x; i < count; i += gridDim.x) {
        const int64_t env = list[i];
        for (int car = 0; car < live.players; car++) {
            const int64_t ci = car * n + env;
#define CRL_CP(f, rows) \
    for (int r = lane; r < (rows); r += 64) live.f[(int64_t)r * M + ci] = stg.f[(int64_t)r * M + ci];
            CRL_CP(body, 30) CRL_CP(jmotor, 4) CRL_CP(jspeed, 4) CRL_CP(jlimit, 4) CRL_CP(jimp, 12) CRL_CP(wgas, 4) CRL_CP(womega, 4) CRL_CP(wphase, 4)
            CRL_CP(wtiles, 4 * kWheelSlots) CRL_CP(visited, 16) CRL_CP(reward, 1) CRL_CP(prev_reward, 1) CRL_CP(visited_count, 1) CRL_CP(last_block, 1)
            CRL_CP(done, 1) CRL_CP(step_count, 1) CRL_CP(first_step, 1) CRL_CP(sleep, 5)
#undef CRL_CP
        }
        if (lane == 0) {
            live.elapsed[env] = stg.elapsed[env];
            if (live.n_contact) live.n_contact[env] = stg.n_contact[env], live.coupled[env] = stg.coupled[env];
            live.map_par[env] ^= 1;
        }
    }
}

void launch_car_commit_list(const CarSoA &live, const CarSoA &stage, const int32_t *list, const int32_t *list_count, int64_t expected, hipStream_t st) {
    int64_t want = expected + expected / 4 + 32;
    want = want > live.n ? live.n : want;
    hipLaunchKernelGGL(car_commit_list_kernel, dim3((unsigned)want), dim3(64), 0, st, live, stage, list, list_count);
}

void launch_car_walk_ahead(const CarSoA &s, const CarTrackSrc &src, hipStream_t st, int budget) {
    hipMemsetAsync(s.walk_count, 0, sizeof(int32_t), st);
    hipLaunchKernelGGL(car_walk_mark_kernel, dim3((unsigned)((s.n + 255) / 256)), dim3(256), 0, st, s, s.walk_list, s.walk_count);
    hipLaunchKernelGGL(car_walk_ahead_kernel, dim3((unsigned)((s.n + 63) / 64)), dim3(64), 0, st, s, src, s.walk_list, s.walk_count, budget);
}

}  // namespace crl
