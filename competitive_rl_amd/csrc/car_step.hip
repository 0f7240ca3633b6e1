// car_step.hip -- one cCarRacingDouble step, one lane per CAR INSTANCE (2 lanes per env).
//
// The two cars of an env only interact through contacts; unless their oriented boxes overlap
// (then car_contact.hip solves them together) each car is an independent Box2D island: wheel model (f64) -> sensor overlap with the
// track tiles (Begin/EndContact -> tile rewards) -> island solve (180 velocity iterations over
// 4 revolute joints, <= 60 position iterations) entirely in registers.  Bound by the sequential
// Gauss-Seidel chain (VALU latency), not by HBM: ~1.3 KB of state per car per step.
#include "car_solver.h"

namespace crl {

__device__ inline double sgn(double v) { return (double)((v > 0) - (v < 0)); }

// ---- convex polygon distance: what b2TestOverlap (GJK distance < radii) decides for sensors
__device__ inline float seg_seg_dist2(V2 p1, V2 q1, V2 p2, V2 q2) {
    const V2 d1 = q1 - p1, d2 = q2 - p2, r = p1 - p2;
    const float a = dot(d1, d1), e = dot(d2, d2), f = dot(d2, r);
    float s, t;
    const float EPS = 1e-12f;
    if (a <= EPS && e <= EPS) return dot(r, r);
    if (a <= EPS) {
        s = 0, t = fminf(fmaxf(f / e, 0.f), 1.f);
    } else {
        const float c = dot(d1, r);
        if (e <= EPS) {
            t = 0, s = fminf(fmaxf(-c / a, 0.f), 1.f);
        } else {
            const float b = dot(d1, d2), den = a * e - b * b;
            s = den != 0 ? fminf(fmaxf((b * f - c * e) / den, 0.f), 1.f) : 0.f;
            t = (b * s + f) / e;
            if (t < 0) t = 0, s = fminf(fmaxf(-c / a, 0.f), 1.f);
            else if (t > 1) t = 1, s = fminf(fmaxf((b - c) / a, 0.f), 1.f);
        }
    }
    const V2 d = (p1 + s * d1) - (p2 + t * d2);
    return dot(d, d);
}

template <int NP>
__device__ inline bool point_in_convex(V2 p, const V2 (&poly)[NP]) {
    bool in = true;
#pragma unroll
    for (int i = 0; i < NP; i++) {
        const V2 a = poly[i], b = poly[i + 1 < NP ? i + 1 : 0];
        in = in && !(cross(b - a, p - a) < 0);
    }
    return in;
}

__device__ inline float poly_dist2(const V2 (&A)[4], const V2 (&B)[5]) {
    if (point_in_convex<5>(A[0], B) || point_in_convex<4>(B[0], A)) return 0.f;
    float best = 3.4e38f;
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 5; j++) best = fminf(best, seg_seg_dist2(A[i], A[i + 1 < 4 ? i + 1 : 0], B[j], B[j + 1 < 5 ? j + 1 : 0]));
    return best;
}

// Second-level "could the cars touch" test: world AABBs of the 8 fixtures of each car (4 hull
// polygons, 4 wheels; wheels do not collide with wheels), grown by more than the polygon radii.
// No overlapping pair => b2CollidePolygons would find no manifold point for this env.
__device__ inline bool fixtures_near(const CarSoA &s, const CarConsts &K, int64_t M, int64_t c0, int64_t c1) {
    // (fully unrolled: the 16 boxes stay in registers)
    float bb[2][8][4];
#pragma unroll
    for (int k = 0; k < 2; k++) {
        const int64_t ci = k ? c1 : c0;
#pragma unroll
        for (int f = 0; f < 8; f++) {
            const int o = f < 4 ? 0 : 6 + 6 * (f - 4);
            const float cx = s.body[(o + 0) * M + ci], cy = s.body[(o + 1) * M + ci], a = s.body[(o + 2) * M + ci];
            float sn, cs;
            crl_sincosf(a, &sn, &cs);
            const V2 lc = f < 4 ? mk(K.hull_lc[0], K.hull_lc[1]) : mk(0.f, 0.f);
            const V2 p = mk(cx, cy) - rotv(sn, cs, lc);
            const int nv = f < 4 ? K.hull_n[f] : 4;
            float x0 = 3.4e38f, y0 = 3.4e38f, x1 = -3.4e38f, y1 = -3.4e38f;
#pragma unroll
            for (int i = 0; i < 8; i++) {
                if (i < nv) {
                    const V2 v = f < 4 ? mk(K.hull_poly[f][i][0], K.hull_poly[f][i][1]) : mk(K.wheel_poly[i < 4 ? i : 0][0], K.wheel_poly[i < 4 ? i : 0][1]);
                    const V2 wv = rotv(sn, cs, v) + p;
                    x0 = fminf(x0, wv.x), y0 = fminf(y0, wv.y), x1 = fmaxf(x1, wv.x), y1 = fmaxf(y1, wv.y);
                }
            }
            bb[k][f][0] = x0 - 0.03f, bb[k][f][1] = y0 - 0.03f, bb[k][f][2] = x1 + 0.03f, bb[k][f][3] = y1 + 0.03f;
        }
    }
    bool any = false;
#pragma unroll
    for (int fa = 0; fa < 8; fa++)
#pragma unroll
        for (int fb = 0; fb < 8; fb++) {
            if (fa >= 4 && fb >= 4) continue;
            any = any || !(bb[0][fa][0] > bb[1][fb][2] || bb[1][fb][0] > bb[0][fa][2] || bb[0][fa][1] > bb[1][fb][3] ||
                           bb[1][fb][1] > bb[0][fa][3]);
        }
    return any;
}

// world.Step -> b2Island::Solve for every car that is an island of its own (one lane per car
// instance, state in registers); the cars flagged by car_step_kernel are solved together in
// car_coupled_kernel instead.
__global__ __launch_bounds__(64) void car_solve_kernel(CarSoA s, CarConsts K) {
    const int64_t M = (int64_t)s.players * s.n;
    const int64_t ci = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (ci >= M) return;
    const int car = ci >= s.n ? 1 : 0;
    const int64_t env = ci - car * s.n;
    if (s.players == 2 && s.coupled[env]) return;
    CarRegs cr;
    load_car(s, M, ci, cr);
#pragma unroll
    for (int w = 0; w < 4; w++) cr.fx[w] = s.wforce[(2 * w + 0) * M + ci], cr.fy[w] = s.wforce[(2 * w + 1) * M + ci];
    const float h = (float)(1.0 / CAR_FPS);
    const float dt_ratio = s.first_step[ci] ? 0.0f : (1.0f / h) * h;
    float slp[5];  // b2Body::m_sleepTime
#pragma unroll
    for (int b = 0; b < 5; b++) slp[b] = s.sleep[b * M + ci];
    island_solve(cr, K, h, dt_ratio, slp);
#pragma unroll
    for (int b = 0; b < 5; b++) s.sleep[b * M + ci] = slp[b];
    store_car(s, M, ci, cr);
    s.first_step[ci] = 0;
}

static constexpr int kNearCap = 32;

__global__ __launch_bounds__(64) void car_step_kernel(CarSoA s, CarConsts K, const float *__restrict__ actions,
                                                      float *__restrict__ rew_out, uint8_t *__restrict__ done_car, int sub,
                                                      int repeat) {
    const int64_t M = (int64_t)s.players * s.n;
    const int64_t ci = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (ci >= M) return;
    const int car = ci >= s.n ? 1 : 0;
    const int64_t env = ci - car * s.n;

    // ---- load
    Body H, Wb[4];
    {
        float *b = s.body + ci;
        H.cx = b[0 * M], H.cy = b[1 * M], H.a = b[2 * M], H.vx = b[3 * M], H.vy = b[4 * M], H.w = b[5 * M];
#pragma unroll
        for (int w = 0; w < 4; w++) {
            const int o = 6 + 6 * w;
            Wb[w].cx = b[(o + 0) * M], Wb[w].cy = b[(o + 1) * M], Wb[w].a = b[(o + 2) * M];
            Wb[w].vx = b[(o + 3) * M], Wb[w].vy = b[(o + 4) * M], Wb[w].w = b[(o + 5) * M];
        }
    }
    float motor_speed[4];
    double gas[4], omega[4], phase[4];
    int16_t wt[4][kWheelSlots];
#pragma unroll
    for (int w = 0; w < 4; w++) {
        motor_speed[w] = s.jspeed[w * M + ci];
        gas[w] = s.wgas[w * M + ci], omega[w] = s.womega[w * M + ci], phase[w] = s.wphase[w * M + ci];
#pragma unroll
        for (int k = 0; k < kWheelSlots; k++) wt[w][k] = s.wtiles[(w * kWheelSlots + k) * M + ci];
    }
    double reward = s.reward[ci], prev_reward = s.prev_reward[ci];
    const int visited_count = s.visited_count[ci];
    int done = s.done[ci];
    int step_count = s.step_count[ci];
    const int ntiles = s.ntiles[env];

    // ---- CarRacing.step: controls for every car, done or not (crmp:549-556)
    const float2 act = reinterpret_cast<const float2 *>(actions)[env * s.players + car];
    double a0 = fmax(fmin((double)act.x, 1.0), -1.0), a1 = fmax(fmin((double)act.y, 1.0), -1.0), a2;
    if (a1 > 0) a2 = 0;
    else a2 = a1, a1 = 0;
    const double steer_t = -a0, gas_t = fabs(a1), brake = fabs(a2);
    if (sub == 0) {  // the controls are applied once per CarRacing.step, before the action-repeat loop
        const double g = gas_t < 0 ? 0 : gas_t > 1 ? 1 : gas_t;
#pragma unroll
        for (int w = 2; w < 4; w++) {
            double diff = g - gas[w];
            if (diff > 0.1) diff = 0.1;
            gas[w] += diff;
        }
    }
    float fx[4] = {0, 0, 0, 0}, fy[4] = {0, 0, 0, 0};
    double step_reward = 0.0;
    if (!done) {
        const double dt = 1.0 / CAR_FPS;
#pragma unroll
        for (int w = 0; w < 4; w++) {  // Car.step (car_dynamics.py:159-234)
            const double steer = w < 2 ? steer_t : 0.0;
            const double ja = (double)(Wb[w].a - H.a - 0.0f);
            const double d = steer - ja;
            motor_speed[w] = (float)(sgn(d) * fmin(50.0 * fabs(d), 3.0));
            bool on_road = false;
#pragma unroll
            for (int k = 0; k < kWheelSlots; k++) on_road = on_road || wt[w][k] >= 0;
            double friction_limit = CAR_FRICTION_LIMIT * 0.6;
            if (on_road) friction_limit = fmax(friction_limit, CAR_FRICTION_LIMIT * 1.0);
            float qs, qc;
            crl_sincosf(Wb[w].a, &qs, &qc);
            const double forw0 = (double)(qc * 0.0f - qs * 1.0f), forw1 = (double)(qs * 0.0f + qc * 1.0f);
            const double side0 = (double)(qc * 1.0f - qs * 0.0f), side1 = (double)(qs * 1.0f + qc * 0.0f);
            const double vx = (double)Wb[w].vx, vy = (double)Wb[w].vy;
            const double vf = forw0 * vx + forw1 * vy, vs = side0 * vx + side1 * vy;
            double om = omega[w];
            om += dt * CAR_ENGINE_POWER * gas[w] / CAR_WHEEL_MOI / (fabs(om) + 5.0);
            if (brake >= 0.9) om = 0;
            else if (brake > 0) {
                const double dir = -sgn(om);
                double val = 15 * brake;
                if (fabs(val) > fabs(om)) val = fabs(om);
                om += dir * val;
            }
            phase[w] += om * dt;
            const double wheel_rad = 1.0 * CAR_WHEEL_R * CAR_SIZE;
            const double vr = om * wheel_rad;
            double f_force = -vf + vr, p_force = -vs;
            f_force *= 205000 * CAR_SIZE * CAR_SIZE, p_force *= 205000 * CAR_SIZE * CAR_SIZE;
            double fo = sqrt(f_force * f_force + p_force * p_force);
            if (fabs(fo) > friction_limit) {
                f_force /= fo, p_force /= fo;
                fo = friction_limit;
                f_force *= fo, p_force *= fo;
            }
            om -= dt * f_force * wheel_rad / CAR_WHEEL_MOI;
            omega[w] = om;
            fx[w] += (float)(p_force * side0 + f_force * forw0), fy[w] += (float)(p_force * side1 + f_force * forw1);
        }
        reward -= 0.1 / repeat;
        step_reward += reward - prev_reward;
        prev_reward = reward;
        float hs, hc;
        crl_sincosf(H.a, &hs, &hc);
        const V2 p = mk(H.cx, H.cy) - rotv(hs, hc, mk(K.hull_lc[0], K.hull_lc[1]));  // hull.position
        if (visited_count == ntiles) done = 1;
        if (fabs((double)p.x) > CAR_PLAYFIELD || fabs((double)p.y) > CAR_PLAYFIELD) done = 1;
        if (step_count > 1000) done = 1;
    }

    // ---- world.Step's Collide (sensor contacts -> tile rewards) does not feed this step's solve: it runs in
    // car_sensor_kernel NEXT TO the island solve, on the wheel transforms the step starts from, handed over here
    // (the solve overwrites the bodies)
#pragma unroll
    for (int w = 0; w < 4; w++) s.wsnap[(3 * w + 0) * M + ci] = Wb[w].cx, s.wsnap[(3 * w + 1) * M + ci] = Wb[w].cy, s.wsnap[(3 * w + 2) * M + ci] = Wb[w].a;

    // ---- world.Step's island solve happens in car_solve_kernel (cars on their own) or
    // car_coupled_kernel (cars whose fixtures may touch): this kernel only decides which, and hands
    // over the tyre forces and the joint motor targets.  Bodies and joint impulses are not modified
    // here, so both lanes of an env read the same pre-solve poses of both cars.
    bool coupled = false;
    if (s.players == 2 && s.contacts_enabled) {
        const int64_t c0 = env, c1 = s.n + env;
        coupled = cars_near(K, s.body[0 * M + c0], s.body[1 * M + c0], s.body[2 * M + c0], s.body[0 * M + c1], s.body[1 * M + c1],
                            s.body[2 * M + c1]);
        if (coupled) coupled = fixtures_near(s, K, M, c0, c1);
    }
#pragma unroll
    for (int w = 0; w < 4; w++) s.wforce[(2 * w + 0) * M + ci] = fx[w], s.wforce[(2 * w + 1) * M + ci] = fy[w];
    if (car == 0 && s.players == 2) {
        s.coupled[env] = coupled ? 1 : 0;
        if (coupled) s.coupled_list[atomicAdd(s.coupled_count, 1)] = (int32_t)env;
        else s.n_contact[env] = 0;
    }
    step_count += 1;

    // ---- store (wheel attributes, sensor contacts, bookkeeping; the motor targets for the solver)
#pragma unroll
    for (int w = 0; w < 4; w++) {
        s.jspeed[w * M + ci] = motor_speed[w];
        s.wgas[w * M + ci] = gas[w], s.womega[w * M + ci] = omega[w], s.wphase[w * M + ci] = phase[w];
    }
    s.reward[ci] = reward, s.prev_reward[ci] = prev_reward;
    s.done[ci] = done;
    s.step_count[ci] = step_count;
    if (rew_out) {
        // step_rewards accumulate over the repeats in f64 (crmp:584); the running sum is kept in prev_step
        const double acc = (sub == 0 ? 0.0 : s.step_acc[ci]) + step_reward;
        s.step_acc[ci] = acc;
        rew_out[env * s.players + car] = (float)acc;
    }
    if (done_car) done_car[env * s.players + car] = (uint8_t)done;
}

// world.Step's Collide for the wheel sensors (FrictionDetector, crmp:111-153): Begin / EndContact of every wheel with the
// track tiles at the transforms the step STARTS from (car_step_kernel's snapshot), tile rewards, road_visited.  Nothing in
// here feeds this step's solve, so the kernel runs beside car_solve_kernel / car_coupled_kernel on a stream of its own.
__global__ __launch_bounds__(64) void car_sensor_kernel(CarSoA s, CarConsts K) {
    __shared__ int16_t near_list[kNearCap][64];  // per lane: tiles whose AABB meets the car's (sensor broadphase)
    const int64_t M = (int64_t)s.players * s.n;
    const int64_t ci = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (ci >= M) return;
    const int car = ci >= s.n ? 1 : 0;
    const int64_t env = ci - car * s.n;
    Body Wb[4];
    int16_t wt[4][kWheelSlots];
#pragma unroll
    for (int w = 0; w < 4; w++) {
        Wb[w].cx = s.wsnap[(3 * w + 0) * M + ci], Wb[w].cy = s.wsnap[(3 * w + 1) * M + ci], Wb[w].a = s.wsnap[(3 * w + 2) * M + ci];
#pragma unroll
        for (int k = 0; k < kWheelSlots; k++) wt[w][k] = s.wtiles[(w * kWheelSlots + k) * M + ci];
    }
    double reward = s.reward[ci];
    int visited_count = s.visited_count[ci], last_block = s.last_block[ci];
    const int ntiles = s.ntiles[env];
    {
        const float R = 0.02f + 10.0f * 1.1920929e-07f;
        V2 wp[4][4];
        float wx0[4], wy0[4], wx1[4], wy1[4];
        float cx0 = 3.4e38f, cy0 = 3.4e38f, cx1 = -3.4e38f, cy1 = -3.4e38f;
#pragma unroll
        for (int w = 0; w < 4; w++) {
            float qs, qc;
            crl_sincosf(Wb[w].a, &qs, &qc);
            wx0[w] = wy0[w] = 3.4e38f, wx1[w] = wy1[w] = -3.4e38f;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                wp[w][k] = rotv(qs, qc, mk(K.wheel_poly[k][0], K.wheel_poly[k][1])) + mk(Wb[w].cx, Wb[w].cy);
                wx0[w] = fminf(wx0[w], wp[w][k].x), wy0[w] = fminf(wy0[w], wp[w][k].y);
                wx1[w] = fmaxf(wx1[w], wp[w][k].x), wy1[w] = fmaxf(wy1[w], wp[w][k].y);
            }
            cx0 = fminf(cx0, wx0[w]), cy0 = fminf(cy0, wy0[w]), cx1 = fmaxf(cx1, wx1[w]), cy1 = fmaxf(cy1, wy1[w]);
        }
        // touched tiles lie in [tmin, tmax]: two compares rule out "was touching" for most tiles
        int tmin = 1 << 30, tmax = -1;
#pragma unroll
        for (int w = 0; w < 4; w++)
#pragma unroll
            for (int k = 0; k < kWheelSlots; k++)
                if (wt[w][k] >= 0) tmin = min(tmin, (int)wt[w][k]), tmax = max(tmax, (int)wt[w][k]);
        // Broadphase over the track, then the narrow phase -- in two separate loops.  Every lane has
        // its own track and position, so the tiles near its car sit at unrelated indices; testing them
        // inside the scan would make the wavefront run the expensive body for the union of all lanes'
        // tiles (nearly every iteration).  The scan only records the tile indices (ascending, which is
        // also the order the Begin/End events are raised in) in a per-lane LDS list; the second loop
        // walks the k-th entry of every lane together.
        int cnt = 0;
#pragma unroll 1
        for (int t0 = 0; t0 < ntiles; t0 += 8) {
            float4 bbs[8];
#pragma unroll
            for (int j = 0; j < 8; j++) bbs[j] = s.tile_aabb[(int64_t)min(t0 + j, ntiles - 1) * s.n + env];
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const int t = t0 + j;
                const float4 bb = bbs[j];
                const bool near_car = !(cx0 > bb.z + 0.05f || cx1 < bb.x - 0.05f || cy0 > bb.w + 0.05f || cy1 < bb.y - 0.05f);
                bool was_any = false;
                if (t >= tmin && t <= tmax) {
#pragma unroll
                    for (int w = 0; w < 4; w++)
#pragma unroll
                        for (int k = 0; k < kWheelSlots; k++) was_any = was_any || wt[w][k] == t;
                }
                if (t < ntiles && (near_car || was_any)) {
                    if (cnt < kNearCap) near_list[cnt][threadIdx.x] = (int16_t)t;
                    cnt++;
                }
            }
        }
        auto narrow = [&](int t) {
            const float4 bb = s.tile_aabb[(int64_t)t * s.n + env];
            V2 tp[5];
#pragma unroll
            for (int k = 0; k < 5; k++)
                tp[k] = mk(s.tile_poly[((int64_t)t * 10 + 2 * k) * s.n + env], s.tile_poly[((int64_t)t * 10 + 2 * k + 1) * s.n + env]);
#pragma unroll
            for (int w = 0; w < 4; w++) {
                bool was = false;
#pragma unroll
                for (int k = 0; k < kWheelSlots; k++) was = was || wt[w][k] == t;
                bool now = false;
                if (!(wx0[w] > bb.z + 0.05f || wx1[w] < bb.x - 0.05f || wy0[w] > bb.w + 0.05f || wy1[w] < bb.y - 0.05f))
                    now = poly_dist2(wp[w], tp) < R * R;
                if (now && !was) {  // BeginContact -> FrictionDetector._contact (crmp:111-153)
                    bool placed = false;
#pragma unroll
                    for (int k = 0; k < kWheelSlots; k++)
                        if (!placed && wt[w][k] < 0) wt[w][k] = (int16_t)t, placed = true;
                    uint32_t *vw = s.visited + (int64_t)(t >> 5) * M + ci;
                    const uint32_t bit = 1u << (t & 31), cur = *vw;
                    if (!(cur & bit)) {
                        const int last_blk = last_block < 0 ? 0 : last_block;
                        if (t - last_blk < 50) {
                            last_block = t;
                            reward += 1000.0 / ntiles;
                        }
                        *vw = cur | bit;
                        visited_count += 1;
                    }
                } else if (!now && was) {  // EndContact
#pragma unroll
                    for (int k = 0; k < kWheelSlots; k++)
                        if (wt[w][k] == t) wt[w][k] = -1;
                }
            }
        };
#pragma unroll 1
        for (int k = 0; k < kNearCap; k++) {
            if (!__any(k < cnt)) break;
            if (k < cnt) narrow(near_list[k][threadIdx.x]);
        }
        if (cnt > kNearCap) {  // more near tiles than list slots (not seen in practice): finish in order
            const int last = near_list[kNearCap - 1][threadIdx.x];
#pragma unroll 1
            for (int t = last + 1; t < ntiles; t++) {
                const float4 bb = s.tile_aabb[(int64_t)t * s.n + env];
                const bool near_car = !(cx0 > bb.z + 0.05f || cx1 < bb.x - 0.05f || cy0 > bb.w + 0.05f || cy1 < bb.y - 0.05f);
                bool was_any = false;
#pragma unroll
                for (int w = 0; w < 4; w++)
#pragma unroll
                    for (int k = 0; k < kWheelSlots; k++) was_any = was_any || wt[w][k] == t;
                if (near_car || was_any) narrow(t);
            }
        }
    }

#pragma unroll
    for (int w = 0; w < 4; w++)
#pragma unroll
        for (int k = 0; k < kWheelSlots; k++) s.wtiles[(w * kWheelSlots + k) * M + ci] = wt[w][k];
    s.reward[ci] = reward;
    s.visited_count[ci] = visited_count, s.last_block[ci] = last_block;
}

// Env-level bookkeeping after the physics: gym TimeLimit (max_episode_steps = 1000,
// car_racing/register.py:15-26), then which cars end the env's episode -- any of them under
// FlattenMultiAgentObservation (atari_wrappers.py:329-330, make_car_racing_double), car 0 alone under
// CarRacingWrapper (make_competitive_car_racing.py:24-33: `d[0]`; a finished opponent just stays frozen,
// crmp:578-579).  Also captures info["num_steps"] = CarRacing.step_count (crmp:616-620) before the auto-reset.
__global__ __launch_bounds__(256) void car_post_kernel(CarSoA s, const uint8_t *__restrict__ done_car,
                                                       uint8_t *__restrict__ done_env, uint8_t *__restrict__ slow_env,
                                                       int32_t *__restrict__ info_steps, int max_episode_steps, int car0_only) {
    const int64_t env = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (env >= s.n) return;
    const int el = s.elapsed[env] + 1;
    bool d = el >= max_episode_steps;
    for (int c = 0; c < (car0_only ? 1 : s.players); c++) d = d || done_car[s.players * env + c];
    s.elapsed[env] = el;
    done_env[env] = d ? 1 : 0;
    if (info_steps) info_steps[env] = s.step_count[env];  // one world clock per env: both cars carry the same count
    // step-pipeline class: 0 = frame can be drawn now, 1 = after the coupled solve, 2 = after the reset
    if (slow_env) slow_env[env] = d ? 2 : ((s.coupled && s.coupled[env]) ? 1 : 0);
}

void launch_car_step(const CarSoA &s, const CarConsts &k, const float *actions, float *rew, uint8_t *done_car, int sub, int repeat,
                     hipStream_t st) {
    const int64_t M = (int64_t)s.players * s.n;
    if (s.coupled_count) hipMemsetAsync(s.coupled_count, 0, sizeof(int32_t), st);
    hipLaunchKernelGGL(car_step_kernel, dim3((unsigned)((M + 63) / 64)), dim3(64), 0, st, s, k, actions, rew, done_car, sub, repeat);
}

void launch_car_sensors(const CarSoA &s, const CarConsts &k, hipStream_t st) {
    const int64_t M = (int64_t)s.players * s.n;
    hipLaunchKernelGGL(car_sensor_kernel, dim3((unsigned)((M + 63) / 64)), dim3(64), 0, st, s, k);
}

void launch_car_solve(const CarSoA &s, const CarConsts &k, hipStream_t st) {
    const int64_t M = (int64_t)s.players * s.n;
    hipLaunchKernelGGL(car_solve_kernel, dim3((unsigned)((M + 63) / 64)), dim3(64), 0, st, s, k);
}

void launch_car_post(const CarSoA &s, const uint8_t *done_car, uint8_t *done_env, uint8_t *slow_env, int32_t *info_steps,
                     int max_episode_steps, bool car0_only, hipStream_t st) {
    hipLaunchKernelGGL(car_post_kernel, dim3((unsigned)((s.n + 255) / 256)), dim3(256), 0, st, s, done_car, done_env, slow_env,
                       info_steps, max_episode_steps, car0_only ? 1 : 0);
}

}  // namespace crl
