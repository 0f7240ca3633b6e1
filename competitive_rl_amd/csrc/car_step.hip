// car_step.hip -- the dynamics of one cCarRacingDouble step, split over three kernels:
//   car_step_kernel    one lane per CAR INSTANCE (2 lanes per env): controls, wheel model (f64), reward / done rules, hand-off
//                      decision (do the two cars' fixtures come near each other?), snapshot of the wheel transforms;
//   car_sensor_kernel  one lane per WHEEL: world.Step's Collide for the wheel sensors (Begin/EndContact with the track tiles ->
//                      tile rewards), on a stream of its own beside the solve -- it feeds nothing into this step's solve;
//   car_solve_kernel   one lane per car instance that is an island of its own: Box2D island solve (180 velocity iterations over 4
//                      revolute joints, <= 60 position iterations) entirely in registers.  (Cars whose fixtures may touch are
//                      solved together by car_contact.hip.)
// Nothing here is HBM-bound (~1.3 KB of state per car per step): the solve is the issue rate of one wavefront per SIMD over a
// sequential Gauss-Seidel chain, the sensors are a latency chain of small loads.
#include <stdlib.h>

#include "car_solver.h"
#include "crl_internal.h"

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
// (a routing decision like cars_near: hardware sine / cosine, and a margin of 0.04 where the contact margin is 0.02)
__device__ inline bool fixtures_near(const float *__restrict__ body, const CarConsts &K, int64_t M, int64_t c0, int64_t c1) {
    // (fully unrolled: the 16 boxes stay in registers)
    float bb[2][8][4];
#pragma unroll
    for (int k = 0; k < 2; k++) {
        const int64_t ci = k ? c1 : c0;
#pragma unroll
        for (int f = 0; f < 8; f++) {
            const int o = f < 4 ? 0 : 6 + 6 * (f - 4);
            const float cx = body[(o + 0) * M + ci], cy = body[(o + 1) * M + ci], a = body[(o + 2) * M + ci];
            float sn, cs;
            fast_sincosf(a, &sn, &cs);
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
            bb[k][f][0] = x0 - 0.04f, bb[k][f][1] = y0 - 0.04f, bb[k][f][2] = x1 + 0.04f, bb[k][f][3] = y1 + 0.04f;
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
template <bool FM>
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
    island_solve<FM>(cr, K, h, dt_ratio, slp);
#pragma unroll
    for (int b = 0; b < 5; b++) s.sleep[b * M + ci] = slp[b];
    store_car(s, M, ci, cr);
    s.first_step[ci] = 0;
}

__global__ __launch_bounds__(64) void car_step_kernel(CarSoA s, CarConsts K, const float *__restrict__ actions,
                                                      float *__restrict__ rew_out, uint8_t *__restrict__ done_car, int sub,
                                                      int repeat, int do_broad) {
    __builtin_amdgcn_s_setprio(2);  // first kernel of the step's longest chain; the collide-ahead of the step may still be running beside it
    const int64_t M = (int64_t)s.players * s.n;
    const int64_t ci = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (blockIdx.x == 0 && threadIdx.x < 16 && s.zero_next) s.zero_next[threadIdx.x] = 0;  // the NEXT step's counters (the other parity's block: nobody reads it now)
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
    // do_broad == 0: the decision (and the narrow phase) were made AHEAD, at the end of the previous step, from the same poses
    // (car_broad_kernel): this kernel then only takes the flag
    bool coupled = false;
    if (s.players == 2 && s.contacts_enabled) {
        if (do_broad) {
            const int64_t c0 = env, c1 = s.n + env;
            coupled = cars_near(K, s.body[0 * M + c0], s.body[1 * M + c0], s.body[2 * M + c0], s.body[0 * M + c1], s.body[1 * M + c1],
                                s.body[2 * M + c1]);
            if (coupled) coupled = fixtures_near(s.body, K, M, c0, c1);
        }
        // (do_broad == 0: the flag is NOT read here -- car_broad_kernel wrote it on another stream that this kernel is not
        // ordered behind; car_post_kernel, which runs behind it, destroys the contacts of the envs that are no longer coupled)
    }
#pragma unroll
    for (int w = 0; w < 4; w++) s.wforce[(2 * w + 0) * M + ci] = fx[w], s.wforce[(2 * w + 1) * M + ci] = fy[w];
    if (s.players == 2 && do_broad) {
        // one atomic per WAVEFRONT, not per coupled env: a tenth of 16 384 envs on one address took longer than the rest of the kernel
        const bool mine = car == 0 && coupled;
        const unsigned long long m = __ballot(mine);
        if (m) {
            const int lane = threadIdx.x & 63;
            int base = 0;
            if (lane == (int)__ffsll((long long)m) - 1) base = atomicAdd(s.coupled_count, (int)__popcll(m));
            base = __shfl(base, (int)__ffsll((long long)m) - 1);
            if (mine) s.coupled_list[base + (int)__popcll(m & ((1ull << lane) - 1ull))] = (int32_t)env;
        }
        if (car == 0) {
            s.coupled[env] = coupled ? 1 : 0;
            if (!coupled) s.n_contact[env] = 0;
        }
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

// The broadphase of the NEXT step, run at the end of this one (the poses a step's Collide sees are the ones the previous solve
// left: Car.step moves nothing): one lane per env, the same two tests as in car_step_kernel, the coupled flags and the compacted
// list into the next step's counter block.  car_narrow_kernel follows on the same stream -- the step's longest chain then starts
// with the touching solve instead of with two more kernels.
// fresh_body / cls (optional): an env of class 3 (finished while coupled) is collided on its STAGED bodies -- the new episode, which
// its commit is about to make current on another stream.
// touching_now / manifolds_now (optional): the kernel runs BEFORE this step's touching solve is in (beside it, so that only the
// narrow phase follows the solve): an env whose cars touch in this step -- its poses are not final yet -- is filed as coupled
// without a test (class 3: tested on its staged bodies, but filed behind the solve all the same -- see `late` below).  The flag only routes: the narrow phase decides what touches, and a coupled env in which nothing does is solved
// as two islands of their own, bit for bit like the per-car kernel.
__global__ __launch_bounds__(64) void car_broad_kernel(CarSoA s, CarConsts K, const float *__restrict__ fresh_body,
                                                       const uint8_t *__restrict__ cls, const int32_t *__restrict__ touching_now,
                                                       const int32_t *__restrict__ manifolds_now) {
    const int64_t env = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t M = 2 * s.n;
    bool coupled = false;
    if (env < s.n) {
        const int64_t c0 = env, c1 = s.n + env;
        const bool fresh = cls && cls[env] == 3;
        const float *body = fresh ? fresh_body : s.body;
        if (!fresh && touching_now && touching_now[env] && manifolds_now[env] > 0) {
            coupled = true;
        } else {
            coupled = cars_near(K, body[0 * M + c0], body[1 * M + c0], body[2 * M + c0], body[0 * M + c1], body[1 * M + c1], body[2 * M + c1]);
            if (coupled) coupled = fixtures_near(body, K, M, c0, c1);
        }
        s.coupled[env] = coupled ? 1 : 0;
    }
    // Two lists in one array (round 5): the envs whose poses are FINAL now go to the front (count [0]) -- their narrow phase runs right
    // behind this kernel, beside the touching solve --, the envs that touch in this step (filed untested above) to the back, from the end
    // downwards (count [7]): only their narrow phase is left behind the solve.
    // (round 6, ADVICE r05: an env that finished while its cars touch -- class 3, collided on its staged bodies -- goes to the back as well: the
    // narrow phase writes the env's single-buffered manifold count and manifolds, which a late pass of THIS step's touching solve still
    // reads when its lists are longer than one pass of its grid; the narrow phase picks the staged bodies by class in either phase)
    const bool late = coupled && touching_now && touching_now[env] && manifolds_now[env] > 0;
    const int lane = threadIdx.x & 63;
    const unsigned long long m = __ballot(coupled && !late), ml = __ballot(late);
    if (m) {
        int base = 0;
        if (lane == (int)__ffsll((long long)m) - 1) base = atomicAdd(s.coupled_count, (int)__popcll(m));
        base = __shfl(base, (int)__ffsll((long long)m) - 1);
        if (coupled && !late) s.coupled_list[base + (int)__popcll(m & ((1ull << lane) - 1ull))] = (int32_t)env;
    }
    if (ml) {
        int base = 0;
        if (lane == (int)__ffsll((long long)ml) - 1) base = atomicAdd(s.coupled_count + 7, (int)__popcll(ml));
        base = __shfl(base, (int)__ffsll((long long)ml) - 1);
        if (late) s.coupled_list[s.n - 1 - (base + (int)__popcll(ml & ((1ull << lane) - 1ull)))] = (int32_t)env;
    }
}

void launch_car_broad(const CarSoA &s, const CarConsts &k, hipStream_t st, const float *fresh_body, const uint8_t *cls,
                      const int32_t *touching_now, const int32_t *manifolds_now) {
    hipLaunchKernelGGL(car_broad_kernel, dim3((unsigned)((s.n + 63) / 64)), dim3(64), 0, st, s, k, fresh_body, cls, touching_now, manifolds_now);
}

// world.Step's Collide for the wheel sensors (FrictionDetector, crmp:111-153): Begin / EndContact of every wheel with the
// track tiles at the transforms the step STARTS from (car_step_kernel's snapshot), tile rewards, road_visited.  Nothing in
// here feeds this step's solve, so the kernel runs beside car_solve_kernel / car_coupled_kernel on a stream of its own.
//
// One lane per WHEEL (4 lanes = a car, 16 cars per wavefront).  (1) broadphase: the car's four lanes scan a quarter of the
// track each (ascending ranges) against all four wheel boxes and file the hits in the OWNER wheel's sub-list for that
// quarter, so an owner reads its near tiles in ascending order; (2) narrow phase: every wheel walks its own few tiles
// (polygon distance, the expensive part: one wheel per lane instead of four per lane); its BeginContacts are noted;
// (3) the car's first lane replays the BeginContacts of the four wheels in the order Box2D raises them here (tile
// ascending, wheel ascending): road_visited, the 50-tile rule, the reward.  A list that overflows sends the whole car
// through the plain serial loop (never seen in practice; CRL_CAR_SENSOR_SERIAL=1 forces it, for the tests).
static constexpr int kSubCap = 6, kBeginCap = 8;

struct SensorBooks {
    double reward;
    int visited_count, last_block;
};

// BeginContact -> FrictionDetector._contact (crmp:111-153) for one (wheel, tile) pair, bookkeeping part
__device__ inline void tile_begin(const CarSoA &s, int64_t M, int64_t ci, int ntiles, int t, SensorBooks &b) {
    uint32_t *vw = s.visited + (int64_t)(t >> 5) * M + ci;
    const uint32_t bit = 1u << (t & 31), cur = *vw;
    if (!(cur & bit)) {
        const int last_blk = b.last_block < 0 ? 0 : b.last_block;
        if (t - last_blk < 50) {
            b.last_block = t;
            b.reward += 1000.0 / ntiles;
        }
        *vw = cur | bit;
        b.visited_count += 1;
    }
}

__device__ inline void wheel_shape(const CarConsts &K, float cx, float cy, float a, V2 (&wp)[4], float4 &box) {
    float qs, qc;
    crl_sincosf(a, &qs, &qc);
    box = make_float4(3.4e38f, 3.4e38f, -3.4e38f, -3.4e38f);
#pragma unroll
    for (int k = 0; k < 4; k++) {
        wp[k] = rotv(qs, qc, mk(K.wheel_poly[k][0], K.wheel_poly[k][1])) + mk(cx, cy);
        box.x = fminf(box.x, wp[k].x), box.y = fminf(box.y, wp[k].y), box.z = fmaxf(box.z, wp[k].x), box.w = fmaxf(box.w, wp[k].y);
    }
}

__device__ inline bool box_near(const float4 &wb, const float4 &bb) {  // the wheel's box against a tile's, grown by 0.05
    return !(wb.x > bb.z + 0.05f || wb.z < bb.x - 0.05f || wb.y > bb.w + 0.05f || wb.w < bb.y - 0.05f);
}

__device__ inline void load_tile_poly(const CarSoA &s, int64_t env, int t, V2 (&tp)[5]) {
#pragma unroll
    for (int k = 0; k < 5; k++)
        tp[k] = mk(s.tile_poly[((int64_t)t * 10 + 2 * k) * s.n + env], s.tile_poly[((int64_t)t * 10 + 2 * k + 1) * s.n + env]);
}

// Overflow path: the whole car in one lane, tile by tile.  A kernel of its own (launched behind car_sensor_kernel, exits at
// once for every car that did not overflow) so that its run-time-indexed arrays cost the main kernel neither scratch
// memory nor registers.
__global__ __launch_bounds__(64) void car_sensor_serial_kernel(CarSoA s, CarConsts K) {
    const int64_t M = (int64_t)s.players * s.n;
    const int64_t ci = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (ci >= M || !s.sensor_ovf[ci]) return;
    const int car = ci >= s.n ? 1 : 0;
    const int64_t env = ci - car * s.n;
    const int ntiles = s.ntiles[env];
    SensorBooks b;
    b.reward = s.reward[ci], b.visited_count = s.visited_count[ci], b.last_block = s.last_block[ci];
    const float R = 0.02f + 10.0f * 1.1920929e-07f;
    V2 wp[4][4];
    float4 wb[4];
    int16_t wt[4][kWheelSlots];
    for (int w = 0; w < 4; w++) {
        wheel_shape(K, s.wsnap[(3 * w + 0) * M + ci], s.wsnap[(3 * w + 1) * M + ci], s.wsnap[(3 * w + 2) * M + ci], wp[w], wb[w]);
        for (int k = 0; k < kWheelSlots; k++) wt[w][k] = s.wtiles[(w * kWheelSlots + k) * M + ci];
    }
#pragma unroll 1
    for (int t = 0; t < ntiles; t++) {
        const float4 bb = s.tile_aabb[(int64_t)t * s.n + env];
        bool any = false;
        for (int w = 0; w < 4; w++) {
            any = any || box_near(wb[w], bb);
            for (int k = 0; k < kWheelSlots; k++) any = any || wt[w][k] == t;
        }
        if (!any) continue;
        V2 tp[5];
        load_tile_poly(s, env, t, tp);
        for (int w = 0; w < 4; w++) {
            bool was = false;
            for (int k = 0; k < kWheelSlots; k++) was = was || wt[w][k] == t;
            bool now = false;
            if (box_near(wb[w], bb)) now = poly_dist2(wp[w], tp) < R * R;
            if (now && !was) {
                bool placed = false;
                for (int k = 0; k < kWheelSlots; k++)
                    if (!placed && wt[w][k] < 0) wt[w][k] = (int16_t)t, placed = true;
                if (!placed) atomicAdd(s.cap_hits, 1);
                tile_begin(s, M, ci, ntiles, t, b);
            } else if (!now && was) {
                for (int k = 0; k < kWheelSlots; k++)
                    if (wt[w][k] == t) wt[w][k] = -1;
            }
        }
    }
    for (int w = 0; w < 4; w++)
        for (int k = 0; k < kWheelSlots; k++) s.wtiles[(w * kWheelSlots + k) * M + ci] = wt[w][k];
    s.reward[ci] = b.reward;
    s.visited_count[ci] = b.visited_count, s.last_block[ci] = b.last_block;
}

// (one wavefront per workgroup and at most 128 registers: the kernel has to fit on SIMDs next to the register-heavy solve)
__global__ __launch_bounds__(64, 4) void car_sensor_kernel(CarSoA s, CarConsts K, int force_serial_arg) {
    const int force_serial = (force_serial_arg & 1) | CRL_ABL(force_serial_arg & 14);  // bit 0: every car through the serial kernel (same results)
    __shared__ int16_t sub[kSubCap][4][64];  // [entry][quarter of the track][owner lane]
    __shared__ uint8_t subcnt[4][64];
    __shared__ int16_t beg[kBeginCap][64];   // an owner's BeginContacts of this step, ascending
    __shared__ uint8_t begcnt[64], ovf_l[64];
    const int tid = threadIdx.x, w = tid & 3, q0 = tid & ~3;
    const int64_t M = (int64_t)s.players * s.n;
    const int64_t ci_raw = ((int64_t)blockIdx.x * 64 + tid) >> 2;
    const bool valid = ci_raw < M;
    const int64_t ci = valid ? ci_raw : M - 1;
    const int car = ci >= s.n ? 1 : 0;
    const int64_t env = ci - car * s.n;
    const float R = 0.02f + 10.0f * 1.1920929e-07f;
    const int ntiles = s.ntiles[env];

    // ---- this lane's wheel: polygon, box, the tiles it touches (and whether each of those is still near)
    V2 wp[4];
    float4 wb;
    wheel_shape(K, s.wsnap[(3 * w + 0) * M + ci], s.wsnap[(3 * w + 1) * M + ci], s.wsnap[(3 * w + 2) * M + ci], wp, wb);
    int wt[kWheelSlots];
    bool slot_near[kWheelSlots];
#pragma unroll
    for (int k = 0; k < kWheelSlots; k++) wt[k] = s.wtiles[(w * kWheelSlots + k) * M + ci];
#pragma unroll
    for (int k = 0; k < kWheelSlots; k++) {
        const float4 bb = s.tile_aabb[(int64_t)max(wt[k], 0) * s.n + env];
        slot_near[k] = wt[k] >= 0 && wt[k] < ntiles && box_near(wb, bb);
    }

    // ---- (1) broadphase: quarter w of the track against the car's four wheel boxes
    float4 qb[4];
#pragma unroll
    for (int o = 0; o < 4; o++)
        qb[o] = make_float4(__shfl(wb.x, q0 + o, 64), __shfl(wb.y, q0 + o, 64), __shfl(wb.z, q0 + o, 64), __shfl(wb.w, q0 + o, 64));
    const int per = (ntiles + 3) >> 2, tbeg = w * per, tend = min(tbeg + per, ntiles);
    int c4[4] = {0, 0, 0, 0};
    bool ovf = (force_serial & 1) != 0;  // bits 2, 4, 8: timing ablations (wrong results): no polygon distance / no polygon loads / no scan
    // two passes: which 8-tile blocks of the quarter come near any of the car's wheel boxes (one united box per block: 16 bytes
    // instead of 128), then only those blocks' tiles, in ascending order as before.  (A wavefront holds 16 cars at 16 different
    // places: testing every tile box cost 4.8 KB per car and step and was the largest fetch of the whole step.)
    uint32_t hit = 0;
    const int b0 = tbeg >> 3, nblk = tend > tbeg ? ((tend - 1) >> 3) - b0 + 1 : 0;  // <= 17 for 512 tiles
    if (!(force_serial & 8)) {
#pragma unroll 1
        for (int k0 = 0; k0 < nblk; k0 += 6) {
            float4 B[6];
#pragma unroll
            for (int j = 0; j < 6; j++) B[j] = s.tile_blk[(int64_t)(b0 + min(k0 + j, nblk - 1)) * s.n + env];
#pragma unroll
            for (int j = 0; j < 6; j++) {
                const bool any = box_near(qb[0], B[j]) || box_near(qb[1], B[j]) || box_near(qb[2], B[j]) || box_near(qb[3], B[j]);
                if (any && k0 + j < nblk) hit |= 1u << (k0 + j);
            }
        }
    }
#pragma unroll 1
    while (__any(hit != 0u)) {
        const bool on = hit != 0u;
        const int kb = on ? __builtin_ctz(hit) : 0;
        hit &= hit - 1u;
        const int base = (b0 + kb) * 8;
        float4 bbs[8];
#pragma unroll
        for (int j = 0; j < 8; j++) bbs[j] = s.tile_aabb[(int64_t)min(max(base + j, tbeg), max(tend - 1, 0)) * s.n + env];
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const int t = base + j;
            if (!on || t < tbeg || t >= tend) continue;
#pragma unroll
            for (int o = 0; o < 4; o++) {
                if (box_near(qb[o], bbs[j])) {
                    if (c4[o] < kSubCap) sub[c4[o]][w][q0 + o] = (int16_t)t;
                    else ovf = true;
                    c4[o]++;
                }
            }
        }
    }
#pragma unroll
    for (int o = 0; o < 4; o++) subcnt[w][q0 + o] = (uint8_t)min(c4[o], kSubCap);
    __syncthreads();

    // ---- (2) narrow phase over this wheel's near tiles, ascending; tiles it touched that are no longer near end as
    // their index is passed (the order only matters for which slot a BeginContact finds free)
    const int n0 = subcnt[0][tid], n1 = n0 + subcnt[1][tid], n2 = n1 + subcnt[2][tid], n3 = n2 + subcnt[3][tid];
    int nb = 0;
#pragma unroll 1
    for (int f = 0; __any(f < n3); f++) {
        if (f >= n3) continue;
        const int q = (f >= n0) + (f >= n1) + (f >= n2);
        const int t = sub[f - (q == 0 ? 0 : q == 1 ? n0 : q == 2 ? n1 : n2)][q][tid];
        V2 tp[5];
        if (!(force_serial & 4)) load_tile_poly(s, env, t, tp);
        else
            for (int k = 0; k < 5; k++) tp[k] = mk((float)(t + k), (float)k * wb.x);
        bool was = false;
#pragma unroll
        for (int k = 0; k < kWheelSlots; k++) {
            if (wt[k] >= 0 && wt[k] < t && !slot_near[k]) wt[k] = -1;  // EndContact of a tile left behind
            was = was || wt[k] == t;
        }
        const bool now = (force_serial & 2) ? tp[0].x < wb.x : poly_dist2(wp, tp) < R * R;
        if (now && !was) {
            bool placed = false;
#pragma unroll
            for (int k = 0; k < kWheelSlots; k++)
                if (!placed && wt[k] < 0) wt[k] = t, slot_near[k] = true, placed = true;
            if (!placed) atomicAdd(s.cap_hits, 1);  // (never on a real track: a 0.6 x 1.1 wheel overlaps two or three 3.5-long tiles)
            if (nb < kBeginCap) beg[nb][tid] = (int16_t)t;
            else ovf = true;
            nb++;
        } else if (!now && was) {
#pragma unroll
            for (int k = 0; k < kWheelSlots; k++)
                if (wt[k] == t) wt[k] = -1;
        }
    }
#pragma unroll
    for (int k = 0; k < kWheelSlots; k++)
        if (wt[k] >= 0 && !slot_near[k]) wt[k] = -1;
    begcnt[tid] = (uint8_t)min(nb, kBeginCap);
    ovf_l[tid] = ovf ? 1 : 0;
    __syncthreads();

    // ---- (3) per car: the BeginContacts of its wheels in Box2D's order here (tile, then wheel)
    const bool car_ovf = ovf_l[q0] | ovf_l[q0 + 1] | ovf_l[q0 + 2] | ovf_l[q0 + 3];
    if (valid && !car_ovf) {
#pragma unroll
        for (int k = 0; k < kWheelSlots; k++) s.wtiles[(w * kWheelSlots + k) * M + ci] = (int16_t)wt[k];
    }
    if (valid && w == 0) s.sensor_ovf[ci] = car_ovf ? 1 : 0;  // car_sensor_serial_kernel redoes the car from scratch
    if (valid && w == 0 && !car_ovf) {
        SensorBooks b;
        b.reward = s.reward[ci], b.visited_count = s.visited_count[ci], b.last_block = s.last_block[ci];
        {
            int ix[4] = {0, 0, 0, 0};
            const int cn[4] = {begcnt[q0], begcnt[q0 + 1], begcnt[q0 + 2], begcnt[q0 + 3]};
#pragma unroll 1
            for (int it = 0; it < 4 * kBeginCap; it++) {
                int best = 1 << 30, bo = -1;
#pragma unroll
                for (int o = 0; o < 4; o++) {
                    const int t = ix[o] < cn[o] ? (int)beg[min(ix[o], kBeginCap - 1)][q0 + o] : (1 << 30);
                    if (t < best) best = t, bo = o;
                }
                if (bo < 0) break;
#pragma unroll
                for (int o = 0; o < 4; o++)
                    if (o == bo) ix[o]++;
                tile_begin(s, M, ci, ntiles, best, b);
            }
        }
        s.reward[ci] = b.reward;
        s.visited_count[ci] = b.visited_count, s.last_block[ci] = b.last_block;
    }
}

// Env-level bookkeeping after the physics: gym TimeLimit (max_episode_steps = 1000,
// car_racing/register.py:15-26), then which cars end the env's episode -- any of them under
// FlattenMultiAgentObservation (atari_wrappers.py:329-330, make_car_racing_double), car 0 alone under
// CarRacingWrapper (make_competitive_car_racing.py:24-33: `d[0]`; a finished opponent just stays frozen,
// crmp:578-579).  Also captures info["num_steps"] = CarRacing.step_count (crmp:616-620) before the auto-reset.
__global__ __launch_bounds__(256) void car_post_kernel(CarSoA s, const uint8_t *__restrict__ done_car,
                                                       uint8_t *__restrict__ done_env, uint8_t *__restrict__ done_out, uint8_t *__restrict__ slow_env,
                                                       int32_t *__restrict__ info_steps, int32_t *__restrict__ info_elapsed, int max_episode_steps, int car0_only,
                                                       int32_t *__restrict__ class_list, int32_t *__restrict__ class_count) {
    const int64_t env = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int cls = 0;
    if (env < s.n) {
        const int el = s.elapsed[env] + 1;
        bool d = el >= max_episode_steps;
        for (int c = 0; c < (car0_only ? 1 : s.players); c++) d = d || done_car[s.players * env + c];
        s.elapsed[env] = el;
        done_env[env] = d ? 1 : 0;
        if (done_out) done_out[env] = d ? 1 : 0;  // the caller's done tensor (was a device-to-device copy at the end of the step)
        if (info_elapsed) info_elapsed[env] = el;  // gym TimeLimit._elapsed_steps after this step, before the auto-reset zeroes it
        if (info_steps) info_steps[env] = s.step_count[env];  // one world clock per env: both cars carry the same count
        // step-pipeline class: 0 = frame can be drawn now, 1 = after the coupled solve, 2 = finished (terminal frame, reset, first
        // frame), 3 = finished and coupled (the same, after the coupled solve)
        const bool cp = s.coupled && s.coupled[env];
        // Box2D destroys the contacts of fixtures whose boxes no longer overlap (b2ContactManager::Collide).  Here rather than in
        // car_step_kernel: with the collide-ahead the flag was written by car_broad_kernel on THIS kernel's stream (side2), which the
        // caller's stream is not ordered behind; nothing reads n_contact of a non-coupled env before the step's join.
        if (s.coupled && s.n_contact && !cp) s.n_contact[env] = 0;
        cls = d ? (cp ? 3 : 2) : (cp ? 1 : 0);
        if (slow_env) slow_env[env] = (uint8_t)cls;
    }
    // the two small classes also as compacted lists (any order): their frames are drawn by launches sized to the lists.
    // One atomic per wavefront and class, not per env (thousands of them on two addresses took 20 us).
    if (class_list) {
        const int lane = threadIdx.x & 63;
#pragma unroll
        for (int k = 2; k <= 4; k++) {  // (class 1, the coupled envs, is listed by the narrow phase: near-only and touching); "4" = every finished env
            const unsigned long long m = __ballot(k == 4 ? cls >= 2 : cls == k);
            if (!m) continue;
            int base = 0;
            if (lane == 0) base = atomicAdd(&class_count[k - 1], (int)__popcll(m));
            base = __shfl(base, 0);
            if (k == 4 ? cls >= 2 : cls == k) class_list[(int64_t)(k - 1) * s.n + base + (int)__popcll(m & ((1ull << lane) - 1ull))] = (int32_t)env;
        }
    }
}

void launch_car_step(const CarSoA &s, const CarConsts &k, const float *actions, float *rew, uint8_t *done_car, int sub, int repeat,
                     hipStream_t st, bool do_broad) {
    const int64_t M = (int64_t)s.players * s.n;
    hipLaunchKernelGGL(car_step_kernel, dim3((unsigned)((M + 63) / 64)), dim3(64), 0, st, s, k, actions, rew, done_car, sub, repeat, do_broad ? 1 : 0);
}

void launch_car_sensors(const CarSoA &s, const CarConsts &k, hipStream_t st) {
    const int64_t M = (int64_t)s.players * s.n;
    static const int force_serial = getenv("CRL_CAR_SENSOR_SERIAL") ? atoi(getenv("CRL_CAR_SENSOR_SERIAL")) : 0;
    hipLaunchKernelGGL(car_sensor_kernel, dim3((unsigned)((4 * M + 63) / 64)), dim3(64), 0, st, s, k, force_serial);
    hipLaunchKernelGGL(car_sensor_serial_kernel, dim3((unsigned)((M + 63) / 64)), dim3(64), 0, st, s, k);
}

void launch_car_solve(const CarSoA &s, const CarConsts &k, hipStream_t st) {
    const int64_t M = (int64_t)s.players * s.n;
    if (s.fma) hipLaunchKernelGGL(car_solve_kernel<true>, dim3((unsigned)((M + 63) / 64)), dim3(64), 0, st, s, k);
    else hipLaunchKernelGGL(car_solve_kernel<false>, dim3((unsigned)((M + 63) / 64)), dim3(64), 0, st, s, k);
}

void launch_car_post(const CarSoA &s, const uint8_t *done_car, uint8_t *done_env, uint8_t *done_out, uint8_t *slow_env, int32_t *info_steps,
                     int32_t *info_elapsed, int max_episode_steps, bool car0_only, hipStream_t st, int32_t *class_list, int32_t *class_count) {
    hipLaunchKernelGGL(car_post_kernel, dim3((unsigned)((s.n + 255) / 256)), dim3(256), 0, st, s, done_car, done_env, done_out, slow_env,
                       info_steps, info_elapsed, max_episode_steps, car0_only ? 1 : 0, class_list, class_count);
}

}  // namespace crl
