// car_step.hip -- one cCarRacingDouble step, one lane per CAR INSTANCE (2 lanes per env).
//
// The two cars of an env only interact through contacts, which this build does not model,
// so each car is an independent Box2D island: wheel model (f64) -> sensor overlap with the
// track tiles (Begin/EndContact -> tile rewards) -> island solve (180 velocity iterations over
// 4 revolute joints, <= 60 position iterations) entirely in registers.  Bound by the sequential
// Gauss-Seidel chain (VALU latency), not by HBM: ~1.3 KB of state per car per step.
#include "car_device.h"

namespace crl {

__device__ inline double sgn(double v) { return (double)((v > 0) - (v < 0)); }

#define LINEAR_SLOP 0.005f
#define ANGULAR_SLOP (2.0f / 180.0f * 3.14159265359f)
#define MAX_ANGULAR_CORRECTION (8.0f / 180.0f * 3.14159265359f)
#define MAX_TRANSLATION 2.0f
#define MAX_ROTATION (0.5f * 3.14159265359f)
#define LOWER_ANGLE (-0.4f)
#define UPPER_ANGLE (+0.4f)
#define MAX_MOTOR_TORQUE ((float)(180 * 900 * CAR_SIZE * CAR_SIZE))
enum { LIM_INACTIVE = 0, LIM_LOWER = 1, LIM_UPPER = 2 };

struct M33 {  // columns ex, ey, ez as in b2Mat33
    float ex[3], ey[3], ez[3];
};

__device__ inline V2 solve22(const M33 &m, V2 b) {
    const float a11 = m.ex[0], a12 = m.ey[0], a21 = m.ex[1], a22 = m.ey[1];
    float det = a11 * a22 - a12 * a21;
    if (det != 0.0f) det = 1.0f / det;
    return mk(det * (a22 * b.x - a12 * b.y), det * (a11 * b.y - a21 * b.x));
}

__device__ inline void solve33(const M33 &m, const float b[3], float x[3]) {
    const float *ex = m.ex, *ey = m.ey, *ez = m.ez;
    const float cyz[3] = {ey[1] * ez[2] - ey[2] * ez[1], ey[2] * ez[0] - ey[0] * ez[2], ey[0] * ez[1] - ey[1] * ez[0]};
    float det = ex[0] * cyz[0] + ex[1] * cyz[1] + ex[2] * cyz[2];
    if (det != 0.0f) det = 1.0f / det;
    const float cbz[3] = {b[1] * ez[2] - b[2] * ez[1], b[2] * ez[0] - b[0] * ez[2], b[0] * ez[1] - b[1] * ez[0]};
    const float cyb[3] = {ey[1] * b[2] - ey[2] * b[1], ey[2] * b[0] - ey[0] * b[2], ey[0] * b[1] - ey[1] * b[0]};
    x[0] = det * (b[0] * cyz[0] + b[1] * cyz[1] + b[2] * cyz[2]);
    x[1] = det * (ex[0] * cbz[0] + ex[1] * cbz[1] + ex[2] * cbz[2]);
    x[2] = det * (ex[0] * cyb[0] + ex[1] * cyb[1] + ex[2] * cyb[2]);
}

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

__global__ __launch_bounds__(64) void car_step_kernel(CarSoA s, CarConsts K, const float *__restrict__ actions,
                                                      float *__restrict__ rew_out, uint8_t *__restrict__ done_car) {
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
    float imp[4][3], motor_imp[4], motor_speed[4];
    int lim[4];
    double gas[4], omega[4], phase[4];
    int16_t wt[4][kWheelSlots];
#pragma unroll
    for (int w = 0; w < 4; w++) {
        imp[w][0] = s.jimp[(3 * w + 0) * M + ci], imp[w][1] = s.jimp[(3 * w + 1) * M + ci], imp[w][2] = s.jimp[(3 * w + 2) * M + ci];
        motor_imp[w] = s.jmotor[w * M + ci], motor_speed[w] = s.jspeed[w * M + ci], lim[w] = s.jlimit[w * M + ci];
        gas[w] = s.wgas[w * M + ci], omega[w] = s.womega[w * M + ci], phase[w] = s.wphase[w * M + ci];
#pragma unroll
        for (int k = 0; k < kWheelSlots; k++) wt[w][k] = s.wtiles[(w * kWheelSlots + k) * M + ci];
    }
    double reward = s.reward[ci], prev_reward = s.prev_reward[ci];
    int visited_count = s.visited_count[ci], last_block = s.last_block[ci], done = s.done[ci];
    int step_count = s.step_count[ci];
    const int first_step = s.first_step[ci];
    const int ntiles = s.ntiles[env];

    // ---- CarRacing.step: controls for every car, done or not (crmp:549-556)
    const float2 act = reinterpret_cast<const float2 *>(actions)[env * s.players + car];
    double a0 = fmax(fmin((double)act.x, 1.0), -1.0), a1 = fmax(fmin((double)act.y, 1.0), -1.0), a2;
    if (a1 > 0) a2 = 0;
    else a2 = a1, a1 = 0;
    const double steer_t = -a0, gas_t = fabs(a1), brake = fabs(a2);
    {
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
            const float qs = sinf(Wb[w].a), qc = cosf(Wb[w].a);
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
        reward -= 0.1 / 1;
        step_reward += reward - prev_reward;
        prev_reward = reward;
        const float hs = sinf(H.a), hc = cosf(H.a);
        const V2 p = mk(H.cx, H.cy) - rotv(hs, hc, mk(K.hull_lc[0], K.hull_lc[1]));  // hull.position
        if (visited_count == ntiles) done = 1;
        if (fabs((double)p.x) > CAR_PLAYFIELD || fabs((double)p.y) > CAR_PLAYFIELD) done = 1;
        if (step_count > 1000) done = 1;
    }

    // ---- world.Step: Collide (sensor contacts at the transforms the step starts from)
    {
        const float R = 0.02f + 10.0f * 1.1920929e-07f;
        V2 wp[4][4];
        float wx0[4], wy0[4], wx1[4], wy1[4];
        float cx0 = 3.4e38f, cy0 = 3.4e38f, cx1 = -3.4e38f, cy1 = -3.4e38f;
#pragma unroll
        for (int w = 0; w < 4; w++) {
            const float qs = sinf(Wb[w].a), qc = cosf(Wb[w].a);
            wx0[w] = wy0[w] = 3.4e38f, wx1[w] = wy1[w] = -3.4e38f;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                wp[w][k] = rotv(qs, qc, mk(K.wheel_poly[k][0], K.wheel_poly[k][1])) + mk(Wb[w].cx, Wb[w].cy);
                wx0[w] = fminf(wx0[w], wp[w][k].x), wy0[w] = fminf(wy0[w], wp[w][k].y);
                wx1[w] = fmaxf(wx1[w], wp[w][k].x), wy1[w] = fmaxf(wy1[w], wp[w][k].y);
            }
            cx0 = fminf(cx0, wx0[w]), cy0 = fminf(cy0, wy0[w]), cx1 = fmaxf(cx1, wx1[w]), cy1 = fmaxf(cy1, wy1[w]);
        }
#pragma unroll 1
        for (int t = 0; t < ntiles; t++) {
            const float4 bb = s.tile_aabb[(int64_t)t * s.n + env];
            const bool near_car = !(cx0 > bb.z + 0.05f || cx1 < bb.x - 0.05f || cy0 > bb.w + 0.05f || cy1 < bb.y - 0.05f);
            bool was_any = false;
#pragma unroll
            for (int w = 0; w < 4; w++)
#pragma unroll
                for (int k = 0; k < kWheelSlots; k++) was_any = was_any || wt[w][k] == t;
            if (!near_car && !was_any) continue;
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
        }
    }

    // ---- world.Step: b2Island::Solve for this car (joints in island order j3, j2, j1, j0)
    {
        const float h = (float)(1.0 / CAR_FPS);
        const float dt_ratio = first_step ? 0.0f : (1.0f / h) * h;
        const float mA = K.hull_inv_mass, iA = K.hull_inv_I, mB = K.wheel_inv_mass, iB = K.wheel_inv_I;
        const V2 lcA = mk(K.hull_lc[0], K.hull_lc[1]);
        // integrate velocities (hull has no applied force; wheels carry the tyre forces)
#pragma unroll
        for (int w = 0; w < 4; w++) Wb[w].vx += h * (mB * fx[w]), Wb[w].vy += h * (mB * fy[w]);
        V2 rA[4];
        M33 mass[4];
        float motorMass = iA + iB;
        if (motorMass > 0.0f) motorMass = 1.0f / motorMass;
        // InitVelocityConstraints + warm start
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int w = 3 - q;
            const float sA = sinf(H.a), cA = cosf(H.a);
            rA[w] = rotv(sA, cA, mk(K.anchor[w][0], K.anchor[w][1]) - lcA);
            const V2 r = rA[w];
            M33 &m = mass[w];
            m.ex[0] = mA + mB + r.y * r.y * iA + 0.0f * 0.0f * iB;
            m.ey[0] = -r.y * r.x * iA - 0.0f * 0.0f * iB;
            m.ez[0] = -r.y * iA - 0.0f * iB;
            m.ex[1] = m.ey[0];
            m.ey[1] = mA + mB + r.x * r.x * iA + 0.0f * 0.0f * iB;
            m.ez[1] = r.x * iA + 0.0f * iB;
            m.ex[2] = m.ez[0], m.ey[2] = m.ez[1], m.ez[2] = iA + iB;
            const float ja = Wb[w].a - H.a - 0.0f;
            if (ja <= LOWER_ANGLE) {
                if (lim[w] != LIM_LOWER) imp[w][2] = 0;
                lim[w] = LIM_LOWER;
            } else if (ja >= UPPER_ANGLE) {
                if (lim[w] != LIM_UPPER) imp[w][2] = 0;
                lim[w] = LIM_UPPER;
            } else {
                lim[w] = LIM_INACTIVE, imp[w][2] = 0;
            }
            imp[w][0] *= dt_ratio, imp[w][1] *= dt_ratio, imp[w][2] *= dt_ratio, motor_imp[w] *= dt_ratio;
            const V2 P = mk(imp[w][0], imp[w][1]);
            H.vx -= mA * P.x, H.vy -= mA * P.y;
            H.w -= iA * (cross(r, P) + motor_imp[w] + imp[w][2]);
            Wb[w].vx += mB * P.x, Wb[w].vy += mB * P.y;
            Wb[w].w += iB * (cross(mk(0.f, 0.f), P) + motor_imp[w] + imp[w][2]);
        }
        // velocity iterations
#pragma unroll 1
        for (int it = 0; it < 180; it++) {
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const int w = 3 - q;
                const V2 r = rA[w], rB = mk(0.f, 0.f);
                {  // motor
                    const float Cdot = Wb[w].w - H.w - motor_speed[w];
                    float impulse = -motorMass * Cdot;
                    const float old = motor_imp[w], maxI = h * MAX_MOTOR_TORQUE;
                    float ni = old + impulse;
                    ni = ni < -maxI ? -maxI : ni > maxI ? maxI : ni;
                    motor_imp[w] = ni;
                    impulse = ni - old;
                    H.w -= iA * impulse, Wb[w].w += iB * impulse;
                }
                const V2 vA = mk(H.vx, H.vy), vB = mk(Wb[w].vx, Wb[w].vy);
                if (lim[w] != LIM_INACTIVE) {
                    const V2 Cdot1 = ((vB + scross(Wb[w].w, rB)) - vA) - scross(H.w, r);
                    const float Cdot2 = Wb[w].w - H.w;
                    const float b[3] = {Cdot1.x, Cdot1.y, Cdot2};
                    float im[3];
                    solve33(mass[w], b, im);
                    im[0] = -im[0], im[1] = -im[1], im[2] = -im[2];
                    const float newI = imp[w][2] + im[2];
                    const bool lower = lim[w] == LIM_LOWER;
                    if (lower ? newI < 0.0f : newI > 0.0f) {
                        const V2 rhs = (-1.0f * Cdot1) + imp[w][2] * mk(mass[w].ez[0], mass[w].ez[1]);
                        const V2 red = solve22(mass[w], rhs);
                        im[0] = red.x, im[1] = red.y, im[2] = -imp[w][2];
                        imp[w][0] += red.x, imp[w][1] += red.y, imp[w][2] = 0;
                    } else {
                        imp[w][0] += im[0], imp[w][1] += im[1], imp[w][2] += im[2];
                    }
                    const V2 P = mk(im[0], im[1]);
                    H.vx -= mA * P.x, H.vy -= mA * P.y, H.w -= iA * (cross(r, P) + im[2]);
                    Wb[w].vx += mB * P.x, Wb[w].vy += mB * P.y, Wb[w].w += iB * (cross(rB, P) + im[2]);
                } else {
                    const V2 Cdot = ((vB + scross(Wb[w].w, rB)) - vA) - scross(H.w, r);
                    const V2 im = solve22(mass[w], -1.0f * Cdot);
                    imp[w][0] += im.x, imp[w][1] += im.y;
                    H.vx -= mA * im.x, H.vy -= mA * im.y, H.w -= iA * cross(r, im);
                    Wb[w].vx += mB * im.x, Wb[w].vy += mB * im.y, Wb[w].w += iB * cross(rB, im);
                }
            }
        }
        // integrate positions
        auto integrate = [&](Body &b) {
            const V2 tr = mk(h * b.vx, h * b.vy);
            if (dot(tr, tr) > MAX_TRANSLATION * MAX_TRANSLATION) {
                const float ratio = MAX_TRANSLATION / sqrtf(dot(tr, tr));
                b.vx *= ratio, b.vy *= ratio;
            }
            const float ro = h * b.w;
            if (ro * ro > MAX_ROTATION * MAX_ROTATION) b.w *= MAX_ROTATION / fabsf(ro);
            b.cx += h * b.vx, b.cy += h * b.vy, b.a += h * b.w;
        };
        integrate(H);
#pragma unroll
        for (int w = 0; w < 4; w++) integrate(Wb[w]);
        // position iterations
#pragma unroll 1
        for (int it = 0; it < 60; it++) {
            bool ok = true;
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const int w = 3 - q;
                float angErr = 0;
                if (lim[w] != LIM_INACTIVE) {
                    const float angle = Wb[w].a - H.a - 0.0f;
                    float C;
                    if (lim[w] == LIM_LOWER) {
                        C = angle - LOWER_ANGLE, angErr = -C;
                        C = fminf(fmaxf(C + ANGULAR_SLOP, -MAX_ANGULAR_CORRECTION), 0.0f);
                    } else {
                        C = angle - UPPER_ANGLE, angErr = C;
                        C = fminf(fmaxf(C - ANGULAR_SLOP, 0.0f), MAX_ANGULAR_CORRECTION);
                    }
                    const float li = -motorMass * C;
                    H.a -= iA * li, Wb[w].a += iB * li;
                }
                const float sA = sinf(H.a), cA = cosf(H.a);
                const V2 r = rotv(sA, cA, mk(K.anchor[w][0], K.anchor[w][1]) - lcA), rB = mk(0.f, 0.f);
                const V2 C = ((mk(Wb[w].cx, Wb[w].cy) + rB) - mk(H.cx, H.cy)) - r;
                const float posErr = sqrtf(dot(C, C));
                M33 k;
                k.ex[0] = mA + mB + iA * r.y * r.y + iB * rB.y * rB.y;
                k.ex[1] = -iA * r.x * r.y - iB * rB.x * rB.y;
                k.ey[0] = k.ex[1];
                k.ey[1] = mA + mB + iA * r.x * r.x + iB * rB.x * rB.x;
                const V2 im = -1.0f * solve22(k, C);
                H.cx -= mA * im.x, H.cy -= mA * im.y, H.a -= iA * cross(r, im);
                Wb[w].cx += mB * im.x, Wb[w].cy += mB * im.y, Wb[w].a += iB * cross(rB, im);
                ok = ok && posErr <= LINEAR_SLOP && angErr <= ANGULAR_SLOP;
            }
            if (ok) break;
        }
    }
    step_count += 1;

    // ---- store
    {
        float *b = s.body + ci;
        b[0 * M] = H.cx, b[1 * M] = H.cy, b[2 * M] = H.a, b[3 * M] = H.vx, b[4 * M] = H.vy, b[5 * M] = H.w;
#pragma unroll
        for (int w = 0; w < 4; w++) {
            const int o = 6 + 6 * w;
            b[(o + 0) * M] = Wb[w].cx, b[(o + 1) * M] = Wb[w].cy, b[(o + 2) * M] = Wb[w].a;
            b[(o + 3) * M] = Wb[w].vx, b[(o + 4) * M] = Wb[w].vy, b[(o + 5) * M] = Wb[w].w;
        }
    }
#pragma unroll
    for (int w = 0; w < 4; w++) {
        s.jimp[(3 * w + 0) * M + ci] = imp[w][0], s.jimp[(3 * w + 1) * M + ci] = imp[w][1], s.jimp[(3 * w + 2) * M + ci] = imp[w][2];
        s.jmotor[w * M + ci] = motor_imp[w], s.jspeed[w * M + ci] = motor_speed[w], s.jlimit[w * M + ci] = lim[w];
        s.wgas[w * M + ci] = gas[w], s.womega[w * M + ci] = omega[w], s.wphase[w * M + ci] = phase[w];
#pragma unroll
        for (int k = 0; k < kWheelSlots; k++) s.wtiles[(w * kWheelSlots + k) * M + ci] = wt[w][k];
    }
    s.reward[ci] = reward, s.prev_reward[ci] = prev_reward;
    s.visited_count[ci] = visited_count, s.last_block[ci] = last_block, s.done[ci] = done;
    s.step_count[ci] = step_count, s.first_step[ci] = 0;
    if (rew_out) rew_out[env * s.players + car] = (float)step_reward;
    if (done_car) done_car[env * s.players + car] = (uint8_t)done;
}

// Env-level bookkeeping after the physics: gym TimeLimit (max_episode_steps = 1000,
// car_racing/register.py:15-26) and FlattenMultiAgentObservation's done = any (atari_wrappers.py:329-330).
__global__ __launch_bounds__(256) void car_post_kernel(CarSoA s, const uint8_t *__restrict__ done_car,
                                                       uint8_t *__restrict__ done_env, int max_episode_steps) {
    const int64_t env = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (env >= s.n) return;
    const int el = s.elapsed[env] + 1;
    bool d = el >= max_episode_steps;
    for (int c = 0; c < s.players; c++) d = d || done_car[s.players * env + c];
    s.elapsed[env] = el;
    done_env[env] = d ? 1 : 0;
}

void launch_car_step(const CarSoA &s, const CarConsts &k, const float *actions, float *rew, uint8_t *done_car, hipStream_t st) {
    const int64_t M = (int64_t)s.players * s.n;
    hipLaunchKernelGGL(car_step_kernel, dim3((unsigned)((M + 63) / 64)), dim3(64), 0, st, s, k, actions, rew, done_car);
}

void launch_car_post(const CarSoA &s, const uint8_t *done_car, uint8_t *done_env, int max_episode_steps, hipStream_t st) {
    hipLaunchKernelGGL(car_post_kernel, dim3((unsigned)((s.n + 255) / 256)), dim3(256), 0, st, s, done_car, done_env,
                       max_episode_steps);
}

}  // namespace crl
