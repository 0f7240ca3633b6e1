// car_solver.h -- Box2D 2.3 island solve for one car (hull + 4 wheels, 4 revolute joints), split
// into its phases so that the per-car kernel and the coupled (car-car contact) kernel share it.
// Restated from b2Island::Solve / b2RevoluteJoint (third-party box2d-py ~=2.3.5; see car_device.h).
#pragma once
#include "car_device.h"

namespace crl {

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

// ---- the two arithmetics of the island solver (template parameter FM; CRL_FLAG_CAR_FMA picks FM = true per context).
// Scope: b2Island::Solve's integrators and the bodies of the velocity / position iterations (joints here, contacts in
// car_contact.hip).  Inside it every mad / nmad site is two roundings when FM is false -- Box2D's own operation order, the default --
// and ONE fused multiply-add when FM is true: the 180 velocity iterations are a dependent chain issued by a lone wavefront at ~5
// cycles per instruction, and the fused form has 0.46 x the instructions (tools/solve_chain_probe.hip: 920 -> 425 cycles per
// iteration).  The CPU checker of tests/ carries the same sites as MAD / NMAD (its -DCRL_FMA build), so either arithmetic is checked at
// tolerance 0.  Constraint initialisation, warm start and Collide are outside the scope and identical in both.
// The terms that are exactly zero because a wheel's joint anchor is the wheel's centre (rB = 0: wB x rB, rB x P) are not evaluated
// in either mode: they add +-0 (the checker's build `norb` == its default build bit for bit, tests/test_car_solver_variants.py).
template <bool FM>
__device__ __forceinline__ float mad(float a, float b, float c) {  // a * b + c
    if constexpr (FM) return __builtin_fmaf(a, b, c);
    else return a * b + c;
}
template <bool FM>
__device__ __forceinline__ float nmad(float a, float b, float c) {  // c - a * b
    if constexpr (FM) return __builtin_fmaf(-a, b, c);
    else return c - a * b;
}
// dot, cross and rotation inside the scope: the FIRST product is the fused one
template <bool FM>
__device__ __forceinline__ float fdot(V2 a, V2 b) { return mad<FM>(a.x, b.x, a.y * b.y); }
template <bool FM>
__device__ __forceinline__ float fcross(V2 a, V2 b) { return mad<FM>(a.x, b.y, -(a.y * b.x)); }
template <bool FM>
__device__ __forceinline__ V2 frot(float s, float c, V2 v) { return mk(mad<FM>(c, v.x, -(s * v.y)), mad<FM>(s, v.x, c * v.y)); }

template <bool FM>
__device__ inline V2 solve22(const M33 &m, V2 b) {
    const float a11 = m.ex[0], a12 = m.ey[0], a21 = m.ex[1], a22 = m.ey[1];
    float det = a11 * a22 - a12 * a21;  // (matrix only: never contracted, the velocity loop gets it from isl_joints_init)
    if (det != 0.0f) det = 1.0f / det;
    return mk(det * mad<FM>(a22, b.x, -(a12 * b.y)), det * mad<FM>(a11, b.y, -(a21 * b.x)));
}

// The joint matrices do not change during a step, so the parts of Solve22 / Solve33 that only depend on the
// matrix (determinant reciprocal, ey x ez) are taken out of the 180 velocity iterations: same operations on the
// same values, computed once (isl_joints_init) instead of per iteration.
__device__ inline float det22_of(const M33 &m) {
    float det = m.ex[0] * m.ey[1] - m.ey[0] * m.ex[1];
    if (det != 0.0f) det = 1.0f / det;
    return det;
}
template <bool FM>
__device__ __forceinline__ V2 solve22_pre(const M33 &m, float det, V2 b) {
    const float a11 = m.ex[0], a12 = m.ey[0], a21 = m.ex[1], a22 = m.ey[1];
    return mk(det * mad<FM>(a22, b.x, -(a12 * b.y)), det * mad<FM>(a11, b.y, -(a21 * b.x)));
}
__device__ inline float det33_of(const M33 &m, float cyz[3]) {
    const float *ex = m.ex, *ey = m.ey, *ez = m.ez;
    cyz[0] = ey[1] * ez[2] - ey[2] * ez[1], cyz[1] = ey[2] * ez[0] - ey[0] * ez[2], cyz[2] = ey[0] * ez[1] - ey[1] * ez[0];
    float det = ex[0] * cyz[0] + ex[1] * cyz[1] + ex[2] * cyz[2];
    if (det != 0.0f) det = 1.0f / det;
    return det;
}
template <bool FM>
__device__ __forceinline__ void solve33_pre(const M33 &m, const float cyz[3], float det, const float b[3], float x[3]) {
    const float *ex = m.ex, *ey = m.ey, *ez = m.ez;
    const float cbz[3] = {mad<FM>(b[1], ez[2], -(b[2] * ez[1])), mad<FM>(b[2], ez[0], -(b[0] * ez[2])), mad<FM>(b[0], ez[1], -(b[1] * ez[0]))};
    const float cyb[3] = {mad<FM>(ey[1], b[2], -(ey[2] * b[1])), mad<FM>(ey[2], b[0], -(ey[0] * b[2])), mad<FM>(ey[0], b[1], -(ey[1] * b[0]))};
    x[0] = det * mad<FM>(b[2], cyz[2], mad<FM>(b[1], cyz[1], b[0] * cyz[0]));
    x[1] = det * mad<FM>(ex[2], cbz[2], mad<FM>(ex[1], cbz[1], ex[0] * cbz[0]));
    x[2] = det * mad<FM>(ex[2], cyb[2], mad<FM>(ex[1], cyb[1], ex[0] * cyb[0]));
}

// One car's solver state while it lives in registers.
struct CarRegs {
    Body H, W[4];
    float imp[4][3], motor_imp[4], motor_speed[4];
    int lim[4];
    float fx[4], fy[4];  // tyre forces applied to the wheels this step
};

struct JointTmp {
    V2 rA[4];
    M33 mass[4];
    float det22[4], det33[4], cyz[4][3];  // matrix-only parts of Solve22 / Solve33
    float motorMass;
};

#define ISL_CONSTS                                                                                                     \
    const float mA = K.hull_inv_mass, iA = K.hull_inv_I, mB = K.wheel_inv_mass, iB = K.wheel_inv_I;                   \
    const V2 lcA = mk(K.hull_lc[0], K.hull_lc[1]);                                                                     \
    (void)mA, (void)iA, (void)mB, (void)iB, (void)lcA

// integrate velocities (the hull carries no applied force; the wheels carry the tyre forces)
template <bool FM>
__device__ inline void isl_integrate_vel(CarRegs &c, const CarConsts &K, float h) {
    const float mB = K.wheel_inv_mass;
#pragma unroll
    for (int w = 0; w < 4; w++) c.W[w].vx = mad<FM>(h, mB * c.fx[w], c.W[w].vx), c.W[w].vy = mad<FM>(h, mB * c.fy[w], c.W[w].vy);
}

// b2RevoluteJoint::InitVelocityConstraints + warm start, joints in island order j3, j2, j1, j0
__device__ inline void isl_joints_init(CarRegs &c, JointTmp &j, const CarConsts &K, float dt_ratio) {
    ISL_CONSTS;
    j.motorMass = iA + iB;
    if (j.motorMass > 0.0f) j.motorMass = 1.0f / j.motorMass;
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const int w = 3 - q;
        float sA, cA;
        crl_sincosf(c.H.a, &sA, &cA);
        j.rA[w] = rotv(sA, cA, mk(K.anchor[w][0], K.anchor[w][1]) - lcA);
        const V2 r = j.rA[w];
        M33 &m = j.mass[w];
        m.ex[0] = mA + mB + r.y * r.y * iA + 0.0f * 0.0f * iB;
        m.ey[0] = -r.y * r.x * iA - 0.0f * 0.0f * iB;
        m.ez[0] = -r.y * iA - 0.0f * iB;
        m.ex[1] = m.ey[0];
        m.ey[1] = mA + mB + r.x * r.x * iA + 0.0f * 0.0f * iB;
        m.ez[1] = r.x * iA + 0.0f * iB;
        m.ex[2] = m.ez[0], m.ey[2] = m.ez[1], m.ez[2] = iA + iB;
        j.det22[w] = det22_of(m), j.det33[w] = det33_of(m, j.cyz[w]);
        const float ja = c.W[w].a - c.H.a - 0.0f;
        if (ja <= LOWER_ANGLE) {
            if (c.lim[w] != LIM_LOWER) c.imp[w][2] = 0;
            c.lim[w] = LIM_LOWER;
        } else if (ja >= UPPER_ANGLE) {
            if (c.lim[w] != LIM_UPPER) c.imp[w][2] = 0;
            c.lim[w] = LIM_UPPER;
        } else {
            c.lim[w] = LIM_INACTIVE, c.imp[w][2] = 0;
        }
        c.imp[w][0] *= dt_ratio, c.imp[w][1] *= dt_ratio, c.imp[w][2] *= dt_ratio, c.motor_imp[w] *= dt_ratio;
        const V2 P = mk(c.imp[w][0], c.imp[w][1]);
        c.H.vx -= mA * P.x, c.H.vy -= mA * P.y;
        c.H.w -= iA * (cross(r, P) + c.motor_imp[w] + c.imp[w][2]);
        c.W[w].vx += mB * P.x, c.W[w].vy += mB * P.y;
        c.W[w].w += iB * (cross(mk(0.f, 0.f), P) + c.motor_imp[w] + c.imp[w][2]);
    }
}

// the motor row of one joint: the impulse is clamped to +- h * maxMotorTorque (> 0: the median of three is the clamp)
template <bool FM>
__device__ __forceinline__ void joint_motor(CarRegs &c, const JointTmp &j, const int w, const float iA, const float iB, const float h) {
    const float Cdot = c.W[w].w - c.H.w - c.motor_speed[w];
    const float old = c.motor_imp[w], maxI = h * MAX_MOTOR_TORQUE;
    const float ni = __builtin_amdgcn_fmed3f(mad<FM>(-j.motorMass, Cdot, old), -maxI, maxI);
    c.motor_imp[w] = ni;
    const float impulse = ni - old;
    c.H.w = nmad<FM>(iA, impulse, c.H.w), c.W[w].w = mad<FM>(iB, impulse, c.W[w].w);
}
// vB + wB x rB - vA - wA x rA at a wheel joint (rB = 0)
template <bool FM>
__device__ __forceinline__ V2 joint_rel_vel(const CarRegs &c, const int w, const V2 r) {
    return mk(mad<FM>(c.H.w, r.y, c.W[w].vx - c.H.vx), nmad<FM>(c.H.w, r.x, c.W[w].vy - c.H.vy));
}
// the point constraint's impulse (and the limit row's `iz`, 0 when the limit is inactive) applied to hull and wheel
template <bool FM>
__device__ __forceinline__ void joint_apply(CarRegs &c, const int w, const V2 r, const V2 P, const float mA, const float iA, const float mB) {
    c.H.vx = nmad<FM>(mA, P.x, c.H.vx), c.H.vy = nmad<FM>(mA, P.y, c.H.vy), c.H.w = nmad<FM>(iA, fcross<FM>(r, P), c.H.w);
    c.W[w].vx = mad<FM>(mB, P.x, c.W[w].vx), c.W[w].vy = mad<FM>(mB, P.y, c.W[w].vy);
}

// one velocity iteration over the 4 joints (motor, then limit / point constraint)
template <bool FM>
__device__ inline void isl_joints_vel(CarRegs &c, const JointTmp &j, const CarConsts &K, float h) {
    ISL_CONSTS;
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const int w = 3 - q;
        const V2 r = j.rA[w];
        joint_motor<FM>(c, j, w, iA, iB, h);
        if (c.lim[w] != LIM_INACTIVE) {
            const V2 Cdot1 = joint_rel_vel<FM>(c, w, r);
            const float Cdot2 = c.W[w].w - c.H.w;
            const float b[3] = {Cdot1.x, Cdot1.y, Cdot2};
            float im[3];
            solve33_pre<FM>(j.mass[w], j.cyz[w], j.det33[w], b, im);
            im[0] = -im[0], im[1] = -im[1], im[2] = -im[2];
            const float newI = c.imp[w][2] + im[2];
            const bool lower = c.lim[w] == LIM_LOWER;
            if (lower ? newI < 0.0f : newI > 0.0f) {
                const V2 rhs = mk(mad<FM>(c.imp[w][2], j.mass[w].ez[0], -Cdot1.x), mad<FM>(c.imp[w][2], j.mass[w].ez[1], -Cdot1.y));
                const V2 red = solve22_pre<FM>(j.mass[w], j.det22[w], rhs);
                im[0] = red.x, im[1] = red.y, im[2] = -c.imp[w][2];
                c.imp[w][0] += red.x, c.imp[w][1] += red.y, c.imp[w][2] = 0;
            } else {
                c.imp[w][0] += im[0], c.imp[w][1] += im[1], c.imp[w][2] += im[2];
            }
            const V2 P = mk(im[0], im[1]);
            c.H.vx = nmad<FM>(mA, P.x, c.H.vx), c.H.vy = nmad<FM>(mA, P.y, c.H.vy), c.H.w = nmad<FM>(iA, fcross<FM>(r, P) + im[2], c.H.w);
            c.W[w].vx = mad<FM>(mB, P.x, c.W[w].vx), c.W[w].vy = mad<FM>(mB, P.y, c.W[w].vy), c.W[w].w = mad<FM>(iB, im[2], c.W[w].w);
        } else {
            const V2 Cdot = joint_rel_vel<FM>(c, w, r);
            const V2 im = solve22_pre<FM>(j.mass[w], j.det22[w], -1.0f * Cdot);
            c.imp[w][0] += im.x, c.imp[w][1] += im.y;
            joint_apply<FM>(c, w, r, im, mA, iA, mB);
        }
    }
}

// Two more forms of the same iteration -- same operations on the same values per lane, different control flow (a dependent
// chain of ~200 instructions issued by one wavefront: what counts is how many instructions the wavefront walks through, and
// every divergent branch is walked by all lanes).  tools/solve_chain_probe.hip: cycles per iteration of a lone wavefront
//   isl_joints_vel (branches per joint)   1 169 no limit active | 1 811 some lanes at a steering limit
//   isl_joints_vel_in  (no limit code)      932
//   isl_joints_vel_sel (selects)          1 377                 | 1 377
// (round 4's numbers, with the rB terms and the compare-and-select clamp still in; round 5's are in docs/LAB_NOTES_r05.md)
// _in: valid while NO joint of the lane's car is at a limit.  _sel: valid while the REAR joints (2, 3: no steering, they never
// reach their limits) are not at a limit; the two steered joints compute the 3x3 and the 2x2 answer and select.
// isl_joint_mode() picks per wavefront.
template <bool FM>
__device__ __forceinline__ void isl_joints_vel_in(CarRegs &c, const JointTmp &j, const CarConsts &K, float h) {
    ISL_CONSTS;
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const int w = 3 - q;
        const V2 r = j.rA[w];
        joint_motor<FM>(c, j, w, iA, iB, h);
        const V2 Cdot = joint_rel_vel<FM>(c, w, r);
        const V2 im = solve22_pre<FM>(j.mass[w], j.det22[w], -1.0f * Cdot);
        c.imp[w][0] += im.x, c.imp[w][1] += im.y;
        joint_apply<FM>(c, w, r, im, mA, iA, mB);
    }
}

template <bool FM>
__device__ __forceinline__ void isl_joints_vel_sel(CarRegs &c, const JointTmp &j, const CarConsts &K, float h) {
    ISL_CONSTS;
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const int w = 3 - q;
        const V2 r = j.rA[w];
        joint_motor<FM>(c, j, w, iA, iB, h);
        const V2 Cdot1 = joint_rel_vel<FM>(c, w, r);
        const V2 im2 = solve22_pre<FM>(j.mass[w], j.det22[w], -1.0f * Cdot1);  // the answer while the limit is inactive
        float ix = im2.x, iy = im2.y, iz = 0.0f;
        float n0 = c.imp[w][0] + im2.x, n1 = c.imp[w][1] + im2.y, n2 = c.imp[w][2];
        bool act = false;
        if (w < 2) {
            act = c.lim[w] != LIM_INACTIVE;
            const float Cdot2 = c.W[w].w - c.H.w;
            const float b[3] = {Cdot1.x, Cdot1.y, Cdot2};
            float im[3];
            solve33_pre<FM>(j.mass[w], j.cyz[w], j.det33[w], b, im);
            im[0] = -im[0], im[1] = -im[1], im[2] = -im[2];
            const float newI = c.imp[w][2] + im[2];
            const bool lower = c.lim[w] == LIM_LOWER;
            const bool clampz = lower ? newI < 0.0f : newI > 0.0f;
            const V2 rhs = mk(mad<FM>(c.imp[w][2], j.mass[w].ez[0], -Cdot1.x), mad<FM>(c.imp[w][2], j.mass[w].ez[1], -Cdot1.y));
            const V2 red = solve22_pre<FM>(j.mass[w], j.det22[w], rhs);
            const float ax = clampz ? red.x : im[0], ay = clampz ? red.y : im[1], az = clampz ? -c.imp[w][2] : im[2];
            const float a0 = c.imp[w][0] + ax, a1 = c.imp[w][1] + ay, a2 = clampz ? 0.0f : c.imp[w][2] + im[2];
            ix = act ? ax : ix, iy = act ? ay : iy, iz = act ? az : iz;
            n0 = act ? a0 : n0, n1 = act ? a1 : n1, n2 = act ? a2 : n2;
        }
        c.imp[w][0] = n0, c.imp[w][1] = n1, c.imp[w][2] = n2;
        const V2 P = mk(ix, iy);
        // (the two paths apply the impulse with the same operations except for the + im[2] of the 3x3 one)
        const float cr = fcross<FM>(r, P);
        const float hw_act = nmad<FM>(iA, cr + iz, c.H.w), hw_in = nmad<FM>(iA, cr, c.H.w);
        const float ww_act = mad<FM>(iB, iz, c.W[w].w);
        c.H.vx = nmad<FM>(mA, P.x, c.H.vx), c.H.vy = nmad<FM>(mA, P.y, c.H.vy);
        c.W[w].vx = mad<FM>(mB, P.x, c.W[w].vx), c.W[w].vy = mad<FM>(mB, P.y, c.W[w].vy);
        c.H.w = act ? hw_act : hw_in, c.W[w].w = act ? ww_act : c.W[w].w;
    }
}

// 0: general form, 1: no limit active anywhere in the wavefront, 2: only steered joints at a limit (wave-uniform)
__device__ __forceinline__ int isl_joint_mode(const CarRegs &c) {
    const bool rear = c.lim[2] != LIM_INACTIVE || c.lim[3] != LIM_INACTIVE, front = c.lim[0] != LIM_INACTIVE || c.lim[1] != LIM_INACTIVE;
    return __any(rear) ? 0 : __any(front) ? 2 : 1;
}
template <bool FM>
__device__ __forceinline__ void isl_joints_vel_mode(int mode, CarRegs &c, const JointTmp &j, const CarConsts &K, float h) {
    if (mode == 1) isl_joints_vel_in<FM>(c, j, K, h);
    else if (mode == 2) isl_joints_vel_sel<FM>(c, j, K, h);
    else isl_joints_vel<FM>(c, j, K, h);
}

template <bool FM>
__device__ inline void integrate_body(Body &b, float h) {
    const V2 tr = mk(h * b.vx, h * b.vy);
    if (fdot<FM>(tr, tr) > MAX_TRANSLATION * MAX_TRANSLATION) {
        const float ratio = MAX_TRANSLATION / sqrtf(fdot<FM>(tr, tr));
        b.vx *= ratio, b.vy *= ratio;
    }
    const float ro = h * b.w;
    if (ro * ro > MAX_ROTATION * MAX_ROTATION) b.w *= MAX_ROTATION / fabsf(ro);
    b.cx = mad<FM>(h, b.vx, b.cx), b.cy = mad<FM>(h, b.vy, b.cy), b.a = mad<FM>(h, b.w, b.a);
}

template <bool FM>
__device__ inline void isl_integrate_pos(CarRegs &c, float h) {
    integrate_body<FM>(c.H, h);
#pragma unroll
    for (int w = 0; w < 4; w++) integrate_body<FM>(c.W[w], h);
}

// one position iteration over the 4 joints; true when all are within slop
template <bool FM>
__device__ inline bool isl_joints_pos(CarRegs &c, const CarConsts &K) {
    ISL_CONSTS;
    float motorMass = iA + iB;
    if (motorMass > 0.0f) motorMass = 1.0f / motorMass;
    bool ok = true;
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const int w = 3 - q;
        float angErr = 0;
        if (c.lim[w] != LIM_INACTIVE) {
            const float angle = c.W[w].a - c.H.a - 0.0f;
            float C;
            if (c.lim[w] == LIM_LOWER) {
                C = angle - LOWER_ANGLE, angErr = -C;
                C = fminf(fmaxf(C + ANGULAR_SLOP, -MAX_ANGULAR_CORRECTION), 0.0f);
            } else {
                C = angle - UPPER_ANGLE, angErr = C;
                C = fminf(fmaxf(C - ANGULAR_SLOP, 0.0f), MAX_ANGULAR_CORRECTION);
            }
            const float li = -motorMass * C;
            c.H.a = nmad<FM>(iA, li, c.H.a), c.W[w].a = mad<FM>(iB, li, c.W[w].a);
        }
        float sA, cA;
        crl_sincosf(c.H.a, &sA, &cA);
        const V2 r = frot<FM>(sA, cA, mk(K.anchor[w][0], K.anchor[w][1]) - lcA);
        const V2 C = (mk(c.W[w].cx, c.W[w].cy) - mk(c.H.cx, c.H.cy)) - r;
        const float posErr = sqrtf(fdot<FM>(C, C));
        M33 k;
        k.ex[0] = mad<FM>(iA * r.y, r.y, mA + mB);
        k.ex[1] = -iA * r.x * r.y;
        k.ey[0] = k.ex[1];
        k.ey[1] = mad<FM>(iA * r.x, r.x, mA + mB);
        const V2 im = -1.0f * solve22<FM>(k, C);
        c.H.cx = nmad<FM>(mA, im.x, c.H.cx), c.H.cy = nmad<FM>(mA, im.y, c.H.cy), c.H.a = nmad<FM>(iA, fcross<FM>(r, im), c.H.a);
        c.W[w].cx = mad<FM>(mB, im.x, c.W[w].cx), c.W[w].cy = mad<FM>(mB, im.y, c.W[w].cy);
        ok = ok && posErr <= LINEAR_SLOP && angErr <= ANGULAR_SLOP;
    }
    return ok;
}

// End of b2Island::Solve ("if (allowSleep)", Box2D 2.3 b2Island.cpp): a body slower than the sleep
// tolerances accumulates m_sleepTime; once every body of the island has been still for
// b2_timeToSleep and the position solver converged the island is put to sleep, which zeroes the
// velocities (b2Body::SetAwake(false)).  Car.step wakes every body again on the next step
// (car_dynamics.py:159-234: joint.motorSpeed assignment, ApplyForceToCenter(..., True)), so the
// visible effect is the velocity reset.  sleep[] = m_sleepTime of hull, wheels 0-3.
#define LIN_SLEEP_TOL 0.01f
#define ANG_SLEEP_TOL (2.0f / 180.0f * 3.14159265359f)
#define TIME_TO_SLEEP 0.5f
__device__ inline float isl_sleep_scan(const CarRegs &c, float *sleep, float h) {
    float min_sleep = 3.402823466e+38f;
    const float lin2 = LIN_SLEEP_TOL * LIN_SLEEP_TOL, ang2 = ANG_SLEEP_TOL * ANG_SLEEP_TOL;
#pragma unroll
    for (int b = 0; b < 5; b++) {
        const Body &B = b == 0 ? c.H : c.W[b == 0 ? 0 : b - 1];
        if (B.w * B.w > ang2 || B.vx * B.vx + B.vy * B.vy > lin2) {
            sleep[b] = 0.0f;
            min_sleep = 0.0f;
        } else {
            sleep[b] += h;
            min_sleep = fminf(min_sleep, sleep[b]);
        }
    }
    return min_sleep;
}
// the same in two halves: what the velocity iterations leave final (velocities, joint impulses and states) and the poses
__device__ inline void store_car_vel(const CarSoA &s, int64_t M, int64_t ci, const CarRegs &c) {
    float *b = s.body + ci;
    b[3 * M] = c.H.vx, b[4 * M] = c.H.vy, b[5 * M] = c.H.w;
#pragma unroll
    for (int w = 0; w < 4; w++) {
        const int o = 6 + 6 * w;
        b[(o + 3) * M] = c.W[w].vx, b[(o + 4) * M] = c.W[w].vy, b[(o + 5) * M] = c.W[w].w;
        s.jimp[(3 * w + 0) * M + ci] = c.imp[w][0], s.jimp[(3 * w + 1) * M + ci] = c.imp[w][1], s.jimp[(3 * w + 2) * M + ci] = c.imp[w][2];
        s.jmotor[w * M + ci] = c.motor_imp[w], s.jspeed[w * M + ci] = c.motor_speed[w], s.jlimit[w * M + ci] = c.lim[w];
    }
}
__device__ inline void store_car_pos(const CarSoA &s, int64_t M, int64_t ci, const CarRegs &c) {
    float *b = s.body + ci;
    b[0 * M] = c.H.cx, b[1 * M] = c.H.cy, b[2 * M] = c.H.a;
#pragma unroll
    for (int w = 0; w < 4; w++) {
        const int o = 6 + 6 * w;
        b[(o + 0) * M] = c.W[w].cx, b[(o + 1) * M] = c.W[w].cy, b[(o + 2) * M] = c.W[w].a;
    }
}
__device__ inline void isl_put_to_sleep(CarRegs &c, float *sleep) {
    c.H.vx = c.H.vy = c.H.w = 0.0f;
#pragma unroll
    for (int w = 0; w < 4; w++) c.W[w].vx = c.W[w].vy = c.W[w].w = 0.0f;
#pragma unroll
    for (int b = 0; b < 5; b++) sleep[b] = 0.0f;
}

// b2Island::Solve for one car on its own
template <bool FM>
__device__ inline void island_solve(CarRegs &c, const CarConsts &K, float h, float dt_ratio, float *sleep) {
    JointTmp j;
    isl_integrate_vel<FM>(c, K, h);
    isl_joints_init(c, j, K, dt_ratio);
    const int mode = isl_joint_mode(c);  // (the limit states are fixed by isl_joints_init for the whole step)
    if (mode == 1) {
#pragma unroll 1
        for (int it = 0; it < 180; it++) isl_joints_vel_in<FM>(c, j, K, h);
    } else if (mode == 2) {
#pragma unroll 1
        for (int it = 0; it < 180; it++) isl_joints_vel_sel<FM>(c, j, K, h);
    } else {
#pragma unroll 1
        for (int it = 0; it < 180; it++) isl_joints_vel<FM>(c, j, K, h);
    }
    isl_integrate_pos<FM>(c, h);
    bool solved = false;
#pragma unroll 1
    for (int it = 0; it < 60; it++)
        if (isl_joints_pos<FM>(c, K)) {
            solved = true;
            break;
        }
    if (isl_sleep_scan(c, sleep, h) >= TIME_TO_SLEEP && solved) isl_put_to_sleep(c, sleep);
}

__device__ inline void load_car(const CarSoA &s, int64_t M, int64_t ci, CarRegs &c) {
    const float *b = s.body + ci;
    c.H.cx = b[0 * M], c.H.cy = b[1 * M], c.H.a = b[2 * M], c.H.vx = b[3 * M], c.H.vy = b[4 * M], c.H.w = b[5 * M];
#pragma unroll
    for (int w = 0; w < 4; w++) {
        const int o = 6 + 6 * w;
        c.W[w].cx = b[(o + 0) * M], c.W[w].cy = b[(o + 1) * M], c.W[w].a = b[(o + 2) * M];
        c.W[w].vx = b[(o + 3) * M], c.W[w].vy = b[(o + 4) * M], c.W[w].w = b[(o + 5) * M];
        c.imp[w][0] = s.jimp[(3 * w + 0) * M + ci], c.imp[w][1] = s.jimp[(3 * w + 1) * M + ci], c.imp[w][2] = s.jimp[(3 * w + 2) * M + ci];
        c.motor_imp[w] = s.jmotor[w * M + ci], c.motor_speed[w] = s.jspeed[w * M + ci], c.lim[w] = s.jlimit[w * M + ci];
        c.fx[w] = 0.f, c.fy[w] = 0.f;
    }
}

__device__ inline void store_car(const CarSoA &s, int64_t M, int64_t ci, const CarRegs &c) {
    float *b = s.body + ci;
    b[0 * M] = c.H.cx, b[1 * M] = c.H.cy, b[2 * M] = c.H.a, b[3 * M] = c.H.vx, b[4 * M] = c.H.vy, b[5 * M] = c.H.w;
#pragma unroll
    for (int w = 0; w < 4; w++) {
        const int o = 6 + 6 * w;
        b[(o + 0) * M] = c.W[w].cx, b[(o + 1) * M] = c.W[w].cy, b[(o + 2) * M] = c.W[w].a;
        b[(o + 3) * M] = c.W[w].vx, b[(o + 4) * M] = c.W[w].vy, b[(o + 5) * M] = c.W[w].w;
        s.jimp[(3 * w + 0) * M + ci] = c.imp[w][0], s.jimp[(3 * w + 1) * M + ci] = c.imp[w][1], s.jimp[(3 * w + 2) * M + ci] = c.imp[w][2];
        s.jmotor[w * M + ci] = c.motor_imp[w], s.jspeed[w * M + ci] = c.motor_speed[w], s.jlimit[w * M + ci] = c.lim[w];
    }
}

}  // namespace crl
