/*
 * car_oracle.c -- CPU restatement of the reference's cCarRacingDouble step path.
 *
 * TEST INFRASTRUCTURE ONLY (tests/, __graft_entry__.smoke(), bench.py cpu_baseline).
 *
 * Pinning status (DESIGN.md "Oracle"):
 *   - track generation, wheel model, action mapping, tile-visit reward rule: PINNED to the
 *     reference's own Python via tests/golden/car_{track,wheels,rules}.npz.
 *   - b2World.Step (rigid bodies, revolute joints, sensor overlap): box2d-py ~=2.3.5 is a
 *     third-party dependency that is neither vendored nor installable; its published
 *     algorithm (Box2D 2.3 b2Island::Solve / b2RevoluteJoint / b2PolygonShape::ComputeMass)
 *     is restated here in float32 (joints, wheel-tile sensors, car-car contacts).  PARITY UNPINNED.
 *   - observation raster: restated literally (pre-rastered palette map at reset, 192 x 192 crop,
 *     pygame's nearest-neighbour rotate, blit, car polygons, indicator bars).  The reference's own
 *     Python around the pygame calls is PINNED via tests/golden/car_obs.npz; pygame 1.9.6 itself
 *     (third-party, not installable) is restated from its published source, UNPINNED.
 *   - liboracle.so evaluates sin / cos / atan2 through include/crl_rot.h and include/crl_f64.h (shared
 *     with the HIP kernels); liboracle_libm.so (-DCRL_LIBM) calls the host libm like the reference
 *     does and is the build the reference-recorded fixtures are checked against bit for bit.
 *
 * Citations: car_racing/car_racing_multi_players.py = "crmp", car_racing/car_dynamics.py = "cd".
 * Build: -O2 -ffp-contract=off (f64 parts are CPython arithmetic, f32 parts Box2D's).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "car_oracle.h"
/* Transcendentals.  Default build: the evaluations shared with the HIP kernels (include/crl_rot.h for
 * Box2D's float32 b2Rot, include/crl_f64.h for CPython's math.sin / cos / atan2 and pygame's C doubles),
 * so oracle and GPU agree bit for bit.  -DCRL_LIBM (liboracle_libm.so): the host's libm, i.e. what the
 * reference itself calls -- that build is the one pinned bit for bit to the fixtures recorded from
 * the reference's Python; tests/test_oracle_libm_delta.py measures the distance between the two. */
#ifdef CRL_LIBM
static inline void rot_sincosf(float a, float *s, float *c) { *s = sinf(a), *c = cosf(a); }
#define m_sin sin
#define m_cos cos
#define m_atan2 atan2
#define t_sin sin
#define t_cos cos
#define t_atan2 atan2
#else
#include "../include/crl_f64.h"
#include "../include/crl_rot.h"
#define rot_sincosf crl_sincosf
#define m_sin crl_sin /* camera, pygame's rotate: the correctly rounded double-double evaluations */
#define m_cos crl_cos
#define m_atan2 crl_atan2
#define t_sin crl_sin_fast /* _create_track's walk: the plain-double evaluations (<= 3 ulp), as the GPU walks it */
#define t_cos crl_cos_fast
#define t_atan2 crl_atan2_fast
#endif

/* ---- constants (crmp:54-88, cd:17-51) */
#define SCALE 6.0
#define TRACK_RAD (900 / SCALE)
#define PLAYFIELD (2000 / SCALE)
#define FPS 50
#define TRACK_DETAIL_STEP (21 / SCALE)
#define TRACK_TURN_RATE 0.31
#define TRACK_WIDTH (40 / SCALE)
#define BORDER (8 / SCALE)
#define BORDER_MIN_COUNT 4
#define SIZE 0.02
#define ENGINE_POWER (100000000 * SIZE * SIZE)
#define WHEEL_MOMENT_OF_INERTIA (4000 * SIZE * SIZE)
#define FRICTION_LIMIT (1000000 * SIZE * SIZE)
#define WHEEL_R 27
#define WHEEL_W 14

static const double WHEELPOS[4][2] = {{-55, +80}, {+55, +80}, {-55, -82}, {+55, -82}};
static const double HULL1[4][2] = {{-60, 130}, {60, 130}, {60, 110}, {-60, 110}};
static const double HULL2[4][2] = {{-15, 120}, {15, 120}, {20, 20}, {-20, 20}};
static const double HULL3[8][2] = {{25, 20}, {50, -10}, {50, -40}, {20, -90}, {-20, -90}, {-50, -40}, {-50, -10}, {-25, 20}};
static const double HULL4[4][2] = {{-50, -120}, {50, -120}, {50, -90}, {-50, -90}};

/* ------------------------------------------------------------------ track (crmp:262-452) */
static double sgn(double v) { return (v > 0) - (v < 0); }

int car_oracle_create_track(const double *u /*24 draws*/, car_track *out) {
    const int CHECKPOINTS = 12;
    double cp[12][3];
    double start_alpha = 0;
    int ui = 0;
    for (int c = 0; c < CHECKPOINTS; c++) {
        double noise = 0 + (2 * M_PI * 1 / CHECKPOINTS - 0) * u[ui++];
        double alpha = 2 * M_PI * c / CHECKPOINTS + noise;
        double rad = TRACK_RAD / 3 + (TRACK_RAD - TRACK_RAD / 3) * u[ui++];
        if (c == 0) alpha = 0, rad = 1.5 * TRACK_RAD;
        if (c == CHECKPOINTS - 1) {
            alpha = 2 * M_PI * c / CHECKPOINTS;
            start_alpha = 2 * M_PI * (-0.5) / CHECKPOINTS;
            rad = 1.5 * TRACK_RAD;
        }
        cp[c][0] = alpha, cp[c][1] = rad * t_cos(alpha), cp[c][2] = rad * t_sin(alpha);
    }
    static __thread double tr[2600][4];
    double x = 1.5 * TRACK_RAD, y = 0, beta = 0;
    long dest_i = 0;
    int laps = 0, n = 0, no_freeze = 2500, visited_other_side = 0;
    for (;;) {
        double alpha = t_atan2(y, x);
        if (visited_other_side && alpha > 0) laps++, visited_other_side = 0;
        if (alpha < 0) visited_other_side = 1, alpha += 2 * M_PI;
        double dest_alpha, dest_x, dest_y;
        for (;;) {
            int failed = 1;
            for (;;) {
                dest_alpha = cp[dest_i % CHECKPOINTS][0], dest_x = cp[dest_i % CHECKPOINTS][1], dest_y = cp[dest_i % CHECKPOINTS][2];
                if (alpha <= dest_alpha) { failed = 0; break; }
                dest_i++;
                if (dest_i % CHECKPOINTS == 0) break;
            }
            if (!failed) break;
            alpha -= 2 * M_PI;
        }
        double r1x = t_cos(beta), r1y = t_sin(beta), p1x = -r1y, p1y = r1x;
        double dest_dx = dest_x - x, dest_dy = dest_y - y;
        double proj = r1x * dest_dx + r1y * dest_dy;
        while (beta - alpha > 1.5 * M_PI) beta -= 2 * M_PI;
        while (beta - alpha < -1.5 * M_PI) beta += 2 * M_PI;
        double prev_beta = beta;
        proj *= SCALE;
        if (proj > 0.3) beta -= fmin(TRACK_TURN_RATE, fabs(0.001 * proj));
        if (proj < -0.3) beta += fmin(TRACK_TURN_RATE, fabs(0.001 * proj));
        x += p1x * TRACK_DETAIL_STEP;
        y += p1y * TRACK_DETAIL_STEP;
        tr[n][0] = alpha, tr[n][1] = prev_beta * 0.5 + beta * 0.5, tr[n][2] = x, tr[n][3] = y;
        n++;
        if (laps > 4) break;
        if (--no_freeze == 0) break;
    }
    int i1 = -1, i2 = -1, i = n;
    for (;;) {
        i--;
        if (i == 0) return 0;
        int pass = tr[i][0] > start_alpha && tr[i - 1][0] <= start_alpha;
        if (pass && i2 == -1) i2 = i;
        else if (pass && i1 == -1) { i1 = i; break; }
    }
    int len = (i2 - 1) - i1;
    if (len <= 0 || len > CAR_MAX_TILES) return 0;
    double(*t)[4] = &tr[i1];
    double fb = t[0][1], fpx = t_cos(fb), fpy = t_sin(fb);
    double a = fpx * (t[0][2] - t[len - 1][2]), b = fpy * (t[0][3] - t[len - 1][3]);
    double glued = sqrt(a * a + b * b);
    if (glued > TRACK_DETAIL_STEP) return 0;
    out->n = len;
    for (int k = 0; k < len; k++) memcpy(out->track[k], t[k], sizeof(double) * 4);
    /* red-white border on hard turns */
    uint8_t border[CAR_MAX_TILES];
    for (int k = 0; k < len; k++) {
        int good = 1;
        double oneside = 0;
        for (int neg = 0; neg < BORDER_MIN_COUNT; neg++) {
            double b1 = t[((k - neg - 0) % len + len) % len][1], b2 = t[((k - neg - 1) % len + len) % len][1];
            good &= fabs(b1 - b2) > TRACK_TURN_RATE * 0.2;
            oneside += sgn(b1 - b2);
        }
        good &= fabs(oneside) == BORDER_MIN_COUNT;
        border[k] = (uint8_t)good;
    }
    for (int k = 0; k < len; k++)
        for (int neg = 0; neg < BORDER_MIN_COUNT; neg++) border[((k - neg) % len + len) % len] |= border[k];
    /* tiles: i = len-1 .. 0 between track[i] and track[i-1] */
    for (int k = len - 1; k >= 0; k--) {
        const double *p1 = t[k], *p2 = t[((k - 1) % len + len) % len];
        double b1 = p1[1], x1 = p1[2], y1 = p1[3], b2 = p2[1], x2 = p2[2], y2 = p2[3];
        double v[5][2] = {
            {x1 - TRACK_WIDTH * t_cos(b1), y1 - TRACK_WIDTH * t_sin(b1)},
            {x1 - TRACK_WIDTH / 2 * t_cos(b1 - M_PI / 2), y1 - TRACK_WIDTH / 2 * t_sin(b1 - M_PI / 2)},
            {x1 + TRACK_WIDTH * t_cos(b1), y1 + TRACK_WIDTH * t_sin(b1)},
            {x2 + TRACK_WIDTH * t_cos(b2), y2 + TRACK_WIDTH * t_sin(b2)},
            {x2 - TRACK_WIDTH * t_cos(b2), y2 - TRACK_WIDTH * t_sin(b2)},
        };
        memcpy(out->tile[k], v, sizeof(v));
        out->border[k] = border[k];
        if (border[k]) {
            double side = sgn(b2 - b1);
            double bp[4][2] = {
                {x1 + side * TRACK_WIDTH * t_cos(b1), y1 + side * TRACK_WIDTH * t_sin(b1)},
                {x1 + side * (TRACK_WIDTH + BORDER) * t_cos(b1), y1 + side * (TRACK_WIDTH + BORDER) * t_sin(b1)},
                {x2 + side * (TRACK_WIDTH + BORDER) * t_cos(b2), y2 + side * (TRACK_WIDTH + BORDER) * t_sin(b2)},
                {x2 + side * TRACK_WIDTH * t_cos(b2), y2 + side * TRACK_WIDTH * t_sin(b2)},
            };
            memcpy(out->border_poly[k], bp, sizeof(bp));
        } else {
            memset(out->border_poly[k], 0, sizeof(out->border_poly[k]));
        }
    }
    return 1;
}

/* ------------------------------------------------------------------ Box2D 2.3 pieces (float32) */
typedef struct { float x, y; } v2;
static inline v2 V(float x, float y) { v2 r = {x, y}; return r; }
static inline v2 vadd(v2 a, v2 b) { return V(a.x + b.x, a.y + b.y); }
static inline v2 vsub(v2 a, v2 b) { return V(a.x - b.x, a.y - b.y); }
static inline v2 vmul(float s, v2 a) { return V(s * a.x, s * a.y); }
static inline float vdot(v2 a, v2 b) { return a.x * b.x + a.y * b.y; }
static inline float vcross(v2 a, v2 b) { return a.x * b.y - a.y * b.x; }
static inline v2 scross(float s, v2 a) { return V(-s * a.y, s * a.x); } /* b2Cross(float, vec) */
static inline v2 rot(float s, float c, v2 v) { return V(c * v.x - s * v.y, s * v.x + c * v.y); }

/* ---- the CONTRACTED arithmetic of the island solver: -DCRL_FMA (liboracle_fma.so, `make -C oracle fma`).
 * Scope: b2Island::Solve's integrators and the bodies of the velocity / position iterations (revolute joints and contacts);
 * constraint initialisation, warm start, Collide and everything outside world.Step are the same in both builds.  Inside
 * that scope every `a * b + c` / `c - a * b` site below is ONE fused multiply-add (a single rounding) in the FMA build and
 * two roundings in the default build, and the terms that are exactly zero because a wheel's joint anchor is its centre
 * (rB = 0) are not evaluated in the FMA build.  MAD / NMAD expand to the default build's ORIGINAL expressions (IEEE addition
 * and multiplication commute exactly and x + (-y) == x - y): tools/oracle_regress.py keeps a digest of the default
 * build's states from before the macros went in.  The HIP kernels follow the same definition behind CRL_FLAG_CAR_FMA
 * (csrc/car_solver.h, car_contact.hip: template parameter FM), so HIP(fma) == liboracle_fma.so at tolerance 0. */
#if defined(CRL_FMA) && !defined(CRL_NO_RB)
#define CRL_NO_RB 1 /* the exactly-zero rB terms of the wheel joints are not evaluated (alone: `make -C oracle norb`, a default build
                     * without them -- tests/test_oracle_car_physics.py shows its states are the default build's, bit for bit) */
#endif
#ifdef CRL_FMA
#define MAD(a, b, c) fmaf((a), (b), (c))   /* a * b + c */
#define NMAD(a, b, c) fmaf(-(a), (b), (c)) /* c - a * b */
#else
#define MAD(a, b, c) ((a) * (b) + (c))
#define NMAD(a, b, c) ((c) - (a) * (b))
#endif
/* dot, cross and rotation inside the scope: the FIRST product is the fused one */
static inline float fdot(v2 a, v2 b) { return MAD(a.x, b.x, a.y * b.y); }
static inline float fcross(v2 a, v2 b) { return MAD(a.x, b.y, -(a.y * b.x)); }
static inline v2 frot(float s, float c, v2 v) { return V(MAD(c, v.x, -(s * v.y)), MAD(s, v.x, c * v.y)); }

/* b2PolygonShape::ComputeMass: mass, centroid, inertia about the shape origin */
static void poly_mass(const v2 *vs, int n, float density, float *mass, v2 *center, float *I) {
    v2 c = V(0, 0), s = V(0, 0);
    float area = 0, in = 0;
    for (int i = 0; i < n; i++) s = vadd(s, vs[i]);
    s = vmul(1.0f / n, s);
    const float k_inv3 = 1.0f / 3.0f;
    for (int i = 0; i < n; i++) {
        v2 e1 = vsub(vs[i], s), e2 = vsub(vs[i + 1 < n ? i + 1 : 0], s);
        float D = vcross(e1, e2), ta = 0.5f * D;
        area += ta;
        c = vadd(c, vmul(ta * k_inv3, vadd(e1, e2)));
        float intx2 = e1.x * e1.x + e2.x * e1.x + e2.x * e2.x, inty2 = e1.y * e1.y + e2.y * e1.y + e2.y * e2.y;
        in += (0.25f * k_inv3 * D) * (intx2 + inty2);
    }
    *mass = density * area;
    c = vmul(1.0f / area, c);
    *center = vadd(c, s);
    *I = density * in;
    *I += *mass * (vdot(*center, *center) - vdot(c, c));
}

/* b2PolygonShape::Set orders vertices counter-clockwise (convex hull); the car polygons
 * are convex, so reversing a clockwise list is the same hull. */
static int make_ccw(const double (*src)[2], int n, double scale, v2 *dst) {
    double area = 0;
    for (int i = 0; i < n; i++) {
        int j = (i + 1) % n;
        area += src[i][0] * src[j][1] - src[j][0] * src[i][1];
    }
    for (int i = 0; i < n; i++) {
        int k = area > 0 ? i : n - 1 - i;
        dst[i] = V((float)(src[k][0] * scale), (float)(src[k][1] * scale));
    }
    return n;
}

static car_consts K;
static int K_ready = 0;

const car_consts *car_oracle_consts(void) {
    if (K_ready) return &K;
    /* hull: 4 fixtures, density 1 (cd:61-71); b2Body::ResetMassData */
    const double(*polys[4])[2] = {HULL1, HULL2, HULL3, HULL4};
    const int cnt[4] = {4, 4, 8, 4};
    float mass = 0, I = 0;
    v2 lc = V(0, 0);
    for (int f = 0; f < 4; f++) {
        K.hull_n[f] = make_ccw(polys[f], cnt[f], SIZE, (v2 *)K.hull_poly[f]);
        float m, i;
        v2 c;
        poly_mass((v2 *)K.hull_poly[f], cnt[f], 1.0f, &m, &c, &i);
        mass += m, lc = vadd(lc, vmul(m, c)), I += i;
    }
    K.hull_mass = mass, K.hull_inv_mass = 1.0f / mass;
    lc = vmul(K.hull_inv_mass, lc);
    K.hull_lc[0] = lc.x, K.hull_lc[1] = lc.y;
    I -= mass * vdot(lc, lc);
    K.hull_I = I, K.hull_inv_I = 1.0f / I;
    /* wheel: box (+-14, +-27) * SIZE, density 0.1 (cd:83-97) */
    const double wp[4][2] = {{-WHEEL_W, +WHEEL_R}, {+WHEEL_W, +WHEEL_R}, {+WHEEL_W, -WHEEL_R}, {-WHEEL_W, -WHEEL_R}};
    make_ccw(wp, 4, SIZE, (v2 *)K.wheel_poly);
    float m, i;
    v2 c;
    poly_mass((v2 *)K.wheel_poly, 4, 0.1f, &m, &c, &i);
    i -= m * vdot(c, c);
    K.wheel_mass = m, K.wheel_inv_mass = 1.0f / m, K.wheel_I = i, K.wheel_inv_I = 1.0f / i;
    for (int w = 0; w < 4; w++) K.anchor[w][0] = (float)(WHEELPOS[w][0] * SIZE), K.anchor[w][1] = (float)(WHEELPOS[w][1] * SIZE);
    K_ready = 1;
    return &K;
}

/* b2Body state <-> transform: xf.q = b2Rot(a); xf.p = c - q * localCenter */
static void body_xf(const car_body *b, v2 lc, float *s, float *c, v2 *p) {
    rot_sincosf(b->a, s, c);
    *p = vsub(V(b->cx, b->cy), rot(*s, *c, lc));
}

/* Car.__init__ (cd:55-129): hull + 4 wheels at the birth place, all at init_angle */
void car_oracle_place(car_state *car, double init_angle, double init_x, double init_y, int birth_place_index) {
    car_oracle_consts();
    memset(car, 0, sizeof(*car));
    init_x -= birth_place_index % 2 * 5;
    init_y -= floor(birth_place_index / 2.0) * 10;
    float a = (float)init_angle, s, c;
    rot_sincosf(a, &s, &c);
    v2 p = V((float)init_x, (float)init_y);
    v2 com = vadd(p, rot(s, c, V(K.hull_lc[0], K.hull_lc[1])));
    car->hull.cx = com.x, car->hull.cy = com.y, car->hull.a = a;
    for (int w = 0; w < 4; w++) {
        /* position=(init_x + wx*SIZE, init_y + wy*SIZE): NOT rotated by init_angle (cd:86) */
        car->wheel[w].cx = (float)(init_x + WHEELPOS[w][0] * SIZE), car->wheel[w].cy = (float)(init_y + WHEELPOS[w][1] * SIZE);
        car->wheel[w].a = a;
    }
}

/* CarRacing.process_action (crmp:527-540) */
void car_oracle_process_action(const double a[2], double out[3]) {
    double a0 = fmax(fmin(a[0], 1), -1), a1 = fmax(fmin(a[1], 1), -1), a2;
    if (a1 > 0) a2 = 0;
    else a2 = a1, a1 = 0;
    out[0] = a0, out[1] = fabs(a1), out[2] = fabs(a2);
}

/* Car.steer / gas / brake (cd:131-157) */
void car_oracle_controls(car_state *car, double steer, double gas, double brake) {
    car->steer[0] = car->steer[1] = steer;
    gas = gas < 0 ? 0 : gas > 1 ? 1 : gas;
    for (int w = 2; w < 4; w++) {
        double diff = gas - car->gas[w];
        if (diff > 0.1) diff = 0.1;
        car->gas[w] += diff;
    }
    for (int w = 0; w < 4; w++) car->brake[w] = brake;
}

/* Car.step (cd:159-234), one wheel.  Inputs that the reference reads back from Box2D are
 * float32 values widened to double; outputs motorSpeed / force narrow to float32. */
void car_oracle_wheel(double dt, double steer, double gas, double brake, double joint_angle, double q_sin, double q_cos,
                      double vx, double vy, int on_road, double *omega, double *phase, double *motor_speed, double force[2]) {
    double d = steer - joint_angle;
    *motor_speed = sgn(d) * fmin(50.0 * fabs(d), 3.0);
    double friction_limit = FRICTION_LIMIT * 0.6;
    if (on_road) friction_limit = fmax(friction_limit, FRICTION_LIMIT * 1.0);
    float s = (float)q_sin, c = (float)q_cos; /* the wheel body's b2Rot (xf.q) */
    /* GetWorldVector((0,1)) / ((1,0)) = b2Mul(q, v) in float32 */
    double forw[2] = {(double)(c * 0.0f - s * 1.0f), (double)(s * 0.0f + c * 1.0f)};
    double side[2] = {(double)(c * 1.0f - s * 0.0f), (double)(s * 1.0f + c * 0.0f)};
    double vf = forw[0] * vx + forw[1] * vy, vs = side[0] * vx + side[1] * vy;
    double om = *omega;
    om += dt * ENGINE_POWER * gas / WHEEL_MOMENT_OF_INERTIA / (fabs(om) + 5.0);
    if (brake >= 0.9) om = 0;
    else if (brake > 0) {
        double dir = -sgn(om), val = 15 * brake;
        if (fabs(val) > fabs(om)) val = fabs(om);
        om += dir * val;
    }
    *phase += om * dt;
    double wheel_rad = 1.0 * WHEEL_R * SIZE;
    double vr = om * wheel_rad, f_force = -vf + vr, p_force = -vs;
    f_force *= 205000 * SIZE * SIZE, p_force *= 205000 * SIZE * SIZE;
    double fo = sqrt(f_force * f_force + p_force * p_force);
    if (fabs(fo) > friction_limit) {
        f_force /= fo, p_force /= fo;
        fo = friction_limit;
        f_force *= fo, p_force *= fo;
    }
    om -= dt * f_force * wheel_rad / WHEEL_MOMENT_OF_INERTIA;
    *omega = om;
    force[0] = p_force * side[0] + f_force * forw[0], force[1] = p_force * side[1] + f_force * forw[1];
}

/* ---- convex polygon distance (what b2TestOverlap's GJK distance decides for sensors) */
static float seg_seg_dist2(v2 p1, v2 q1, v2 p2, v2 q2) {
    v2 d1 = vsub(q1, p1), d2 = vsub(q2, p2), r = vsub(p1, p2);
    float a = vdot(d1, d1), e = vdot(d2, d2), f = vdot(d2, r), s, t;
    const float EPS = 1e-12f;
    if (a <= EPS && e <= EPS) return vdot(r, r);
    if (a <= EPS) s = 0, t = fminf(fmaxf(f / e, 0), 1);
    else {
        float c = vdot(d1, r);
        if (e <= EPS) t = 0, s = fminf(fmaxf(-c / a, 0), 1);
        else {
            float b = vdot(d1, d2), den = a * e - b * b;
            s = den != 0 ? fminf(fmaxf((b * f - c * e) / den, 0), 1) : 0;
            t = (b * s + f) / e;
            if (t < 0) t = 0, s = fminf(fmaxf(-c / a, 0), 1);
            else if (t > 1) t = 1, s = fminf(fmaxf((b - c) / a, 0), 1);
        }
    }
    v2 c1 = vadd(p1, vmul(s, d1)), c2 = vadd(p2, vmul(t, d2)), d = vsub(c1, c2);
    return vdot(d, d);
}

static int point_in_convex(v2 p, const v2 *poly, int n) { /* CCW polygon */
    for (int i = 0; i < n; i++) {
        v2 a = poly[i], b = poly[i + 1 < n ? i + 1 : 0];
        if (vcross(vsub(b, a), vsub(p, a)) < 0) return 0;
    }
    return 1;
}

/* squared distance between two convex CCW polygons (0 when they intersect) */
static float poly_dist2(const v2 *A, int nA, const v2 *B, int nB) {
    if (point_in_convex(A[0], B, nB) || point_in_convex(B[0], A, nA)) return 0;
    float best = 3.4e38f;
    for (int i = 0; i < nA; i++)
        for (int j = 0; j < nB; j++) {
            float d = seg_seg_dist2(A[i], A[i + 1 < nA ? i + 1 : 0], B[j], B[j + 1 < nB ? j + 1 : 0]);
            if (d < best) best = d;
        }
    return best;
}

/* ---- revolute joint (b2RevoluteJoint, bodyA = hull, bodyB = wheel) */
enum { LIM_INACTIVE = 0, LIM_LOWER = 1, LIM_UPPER = 2 };
#define LINEAR_SLOP 0.005f
#define ANGULAR_SLOP (2.0f / 180.0f * 3.14159265359f)
#define MAX_ANGULAR_CORRECTION (8.0f / 180.0f * 3.14159265359f)
#define MAX_TRANSLATION 2.0f
#define MAX_ROTATION (0.5f * 3.14159265359f)
#define LOWER_ANGLE (-0.4f)
#define UPPER_ANGLE (+0.4f)
#define MAX_MOTOR_TORQUE ((float)(180 * 900 * SIZE * SIZE))

typedef struct {
    v2 rA, rB;
    float m[3][3]; /* m[col][row]: ex = m[0], ey = m[1], ez = m[2] */
    float motorMass;
} joint_tmp;

static v2 solve22(const float m[3][3], v2 b) {
    float a11 = m[0][0], a12 = m[1][0], a21 = m[0][1], a22 = m[1][1];
    float det = a11 * a22 - a12 * a21; /* (matrix only: not contracted -- the GPU takes it out of the velocity loop) */
    if (det != 0.0f) det = 1.0f / det;
    return V(det * MAD(a22, b.x, -(a12 * b.y)), det * MAD(a11, b.y, -(a21 * b.x)));
}

static void solve33(const float m[3][3], const float b[3], float x[3]) {
    const float *ex = m[0], *ey = m[1], *ez = m[2];
    float cyz[3] = {ey[1] * ez[2] - ey[2] * ez[1], ey[2] * ez[0] - ey[0] * ez[2], ey[0] * ez[1] - ey[1] * ez[0]};
    float det = ex[0] * cyz[0] + ex[1] * cyz[1] + ex[2] * cyz[2];
    if (det != 0.0f) det = 1.0f / det;
    float cbz[3] = {MAD(b[1], ez[2], -(b[2] * ez[1])), MAD(b[2], ez[0], -(b[0] * ez[2])), MAD(b[0], ez[1], -(b[1] * ez[0]))};
    float cyb[3] = {MAD(ey[1], b[2], -(ey[2] * b[1])), MAD(ey[2], b[0], -(ey[0] * b[2])), MAD(ey[0], b[1], -(ey[1] * b[0]))};
    x[0] = det * MAD(b[2], cyz[2], MAD(b[1], cyz[1], b[0] * cyz[0]));
    x[1] = det * MAD(ex[2], cbz[2], MAD(ex[1], cbz[1], ex[0] * cbz[0]));
    x[2] = det * MAD(ex[2], cyb[2], MAD(ex[1], cyb[1], ex[0] * cyb[0]));
}

/* b2Island::Solve, split into its phases so that one car (the usual case) and two cars coupled by
 * contacts run the same code.  Bodies of a car: [hull, w0..w3]; its joints are solved in the order
 * j3, j2, j1, j0 (island build order of b2World::Solve for bodies created hull, w0, w1, w2, w3). */
#define ISL_CONSTS \
    const float mA = K.hull_inv_mass, iA = K.hull_inv_I, mB = K.wheel_inv_mass, iB = K.wheel_inv_I; \
    car_body *H = &car->hull; \
    const v2 lcA = V(K.hull_lc[0], K.hull_lc[1]); \
    (void)mA, (void)iA, (void)mB, (void)iB, (void)H, (void)lcA

/* integrate velocities: v += h * invMass * F (no gravity, no damping, no torque) */
static void isl_integrate_vel(car_state *car, float h) {
    ISL_CONSTS;
    H->vx = MAD(h, mA * H->fx, H->vx), H->vy = MAD(h, mA * H->fy, H->vy);
    for (int w = 0; w < 4; w++) {
        car_body *B = &car->wheel[w];
        B->vx = MAD(h, mB * B->fx, B->vx), B->vy = MAD(h, mB * B->fy, B->vy);
    }
}

/* b2RevoluteJoint::InitVelocityConstraints (+ warm start) for the 4 joints */
static void isl_joints_init(car_state *car, joint_tmp *jt, float dt_ratio) {
    ISL_CONSTS;
    for (int q = 0; q < 4; q++) {
        int w = 3 - q;
        car_body *B = &car->wheel[w];
        joint_tmp *j = &jt[w];
        float sA, cA;
        rot_sincosf(H->a, &sA, &cA);
        j->rA = rot(sA, cA, vsub(V(K.anchor[w][0], K.anchor[w][1]), lcA));
        j->rB = V(0, 0); /* rot(qB, localAnchorB - localCenterB) = 0 */
        v2 rA = j->rA, rB = j->rB;
        j->m[0][0] = mA + mB + rA.y * rA.y * iA + rB.y * rB.y * iB;
        j->m[1][0] = -rA.y * rA.x * iA - rB.y * rB.x * iB;
        j->m[2][0] = -rA.y * iA - rB.y * iB;
        j->m[0][1] = j->m[1][0];
        j->m[1][1] = mA + mB + rA.x * rA.x * iA + rB.x * rB.x * iB;
        j->m[2][1] = rA.x * iA + rB.x * iB;
        j->m[0][2] = j->m[2][0], j->m[1][2] = j->m[2][1], j->m[2][2] = iA + iB;
        j->motorMass = iA + iB;
        if (j->motorMass > 0.0f) j->motorMass = 1.0f / j->motorMass;
        float ja = B->a - H->a - 0.0f;
        if (ja <= LOWER_ANGLE) {
            if (car->limit_state[w] != LIM_LOWER) car->imp[w][2] = 0;
            car->limit_state[w] = LIM_LOWER;
        } else if (ja >= UPPER_ANGLE) {
            if (car->limit_state[w] != LIM_UPPER) car->imp[w][2] = 0;
            car->limit_state[w] = LIM_UPPER;
        } else {
            car->limit_state[w] = LIM_INACTIVE, car->imp[w][2] = 0;
        }
        /* warm start */
        car->imp[w][0] *= dt_ratio, car->imp[w][1] *= dt_ratio, car->imp[w][2] *= dt_ratio, car->motor_imp[w] *= dt_ratio;
        v2 P = V(car->imp[w][0], car->imp[w][1]);
        H->vx -= mA * P.x, H->vy -= mA * P.y;
        H->w -= iA * (vcross(rA, P) + car->motor_imp[w] + car->imp[w][2]);
        B->vx += mB * P.x, B->vy += mB * P.y;
        B->w += iB * (vcross(rB, P) + car->motor_imp[w] + car->imp[w][2]);
    }
}

/* vB + wB x rB - vA - wA x rA at a wheel joint; rB = 0 (the anchor is the wheel's centre) */
static inline v2 joint_rel_vel(const car_body *H, const car_body *B, v2 rA, v2 rB) {
#ifdef CRL_NO_RB
    (void)rB;
    return V(MAD(H->w, rA.y, B->vx - H->vx), NMAD(H->w, rA.x, B->vy - H->vy));
#else
    v2 vA = V(H->vx, H->vy), vB = V(B->vx, B->vy);
    return vsub(vsub(vadd(vB, scross(B->w, rB)), vA), scross(H->w, rA));
#endif
}

/* one velocity iteration: b2RevoluteJoint::SolveVelocityConstraints for the 4 joints */
static void isl_joints_vel(car_state *car, joint_tmp *jt, float h) {
    ISL_CONSTS;
        for (int q = 0; q < 4; q++) {
            int w = 3 - q;
            car_body *B = &car->wheel[w];
            joint_tmp *j = &jt[w];
            v2 rA = j->rA, rB = j->rB;
            { /* motor */
                float Cdot = B->w - H->w - car->motor_speed[w];
                float old = car->motor_imp[w], maxI = h * MAX_MOTOR_TORQUE;
                float ni = MAD(-j->motorMass, Cdot, old);
                ni = ni < -maxI ? -maxI : ni > maxI ? maxI : ni;
                car->motor_imp[w] = ni;
                float impulse = ni - old;
                H->w = NMAD(iA, impulse, H->w), B->w = MAD(iB, impulse, B->w);
            }
            if (car->limit_state[w] != LIM_INACTIVE) {
                v2 Cdot1 = joint_rel_vel(H, B, rA, rB);
                float Cdot2 = B->w - H->w;
                float b[3] = {Cdot1.x, Cdot1.y, Cdot2}, imp[3];
                solve33(j->m, b, imp);
                imp[0] = -imp[0], imp[1] = -imp[1], imp[2] = -imp[2];
                float newI = car->imp[w][2] + imp[2];
                int lower = car->limit_state[w] == LIM_LOWER;
                if (lower ? newI < 0.0f : newI > 0.0f) {
                    v2 rhs = V(MAD(car->imp[w][2], j->m[2][0], -Cdot1.x), MAD(car->imp[w][2], j->m[2][1], -Cdot1.y));
                    v2 red = solve22(j->m, rhs);
                    imp[0] = red.x, imp[1] = red.y, imp[2] = -car->imp[w][2];
                    car->imp[w][0] += red.x, car->imp[w][1] += red.y, car->imp[w][2] = 0;
                } else {
                    car->imp[w][0] += imp[0], car->imp[w][1] += imp[1], car->imp[w][2] += imp[2];
                }
                v2 P = V(imp[0], imp[1]);
                H->vx = NMAD(mA, P.x, H->vx), H->vy = NMAD(mA, P.y, H->vy), H->w = NMAD(iA, fcross(rA, P) + imp[2], H->w);
                B->vx = MAD(mB, P.x, B->vx), B->vy = MAD(mB, P.y, B->vy);
#ifdef CRL_NO_RB
                B->w = MAD(iB, imp[2], B->w);
#else
                B->w += iB * (vcross(rB, P) + imp[2]);
#endif
            } else {
                v2 Cdot = joint_rel_vel(H, B, rA, rB);
                v2 imp = solve22(j->m, vmul(-1.0f, Cdot));
                car->imp[w][0] += imp.x, car->imp[w][1] += imp.y;
                H->vx = NMAD(mA, imp.x, H->vx), H->vy = NMAD(mA, imp.y, H->vy), H->w = NMAD(iA, fcross(rA, imp), H->w);
                B->vx = MAD(mB, imp.x, B->vx), B->vy = MAD(mB, imp.y, B->vy);
#ifndef CRL_NO_RB
                B->w += iB * vcross(rB, imp);
#endif
            }
        }
}

/* integrate positions (with the translation / rotation clamps) */
static void isl_integrate_pos(car_state *car, float h) {
    ISL_CONSTS;
    car_body *all[5] = {H, &car->wheel[0], &car->wheel[1], &car->wheel[2], &car->wheel[3]};
    for (int k = 0; k < 5; k++) {
        car_body *b = all[k];
        v2 tr = V(h * b->vx, h * b->vy);
        if (fdot(tr, tr) > MAX_TRANSLATION * MAX_TRANSLATION) {
            float ratio = MAX_TRANSLATION / sqrtf(fdot(tr, tr));
            b->vx *= ratio, b->vy *= ratio;
        }
        float ro = h * b->w;
        if (ro * ro > MAX_ROTATION * MAX_ROTATION) b->w *= MAX_ROTATION / fabsf(ro);
        b->cx = MAD(h, b->vx, b->cx), b->cy = MAD(h, b->vy, b->cy), b->a = MAD(h, b->w, b->a);
    }
}

/* one position iteration: b2RevoluteJoint::SolvePositionConstraints for the 4 joints */
static int isl_joints_pos(car_state *car) {
    ISL_CONSTS;
    int ok = 1;
        for (int q = 0; q < 4; q++) {
            int w = 3 - q;
            car_body *B = &car->wheel[w];
            float angErr = 0;
            if (car->limit_state[w] != LIM_INACTIVE) {
                float angle = B->a - H->a - 0.0f, C, li;
                float mm = iA + iB;
                if (mm > 0.0f) mm = 1.0f / mm;
                if (car->limit_state[w] == LIM_LOWER) {
                    C = angle - LOWER_ANGLE, angErr = -C;
                    C = fminf(fmaxf(C + ANGULAR_SLOP, -MAX_ANGULAR_CORRECTION), 0.0f);
                } else {
                    C = angle - UPPER_ANGLE, angErr = C;
                    C = fminf(fmaxf(C - ANGULAR_SLOP, 0.0f), MAX_ANGULAR_CORRECTION);
                }
                li = -mm * C;
                H->a = NMAD(iA, li, H->a), B->a = MAD(iB, li, B->a);
            }
            float sA, cA;
            rot_sincosf(H->a, &sA, &cA);
            v2 rA = frot(sA, cA, vsub(V(K.anchor[w][0], K.anchor[w][1]), lcA));
            float k[3][3];
#ifdef CRL_NO_RB
            v2 C = vsub(vsub(V(B->cx, B->cy), V(H->cx, H->cy)), rA);
            k[0][0] = MAD(iA * rA.y, rA.y, mA + mB);
            k[0][1] = -iA * rA.x * rA.y;
            k[1][0] = k[0][1];
            k[1][1] = MAD(iA * rA.x, rA.x, mA + mB);
#else
            v2 rB = V(0, 0);
            v2 C = vsub(vsub(vadd(V(B->cx, B->cy), rB), V(H->cx, H->cy)), rA);
            k[0][0] = mA + mB + iA * rA.y * rA.y + iB * rB.y * rB.y;
            k[0][1] = -iA * rA.x * rA.y - iB * rB.x * rB.y;
            k[1][0] = k[0][1];
            k[1][1] = mA + mB + iA * rA.x * rA.x + iB * rB.x * rB.x;
#endif
            float posErr = sqrtf(fdot(C, C));
            v2 imp = vmul(-1.0f, solve22(k, C));
            H->cx = NMAD(mA, imp.x, H->cx), H->cy = NMAD(mA, imp.y, H->cy), H->a = NMAD(iA, fcross(rA, imp), H->a);
            B->cx = MAD(mB, imp.x, B->cx), B->cy = MAD(mB, imp.y, B->cy);
#ifndef CRL_NO_RB
            B->a += iB * vcross(rB, imp);
#endif
            ok &= posErr <= LINEAR_SLOP && angErr <= ANGULAR_SLOP;
        }
    return ok;
}

static void isl_clear_forces(car_state *car) {
    car->hull.fx = car->hull.fy = 0;
    for (int w = 0; w < 4; w++) car->wheel[w].fx = car->wheel[w].fy = 0;
}

/* End of b2Island::Solve (Box2D 2.3 b2Island.cpp, "if (allowSleep)"): a body slower than the sleep
 * tolerances accumulates m_sleepTime; once every body of the island has been still for
 * b2_timeToSleep and the position solver converged, the island is put to sleep, which zeroes the
 * velocities (b2Body::SetAwake(false)).  Car.step wakes every body again on the next step
 * (joint.motorSpeed assignment and ApplyForceToCenter(..., True), car_dynamics.py:159-234), so the
 * visible effect is the velocity reset. */
#define LIN_SLEEP_TOL 0.01f
#define ANG_SLEEP_TOL (2.0f / 180.0f * 3.14159265359f)
#define TIME_TO_SLEEP 0.5f
static float isl_sleep_scan(car_state *car, float h) {
    float min_sleep = 3.402823466e+38f;
    const float lin2 = LIN_SLEEP_TOL * LIN_SLEEP_TOL, ang2 = ANG_SLEEP_TOL * ANG_SLEEP_TOL;
    for (int b = 0; b < 5; b++) {
        car_body *B = b == 0 ? &car->hull : &car->wheel[b - 1];
        if (B->w * B->w > ang2 || B->vx * B->vx + B->vy * B->vy > lin2) {
            car->sleep_time[b] = 0.0f;
            min_sleep = 0.0f;
        } else {
            car->sleep_time[b] += h;
            min_sleep = fminf(min_sleep, car->sleep_time[b]);
        }
    }
    return min_sleep;
}
static void isl_put_to_sleep(car_state *car) {
    for (int b = 0; b < 5; b++) {
        car_body *B = b == 0 ? &car->hull : &car->wheel[b - 1];
        car->sleep_time[b] = 0.0f;
        B->vx = B->vy = B->w = 0.0f;
    }
}

#ifdef CRL_CYCLE_STATS
static long lone_pos_hist[64];
void car_oracle_lone_pos_hist(long *out) { memcpy(out, lone_pos_hist, sizeof(lone_pos_hist)); }
#endif
static void island_solve(car_state *car, float h, float dt_ratio, int vel_iters, int pos_iters) {
    joint_tmp jt[4];
    isl_integrate_vel(car, h);
    isl_joints_init(car, jt, dt_ratio);
    for (int it = 0; it < vel_iters; it++) isl_joints_vel(car, jt, h);
    isl_integrate_pos(car, h);
    int solved = 0;
#ifdef CRL_CYCLE_STATS
    int used = 0;
#endif
    for (int it = 0; it < pos_iters; it++) {
#ifdef CRL_CYCLE_STATS
        used = it + 1;
#endif
        if (isl_joints_pos(car)) {
            solved = 1;
            break;
        }
    }
#ifdef CRL_CYCLE_STATS
    lone_pos_hist[used]++; /* position iterations a car on its own needed (tools/cycle_stats.py) */
    if (!solved) lone_pos_hist[0]++;
    for (int w = 0; w < 4; w++) if (car->limit_state[w] != LIM_INACTIVE) { lone_pos_hist[61]++; break; }
#endif
    isl_clear_forces(car);
    if (isl_sleep_scan(car, h) >= TIME_TO_SLEEP && solved) isl_put_to_sleep(car);
}


/* ------------------------------------------------------------------ car-car contacts
 * b2CollidePolygons (manifold), b2ContactSolver (velocity: friction then normal with the 2-point
 * block solver; position: Baumgarte pseudo-impulses) restated from Box2D 2.3 [published
 * algorithm, not available here: UNPINNED].  A = a fixture of car 0, B = a fixture of car 1
 * (car 0's proxies are created first).  Wheels do not collide with wheels (category 0x20 / mask 1).
 * Friction = sqrt(0.2 * 0.2), restitution = 0. */
typedef struct { float s, c; v2 p; } xform;
static inline v2 xmul(xform t, v2 v) { return vadd(rot(t.s, t.c, v), t.p); }
static inline v2 qmulT(xform t, v2 v) { return V(t.c * v.x + t.s * v.y, -t.s * v.x + t.c * v.y); }
static inline v2 xmulT(xform t, v2 v) { return qmulT(t, vsub(v, t.p)); }

typedef struct { const float (*v)[2]; int n; v2 nrm[8]; v2 centroid; } shape;
static shape SH[8]; /* 0-3 hull polygons, 4-7 the wheel box */
static int SH_ready = 0;
static void shapes_init(void) {
    if (SH_ready) return;
    car_oracle_consts();
    for (int f = 0; f < 8; f++) {
        SH[f].v = f < 4 ? K.hull_poly[f] : K.wheel_poly, SH[f].n = f < 4 ? K.hull_n[f] : 4;
        for (int i = 0; i < SH[f].n; i++) {
            int j = i + 1 < SH[f].n ? i + 1 : 0;
            v2 e = V(SH[f].v[j][0] - SH[f].v[i][0], SH[f].v[j][1] - SH[f].v[i][1]);
            v2 n = V(e.y, -e.x); /* b2Cross(edge, 1) */
            float len = sqrtf(vdot(n, n));
            SH[f].nrm[i] = vmul(1.0f / len, n);
        }
        { /* b2PolygonShape::Set -> ComputeCentroid (b2PolygonShape.cpp): area-weighted, reference point = origin */
            v2 c = V(0, 0);
            float area = 0;
            for (int i = 0; i < SH[f].n; i++) {
                v2 p2 = V(SH[f].v[i][0], SH[f].v[i][1]), p3 = i + 1 < SH[f].n ? V(SH[f].v[i + 1][0], SH[f].v[i + 1][1]) : V(SH[f].v[0][0], SH[f].v[0][1]);
                float D = vcross(p2, p3), ta = 0.5f * D;
                area += ta;
                c = vadd(c, vmul(ta * (1.0f / 3.0f), vadd(p2, p3)));
            }
            SH[f].centroid = vmul(1.0f / area, c);
        }
    }
    SH_ready = 1;
}

typedef struct { car_body *b; float im, ii; v2 lc; } bref;
static bref body_of(car_env *e, int car, int fixture) {
    bref r;
    if (fixture < 4) r.b = &e->car[car].hull, r.im = K.hull_inv_mass, r.ii = K.hull_inv_I, r.lc = V(K.hull_lc[0], K.hull_lc[1]);
    else r.b = &e->car[car].wheel[fixture - 4], r.im = K.wheel_inv_mass, r.ii = K.wheel_inv_I, r.lc = V(0, 0);
    return r;
}
static xform xf_of(const bref *r) {
    xform t;
    rot_sincosf(r->b->a, &t.s, &t.c);
    t.p = vsub(V(r->b->cx, r->b->cy), rot(t.s, t.c, r->lc));
    return t;
}
/* the same inside the position iterations (the scope of the contracted arithmetic, see MAD) */
static xform xf_of_f(const bref *r) {
    xform t;
    rot_sincosf(r->b->a, &t.s, &t.c);
    t.p = vsub(V(r->b->cx, r->b->cy), frot(t.s, t.c, r->lc));
    return t;
}
static inline v2 xmul_f(xform t, v2 v) { return vadd(frot(t.s, t.c, v), t.p); }

/* b2FindMaxSeparation: which published form?  box2d-py ~=2.3.5 is not in the reference tree, and Box2D changed this function
 * between 2.3.0 and 2.3.1; DESIGN.md section 9 tabulates what is assumed per function.  CRL_B2_COLLIDE selects (compile time):
 *   0 (default, = the HIP kernels): exhaustive search over poly1's edges, dot products in the WORLD frame, flip rule 0.98 / 0.001
 *   1  Box2D 2.3.1+ as published: exhaustive search in poly2's frame (b2MulT(xf2, xf1)), flip rule separationB > separationA + 0.1 linearSlop
 *   2  Box2D 2.3.0 as published: hill climb from the edge facing the other centroid (b2EdgeSeparation), flip rule 0.98 / 0.001
 * The three agree on every manifold except where two edges' separations tie to the last bit or the flip rule is at its threshold;
 * tests/test_oracle_car_physics.py counts the differences over a contact-rich soak (CPU only; builds: make -C oracle variants). */
#ifndef CRL_B2_COLLIDE
#define CRL_B2_COLLIDE 0
#endif
#if CRL_B2_COLLIDE == 1
static float max_separation(int *edge, const shape *p1, xform x1, const shape *p2, xform x2) {
    xform xf; /* b2MulT(xf2, xf1) */
    xf.s = x2.c * x1.s - x2.s * x1.c, xf.c = x2.c * x1.c + x2.s * x1.s;
    xf.p = qmulT(x2, vsub(x1.p, x2.p));
    float best = -3.4e38f;
    int bi = 0;
    for (int i = 0; i < p1->n; i++) {
        v2 n = rot(xf.s, xf.c, p1->nrm[i]), v1 = xmul(xf, V(p1->v[i][0], p1->v[i][1]));
        float si = 3.4e38f;
        for (int j = 0; j < p2->n; j++) {
            float sij = vdot(n, vsub(V(p2->v[j][0], p2->v[j][1]), v1));
            if (sij < si) si = sij;
        }
        if (si > best) best = si, bi = i;
    }
    *edge = bi;
    return best;
}
#elif CRL_B2_COLLIDE == 2
static float edge_separation(const shape *p1, xform x1, int edge1, const shape *p2, xform x2) { /* b2EdgeSeparation */
    v2 normal1World = rot(x1.s, x1.c, p1->nrm[edge1]), normal1 = qmulT(x2, normal1World);
    int index = 0;
    float minDot = 3.4e38f;
    for (int i = 0; i < p2->n; i++) {
        float d = vdot(V(p2->v[i][0], p2->v[i][1]), normal1);
        if (d < minDot) minDot = d, index = i;
    }
    v2 v1 = xmul(x1, V(p1->v[edge1][0], p1->v[edge1][1])), v2_ = xmul(x2, V(p2->v[index][0], p2->v[index][1]));
    return vdot(vsub(v2_, v1), normal1World);
}
static float max_separation(int *edgeIndex, const shape *p1, xform x1, const shape *p2, xform x2) {
    const int count1 = p1->n;
    v2 d = vsub(xmul(x2, p2->centroid), xmul(x1, p1->centroid)), dLocal1 = qmulT(x1, d);
    int edge = 0;
    float maxDot = -3.4e38f;
    for (int i = 0; i < count1; i++) {
        float dt = vdot(p1->nrm[i], dLocal1);
        if (dt > maxDot) maxDot = dt, edge = i;
    }
    float s = edge_separation(p1, x1, edge, p2, x2);
    int prevEdge = edge - 1 >= 0 ? edge - 1 : count1 - 1;
    float sPrev = edge_separation(p1, x1, prevEdge, p2, x2);
    int nextEdge = edge + 1 < count1 ? edge + 1 : 0;
    float sNext = edge_separation(p1, x1, nextEdge, p2, x2);
    int bestEdge, increment;
    float bestSeparation;
    if (sPrev > s && sPrev > sNext) increment = -1, bestEdge = prevEdge, bestSeparation = sPrev;
    else if (sNext > s) increment = 1, bestEdge = nextEdge, bestSeparation = sNext;
    else {
        *edgeIndex = edge;
        return s;
    }
    for (;;) {
        if (increment == -1) edge = bestEdge - 1 >= 0 ? bestEdge - 1 : count1 - 1;
        else edge = bestEdge + 1 < count1 ? bestEdge + 1 : 0;
        s = edge_separation(p1, x1, edge, p2, x2);
        if (s > bestSeparation) bestEdge = edge, bestSeparation = s;
        else break;
    }
    *edgeIndex = bestEdge;
    return bestSeparation;
}
#else
/* exhaustive form, world frame: best edge of poly1 against poly2's vertices */
static float max_separation(int *edge, const shape *p1, xform x1, const shape *p2, xform x2) {
    float best = -3.4e38f;
    int bi = 0;
    for (int i = 0; i < p1->n; i++) {
        v2 n = rot(x1.s, x1.c, p1->nrm[i]), v1 = xmul(x1, V(p1->v[i][0], p1->v[i][1]));
        float si = 3.4e38f;
        for (int j = 0; j < p2->n; j++) {
            float sij = vdot(n, vsub(xmul(x2, V(p2->v[j][0], p2->v[j][1])), v1));
            if (sij < si) si = sij;
        }
        if (si > best) best = si, bi = i;
    }
    *edge = bi;
    return best;
}
#endif

typedef struct { v2 v; uint32_t id; } clipv;
#define MKID(ia, ib, ta, tb) ((uint32_t)(ia) | ((uint32_t)(ib) << 8) | ((uint32_t)(ta) << 16) | ((uint32_t)(tb) << 24))
static int clip_segment(clipv out[2], const clipv in[2], v2 normal, float offset, int vertexIndexA) {
    int n = 0;
    float d0 = vdot(normal, in[0].v) - offset, d1 = vdot(normal, in[1].v) - offset;
    if (d0 <= 0.0f) out[n++] = in[0];
    if (d1 <= 0.0f) out[n++] = in[1];
    if (d0 * d1 < 0.0f) {
        float interp = d0 / (d0 - d1);
        out[n].v = vadd(in[0].v, vmul(interp, vsub(in[1].v, in[0].v)));
        out[n].id = MKID(vertexIndexA, (in[0].id >> 8) & 255u, 0 /*e_vertex*/, 1 /*e_face*/);
        n++;
    }
    return n;
}

/* b2CollidePolygons -> manifold in `c` (count = 0 when not touching) */
static void collide_polygons(car_contact *c, const shape *pa, xform xa, const shape *pb, xform xb) {
    c->count = 0;
    const float totalRadius = 0.02f;
    int edgeA, edgeB;
    float sepA = max_separation(&edgeA, pa, xa, pb, xb);
    if (sepA > totalRadius) return;
    float sepB = max_separation(&edgeB, pb, xb, pa, xa);
    if (sepB > totalRadius) return;
    const shape *p1, *p2;
    xform x1, x2;
    int edge1, flip;
#if CRL_B2_COLLIDE == 1
    const int flip_rule = sepB > sepA + 0.1f * 0.005f; /* k_tol = 0.1f * b2_linearSlop */
#else
    const int flip_rule = sepB > 0.98f * sepA + 0.001f; /* k_relativeTol, k_absoluteTol */
#endif
    if (flip_rule) p1 = pb, p2 = pa, x1 = xb, x2 = xa, edge1 = edgeB, c->type = 1, flip = 1;
    else p1 = pa, p2 = pb, x1 = xa, x2 = xb, edge1 = edgeA, c->type = 0, flip = 0;
    /* b2FindIncidentEdge */
    clipv inc[2];
    {
        v2 n1 = qmulT(x2, rot(x1.s, x1.c, p1->nrm[edge1]));
        int idx = 0;
        float mind = 3.4e38f;
        for (int i = 0; i < p2->n; i++) {
            float d = vdot(n1, p2->nrm[i]);
            if (d < mind) mind = d, idx = i;
        }
        int i1 = idx, i2 = i1 + 1 < p2->n ? i1 + 1 : 0;
        inc[0].v = xmul(x2, V(p2->v[i1][0], p2->v[i1][1])), inc[0].id = MKID(edge1, i1, 1, 0);
        inc[1].v = xmul(x2, V(p2->v[i2][0], p2->v[i2][1])), inc[1].id = MKID(edge1, i2, 1, 0);
    }
    int iv1 = edge1, iv2 = edge1 + 1 < p1->n ? edge1 + 1 : 0;
    v2 v11 = V(p1->v[iv1][0], p1->v[iv1][1]), v12 = V(p1->v[iv2][0], p1->v[iv2][1]);
    v2 lt = vsub(v12, v11);
    lt = vmul(1.0f / sqrtf(vdot(lt, lt)), lt);
    v2 ln = V(lt.y, -lt.x), planePoint = vmul(0.5f, vadd(v11, v12));
    v2 tangent = rot(x1.s, x1.c, lt), normal = V(tangent.y, -tangent.x);
    v11 = xmul(x1, v11), v12 = xmul(x1, v12);
    float frontOffset = vdot(normal, v11);
    float side1 = -vdot(tangent, v11) + totalRadius, side2 = vdot(tangent, v12) + totalRadius;
    clipv c1[2], c2[2];
    if (clip_segment(c1, inc, vmul(-1.0f, tangent), side1, iv1) < 2) return;
    if (clip_segment(c2, c1, tangent, side2, iv2) < 2) return;
    c->ln[0] = ln.x, c->ln[1] = ln.y, c->lp[0] = planePoint.x, c->lp[1] = planePoint.y;
    int n = 0;
    for (int i = 0; i < 2; i++) {
        float sep = vdot(normal, c2[i].v) - frontOffset;
        if (sep <= totalRadius) {
            v2 lpt = xmulT(x2, c2[i].v);
            c->pt[n][0] = lpt.x, c->pt[n][1] = lpt.y;
            uint32_t id = c2[i].id;
            if (flip) id = MKID((id >> 8) & 255u, id & 255u, (id >> 24) & 255u, (id >> 16) & 255u);
            c->id[n] = id;
            n++;
        }
    }
    c->count = n;
}

/* b2ContactManager::Collide for the car-car pairs: new manifolds, impulses carried by id */
static void collide_cars(car_env *e) {
    shapes_init();
    car_contact old[CAR_MAX_CONTACTS];
    int n_old = e->n_contact;
    memcpy(old, e->contact, sizeof(old));
    e->n_contact = 0;
    /* cheap reject: hull centres more than two car lengths apart */
    float dx = e->car[0].hull.cx - e->car[1].hull.cx, dy = e->car[0].hull.cy - e->car[1].hull.cy;
    if (dx * dx + dy * dy > 12.0f * 12.0f) return;
    for (int fa = 0; fa < 8; fa++)
        for (int fb = 0; fb < 8; fb++) {
            if (fa >= 4 && fb >= 4) continue; /* wheel-wheel filtered out */
            bref A = body_of(e, 0, fa), B = body_of(e, 1, fb);
            car_contact c;
            memset(&c, 0, sizeof(c));
            c.pair = fa * 8 + fb;
            collide_polygons(&c, &SH[fa], xf_of(&A), &SH[fb], xf_of(&B));
            if (c.count == 0 || e->n_contact >= CAR_MAX_CONTACTS) continue;
            for (int k = 0; k < n_old; k++)
                if (old[k].pair == c.pair)
                    for (int i = 0; i < c.count; i++)
                        for (int j = 0; j < old[k].count; j++)
                            if (old[k].id[j] == c.id[i]) c.nimp[i] = old[k].nimp[j], c.timp[i] = old[k].timp[j];
            e->contact[e->n_contact++] = c;
        }
}

typedef struct {
    bref A, B;
    v2 normal, rA[2], rB[2];
    float nmass[2], tmass[2], bias[2], K[2][2], invK[2][2];
    int count;
} contact_vc;

static void world_manifold(const car_contact *c, xform xa, xform xb, v2 *normal, v2 pts[2]) {
    const float rA = 0.01f, rB = 0.01f;
    if (c->type == 0) {
        v2 n = rot(xa.s, xa.c, V(c->ln[0], c->ln[1])), plane = xmul(xa, V(c->lp[0], c->lp[1]));
        for (int i = 0; i < c->count; i++) {
            v2 clip = xmul(xb, V(c->pt[i][0], c->pt[i][1]));
            v2 cA = vadd(clip, vmul(rA - vdot(vsub(clip, plane), n), n)), cB = vsub(clip, vmul(rB, n));
            pts[i] = vmul(0.5f, vadd(cA, cB));
        }
        *normal = n;
    } else {
        v2 n = rot(xb.s, xb.c, V(c->ln[0], c->ln[1])), plane = xmul(xb, V(c->lp[0], c->lp[1]));
        for (int i = 0; i < c->count; i++) {
            v2 clip = xmul(xa, V(c->pt[i][0], c->pt[i][1]));
            v2 cB = vadd(clip, vmul(rB - vdot(vsub(clip, plane), n), n)), cA = vsub(clip, vmul(rA, n));
            pts[i] = vmul(0.5f, vadd(cA, cB));
        }
        *normal = vmul(-1.0f, n);
    }
}

static inline v2 bvel(const car_body *b) { return V(b->vx, b->vy); }

/* b2ContactSolver::InitializeVelocityConstraints + WarmStart */
static void contacts_init(car_env *e, contact_vc *vc, float dt_ratio) {
    for (int k = 0; k < e->n_contact; k++) {
        car_contact *c = &e->contact[k];
        contact_vc *q = &vc[k];
        q->A = body_of(e, 0, c->pair >> 3), q->B = body_of(e, 1, c->pair & 7);
        q->count = c->count;
        const float mA = q->A.im, iA = q->A.ii, mB = q->B.im, iB = q->B.ii;
        v2 pts[2];
        world_manifold(c, xf_of(&q->A), xf_of(&q->B), &q->normal, pts);
        v2 cA = V(q->A.b->cx, q->A.b->cy), cB = V(q->B.b->cx, q->B.b->cy);
        v2 tangent = V(q->normal.y, -q->normal.x);
        for (int j = 0; j < c->count; j++) {
            c->nimp[j] *= dt_ratio, c->timp[j] *= dt_ratio;
            q->rA[j] = vsub(pts[j], cA), q->rB[j] = vsub(pts[j], cB);
            float rnA = vcross(q->rA[j], q->normal), rnB = vcross(q->rB[j], q->normal);
            float kN = mA + mB + iA * rnA * rnA + iB * rnB * rnB;
            q->nmass[j] = kN > 0.0f ? 1.0f / kN : 0.0f;
            float rtA = vcross(q->rA[j], tangent), rtB = vcross(q->rB[j], tangent);
            float kT = mA + mB + iA * rtA * rtA + iB * rtB * rtB;
            q->tmass[j] = kT > 0.0f ? 1.0f / kT : 0.0f;
            q->bias[j] = 0.0f; /* restitution 0 */
        }
        if (q->count == 2) {
            float rn1A = vcross(q->rA[0], q->normal), rn1B = vcross(q->rB[0], q->normal);
            float rn2A = vcross(q->rA[1], q->normal), rn2B = vcross(q->rB[1], q->normal);
            float k11 = mA + mB + iA * rn1A * rn1A + iB * rn1B * rn1B, k22 = mA + mB + iA * rn2A * rn2A + iB * rn2B * rn2B;
            float k12 = mA + mB + iA * rn1A * rn2A + iB * rn1B * rn2B;
            if (k11 * k11 < 1000.0f * (k11 * k22 - k12 * k12)) {
                q->K[0][0] = k11, q->K[0][1] = k12, q->K[1][0] = k12, q->K[1][1] = k22; /* K[col][row] */
                float det = k11 * k22 - k12 * k12;
                if (det != 0.0f) det = 1.0f / det;
                q->invK[0][0] = det * k22, q->invK[1][0] = -det * k12, q->invK[0][1] = -det * k12, q->invK[1][1] = det * k11;
            } else {
                q->count = 1;
            }
        }
    }
    for (int k = 0; k < e->n_contact; k++) { /* WarmStart */
        car_contact *c = &e->contact[k];
        contact_vc *q = &vc[k];
        v2 tangent = V(q->normal.y, -q->normal.x);
        for (int j = 0; j < q->count; j++) {
            v2 P = vadd(vmul(c->nimp[j], q->normal), vmul(c->timp[j], tangent));
            q->A.b->w -= q->A.ii * vcross(q->rA[j], P), q->A.b->vx -= q->A.im * P.x, q->A.b->vy -= q->A.im * P.y;
            q->B.b->w += q->B.ii * vcross(q->rB[j], P), q->B.b->vx += q->B.im * P.x, q->B.b->vy += q->B.im * P.y;
        }
    }
}

static void apply_imp(contact_vc *q, int j, v2 P) {
    car_body *A = q->A.b, *B = q->B.b;
    A->vx = NMAD(q->A.im, P.x, A->vx), A->vy = NMAD(q->A.im, P.y, A->vy), A->w = NMAD(q->A.ii, fcross(q->rA[j], P), A->w);
    B->vx = MAD(q->B.im, P.x, B->vx), B->vy = MAD(q->B.im, P.y, B->vy), B->w = MAD(q->B.ii, fcross(q->rB[j], P), B->w);
}
static v2 rel_vel(const contact_vc *q, int j) { /* ((vB + wB x rB) - vA) - wA x rA,  w x r = (-w r.y, w r.x) */
    const car_body *A = q->A.b, *B = q->B.b;
    const v2 rA = q->rA[j], rB = q->rB[j];
    return V(NMAD(-A->w, rA.y, NMAD(B->w, rB.y, B->vx) - A->vx), NMAD(A->w, rA.x, MAD(B->w, rB.x, B->vy) - A->vy));
}

/* b2ContactSolver::SolveVelocityConstraints, one iteration */
static void contacts_vel(car_env *e, contact_vc *vc) {
    const float friction = sqrtf(0.2f * 0.2f);
    for (int k = 0; k < e->n_contact; k++) {
        car_contact *c = &e->contact[k];
        contact_vc *q = &vc[k];
        v2 normal = q->normal, tangent = V(normal.y, -normal.x);
        for (int j = 0; j < q->count; j++) { /* friction first */
            float vt = fdot(rel_vel(q, j), tangent);
            float maxF = friction * c->nimp[j];
            float ni = MAD(q->tmass[j], -vt, c->timp[j]);
            ni = ni < -maxF ? -maxF : ni > maxF ? maxF : ni;
            float lambda = ni - c->timp[j];
            c->timp[j] = ni;
            apply_imp(q, j, vmul(lambda, tangent));
        }
        if (q->count == 1) {
            float vn = fdot(rel_vel(q, 0), normal);
            float ni = fmaxf(MAD(-q->nmass[0], vn - q->bias[0], c->nimp[0]), 0.0f);
            float lambda = ni - c->nimp[0];
            c->nimp[0] = ni;
            apply_imp(q, 0, vmul(lambda, normal));
        } else if (q->count == 2) { /* block solver */
            v2 a = V(c->nimp[0], c->nimp[1]);
            float vn1 = fdot(rel_vel(q, 0), normal), vn2 = fdot(rel_vel(q, 1), normal);
            v2 b = V(vn1 - q->bias[0], vn2 - q->bias[1]);
            b = vsub(b, V(MAD(q->K[0][0], a.x, q->K[1][0] * a.y), MAD(q->K[0][1], a.x, q->K[1][1] * a.y)));
            v2 x;
            int solved = 0;
            x = V(-MAD(q->invK[0][0], b.x, q->invK[1][0] * b.y), -MAD(q->invK[0][1], b.x, q->invK[1][1] * b.y));
            if (x.x >= 0.0f && x.y >= 0.0f) solved = 1;
            if (!solved) {
                x = V(-q->nmass[0] * b.x, 0.0f);
                vn2 = MAD(q->K[0][1], x.x, b.y);
                if (x.x >= 0.0f && vn2 >= 0.0f) solved = 1;
            }
            if (!solved) {
                x = V(0.0f, -q->nmass[1] * b.y);
                vn1 = MAD(q->K[1][0], x.y, b.x);
                if (x.y >= 0.0f && vn1 >= 0.0f) solved = 1;
            }
            if (!solved) {
                x = V(0.0f, 0.0f);
                if (b.x >= 0.0f && b.y >= 0.0f) solved = 1;
            }
            if (solved) {
                v2 d = vsub(x, a);
                v2 P1 = vmul(d.x, normal), P2 = vmul(d.y, normal);
                car_body *A = q->A.b, *B = q->B.b;
                A->vx = NMAD(q->A.im, P1.x + P2.x, A->vx), A->vy = NMAD(q->A.im, P1.y + P2.y, A->vy);
                A->w = NMAD(q->A.ii, fcross(q->rA[0], P1) + fcross(q->rA[1], P2), A->w);
                B->vx = MAD(q->B.im, P1.x + P2.x, B->vx), B->vy = MAD(q->B.im, P1.y + P2.y, B->vy);
                B->w = MAD(q->B.ii, fcross(q->rB[0], P1) + fcross(q->rB[1], P2), B->w);
                c->nimp[0] = x.x, c->nimp[1] = x.y;
            }
        }
    }
}

/* b2ContactSolver::SolvePositionConstraints, one iteration; returns minSeparation >= -3*slop */
#ifdef CRL_CYCLE_STATS
static long pos_turn_hist[2][5];
void car_oracle_pos_turn_hist(long *out) { memcpy(out, pos_turn_hist, sizeof(pos_turn_hist)); }
#endif
static int contacts_pos(car_env *e) {
    float minSep = 0.0f;
    for (int k = 0; k < e->n_contact; k++) {
        car_contact *c = &e->contact[k];
        bref A = body_of(e, 0, c->pair >> 3), B = body_of(e, 1, c->pair & 7);
        for (int j = 0; j < c->count; j++) {
            xform xa = xf_of_f(&A), xb = xf_of_f(&B);
            v2 normal, point;
            float sep;
            if (c->type == 0) {
                normal = frot(xa.s, xa.c, V(c->ln[0], c->ln[1]));
                v2 plane = xmul_f(xa, V(c->lp[0], c->lp[1])), clip = xmul_f(xb, V(c->pt[j][0], c->pt[j][1]));
                sep = fdot(vsub(clip, plane), normal) - 0.01f - 0.01f, point = clip;
            } else {
                normal = frot(xb.s, xb.c, V(c->ln[0], c->ln[1]));
                v2 plane = xmul_f(xb, V(c->lp[0], c->lp[1])), clip = xmul_f(xa, V(c->pt[j][0], c->pt[j][1]));
                sep = fdot(vsub(clip, plane), normal) - 0.01f - 0.01f, point = clip;
                normal = vmul(-1.0f, normal);
            }
            v2 rA = vsub(point, V(A.b->cx, A.b->cy)), rB = vsub(point, V(B.b->cx, B.b->cy));
            if (sep < minSep) minSep = sep;
            float C = fminf(fmaxf(0.2f * (sep + LINEAR_SLOP), -0.2f), 0.0f);
            float rnA = fcross(rA, normal), rnB = fcross(rB, normal);
            float Kn = MAD(B.ii * rnB, rnB, MAD(A.ii * rnA, rnA, A.im + B.im));
            float impulse = Kn > 0.0f ? -C / Kn : 0.0f;
            v2 P = vmul(impulse, normal);
#ifdef CRL_CYCLE_STATS
            const float a0A = A.b->a, a0B = B.b->a;
#endif
            A.b->cx = NMAD(A.im, P.x, A.b->cx), A.b->cy = NMAD(A.im, P.y, A.b->cy), A.b->a = NMAD(A.ii, fcross(rA, P), A.b->a);
            B.b->cx = MAD(B.im, P.x, B.b->cx), B.b->cy = MAD(B.im, P.y, B.b->cy), B.b->a = MAD(B.ii, fcross(rB, P), B.b->a);
#ifdef CRL_CYCLE_STATS
            for (int side = 0; side < 2; side++) { /* how far a contact point turns a body: [0] hull, [1] wheel; bins: == 0, < 2^-10, < 2^-5, < 2^-2, >= 2^-2 */
                const float dd = fabsf(side ? B.b->a - a0B : A.b->a - a0A);
                const int wheel = (side ? (c->pair & 7) : (c->pair >> 3)) >= 4;
                pos_turn_hist[wheel][dd == 0.0f ? 0 : dd < 0x1p-10f ? 1 : dd < 0x1p-5f ? 2 : dd < 0x1p-2f ? 3 : 4]++;
            }
#endif
        }
    }
    return minSep >= -3.0f * LINEAR_SLOP;
}

/* b2Island::Solve for the two cars joined by touching contacts: contacts are initialised and
 * warm-started before the joints; each velocity iteration solves joints (car 1, car 0) then
 * contacts; each position iteration contacts then joints. */
#ifdef CRL_CYCLE_STATS
/* Experiment (tools/cycle_stats.py, `make -C oracle cyc`; docs/LAB_NOTES_r05.md): does the iteration state of a touching island
 * come back to an EARLIER state bit for bit (a cycle of period <= 8)?  Then every later iteration is known without running it.
 * stats: [0] islands, [1 + p] islands whose velocity iterations enter a period-p cycle (p = 1..8), [10] sum of the iteration
 * index where it is first seen, [11] islands that run all 60 position iterations, [12 + p] of those, the ones whose position
 * iterations enter a period-p cycle, [21] sum of the iteration index, [22 + k] histogram of position iterations used (k = 1..60),
 * [90 + nc] islands by manifold count; [100..] the same block again for the islands with >= 2 manifolds */
static long cyc_stats[256];
void car_oracle_cycle_stats(long *out) { memcpy(out, cyc_stats, sizeof(cyc_stats)); }
static int cyc_snap_vel(const car_env *e, float *b) {
    int n = 0;
    for (int c = 0; c < 2; c++) {
        const car_state *q = &e->car[c];
        b[n++] = q->hull.vx, b[n++] = q->hull.vy, b[n++] = q->hull.w;
        for (int w = 0; w < 4; w++) {
            b[n++] = q->wheel[w].vx, b[n++] = q->wheel[w].vy, b[n++] = q->wheel[w].w;
            b[n++] = q->imp[w][0], b[n++] = q->imp[w][1], b[n++] = q->imp[w][2], b[n++] = q->motor_imp[w];
        }
    }
    for (int k = 0; k < e->n_contact; k++)
        for (int j = 0; j < 2; j++) b[n++] = e->contact[k].nimp[j], b[n++] = e->contact[k].timp[j];
    return n;
}
static int cyc_snap_pos(const car_env *e, float *b) {
    int n = 0;
    for (int c = 0; c < 2; c++) {
        const car_state *q = &e->car[c];
        b[n++] = q->hull.cx, b[n++] = q->hull.cy, b[n++] = q->hull.a;
        for (int w = 0; w < 4; w++) b[n++] = q->wheel[w].cx, b[n++] = q->wheel[w].cy, b[n++] = q->wheel[w].a;
    }
    return n;
}
#endif

static void island_solve_coupled(car_env *e, float h, float dt_ratio, int vel_iters, int pos_iters) {
    joint_tmp jt[2][4];
    contact_vc vc[CAR_MAX_CONTACTS];
    isl_integrate_vel(&e->car[1], h), isl_integrate_vel(&e->car[0], h);
    contacts_init(e, vc, dt_ratio);
    isl_joints_init(&e->car[1], jt[1], dt_ratio), isl_joints_init(&e->car[0], jt[0], dt_ratio);
#ifdef CRL_CYCLE_STATS
    float ring[9][128];
    int found_p = 0, found_it = 0;
    const int blk = e->n_contact >= 2 ? 100 : 0;
#endif
    for (int it = 0; it < vel_iters; it++) {
        isl_joints_vel(&e->car[1], jt[1], h), isl_joints_vel(&e->car[0], jt[0], h);
        contacts_vel(e, vc);
#ifdef CRL_CYCLE_STATS
        const int nw = cyc_snap_vel(e, ring[it % 9]);
        for (int p = 1; p <= 8 && p <= it && !found_p; p++)
            if (memcmp(ring[it % 9], ring[(it - p) % 9], nw * sizeof(float)) == 0) found_p = p, found_it = it;
#endif
    }
#ifdef CRL_CYCLE_STATS
#pragma omp critical
    {
        cyc_stats[0]++, cyc_stats[90 + (e->n_contact > 8 ? 8 : e->n_contact)]++;
        if (blk) cyc_stats[100]++;
        if (found_p) {
            cyc_stats[1 + found_p]++, cyc_stats[10] += found_it;
            if (blk) cyc_stats[blk + 1 + found_p]++, cyc_stats[blk + 10] += found_it;
        }
    }
    found_p = found_it = 0;
    int used = pos_iters;
#endif
    isl_integrate_pos(&e->car[1], h), isl_integrate_pos(&e->car[0], h);
    int solved = 0;
    for (int it = 0; it < pos_iters; it++) {
        int cok = contacts_pos(e);
        int j1 = isl_joints_pos(&e->car[1]), j0 = isl_joints_pos(&e->car[0]);
#ifdef CRL_CYCLE_STATS
        const int nw = cyc_snap_pos(e, ring[it % 9]);
        for (int p = 1; p <= 8 && p <= it && !found_p; p++)
            if (memcmp(ring[it % 9], ring[(it - p) % 9], nw * sizeof(float)) == 0) found_p = p, found_it = it;
        if (cok && j1 && j0) used = it + 1;
#endif
        if (cok && j1 && j0) {
            solved = 1;
            break;
        }
    }
#ifdef CRL_CYCLE_STATS
#pragma omp critical
    {
        cyc_stats[22 + used]++;
        if (blk) cyc_stats[blk + 22 + used]++;
        if (!solved) {
            cyc_stats[11]++;
            if (blk) cyc_stats[blk + 11]++;
            if (found_p) {
                cyc_stats[12 + found_p]++, cyc_stats[21] += found_it;
                if (blk) cyc_stats[blk + 12 + found_p]++, cyc_stats[blk + 21] += found_it;
            }
        }
    }
#endif
    isl_clear_forces(&e->car[1]), isl_clear_forces(&e->car[0]);
    const float m1 = isl_sleep_scan(&e->car[1], h), m0 = isl_sleep_scan(&e->car[0], h);
    if (fminf(m1, m0) >= TIME_TO_SLEEP && solved) isl_put_to_sleep(&e->car[1]), isl_put_to_sleep(&e->car[0]);
}

/* ------------------------------------------------------------------ env */
static void tiles_to_f32(car_env *e) {
    for (int t = 0; t < e->trk.n; t++) {
        /* b2PolygonShape::Set -> CCW hull; the 5-gon is convex */
        v2 tmp[5];
        make_ccw(e->trk.tile[t], 5, 1.0, tmp);
        float x0 = 3.4e38f, y0 = 3.4e38f, x1 = -3.4e38f, y1 = -3.4e38f;
        for (int k = 0; k < 5; k++) {
            e->tile32[t][k][0] = tmp[k].x, e->tile32[t][k][1] = tmp[k].y;
            x0 = fminf(x0, tmp[k].x), y0 = fminf(y0, tmp[k].y), x1 = fmaxf(x1, tmp[k].x), y1 = fmaxf(y1, tmp[k].y);
        }
        e->tile_aabb[t][0] = x0, e->tile_aabb[t][1] = y0, e->tile_aabb[t][2] = x1, e->tile_aabb[t][3] = y1;
    }
}

/* CarRacing.reset (crmp:454-525) with an explicit draw stream: `u` holds 24 uniforms per
 * track attempt (attempts are consumed until one succeeds), `shuffle_swap` is the outcome
 * of np.random.shuffle on [0, 1] (1 = swapped).  Returns the number of attempts used, or
 * -1 when max_attempts were exhausted. */
int car_oracle_reset(car_env *e, const double *u, int max_attempts, int shuffle_swap) {
    car_oracle_consts();
    int att = 0, ok = 0;
    while (att < max_attempts && !ok) ok = car_oracle_create_track(u + 24 * att++, &e->trk);
    if (!ok) return -1;
    tiles_to_f32(e);
    int birth[2] = {shuffle_swap ? 1 : 0, shuffle_swap ? 0 : 1};
    for (int k = 0; k < 2; k++) {
        car_oracle_place(&e->car[k], e->trk.track[0][1], e->trk.track[0][2], e->trk.track[0][3], birth[k]);
        e->reward[k] = e->prev_reward[k] = 0, e->tile_visited_count[k] = 0, e->done[k] = 0, e->last_block[k] = -1;
    }
    memset(e->visited, 0, sizeof(e->visited));
    memset(e->wheel_tiles, 0, sizeof(e->wheel_tiles));
    e->t = 0, e->step_count = 0, e->inv_dt0 = 0.0f;
    e->n_contact = 0;
    return att;
}

static void wheel_world_poly(const car_body *b, v2 *out) {
    float s, c;
    rot_sincosf(b->a, &s, &c);
    for (int k = 0; k < 4; k++) out[k] = vadd(rot(s, c, V(K.wheel_poly[k][0], K.wheel_poly[k][1])), V(b->cx, b->cy));
}

/* FrictionDetector._contact (crmp:111-153) driven by sensor overlap at the transforms the
 * step starts from (b2ContactManager::Collide before the island solve). */
/* One Begin/EndContact between wheel w of car c and tile t (crmp:111-153). */
void car_oracle_contact_event(car_env *e, int c, int w, int t, int begin) {
    if (begin) {
        e->wheel_tiles[c][w][t >> 5] |= 1u << (t & 31);
        if (!((e->visited[c][t >> 5] >> (t & 31)) & 1)) {
            int last_blk = e->last_block[c] < 0 ? 0 : e->last_block[c];
            if (t - last_blk < 50) {
                e->last_block[c] = t;
                e->reward[c] += 1000.0 / e->trk.n;
            } /* else: the reference raises (self.verbose missing on the listener, crmp:146); no reward */
            e->visited[c][t >> 5] |= 1u << (t & 31);
            e->tile_visited_count[c] += 1;
        }
    } else {
        e->wheel_tiles[c][w][t >> 5] &= ~(1u << (t & 31));
    }
}

static void collide(car_env *e) {
    const float R2 = (0.02f + 10.0f * 1.1920929e-07f);
    for (int c = 0; c < 2; c++) {
        v2 wp[4][4];
        float bb4[4][4];
        for (int w = 0; w < 4; w++) {
            wheel_world_poly(&e->car[c].wheel[w], wp[w]);
            float x0 = 3.4e38f, y0 = 3.4e38f, x1 = -3.4e38f, y1 = -3.4e38f;
            for (int k = 0; k < 4; k++)
                x0 = fminf(x0, wp[w][k].x), y0 = fminf(y0, wp[w][k].y), x1 = fmaxf(x1, wp[w][k].x), y1 = fmaxf(y1, wp[w][k].y);
            bb4[w][0] = x0, bb4[w][1] = y0, bb4[w][2] = x1, bb4[w][3] = y1;
        }
        /* events are raised tile-major (tile 0..n-1, wheels 0..3 within a tile); Box2D's own
         * order is its contact-list order, which is not reproducible here */
        for (int t = 0; t < e->trk.n; t++)
            for (int w = 0; w < 4; w++) {
                int was = (e->wheel_tiles[c][w][t >> 5] >> (t & 31)) & 1, now = 0;
                const float *bb = e->tile_aabb[t];
                if (!(bb4[w][0] > bb[2] + 0.05f || bb4[w][2] < bb[0] - 0.05f || bb4[w][1] > bb[3] + 0.05f || bb4[w][3] < bb[1] - 0.05f)) {
                    float d2 = poly_dist2(wp[w], 4, (const v2 *)e->tile32[t], 5);
                    now = d2 < R2 * R2;
                }
                if (now != was) car_oracle_contact_event(e, c, w, t, now);
            }
    }
}

int car_oracle_wheel_on_road(const car_env *e, int c, int w) {
    for (int k = 0; k < CAR_MAX_TILES / 32; k++)
        if (e->wheel_tiles[c][w][k]) return 1;
    return 0;
}

/* CarRacing.step (crmp:542-620), action_repeat = 1.  actions[c] = (steer, gas/brake) or
 * NULL for the action-less step that reset() ends with. */
void car_oracle_step_repeat(car_env *e, const double (*actions)[2], int repeat, double step_reward[2], int done[2]) {
    const double dt = 1.0 / FPS;
    step_reward[0] = step_reward[1] = 0.0;
    if (actions) {
        for (int c = 0; c < 2; c++) {
            double a[3];
            car_oracle_process_action(actions[c], a);
            car_oracle_controls(&e->car[c], -a[0], a[1], a[2]);
        }
      for (int rep = 0; rep < repeat; rep++) { /* action repetition (crmp:576) */
        for (int c = 0; c < 2; c++) {
            if (e->done[c]) continue;
            car_state *car = &e->car[c];
            for (int w = 0; w < 4; w++) { /* Car.step */
                car_body *B = &car->wheel[w];
                double ms, f[2];
                double ja = (double)(B->a - car->hull.a - 0.0f);
                float qs, qc;
                rot_sincosf(B->a, &qs, &qc);
                car_oracle_wheel(dt, car->steer[w], car->gas[w], car->brake[w], ja, (double)qs, (double)qc, (double)B->vx, (double)B->vy,
                                 car_oracle_wheel_on_road(e, c, w), &car->omega[w], &car->phase[w], &ms, f);
                car->motor_speed[w] = (float)ms;
                B->fx += (float)f[0], B->fy += (float)f[1];
            }
            e->reward[c] -= 0.1 / repeat;
            step_reward[c] += e->reward[c] - e->prev_reward[c];
            e->prev_reward[c] = e->reward[c];
            float s, co;
            v2 p;
            body_xf(&car->hull, V(K.hull_lc[0], K.hull_lc[1]), &s, &co, &p);
            if (e->tile_visited_count[c] == e->trk.n) e->done[c] = 1;
            if (fabs((double)p.x) > PLAYFIELD || fabs((double)p.y) > PLAYFIELD) e->done[c] = 1;
            if (e->step_count > 1000) e->done[c] = 1;
        }
        /* world.Step(1/FPS, 6*30, 2*30) */
        float h = (float)dt;
        float dt_ratio = e->inv_dt0 * h;
        collide(e);
        if (e->contacts_enabled) collide_cars(e);
        if (e->contacts_enabled && e->n_contact > 0) {
            island_solve_coupled(e, h, dt_ratio, 180, 60);
        } else {
            island_solve(&e->car[1], h, dt_ratio, 180, 60);
            island_solve(&e->car[0], h, dt_ratio, 180, 60);
        }
        e->inv_dt0 = 1.0f / h;
        e->t += dt;
        e->step_count += 1;
      }
    }
    done[0] = e->done[0], done[1] = e->done[1];
}

void car_oracle_step(car_env *e, const double (*actions)[2], double step_reward[2], int done[2]) {
    car_oracle_step_repeat(e, actions, 1, step_reward, done);
}

/* n envs stepped at once over the host's cores (the long parity soaks of tests/: thousands of envs per step) */
void car_oracle_step_batch(car_env *e, long n, const double *actions /*[n][2][2]*/, double *step_reward /*[n][2]*/, int32_t *done /*[n][2]*/) {
    car_oracle_consts();
    shapes_init();
#pragma omp parallel for schedule(dynamic, 4)
    for (long i = 0; i < n; i++) {
        int d[2];
        car_oracle_step(&e[i], actions ? (const double(*)[2])(actions + 4 * i) : 0, step_reward + 2 * i, d);
        done[2 * i] = d[0], done[2 * i + 1] = d[1];
    }
}

/* world.Step alone: Collide of the two cars + the island solve(s), WITHOUT Car.step (the forces the bodies carry are applied as
 * they are, the joint motor targets stay): the physics property tests of tests/test_oracle_car_physics.py (momentum through a
 * collision with the tyre forces off) */
void car_oracle_world_step(car_env *e) {
    car_oracle_consts();
    shapes_init();
    const float h = (float)(1.0 / FPS), dt_ratio = e->inv_dt0 * h;
    if (e->contacts_enabled) collide_cars(e);
    if (e->contacts_enabled && e->n_contact > 0) {
        island_solve_coupled(e, h, dt_ratio, 180, 60);
    } else {
        island_solve(&e->car[1], h, dt_ratio, 180, 60);
        island_solve(&e->car[0], h, dt_ratio, 180, 60);
    }
    e->inv_dt0 = 1.0f / h;
}

/* b2ContactManager::Collide for the two cars of each env, from the poses it holds: the manifolds (without stepping) */
void car_oracle_collide_batch(car_env *e, long n) {
    car_oracle_consts();
    shapes_init();
#pragma omp parallel for schedule(dynamic, 16)
    for (long i = 0; i < n; i++) collide_cars(&e[i]);
}
int car_oracle_collide_variant(void) { return CRL_B2_COLLIDE; }
/* 1: this build solves the islands in the contracted arithmetic (-DCRL_FMA, see MAD above) */
int car_oracle_fma(void) {
#ifdef CRL_FMA
    return 1;
#else
    return 0;
#endif
}

void car_oracle_hull_position(const car_env *e, int c, float out[3]) {
    float s, co;
    v2 p;
    body_xf(&e->car[c].hull, V(K.hull_lc[0], K.hull_lc[1]), &s, &co, &p);
    out[0] = p.x, out[1] = p.y, out[2] = e->car[c].hull.a;
}

int car_oracle_env_size(void) { return (int)sizeof(car_env); }

/* ------------------------------------------------------------------ observation raster
 * CarRacing.get_observation (crmp:622-634), restated the way the reference computes it:
 *   reset:  render_road_for_observation_map (:732-755, called at :519) pre-rasters grass, the
 *           range(-20, 20, 2) squares and every road / border polygon with pygame.draw.polygon
 *           into the 10000 x 10000 `observation_playground` at obs_scale px per world unit;
 *   step:   camera_update("rgb_array") :791-804, camera_view :764-789 = 192 x 192 subsurface at the
 *           int-truncated camera pixel -> pygame.transform.rotate (nearest neighbour, 16.16 fixed
 *           point) -> blit so that the rotated centre lands on (48, 48); then Car.draw_for_pygame
 *           (cd:284-298) for every car, render_indicators_for_pygame :645-670, and the luma
 *           0.299 R + 0.587 G + 0.114 B truncated to uint8.
 * pygame 1.9.6 (setup.py:7) is a third-party dependency that is neither vendored nor installable
 * here: its draw_fillpoly / drawhorzlineclip (draw.c), surf_rotate / rotate / rotate90
 * (transform.c), Rect and blit argument conversion (truncation toward zero) are restated from the
 * published source [from memory]; PARITY AGAINST REAL pygame UNPINNED.  What IS pinned: the
 * reference's own Python around those calls (which polygons, in which order and colours, the
 * camera, the crop rectangle, the car transforms, the indicator geometry) through
 * tests/golden/car_obs.npz, recorded from the reference's get_observation running over a
 * pygame stand-in that implements the same restated primitives.
 *
 * The map is kept as a 7-colour palette (one byte per pixel here) of the WINDOW
 * [org, org + w)^2 of the 10000^2 surface; everything outside the window is grass as long as no
 * polygon leaves it (car_oracle_build_map returns the number of span pixels it had to drop:
 * 0 for every track, tests assert it; org = 0, w = 10000 is the reference's whole surface).
 * The 5-px reward text comes from pre-baked 1-bit strings (car_oracle_set_text). */
#define G_GRASS 161
#define G_LIGHT 176
#define G_WHITE 255
#define G_RED 76
#define G_OWN 60
#define G_OTHER 29
#define G_BLUE 29
#define G_ABS_REAR 44
#define G_GREEN 149
static const uint8_t G_ROAD[3] = {101, 103, 107};
enum { P_GRASS = 0, P_LIGHT = 1, P_ROAD0 = 2, P_WHITE = 5, P_RED = 6 };
/* luma of (102,204,102), (102,229,102), road 102 / 104 / 107, (255,255,255), (255,0,0): each
 * 0.299 R + 0.587 G + 0.114 B in float64, truncated (crmp:631-633) */
static const uint8_t PAL_GRAY[8] = {G_GRASS, G_LIGHT, 101, 103, 107, G_WHITE, G_RED, 0};
#define MAP_SURFACE 10000 /* self.world_size (crmp:216) */

typedef struct { uint8_t *px; int org, w; long dropped; } map_win;

/* pygame draw.c drawhorzlineclip on the 10000^2 surface (clip rect = whole surface), restricted to the window */
static void map_hline(map_win *m, int x1, int y, int x2, uint8_t v) {
    if (y < 0 || y >= MAP_SURFACE) return;
    if (x2 < x1) { int t = x1; x1 = x2; x2 = t; }
    if (x1 < 0) x1 = 0;
    if (x2 > MAP_SURFACE - 1) x2 = MAP_SURFACE - 1;
    if (x2 < 0 || x1 >= MAP_SURFACE) return;
    for (int x = x1; x <= x2; x++) {
        const int wx = x - m->org, wy = y - m->org;
        if (wx < 0 || wy < 0 || wx >= m->w || wy >= m->w) { m->dropped++; continue; }
        m->px[(size_t)wy * m->w + wx] = v;
    }
}

static int cmp_int(const void *a, const void *b) { return *(const int *)a - *(const int *)b; }

/* pygame draw.c draw_fillpoly, scanline by scanline */
static void map_fillpoly(map_win *m, const int *vx, const int *vy, int n, uint8_t v) {
    int miny = vy[0], maxy = vy[0], xs[16];
    for (int i = 1; i < n; i++) {
        if (vy[i] < miny) miny = vy[i];
        if (vy[i] > maxy) maxy = vy[i];
    }
    if (miny == maxy) { /* "Special case: polygon only 1 pixel high." */
        int minx = vx[0], maxx = vx[0];
        for (int i = 1; i < n; i++) {
            if (vx[i] < minx) minx = vx[i];
            if (vx[i] > maxx) maxx = vx[i];
        }
        map_hline(m, minx, miny, maxx, v);
        return;
    }
    for (int y = miny; y <= maxy; y++) {
        int ints = 0;
        for (int i = 0; i < n; i++) {
            const int ind1 = i ? i - 1 : n - 1, ind2 = i;
            int y1 = vy[ind1], y2 = vy[ind2], x1, x2;
            if (y1 < y2) x1 = vx[ind1], x2 = vx[ind2];
            else if (y1 > y2) y2 = vy[ind1], y1 = vy[ind2], x2 = vx[ind1], x1 = vx[ind2];
            else continue;
            if ((y >= y1 && y < y2) || (y == maxy && y > y1 && y <= y2)) xs[ints++] = (y - y1) * (x2 - x1) / (y2 - y1) + x1;
        }
        qsort(xs, ints, sizeof(int), cmp_int);
        for (int i = 0; i + 1 < ints; i += 2) map_hline(m, xs[i], y, xs[i + 1], v);
    }
}

static double obs_scale(void) { return (10 / (100 / sqrt(96.0))) * 1.8; } /* crmp:214-215 */

/* One vertex of a map polygon: (obs_scale * -v + world_size / 2) as a Python float, truncated by
 * pygame's pg_TwoIntsFromObj (crmp:745-753) */
static int map_coord(double v) { return (int)(obs_scale() * -v + MAP_SURFACE / 2.0); }

/* render_road_for_observation_map (crmp:732-755) into map[w * w] (palette indices); returns the
 * number of polygon pixels that fell outside the window */
long car_oracle_build_map(const car_env *e, uint8_t *map, int org, int w) {
    map_win m = {map, org, w, 0};
    memset(map, P_GRASS, (size_t)w * w); /* screen.fill((0.4 * 255, 0.8 * 255, 0.4 * 255)) */
    const double k = PLAYFIELD / 20.0;
    for (int x = -20; x < 20; x += 2)
        for (int y = -20; y < 20; y += 2) {
            const double sq[4][2] = {{k * x + k, k * y + 0}, {k * x + 0, k * y + 0}, {k * x + 0, k * y + k}, {k * x + k, k * y + k}};
            int vx[4], vy[4];
            for (int i = 0; i < 4; i++) vx[i] = map_coord(sq[i][0]), vy[i] = map_coord(sq[i][1]);
            map_fillpoly(&m, vx, vy, 4, P_LIGHT);
        }
    /* road_poly in the order _create_track appends it (crmp:400-441): for i = n-1 .. 0 the tile, then its border */
    for (int i = e->trk.n - 1; i >= 0; i--) {
        int vx[5], vy[5];
        for (int j = 0; j < 5; j++) vx[j] = map_coord(e->trk.tile[i][j][0]), vy[j] = map_coord(e->trk.tile[i][j][1]);
        map_fillpoly(&m, vx, vy, 5, (uint8_t)(P_ROAD0 + i % 3)); /* 255 * (0.4 + 0.01 * (i % 3)) -> 102, 104, 107 */
        if (e->trk.border[i]) {
            for (int j = 0; j < 4; j++) vx[j] = map_coord(e->trk.border_poly[i][j][0]), vy[j] = map_coord(e->trk.border_poly[i][j][1]);
            map_fillpoly(&m, vx, vy, 4, i % 2 == 0 ? P_WHITE : P_RED);
        }
    }
    return m.dropped;
}

/* integer map-space vertices of the polygons of build_map, for the HIP side's tests: out[i][0..4] tile, [5..8] border */
void car_oracle_map_vertices(const car_env *e, int32_t *out /*[n][9][2]*/) {
    for (int i = 0; i < e->trk.n; i++) {
        for (int j = 0; j < 5; j++) out[(i * 9 + j) * 2] = map_coord(e->trk.tile[i][j][0]), out[(i * 9 + j) * 2 + 1] = map_coord(e->trk.tile[i][j][1]);
        for (int j = 0; j < 4; j++) {
            out[(i * 9 + 5 + j) * 2] = e->trk.border[i] ? map_coord(e->trk.border_poly[i][j][0]) : 0;
            out[(i * 9 + 5 + j) * 2 + 1] = e->trk.border[i] ? map_coord(e->trk.border_poly[i][j][1]) : 0;
        }
    }
}

static uint8_t map_at(const uint8_t *map, int org, int w, int x, int y) {
    const int wx = x - org, wy = y - org;
    if (wx < 0 || wy < 0 || wx >= w || wy >= w) return P_GRASS;
    return map[(size_t)wy * w + wx];
}

/* pygame draw_fillpoly membership test for one pixel */
static int fillpoly_hit(const int *px, const int *py, int n, int x, int y) {
    int miny = py[0], maxy = py[0], minx = px[0], maxx = px[0];
    for (int i = 1; i < n; i++) {
        if (py[i] < miny) miny = py[i];
        if (py[i] > maxy) maxy = py[i];
        if (px[i] < minx) minx = px[i];
        if (px[i] > maxx) maxx = px[i];
    }
    if (y < miny || y > maxy) return 0;
    if (miny == maxy) return x >= minx && x <= maxx;
    int xs[16], k = 0;
    for (int i = 0; i < n; i++) {
        int ip = i ? i - 1 : n - 1;
        int y1 = py[ip], y2 = py[i], x1, x2;
        if (y1 < y2) x1 = px[ip], x2 = px[i];
        else if (y1 > y2) y2 = py[ip], y1 = py[i], x2 = px[ip], x1 = px[i];
        else continue;
        if ((y >= y1 && y < y2) || (y == maxy && y > y1 && y <= y2)) xs[k++] = (y - y1) * (x2 - x1) / (y2 - y1) + x1;
    }
    for (int i = 1; i < k; i++) /* insertion sort */
        for (int j = i; j > 0 && xs[j - 1] > xs[j]; j--) { int t = xs[j]; xs[j] = xs[j - 1]; xs[j - 1] = t; }
    for (int i = 0; i + 1 < k; i += 2)
        if (x >= xs[i] && x <= xs[i + 1]) return 1;
    return 0;
}

/* pygame.draw.rect(surface, color, (x, y, w, h)) with float arguments: int-truncated, then
 * filled as the polygon (l,t),(r,t),(r,b),(l,b) with r = x+w-1, b = y+h-1 (negative sizes
 * therefore fill "backwards") */
static void fill_rect(uint8_t *out, double x, double y, double w, double h, uint8_t g) {
    int l = (int)x, t = (int)y, r = (int)x + (int)w - 1, b = (int)y + (int)h - 1;
    int x0 = l < r ? l : r, x1 = l < r ? r : l, y0 = t < b ? t : b, y1 = t < b ? b : t;
    for (int yy = y0 > 0 ? y0 : 0; yy <= y1 && yy < 96; yy++)
        for (int xx = x0 > 0 ? x0 : 0; xx <= x1 && xx < 96; xx++) out[yy * 96 + xx] = g;
}

static const uint32_t *TEXT_BITS = 0; /* [3001][10] reward read-out bitmaps, or NULL */
void car_oracle_set_text(const uint32_t *bits) { TEXT_BITS = bits; }

typedef struct { double angle; float s, c; v2 off; double vx, vy; } camera;

/* camera_update("rgb_array") (crmp:791-804) */
static camera camera_of(const car_env *e, int viewer) {
    car_oracle_consts();
    const car_state *me = &e->car[viewer];
    camera cam;
    cam.angle = (double)me->hull.a;
    cam.vx = (double)me->hull.vx, cam.vy = (double)me->hull.vy;
    if (cam.vx * cam.vx + cam.vy * cam.vy > 0.5 * 0.5) cam.angle = m_atan2(-cam.vx, cam.vy);
    rot_sincosf((float)cam.angle, &cam.s, &cam.c); /* tmp.angle = angle: b2Rot(float32) */
    float hs, hc;
    v2 hp;
    body_xf(&me->hull, V(K.hull_lc[0], K.hull_lc[1]), &hs, &hc, &hp);
    cam.off = vadd(hp, V(cam.c * 0.0f - cam.s * 16.0f, cam.s * 0.0f + cam.c * 16.0f)); /* hull.position + tmp * (0, 16) */
    return cam;
}

/* cars (draw_for_pygame, cd:284-298), indicators (crmp:645-670), reward text: drawn over the background */
static void draw_overlays(const car_env *e, int viewer, const camera *cam, uint8_t *out) {
    const car_state *me = &e->car[viewer];
    const float s = cam->s, c = cam->c;
    const v2 off = cam->off;
    const double vx = cam->vx, vy = cam->vy;
    const float scale_f = (float)obs_scale();
    /* cars: car 0 then car 1; per car wheels then hull (cd:286-298) */
    for (int k = 0; k < 2; k++) {
        const car_state *car = &e->car[k];
        for (int part = 0; part < 8; part++) {
            const car_body *b = part < 4 ? &car->wheel[part] : &car->hull;
            int nv = part < 4 ? 4 : K.hull_n[part - 4];
            const float(*poly)[2] = part < 4 ? K.wheel_poly : K.hull_poly[part - 4];
            float bs, bc;
            v2 bp;
            body_xf(b, part < 4 ? V(0, 0) : V(K.hull_lc[0], K.hull_lc[1]), &bs, &bc, &bp);
            int px[8], py[8];
            for (int i = 0; i < nv; i++) {
                v2 wv = vadd(rot(bs, bc, V(poly[i][0], poly[i][1])), bp);
                v2 d = vsub(wv, off);
                v2 t = rot(-s, c, d); /* tmp.angle = -angle */
                float X = (-scale_f) * t.x + 48.0f, Y = (-scale_f) * t.y + 48.0f;
                px[i] = (int)X, py[i] = (int)Y;
            }
            uint8_t g = part < 4 ? 0 : (k == viewer ? G_OWN : G_OTHER);
            int x0 = 95, x1 = 0, y0 = 95, y1 = 0;
            for (int i = 0; i < nv; i++) {
                if (px[i] < x0) x0 = px[i];
                if (px[i] > x1) x1 = px[i];
                if (py[i] < y0) y0 = py[i];
                if (py[i] > y1) y1 = py[i];
            }
            for (int yy = y0 < 0 ? 0 : y0; yy <= y1 && yy < 96; yy++)
                for (int xx = x0 < 0 ? 0 : x0; xx <= x1 && xx < 96; xx++)
                    if (fillpoly_hit(px, py, nv, xx, yy)) out[yy * 96 + xx] = g;
        }
    }
    /* indicators (render_indicators_for_pygame, width = height = 96) */
    const double S = 96 / 40.0, Hh = 96 / 40.0;
    double true_speed = sqrt(vx * vx + vy * vy);
    fill_rect(out, 0, 96 - 4 * Hh, 96, 4 * Hh * 1000, 0);
    fill_rect(out, 5 * S, 96 - Hh, S, Hh * (-0.02 * true_speed), G_BLUE);
    for (int w = 0; w < 4; w++) fill_rect(out, (7 + w) * S, 96 - Hh, S, Hh * (-0.01 * me->omega[w]), w < 2 ? G_BLUE : G_ABS_REAR);
    double ja = (double)(me->wheel[0].a - me->hull.a - 0.0f);
    fill_rect(out, 20 * S, 96 - 2 * Hh, S * (10.0 * ja), 2 * Hh, G_GREEN);
    fill_rect(out, 30 * S, 96 - 2 * Hh, S * (0.8 * (double)me->hull.w), 2 * Hh, G_RED);
    /* draw_text("%05.0f" % rewards[viewer], 0.96, 91.2, 5-px font, not antialiased, white) */
    if (TEXT_BITS) {
        double r = e->reward[viewer], rr = rint(r);
        int idx = (int)rr - (-999);
        if (rr == 0.0 && (r < 0.0 || (r == 0.0 && signbit(r)))) idx = 3000;
        idx = idx < 0 ? 0 : idx > 3000 ? 3000 : idx;
        for (int row = 0; row < 10 && 91 + row < 96; row++)
            for (int col = 0; col < 32; col++)
                if ((TEXT_BITS[idx * 10 + row] >> col) & 1u) out[(91 + row) * 96 + col] = 255;
    }
}

/* camera_view(mode="rgb_array") (crmp:764-789): where each of the 96 x 96 screen pixels comes from.
 * src_xy[2 * p] / [2 * p + 1] = map pixel (on the 10000^2 surface) shown at screen pixel p, or
 * (-1, -1) where pygame's rotate writes its background colour (the subsurface's first pixel), or
 * (-2, -2) where the blit does not reach; rect_xy = top-left corner of the 192 x 192 subsurface. */
void car_oracle_view_sources(const car_env *e, int viewer, int32_t *src_xy, int32_t *rect_xy) {
    const camera cam = camera_of(e, viewer);
    const int W = 96, H = 96, SW = 2 * W, SH = 2 * H;
    const double pos0 = obs_scale() * -(double)cam.off.x + MAP_SURFACE / 2.0, pos1 = obs_scale() * -(double)cam.off.y + MAP_SURFACE / 2.0;
    const int rx = (int)(pos0 - W), ry = (int)(pos1 - H); /* pygame.Rect(pos[0] - width, pos[1] - height, 2 * width, 2 * height) */
    rect_xy[0] = rx, rect_xy[1] = ry;
    const float angle = (float)(57.295779513 * cam.angle); /* PyArg_ParseTuple "f" */
    if (!fmod((double)angle, (double)90.0f)) {             /* surf_rotate: rotate90(surf, (int)angle) */
        int numturns = ((int)angle / 90) % 4;
        if (numturns < 0) numturns = 4 + numturns;
        const int dw = (numturns % 2) ? SH : SW, dh = (numturns % 2) ? SW : SH;
        const int cx = dw >> 1, cy = dh >> 1; /* camera_view.get_rect().center */
        for (int Y = 0; Y < H; Y++)
            for (int X = 0; X < W; X++) {
                const int x = X - (-cx + W / 2), y = Y - (-cy + H / 2); /* screen.blit(camera_view, (-center[0] + width / 2, ...)) */
                int sx, sy;
                if (numturns == 0) sx = x, sy = y;
                else if (numturns == 1) sx = SW - 1 - y, sy = x;
                else if (numturns == 2) sx = SW - 1 - x, sy = SH - 1 - y;
                else sx = y, sy = SH - 1 - x;
                int32_t *o = src_xy + 2 * (Y * W + X);
                if (x < 0 || y < 0 || x >= dw || y >= dh) o[0] = o[1] = -2;
                else o[0] = rx + sx, o[1] = ry + sy;
            }
        return;
    }
    /* surf_rotate + rotate (pygame 1.9.6 transform.c) */
    const double radangle = angle * .01745329251994329;
    const double sangle = m_sin(radangle), cangle = m_cos(radangle);
    const double x = SW, y = SH, cxd = cangle * x, cyd = cangle * y, sxd = sangle * x, syd = sangle * y;
#define MAX2(a, b) ((a) > (b) ? (a) : (b))
    const int nxmax = (int)(MAX2(MAX2(MAX2(fabs(cxd + syd), fabs(cxd - syd)), fabs(-cxd + syd)), fabs(-cxd - syd)));
    const int nymax = (int)(MAX2(MAX2(MAX2(fabs(sxd + cyd), fabs(sxd - cyd)), fabs(-sxd + cyd)), fabs(-sxd - cyd)));
#undef MAX2
    const int dcy = nymax / 2;
    const int xd = (SW - nxmax) * 32768, yd = (SH - nymax) * 32768; /* ((src->w - dst->w) << 15) */
    const int isin = (int)(sangle * 65536), icos = (int)(cangle * 65536);
    const int ax = (nxmax << 15) - (int)(cangle * ((nxmax - 1) << 15));
    const int ay = (nymax << 15) - (int)(sangle * ((nxmax - 1) << 15));
    const int xmaxval = (SW << 16) - 1, ymaxval = (SH << 16) - 1;
    const int bx = -(nxmax >> 1) + W / 2, by = -(nymax >> 1) + H / 2; /* blit position of the rotated surface */
    for (int Y = 0; Y < H; Y++)
        for (int X = 0; X < W; X++) {
            const int xx = X - bx, yy = Y - by; /* pixel of the rotated surface */
            int32_t *o = src_xy + 2 * (Y * W + X);
            if (xx < 0 || yy < 0 || xx >= nxmax || yy >= nymax) {
                o[0] = o[1] = -2;
                continue;
            }
            const int dx = (ax + (isin * (dcy - yy))) + xd + icos * xx; /* row start, then xx steps of (icos, isin) */
            const int dy = (ay - (icos * (dcy - yy))) + yd + isin * xx;
            if (dx < 0 || dy < 0 || dx > xmaxval || dy > ymaxval) o[0] = o[1] = -1;
            else o[0] = rx + (dx >> 16), o[1] = ry + (dy >> 16);
        }
}

/* the observation of `viewer` from a map built by car_oracle_build_map */
void car_oracle_render(const car_env *e, const uint8_t *map, int org, int w, int viewer, uint8_t *out) {
    int32_t *src = (int32_t *)malloc(sizeof(int32_t) * 96 * 96 * 2), rect[2];
    car_oracle_view_sources(e, viewer, src, rect);
    for (int p = 0; p < 96 * 96; p++) {
        uint8_t pal;
        if (src[2 * p] == -1) pal = map_at(map, org, w, rect[0], rect[1]); /* bgcolor = first pixel of the subsurface */
        else if (src[2 * p] == -2) pal = 7;                                /* screen pixel the blit does not reach (black surface) */
        else pal = map_at(map, org, w, src[2 * p], src[2 * p + 1]);
        out[p] = PAL_GRAY[pal];
    }
    free(src);
    const camera cam = camera_of(e, viewer);
    draw_overlays(e, viewer, &cam, out);
}

/* Round-1/2 definition of the background, kept for comparison (tests report how many pixels it gets
 * differently): every pixel centre classified ANALYTICALLY against the track polygons in world space. */
void car_oracle_render_analytic(const car_env *e, int viewer, uint8_t *out) {
    const camera cam = camera_of(e, viewer);
    const float s = cam.s, c = cam.c;
    const v2 off = cam.off;
    const float inv_scale = (float)(1.0 / obs_scale());
    const float kf = (float)(PLAYFIELD / 20.0);
    for (int sy = 0; sy < 96; sy++)
        for (int sx = 0; sx < 96; sx++) {
            float dx = ((float)sx + 0.5f) - 48.0f, dy = ((float)sy + 0.5f) - 48.0f;
            float rx = c * dx - s * dy, ry = s * dx + c * dy;
            v2 pw = V(off.x - rx * inv_scale, off.y - ry * inv_scale);
            float fx = floorf(pw.x / kf), fy = floorf(pw.y / kf);
            int ix = (int)fx, iy = (int)fy;
            int light = ix >= -20 && ix <= 18 && iy >= -20 && iy <= 18 && (ix & 1) == 0 && (iy & 1) == 0;
            uint8_t g = light ? G_LIGHT : G_GRASS;
            for (int t = 0; t < e->trk.n; t++) {
                const float *bb = e->tile_aabb[t];
                if (e->trk.border[t]) {
                    v2 bp[4];
                    make_ccw(e->trk.border_poly[t], 4, 1.0, bp);
                    if (point_in_convex(pw, bp, 4)) { g = (t % 2 == 0) ? G_WHITE : G_RED; break; }
                }
                if (pw.x < bb[0] || pw.x > bb[2] || pw.y < bb[1] || pw.y > bb[3]) continue;
                if (point_in_convex(pw, (const v2 *)e->tile32[t], 5)) { g = G_ROAD[t % 3]; break; }
            }
            out[sy * 96 + sx] = g;
        }
    draw_overlays(e, viewer, &cam, out);
}

/* the double-precision functions this build uses, for tests/test_f64_math.py */
void car_oracle_f64(int fn, const double *a, const double *b, double *out, long n) {
    if (fn == 6 || fn == 7) { /* b2Rot's float32 sine / cosine (argument and result held in doubles) */
        for (long i = 0; i < n; i++) {
            float sn, cs;
            rot_sincosf((float)a[i], &sn, &cs);
            out[i] = fn == 6 ? sn : cs;
        }
        return;
    }
    for (long i = 0; i < n; i++)
        out[i] = fn == 0 ? m_sin(a[i]) : fn == 1 ? m_cos(a[i]) : fn == 2 ? m_atan2(a[i], b[i]) : fn == 3 ? t_sin(a[i]) : fn == 4 ? t_cos(a[i]) : t_atan2(a[i], b[i]);
}
