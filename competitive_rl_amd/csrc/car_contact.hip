// car_contact.hip -- the two cars of an env solved as ONE Box2D island when they can touch.
//
// Two lanes per coupled env (one per car; car_step_kernel flags an env as coupled when the cars'
// oriented boxes and then fixture boxes overlap, and compacts those envs into a list).  Restates, in float32, Box2D 2.3's b2CollidePolygons (reference-face clipping, <= 2
// manifold points, contact ids for warm starting) and b2ContactSolver (friction then normal with
// the 2-point block solver; Baumgarte position correction) around the same joint phases as the
// single-car island (car_solver.h).  Third-party algorithm, absent from the reference tree
// (box2d-py ~=2.3.5): parity unpinned; the CPU checker restates the same algorithm.
//
// A = a fixture of car 0, B = a fixture of car 1 (fixtures 0-3: hull polygons, 4-7: wheels; wheels do
// not collide with wheels).  friction = sqrt(0.2*0.2), restitution = 0, polygonRadius = 0.01.
#include "car_solver.h"

namespace crl {

struct XF {
    float s, c;
    V2 p;
};
__device__ inline V2 xmul(const XF &t, V2 v) { return rotv(t.s, t.c, v) + t.p; }
__device__ inline V2 qmulT(const XF &t, V2 v) { return mk(t.c * v.x + t.s * v.y, -t.s * v.x + t.c * v.y); }
__device__ inline V2 xmulT(const XF &t, V2 v) { return qmulT(t, v - t.p); }

struct Shape {
    const float (*v)[2];
    int n;
};
__device__ inline Shape shape_of(const CarConsts &K, int f) {
    Shape s;
    s.v = f < 4 ? K.hull_poly[f] : K.wheel_poly, s.n = f < 4 ? K.hull_n[f] : 4;
    return s;
}
__device__ inline V2 shape_vertex(const Shape &s, int i) { return mk(s.v[i][0], s.v[i][1]); }
__device__ inline V2 shape_normal(const Shape &s, int i) {
    const int j = i + 1 < s.n ? i + 1 : 0;
    const V2 e = mk(s.v[j][0] - s.v[i][0], s.v[j][1] - s.v[i][1]);
    const V2 n = mk(e.y, -e.x);
    const float len = sqrtf(dot(n, n));
    return (1.0f / len) * n;
}

struct Contact {  // persisted manifold (car_device.h: kContactWords floats per contact)
    int pair, count, type;
    float ln[2], lp[2], pt[2][2];
    uint32_t id[2];
    float nimp[2], timp[2];
};

struct BRef {
    Body *b;
    float im, ii;
    V2 lc;
};
__device__ inline BRef body_of(CarRegs &c, const CarConsts &K, int fixture) {
    BRef r;
    if (fixture < 4) r.b = &c.H, r.im = K.hull_inv_mass, r.ii = K.hull_inv_I, r.lc = mk(K.hull_lc[0], K.hull_lc[1]);
    else r.b = &c.W[fixture - 4], r.im = K.wheel_inv_mass, r.ii = K.wheel_inv_I, r.lc = mk(0.f, 0.f);
    return r;
}
__device__ inline XF xf_of(const BRef &r) {
    XF t;
    crl_sincosf(r.b->a, &t.s, &t.c);
    t.p = mk(r.b->cx, r.b->cy) - rotv(t.s, t.c, r.lc);
    return t;
}

__device__ inline float max_separation(int &edge, const Shape &p1, const XF &x1, const Shape &p2, const XF &x2) {
    float best = -3.4e38f;
    int bi = 0;
    for (int i = 0; i < p1.n; i++) {
        const V2 n = rotv(x1.s, x1.c, shape_normal(p1, i)), v1 = xmul(x1, shape_vertex(p1, i));
        float si = 3.4e38f;
        for (int j = 0; j < p2.n; j++) {
            const float sij = dot(n, xmul(x2, shape_vertex(p2, j)) - v1);
            if (sij < si) si = sij;
        }
        if (si > best) best = si, bi = i;
    }
    edge = bi;
    return best;
}

struct ClipV {
    V2 v;
    uint32_t id;
};
__device__ inline uint32_t mkid(uint32_t ia, uint32_t ib, uint32_t ta, uint32_t tb) { return ia | (ib << 8) | (ta << 16) | (tb << 24); }

__device__ inline int clip_segment(ClipV out[2], const ClipV in[2], V2 normal, float offset, int vertexIndexA) {
    int n = 0;
    const float d0 = dot(normal, in[0].v) - offset, d1 = dot(normal, in[1].v) - offset;
    if (d0 <= 0.0f) out[n++] = in[0];
    if (d1 <= 0.0f) out[n++] = in[1];
    if (d0 * d1 < 0.0f) {
        const float interp = d0 / (d0 - d1);
        out[n].v = in[0].v + interp * (in[1].v - in[0].v);
        out[n].id = mkid((uint32_t)vertexIndexA, (in[0].id >> 8) & 255u, 0u, 1u);
        n++;
    }
    return n;
}

__device__ void collide_polygons(Contact &c, const Shape &pa, const XF &xa, const Shape &pb, const XF &xb) {
    c.count = 0;
    const float totalRadius = 0.02f;
    int edgeA, edgeB;
    const float sepA = max_separation(edgeA, pa, xa, pb, xb);
    if (sepA > totalRadius) return;
    const float sepB = max_separation(edgeB, pb, xb, pa, xa);
    if (sepB > totalRadius) return;
    const bool flip = sepB > 0.98f * sepA + 0.001f;
    const Shape &p1 = flip ? pb : pa, &p2 = flip ? pa : pb;
    const XF &x1 = flip ? xb : xa, &x2 = flip ? xa : xb;
    const int edge1 = flip ? edgeB : edgeA;
    c.type = flip ? 1 : 0;
    ClipV inc[2];
    {
        const V2 n1 = qmulT(x2, rotv(x1.s, x1.c, shape_normal(p1, edge1)));
        int idx = 0;
        float mind = 3.4e38f;
        for (int i = 0; i < p2.n; i++) {
            const float d = dot(n1, shape_normal(p2, i));
            if (d < mind) mind = d, idx = i;
        }
        const int i1 = idx, i2 = i1 + 1 < p2.n ? i1 + 1 : 0;
        inc[0].v = xmul(x2, shape_vertex(p2, i1)), inc[0].id = mkid((uint32_t)edge1, (uint32_t)i1, 1u, 0u);
        inc[1].v = xmul(x2, shape_vertex(p2, i2)), inc[1].id = mkid((uint32_t)edge1, (uint32_t)i2, 1u, 0u);
    }
    const int iv1 = edge1, iv2 = edge1 + 1 < p1.n ? edge1 + 1 : 0;
    V2 v11 = shape_vertex(p1, iv1), v12 = shape_vertex(p1, iv2);
    V2 lt = v12 - v11;
    lt = (1.0f / sqrtf(dot(lt, lt))) * lt;
    const V2 ln = mk(lt.y, -lt.x), planePoint = 0.5f * (v11 + v12);
    const V2 tangent = rotv(x1.s, x1.c, lt), normal = mk(tangent.y, -tangent.x);
    v11 = xmul(x1, v11), v12 = xmul(x1, v12);
    const float frontOffset = dot(normal, v11);
    const float side1 = -dot(tangent, v11) + totalRadius, side2 = dot(tangent, v12) + totalRadius;
    ClipV c1[2], c2[2];
    if (clip_segment(c1, inc, -1.0f * tangent, side1, iv1) < 2) return;
    if (clip_segment(c2, c1, tangent, side2, iv2) < 2) return;
    c.ln[0] = ln.x, c.ln[1] = ln.y, c.lp[0] = planePoint.x, c.lp[1] = planePoint.y;
    int n = 0;
    for (int i = 0; i < 2; i++) {
        const float sep = dot(normal, c2[i].v) - frontOffset;
        if (sep <= totalRadius) {
            const V2 lpt = xmulT(x2, c2[i].v);
            c.pt[n][0] = lpt.x, c.pt[n][1] = lpt.y;
            uint32_t id = c2[i].id;
            if (flip) id = mkid((id >> 8) & 255u, id & 255u, (id >> 24) & 255u, (id >> 16) & 255u);
            c.id[n] = id;
            n++;
        }
    }
    c.count = n;
}

struct ContactVC {
    V2 normal, rA[2], rB[2];
    float nmass[2], tmass[2], K[2][2], invK[2][2];
    int count;
};

__device__ inline void world_manifold(const Contact &c, const XF &xa, const XF &xb, V2 &normal, V2 pts[2]) {
    const float rA = 0.01f, rB = 0.01f;
    if (c.type == 0) {
        const V2 n = rotv(xa.s, xa.c, mk(c.ln[0], c.ln[1])), plane = xmul(xa, mk(c.lp[0], c.lp[1]));
        for (int i = 0; i < c.count; i++) {
            const V2 clip = xmul(xb, mk(c.pt[i][0], c.pt[i][1]));
            const V2 cA = clip + (rA - dot(clip - plane, n)) * n, cB = clip - rB * n;
            pts[i] = 0.5f * (cA + cB);
        }
        normal = n;
    } else {
        const V2 n = rotv(xb.s, xb.c, mk(c.ln[0], c.ln[1])), plane = xmul(xb, mk(c.lp[0], c.lp[1]));
        for (int i = 0; i < c.count; i++) {
            const V2 clip = xmul(xa, mk(c.pt[i][0], c.pt[i][1]));
            const V2 cB = clip + (rB - dot(clip - plane, n)) * n, cA = clip - rA * n;
            pts[i] = 0.5f * (cA + cB);
        }
        normal = -1.0f * n;
    }
}

__device__ inline V2 bvel(const Body &b) { return mk(b.vx, b.vy); }
__device__ inline void apply_imp(const BRef &A, const BRef &B, V2 rA, V2 rB, V2 P) {
    A.b->vx -= A.im * P.x, A.b->vy -= A.im * P.y, A.b->w -= A.ii * cross(rA, P);
    B.b->vx += B.im * P.x, B.b->vy += B.im * P.y, B.b->w += B.ii * cross(rB, P);
}
__device__ inline V2 rel_vel(const BRef &A, const BRef &B, V2 rA, V2 rB) {
    return ((bvel(*B.b) + scross(B.b->w, rB)) - bvel(*A.b)) - scross(A.b->w, rA);
}

__global__ __launch_bounds__(64) void car_coupled_kernel(CarSoA s, CarConsts K) {
    // dense over the compacted list of coupled envs: a workgroup holds 111 KB of LDS, so the ones
    // past the end of the list must leave at once (the frames of the other envs are being drawn
    // on the same CUs meanwhile)
    // TWO LANES PER ENV: lane 2p holds car 0 of pair p, lane 2p+1 car 1.  The joints of a car only couple its
    // own hull and wheels, so each lane solves its car's joints in registers, both cars at once, exactly like
    // the per-car kernel.  Only the contacts pick their bodies by fixture index at run time: they are solved
    // by the even lane on the LDS copy of the two cars, and the lanes exchange velocities (positions in the
    // position phase) with that copy around every contact pass.  LDS operations of a wavefront complete in
    // order, so the pair needs no barrier.
    const int pair = threadIdx.x >> 1, me = threadIdx.x & 1;
    const int slot = blockIdx.x * 32 + pair;
    if (slot >= *s.coupled_count) return;
    const int64_t env = s.coupled_list[slot];
    const int64_t M = 2 * s.n;
    __shared__ CarRegs sh_car[32][2];
    __shared__ Contact sh_ct[32][kMaxContacts];
    __shared__ ContactVC sh_vc[32][kMaxContacts];
    __shared__ int sh_nc[32];
    CarRegs(&car)[2] = sh_car[pair];
    {
        const int64_t ci = me * s.n + env;
        load_car(s, M, ci, car[me]);
        for (int w = 0; w < 4; w++) car[me].fx[w] = s.wforce[(2 * w + 0) * M + ci], car[me].fy[w] = s.wforce[(2 * w + 1) * M + ci];
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    const int first_step = s.first_step[env];
    const float h = (float)(1.0 / CAR_FPS);
    const float dt_ratio = first_step ? 0.0f : (1.0f / h) * h;

    // ---- Collide: manifolds of the 48 fixture pairs, impulses carried over by contact id
    Contact *ct = sh_ct[pair];
    int nc = 0;
    if (me == 0) {  // the even lane runs the narrow phase for the pair
    // bounding circle of every fixture (world centre, radius): rejects most of the 48 pairs cheaply;
    // circles that do not overlap cannot be within the 0.02 contact margin
    float fcx[2][8], fcy[2][8], frad[2][8];
    for (int k = 0; k < 2; k++)
        for (int f = 0; f < 8; f++) {
            const Shape sh = shape_of(K, f);
            V2 ctr = mk(0.f, 0.f);
            for (int i = 0; i < sh.n; i++) ctr = ctr + shape_vertex(sh, i);
            ctr = (1.0f / sh.n) * ctr;
            float r2 = 0.f;
            for (int i = 0; i < sh.n; i++) r2 = fmaxf(r2, dot(shape_vertex(sh, i) - ctr, shape_vertex(sh, i) - ctr));
            const V2 wc = xmul(xf_of(body_of(car[k], K, f)), ctr);
            fcx[k][f] = wc.x, fcy[k][f] = wc.y, frad[k][f] = sqrtf(r2) + 0.03f;
        }
    {
        float *old = s.contact + env * (int64_t)(kMaxContacts * kContactWords);
        const int n_old = s.n_contact[env];
        for (int fa = 0; fa < 8; fa++)
            for (int fb = 0; fb < 8; fb++) {
                if (fa >= 4 && fb >= 4) continue;
                {
                    const float dx = fcx[0][fa] - fcx[1][fb], dy = fcy[0][fa] - fcy[1][fb], rr = frad[0][fa] + frad[1][fb];
                    if (dx * dx + dy * dy > rr * rr) continue;
                }
                const BRef A = body_of(car[0], K, fa), B = body_of(car[1], K, fb);
                Contact c;
                c.pair = fa * 8 + fb, c.type = 0;
                for (int i = 0; i < 2; i++) c.nimp[i] = c.timp[i] = 0.f, c.id[i] = 0u, c.pt[i][0] = c.pt[i][1] = 0.f;
                c.ln[0] = c.ln[1] = c.lp[0] = c.lp[1] = 0.f;
                collide_polygons(c, shape_of(K, fa), xf_of(A), shape_of(K, fb), xf_of(B));
                if (c.count == 0 || nc >= kMaxContacts) continue;
                for (int k = 0; k < n_old; k++) {
                    const float *o = old + k * kContactWords;
                    if (__float_as_int(o[0]) != c.pair) continue;
                    const int ocount = __float_as_int(o[1]);
                    for (int i = 0; i < c.count; i++)
                        for (int j = 0; j < ocount; j++)
                            if (__float_as_uint(o[11 + j]) == c.id[i]) c.nimp[i] = o[13 + j], c.timp[i] = o[15 + j];
                }
                ct[nc++] = c;
            }
    }

        sh_nc[pair] = nc;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    nc = sh_nc[pair];

    float slp[5];  // b2Body::m_sleepTime of this lane's car
    for (int b = 0; b < 5; b++) slp[b] = s.sleep[b * M + me * s.n + env];
    // The joints always couple the hull with wheel w, so they are solved on REGISTER copies of the two
    // cars (r0, r1), exactly like the per-car kernel; only the contacts pick their bodies by fixture index
    // at run time, and they work on the LDS copy.  Velocities (positions in the position phase) are
    // exchanged between the two copies around every contact pass: 30 independent LDS accesses each way
    // instead of every joint access being a dependent LDS round trip.
    CarRegs r = car[me];
    Body &mH = car[me].H;
    Body(&mW)[4] = car[me].W;
    auto vel_to_lds = [&]() {
        mH.vx = r.H.vx, mH.vy = r.H.vy, mH.w = r.H.w;
#pragma unroll
        for (int w = 0; w < 4; w++) mW[w].vx = r.W[w].vx, mW[w].vy = r.W[w].vy, mW[w].w = r.W[w].w;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    };
    auto vel_from_lds = [&]() {
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        r.H.vx = mH.vx, r.H.vy = mH.vy, r.H.w = mH.w;
#pragma unroll
        for (int w = 0; w < 4; w++) r.W[w].vx = mW[w].vx, r.W[w].vy = mW[w].vy, r.W[w].w = mW[w].w;
    };
    auto pos_to_lds = [&]() {
        mH.cx = r.H.cx, mH.cy = r.H.cy, mH.a = r.H.a;
#pragma unroll
        for (int w = 0; w < 4; w++) mW[w].cx = r.W[w].cx, mW[w].cy = r.W[w].cy, mW[w].a = r.W[w].a;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    };
    auto pos_from_lds = [&]() {
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        r.H.cx = mH.cx, r.H.cy = mH.cy, r.H.a = mH.a;
#pragma unroll
        for (int w = 0; w < 4; w++) r.W[w].cx = mW[w].cx, r.W[w].cy = mW[w].cy, r.W[w].a = mW[w].a;
    };
    // Envs whose boxes overlap but where nothing touches (most of this kernel's envs) are two independent islands, exactly as in the
    // per-car kernel.  A wavefront that only holds such envs runs island_solve(); in a MIXED wavefront they ride along the contact
    // path below instead of running island_solve() beside it (a divergent branch executes both sides one after the other: the
    // per-car solve used to add its 0.24 ms to the ~1 ms contact path of nearly every wavefront).  Riding along is the same
    // arithmetic: the LDS round trips copy values, the contact loops run zero times, and `iso` keeps the two places where one
    // island differs from two -- when the position iterations stop, and who goes to sleep -- per car.
    const bool iso = nc == 0;
    if (!__any(nc != 0)) {
        island_solve(r, K, h, dt_ratio, slp);
    } else {
        JointTmp jt;
        ContactVC *vc = sh_vc[pair];
        isl_integrate_vel(r, K, h);
        vel_to_lds();
        if (me == 0) {  // contact constraints and warm start: even lane, LDS copy
        // b2ContactSolver::InitializeVelocityConstraints, then WarmStart
        for (int k = 0; k < nc; k++) {
            Contact &c = ct[k];
            ContactVC &q = vc[k];
            const BRef A = body_of(car[0], K, c.pair >> 3), B = body_of(car[1], K, c.pair & 7);
            q.count = c.count;
            const float mA = A.im, iA = A.ii, mB = B.im, iB = B.ii;
            V2 pts[2];
            world_manifold(c, xf_of(A), xf_of(B), q.normal, pts);
            const V2 cA = mk(A.b->cx, A.b->cy), cB = mk(B.b->cx, B.b->cy), tangent = mk(q.normal.y, -q.normal.x);
            for (int j = 0; j < c.count; j++) {
                c.nimp[j] *= dt_ratio, c.timp[j] *= dt_ratio;
                q.rA[j] = pts[j] - cA, q.rB[j] = pts[j] - cB;
                const float rnA = cross(q.rA[j], q.normal), rnB = cross(q.rB[j], q.normal);
                const float kN = mA + mB + iA * rnA * rnA + iB * rnB * rnB;
                q.nmass[j] = kN > 0.0f ? 1.0f / kN : 0.0f;
                const float rtA = cross(q.rA[j], tangent), rtB = cross(q.rB[j], tangent);
                const float kT = mA + mB + iA * rtA * rtA + iB * rtB * rtB;
                q.tmass[j] = kT > 0.0f ? 1.0f / kT : 0.0f;
            }
            if (q.count == 2) {
                const float rn1A = cross(q.rA[0], q.normal), rn1B = cross(q.rB[0], q.normal);
                const float rn2A = cross(q.rA[1], q.normal), rn2B = cross(q.rB[1], q.normal);
                const float k11 = mA + mB + iA * rn1A * rn1A + iB * rn1B * rn1B, k22 = mA + mB + iA * rn2A * rn2A + iB * rn2B * rn2B;
                const float k12 = mA + mB + iA * rn1A * rn2A + iB * rn1B * rn2B;
                if (k11 * k11 < 1000.0f * (k11 * k22 - k12 * k12)) {
                    q.K[0][0] = k11, q.K[0][1] = k12, q.K[1][0] = k12, q.K[1][1] = k22;
                    float det = k11 * k22 - k12 * k12;
                    if (det != 0.0f) det = 1.0f / det;
                    q.invK[0][0] = det * k22, q.invK[1][0] = -det * k12, q.invK[0][1] = -det * k12, q.invK[1][1] = det * k11;
                } else {
                    q.count = 1;
                }
            }
        }
        for (int k = 0; k < nc; k++) {
            const Contact &c = ct[k];
            const ContactVC &q = vc[k];
            const BRef A = body_of(car[0], K, c.pair >> 3), B = body_of(car[1], K, c.pair & 7);
            const V2 tangent = mk(q.normal.y, -q.normal.x);
            for (int j = 0; j < q.count; j++) {
                const V2 P = c.nimp[j] * q.normal + c.timp[j] * tangent;
                A.b->w -= A.ii * cross(q.rA[j], P), A.b->vx -= A.im * P.x, A.b->vy -= A.im * P.y;
                B.b->w += B.ii * cross(q.rB[j], P), B.b->vx += B.im * P.x, B.b->vy += B.im * P.y;
            }
        }
        }
        vel_from_lds();  // (the contact warm start changed the velocities)
        isl_joints_init(r, jt, K, dt_ratio);
        const float friction = sqrtf(0.2f * 0.2f);
#pragma unroll 1
        for (int it = 0; it < 180; it++) {
            isl_joints_vel(r, jt, K, h);
            vel_to_lds();
            if (me == 0)
            for (int k = 0; k < nc; k++) {  // b2ContactSolver::SolveVelocityConstraints
                // The two bodies' velocities, the constraint data and the accumulated impulses come into
                // registers with back-to-back LDS reads, the whole contact (friction per point, then the
                // normal constraint) is solved there in Box2D's order, and everything goes back once.
                Contact &c = ct[k];
                const ContactVC &q = vc[k];
                const BRef A = body_of(car[0], K, c.pair >> 3), B = body_of(car[1], K, c.pair & 7);
                V2 vA = mk(A.b->vx, A.b->vy), vB = mk(B.b->vx, B.b->vy);
                float wA = A.b->w, wB = B.b->w;
                const float mA = A.im, iA = A.ii, mB = B.im, iB = B.ii;
                const V2 normal = q.normal, tangent = mk(normal.y, -normal.x);
                const int count = q.count;
                const V2 rA0 = q.rA[0], rB0 = q.rB[0], rA1 = q.rA[1], rB1 = q.rB[1];
                const float tm0 = q.tmass[0], tm1 = q.tmass[1], nm0 = q.nmass[0], nm1 = q.nmass[1];
                const float k00 = q.K[0][0], k01 = q.K[0][1], k10 = q.K[1][0], k11 = q.K[1][1];
                const float ik00 = q.invK[0][0], ik01 = q.invK[0][1], ik10 = q.invK[1][0], ik11 = q.invK[1][1];
                float nimp0 = c.nimp[0], nimp1 = c.nimp[1], timp0 = c.timp[0], timp1 = c.timp[1];
                auto relv = [&](V2 ra, V2 rb) { return ((vB + scross(wB, rb)) - vA) - scross(wA, ra); };
                auto apply = [&](V2 ra, V2 rb, V2 P) {
                    vA.x -= mA * P.x, vA.y -= mA * P.y, wA -= iA * cross(ra, P);
                    vB.x += mB * P.x, vB.y += mB * P.y, wB += iB * cross(rb, P);
                };
#pragma unroll
                for (int j = 0; j < 2; j++) {
                    if (j < count) {
                        const V2 ra = j ? rA1 : rA0, rb = j ? rB1 : rB0;
                        float &timp = j ? timp1 : timp0;
                        const float vt = dot(relv(ra, rb), tangent);
                        float lambda = (j ? tm1 : tm0) * (-vt);
                        const float maxF = friction * (j ? nimp1 : nimp0);
                        float ni = timp + lambda;
                        ni = ni < -maxF ? -maxF : ni > maxF ? maxF : ni;
                        lambda = ni - timp, timp = ni;
                        apply(ra, rb, lambda * tangent);
                    }
                }
                if (count == 1) {
                    const float vn = dot(relv(rA0, rB0), normal);
                    float lambda = -nm0 * (vn - 0.0f);
                    const float ni = fmaxf(nimp0 + lambda, 0.0f);
                    lambda = ni - nimp0, nimp0 = ni;
                    apply(rA0, rB0, lambda * normal);
                } else if (count == 2) {
                    const V2 a = mk(nimp0, nimp1);
                    float vn1 = dot(relv(rA0, rB0), normal), vn2 = dot(relv(rA1, rB1), normal);
                    V2 b = mk(vn1 - 0.0f, vn2 - 0.0f);
                    b = b - mk(k00 * a.x + k10 * a.y, k01 * a.x + k11 * a.y);
                    V2 x = mk(-(ik00 * b.x + ik10 * b.y), -(ik01 * b.x + ik11 * b.y));
                    bool solved = x.x >= 0.0f && x.y >= 0.0f;
                    if (!solved) {
                        x = mk(-nm0 * b.x, 0.0f);
                        vn2 = k01 * x.x + b.y;
                        solved = x.x >= 0.0f && vn2 >= 0.0f;
                    }
                    if (!solved) {
                        x = mk(0.0f, -nm1 * b.y);
                        vn1 = k10 * x.y + b.x;
                        solved = x.y >= 0.0f && vn1 >= 0.0f;
                    }
                    if (!solved) {
                        x = mk(0.0f, 0.0f);
                        solved = b.x >= 0.0f && b.y >= 0.0f;
                    }
                    if (solved) {
                        const V2 d = x - a;
                        const V2 P1 = d.x * normal, P2 = d.y * normal;
                        vA.x -= mA * (P1.x + P2.x), vA.y -= mA * (P1.y + P2.y);
                        wA -= iA * (cross(rA0, P1) + cross(rA1, P2));
                        vB.x += mB * (P1.x + P2.x), vB.y += mB * (P1.y + P2.y);
                        wB += iB * (cross(rB0, P1) + cross(rB1, P2));
                        nimp0 = x.x, nimp1 = x.y;
                    }
                }
                A.b->vx = vA.x, A.b->vy = vA.y, A.b->w = wA, B.b->vx = vB.x, B.b->vy = vB.y, B.b->w = wB;
                c.nimp[0] = nimp0, c.nimp[1] = nimp1, c.timp[0] = timp0, c.timp[1] = timp1;
            }
            vel_from_lds();
        }
        isl_integrate_pos(r, h);
        pos_to_lds();
        bool solved = false;
        bool iso_done = false;  // iso: this car's own position iterations have converged (b2Island::Solve breaks out there)
#pragma unroll 1
        for (int it = 0; it < 60; it++) {
            float minSep = 0.0f;
            if (me == 0)
            for (int k = 0; k < nc; k++) {  // b2ContactSolver::SolvePositionConstraints
                const Contact &c = ct[k];
                const BRef A = body_of(car[0], K, c.pair >> 3), B = body_of(car[1], K, c.pair & 7);
                for (int j = 0; j < c.count; j++) {
                    const XF xa = xf_of(A), xb = xf_of(B);
                    V2 normal, point;
                    float sep;
                    if (c.type == 0) {
                        normal = rotv(xa.s, xa.c, mk(c.ln[0], c.ln[1]));
                        const V2 plane = xmul(xa, mk(c.lp[0], c.lp[1])), clip = xmul(xb, mk(c.pt[j][0], c.pt[j][1]));
                        sep = dot(clip - plane, normal) - 0.01f - 0.01f, point = clip;
                    } else {
                        normal = rotv(xb.s, xb.c, mk(c.ln[0], c.ln[1]));
                        const V2 plane = xmul(xb, mk(c.lp[0], c.lp[1])), clip = xmul(xa, mk(c.pt[j][0], c.pt[j][1]));
                        sep = dot(clip - plane, normal) - 0.01f - 0.01f, point = clip;
                        normal = -1.0f * normal;
                    }
                    const V2 rA = point - mk(A.b->cx, A.b->cy), rB = point - mk(B.b->cx, B.b->cy);
                    if (sep < minSep) minSep = sep;
                    const float C = fminf(fmaxf(0.2f * (sep + LINEAR_SLOP), -0.2f), 0.0f);
                    const float rnA = cross(rA, normal), rnB = cross(rB, normal);
                    const float Kn = A.im + B.im + A.ii * rnA * rnA + B.ii * rnB * rnB;
                    const float impulse = Kn > 0.0f ? -C / Kn : 0.0f;
                    const V2 P = impulse * normal;
                    A.b->cx -= A.im * P.x, A.b->cy -= A.im * P.y, A.b->a -= A.ii * cross(rA, P);
                    B.b->cx += B.im * P.x, B.b->cy += B.im * P.y, B.b->a += B.ii * cross(rB, P);
                }
            }
            const bool cok = minSep >= -3.0f * LINEAR_SLOP;  // (meaningful in the even lane)
            pos_from_lds();
            bool jok = true;
            if (!iso_done) jok = isl_joints_pos(r, K);  // (a converged island of its own is not iterated again)
            pos_to_lds();
            if (iso && jok) iso_done = true;
            // contactsOkay && jointsOkay of the whole island: combine the pair (two islands: both have converged)
            const int mine = iso ? (iso_done ? 1 : 0) : ((me == 0 ? (cok ? 1 : 0) : 1) & (jok ? 1 : 0));
            const int other = __shfl_xor(mine, 1);
            if (mine & other) {
                solved = true;
                break;
            }
        }
        // one island: it sleeps only when all ten bodies have been still long enough; two islands: each on its own
        const float mm = isl_sleep_scan(r, slp, h);
        const float mo = __shfl_xor(mm, 1);
        if (iso) {
            if (mm >= TIME_TO_SLEEP && iso_done) isl_put_to_sleep(r, slp);
        } else if (fminf(mm, mo) >= TIME_TO_SLEEP && solved) {
            isl_put_to_sleep(r, slp);
        }
    }
    for (int b = 0; b < 5; b++) s.sleep[b * M + me * s.n + env] = slp[b];

    // ---- store bodies, joints and the manifolds with their impulses
    store_car(s, M, me * s.n + env, r);
    s.first_step[me * s.n + env] = 0;
    if (me != 0) return;
    s.n_contact[env] = nc;
    float *out = s.contact + env * (int64_t)(kMaxContacts * kContactWords);
    for (int k = 0; k < nc; k++) {
        float *o = out + k * kContactWords;
        const Contact &c = ct[k];
        o[0] = __int_as_float(c.pair), o[1] = __int_as_float(c.count), o[2] = __int_as_float(c.type);
        o[3] = c.ln[0], o[4] = c.ln[1], o[5] = c.lp[0], o[6] = c.lp[1];
        o[7] = c.pt[0][0], o[8] = c.pt[0][1], o[9] = c.pt[1][0], o[10] = c.pt[1][1];
        o[11] = __uint_as_float(c.id[0]), o[12] = __uint_as_float(c.id[1]);
        o[13] = c.nimp[0], o[14] = c.nimp[1], o[15] = c.timp[0], o[16] = c.timp[1];
    }
}

void launch_car_coupled(const CarSoA &s, const CarConsts &k, hipStream_t st) {
    if (s.players != 2 || !s.contacts_enabled) return;
    hipLaunchKernelGGL(car_coupled_kernel, dim3((unsigned)((s.n + 31) / 32)), dim3(64), 0, st, s, k);
}

}  // namespace crl
