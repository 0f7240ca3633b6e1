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
#include <stdlib.h>

#ifdef CRL_ABLATION
#include "car_obs_tile.h"
#endif
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

// ---- the same b2CollidePolygons on polygons whose vertices and edge normals are already in WORLD space (the narrow-phase
// kernel transforms each lane's two fixtures once into LDS: the nested separation search then reads 16 values instead of
// re-transforming a vertex per inner iteration).  Local-space quantities of the manifold (localNormal, localPoint, the
// incident points) are computed from the untransformed vertices exactly as in collide_polygons above.
struct WPoly {
    V2 w[8], n[8];  // world vertices, world edge normals
    int cnt;
};
// MAXV: the largest vertex count among the polygons the wavefront collides this time (4: boxes and the hull's quadrilaterals -- a
// quarter of the 8 x 8 table)
template <int MAXV>
__device__ inline float max_separation_w(int &edge, const WPoly &p1, const WPoly &p2) {
    // both polygons' data comes in with back-to-back LDS reads and is indexed statically (MAXV x MAXV, entries past a polygon's
    // count switched off): a rolled double loop is a chain of ~100-cycle LDS round trips
    V2 n1[MAXV], v1[MAXV], w2[MAXV];
#pragma unroll
    for (int i = 0; i < MAXV; i++) n1[i] = p1.n[i], v1[i] = p1.w[i], w2[i] = p2.w[i];
    const int c1 = p1.cnt, c2 = p2.cnt;
    float best = -3.4e38f;
    int bi = 0;
#pragma unroll
    for (int i = 0; i < MAXV; i++) {
        float si = 3.4e38f;
#pragma unroll
        for (int j = 0; j < MAXV; j++) {
            const float sij = dot(n1[i], w2[j] - v1[i]);
            if (j < c2 && sij < si) si = sij;
        }
        if (i < c1 && si > best) best = si, bi = i;
    }
    edge = bi;
    return best;
}
// nl: local edge normals of every fixture shape (table in LDS, [fixture][edge]), as shape_normal computes them
template <int MAXV>
__device__ void collide_polygons_w(Contact &c, const Shape &pa, const XF &xa, const WPoly &wa, const V2 *nla, const Shape &pb, const XF &xb,
                                   const WPoly &wb, const V2 *nlb) {
    c.count = 0;
    const float totalRadius = 0.02f;
    int edgeA, edgeB;
    const float sepA = max_separation_w<MAXV>(edgeA, wa, wb);
    if (sepA > totalRadius) return;
    const float sepB = max_separation_w<MAXV>(edgeB, wb, wa);
    if (sepB > totalRadius) return;
    const bool flip = sepB > 0.98f * sepA + 0.001f;
    const Shape &p1 = flip ? pb : pa, &p2 = flip ? pa : pb;
    const XF &x1 = flip ? xb : xa, &x2 = flip ? xa : xb;
    const WPoly &w1 = flip ? wb : wa, &w2 = flip ? wa : wb;
    const V2 *nl2 = flip ? nla : nlb;
    const int edge1 = flip ? edgeB : edgeA;
    c.type = flip ? 1 : 0;
    ClipV inc[2];
    {
        const V2 n1 = qmulT(x2, w1.n[edge1]);
        int idx = 0;
        float mind = 3.4e38f;
        for (int i = 0; i < p2.n; i++) {
            const float d = dot(n1, nl2[i]);
            if (d < mind) mind = d, idx = i;
        }
        const int i1 = idx, i2 = i1 + 1 < p2.n ? i1 + 1 : 0;
        inc[0].v = w2.w[i1], inc[0].id = mkid((uint32_t)edge1, (uint32_t)i1, 1u, 0u);
        inc[1].v = w2.w[i2], inc[1].id = mkid((uint32_t)edge1, (uint32_t)i2, 1u, 0u);
    }
    const int iv1 = edge1, iv2 = edge1 + 1 < p1.n ? edge1 + 1 : 0;
    V2 v11 = shape_vertex(p1, iv1), v12 = shape_vertex(p1, iv2);
    V2 lt = v12 - v11;
    lt = (1.0f / sqrtf(dot(lt, lt))) * lt;
    const V2 ln = mk(lt.y, -lt.x), planePoint = 0.5f * (v11 + v12);
    const V2 tangent = rotv(x1.s, x1.c, lt), normal = mk(tangent.y, -tangent.x);
    v11 = w1.w[iv1], v12 = w1.w[iv2];
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

// ---- (1) narrow phase: ONE WAVEFRONT per coupled env, one lane per fixture pair (fa = lane / 8 of car 0, fb = lane % 8 of
// car 1; the 16 wheel-wheel pairs do not collide).  b2CollidePolygons per lane, then the touching pairs are compacted in
// (fa, fb) order -- the order the contacts are solved in -- with a ballot; at most kMaxContacts are kept.  Accumulated
// impulses are carried over from last step's manifolds by contact id (warm starting).  The env goes to one of two lists:
// `touch` (one island with contacts: car_touch_kernel) or `near` (the boxes overlap but nothing touches: two islands of
// their own, car_near_kernel = the per-car solve).
// phase (round 5): the broadphase that runs ahead files the coupled envs in two lists -- front: poses final (count [0]), back: envs that touch in
// the current step (count [7], car_broad_kernel).  0: both lists; 1: the front list only (beside the touching solve); 2: the back list only
// (behind it).  The output lists are appended to in any order, so two launches fill them like one.
__global__ __launch_bounds__(64) void car_narrow_kernel(CarSoA s, CarConsts Kv, int urgent, const float *__restrict__ fresh_body,
                                                        const uint8_t *__restrict__ cls, int phase) {
    // (the fixture tables are indexed per lane at run time: a by-value kernel argument would first be copied to every lane's
    // scratch, and reading them from device memory makes every vertex a dependent ~200-cycle load: stage them in LDS)
    __shared__ CarConsts Ks;
    const int lane = threadIdx.x;
#ifdef CRL_ABLATION
    const unsigned long long sn0 = __builtin_readcyclecounter();
    unsigned long long acc1 = 0, acc2 = 0, acc3 = 0, nslot = 0;
#endif
    if (urgent) __builtin_amdgcn_s_setprio(3);  // in front of the touching solve it heads the step's critical path; run ahead, at the end of the
                                                // previous step, it must not take issue slots from that step's last frames and the next car_step_kernel
    const int n_front = s.coupled_count[0], n_back = s.coupled_count[7];
    const int slot0 = phase == 2 ? n_front : 0, slot1 = phase == 1 ? n_front : n_front + n_back;  // slots [0, n_front) = the front list, then the back list
    {
        if (blockIdx.x == 0 && lane == 0 && s.coupled_to_host && phase != 1) *s.coupled_to_host = n_front + n_back;
        if ((int)blockIdx.x >= slot1 - slot0) return;  // (before the tables are staged: most of the grid has nothing to do)
    }
    {
        const uint32_t *src = reinterpret_cast<const uint32_t *>(s.consts_dev);
        uint32_t *dst = reinterpret_cast<uint32_t *>(&Ks);
        for (int i = lane; i < (int)(sizeof(CarConsts) / 4); i += 64) dst[i] = src[i];
    }
    __syncthreads();
    const CarConsts &K = Ks;
    (void)Kv;
    // local edge normals of the five distinct shapes (hull polygons 0-3, wheel box): one lane per edge, once per workgroup
    __shared__ V2 nl[8][8];
    __shared__ float circ[5][3];  // bounding circle of each shape (local centre, radius + margin): rejects most pairs cheaply
    if (lane < 40) {
        const int f = lane >> 3, e = lane & 7;  // f = 4: every wheel
        const Shape sh = shape_of(K, f);
        if (e < sh.n) nl[f][e] = shape_normal(sh, e);
        if (e == 0) {
            V2 ctr = mk(0.f, 0.f);
            for (int i = 0; i < sh.n; i++) ctr = ctr + shape_vertex(sh, i);
            ctr = (1.0f / sh.n) * ctr;
            float r2 = 0.f;
            for (int i = 0; i < sh.n; i++) r2 = fmaxf(r2, dot(shape_vertex(sh, i) - ctr, shape_vertex(sh, i) - ctr));
            circ[f][0] = ctr.x, circ[f][1] = ctr.y, circ[f][2] = sqrtf(r2) + 0.03f;
        }
    }
    __syncthreads();
    // the 16 fixtures of the env in world space (car 0's at [f], car 1's at [8 + f]): built once by 16 lanes, read by every pair.
    // (One copy per PAIR -- 17 KB per workgroup -- allowed eight workgroups per CU: 2 300 coupled envs took two rounds of the kernel.)
    __shared__ WPoly wp[16];
    const int64_t M = 2 * s.n;
#ifdef CRL_ABLATION
    const unsigned long long sn1 = __builtin_readcyclecounter();
#endif
    for (int slot = slot0 + (int)blockIdx.x; slot < slot1; slot += gridDim.x) {
#ifdef CRL_ABLATION
        const unsigned long long sa = __builtin_readcyclecounter();
        unsigned long long sb = sa;
#endif
        const int64_t env = slot < n_front ? s.coupled_list[slot] : s.coupled_list[s.n - 1 - (slot - n_front)];
        // (collide-ahead) an env that finished while coupled: its new episode's bodies, still staged; no manifolds to warm-start from
        const bool fresh = cls && cls[env] == 3;
        const float *body = fresh ? fresh_body : s.body;
        const int fa = lane >> 3, fb = lane & 7;
        const bool active = !(fa >= 4 && fb >= 4);
        Contact c;
        c.pair = fa * 8 + fb, c.type = 0, c.count = 0;
        for (int i = 0; i < 2; i++) c.nimp[i] = c.timp[i] = 0.f, c.id[i] = 0u, c.pt[i][0] = c.pt[i][1] = 0.f;
        c.ln[0] = c.ln[1] = c.lp[0] = c.lp[1] = 0.f;
        if (active) {
            XF xf[2];
            float ccx[2], ccy[2], crad[2];
#pragma unroll
            for (int k = 0; k < 2; k++) {
                const int f = k ? fb : fa;
                const int64_t ci = k * s.n + env;
                const int o = f < 4 ? 0 : 6 + 6 * (f - 4);
                const float bx = body[(o + 0) * M + ci], by = body[(o + 1) * M + ci], ba = body[(o + 2) * M + ci];
                crl_sincosf(ba, &xf[k].s, &xf[k].c);
                const V2 lc = f < 4 ? mk(K.hull_lc[0], K.hull_lc[1]) : mk(0.f, 0.f);
                xf[k].p = mk(bx, by) - rotv(xf[k].s, xf[k].c, lc);
                // bounding circle of the fixture (world centre, radius): circles that do not overlap cannot be within the
                // 0.02 contact margin
                const float *cq = circ[f < 4 ? f : 4];
                const V2 wc = xmul(xf[k], mk(cq[0], cq[1]));
                ccx[k] = wc.x, ccy[k] = wc.y, crad[k] = cq[2];
            }
            const float dx = ccx[0] - ccx[1], dy = ccy[0] - ccy[1], rr = crad[0] + crad[1];
#ifdef CRL_ABLATION
            sb = __builtin_readcyclecounter();
#endif
            const bool cand = !(dx * dx + dy * dy > rr * rr);
            // (wave-uniform: does any candidate pair of this env involve the hull's octagon?)
            const bool big = __any(cand && (shape_of(K, fa).n > 4 || shape_of(K, fb).n > 4));
            if (__any(cand)) {  // (wave-uniform) every fixture into world space once: vertices, and normals as rotv(q, local normal)
#pragma unroll
                for (int k = 0; k < 2; k++) {
                    if (k == 0 ? fb == 0 : fa == 0) {  // this lane builds car k's fixture (pairs (fa, 0) and (0, fb) are active ones)
                        const int f = k ? fb : fa;
                        const Shape sh = shape_of(K, f);
                        WPoly &w = wp[8 * k + f];
                        w.cnt = sh.n;
                        for (int i = 0; i < sh.n; i++) w.w[i] = xmul(xf[k], shape_vertex(sh, i)), w.n[i] = rotv(xf[k].s, xf[k].c, nl[f < 4 ? f : 4][i]);
                    }
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
            if (cand) {
                if (big) collide_polygons_w<8>(c, shape_of(K, fa), xf[0], wp[fa], nl[fa < 4 ? fa : 4], shape_of(K, fb), xf[1], wp[8 + fb], nl[fb < 4 ? fb : 4]);
                else collide_polygons_w<4>(c, shape_of(K, fa), xf[0], wp[fa], nl[fa < 4 ? fa : 4], shape_of(K, fb), xf[1], wp[8 + fb], nl[fb < 4 ? fb : 4]);
            }
        }
#ifdef CRL_ABLATION
        const unsigned long long sc = __builtin_readcyclecounter();
#endif
        const bool hit = active && c.count > 0;
        const unsigned long long m = __ballot(hit);
        const int rank = (int)__popcll(m & ((1ull << lane) - 1ull));
        const int nc = min((int)__popcll(m), kMaxContacts);
        if (hit && rank < kMaxContacts) {
            const float *old = s.contact + env * (int64_t)(kMaxContacts * kContactWords);
            const int n_old = fresh ? 0 : s.n_contact[env];
            for (int k = 0; k < n_old; k++) {
                const float *o = old + k * kContactWords;
                if (__float_as_int(o[0]) != c.pair) continue;
                const int ocount = __float_as_int(o[1]);
                for (int i = 0; i < c.count; i++)
                    for (int j = 0; j < ocount; j++)
                        if (__float_as_uint(o[11 + j]) == c.id[i]) c.nimp[i] = o[13 + j], c.timp[i] = o[15 + j];
            }
            float *o = s.contact_new + (env * kMaxContacts + rank) * (int64_t)kContactWords;
            o[0] = __int_as_float(c.pair), o[1] = __int_as_float(c.count), o[2] = __int_as_float(c.type);
            o[3] = c.ln[0], o[4] = c.ln[1], o[5] = c.lp[0], o[6] = c.lp[1];
            o[7] = c.pt[0][0], o[8] = c.pt[0][1], o[9] = c.pt[1][0], o[10] = c.pt[1][1];
            o[11] = __uint_as_float(c.id[0]), o[12] = __uint_as_float(c.id[1]);
            o[13] = c.nimp[0], o[14] = c.nimp[1], o[15] = c.timp[0], o[16] = c.timp[1];
        }
        if (lane == 0) {
            if ((int)__popcll(m) > kMaxContacts) atomicAdd(s.cap_hits + 1, 1);
            s.nc_new[env] = nc;
            // [1] near-only; touching by manifold count: [2] one, [3] two, [4] three or more
            const int cls = nc == 0 ? 0 : nc >= 3 ? 3 : nc;
            int32_t *lst = nc ? s.touch_list + (int64_t)(cls - 1) * s.n : s.near_list;
            lst[atomicAdd(s.coupled_count + 1 + cls, 1)] = (int32_t)env;
            if (nc) s.touch_all[atomicAdd(s.coupled_count + 5, 1)] = (int32_t)env;
            if (nc >= 2) s.touch_multi[atomicAdd(s.coupled_count + 6, 1)] = (int32_t)env;
        }
#ifdef CRL_ABLATION
        const unsigned long long sd = __builtin_readcyclecounter();
        acc1 += sb - sa, acc2 += sc - sb, acc3 += sd - sc, nslot++;
#endif
    }
#ifdef CRL_ABLATION
    if (lane == 0 && s.stamps && nslot) {
        unsigned long long *q = s.stamps + 32;
        atomicAdd(q + 0, sn1 - sn0), atomicAdd(q + 1, acc1), atomicAdd(q + 2, acc2), atomicAdd(q + 3, acc3), atomicAdd(q + 7, nslot);
    }
#endif
}

// ---- (2) envs whose boxes overlap but where nothing touches: two islands of their own, exactly the per-car kernel,
// two lanes per env over the compacted list
template <bool FM>
__global__ __launch_bounds__(64) void car_near_kernel(CarSoA s, CarConsts K) {
    const int64_t M = 2 * s.n;
    const int count = s.coupled_count[1];
    for (int slot = blockIdx.x * 32 + (threadIdx.x >> 1); slot < count; slot += gridDim.x * 32) {
        const int64_t env = s.near_list[slot];
        const int64_t ci = (int64_t)(threadIdx.x & 1) * s.n + env;
        CarRegs cr;
        load_car(s, M, ci, cr);
#pragma unroll
        for (int w = 0; w < 4; w++) cr.fx[w] = s.wforce[(2 * w + 0) * M + ci], cr.fy[w] = s.wforce[(2 * w + 1) * M + ci];
        const float h = (float)(1.0 / CAR_FPS);
        const float dt_ratio = s.first_step[ci] ? 0.0f : (1.0f / h) * h;
        float slp[5];
#pragma unroll
        for (int b = 0; b < 5; b++) slp[b] = s.sleep[b * M + ci];
        island_solve<FM>(cr, K, h, dt_ratio, slp);
#pragma unroll
        for (int b = 0; b < 5; b++) s.sleep[b * M + ci] = slp[b];
        store_car(s, M, ci, cr);
        s.first_step[ci] = 0;
        if ((threadIdx.x & 1) == 0) s.n_contact[env] = 0;
    }
}

// ---- (3) touching envs: one Box2D island of ten bodies, eight joints and nc contacts.
// TWO LANES PER ENV: lane 2p holds car 0 of pair p ("A side"), lane 2p+1 car 1 ("B side"), state in registers.  The joints
// of a car only couple its own hull and wheels: each lane solves its car's joints exactly like the per-car kernel, both
// cars at once.  A contact couples ONE body of each car, picked by fixture index at run time: each lane selects its own
// body's velocity (five candidates, v_cndmask), the two lanes swap those three values through DPP, BOTH lanes then solve
// the contact -- same operands, same instructions, one pass of the wavefront -- and each lane writes its own body back.
// Everything a contact needs in the 180 velocity iterations (constraint rows, accumulated impulses, manifold) lives in
// registers for the first NK contacts; the narrow phase files the envs by manifold count (1, 2, >= 3: 94 % / 6 % / 0.2 % of the
// touching envs), so a wavefront runs the instance for ITS count and the loop over contacts is unrolled.  Manifolds past
// the third (not seen in 10^4 touching env-steps, possible) go through the same code from their LDS rows.
struct TouchC {  // one contact's velocity-constraint rows in LDS (where InitializeVelocityConstraints leaves them)
    float nimp0, nimp1, timp0, timp1;
    float nx, ny, rA0x, rA0y;
    float rB0x, rB0y, rA1x, rA1y;
    float rB1x, rB1y, tm0, tm1;
    float nm0, nm1, k00, k01;
    float k10, k11, ik00, ik01;
    float ik10, ik11;
    int count, pair;
};
static_assert(sizeof(TouchC) == 112, "TouchC rows");

struct KC {  // one contact in registers
    float nimp0, nimp1, timp0, timp1;
    float nx, ny, rA0x, rA0y, rB0x, rB0y, rA1x, rA1y, rB1x, rB1y, tm0, tm1, nm0, nm1, k00, k01, k10, k11, ik00, ik01, ik10, ik11;
    int count;  // velocity-constraint points (the block solver may have dropped the second)
    int bi;     // this lane's body of the contact: 0 = hull, 1..4 = wheels 0..3
    float mA, iA, mB, iB;
    bool ok;  // the env has this contact (k < nc)
    // (the manifold itself -- what the position pass needs -- stays in its LDS row: ten registers per contact that nothing reads in
    // the 180 velocity iterations were what the allocator spilled to scratch and reloaded one by one afterwards)
};

__device__ __forceinline__ float lane_swap(float x) {  // the value of the other lane of the pair (DPP quad_perm [1,0,3,2])
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0xB1, 0xF, 0xF, false));
}

// body `bi` of a car in registers: 0 = hull, 1..4 = wheels 0..3
// (one v_cndmask per candidate: written as a sequence of simple selects -- a nested ?: chain comes out as branches)
__device__ __forceinline__ float sel5(int bi, float h, float w0, float w1, float w2, float w3) {
    float x = h;
    x = bi == 1 ? w0 : x;
    x = bi == 2 ? w1 : x;
    x = bi == 3 ? w2 : x;
    x = bi == 4 ? w3 : x;
    return x;
}
#define CRL_SEL(field, bi) sel5(bi, r.H.field, r.W[0].field, r.W[1].field, r.W[2].field, r.W[3].field)
#define CRL_PUT(field, bi, ok, val)                                                   \
    {                                                                                 \
        const int bj_ = (ok) ? (bi) : -1;                                             \
        r.H.field = bj_ == 0 ? (val) : r.H.field;                                     \
        r.W[0].field = bj_ == 1 ? (val) : r.W[0].field;                               \
        r.W[1].field = bj_ == 2 ? (val) : r.W[1].field;                               \
        r.W[2].field = bj_ == 3 ? (val) : r.W[2].field;                               \
        r.W[3].field = bj_ == 4 ? (val) : r.W[3].field;                               \
    }

__device__ __forceinline__ KC kc_load(const TouchC &t, const Contact &c, int me, bool ok, const CarConsts &K) {
    KC q;
    q.nimp0 = t.nimp0, q.nimp1 = t.nimp1, q.timp0 = t.timp0, q.timp1 = t.timp1;
    q.nx = t.nx, q.ny = t.ny, q.rA0x = t.rA0x, q.rA0y = t.rA0y, q.rB0x = t.rB0x, q.rB0y = t.rB0y, q.rA1x = t.rA1x, q.rA1y = t.rA1y;
    q.rB1x = t.rB1x, q.rB1y = t.rB1y, q.tm0 = t.tm0, q.tm1 = t.tm1, q.nm0 = t.nm0, q.nm1 = t.nm1;
    q.k00 = t.k00, q.k01 = t.k01, q.k10 = t.k10, q.k11 = t.k11, q.ik00 = t.ik00, q.ik01 = t.ik01, q.ik10 = t.ik10, q.ik11 = t.ik11;
    q.count = t.count;
    const int fA = t.pair >> 3, fB = t.pair & 7, f = me ? fB : fA;
    q.bi = f < 4 ? 0 : f - 3;
    q.mA = fA < 4 ? K.hull_inv_mass : K.wheel_inv_mass, q.iA = fA < 4 ? K.hull_inv_I : K.wheel_inv_I;
    q.mB = fB < 4 ? K.hull_inv_mass : K.wheel_inv_mass, q.iB = fB < 4 ? K.hull_inv_I : K.wheel_inv_I;
    q.ok = ok;
    (void)c;
    return q;
}

// b2ContactSolver::SolveVelocityConstraints for one contact: friction per point, then the normal constraint (one point, or
// the two-point block solver), in Box2D's order.  any1 / any2: some lane of the wavefront has a one- / two-point constraint.
template <bool FM>
__device__ __forceinline__ void contact_vel(CarRegs &r, KC &q, const int me, const float friction, const bool any1, const bool any2) {
    const int bi = q.bi;
    const float ovx = CRL_SEL(vx, bi), ovy = CRL_SEL(vy, bi), ow = CRL_SEL(w, bi);
    const float pvx = lane_swap(ovx), pvy = lane_swap(ovy), pw = lane_swap(ow);
    V2 vA = me ? mk(pvx, pvy) : mk(ovx, ovy), vB = me ? mk(ovx, ovy) : mk(pvx, pvy);
    float wA = me ? pw : ow, wB = me ? ow : pw;
    const float mA = q.mA, iA = q.iA, mB = q.mB, iB = q.iB;
    const V2 normal = mk(q.nx, q.ny), tangent = mk(normal.y, -normal.x);
    const V2 rA0 = mk(q.rA0x, q.rA0y), rB0 = mk(q.rB0x, q.rB0y), rA1 = mk(q.rA1x, q.rA1y), rB1 = mk(q.rB1x, q.rB1y);
    const int count = q.count;
    // ((vB + wB x rb) - vA) - wA x ra,  w x r = (-w r.y, w r.x)   (mad / nmad: car_solver.h)
    auto relv = [&](V2 ra, V2 rb) {
        return mk(nmad<FM>(-wA, ra.y, nmad<FM>(wB, rb.y, vB.x) - vA.x), nmad<FM>(wA, ra.x, mad<FM>(wB, rb.x, vB.y) - vA.y));
    };
    auto apply = [&](V2 ra, V2 rb, V2 P) {
        vA.x = nmad<FM>(mA, P.x, vA.x), vA.y = nmad<FM>(mA, P.y, vA.y), wA = nmad<FM>(iA, fcross<FM>(ra, P), wA);
        vB.x = mad<FM>(mB, P.x, vB.x), vB.y = mad<FM>(mB, P.y, vB.y), wB = mad<FM>(iB, fcross<FM>(rb, P), wB);
    };
    {  // friction, point 0 (every constraint has it)
        const float vt = fdot<FM>(relv(rA0, rB0), tangent);
        const float maxF = friction * q.nimp0;
        float ni = mad<FM>(q.tm0, -vt, q.timp0);
        ni = ni < -maxF ? -maxF : ni > maxF ? maxF : ni;  // (maxF may be +0: the compare chain keeps the limit's sign, a median would not)
        const float lambda = ni - q.timp0;
        q.timp0 = ni;
        apply(rA0, rB0, lambda * tangent);
    }
    if (any2) {
        if (count == 2) {  // friction, point 1
            const float vt = fdot<FM>(relv(rA1, rB1), tangent);
            const float maxF = friction * q.nimp1;
            float ni = mad<FM>(q.tm1, -vt, q.timp1);
            ni = ni < -maxF ? -maxF : ni > maxF ? maxF : ni;
            const float lambda = ni - q.timp1;
            q.timp1 = ni;
            apply(rA1, rB1, lambda * tangent);
        }
    }
    if (any1) {
        if (count == 1) {
            const float vn = fdot<FM>(relv(rA0, rB0), normal);
            const float ni = fmaxf(mad<FM>(-q.nm0, vn - 0.0f, q.nimp0), 0.0f);
            const float lambda = ni - q.nimp0;
            q.nimp0 = ni;
            apply(rA0, rB0, lambda * normal);
        }
    }
    if (any2) {
        if (count == 2) {
            const V2 a = mk(q.nimp0, q.nimp1);
            float vn1 = fdot<FM>(relv(rA0, rB0), normal), vn2 = fdot<FM>(relv(rA1, rB1), normal);
            V2 b = mk(vn1 - 0.0f, vn2 - 0.0f);
            b = b - mk(mad<FM>(q.k00, a.x, q.k10 * a.y), mad<FM>(q.k01, a.x, q.k11 * a.y));
            V2 x = mk(-mad<FM>(q.ik00, b.x, q.ik10 * b.y), -mad<FM>(q.ik01, b.x, q.ik11 * b.y));
            bool solved = x.x >= 0.0f && x.y >= 0.0f;
            if (!solved) {
                x = mk(-q.nm0 * b.x, 0.0f);
                vn2 = mad<FM>(q.k01, x.x, b.y);
                solved = x.x >= 0.0f && vn2 >= 0.0f;
            }
            if (!solved) {
                x = mk(0.0f, -q.nm1 * b.y);
                vn1 = mad<FM>(q.k10, x.y, b.x);
                solved = x.y >= 0.0f && vn1 >= 0.0f;
            }
            if (!solved) {
                x = mk(0.0f, 0.0f);
                solved = b.x >= 0.0f && b.y >= 0.0f;
            }
            if (solved) {
                const V2 d = x - a;
                const V2 P1 = d.x * normal, P2 = d.y * normal;
                vA.x = nmad<FM>(mA, P1.x + P2.x, vA.x), vA.y = nmad<FM>(mA, P1.y + P2.y, vA.y);
                wA = nmad<FM>(iA, fcross<FM>(rA0, P1) + fcross<FM>(rA1, P2), wA);
                vB.x = mad<FM>(mB, P1.x + P2.x, vB.x), vB.y = mad<FM>(mB, P1.y + P2.y, vB.y);
                wB = mad<FM>(iB, fcross<FM>(rB0, P1) + fcross<FM>(rB1, P2), wB);
                q.nimp0 = x.x, q.nimp1 = x.y;
            }
        }
    }
    const float nvx = me ? vB.x : vA.x, nvy = me ? vB.y : vA.y, nw = me ? wB : wA;
    CRL_PUT(vx, bi, q.ok, nvx) CRL_PUT(vy, bi, q.ok, nvy) CRL_PUT(w, bi, q.ok, nw)
}

// b2ContactSolver::SolvePositionConstraints for one contact (both manifold points)
template <bool FM>
__device__ __forceinline__ void contact_pos(CarRegs &r, const KC &q, const Contact &c, const int me, const V2 hlc, float &minSep) {
    const int bi = q.bi;
    const int ccount = c.count, ctype = c.type;
    const float lnx = c.ln[0], lny = c.ln[1], lpx = c.lp[0], lpy = c.lp[1];
    const float mA = q.mA, iA = q.iA, mB = q.mB, iB = q.iB;
    const V2 mylc = bi == 0 ? hlc : mk(0.f, 0.f);
#pragma unroll
    for (int j = 0; j < 2; j++) {
        const bool ok = q.ok && j < ccount;
        if (!__any(ok)) continue;  // (no island of this wavefront has this point: with one env per wavefront, exactly this island)
        // this lane's body (centre, angle) -> its transform; the partner's through DPP
        const float ocx = CRL_SEL(cx, bi), ocy = CRL_SEL(cy, bi), oa = CRL_SEL(a, bi);
        XF ox;
        crl_sincosf(oa, &ox.s, &ox.c);
        ox.p = mk(ocx, ocy) - frot<FM>(ox.s, ox.c, mylc);
        XF px;
        px.s = lane_swap(ox.s), px.c = lane_swap(ox.c), px.p.x = lane_swap(ox.p.x), px.p.y = lane_swap(ox.p.y);
        const float pcx = lane_swap(ocx), pcy = lane_swap(ocy), pa = lane_swap(oa);
        const XF xa = me ? px : ox, xb = me ? ox : px;
        V2 cA = me ? mk(pcx, pcy) : mk(ocx, ocy), cB = me ? mk(ocx, ocy) : mk(pcx, pcy);
        float aA = me ? pa : oa, aB = me ? oa : pa;
        const V2 lpt = mk(c.pt[j][0], c.pt[j][1]);
        V2 normal, point;
        float sep;
        if (ctype == 0) {
            normal = frot<FM>(xa.s, xa.c, mk(lnx, lny));
            const V2 plane = frot<FM>(xa.s, xa.c, mk(lpx, lpy)) + xa.p, clip = frot<FM>(xb.s, xb.c, lpt) + xb.p;
            sep = fdot<FM>(clip - plane, normal) - 0.01f - 0.01f, point = clip;
        } else {
            normal = frot<FM>(xb.s, xb.c, mk(lnx, lny));
            const V2 plane = frot<FM>(xb.s, xb.c, mk(lpx, lpy)) + xb.p, clip = frot<FM>(xa.s, xa.c, lpt) + xa.p;
            sep = fdot<FM>(clip - plane, normal) - 0.01f - 0.01f, point = clip;
            normal = -1.0f * normal;
        }
        const V2 rA = point - cA, rB = point - cB;
        if (ok && sep < minSep) minSep = sep;
        const float C = fminf(fmaxf(0.2f * (sep + LINEAR_SLOP), -0.2f), 0.0f);
        const float rnA = fcross<FM>(rA, normal), rnB = fcross<FM>(rB, normal);
        const float Kn = mad<FM>(iB * rnB, rnB, mad<FM>(iA * rnA, rnA, mA + mB));
        const float impulse = Kn > 0.0f ? -C / Kn : 0.0f;
        const V2 P = impulse * normal;
        cA.x = nmad<FM>(mA, P.x, cA.x), cA.y = nmad<FM>(mA, P.y, cA.y), aA = nmad<FM>(iA, fcross<FM>(rA, P), aA);
        cB.x = mad<FM>(mB, P.x, cB.x), cB.y = mad<FM>(mB, P.y, cB.y), aB = mad<FM>(iB, fcross<FM>(rB, P), aB);
        const float ncx = me ? cB.x : cA.x, ncy = me ? cB.y : cA.y, na = me ? aB : aA;
        CRL_PUT(cx, bi, ok, ncx) CRL_PUT(cy, bi, ok, ncy) CRL_PUT(a, bi, ok, na)
    }
}

// NK = contacts kept in registers (the list this wavefront serves holds envs with nc == NK, or nc >= 3 for NK == 3)
// EPW = envs per wavefront: 32 (one per lane pair) down to 1 -- every lane pair works on the SAME env and pair 0 stores it.  One env
// per wavefront wastes lanes, not time: the chip has a thousand SIMDs for a few hundred touching envs, every wave-uniform
// shortcut (which constraint kinds occur, how many manifolds, position iterations until THIS island has converged) becomes
// exact, and the kernel -- which ends with its slowest wavefront -- no longer pays for the union of its envs' worst cases.
// TAIL: the env may have more manifolds than the NK kept in registers; those go through their LDS rows every iteration
// (44 LDS reads per row and iteration -- still far cheaper than what the register allocator does when three rows, five bodies and
// four joints do not fit 512 registers: NK = 3 spilled 167 of them, 66 scratch accesses inside the velocity loop)
template <int NK, int EPW, bool TAIL, bool FM>
__device__ __forceinline__ void touch_solve(const CarSoA &s, const CarConsts &K, const int32_t *list, const int list_count, const int slot_base,
                                            CarRegs (*sh_car)[2], Contact (*sh_ct)[kMaxContacts], TouchC (*sh_tc)[kMaxContacts]) {
    const int me = threadIdx.x & 1;
    const int pair = (threadIdx.x >> 1) % EPW;  // (lane pairs past EPW repeat the first ones' envs and store nothing)
    const int slot = slot_base + pair;
    const bool live = slot < list_count && (threadIdx.x >> 1) < EPW;
#ifdef CRL_ABLATION
    const unsigned long long st0 = __builtin_readcyclecounter();
#endif
    // (lanes past the end of the list ride along on the last env's data and store nothing: the DPP swaps and the uniform
    // loop bounds below want every lane of the wavefront in step)
    const int64_t env = list[slot < list_count ? slot : list_count - 1];
    const int64_t M = 2 * s.n;
    CarRegs(&car)[2] = sh_car[pair];
    {
        const int64_t ci = me * s.n + env;
        load_car(s, M, ci, car[me]);
        for (int w = 0; w < 4; w++) car[me].fx[w] = s.wforce[(2 * w + 0) * M + ci], car[me].fy[w] = s.wforce[(2 * w + 1) * M + ci];
    }
    const int first_step = s.first_step[env];
    const float h = (float)(1.0 / CAR_FPS);
    const float dt_ratio = first_step ? 0.0f : (1.0f / h) * h;
    Contact *ct = sh_ct[pair];
    TouchC *tc = sh_tc[pair];
    const int nc = s.nc_new[env];
    int nc_wave = nc;  // contacts past NK: the wavefront's loop runs to the largest count in it
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) nc_wave = max(nc_wave, __shfl_xor(nc_wave, d));
    constexpr int kCW = (int)(sizeof(Contact) / 4);  // a Contact is its first 17 persisted words, in order (kContactWords = the stride in memory)
    static_assert(sizeof(Contact) == 17 * 4 && kCW <= kContactWords, "Contact layout");
    if (EPW == 1) {  // one env per wavefront: every lane mirrors it -- the words of its manifolds go one per lane (a lone lane's
                     // 17 loads per manifold, one after the other, were a quarter of the setup of an eight-manifold island)
        const uint32_t *in = reinterpret_cast<const uint32_t *>(s.contact_new + env * (int64_t)(kMaxContacts * kContactWords));
        uint32_t *dst = reinterpret_cast<uint32_t *>(ct);
        for (int i = threadIdx.x; i < nc * kCW; i += 64) dst[i] = in[(i / kCW) * kContactWords + i % kCW];
    } else if (me == 0) {  // this step's manifolds, from the narrow phase
        const float *in = s.contact_new + env * (int64_t)(kMaxContacts * kContactWords);
        for (int k = 0; k < nc; k++) {
            const float *o = in + k * kContactWords;
            Contact &c = ct[k];
            c.pair = __float_as_int(o[0]), c.count = __float_as_int(o[1]), c.type = __float_as_int(o[2]);
            c.ln[0] = o[3], c.ln[1] = o[4], c.lp[0] = o[5], c.lp[1] = o[6];
            c.pt[0][0] = o[7], c.pt[0][1] = o[8], c.pt[1][0] = o[9], c.pt[1][1] = o[10];
            c.id[0] = __float_as_uint(o[11]), c.id[1] = __float_as_uint(o[12]);
            c.nimp[0] = o[13], c.nimp[1] = o[14], c.timp[0] = o[15], c.timp[1] = o[16];
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");

    CarRegs r = car[me];
    Body &mH = car[me].H;
    Body(&mW)[4] = car[me].W;
    JointTmp jt;
    isl_integrate_vel<FM>(r, K, h);
    {  // b2ContactSolver::InitializeVelocityConstraints + WarmStart, once per step: even lane, on an LDS copy of both cars
        mH.vx = r.H.vx, mH.vy = r.H.vy, mH.w = r.H.w;
#pragma unroll
        for (int w = 0; w < 4; w++) mW[w].vx = r.W[w].vx, mW[w].vy = r.W[w].vy, mW[w].w = r.W[w].w;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        // EPW == 1: manifold k's constraint rows by lane k (they depend on the poses and the manifold only); else lane 0 of the pair
        if (EPW == 1 ? (int)threadIdx.x < nc : me == 0) {
            for (int k = EPW == 1 ? (int)threadIdx.x : 0; k < (EPW == 1 ? (int)threadIdx.x + 1 : nc); k++) {
                Contact &c = ct[k];
                ContactVC q;
                const BRef A = body_of(car[0], K, c.pair >> 3), B = body_of(car[1], K, c.pair & 7);
                q.count = c.count;
                const float mA = A.im, iA = A.ii, mB = B.im, iB = B.ii;
                V2 pts[2];
                pts[0] = pts[1] = mk(0.f, 0.f);
                world_manifold(c, xf_of(A), xf_of(B), q.normal, pts);
                const V2 cA = mk(A.b->cx, A.b->cy), cB = mk(B.b->cx, B.b->cy), tangent = mk(q.normal.y, -q.normal.x);
                q.rA[0] = q.rA[1] = q.rB[0] = q.rB[1] = mk(0.f, 0.f);
                q.nmass[0] = q.nmass[1] = q.tmass[0] = q.tmass[1] = 0.f;
                q.K[0][0] = q.K[0][1] = q.K[1][0] = q.K[1][1] = q.invK[0][0] = q.invK[0][1] = q.invK[1][0] = q.invK[1][1] = 0.f;
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
                TouchC &t = tc[k];
                t.nimp0 = c.nimp[0], t.nimp1 = c.nimp[1], t.timp0 = c.timp[0], t.timp1 = c.timp[1];
                t.nx = q.normal.x, t.ny = q.normal.y, t.rA0x = q.rA[0].x, t.rA0y = q.rA[0].y;
                t.rB0x = q.rB[0].x, t.rB0y = q.rB[0].y, t.rA1x = q.rA[1].x, t.rA1y = q.rA[1].y;
                t.rB1x = q.rB[1].x, t.rB1y = q.rB[1].y, t.tm0 = q.tmass[0], t.tm1 = q.tmass[1];
                t.nm0 = q.nmass[0], t.nm1 = q.nmass[1], t.k00 = q.K[0][0], t.k01 = q.K[0][1];
                t.k10 = q.K[1][0], t.k11 = q.K[1][1], t.ik00 = q.invK[0][0], t.ik01 = q.invK[0][1];
                t.ik10 = q.invK[1][0], t.ik11 = q.invK[1][1], t.count = q.count, t.pair = c.pair;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        if (me == 0) {  // warm start: in contact order (the bodies' velocities chain through it)
            for (int k = 0; k < nc; k++) {
                const TouchC &t = tc[k];
                const BRef A = body_of(car[0], K, t.pair >> 3), B = body_of(car[1], K, t.pair & 7);
                const V2 normal = mk(t.nx, t.ny), tangent = mk(normal.y, -normal.x);
                for (int j = 0; j < t.count; j++) {
                    const V2 rA = j ? mk(t.rA1x, t.rA1y) : mk(t.rA0x, t.rA0y), rB = j ? mk(t.rB1x, t.rB1y) : mk(t.rB0x, t.rB0y);
                    const V2 P = (j ? t.nimp1 : t.nimp0) * normal + (j ? t.timp1 : t.timp0) * tangent;
                    A.b->w -= A.ii * cross(rA, P), A.b->vx -= A.im * P.x, A.b->vy -= A.im * P.y;
                    B.b->w += B.ii * cross(rB, P), B.b->vx += B.im * P.x, B.b->vy += B.im * P.y;
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");  // (the contact warm start changed the velocities)
        r.H.vx = mH.vx, r.H.vy = mH.vy, r.H.w = mH.w;
#pragma unroll
        for (int w = 0; w < 4; w++) r.W[w].vx = mW[w].vx, r.W[w].vy = mW[w].vy, r.W[w].w = mW[w].w;
    }
    isl_joints_init(r, jt, K, dt_ratio);
    const float friction = sqrtf(0.2f * 0.2f);
    // the first NK contacts into registers (rows past the env's count: contact 0's, switched off)
    KC kc[NK];
    bool any1 = false, any2 = false;
#pragma unroll
    for (int k = 0; k < NK; k++) {
        const bool ok = k < nc;
        kc[k] = kc_load(tc[ok ? k : 0], ct[ok ? k : 0], me, ok, K);
        any1 = any1 || __any(kc[k].count == 1), any2 = any2 || __any(kc[k].count == 2);
    }
    bool tail1 = false, tail2 = false;  // contacts past NK (TAIL only)
    for (int k = NK; k < nc_wave; k++) tail1 = tail1 || __any(k < nc && tc[k < nc ? k : 0].count == 1), tail2 = tail2 || __any(k < nc && tc[k < nc ? k : 0].count == 2);
#ifdef CRL_ABLATION
    const unsigned long long st1 = __builtin_readcyclecounter();
#endif
    const int jmode = isl_joint_mode(r);
#pragma unroll 1
    for (int it = 0; it < 180; it++) {
        isl_joints_vel_mode<FM>(jmode, r, jt, K, h);
#pragma unroll
        for (int k = 0; k < NK; k++) contact_vel<FM>(r, kc[k], me, friction, any1, any2);
        if (TAIL) {
#pragma unroll 1
            for (int k = NK; k < nc_wave; k++) {
                const bool ok = k < nc;
                KC q = kc_load(tc[ok ? k : 0], ct[ok ? k : 0], me, ok, K);
                contact_vel<FM>(r, q, me, friction, tail1, tail2);
                if (ok && me == 0) tc[k].nimp0 = q.nimp0, tc[k].nimp1 = q.nimp1, tc[k].timp0 = q.timp0, tc[k].timp1 = q.timp1;
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            }
        }
    }
    if (me == 0) {
#pragma unroll
        for (int k = 0; k < NK; k++)
            if (k < nc) ct[k].nimp[0] = kc[k].nimp0, ct[k].nimp[1] = kc[k].nimp1, ct[k].timp[0] = kc[k].timp0, ct[k].timp[1] = kc[k].timp1;
        for (int k = NK; k < nc; k++) ct[k].nimp[0] = tc[k].nimp0, ct[k].nimp[1] = tc[k].nimp1, ct[k].timp[0] = tc[k].timp0, ct[k].timp[1] = tc[k].timp1;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
#ifdef CRL_ABLATION
    const unsigned long long st2 = __builtin_readcyclecounter();
    int pos_iters = 0;
#endif
    // (the poses did not change since the car was loaded: from the LDS copy again, so that they need no registers across the
    // velocity iterations)
    r.H.cx = mH.cx, r.H.cy = mH.cy, r.H.a = mH.a;
#pragma unroll
    for (int w = 0; w < 4; w++) r.W[w].cx = mW[w].cx, r.W[w].cy = mW[w].cy, r.W[w].a = mW[w].a;
    // Behind the position integration velocities, joint impulses and limit states are final: stored then, while they are in registers (the position
    // iterations need every register for their sines and cosines; what they do not touch went to scratch and came back one
    // value at a time for the stores at the end: 30-45 k cycles behind the last iteration of the step's slowest islands).
    // The island's sleep scan only reads the velocities: here as well.
    isl_integrate_pos<FM>(r, h);  // (clamps velocities that would move a body too far in one step: final only now)
    if (live) store_car_vel(s, M, me * s.n + env, r);
    float slp[5];  // b2Body::m_sleepTime of this lane's car
    for (int b = 0; b < 5; b++) slp[b] = s.sleep[b * M + me * s.n + env];
    const float mm = isl_sleep_scan(r, slp, h);
    bool solved = false;
    const V2 hlc = mk(K.hull_lc[0], K.hull_lc[1]);
#pragma unroll 1
    for (int it = 0; it < 60; it++) {
        float minSep = 0.0f;
#ifdef CRL_ABLATION
        pos_iters = it + 1;
#endif
#pragma unroll
        for (int k = 0; k < NK; k++) contact_pos<FM>(r, kc[k], ct[k < nc ? k : 0], me, hlc, minSep);
        if (TAIL) {
#pragma unroll 1
            for (int k = NK; k < nc_wave; k++) {
                const bool ok = k < nc;
                const KC q = kc_load(tc[ok ? k : 0], ct[ok ? k : 0], me, ok, K);
                contact_pos<FM>(r, q, ct[ok ? k : 0], me, hlc, minSep);
            }
        }
        const bool cok = minSep >= -3.0f * LINEAR_SLOP;
        const bool jok = isl_joints_pos<FM>(r, K);
        // contactsOkay && jointsOkay of the whole island: combine the pair
        const int mine = (cok ? 1 : 0) & (jok ? 1 : 0);
        const int other = __shfl_xor(mine, 1);
        if (mine & other) {  // (both lanes of the pair leave together; the wavefront goes on for the other islands)
            solved = true;
            break;
        }
    }
#ifdef CRL_ABLATION
    const unsigned long long st3 = __builtin_readcyclecounter();
    int pi_wave = pos_iters;
    for (int d = 1; d < 64; d <<= 1) pi_wave = max(pi_wave, __shfl_xor(pi_wave, d));
#endif
    // one island: it sleeps only when all ten bodies have been still long enough
    const float mo = __shfl_xor(mm, 1);
    if (fminf(mm, mo) >= TIME_TO_SLEEP && solved) {
        isl_put_to_sleep(r, slp);
        if (live) store_car_vel(s, M, me * s.n + env, r);  // (the velocities again: zero)
    }
    if (EPW == 1) {  // the manifolds with their impulses, one word per lane (every lane mirrors the env)
        const uint32_t *src = reinterpret_cast<const uint32_t *>(ct);
        uint32_t *out = reinterpret_cast<uint32_t *>(s.contact + env * (int64_t)(kMaxContacts * kContactWords));
        for (int i = threadIdx.x; i < nc * kCW; i += 64) out[(i / kCW) * kContactWords + i % kCW] = src[i];
    }
    if (!live) return;
#ifdef CRL_ABLATION
    const unsigned long long st3b = __builtin_readcyclecounter();
#endif
    for (int b = 0; b < 5; b++) s.sleep[b * M + me * s.n + env] = slp[b];

    // ---- store bodies, joints and the manifolds with their impulses
    store_car_pos(s, M, me * s.n + env, r);
    s.first_step[me * s.n + env] = 0;
#ifdef CRL_ABLATION
    const unsigned long long st3c = __builtin_readcyclecounter();
#endif
    if (me != 0) return;
    s.n_contact[env] = nc;
    float *out = s.contact + env * (int64_t)(kMaxContacts * kContactWords);
    for (int k = 0; k < (EPW == 1 ? 0 : nc); k++) {
        float *o = out + k * kContactWords;
        const Contact &c = ct[k];
        o[0] = __int_as_float(c.pair), o[1] = __int_as_float(c.count), o[2] = __int_as_float(c.type);
        o[3] = c.ln[0], o[4] = c.ln[1], o[5] = c.lp[0], o[6] = c.lp[1];
        o[7] = c.pt[0][0], o[8] = c.pt[0][1], o[9] = c.pt[1][0], o[10] = c.pt[1][1];
        o[11] = __uint_as_float(c.id[0]), o[12] = __uint_as_float(c.id[1]);
        o[13] = c.nimp[0], o[14] = c.nimp[1], o[15] = c.timp[0], o[16] = c.timp[1];
    }
#ifdef CRL_ABLATION
    if (threadIdx.x == 0 && s.stamps) {
        const unsigned long long st4 = __builtin_readcyclecounter();
        unsigned long long *q = s.stamps + 8 * (NK - 1);
        atomicAdd(q + 0, st1 - st0), atomicAdd(q + 1, st2 - st1), atomicAdd(q + 2, st3 - st2), atomicAdd(q + 3, st4 - st3);
        atomicAdd(q + 4, (unsigned long long)pi_wave), atomicMax(q + 5, st4 - st0), atomicAdd(q + 6, st3c - st3b), atomicAdd(q + 7, 1ull);
        unsigned long long *m = s.stamps + 40 + 4 * (NK - 1);  // per class: longest velocity phase, longest position phase, wavefronts that ran all 60 position iterations
        atomicMax(m + 0, st2 - st1), atomicMax(m + 1, st3 - st2), atomicAdd(m + 2, pi_wave >= 60 ? 1ull : 0ull), atomicMax(m + 3, st1 - st0);
    }
#endif
}
#undef CRL_SEL
#undef CRL_PUT

// one launch, blockIdx.y = manifold-count class (0: nc == 1, 1: nc == 2, 2: nc >= 3); the workgroups loop over the class's list.
// EPW1 = envs per wavefront of class 0 (94 % of the touching envs); two manifolds or more (the slowest islands): one env per wavefront.
// NK2 / NK3: manifolds kept in registers for classes 1 and 2 (the rest through LDS).
// (Round 4 tried an epilogue in which the wavefront that solved an island also draws that env's two frames, behind an epoch word
// published by the wheel sensors' stream: one tile takes a lone wavefront ~75 us, so two or eight tiles in a row behind a solve are
// slower than the list launch that draws them side by side -- and a kernel that spins on another kernel's output deadlocks as soon
// as both are only partly dispatched.  docs/LAB_NOTES_r04.md.)
#ifdef CRL_ABLATION
// Epilogue of the touching solve (round 5, profiling build only: measured, not kept): the solving wavefront also prepares the VIEWS of its
// envs' frames -- camera (double-double atan2 / sincos, ~15 us of latency) and the car polygons' scanline spans (~20 us) -- from the poses
// it has just stored, into the arrays the bulk path's camera / polygon kernels fill (s.view, s.view_rec, s.view_cnt).
// touch_view == 1: every class, and the frame launch behind the solve only gathers: + 2-3 % per step (the 35 us land on the wavefronts that
// end the kernel).  touch_view == 2: the one-manifold class only (eight envs per wavefront, done long before the slow islands; behind the
// solve car_view_list_kernel computes the ~110 multi-manifold envs' views from touch_multi -- THIS step's narrow-phase lists on both sides,
// no manifold counts read while the next narrow phase rewrites them): +- 0 in four alternating pairs of shipped-flavour builds.
__device__ __forceinline__ void touch_view_epilogue(const CarSoA &s, const CarConsts &K, const int32_t *list, int count, int base, int epw) {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "agent");  // the solved poses were stored by this wavefront: completed, and not served from a stale L1 line
    const int lane = threadIdx.x;
    const int ntile = 2 * min(epw, count - base);
    for (int t0 = 0; t0 < ntile; t0 += 4) {
        const int tl = t0 + (lane >> 4);
        if (tl >= ntile) continue;
        const int64_t env = list[base + (tl >> 1)];
        const int viewer = tl & 1, q = lane & 15;
        const int64_t t = env * 2 + viewer;
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
#endif

template <int EPW1, int NK2, int NK3, bool FM>
__global__ __launch_bounds__(64) void car_touch_kernel(CarSoA s, CarConsts K, int cls0) {
    const int cls = cls0 + blockIdx.y;
    const int count = s.coupled_count[2 + cls];
    const int epw = cls == 0 ? EPW1 : 1;
    if ((int)blockIdx.x * epw >= count) return;
    __builtin_amdgcn_s_setprio(3);
    __shared__ __attribute__((aligned(16))) CarRegs sh_car[EPW1][2];
    __shared__ __attribute__((aligned(16))) Contact sh_ct[EPW1][kMaxContacts];
    __shared__ __attribute__((aligned(16))) TouchC sh_tc[EPW1][kMaxContacts];
    const int32_t *list = s.touch_list + (int64_t)cls * s.n;
    for (int base = blockIdx.x * epw; base < count; base += gridDim.x * epw) {
        if (cls == 0) touch_solve<1, EPW1, false, FM>(s, K, list, count, base, sh_car, sh_ct, sh_tc);
        else if (cls == 1) touch_solve<NK2, 1, (NK2 < 2), FM>(s, K, list, count, base, sh_car, sh_ct, sh_tc);
        else touch_solve<NK3, 1, true, FM>(s, K, list, count, base, sh_car, sh_ct, sh_tc);
#ifdef CRL_ABLATION
        if (s.touch_view == 1 || (s.touch_view == 2 && cls == 0)) touch_view_epilogue(s, K, list, count, base, epw);
#endif
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    }
}

// world.Step of the coupled envs.  `near_st` (may equal `st`): where the near-only envs are solved, beside the touching ones.
void launch_car_narrow(const CarSoA &s, const CarConsts &k, hipStream_t st, bool urgent, const float *fresh_body, const uint8_t *cls, int phase) {
    const unsigned cap = (unsigned)(s.n < 4096 ? s.n : 4096);
    hipLaunchKernelGGL(car_narrow_kernel, dim3(cap), dim3(64), 0, st, s, k, urgent ? 1 : 0, fresh_body, cls, phase);
}

// skip_narrow: the narrow phase of this step already ran (ahead, at the end of the previous step); the caller has ordered `st`,
// `near_st` and `one_st` behind it
void launch_car_coupled(const CarSoA &s, const CarConsts &k, hipStream_t st, hipStream_t near_st, hipEvent_t ev_narrow, bool skip_narrow) {
    if (s.players != 2 || !s.contacts_enabled) return;
    if (!near_st) near_st = st;
    if (!skip_narrow) {
        launch_car_narrow(s, k, st, true);
        if (near_st != st) {
            hipEventRecord(ev_narrow, st);
            hipStreamWaitEvent(near_st, ev_narrow, 0);
        }
    }
    const unsigned gn = (unsigned)((s.n + 31) / 32 < 512 ? (s.n + 31) / 32 : 512);
    const unsigned g = (unsigned)((s.n + 31) / 32 < 256 ? (s.n + 31) / 32 : 256);
#ifndef CRL_EPW1
#define CRL_EPW1 8  // envs per wavefront of the one-manifold class (build-time experiment switch: python -m competitive_rl_amd.build --variant e4 -DCRL_EPW1=4)
#endif
    if (s.fma) {  // CRL_FLAG_CAR_FMA: the iterations in fused multiply-adds (car_solver.h)
        hipLaunchKernelGGL(car_near_kernel<true>, dim3(gn), dim3(64), 0, near_st, s, k);
        hipLaunchKernelGGL((car_touch_kernel<CRL_EPW1, 2, 3, true>), dim3(g, 3), dim3(64), 0, st, s, k, 0);
    } else {
        hipLaunchKernelGGL(car_near_kernel<false>, dim3(gn), dim3(64), 0, near_st, s, k);
        hipLaunchKernelGGL((car_touch_kernel<CRL_EPW1, 2, 3, false>), dim3(g, 3), dim3(64), 0, st, s, k, 0);
    }
}

}  // namespace crl
