/*
 * crl_rot.h -- the one sine/cosine evaluation of the CarRacing float32 physics.
 *
 * Box2D builds a rotation as b2Rot(angle) = (sinf(angle), cosf(angle)) (b2Math.h), so the solver's
 * results depend on libm's last bit -- which differs between glibc (the reference's hosts, the CPU
 * oracle) and the GPU's device library.  Both the HIP kernels (competitive_rl_amd/csrc/car_*.hip)
 * and the CPU oracle (oracle/car_oracle.c) therefore evaluate THIS function: Cephes' single-precision
 * kernels (three-constant Cody-Waite reduction by pi/4, degree-7 / degree-8 polynomials, <= 2 ulp
 * for |x| < 8192), written as individual float multiplies and adds.  Both sides are compiled with
 * -ffp-contract=off, so every operation rounds once and the float32 state of a car is reproducible
 * bit for bit across CPU and GPU.  (Against any particular libm the difference is that libm's own
 * last-bit error; parity against real Box2D builds is unpinned either way)
 */
#ifndef CRL_ROT_H_
#define CRL_ROT_H_

#if defined(__HIPCC__) || defined(__CUDACC__)
#define CRL_ROT_FN __host__ __device__ static inline
#else
#define CRL_ROT_FN static inline
#endif

CRL_ROT_FN void crl_sincosf(float x, float *sn, float *cs) {
    const float ax = x < 0.0f ? -x : x;
    int j = (int)(ax * 1.27323954473516f); /* 4/pi */
    j = (j + 1) & ~1;                      /* nearest even octant: ax = j * pi/4 + r, |r| <= pi/4 */
    const float y = (float)j;
    const float r = ((ax - y * 0.78515625f) - y * 2.4187564849853515625e-4f) - y * 3.77489497744594108e-8f;
    const float z = r * r;
    const float ps = ((-1.9515295891e-4f * z + 8.3321608736e-3f) * z - 1.6666654611e-1f) * z * r + r;
    const float pc = ((2.443315711809948e-5f * z - 1.388731625493765e-3f) * z + 4.166664568298827e-2f) * z * z - 0.5f * z + 1.0f;
    const int q = (j >> 1) & 3;
    float s = (q & 1) ? pc : ps, c = (q & 1) ? ps : pc;
    if (q == 2 || q == 3) s = -s;
    if (q == 1 || q == 2) c = -c;
    *sn = x < 0.0f ? -s : s;
    *cs = c;
}

#endif /* CRL_ROT_H_ */
