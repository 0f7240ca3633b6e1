/* CPU twin of crl_selftest_sincosf (competitive_rl_amd/csrc/crl_selftest.hip): every float32 argument of crl_sincosf (stride 1) against the
 * double-double evaluation rounded once; also counts how often the host sinf is NOT the correctly rounded value (what the checker is able to see).
 *   gcc -O2 -fopenmp -ffp-contract=off -Iinclude tools/sincosf_sweep_cpu.c -o /tmp/sweep -lm && /tmp/sweep 1 */
#include <math.h>
#include <stdlib.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include "crl_f64.h"
#include "crl_rot.h"
static inline uint32_t f2u(float f){uint32_t u;memcpy(&u,&f,4);return u;}
static inline float u2f(uint32_t u){float f;memcpy(&f,&u,4);return f;}
static int check_cr32(float got, crl_dd v) {
    const float f = (float)v.h;
    const float dn = nextafterf(f, -INFINITY), up = nextafterf(f, INFINITY);
    const double mdn = 0.5 * ((double)f + (double)dn), mup = 0.5 * ((double)f + (double)up);
    crl_dd a = crl_two_sum(v.h, -mdn), b = crl_two_sum(v.h, -mup);
    const double da = a.h + (a.l + v.l), db = b.h + (b.l + v.l);
    const double tol = fabs(v.h) * 0x1p-80;
    if (fabs(da) <= tol || fabs(db) <= tol) return 2;
    float want = f;
    if (da < 0.0) want = dn; else if (db > 0.0) want = up;
    return f2u(got) == f2u(want) ? 0 : 1;
}
int main(int argc, char **argv) {
    int stride = argc > 1 ? atoi(argv[1]) : 16;
    unsigned long long tested = 0, bs = 0, bc = 0, und = 0, gl = 0;
#pragma omp parallel for reduction(+:tested,bs,bc,und,gl) schedule(static)
    for (long long i = 0; i < (1ll << 32); i += stride) {
        float x = u2f((uint32_t)i);
        if (!(fabsf(x) < 1647099.0f) || x == 0.0f) continue;
        float s, c; crl_sincosf(x, &s, &c);
        crl_dd ds, dc; crl_sincos_dd((double)x, &ds, &dc);
        int rs = check_cr32(s, ds), rc = check_cr32(c, dc);
        tested++; bs += rs == 1; bc += rc == 1; und += (rs == 2) + (rc == 2);
        if (rs == 1 || rc == 1) printf("MISS bits 0x%08x x=%a sin=%a cos=%a dd_s=(%a,%a) dd_c=(%a,%a)\n", (unsigned)i, x, s, c, ds.h, ds.l, dc.h, dc.l);
        gl += check_cr32(sinf(x), ds) == 1;
    }
    printf("tested %llu bad_s %llu bad_c %llu undecided %llu glibc_sinf_miss %llu\n", tested, bs, bc, und, gl);
}
