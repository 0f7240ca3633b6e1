// crl_selftest.hip -- device self-tests of the shared math (include/crl_rot.h), exported through the C ABI.
//
// crl_selftest_sincosf: every parity claim of the CarRacing float32 physics rests on crl_sincosf being THE correctly rounded
// float32 sine / cosine (a correctly rounded function has one right answer, whatever evaluates it).  A sample cannot show that; the
// argument space is small enough to sweep: this kernel evaluates crl_sincosf on every float32 bit pattern of a range and compares
// it with the double-double evaluation of include/crl_f64.h (relative error ~2^-95) rounded ONCE to float32 -- the rounding is
// decided against the midpoints to the neighbouring floats in double-double, and an argument whose exact value lies too close to
// a midpoint for the double-double error bound to decide is COUNTED as undecided instead of trusted.
#include <math.h>

#include "../../include/crl_f64.h"
#include "../../include/crl_rot.h"
#include "crl_internal.h"

namespace crl {

// 0 = `got` is the correctly rounded float32 of v = h + l, 1 = it is not, 2 = too close to a rounding boundary to decide
__device__ inline int check_cr32(float got, crl_dd v) {
    const float f = (float)v.h;  // candidate: round-to-nearest of the leading double
    const float dn = nextafterf(f, -INFINITY), up = nextafterf(f, INFINITY);
    // midpoints to the neighbours (exact in double: two adjacent float32 values have a 25-bit midpoint)
    const double mdn = 0.5 * ((double)f + (double)dn), mup = 0.5 * ((double)f + (double)up);
    // v - mid, exactly enough: two_sum of the leading doubles, then the low word
    crl_dd a = crl_two_sum(v.h, -mdn), b = crl_two_sum(v.h, -mup);
    const double da = a.h + (a.l + v.l), db = b.h + (b.l + v.l);
    const double tol = fabs(v.h) * 0x1p-80;  // 2^15 times the double-double evaluation's error bound
    if (fabs(da) <= tol || fabs(db) <= tol) return 2;
    float want = f;
    if (da < 0.0) want = dn;       // (only when the low word carries v across the midpoint the leading double rounded from)
    else if (db > 0.0) want = up;
    return __float_as_uint(got) == __float_as_uint(want) ? 0 : 1;
}

// out: [0] arguments tested, [1] sine mismatches, [2] cosine mismatches, [3] undecided, [4] first mismatching bit pattern + 1
__global__ __launch_bounds__(256) void sincosf_sweep_kernel(uint64_t first, uint64_t count, unsigned long long *out) {
    unsigned long long tested = 0, bad_s = 0, bad_c = 0, und = 0, first_bad = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t bits = (uint32_t)(first + i);
        const float x = __uint_as_float(bits);
        if (!(fabsf(x) < 1647099.0f)) continue;  // the documented domain |x| < 2^20 * pi/2 (also skips inf / nan)
        float s, c;
        crl_sincosf(x, &s, &c);
        tested++;
        if (x == 0.0f) {  // exact: sin(+-0) = +-0, cos = 1
            if (__float_as_uint(s) != bits) bad_s++;
            if (c != 1.0f) bad_c++;
            continue;
        }
        crl_dd ds, dc;
        if (!crl_sincos_dd((double)x, &ds, &dc)) {
            und++;
            continue;
        }
        const int rs = check_cr32(s, ds), rc = check_cr32(c, dc);
        bad_s += rs == 1, bad_c += rc == 1, und += (rs == 2) + (rc == 2);
        if ((rs == 1 || rc == 1) && !first_bad) first_bad = (unsigned long long)bits + 1ull;
    }
    // per-wavefront reduction, one atomic per counter and wavefront
    for (int d = 32; d; d >>= 1) {
        tested += __shfl_xor(tested, d), bad_s += __shfl_xor(bad_s, d), bad_c += __shfl_xor(bad_c, d), und += __shfl_xor(und, d);
    }
    if ((threadIdx.x & 63) == 0) {
        atomicAdd(out + 0, tested), atomicAdd(out + 1, bad_s), atomicAdd(out + 2, bad_c), atomicAdd(out + 3, und);
    }
    if (first_bad) atomicCAS(out + 4, 0ull, first_bad);
}

}  // namespace crl

extern "C" int crl_selftest_sincosf(int32_t device, uint64_t first_bits, uint64_t count, uint64_t *out5_host) {
    crl_fail_no_ctx();
    if (!out5_host || count == 0 || first_bits + count > (1ull << 32)) return crl_fail(CRL_EINVAL, "crl_selftest_sincosf: bad range");
    if (hipSetDevice(device) != hipSuccess) return crl_fail(CRL_EHIP, "crl_selftest_sincosf: hipSetDevice(%d)", device);
    unsigned long long *dev = nullptr;
    if (hipMalloc((void **)&dev, 5 * sizeof(unsigned long long)) != hipSuccess) return crl_fail(CRL_ENOMEM, "crl_selftest_sincosf: hipMalloc");
    hipMemset(dev, 0, 5 * sizeof(unsigned long long));
    const uint64_t want = (count + 255) / 256;
    hipLaunchKernelGGL(crl::sincosf_sweep_kernel, dim3((unsigned)(want < 65536 ? want : 65536)), dim3(256), 0, nullptr, first_bits, count, dev);
    unsigned long long h[5];
    const hipError_t e = hipMemcpy(h, dev, sizeof(h), hipMemcpyDeviceToHost);
    hipFree(dev);
    if (e != hipSuccess) return crl_fail(CRL_EHIP, "crl_selftest_sincosf: %s", hipGetErrorString(e));
    for (int i = 0; i < 5; i++) out5_host[i] = h[i];
    return CRL_OK;
}
