"""include/crl_f64.h (the double-precision sin / cos / atan2 shared by the HIP kernels and liboracle.so) against
mpmath, and how often the host libm -- what the reference itself calls -- differs from it."""
import numpy as np
import pytest

from oracle import car_oracle as co

mp = pytest.importorskip("mpmath")


def _args(n, seed):
    rs = np.random.RandomState(seed)
    x = np.concatenate([rs.uniform(-1, 1, n // 4), rs.uniform(-10, 10, n // 4), rs.uniform(-400, 400, n // 4),
                        rs.uniform(-1e-3, 1e-3, n // 8), rs.uniform(-1e5, 1e5, n // 8)])
    y, z = rs.uniform(-300, 300, len(x)), rs.uniform(-300, 300, len(x))
    return x, y, z


def test_crl_f64_is_correctly_rounded_on_a_sample():
    mp.mp.prec = 200
    x, y, z = _args(12000, 1)
    s, c, a = co.f64(0, x), co.f64(1, x), co.f64(2, y, z)
    for i in range(len(x)):
        assert float(mp.sin(mp.mpf(float(x[i])))) == s[i], (x[i], s[i])
        assert float(mp.cos(mp.mpf(float(x[i])))) == c[i], (x[i], c[i])
        assert float(mp.atan2(mp.mpf(float(y[i])), mp.mpf(float(z[i])))) == a[i], (y[i], z[i], a[i])


def test_crl_sincosf_is_correctly_rounded_on_a_sample():
    """include/crl_rot.h: b2Rot's float32 sine / cosine, against mpmath rounded once to float32; and how often glibc's
    sinf / cosf return the neighbouring float instead"""
    mp.mp.prec = 120
    rs = np.random.RandomState(5)
    x = np.concatenate([rs.uniform(-4, 4, 8000), rs.uniform(-80, 80, 8000), rs.uniform(-1e-3, 1e-3, 2000),
                        rs.uniform(-2000, 2000, 2000)]).astype(np.float32).astype(np.float64)
    s, c = co.f64(6, x), co.f64(7, x)
    sl, cl = co.f64(6, x, libm=True), co.f64(7, x, libm=True)

    def f32(v):  # mpmath -> nearest float32 (round half even): through a 24-bit mpf
        with mp.workprec(24):
            return float(+v)

    for i in range(len(x)):
        xs = mp.mpf(float(x[i]))
        assert f32(mp.sin(xs)) == s[i], (x[i], s[i])
        assert f32(mp.cos(xs)) == c[i], (x[i], c[i])
    miss = float(((sl != s) | (cl != c)).mean())
    print("fraction of arguments where glibc sinf / cosf is not the correctly rounded float:", miss)
    assert miss < 0.05
    assert (np.abs(sl - s) <= np.spacing(np.abs(s).astype(np.float32)).astype(np.float64)).all()


def test_fast_variants_stay_within_three_ulp():
    """crl_*_fast (the track walk): plain-double evaluations, compared with the correctly rounded ones"""
    x, y, z = _args(400000, 3)
    worst = {}
    for fn, name in ((0, "sin"), (1, "cos"), (2, "atan2")):
        a = co.f64(fn + 3, x if fn < 2 else y, None if fn < 2 else z)
        b = co.f64(fn, x if fn < 2 else y, None if fn < 2 else z)
        ulp = np.spacing(np.abs(b))
        worst[name] = (float((np.abs(a - b) / ulp).max()), float((a != b).mean()))
    print("fast vs correctly rounded: (max ulp, fraction differing)", worst)
    assert all(v[0] <= 3.0 for v in worst.values()), worst


def test_special_arguments():
    assert co.f64(0, [0.0, -0.0]).tolist() == [0.0, -0.0] and np.signbit(co.f64(0, [-0.0]))[0]
    assert co.f64(1, [0.0])[0] == 1.0
    pi = float(np.pi)
    got = co.f64(2, [0.0, -0.0, 0.0, -0.0, 1.0, -1.0, 2.5, 0.0], [1.0, 1.0, -1.0, -1.0, 0.0, 0.0, 2.5, 0.0])
    assert got.tolist() == [0.0, -0.0, pi, -pi, pi / 2, -pi / 2, pi / 4, 0.0]
    assert np.isnan(co.f64(0, [np.inf, np.nan, 1e300])).all()


def test_distance_to_the_host_libm():
    """glibc's sin / cos / atan2 are documented to < 1 ulp, not correctly rounded: about one call in a thousand
    returns the neighbouring double.  That is the whole distance between liboracle.so and liboracle_libm.so."""
    x, y, z = _args(400000, 2)
    out = {}
    for fn, name in ((0, "sin"), (1, "cos"), (2, "atan2")):
        a = co.f64(fn, x if fn < 2 else y, None if fn < 2 else z)
        b = co.f64(fn, x if fn < 2 else y, None if fn < 2 else z, libm=True)
        diff = a != b
        ulp = np.spacing(np.abs(b))
        assert (np.abs(a - b) <= ulp).all(), name  # never more than one unit in the last place
        out[name] = float(diff.mean())
    print("fraction of calls where the host libm differs from the correctly rounded value:", out)
    assert all(v < 5e-3 for v in out.values())


@pytest.mark.gpu
def test_crl_sincosf_is_correctly_rounded_for_every_float32_argument():
    """the sample above is 20 000 arguments; the space is small enough to sweep.  crl_selftest_sincosf (include/crl.h)
    evaluates crl_sincosf on EVERY float32 bit pattern of its domain (|x| < 2^20 * pi/2: 2.5e9 arguments) on the GPU and
    compares with the double-double evaluation of include/crl_f64.h (~2^-95) rounded once to float32: 0 mismatches, and no
    argument so close to a rounding boundary that the double-double bound could not decide."""
    import ctypes as C

    import torch

    from competitive_rl_amd import _native as N

    if not torch.cuda.is_available():
        pytest.fail("no GPU visible: -m gpu tests must run on the MI355X box")
    L = N.load()
    tot = np.zeros(5, np.uint64)
    for first in range(0, 1 << 32, 1 << 30):  # four launches of 2^30 bit patterns
        out = np.zeros(5, np.uint64)
        N.check(L.crl_selftest_sincosf(0, first, 1 << 30, out.ctypes.data_as(C.c_void_p)))
        tot[:4] += out[:4]
        tot[4] = tot[4] or out[4]
    tested, bad_s, bad_c, undecided, first_bad = (int(v) for v in tot)
    print("crl_sincosf sweep: tested", tested, "sin mismatches", bad_s, "cos mismatches", bad_c, "undecided", undecided)
    # 2 x (bit patterns below 0x49C90FDB = 1647099.0f's neighbourhood) incl. both zeros
    assert tested > 2_400_000_000, tested
    assert bad_s == 0 and bad_c == 0, (bad_s, bad_c, hex(first_bad - 1))
    assert undecided == 0, undecided
