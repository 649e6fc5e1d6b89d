"""devmath (oracle/oracle_devmath.h, the CPU twin of terra_amd/csrc/dev_math.h) against this host's libm.

sinf/cosf must equal glibc on EVERY argument the diffuse sampler can produce, so
that diffuse-only scenes are bit-identical to the reference end to end.
powf/acosf/atan2f restate the published algorithms glibc 2.35 uses (table+polynomial powf,
fdlibm float acosf, atanf and atan2f) and must agree with this host's libm on every sampled argument
(NaN results: NaN-ness only, payload/sign unspecified)."""
import ctypes as C

import numpy as np


def _eval(orc, fn, mode, x, y=None):
    f = orc.fn("orc_math_eval", None, [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p])
    x = np.ascontiguousarray(x, np.float32)
    y = np.ascontiguousarray(y if y is not None else x, np.float32)
    out = np.zeros_like(x)
    f(fn, mode, len(x), x.ctypes.data, y.ctypes.data, out.ctypes.data)
    return out


def test_sincos_exhaustive_on_sampler_domain(H, orc_lib):
    f = orc_lib.fn("orc_devmath_sincos_domain_check", None, [C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)])
    bs, bc = C.c_uint64(1), C.c_uint64(1)
    f(C.byref(bs), C.byref(bc))
    assert (bs.value, bc.value) == (0, 0)


def test_sincos_general_arguments(H, orc_lib):
    r = H.rng(5)
    x = np.concatenate([r.uniform(-100, 100, 400000), r.uniform(-1e-3, 1e-3, 50000), [0.0, -0.0, 0.78539816, 1.5707964, 3.1415927, 6.2831855]]).astype(np.float32)
    for fn in (0, 1):
        a, b = _eval(orc_lib, fn, 0, x), _eval(orc_lib, fn, 1, x)
        assert H.same_bits(a, b), f"fn {fn}: {(H.bits(a) != H.bits(b)).sum()} mismatches"


def _ulp_diff(a, b):
    ia = H_bits(a).astype(np.int64); ib = H_bits(b).astype(np.int64)
    return np.abs(ia - ib)


def H_bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def test_powf_close_to_libm(H, orc_lib):
    r = H.rng(6)
    n = 400000
    # the renderer's uses: (1-e2)^(1/(n+1)), cos^n, colour^(1/gamma)
    x = np.concatenate([r.uniform(0, 1, n // 2), r.uniform(0, 16, n // 4), r.uniform(-1, 1, n // 4)]).astype(np.float32)
    y = np.concatenate([r.uniform(0, 1, n // 2), np.full(n // 4, 1 / 2.2), np.round(r.uniform(1, 60, n // 4))]).astype(np.float32)
    a, b = _eval(orc_lib, 2, 0, x, y), _eval(orc_lib, 2, 1, x, y)
    fin = np.isfinite(a) & np.isfinite(b)
    assert np.array_equal(np.isnan(a), np.isnan(b))
    d = _ulp_diff(a[fin], b[fin])
    rate = (d != 0).mean()
    print(f"powf: mismatch rate {rate:.2e}, max ulp {d.max()}")
    assert d.max() == 0 and rate == 0
    # special cases
    xs = np.array([0, 0, 1, 2, -2, -2, -2, np.inf, 0.5, 2, -1, np.nan, 3], np.float32)
    ys = np.array([0, 2, 5, 0, 2, 3, 0.5, 2, np.inf, -np.inf, 3, 1, np.nan], np.float32)
    a, b = _eval(orc_lib, 2, 0, xs, ys), _eval(orc_lib, 2, 1, xs, ys)
    assert np.array_equal(np.isnan(a), np.isnan(b)) and np.array_equal(a[~np.isnan(a)], b[~np.isnan(b)])


def test_acosf_close_to_libm(H, orc_lib):
    r = H.rng(7)
    x = np.concatenate([r.uniform(-1, 1, 400000), [1.0, -1.0, 0.0, 0.5, -0.5, 1.5, -1.5]]).astype(np.float32)
    a, b = _eval(orc_lib, 3, 0, x), _eval(orc_lib, 3, 1, x)
    assert np.array_equal(np.isnan(a), np.isnan(b))
    fin = ~np.isnan(a)
    d = _ulp_diff(a[fin], b[fin])
    rate = (d != 0).mean()
    print(f"acosf: mismatch rate {rate:.2e}, max ulp {d.max()}")
    assert d.max() == 0 and rate == 0


def test_atan2f_equals_libm(H, orc_lib):
    """the lat-long environment lookup (reference src/Terra.c:468-477) is the only atan2f user"""
    r = H.rng(8)
    n = 500000
    raw = r.randint(0, 2 ** 32, (2, n), dtype=np.uint64).astype(np.uint32).view(np.float32)
    y = np.concatenate([r.uniform(-1, 1, n), raw[0], [0, -0.0, 0, -0.0, 1, -1, np.inf, -np.inf, np.inf, 1, 1e-30]]).astype(np.float32)
    x = np.concatenate([r.uniform(-1, 1, n), raw[1], [1, 1, -1, -1, 0, 0, np.inf, -np.inf, 1, -np.inf, -3e30]]).astype(np.float32)
    a, b = _eval(orc_lib, 4, 0, y, x), _eval(orc_lib, 4, 1, y, x)
    assert np.array_equal(np.isnan(a), np.isnan(b))
    ok = ~np.isnan(a)
    assert np.array_equal(H_bits(a[ok]), H_bits(b[ok]))
