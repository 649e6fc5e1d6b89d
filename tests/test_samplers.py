"""SURVEY.md 8f N4, unit level: the reference's stratified / Halton samplers and 1D / 2D distributions
(src/Terra.c:703-755, 760-846). Chain: compiled reference -> tests/golden/samplers.npz -> oracle (CPU) -> device (GPU),
bit for bit. Nothing on the render path uses them (the reference constructs a sampler per pixel and never draws from it)."""
import numpy as np
import pytest


def G(H):
    return np.load(H.GOLDEN / "samplers.npz")


def check_against_golden(H, U):
    g = G(H)
    for strata, samples in ((1, 1), (2, 3), (4, 16), (7, 2)):
        got = U.stratified(g["seeds"], strata, samples, strata * strata * samples)
        assert H.same_bits(got, g[f"strat_{strata}_{samples}"]), (strata, samples)
    assert H.same_bits(U.halton(0, 4096), g["halton_0"]) and H.same_bits(U.halton(2 ** 30, 512), g["halton_far"])
    tables, e, t2, e12 = H.sampler_cases()
    assert H.same_bits(e, g["e"]) and H.same_bits(e12, g["e12"])
    for name, f in tables.items():
        out = U.distribution_1d(f, e)
        for k, v in out.items():
            want = g[f"d1_{name}_{k}"]
            assert (np.array_equal(v, want) if v.dtype == np.uint32 else H.same_bits(v, want)), (name, k)
    for name, f in t2.items():
        out = U.distribution_2d(f, e12)
        for k, v in out.items():
            assert H.same_bits(v, g[f"d2_{name}_{k}"]), (name, k)


def test_oracle_equals_the_reference_goldens(H, orc_lib):
    check_against_golden(H, H.Unit("orc"))


def test_sampler_properties(H, orc_lib):
    U = H.Unit("orc")
    s = U.stratified([3], 4, 16, 256)[0]
    cell = np.floor(s * 4).astype(int)                       # sample k lies in stratum k // 16 = (x, y) with x fastest
    k = np.arange(256) // 16
    assert np.array_equal(cell[:, 0], k % 4) and np.array_equal(cell[:, 1], k // 4) and (s < 1).all() and (s >= 0).all()
    h = U.halton(0, 9)
    assert np.allclose(h[:, 1], [0, .5, .25, .75, .125, .625, .375, .875, .0625]) and np.allclose(h[:4, 0], [0, 1 / 3, 2 / 3, 1 / 9], atol=1e-7)
    f = np.array([1, 3, 0, 4], np.float32)
    out = U.distribution_1d(f, np.array([0.0, 0.124, 0.125, 0.49, 0.5, 0.99], np.float32))
    assert np.array_equal(out["idx"], [0, 0, 1, 1, 3, 3]) and np.allclose(out["pdf"], f[out["idx"]] / 8) and out["integral"][0] == 8
    assert (np.diff(out["x"]) >= 0).all() and np.allclose(out["x"][0], 0) and out["x"][-1] < 1
    # a variate not below the last cdf entry finds no bucket (the reference asserts there): FLT_MAX
    assert U.distribution_1d(f, np.array([1.0], np.float32))["x"][0] == np.finfo(np.float32).max


def test_oracle_equals_the_reference_on_random_tables(H, orc_lib, ref_lib):
    r = H.rng(92); O, R = H.Unit("orc"), H.Unit("ref")
    for _ in range(30):
        n = int(r.randint(1, 300)); f = (r.uniform(0, 1, n) ** r.randint(1, 5)).astype(np.float32); f[0] += 0.01
        e = (r.randint(0, 2 ** 24, size=64).astype(np.float32) * np.float32(2.0 ** -24)).astype(np.float32)
        a, b = O.distribution_1d(f, e), R.distribution_1d(f, e)
        assert all((np.array_equal(a[k], b[k]) if a[k].dtype == np.uint32 else H.same_bits(a[k], b[k])) for k in a)
        ny, nx = int(r.randint(1, 20)), int(r.randint(1, 40)); f2 = r.uniform(0.01, 1, (ny, nx)).astype(np.float32)
        e12 = np.stack([e[:32], e[32:]], axis=1)
        a, b = O.distribution_2d(f2, e12), R.distribution_2d(f2, e12)
        assert all(H.same_bits(a[k], b[k]) for k in a)
        seeds = r.randint(0, 2 ** 32, size=3).astype(np.uint32); strata, samples = int(r.randint(1, 9)), int(r.randint(1, 20))
        assert H.same_bits(O.stratified(seeds, strata, samples, strata * strata * samples), R.stratified(seeds, strata, samples, strata * strata * samples))
        first = int(r.randint(0, 2 ** 31 - 100))
        assert H.same_bits(O.halton(first, 64), R.halton(first, 64))


@pytest.mark.gpu
def test_device_equals_the_reference_goldens(H, amd_lib):
    check_against_golden(H, H.Unit("amd"))


@pytest.mark.gpu
def test_device_equals_the_oracle_on_random_and_irregular_tables(H, amd_lib, orc_lib):
    r = H.rng(93); O, A = H.Unit("orc"), H.Unit("amd")
    for it in range(12):
        n = int(r.choice([1, 2, 33, 257, 5000])); f = (r.uniform(0, 1, n) ** 3).astype(np.float32)
        if it % 3 == 1 and n > 2:
            f[r.randint(0, n, size=n // 3)] = 0                     # empty buckets
        if it % 3 == 2 and n > 2:
            f[1] = -0.25                                              # a negative entry: non-monotone cdf, the device takes the linear scan
        e = (r.randint(0, 2 ** 24, size=500).astype(np.float32) * np.float32(2.0 ** -24)).astype(np.float32); e[0] = 1.0
        a, b = O.distribution_1d(f, e), A.distribution_1d(f, e)
        for k in a:
            assert (np.array_equal(a[k], b[k]) if a[k].dtype == np.uint32 else H.same_bits(a[k], b[k])), (it, n, k)
        ny, nx = int(r.choice([1, 3, 64])), int(r.choice([1, 17, 200])); f2 = r.uniform(0, 1, (ny, nx)).astype(np.float32)
        e12 = np.stack([e[:250], e[250:]], axis=1)
        a, b = O.distribution_2d(f2, e12), A.distribution_2d(f2, e12)
        for k in a:
            assert H.same_bits(a[k], b[k]), (it, ny, nx, k)
        seeds = r.randint(0, 2 ** 32, size=70).astype(np.uint32); strata, samples = int(r.randint(1, 9)), int(r.randint(1, 20))
        assert H.same_bits(O.stratified(seeds, strata, samples, strata * strata * samples), A.stratified(seeds, strata, samples, strata * strata * samples))
        first = int(r.randint(0, 2 ** 31 - 3000))
        assert H.same_bits(O.halton(first, 3000), A.halton(first, 3000))
    # all-zero table: integral 0, cdf 0/0 = NaN, nothing is ever found
    z = A.distribution_1d(np.zeros(5, np.float32), np.array([0.0, 0.5], np.float32))
    assert (z["x"] == np.finfo(np.float32).max).all() and np.isnan(z["cdf"]).all()
    # the stratified sampler refuses to run past its strata, as the reference's assert does
    import ctypes as C
    from terra_amd import runtime
    L = runtime.load()
    out = np.zeros((1, 5, 2), np.float32); seeds = np.zeros(1, np.uint32)
    assert L.fn("terra_amd_unit_stratified", C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p])(seeds.ctypes.data, 1, 2, 1, 5, out.ctypes.data) < 0
    assert "asserts" in runtime.last_error(); L.clear_error()
