"""GPU parity, unit level: every device function of the path, called through the
C-ABI's terra_amd_unit_* entry points, against (a) the golden vectors dumped from
the compiled reference and (b) the oracle on fresh seeded inputs. Bit-exact
everywhere (floats compared by bit pattern; NaN results by NaN-ness)."""
import ctypes as C

import numpy as np
import pytest

from terra_amd import api, runtime, scenes

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def L(amd_lib):
    lib = runtime.load()
    assert lib.device_count() > 0, "gpu tests need a visible MI355X: " + runtime.last_error()
    return lib


@pytest.fixture(scope="module")
def U(H, L):
    return H.Unit("amd")


def G(H, name):
    return np.load(H.GOLDEN / f"{name}.npz")


def same(H, a, b):
    a = np.asarray(a, np.float32); b = np.asarray(b, np.float32)
    nan = np.isnan(a)
    return np.array_equal(nan, np.isnan(b)) and np.array_equal(H.bits(a)[~nan], H.bits(b)[~nan])


def test_pcg(H, U):
    g = G(H, "pcg")
    assert H.same_bits(U.pcg(g["seeds"], 64), g["floats"])


def test_stream_keys(H, L, orc_lib):
    r = H.rng(1)
    n = 4096
    pix = r.randint(0, 3840 * 2160, n).astype(np.uint64); k = (r.randint(0, 5000, n) * (r.uniform(size=n) < 0.5)).astype(np.uint64)
    out = np.zeros((n, 3), np.uint64)
    rc = L.fn("terra_amd_unit_stream_keys", C.c_int, [C.c_uint64, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p])(scenes.FRAME_SEED, pix.ctypes.data, k.ctypes.data, n, out.ctypes.data)
    assert rc == 0, runtime.last_error()
    f = orc_lib.fn("orc_pixel_stream_key", None, [C.c_uint64, C.c_uint64, C.c_uint64, C.c_void_p])
    want = np.zeros((n, 3), np.uint64)
    for i in range(n):
        f(scenes.FRAME_SEED, int(pix[i]), int(k[i]), want[i].ctypes.data)
    assert np.array_equal(out, want)


@pytest.mark.parametrize("name", ["camera", "camera_tilted"])
def test_camera(H, U, name):
    g = G(H, name)
    if name == "camera":
        cam = scenes.camera_of(scenes.cornell_box())
    else:
        cam = api.TerraCamera(); cam.position = api.f3((0.3, 1.2, -2.0)); cam.direction = api.f3((0.2, -0.1, 1.0)); cam.up = api.f3((0.05, 1.0, 0.0)); cam.fov = 60.0
    assert H.same_bits(U.camera_dirs(cam, 1920, 1080, g["xy"], float(g["jitter"]), g["r"]), g["dirs"])


def test_ray_aabb(H, U):
    g = G(H, "ray_aabb")
    hit, tmin, tmax = U.ray_aabb(g["o"], g["d"], g["boxes"])
    assert np.array_equal(hit, g["hit"]) and same(H, tmin, g["tmin"]) and same(H, tmax, g["tmax"])


def test_watertight_and_moller_trumbore_golden(H, U):
    g = G(H, "watertight")
    hit, out = U.watertight(g["o"], g["d"], g["tris"])
    assert np.array_equal(hit, g["hit"]) and H.same_bits(out, g["out"])
    g = G(H, "moller_trumbore")
    hit, out = U.moller_trumbore(g["o"], g["d"], g["tris"])
    assert np.array_equal(hit, g["hit"]) and H.same_bits(out, g["out"])


def test_watertight_fresh_inputs_vs_oracle(H, U, orc_lib):
    o, d, t = H.watertight_cases(seed=991, n_random=20000)
    a, b = U.watertight(o, d, t), H.Unit("orc").watertight(o, d, t)
    assert np.array_equal(a[0], b[0]) and H.same_bits(a[1], b[1])
    o, d, boxes = H.aabb_cases(seed=992, n_random=20000)
    a, b = U.ray_aabb(o, d, boxes), H.Unit("orc").ray_aabb(o, d, boxes)
    assert np.array_equal(a[0], b[0]) and same(H, a[1], b[1]) and same(H, a[2], b[2])


def test_bvh_traverse_and_raycast_golden(H, L, U):
    scene = scenes.build_scene(L, scenes.cornell_box(256, 256, 4))
    assert runtime.last_error() == "" or "error" not in runtime.last_error()
    g = G(H, "bvh_traverse")
    found, prim, point = U.bvh_traverse(scene, g["o"], g["d"])
    assert np.array_equal(found, g["found"]) and np.array_equal(prim, g["prim"]) and H.same_bits(point, g["point"])
    g = G(H, "raycast")
    obj, tri, point, surf = U.raycast(scene, g["o"], g["d"])
    assert np.array_equal(obj, g["obj"]) and np.array_equal(tri, g["tri"]) and H.same_bits(point, g["point"])
    m = obj >= 0
    assert H.same_bits(surf[m][:, :22], g["surface"][m][:, :22]) and H.same_bits(surf[m][:, 23:26], g["surface"][m][:, 23:26])
    L.scene_destroy(scene)


@pytest.mark.parametrize("kind_id,name", [(0, "diffuse"), (1, "phong")])
def test_bsdf_golden(H, U, kind_id, name):
    g = G(H, f"bsdf_{name}")
    wi, pdf, f, surf = U.bsdf(kind_id, g["surfaces"], g["e"], g["wo"])
    assert same(H, wi, g["wi"]) and same(H, pdf, g["pdf"]) and same(H, f, g["f"])
    assert H.same_bits(surf[:, 32], g["pick"])


def test_phong_fractional_exponent_nan_lobes_match_oracle(H, U, orc_lib):
    """powf(negative, fractional) is NaN in the reference's Phong pdf/eval; NaN-ness must match"""
    surf, e, wo = H.bsdf_cases(77, 4096, 1)
    surf[:, 29:32] = 7.5
    a, b = U.bsdf(1, surf, e, wo), H.Unit("orc").bsdf(1, surf, e, wo)
    for x, y in zip(a[:3], b[:3]):
        assert same(H, x, y)
    assert np.isnan(a[2]).any()      # eval of a diffuse-picked direction behind the mirror direction: powf(negative, 7.5)


@pytest.mark.parametrize("sname", ["cornell", "phong"])
@pytest.mark.parametrize("integ", range(7))
def test_trace_golden(H, L, U, sname, integ):
    g = G(H, f"trace_{sname}_{integ}")
    mk = scenes.cornell_box if sname == "cornell" else scenes.cornell_phong
    scene = scenes.build_scene(L, mk(64, 64, 1, integrator=integ))
    rad, calls = U.trace(scene, g["o"], g["d"], g["stateB"], g["incB"])
    assert np.array_equal(calls, g["rand_calls"].astype(np.uint32))
    assert same(H, rad, g["radiance"])
    L.scene_destroy(scene)


def test_tonemap_vs_oracle(H, L, orc_lib, devmath_mode):
    r = H.rng(8)
    colors = np.concatenate([r.uniform(0, 20, size=(4000, 3)), r.uniform(0, 0.01, size=(500, 3)), [[0, 0, 0], [1, 1, 1], [0.004, 0.003, 15.0]]]).astype(np.float32)
    f = orc_lib.fn("orc_tonemap", api.TerraFloat3, [C.POINTER(api.TerraFloat3), C.c_int, C.c_float])
    for op in range(5):
        for gamma in (2.2, 1.0, 1.8):
            got = colors.copy()
            rc = L.fn("terra_amd_unit_tonemap", C.c_int, [C.c_int, C.c_float, C.c_int, C.c_void_p])(op, gamma, len(got), got.ctypes.data)
            assert rc == 0, runtime.last_error()
            want = np.array([f(C.byref(api.f3(c)), op, gamma).tuple() for c in colors[:600]], np.float32)
            assert same(H, got[:600], want), (op, gamma)


def test_device_math_vs_oracle_and_libm(H, L, orc_lib):
    r = H.rng(9)
    f = L.fn("terra_amd_unit_math", C.c_int, [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p])
    ev = orc_lib.fn("orc_math_eval", None, [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p])
    n = 300000
    k = r.randint(0, 2 ** 24, n).astype(np.float32) * np.float32(2.0 ** -24)
    cases = {
        0: (np.float32(2 * 3.1416926535) * k, None), 1: (np.float32(2 * 3.1416926535) * k, None),
        2: (np.concatenate([r.uniform(0, 1, n // 2), r.uniform(-1, 1, n // 4), r.uniform(0, 30, n // 4)]).astype(np.float32),
            np.concatenate([1 / (1 + np.round(r.uniform(0, 80, n // 2))), np.round(r.uniform(0, 60, n // 4)), np.full(n // 4, 1 / 2.2)]).astype(np.float32)),
        3: (r.uniform(-1.01, 1.01, n).astype(np.float32), None),
        4: (np.concatenate([r.uniform(-1, 1, n - 8), [0, -0.0, 0, 1, -1, 1e-30, 3e30, -2]]).astype(np.float32),
            np.concatenate([r.uniform(-1, 1, n - 8), [1, -1, 0, 0, 0, -3e30, 1e-30, 1]]).astype(np.float32)),
    }
    for fn, (x, y) in cases.items():
        x = np.ascontiguousarray(x, np.float32); yy = np.ascontiguousarray(y if y is not None else x, np.float32)
        got = np.zeros_like(x); dm = np.zeros_like(x); lm = np.zeros_like(x)
        assert f(fn, len(x), x.ctypes.data, yy.ctypes.data, got.ctypes.data) == 0, runtime.last_error()
        ev(fn, 1, len(x), x.ctypes.data, yy.ctypes.data, dm.ctypes.data)
        ev(fn, 0, len(x), x.ctypes.data, yy.ctypes.data, lm.ctypes.data)
        assert same(H, got, dm), fn
        assert same(H, got, lm), fn
