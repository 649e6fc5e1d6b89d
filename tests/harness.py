"""Shared test plumbing: library loading, one uniform Python face over the three
implementations' unit-level entry points, and deterministic input generators.

Backends
  "ref" : the compiled reference, oracle/_ref/libterra_ref.so (only where /root/reference exists)
  "orc" : the CPU restatement, oracle/liboracle.so
  "amd" : the product, terra_amd/libterra_amd.so (device entry points need a GPU)

Only tests (and smoke/bench's checker leg) may touch oracle/.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

from terra_amd import api, scenes  # noqa: E402

REFERENCE_DIR = Path("/root/reference")
REF_SO = ROOT / "oracle" / "_ref" / "libterra_ref.so"
ORC_SO = ROOT / "oracle" / "liboracle.so"
AMD_SO = ROOT / "terra_amd" / "libterra_amd.so"
GOLDEN = ROOT / "tests" / "golden"

F3P = C.POINTER(api.TerraFloat3)
c_f = C.c_float
c_sz = C.c_size_t
RENDER_PIXELS_SIG = [C.POINTER(api.TerraCamera), C.c_void_p, C.POINTER(api.TerraFramebuffer), c_sz, c_sz, c_sz, c_sz, C.c_uint64, C.c_void_p]


def have_reference() -> bool:
    return REFERENCE_DIR.is_dir()


def build_oracle() -> None:
    subprocess.run(["make", "-C", str(ROOT / "oracle")], check=True, capture_output=True)


def build_reference() -> None:
    subprocess.run(["make", "-C", str(ROOT / "oracle"), "ref"], check=True, capture_output=True)


_libs = {}


def lib(kind: str) -> api.TerraLib:
    if kind in _libs:
        return _libs[kind]
    if kind == "orc":
        if not ORC_SO.exists():
            build_oracle()
        L = api.TerraLib(ORC_SO, "orc_")
    elif kind == "ref":
        if not REF_SO.exists():
            build_reference()
        L = api.TerraLib(REF_SO, "terra_")
    elif kind == "amd":
        try:
            import torch  # noqa: F401  -- first: the library must bind to the HIP runtime torch loads (terra_amd/runtime.py)
        except ImportError:
            pass
        if not AMD_SO.exists():
            from terra_amd import build as _b
            _b.build()
        L = api.TerraLib(AMD_SO, "terra_")
    else:
        raise KeyError(kind)
    _libs[kind] = L
    return L


def bits(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def same_bits(a, b) -> bool:
    return np.array_equal(bits(a), bits(b))


def fnv1a(arr: np.ndarray) -> int:
    """64-bit FNV-1a over the bytes of arr (vectorised per byte position is not possible; small inputs only)."""
    h = 0xCBF29CE484222325
    for byte in np.ascontiguousarray(arr).view(np.uint8).ravel().tolist():
        h = ((h ^ byte) * 0x100000001B3) & 0xFFFFFFFFFFFFFFFF
    return h


def digest(arr: np.ndarray) -> str:
    import hashlib
    return hashlib.sha256(np.ascontiguousarray(arr).tobytes()).hexdigest()


# ---------------------------------------------------------------------------
# uniform unit-level face
# ---------------------------------------------------------------------------

class Unit:
    """Unit-level entry points of one backend with numpy in / numpy out."""

    def __init__(self, kind: str):
        self.kind = kind
        self.L = lib(kind)
        self.p = {"ref": "ref_", "orc": "orc_", "amd": "terra_amd_unit_"}[kind]

    # -- helpers
    def _f(self, name, res, args):
        return self.L.fn(self.p + name, res, args)

    @staticmethod
    def _v(a, i):
        return api.TerraFloat3(float(a[i, 0]), float(a[i, 1]), float(a[i, 2]))

    # -- PCG
    def pcg(self, seeds, n):
        seeds = np.asarray(seeds, np.uint32)
        out = np.zeros((len(seeds), n), np.float32)
        if self.kind == "amd":
            rc = self._f("pcg", C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p])(seeds.ctypes.data, len(seeds), n, out.ctypes.data)
            assert rc == 0, last_error()
        else:
            f = self._f("pcg_floats", None, [C.c_uint32, C.c_int, C.c_void_p])
            for i, s in enumerate(seeds):
                f(int(s), n, out[i].ctypes.data)
        return out

    def ray_aabb(self, o, d, boxes):
        n = len(o)
        hit = np.zeros(n, np.int32); tmin = np.zeros(n, np.float32); tmax = np.zeros(n, np.float32)
        if self.kind == "amd":
            rc = self._f("ray_aabb", C.c_int, [C.c_int] + [C.c_void_p] * 6)(n, o.ctypes.data, d.ctypes.data, boxes.ctypes.data, hit.ctypes.data, tmin.ctypes.data, tmax.ctypes.data)
            assert rc == 0, last_error()
        else:
            f = self._f("ray_aabb", C.c_int, [F3P, F3P, C.POINTER(api.TerraAABB), C.POINTER(c_f), C.POINTER(c_f)])
            for i in range(n):
                box = api.TerraAABB(self._v(boxes[:, :3], i), self._v(boxes[:, 3:], i))
                a, b = c_f(0), c_f(0)
                hit[i] = f(C.byref(self._v(o, i)), C.byref(self._v(d, i)), C.byref(box), C.byref(a), C.byref(b))
                if hit[i]:
                    tmin[i], tmax[i] = a.value, b.value
        return hit, tmin, tmax

    def _tri_call(self, name, width, o, d, tris):
        n = len(o)
        hit = np.zeros(n, np.int32); out = np.zeros((n, width), np.float32)
        if self.kind == "amd":
            rc = self._f(name, C.c_int, [C.c_int] + [C.c_void_p] * 5)(n, o.ctypes.data, d.ctypes.data, tris.ctypes.data, hit.ctypes.data, out.ctypes.data)
            assert rc == 0, last_error()
        else:
            f = self._f(name, C.c_int, [F3P, F3P, C.c_void_p, C.c_void_p])
            for i in range(n):
                hit[i] = f(C.byref(self._v(o, i)), C.byref(self._v(d, i)), tris[i].ctypes.data, out[i].ctypes.data)
                if not hit[i]:
                    out[i] = 0
        return hit, out

    def watertight(self, o, d, tris):
        return self._tri_call("watertight", 8, o, d, tris)

    def moller_trumbore(self, o, d, tris):
        return self._tri_call("moller_trumbore", 4, o, d, tris)

    def bvh_traverse(self, scene, o, d):
        n = len(o)
        found = np.zeros(n, np.int32); prim = np.zeros(n, np.uint32); point = np.zeros((n, 3), np.float32)
        if self.kind == "amd":
            rc = self._f("bvh_traverse", C.c_int, [C.c_void_p, C.c_int] + [C.c_void_p] * 5)(scene, n, o.ctypes.data, d.ctypes.data, found.ctypes.data, prim.ctypes.data, point.ctypes.data)
            assert rc == 0, last_error()
        else:
            f = self._f("bvh_traverse", C.c_int, [C.c_void_p, F3P, F3P, F3P, C.POINTER(C.c_uint32)])
            for i in range(n):
                p = api.TerraFloat3(); pr = C.c_uint32(0)
                found[i] = f(scene, C.byref(self._v(o, i)), C.byref(self._v(d, i)), C.byref(p), C.byref(pr))
                prim[i] = pr.value if found[i] else 0
                point[i] = p.tuple()
        return found, prim, point

    def bvh_traverse_fast(self, scene, o, d):
        """the fast tree's traversal (device only): found, prim, point as bvh_traverse, plus nodes visited per ray"""
        assert self.kind == "amd"
        n = len(o)
        found = np.zeros(n, np.int32); prim = np.zeros(n, np.uint32); point = np.zeros((n, 3), np.float32); nodes = np.zeros(n, np.uint32)
        rc = self._f("bvh_traverse_fast", C.c_int, [C.c_void_p, C.c_int] + [C.c_void_p] * 6)(scene, n, o.ctypes.data, d.ctypes.data, found.ctypes.data, prim.ctypes.data, point.ctypes.data, nodes.ctypes.data)
        assert rc == 0, last_error()
        return found, prim, point, nodes

    def raycast(self, scene, o, d):
        n = len(o)
        obj = np.zeros(n, np.int32); tri = np.zeros(n, np.int32); point = np.zeros((n, 3), np.float32); surf = np.zeros((n, 47), np.float32)
        if self.kind == "amd":
            rc = self._f("raycast", C.c_int, [C.c_void_p, C.c_int] + [C.c_void_p] * 6)(scene, n, o.ctypes.data, d.ctypes.data, obj.ctypes.data, tri.ctypes.data, point.ctypes.data, surf.ctypes.data)
            assert rc == 0, last_error()
        else:
            f = self._f("raycast", C.c_int, [C.c_void_p, F3P, F3P, C.POINTER(api.TerraShadingSurface), F3P, C.POINTER(C.c_int)])
            for i in range(n):
                s = api.TerraShadingSurface(); p = api.TerraFloat3(); t = C.c_int(0)
                obj[i] = f(scene, C.byref(self._v(o, i)), C.byref(self._v(d, i)), C.byref(s), C.byref(p), C.byref(t))
                point[i] = p.tuple()
                if obj[i] >= 0:
                    tri[i] = t.value
                    surf[i] = np.frombuffer(bytes(s), dtype=np.float32)
        return obj, tri, point, surf

    # -- samplers and distributions (unit level)
    def stratified(self, seeds, strata, samples, n):
        seeds = np.asarray(seeds, np.uint32)
        out = np.zeros((len(seeds), n, 2), np.float32)
        if self.kind == "amd":
            rc = self._f("stratified", C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p])(seeds.ctypes.data, len(seeds), strata, samples, n, out.ctypes.data)
            assert rc == 0, last_error()
        else:
            f = self._f("stratified_pairs", None, [C.c_uint32, C.c_int, C.c_int, C.c_int, C.c_void_p])
            for i, sd in enumerate(seeds):
                f(int(sd), strata, samples, n, out[i].ctypes.data)
        return out

    def halton(self, first, n):
        out = np.zeros((n, 2), np.float32)
        if self.kind == "amd":
            rc = self._f("halton", C.c_int, [C.c_int, C.c_int, C.c_void_p])(first, n, out.ctypes.data)
            assert rc == 0, last_error()
        else:
            self._f("halton_pairs", None, [C.c_int, C.c_int, C.c_void_p])(first, n, out.ctypes.data)
        return out

    def distribution_1d(self, f, e):
        f = np.ascontiguousarray(f, np.float32); e = np.ascontiguousarray(e, np.float32); n, m = len(f), len(e)
        x = np.zeros(m, np.float32); pdf = np.zeros(m, np.float32); idx = np.zeros(m, np.uint32); cdf = np.zeros(n, np.float32); integral = np.zeros(1, np.float32)
        sig = [C.c_void_p, c_sz, C.c_void_p, C.c_int] + [C.c_void_p] * 5
        res = self._f("distribution_1d", C.c_int if self.kind == "amd" else None, sig)(f.ctypes.data, n, e.ctypes.data, m, x.ctypes.data, pdf.ctypes.data, idx.ctypes.data, cdf.ctypes.data, integral.ctypes.data)
        if self.kind == "amd":
            assert res == 0, last_error()
        return dict(x=x, pdf=pdf, idx=idx, cdf=cdf, integral=integral)

    def distribution_2d(self, f, e12):
        f = np.ascontiguousarray(f, np.float32); e12 = np.ascontiguousarray(e12, np.float32); ny, nx = f.shape; m = len(e12)
        xy = np.zeros((m, 2), np.float32); pdf = np.zeros(m, np.float32); mcdf = np.zeros(ny, np.float32)
        sig = [C.c_void_p, c_sz, c_sz, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        res = self._f("distribution_2d", C.c_int if self.kind == "amd" else None, sig)(f.ctypes.data, nx, ny, e12.ctypes.data, m, xy.ctypes.data, pdf.ctypes.data, mcdf.ctypes.data)
        if self.kind == "amd":
            assert res == 0, last_error()
        return dict(xy=xy, pdf=pdf, marginal_cdf=mcdf)

    def trace(self, scene, o, d, stateB, incB):
        n = len(o)
        rad = np.zeros((n, 3), np.float32); calls = np.zeros(n, np.uint32)
        if self.kind == "amd":
            rc = self._f("trace", C.c_int, [C.c_void_p, C.c_int] + [C.c_void_p] * 6)(scene, n, o.ctypes.data, d.ctypes.data, stateB.ctypes.data, incB.ctypes.data, rad.ctypes.data, calls.ctypes.data)
            assert rc == 0, last_error()
        else:
            f = self._f("trace_one", api.TerraFloat3, [C.c_void_p, F3P, F3P, C.c_uint64, C.c_uint64, C.POINTER(C.c_uint32)])
            for i in range(n):
                c = C.c_uint32(0)
                L = f(scene, C.byref(self._v(o, i)), C.byref(self._v(d, i)), int(stateB[i]), int(incB[i]), C.byref(c))
                rad[i] = L.tuple(); calls[i] = c.value
        return rad, calls

    def bsdf(self, kind_id, surfaces47, e, wo):
        """sample, then pdf and eval at the sampled direction; returns wi, pdf, f and the (possibly modified) surfaces."""
        n = len(e)
        surf = np.array(surfaces47, np.float32, copy=True)
        wi = np.zeros((n, 3), np.float32); pdf = np.zeros(n, np.float32); f = np.zeros((n, 3), np.float32)
        if self.kind == "amd":
            rc = self._f("bsdf", C.c_int, [C.c_int, C.c_int] + [C.c_void_p] * 6)(kind_id, n, surf.ctypes.data, e.ctypes.data, wo.ctypes.data, wi.ctypes.data, pdf.ctypes.data, f.ctypes.data)
            assert rc == 0, last_error()
            return wi, pdf, f, surf
        b = api.TerraBSDF()
        [self.L.bsdf_diffuse_init, self.L.bsdf_phong_init, getattr(self.L, "bsdf_ggx_init", None), getattr(self.L, "bsdf_glass_init", None)][kind_id](C.byref(b))
        SP = C.POINTER(api.TerraShadingSurface)
        fs = C.CFUNCTYPE(api.TerraFloat3, SP, c_f, c_f, c_f, F3P)(b.sample)
        fp = C.CFUNCTYPE(c_f, SP, F3P, F3P)(b.pdf)
        fe = C.CFUNCTYPE(api.TerraFloat3, SP, F3P, F3P)(b.eval)
        for i in range(n):
            s = api.TerraShadingSurface.from_buffer_copy(surf[i].tobytes())
            w = self._v(wo, i)
            r = fs(C.byref(s), float(e[i, 0]), float(e[i, 1]), float(e[i, 2]), C.byref(w))
            wi[i] = r.tuple()
            pdf[i] = fp(C.byref(s), C.byref(r), C.byref(w))
            f[i] = fe(C.byref(s), C.byref(r), C.byref(w)).tuple()
            surf[i] = np.frombuffer(bytes(s), dtype=np.float32)
        return wi, pdf, f, surf

    def camera_dirs(self, cam: api.TerraCamera, W, H, xy, jitter, r):
        """world-space primary directions (camera sample rotated by the camera frame)."""
        n = len(xy)
        out = np.zeros((n, 3), np.float32)
        if self.kind == "amd":
            xy = np.ascontiguousarray(xy, np.uint32); r = np.ascontiguousarray(r, np.float32)
            rc = self._f("camera", C.c_int, [C.POINTER(api.TerraCamera), c_sz, c_sz, C.c_int, C.c_void_p, c_f, C.c_void_p, C.c_void_p])(C.byref(cam), W, H, n, xy.ctypes.data, jitter, r.ctypes.data, out.ctypes.data)
            assert rc == 0, last_error()
            return out
        if self.kind == "ref":
            fb = api.TerraFramebuffer(); fb.width = W; fb.height = H
            fs = self.L.fn("terra_camera_perspective_sample", api.TerraFloat3, [C.POINTER(api.TerraCamera), C.POINTER(api.TerraFramebuffer), c_sz, c_sz, c_f, c_f, c_f])
            ff = self.L.fn("terra_camera_to_world_frame", api.TerraFloat4x4, [C.POINTER(api.TerraCamera)])
            sample = lambda x, y, a, b: fs(C.byref(cam), C.byref(fb), x, y, jitter, a, b)
        else:
            fs = self.L.fn("orc_camera_sample", api.TerraFloat3, [C.POINTER(api.TerraCamera), c_sz, c_sz, c_sz, c_sz, c_f, c_f, c_f])
            ff = self.L.fn("orc_camera_frame", api.TerraFloat4x4, [C.POINTER(api.TerraCamera)])
            sample = lambda x, y, a, b: fs(C.byref(cam), W, H, x, y, jitter, a, b)
        m = ff(C.byref(cam))
        rows = np.array([[m.rows[i].x, m.rows[i].y, m.rows[i].z] for i in range(3)], np.float32)
        for i in range(n):
            v = sample(int(xy[i, 0]), int(xy[i, 1]), float(r[i, 0]), float(r[i, 1]))
            v = np.array(v.tuple(), np.float32)
            # terra_transformf3: row . v with float32 left-to-right sums
            out[i] = [np.float32(np.float32(np.float32(rows[k, 0] * v[0]) + np.float32(rows[k, 1] * v[1])) + np.float32(rows[k, 2] * v[2])) for k in range(3)]
        return out

    def bvh_nodes(self, scene) -> np.ndarray:
        """reference-layout node array as (n, 16) uint32 words"""
        if self.kind == "amd":
            f = self.L.fn("terra_amd_scene_bvh_nodes", C.c_int, [C.c_void_p, C.c_void_p, C.c_int])
            n = f(scene, None, 0)
            out = np.zeros((n, 16), np.uint32)
            assert f(scene, out.ctypes.data, n) == n
            return out
        n = self._f("bvh_node_count", C.c_int, [C.c_void_p])(scene)
        ptr = self._f("bvh_nodes", C.c_void_p, [C.c_void_p])(scene)
        buf = (C.c_uint32 * (16 * n)).from_address(ptr)
        return np.frombuffer(buf, dtype=np.uint32).reshape(n, 16).copy()

    def render_pixels(self, d: scenes.SceneDesc, passes=1, frame_seed=scenes.FRAME_SEED, rect=None, want_calls=True, threads=0, sum_calls=False):
        """Full per-pixel-stream render through this backend's CPU path (ref/orc only).
        threads > 0: the *_render_pixels_mt entry (rows dealt to pthreads; per-pixel streams make the result thread independent)."""
        assert self.kind in ("ref", "orc")
        if threads > 0:
            fmt = self._f("render_pixels_mt", None, RENDER_PIXELS_SIG + [C.c_int])
            f = lambda *a: fmt(*a, threads)
        else:
            f = self._f("render_pixels", None, RENDER_PIXELS_SIG)
        scene = scenes.build_scene(self.L, d)
        fb = api.Framebuffer(self.L, d.width, d.height)
        cam = scenes.camera_of(d)
        calls = np.zeros((d.height, d.width), np.uint32)
        x, y, w, h = rect if rect else (0, 0, d.width, d.height)
        total = np.zeros((d.height, d.width), np.uint64)
        for _ in range(passes):
            f(C.byref(cam), scene, C.byref(fb.fb), x, y, w, h, frame_seed, calls.ctypes.data if want_calls else None)
            total += calls
        # rand_calls: the LAST pass's per-pixel stream-B draws (what a plain device call reports), or with sum_calls the
        # total over the passes (what one split device call reports: its chunks are the passes)
        out = dict(pixels=fb.pixels.copy(), acc=fb.results["acc"].copy(), samples=fb.results["samples"].copy(), rand_calls=total if sum_calls else calls)
        fb.destroy()
        self.L.scene_destroy(scene)
        return out


def last_error() -> str:
    L = lib("amd")
    return L.fn("terra_amd_last_error", C.c_char_p, [])().decode()


def set_oracle_math(mode: int) -> None:
    lib("orc").fn("orc_set_math_mode", None, [C.c_int])(mode)


# ---------------------------------------------------------------------------
# deterministic inputs (legacy RandomState: bit-stable across numpy versions)
# ---------------------------------------------------------------------------

def rng(seed):
    return np.random.RandomState(seed)


def unit_dirs(r, n):
    v = r.normal(size=(n, 3)).astype(np.float32)
    v /= np.linalg.norm(v, axis=1, keepdims=True).astype(np.float32)
    return np.ascontiguousarray(v, np.float32)


def watertight_cases(seed=11, n_random=3000):
    """(origins, dirs, tris): random pairs + rays aimed at random points inside the triangle
    + exact edge/vertex/coplanar/behind/axis-tie/negative-zero cases."""
    r = rng(seed)
    o = r.uniform(-2, 2, size=(n_random, 3)).astype(np.float32)
    tris = r.uniform(-2, 2, size=(n_random, 3, 3)).astype(np.float32)
    bary = r.dirichlet([1, 1, 1], size=n_random).astype(np.float32)
    target = np.einsum("nk,nkc->nc", bary, tris).astype(np.float32)
    d = (target - o).astype(np.float32)
    half = n_random // 2
    d[half:] = unit_dirs(r, n_random - half)                       # second half: random directions (mostly misses)
    d[: half // 2] /= np.linalg.norm(d[: half // 2], axis=1, keepdims=True)   # first quarter normalised, second quarter not
    special_o, special_d, special_t = [], [], []
    T = [[0, 0, 1], [1, 0, 1], [0, 1, 1]]
    def add(oo, dd, tt=T):
        special_o.append(oo); special_d.append(dd); special_t.append(tt)
    add([0.5, 0, 0], [0, 0, 1])            # edge AB: one barycentric exactly 0 -> double fallback
    add([0, 0.5, 0], [0, 0, 1])            # edge CA
    add([0.5, 0.5, 0], [0, 0, 1])          # edge BC
    add([0, 0, 0], [0, 0, 1])              # vertex A
    add([1, 0, 0], [0, 0, 1])              # vertex B
    add([0, 1, 0], [0, 0, 1])              # vertex C
    add([0.25, 0.25, 0], [0, 0, 1])        # interior
    add([0.25, 0.25, 2], [0, 0, 1])        # behind origin
    add([0.25, 0.25, 2], [0, 0, -1])       # negative major axis (x/y swap)
    add([0.25, 0.25, 0], [0, 0, -1])       # pointing away
    add([-1, 0.25, 1], [1, 0, 0])          # coplanar ray: det == 0
    add([0.25, 0.25, 1], [0, 0, 1])        # origin on the triangle: depth 0 passes
    add([0.25, 0.25, 0], [1, 1, 1])        # all |d| equal: tie -> z
    add([0.25, 0.25, 0], [1, 1, 0.5])      # x == y tie -> y
    add([0.25, 0.25, 0], [-0.0, 0.0, 1])   # negative zero component
    add([0.5, 0, 0], [0, 0, 2])            # unnormalised
    add([0.5, -1e-8, 0], [0, 0, 1])        # just outside the edge
    add([0.5, 1e-8, 0], [0, 0, 1])         # just inside
    add([0.3, 0.3, 0], [1e-3, -1e-3, 1])
    add([0.25, 0.25, 0], [0, 0, 1], [[0, 0, 1], [0, 1, 1], [1, 0, 1]])   # opposite winding
    add([0.25, 0.25, 0], [0, 0, 1], [[0, 0, 1], [0, 0, 1], [1, 0, 1]])   # degenerate triangle
    so = np.array(special_o, np.float32); sd = np.array(special_d, np.float32); st = np.array(special_t, np.float32)
    return (np.ascontiguousarray(np.concatenate([o, so])), np.ascontiguousarray(np.concatenate([d, sd])),
            np.ascontiguousarray(np.concatenate([tris, st]).reshape(-1, 9)))


def aabb_cases(seed=12, n_random=2000):
    r = rng(seed)
    o = r.uniform(-3, 3, size=(n_random, 3)).astype(np.float32)
    lo = r.uniform(-2, 1, size=(n_random, 3)).astype(np.float32)
    hi = (lo + r.uniform(0.01, 2, size=(n_random, 3))).astype(np.float32)
    d = unit_dirs(r, n_random)
    aim = n_random // 2
    tgt = (lo[:aim] + (hi[:aim] - lo[:aim]) * r.uniform(0, 1, size=(aim, 3))).astype(np.float32)
    d[:aim] = (tgt - o[:aim]).astype(np.float32)
    so, sd, sb = [], [], []
    B = [-1, -1, -1, 1, 1, 1]
    def add(oo, dd, bb=B):
        so.append(oo); sd.append(dd); sb.append(bb)
    add([0, 0, -3], [0, 0, 1])            # axis parallel: two inv components are +inf
    add([0, 0, -3], [0, 0, -1])           # pointing away
    add([2, 0, -3], [0, 0, 1])            # parallel, outside the slab
    add([1, 0, -3], [0, 0, 1])            # parallel, ON the slab plane: 0 * inf = NaN
    add([-1, 0, -3], [0, 0, 1])           # on the min plane
    add([1, 1, -3], [0, 0, 1])            # on an edge
    add([0, 0, 0], [0, 0, 1])             # origin inside
    add([0, 0, 0], [1, 1, 1])
    add([0, 0, 1], [0, 0, 1])             # origin on the exit plane: tmax = 0
    add([0, 0, -3], [-0.0, 0.0, 1])       # -0 -> -inf
    add([0, 0, -3], [1e-30, 0, 1])        # huge inv
    add([0, 0, -3], [0, 0, 1], [-1, -1, -1, -1, 1, 1])    # zero-thickness box
    add([0, 0, -3], [0, 0, 1], [3.4e38, 3.4e38, 3.4e38, -3.4e38, -3.4e38, -3.4e38])   # empty (inverted) box
    o = np.concatenate([o, np.array(so, np.float32)]); d = np.concatenate([d, np.array(sd, np.float32)])
    boxes = np.concatenate([np.concatenate([lo, hi], axis=1), np.array(sb, np.float32)])
    return np.ascontiguousarray(o, np.float32), np.ascontiguousarray(d, np.float32), np.ascontiguousarray(boxes, np.float32)


def scene_rays(seed, n, box=((-1.2, -0.2, -3.5), (1.2, 2.2, 1.2))):
    """rays with origins in/around the Cornell room; half aimed at the room's interior"""
    r = rng(seed)
    lo, hi = np.array(box[0], np.float32), np.array(box[1], np.float32)
    o = (lo + (hi - lo) * r.uniform(0, 1, size=(n, 3))).astype(np.float32)
    d = unit_dirs(r, n)
    tgt = (np.array([-1, 0, -1], np.float32) + np.array([2, 2, 2], np.float32) * r.uniform(0, 1, size=(n // 2, 3))).astype(np.float32)
    dd = (tgt - o[: n // 2]).astype(np.float32)
    dd /= np.linalg.norm(dd, axis=1, keepdims=True)
    d[: n // 2] = dd
    # a few axis-parallel rays from inside the room
    k = min(16, n)
    o[-k:] = np.array([0.1, 1.0, -0.3], np.float32)
    axes = np.array([[1, 0, 0], [-1, 0, 0], [0, 1, 0], [0, -1, 0], [0, 0, 1], [0, 0, -1], [1, 1, 0], [0, 1, 1]], np.float32)
    d[-k:] = np.resize(axes, (k, 3))
    return np.ascontiguousarray(o, np.float32), np.ascontiguousarray(d, np.float32)


def stream_states(seed, n):
    r = rng(seed)
    st = r.randint(0, 2 ** 32, size=(n, 2)).astype(np.uint64)
    state = (st[:, 0] << np.uint64(32)) | st[:, 1]
    inc = (r.randint(0, 2 ** 31, size=n).astype(np.uint64) << np.uint64(1)) | np.uint64(1)
    return np.ascontiguousarray(state), np.ascontiguousarray(inc)


def bsdf_cases(seed, n, kind_id):
    """surfaces (47 floats: basis of a random unit normal via the backend-independent formula), variates, wo"""
    r = rng(seed)
    nrm = unit_dirs(r, n)
    nrm[:6] = np.array([[0, 1, 0], [0, -1, 0], [1, 0, 0], [-1, 0, 0], [0, 0, 1], [0, 0, -1]], np.float32)
    surf = np.zeros((n, 47), np.float32)
    for i in range(n):
        m = api.TerraShadingSurface()
        basis = _basis(nrm[i])
        surf[i, 0:16] = basis
    surf[:, 16:19] = nrm
    surf[:, 19:22] = 0
    surf[:, 22] = 1.5
    if kind_id == 0:
        surf[:, 23:26] = r.uniform(0, 1, size=(n, 3))
    elif kind_id == 2:      # GGX: F0, roughness
        surf[:, 23:26] = r.uniform(0.2, 1, size=(n, 3))
        surf[:, 26:29] = r.uniform(0.02, 0.9, size=(n, 1))
    elif kind_id == 3:      # glass: tint, ior in slot 22, scratch slots zero
        surf[:, 23:26] = r.uniform(0.5, 1, size=(n, 3))
        surf[:, 22] = r.uniform(1.1, 2.4, size=n)
    else:
        surf[:, 23:26] = r.uniform(0, 1, size=(n, 3))       # specular colour
        surf[:, 26:29] = r.uniform(0, 1, size=(n, 3))       # albedo
        surf[:, 29:32] = np.round(r.uniform(1, 60, size=(n, 1)))   # integral exponents (fractional ones give NaN lobes in the reference)
    e = (r.randint(0, 2 ** 24, size=(n, 3)).astype(np.float32) * np.float32(2.0 ** -24)).astype(np.float32)
    wo = unit_dirs(r, n)
    flip = np.einsum("nc,nc->n", wo, nrm) < 0
    if kind_id == 3:
        flip[n // 2:] = ~flip[n // 2:]          # glass: half of the cases arrive from inside the medium
    wo[flip] = -wo[flip]
    return np.ascontiguousarray(surf), np.ascontiguousarray(e), np.ascontiguousarray(wo)


def _basis(n):
    """terra_f4x4_basis in float32 (reference include/TerraMath.inl:251-272), row-major 16 floats"""
    f = np.float32
    nx, ny, nz = f(n[0]), f(n[1]), f(n[2])
    if abs(nx) > abs(ny):
        k = np.sqrt(f(f(nx * nx) + f(nz * nz)), dtype=np.float32)
        t = np.array([f(nz * k), f(f(0) * k), f(f(-nx) * k)], np.float32)
    else:
        k = np.sqrt(f(f(ny * ny) + f(nz * nz)), dtype=np.float32)
        t = np.array([f(f(0) * k), f(f(-nz) * k), f(ny * k)], np.float32)
    b = np.array([f(f(ny * t[2]) - f(nz * t[1])), f(f(nz * t[0]) - f(nx * t[2])), f(f(nx * t[1]) - f(ny * t[0]))], np.float32)
    return np.array([t[0], nx, b[0], 0, t[1], ny, b[1], 0, t[2], nz, b[2], 0, 0, 0, 0, 1], np.float32)


def sampler_cases():
    """inputs of the N4 fixtures (tests/golden/samplers.npz): table sizes from 1 to 1000, tables with zero buckets, variates on the
    u24 grid incl. 0 and the largest value below 1"""
    r = rng(91)
    u24 = lambda n: (r.randint(0, 2 ** 24, size=n).astype(np.float32) * np.float32(2.0 ** -24)).astype(np.float32)
    tables = {"n1": np.array([2.5], np.float32), "n7": r.uniform(0.1, 3, 7).astype(np.float32), "n64": r.uniform(0, 1, 64).astype(np.float32),
              "n1000": (r.uniform(0, 1, 1000) ** 4).astype(np.float32), "zeros": np.array([0, 0, 1, 0, 2, 0, 0, 3, 0], np.float32)}
    e = u24(400); e[:3] = [0.0, np.float32(1) - np.float32(2.0 ** -24), 0.5]
    t2 = {"t16x8": r.uniform(0, 1, (8, 16)).astype(np.float32), "t64x32": (r.uniform(0, 1, (32, 64)) ** 3).astype(np.float32), "t5x3": np.array([[0, 1, 0, 2, 0], [3, 0, 0, 0, 1], [0.5, 0.5, 0.5, 0.5, 0.5]], np.float32)}
    e12 = np.stack([u24(300), u24(300)], axis=1); e12[:2] = [[0.0, 0.0], [np.float32(1) - np.float32(2.0 ** -24)] * 2]
    return tables, e, t2, np.ascontiguousarray(e12, np.float32)
