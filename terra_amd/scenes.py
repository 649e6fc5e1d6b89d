"""Synthetic scenes of BASELINE.json's configs, as plain data, and the code that
feeds them through the Terra.h API (`build_scene`), exactly as a C client would:
terra_scene_create -> terra_scene_add_object (fill in place) ->
terra_scene_get_options -> terra_scene_commit  (reference src/Terra.c:130-255).

Scene definitions follow SURVEY.md section 8d. All geometry is generated from
integer arithmetic / an integer hash so every backend (reference, oracle,
device) sees bit-identical float inputs without any files.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field
from typing import List, Optional

import numpy as np

from . import api

FRAME_SEED = 0x5EED0001
LAST_COMMIT_MS = 0.0


@dataclass
class Material:
    kind: str = "diffuse"                 # "diffuse" | "phong" | "ggx" | "glass" (the last two: this repo's own presets)
    roughness: float = 0.2
    albedo_texture: Optional["TextureDesc"] = None       # replaces the constant albedo (slot 0 diffuse, slot 1 Phong)
    emissive_texture: Optional["TextureDesc"] = None
    albedo: tuple = (0.73, 0.73, 0.73)
    emissive: tuple = (0.0, 0.0, 0.0)
    specular_color: tuple = (0.0, 0.0, 0.0)
    specular_intensity: float = 1.0
    ior: float = 1.5


@dataclass
class TextureDesc:
    """A TerraTexture: data (h, w, c) uint8 or float32, c <= 3; lookups are in TEXEL units (reference src/Terra.c:410-414)."""
    data: np.ndarray
    filter: int = 0            # kTerraFilterPoint / 1 = bilinear
    address_mode: int = 0      # wrap / 1 mirror / 2 clamp


@dataclass
class ObjectDesc:
    triangles: np.ndarray                  # (n, 3, 3) float32: a, b, c
    normals: np.ndarray                    # (n, 3, 3) float32: per-vertex normals
    texcoords: np.ndarray                  # (n, 3, 2) float32
    material: Material = field(default_factory=Material)
    name: str = ""


@dataclass
class SceneDesc:
    objects: List[ObjectDesc]
    camera_position: tuple = (0.0, 1.0, -3.4)
    camera_direction: tuple = (0.0, 0.0, 1.0)
    camera_up: tuple = (0.0, 1.0, 0.0)
    camera_fov: float = 45.0
    width: int = 256
    height: int = 256
    spp: int = 4
    bounces: int = 8
    integrator: int = api.kTerraIntegratorSimple
    tonemap: int = api.kTerraTonemappingOperatorNone
    sampling: int = api.kTerraSamplingMethodRandom
    strata: int = 4
    jitter: float = 0.5
    exposure: float = 1.0
    gamma: float = 2.2
    environment: tuple = (0.0, 0.0, 0.0)
    environment_texture: Optional["TextureDesc"] = None   # lat-long map bound with terra_attribute_init_cubemap
    environment_lighting: bool = False     # extension (terra_amd_/orc_set_environment_lighting); the reference drops the environment term
    environment_sampling: bool = False     # extension (terra_amd_/orc_set_environment_sampling): Direct / Direct+MIS sample a lat-long environment through a TerraDistribution2D
    sampler_integration: bool = False      # extension (terra_amd_/orc_set_sampler_integration); the reference never draws from the pixel sampler it constructs
    name: str = "scene"

    @property
    def triangle_count(self) -> int:
        return int(sum(len(o.triangles) for o in self.objects))


# --------------------------------------------------------------------------
# geometry helpers
# --------------------------------------------------------------------------

def _quad(p0, p1, p2, p3, n):
    """Two triangles (p0,p1,p2),(p0,p2,p3) with constant vertex normal n."""
    tris = np.array([[p0, p1, p2], [p0, p2, p3]], dtype=np.float32)
    nrm = np.broadcast_to(np.asarray(n, dtype=np.float32), (2, 3, 3)).copy()
    return tris, nrm


def _merge(parts):
    tris = np.concatenate([p[0] for p in parts]).astype(np.float32)
    nrm = np.concatenate([p[1] for p in parts]).astype(np.float32)
    uv = np.zeros((len(tris), 3, 2), dtype=np.float32)
    return tris, nrm, uv


def _open_box(x0, x1, y0, y1, z0, z1):
    """Axis-aligned box without its bottom face: 5 faces, outward normals."""
    parts = [
        _quad((x0, y1, z0), (x1, y1, z0), (x1, y1, z1), (x0, y1, z1), (0, 1, 0)),    # top
        _quad((x0, y0, z0), (x1, y0, z0), (x1, y1, z0), (x0, y1, z0), (0, 0, -1)),   # front (-z)
        _quad((x0, y0, z1), (x1, y0, z1), (x1, y1, z1), (x0, y1, z1), (0, 0, 1)),    # back (+z)
        _quad((x0, y0, z0), (x0, y0, z1), (x0, y1, z1), (x0, y1, z0), (-1, 0, 0)),   # left
        _quad((x1, y0, z0), (x1, y0, z1), (x1, y1, z1), (x1, y1, z0), (1, 0, 0)),    # right
    ]
    return _merge(parts)


# --------------------------------------------------------------------------
# Cornell-32 (configs 1, 2)
# --------------------------------------------------------------------------

def cornell_box(width=256, height=256, spp=4, bounces=8, integrator=api.kTerraIntegratorSimple, **kw) -> SceneDesc:
    """SURVEY.md section 8d 'Cornell-32': room x[-1,1] y[0,2] z[-1,1] open towards -z,
    white floor/ceiling/back, red left, green right, 0.5x0.5 ceiling light at
    y=1.99 with emissive (15,15,15), two axis-aligned white boxes of 5 faces:
    6 objects, 32 triangles. Vertex normals point into the room / out of the
    boxes (the reference does not face-forward normals, src/Terra.c:1741-1746)."""
    white = _merge([
        _quad((-1, 0, -1), (1, 0, -1), (1, 0, 1), (-1, 0, 1), (0, 1, 0)),      # floor
        _quad((-1, 2, -1), (1, 2, -1), (1, 2, 1), (-1, 2, 1), (0, -1, 0)),     # ceiling
        _quad((-1, 0, 1), (1, 0, 1), (1, 2, 1), (-1, 2, 1), (0, 0, -1)),       # back wall
    ])
    red = _merge([_quad((-1, 0, -1), (-1, 0, 1), (-1, 2, 1), (-1, 2, -1), (1, 0, 0))])
    green = _merge([_quad((1, 0, -1), (1, 0, 1), (1, 2, 1), (1, 2, -1), (-1, 0, 0))])
    light = _merge([_quad((-0.25, 1.99, -0.25), (0.25, 1.99, -0.25), (0.25, 1.99, 0.25), (-0.25, 1.99, 0.25), (0, -1, 0))])
    short_box = _open_box(0.15, 0.75, 0.0, 0.6, -0.65, -0.05)
    tall_box = _open_box(-0.75, -0.15, 0.0, 1.2, 0.05, 0.65)
    objs = [
        ObjectDesc(*white, Material(albedo=(0.73, 0.73, 0.73)), "white"),
        ObjectDesc(*red, Material(albedo=(0.65, 0.05, 0.05)), "red"),
        ObjectDesc(*green, Material(albedo=(0.12, 0.45, 0.15)), "green"),
        ObjectDesc(*light, Material(albedo=(0.78, 0.78, 0.78), emissive=(15.0, 15.0, 15.0)), "light"),
        ObjectDesc(*short_box, Material(albedo=(0.73, 0.73, 0.73)), "short_box"),
        ObjectDesc(*tall_box, Material(albedo=(0.73, 0.73, 0.73)), "tall_box"),
    ]
    d = SceneDesc(objects=objs, width=width, height=height, spp=spp, bounces=bounces, integrator=integrator, name="cornell32")
    for k, v in kw.items():
        setattr(d, k, v)
    assert d.triangle_count == 32
    return d


def cornell_textured(width=256, height=256, spp=4, bounces=8, integrator=api.kTerraIntegratorSimple, mirror=False, **kw) -> SceneDesc:
    """Cornell-32 with textured attributes: an 8x8 byte checker on the white object (point filter,
    wrap), a 4x4 float texture on the red wall (bilinear, clamp), a Phong box with a bilinear byte texture as its albedo
    (wrap; `mirror=True` switches it to the mirror mode, whose reference implementation reads one past the end
    -- src/Terra.c:385-386 -- so that variant is only compared device vs oracle), and a textured emissive on the light. Texcoords are in texel units, as the reference samples them."""
    d = cornell_box(width, height, spp, bounces, integrator)
    r = np.random.RandomState(5)
    def planar_uv(tris, ax0, ax1, scale, offset=0.0):
        uv = np.zeros((len(tris), 3, 2), np.float32)
        uv[..., 0] = (tris[..., ax0] + 1.0) * scale + offset
        uv[..., 1] = (tris[..., ax1] + 1.0) * scale + offset
        return uv
    checker = np.zeros((8, 8, 3), np.uint8)
    ii, jj = np.meshgrid(np.arange(8), np.arange(8), indexing="ij")
    checker[(ii + jj) % 2 == 0] = (230, 225, 210); checker[(ii + jj) % 2 == 1] = (60, 90, 140)
    white = d.objects[0]
    uvw = planar_uv(white.triangles, 0, 2, 4.0)
    uvw[4:6] = planar_uv(white.triangles[4:6], 0, 1, 4.0)          # back wall: x,y
    d.objects[0] = ObjectDesc(white.triangles, white.normals, uvw, Material(albedo_texture=TextureDesc(checker, 0, 0)), "white_checker")
    red = d.objects[1]
    hdr = r.uniform(0.05, 0.9, size=(4, 4, 3)).astype(np.float32)
    d.objects[1] = ObjectDesc(red.triangles, red.normals, planar_uv(red.triangles, 2, 1, 1.5, 0.25), Material(albedo_texture=TextureDesc(hdr, 1, 2)), "red_hdr_bilinear")
    light = d.objects[3]
    glow = np.array([[[15, 14, 12], [9, 12, 15]], [[15, 15, 15], [12, 9, 6]]], np.float32)
    d.objects[3] = ObjectDesc(light.triangles, light.normals, planar_uv(light.triangles, 0, 2, 4.0, -2.9), Material(albedo=(0.78, 0.78, 0.78), emissive_texture=TextureDesc(glow, 0, 2)), "light_textured")
    box = d.objects[4]
    stripes = r.randint(30, 255, size=(4, 6, 3)).astype(np.uint8)
    d.objects[4] = ObjectDesc(box.triangles, box.normals, planar_uv(box.triangles, 0, 1, 9.0), Material(kind="phong", specular_color=(0.4, 0.4, 0.4), specular_intensity=20.0, albedo_texture=TextureDesc(stripes, 1, 1 if mirror else 0)), "phong_tex")
    d.name = "cornell32_textured"
    for k, v in kw.items():
        setattr(d, k, v)
    return d


def cornell_phong(width=256, height=256, spp=4, bounces=8, integrator=api.kTerraIntegratorSimple, **kw) -> SceneDesc:
    """Cornell-32 with the two boxes switched to the Phong preset (the
    reference's only live 'specular' BSDF, src/TerraPresets.c:66-146)."""
    d = cornell_box(width, height, spp, bounces, integrator, **kw)
    d.objects[4].material = Material(kind="phong", albedo=(0.35, 0.35, 0.35), specular_color=(0.6, 0.6, 0.6), specular_intensity=40.0)
    d.objects[5].material = Material(kind="phong", albedo=(0.10, 0.20, 0.45), specular_color=(0.5, 0.5, 0.5), specular_intensity=8.0)
    d.name = "cornell32_phong"
    return d


def _uv_sphere(center, radius, n=32):
    """n x n UV sphere (n a power of two), outward vertex normals, libm-free; 2*n*n - 2*n triangles"""
    c2, s2 = _unit_circle(2 * n)                 # latitude: k*pi/n, k = 0..n
    lat_c, lat_s = c2[: n + 1].copy(), s2[: n + 1].copy()
    lat_c[n], lat_s[n] = -1.0, 0.0
    lon_c, lon_s = _unit_circle(n)
    lon_c = np.append(lon_c, lon_c[0]); lon_s = np.append(lon_s, lon_s[0])
    N = np.stack([lat_s[:, None] * lon_c[None, :], np.broadcast_to(lat_c[:, None], (n + 1, n + 1)), lat_s[:, None] * lon_s[None, :]], axis=-1)
    P = np.asarray(center, np.float64) + radius * N
    tris, nrm = _grid(P, N)
    e1 = tris[:, 1] - tris[:, 0]; e2 = tris[:, 2] - tris[:, 0]
    keep = (np.cross(e1, e2) ** 2).sum(axis=1) > 0          # drop the degenerate triangles at the poles
    return tris[keep], nrm[keep], np.zeros((int(keep.sum()), 3, 2), np.float32)


def cornell_spheres(width=1920, height=1080, spp=1024, bounces=8, integrator=api.kTerraIntegratorSimple, **kw) -> SceneDesc:
    """BASELINE.json configs[3] / SURVEY.md section 8d 'Config 4': Cornell-32 with the two boxes replaced
    in place by a glass sphere (ior 1.5) and a GGX metal sphere (alpha 0.2), ~2k triangles each.
    Both BSDFs are this repo's own definitions (no runnable reference: SURVEY.md A14)."""
    d = cornell_box(width, height, spp, bounces, integrator)
    glass = _uv_sphere((0.45, 0.45, -0.35), 0.4)
    metal = _uv_sphere((-0.45, 0.5, 0.35), 0.5)
    d.objects = d.objects[:4] + [
        ObjectDesc(*glass, Material(kind="glass", albedo=(0.98, 0.98, 0.98), ior=1.5), "glass_sphere"),
        ObjectDesc(*metal, Material(kind="ggx", specular_color=(0.95, 0.78, 0.45), roughness=0.2), "metal_sphere"),
    ]
    d.name = "cornell_spheres"
    for k, v in kw.items():
        setattr(d, k, v)
    return d


# --------------------------------------------------------------------------
# Sponza-class hall, ~100k triangles (configs 3, 5)
# --------------------------------------------------------------------------
# Generated with integer arithmetic, an integer hash and IEEE + - * / sqrt only (no
# libm), so the float32 vertex data are bit-identical on every machine.

def hash32(x):
    x = np.asarray(x, dtype=np.uint32).copy()
    x ^= x >> np.uint32(16); x *= np.uint32(0x7feb352d)
    x ^= x >> np.uint32(15); x *= np.uint32(0x846ca68b)
    x ^= x >> np.uint32(16)
    return x


def _unit_circle(n):
    """cos/sin of 2*pi*k/n for n a power of two, from half-angle square roots and one rotation recurrence"""
    c, s, m = 0.0, 1.0, 4             # angle pi/2
    while m < n:
        c, s = np.sqrt((1.0 + c) / 2.0), np.sqrt((1.0 - c) / 2.0)
        m *= 2
    cs = np.zeros(n); sn = np.zeros(n)
    cc, ss = 1.0, 0.0
    for k in range(n):
        cs[k], sn[k] = cc, ss
        cc, ss = cc * c - ss * s, cc * s + ss * c
    return cs, sn


def _grid(P, N=None, flip=False):
    """P: (ny, nx, 3) vertex grid -> triangles (2*(ny-1)*(nx-1), 3, 3) and per-vertex normals (given or face)"""
    a, b, c, d = P[:-1, :-1], P[:-1, 1:], P[1:, 1:], P[1:, :-1]
    t1 = np.stack([a, b, c], axis=-2); t2 = np.stack([a, c, d], axis=-2)
    tris = np.concatenate([t1.reshape(-1, 3, 3), t2.reshape(-1, 3, 3)]).astype(np.float32)
    if N is not None:
        na, nb, nc, nd = N[:-1, :-1], N[:-1, 1:], N[1:, 1:], N[1:, :-1]
        n1 = np.stack([na, nb, nc], axis=-2); n2 = np.stack([na, nc, nd], axis=-2)
        nrm = np.concatenate([n1.reshape(-1, 3, 3), n2.reshape(-1, 3, 3)])
    else:
        f = np.cross(tris[:, 1] - tris[:, 0], tris[:, 2] - tris[:, 0]).astype(np.float64)
        f /= np.sqrt((f * f).sum(axis=1, keepdims=True))
        nrm = np.repeat(f[:, None, :], 3, axis=1)
    if flip:
        nrm = -nrm
    return tris, nrm.astype(np.float32)


def _plane(origin, du, dv, nu, nv, normal):
    u = np.arange(nu + 1, dtype=np.float64) / nu; v = np.arange(nv + 1, dtype=np.float64) / nv
    P = np.asarray(origin, np.float64) + v[:, None, None] * np.asarray(dv, np.float64) + u[None, :, None] * np.asarray(du, np.float64)
    N = np.broadcast_to(np.asarray(normal, np.float64), P.shape)
    return _grid(P, N)


def _albedo(seed, lo=0.35, hi=0.8):
    h = hash32(np.arange(3, dtype=np.uint32) + np.uint32(seed * 977)).astype(np.float64) / 4294967296.0
    return tuple(float(np.float32(lo + (hi - lo) * x)) for x in h)


def sponza_hall(width=1920, height=1080, spp=256, bounces=8, integrator=api.kTerraIntegratorSimple, detail=1.0, **kw) -> SceneDesc:
    """SURVEY.md section 8d 'Sponza-class-100k': hall x[-10,10] y[0,8] z[-5,5], heightfield floor
    (hash noise, amplitude 0.05), tessellated walls/ceiling, 2x12 columns (32-gon prisms with flared
    capitals), arches between columns, two gallery levels, 3 ceiling quad lights; 6 objects,
    about 100,000 triangles at detail=1 (detail scales the tessellation for small test scenes)."""
    def n(v):
        return max(1, int(round(v * detail)))
    parts = {}
    # floor heightfield
    nx, nz = n(160), n(80)
    ix, iz = np.meshgrid(np.arange(nx + 1), np.arange(nz + 1))
    hval = hash32((ix * 7919 + iz * 104729).astype(np.uint32)).astype(np.float64) / 4294967296.0
    P = np.stack([-10.0 + 20.0 * ix / nx, 0.05 * hval, -5.0 + 10.0 * iz / nz], axis=-1)
    hx = np.zeros_like(hval); hz = np.zeros_like(hval)
    hx[:, 1:-1] = (P[:, 2:, 1] - P[:, :-2, 1]) / (P[:, 2:, 0] - P[:, :-2, 0])
    hz[1:-1, :] = (P[2:, :, 1] - P[:-2, :, 1]) / (P[2:, :, 2] - P[:-2, :, 2])
    N = np.stack([-hx, np.ones_like(hx), -hz], axis=-1); N /= np.sqrt((N * N).sum(axis=-1, keepdims=True))
    parts["floor"] = [_grid(P, N)]
    # shell: ceiling, two long walls, two end walls (normals into the hall)
    parts["shell"] = [
        _plane((-10, 8, -5), (20, 0, 0), (0, 0, 10), n(80), n(40), (0, -1, 0)),
        _plane((-10, 0, 5), (20, 0, 0), (0, 8, 0), n(80), n(32), (0, 0, -1)),
        _plane((-10, 0, -5), (20, 0, 0), (0, 8, 0), n(80), n(32), (0, 0, 1)),
        _plane((-10, 0, -5), (0, 0, 10), (0, 8, 0), n(40), n(32), (1, 0, 0)),
        _plane((10, 0, -5), (0, 0, 10), (0, 8, 0), n(40), n(32), (-1, 0, 0)),
    ]
    # columns: 2 rows x 12, radius 0.3, height 5, flared capital to radius 0.5 over the top 0.5
    seg = 32 if detail >= 0.5 else 8
    cs, sn = _unit_circle(seg)
    cs = np.append(cs, cs[0]); sn = np.append(sn, sn[0])
    rings = n(24)
    ys = np.concatenate([np.arange(rings + 1) * (4.5 / rings), 4.5 + np.arange(1, 5) * 0.125])
    rad = np.concatenate([np.full(rings + 1, 0.3), 0.3 + np.arange(1, 5) * 0.05])
    slope = np.concatenate([np.zeros(rings + 1), np.full(4, 0.4)])          # radial growth per unit height in the capital
    cols = []
    for row, zc in enumerate((-2.5, 2.5)):
        for k in range(12):
            xc = -8.25 + 1.5 * k
            P = np.stack([xc + rad[:, None] * cs[None, :], np.broadcast_to(ys[:, None], (len(ys), seg + 1)), zc + rad[:, None] * sn[None, :]], axis=-1)
            N = np.stack([np.broadcast_to(cs[None, :], (len(ys), seg + 1)), np.broadcast_to(-slope[:, None], (len(ys), seg + 1)), np.broadcast_to(sn[None, :], (len(ys), seg + 1))], axis=-1).copy()
            N /= np.sqrt((N * N).sum(axis=-1, keepdims=True))
            cols.append(_grid(P, N))
    parts["columns"] = cols
    # arches: half rings spanning neighbouring columns (in x), square-ish cross-section as a 6-gon tube
    a_seg = n(16)
    th_c, th_s = _unit_circle(4 * max(4, a_seg))       # quarter resolution; take the upper half circle
    half = len(th_c) // 2
    tc, ts = th_c[: half + 1], th_s[: half + 1]
    step = max(1, half // a_seg)
    tc, ts = tc[::step], ts[::step]
    c8, s8 = _unit_circle(8)
    pick6 = [0, 1, 2, 4, 5, 6, 0]                       # a closed 6-gon cross-section from the octagon's vertices
    hc, hs = c8[pick6], s8[pick6]
    arches = []
    for zc in (-2.5, 2.5):
        for k in range(11):
            xm = -8.25 + 1.5 * k + 0.75
            R, r = 0.75, 0.08
            cx = xm + (R + r * hc[None, :]) * tc[:, None]
            cy = 5.0 + (R + r * hc[None, :]) * ts[:, None]
            cz = zc + r * hs[None, :] * np.ones_like(tc)[:, None]
            P = np.stack([cx, cy, cz], axis=-1)
            N = np.stack([hc[None, :] * tc[:, None], hc[None, :] * ts[:, None], hs[None, :] * np.ones_like(tc)[:, None]], axis=-1)
            N /= np.sqrt((N * N).sum(axis=-1, keepdims=True))
            arches.append(_grid(P, N))
    parts["arches"] = arches
    # galleries: two levels on both sides: top and bottom faces + inner edge
    gal = []
    for y0 in (3.0, 5.9):
        for zs, z0, z1 in ((-1, -5.0, -3.2), (1, 3.2, 5.0)):
            gal.append(_plane((-10, y0 + 0.2, z0), (20, 0, 0), (0, 0, z1 - z0), n(40), n(4), (0, 1, 0)))
            gal.append(_plane((-10, y0, z0), (20, 0, 0), (0, 0, z1 - z0), n(40), n(4), (0, -1, 0)))
            ze = z1 if zs < 0 else z0
            gal.append(_plane((-10, y0, ze), (20, 0, 0), (0, 0.2, 0), n(40), 1, (0, 0, -float(zs))))
    parts["galleries"] = gal
    # lights: three ceiling quads
    parts["lights"] = [_plane((xc - 0.6, 7.98, -0.6), (1.2, 0, 0), (0, 0, 1.2), 1, 1, (0, -1, 0)) for xc in (-6.0, 0.0, 6.0)]

    objs = []
    for i, (name, plist) in enumerate(parts.items()):
        tris = np.concatenate([p[0] for p in plist]).astype(np.float32)
        nrm = np.concatenate([p[1] for p in plist]).astype(np.float32)
        uv = np.zeros((len(tris), 3, 2), np.float32)
        mat = Material(albedo=(0.78, 0.78, 0.78), emissive=(12.0, 11.0, 9.0)) if name == "lights" else Material(albedo=_albedo(i + 1))
        objs.append(ObjectDesc(np.ascontiguousarray(tris), np.ascontiguousarray(nrm), uv, mat, name))
    d = SceneDesc(objects=objs, width=width, height=height, spp=spp, bounces=bounces, integrator=integrator,
                  camera_position=(-9.2, 2.2, 0.35), camera_direction=(1.0, 0.06, -0.04), camera_up=(0.0, 1.0, 0.0), camera_fov=60.0,
                  name="sponza_hall" if detail == 1.0 else f"sponza_hall_d{detail}")
    for k2, v in kw.items():
        setattr(d, k2, v)
    return d


# --------------------------------------------------------------------------
# feeding a SceneDesc through the C API
# --------------------------------------------------------------------------

_keepalive = []      # TerraTexture structs must outlive the scenes that borrow them (reference src/Terra.c:294-304)


def texture_attribute(lib: api.TerraLib, td: TextureDesc, latlong: bool = False) -> api.TerraAttribute:
    data = np.ascontiguousarray(td.data)
    h, w, c = data.shape
    tex = api.TerraTexture()
    if data.dtype == np.uint8:
        lib.texture_init(C.byref(tex), w, h, c, data.ctypes.data)
    else:
        data = data.astype(np.float32)
        lib.texture_init_hdr(C.byref(tex), w, h, c, data.ctypes.data)
    tex.filter = td.filter
    tex.address_mode = td.address_mode
    a = api.TerraAttribute()
    (lib.attribute_init_cubemap if latlong else lib.attribute_init_texture)(C.byref(a), C.byref(tex))
    _keepalive.append((tex, a))
    return a


def fill_object(lib: api.TerraLib, obj: api.TerraObject, od: ObjectDesc) -> None:
    n = len(od.triangles)
    tris = np.ascontiguousarray(od.triangles, dtype=np.float32).reshape(n, 9)
    props = np.concatenate([np.asarray(od.normals, np.float32).reshape(n, 9),
                            np.asarray(od.texcoords, np.float32).reshape(n, 6)], axis=1)
    props = np.ascontiguousarray(props, dtype=np.float32)
    C.memmove(obj.triangles, tris.ctypes.data, tris.nbytes)
    C.memmove(obj.properties, props.ctypes.data, props.nbytes)
    m = od.material
    mat = obj.material
    mat.ior = m.ior
    mat.enable_bump_map_attr = False
    mat.enable_normal_map_attr = False
    mat.emissive = texture_attribute(lib, m.emissive_texture) if m.emissive_texture is not None else api.const_attribute(lib, m.emissive)
    albedo_attr = texture_attribute(lib, m.albedo_texture) if m.albedo_texture is not None else api.const_attribute(lib, m.albedo)
    if m.kind == "diffuse":
        mat.attributes[api.TERRA_DIFFUSE_ALBEDO] = albedo_attr
        mat.attributes_count = api.TERRA_DIFFUSE_END
        lib.bsdf_diffuse_init(C.byref(mat.bsdf))
    elif m.kind == "phong":
        mat.attributes[api.TERRA_PHONG_ALBEDO] = albedo_attr
        mat.attributes[api.TERRA_PHONG_SPECULAR_COLOR] = api.const_attribute(lib, m.specular_color)
        mat.attributes[api.TERRA_PHONG_SPECULAR_INTENSITY] = api.const_attribute(lib, (m.specular_intensity,) * 3)
        mat.attributes[api.TERRA_PHONG_SAMPLE_PICK] = api.const_attribute(lib, (0.0, 0.0, 0.0))
        mat.attributes_count = api.TERRA_PHONG_END
        lib.bsdf_phong_init(C.byref(mat.bsdf))
    elif m.kind == "ggx":
        mat.attributes[api.TERRA_GGX_F0] = api.const_attribute(lib, m.specular_color)
        mat.attributes[api.TERRA_GGX_ROUGHNESS] = api.const_attribute(lib, (m.roughness,) * 3)
        mat.attributes_count = api.TERRA_GGX_END
        lib.bsdf_ggx_init(C.byref(mat.bsdf))
    elif m.kind == "glass":
        mat.attributes[api.TERRA_GLASS_TINT] = api.const_attribute(lib, m.albedo)
        for slot in (api.TERRA_GLASS_UNUSED, api.TERRA_GLASS_SAMPLE_DIR, api.TERRA_GLASS_SAMPLE_PROB):
            mat.attributes[slot] = api.const_attribute(lib, (0.0, 0.0, 0.0))
        mat.attributes_count = api.TERRA_GLASS_END
        lib.bsdf_glass_init(C.byref(mat.bsdf))
    else:
        raise ValueError(f"unknown material kind {m.kind!r}")


def apply_options(lib: api.TerraLib, scene, d: SceneDesc) -> None:
    o = lib.scene_get_options(scene).contents
    o.environment_map = texture_attribute(lib, d.environment_texture, latlong=True) if d.environment_texture is not None else api.const_attribute(lib, d.environment)
    o.tonemapping_operator = d.tonemap
    o.accelerator = api.kTerraAcceleratorBVH
    o.sampling_method = d.sampling
    o.integrator = d.integrator
    o.subpixel_jitter = d.jitter
    o.samples_per_pixel = d.spp
    o.bounces = d.bounces
    o.strata = d.strata
    o.manual_exposure = d.exposure
    o.gamma = d.gamma


def build_scene(lib: api.TerraLib, d: SceneDesc, tree_mode=None, tree_builder=None, debug_shrink=None, counters=True):
    """Returns a committed HTerraScene (c_void_p value) owned by `lib`.
    tree_mode: terra_amd_set_tree_mode of the product (0 replica traversal of the reference tree, 1 fast tree, 2 automatic);
    None leaves the library's default (automatic). The reference and the oracle only have the reference traversal.
    counters: enable the product's work counters (terra_amd_set_work_counters; off by default in the library, on by default HERE because the tests and the
    tools read terra_amd_get_stats); bench.py times with counters=False."""
    scene = lib.scene_create()
    if tree_mode is not None and lib.has("terra_amd_set_tree_mode"):
        f = lib.fn("terra_amd_set_tree_mode", C.c_int, [C.c_void_p, C.c_int])
        assert f(scene, tree_mode) == 0
    if tree_builder is not None and lib.has("terra_amd_set_tree_builder"):      # 0 host binned SAH, 1 device LBVH (the fast tree only)
        assert lib.fn("terra_amd_set_tree_builder", C.c_int, [C.c_void_p, C.c_int])(scene, tree_builder) == 0
    if counters and lib.has("terra_amd_set_work_counters"):      # the product's work counters are off by default (instrumentation); tests and tools want them
        lib.fn("terra_amd_set_work_counters", C.c_int, [C.c_void_p, C.c_int])(scene, 1)
    if debug_shrink:                # the product's test hook (terra_amd.h terra_amd_debug_shrink_reference_boxes)
        assert lib.fn("terra_amd_debug_shrink_reference_boxes", C.c_int, [C.c_void_p, C.c_float])(scene, float(debug_shrink)) == 0
    if d.environment_lighting:      # the reference has no such switch: its environment term never reaches the image
        sym = {"terra_": "terra_amd_set_environment_lighting", "orc_": "orc_set_environment_lighting"}[lib.prefix]
        if not lib.has(sym):
            raise ValueError("environment lighting is an extension of libterra_amd.so / the oracle; this library has no such switch")
        lib.fn(sym, C.c_int if lib.prefix == "terra_" else None, [C.c_void_p, C.c_int])(scene, 1)
    if d.environment_sampling:      # nothing in the reference calls its TerraDistribution2D: no such switch there
        sym = {"terra_": "terra_amd_set_environment_sampling", "orc_": "orc_set_environment_sampling"}[lib.prefix]
        if not lib.has(sym):
            raise ValueError("environment sampling is an extension of libterra_amd.so / the oracle; this library has no such switch")
        lib.fn(sym, C.c_int if lib.prefix == "terra_" else None, [C.c_void_p, C.c_int])(scene, 1)
    if d.sampler_integration:       # the reference constructs the sampler and never draws from it: no such switch there
        sym = {"terra_": "terra_amd_set_sampler_integration", "orc_": "orc_set_sampler_integration"}[lib.prefix]
        if not lib.has(sym):
            raise ValueError("sampler integration is an extension of libterra_amd.so / the oracle; this library has no such switch")
        lib.fn(sym, C.c_int if lib.prefix == "terra_" else None, [C.c_void_p, C.c_int])(scene, 1)
    for od in d.objects:
        obj = lib.scene_add_object(scene, len(od.triangles)).contents
        fill_object(lib, obj, od)
    apply_options(lib, scene, d)
    import time
    t = time.perf_counter()
    lib.scene_commit(scene)
    global LAST_COMMIT_MS
    LAST_COMMIT_MS = (time.perf_counter() - t) * 1e3       # wall time of terra_scene_commit alone: host tree build(s), flattening, upload (bench.py reports it)
    return scene


def camera_of(d: SceneDesc) -> api.TerraCamera:
    cam = api.TerraCamera()
    cam.position = api.f3(d.camera_position)
    cam.direction = api.f3(d.camera_direction)
    cam.up = api.f3(d.camera_up)
    cam.fov = d.camera_fov
    return cam
