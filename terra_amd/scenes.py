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


@dataclass
class Material:
    kind: str = "diffuse"                 # "diffuse" | "phong"
    albedo: tuple = (0.73, 0.73, 0.73)
    emissive: tuple = (0.0, 0.0, 0.0)
    specular_color: tuple = (0.0, 0.0, 0.0)
    specular_intensity: float = 1.0
    ior: float = 1.5


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


def cornell_phong(width=256, height=256, spp=4, bounces=8, integrator=api.kTerraIntegratorSimple, **kw) -> SceneDesc:
    """Cornell-32 with the two boxes switched to the Phong preset (the
    reference's only live 'specular' BSDF, src/TerraPresets.c:66-146)."""
    d = cornell_box(width, height, spp, bounces, integrator, **kw)
    d.objects[4].material = Material(kind="phong", albedo=(0.35, 0.35, 0.35), specular_color=(0.6, 0.6, 0.6), specular_intensity=40.0)
    d.objects[5].material = Material(kind="phong", albedo=(0.10, 0.20, 0.45), specular_color=(0.5, 0.5, 0.5), specular_intensity=8.0)
    d.name = "cornell32_phong"
    return d


# --------------------------------------------------------------------------
# feeding a SceneDesc through the C API
# --------------------------------------------------------------------------

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
    mat.emissive = api.const_attribute(lib, m.emissive)
    if m.kind == "diffuse":
        mat.attributes[api.TERRA_DIFFUSE_ALBEDO] = api.const_attribute(lib, m.albedo)
        mat.attributes_count = api.TERRA_DIFFUSE_END
        lib.bsdf_diffuse_init(C.byref(mat.bsdf))
    elif m.kind == "phong":
        mat.attributes[api.TERRA_PHONG_ALBEDO] = api.const_attribute(lib, m.albedo)
        mat.attributes[api.TERRA_PHONG_SPECULAR_COLOR] = api.const_attribute(lib, m.specular_color)
        mat.attributes[api.TERRA_PHONG_SPECULAR_INTENSITY] = api.const_attribute(lib, (m.specular_intensity,) * 3)
        mat.attributes[api.TERRA_PHONG_SAMPLE_PICK] = api.const_attribute(lib, (0.0, 0.0, 0.0))
        mat.attributes_count = api.TERRA_PHONG_END
        lib.bsdf_phong_init(C.byref(mat.bsdf))
    else:
        raise ValueError(f"unknown material kind {m.kind!r}")


def apply_options(lib: api.TerraLib, scene, d: SceneDesc) -> None:
    o = lib.scene_get_options(scene).contents
    o.environment_map = api.const_attribute(lib, d.environment)
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


def build_scene(lib: api.TerraLib, d: SceneDesc):
    """Returns a committed HTerraScene (c_void_p value) owned by `lib`."""
    scene = lib.scene_create()
    for od in d.objects:
        obj = lib.scene_add_object(scene, len(od.triangles)).contents
        fill_object(lib, obj, od)
    apply_options(lib, scene, d)
    lib.scene_commit(scene)
    return scene


def camera_of(d: SceneDesc) -> api.TerraCamera:
    cam = api.TerraCamera()
    cam.position = api.f3(d.camera_position)
    cam.direction = api.f3(d.camera_direction)
    cam.up = api.f3(d.camera_up)
    cam.fov = d.camera_fov
    return cam
