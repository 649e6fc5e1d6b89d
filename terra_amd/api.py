"""ctypes mirror of the Terra C API (include/Terra.h, include/TerraPresets.h).

Host-side mirror of the reference's client interface for the hot path: the same
type names, field names and entry points as reference include/Terra.h:36-245, so
tests read like a C client of the reference would. One `TerraLib` instance binds
ONE shared library that exports that API:

* the product, ``terra_amd/libterra_amd.so`` (prefix ``terra_``), or
* in tests only, the compiled reference ``oracle/_ref/libterra_ref.so``
  (prefix ``terra_``) and the CPU restatement ``oracle/liboracle.so``
  (prefix ``orc_``).

Nothing here computes anything: it is plumbing over plain pointers and sizes.
"""
from __future__ import annotations

import ctypes as C
from ctypes import POINTER, Structure, c_bool, c_float, c_int, c_size_t, c_uint8, c_uint16, c_uint32, c_uint64, c_void_p

import numpy as np

MAX_ATTRIBUTES = 8

# enum values, reference include/Terra.h:57-69,131-157
kTerraTonemappingOperatorNone, kTerraTonemappingOperatorLinear, kTerraTonemappingOperatorReinhard, \
    kTerraTonemappingOperatorFilmic, kTerraTonemappingOperatorUncharted2 = range(5)
kTerraAcceleratorBVH = 0
kTerraSamplingMethodRandom, kTerraSamplingMethodStratified, kTerraSamplingMethodHalton = range(3)
kTerraIntegratorSimple, kTerraIntegratorDirect, kTerraIntegratorDirectMis, kTerraIntegratorDebugMono, \
    kTerraIntegratorDebugDepth, kTerraIntegratorDebugNormals, kTerraIntegratorDebugMisWeights = range(7)

# preset attribute slots, reference include/TerraPresets.h:11-26
TERRA_DIFFUSE_ALBEDO, TERRA_DIFFUSE_END = 0, 1
TERRA_PHONG_SPECULAR_COLOR, TERRA_PHONG_ALBEDO, TERRA_PHONG_SPECULAR_INTENSITY, TERRA_PHONG_SAMPLE_PICK, TERRA_PHONG_END = 0, 1, 2, 3, 4
TERRA_GGX_F0, TERRA_GGX_ROUGHNESS, TERRA_GGX_END = 0, 1, 2
TERRA_GLASS_TINT, TERRA_GLASS_UNUSED, TERRA_GLASS_SAMPLE_DIR, TERRA_GLASS_SAMPLE_PROB, TERRA_GLASS_END = 0, 1, 2, 3, 4


class TerraFloat2(Structure):
    _fields_ = [("x", c_float), ("y", c_float)]


class TerraFloat3(Structure):
    _fields_ = [("x", c_float), ("y", c_float), ("z", c_float)]

    def __init__(self, x=0.0, y=0.0, z=0.0):
        super().__init__(x, y, z)

    def tuple(self):
        return (self.x, self.y, self.z)


class TerraFloat4(Structure):
    _fields_ = [("x", c_float), ("y", c_float), ("z", c_float), ("w", c_float)]


class TerraFloat4x4(Structure):
    _fields_ = [("rows", TerraFloat4 * 4)]


class TerraShadingSurface(Structure):
    _fields_ = [("transform", TerraFloat4x4), ("normal", TerraFloat3), ("emissive", TerraFloat3),
                ("ior", c_float), ("attributes", TerraFloat3 * MAX_ATTRIBUTES)]


class TerraBSDF(Structure):
    _fields_ = [("sample", c_void_p), ("pdf", c_void_p), ("eval", c_void_p)]


class TerraTexture(Structure):
    _fields_ = [("pixels", c_void_p), ("width", c_uint16), ("height", c_uint16), ("components", c_uint8),
                ("depth", c_uint8), ("filter", c_uint8), ("address_mode", c_uint8)]


class TerraAttribute(Structure):
    _fields_ = [("state", c_void_p), ("finalize", c_void_p), ("eval", c_void_p), ("value", TerraFloat3)]


class TerraMaterial(Structure):
    _fields_ = [("bsdf", TerraBSDF), ("ior", c_float), ("emissive", TerraAttribute),
                ("attributes", TerraAttribute * MAX_ATTRIBUTES), ("attributes_count", c_size_t),
                ("enable_bump_map_attr", c_bool), ("enable_normal_map_attr", c_bool)]


class TerraAABB(Structure):
    _fields_ = [("min", TerraFloat3), ("max", TerraFloat3)]


class TerraTriangle(Structure):
    _fields_ = [("a", TerraFloat3), ("b", TerraFloat3), ("c", TerraFloat3)]


class TerraTriangleProperties(Structure):
    _fields_ = [("normal_a", TerraFloat3), ("normal_b", TerraFloat3), ("normal_c", TerraFloat3),
                ("texcoord_a", TerraFloat2), ("texcoord_b", TerraFloat2), ("texcoord_c", TerraFloat2)]


class TerraObject(Structure):
    _fields_ = [("triangles", POINTER(TerraTriangle)), ("properties", POINTER(TerraTriangleProperties)),
                ("triangles_count", c_size_t), ("material", TerraMaterial)]


class TerraSceneOptions(Structure):
    _fields_ = [("environment_map", TerraAttribute), ("tonemapping_operator", c_int), ("accelerator", c_int),
                ("sampling_method", c_int), ("integrator", c_int), ("subpixel_jitter", c_float),
                ("samples_per_pixel", c_size_t), ("bounces", c_size_t), ("strata", c_size_t),
                ("manual_exposure", c_float), ("gamma", c_float)]


class TerraCamera(Structure):
    _fields_ = [("position", TerraFloat3), ("direction", TerraFloat3), ("up", TerraFloat3), ("fov", c_float)]


class TerraRawIntegrationResult(Structure):
    _fields_ = [("acc", TerraFloat3), ("samples", c_int)]


class TerraFramebuffer(Structure):
    _fields_ = [("pixels", POINTER(TerraFloat3)), ("results", POINTER(TerraRawIntegrationResult)),
                ("width", c_size_t), ("height", c_size_t)]


# sizes pinned by include/Terra.h's TERRA_ABI_ASSERT block
ABI_SIZES = {
    TerraFloat3: 12, TerraFloat4x4: 64, TerraShadingSurface: 188, TerraBSDF: 24, TerraTexture: 16,
    TerraAttribute: 40, TerraMaterial: 408, TerraAABB: 24, TerraTriangle: 36, TerraTriangleProperties: 60,
    TerraObject: 432, TerraSceneOptions: 96, TerraCamera: 40, TerraRawIntegrationResult: 16, TerraFramebuffer: 32,
}
for _t, _s in ABI_SIZES.items():
    assert C.sizeof(_t) == _s, (_t.__name__, C.sizeof(_t), _s)

# numpy views of the array element types
TRIANGLE_DTYPE = np.dtype((np.float32, (3, 3)))          # a, b, c
PROPERTIES_DTYPE = np.dtype((np.float32, (15,)))          # na nb nc (9) + ta tb tc (6)
RESULT_DTYPE = np.dtype([("acc", np.float32, (3,)), ("samples", np.int32)])
assert RESULT_DTYPE.itemsize == 16

# entry points of include/Terra.h + include/TerraPresets.h, name -> (restype, argtypes)
API_SIGNATURES = {
    "scene_create": (c_void_p, []),
    "scene_add_object": (POINTER(TerraObject), [c_void_p, c_size_t]),
    "scene_count_objects": (c_size_t, [c_void_p]),
    "scene_commit": (None, [c_void_p]),
    "scene_clear": (None, [c_void_p]),
    "scene_get_options": (POINTER(TerraSceneOptions), [c_void_p]),
    "scene_destroy": (None, [c_void_p]),
    "framebuffer_create": (c_bool, [POINTER(TerraFramebuffer), c_size_t, c_size_t]),
    "framebuffer_clear": (None, [POINTER(TerraFramebuffer)]),
    "framebuffer_destroy": (None, [POINTER(TerraFramebuffer)]),
    "texture_init": (c_bool, [POINTER(TerraTexture), c_size_t, c_size_t, c_size_t, c_void_p]),
    "texture_init_hdr": (c_bool, [POINTER(TerraTexture), c_size_t, c_size_t, c_size_t, c_void_p]),
    "texture_read": (TerraFloat3, [POINTER(TerraTexture), c_size_t, c_size_t]),
    "texture_sample": (TerraFloat3, [c_void_p, c_void_p, c_void_p]),
    "texture_sample_latlong": (TerraFloat3, [c_void_p, c_void_p, c_void_p]),
    "texture_destroy": (None, [POINTER(TerraTexture)]),
    "texture_finalize": (None, [c_void_p]),
    "attribute_init_constant": (None, [POINTER(TerraAttribute), POINTER(TerraFloat3)]),
    "attribute_init_texture": (None, [POINTER(TerraAttribute), POINTER(TerraTexture)]),
    "attribute_init_cubemap": (None, [POINTER(TerraAttribute), POINTER(TerraTexture)]),
    "render": (None, [POINTER(TerraCamera), c_void_p, POINTER(TerraFramebuffer), c_size_t, c_size_t, c_size_t, c_size_t]),
    "malloc": (c_void_p, [c_size_t]),
    "realloc": (c_void_p, [c_void_p, c_size_t]),
    "free": (None, [c_void_p]),
    "log": (None, None),  # variadic
    "bsdf_diffuse_init": (None, [POINTER(TerraBSDF)]),
    "bsdf_phong_init": (None, [POINTER(TerraBSDF)]),
    "bsdf_ggx_init": (None, [POINTER(TerraBSDF)]),
    "bsdf_glass_init": (None, [POINTER(TerraBSDF)]),
}
# not part of the reference API: absent from the compiled reference
OPTIONAL_SYMBOLS = {"bsdf_ggx_init", "bsdf_glass_init"}


class TerraLib:
    """Binds one shared library exporting the Terra.h API under `prefix`."""

    def __init__(self, path: str, prefix: str = "terra_"):
        self.path = str(path)
        self.prefix = prefix
        self.dll = C.CDLL(self.path, mode=getattr(C, "RTLD_LOCAL", 0))
        self.missing = []
        for name, (res, args) in API_SIGNATURES.items():
            sym = prefix + name
            try:
                fn = getattr(self.dll, sym)
            except AttributeError:
                if name not in OPTIONAL_SYMBOLS:
                    self.missing.append(sym)
                continue
            fn.restype = res
            if args is not None:
                fn.argtypes = args
            setattr(self, name, fn)

    def fn(self, symbol: str, restype, argtypes):
        """Bind an extra (non-Terra.h) symbol, e.g. the terra_amd_* or ref_* ones."""
        f = getattr(self.dll, symbol)
        f.restype = restype
        f.argtypes = argtypes
        return f

    def has(self, symbol: str) -> bool:
        try:
            getattr(self.dll, symbol)
            return True
        except AttributeError:
            return False


def f3(v) -> TerraFloat3:
    return TerraFloat3(float(v[0]), float(v[1]), float(v[2]))


def const_attribute(lib: TerraLib, value) -> TerraAttribute:
    a = TerraAttribute()
    v = f3(value)
    lib.attribute_init_constant(C.byref(a), C.byref(v))
    return a


class Framebuffer:
    """Owns a TerraFramebuffer created by `lib` and exposes numpy views of it."""

    def __init__(self, lib: TerraLib, width: int, height: int):
        self.lib = lib
        self.fb = TerraFramebuffer()
        if not lib.framebuffer_create(C.byref(self.fb), width, height):
            raise ValueError("terra_framebuffer_create failed")
        self.width, self.height = width, height

    @property
    def pixels(self) -> np.ndarray:
        n = self.width * self.height
        buf = (c_float * (3 * n)).from_address(C.addressof(self.fb.pixels.contents))
        return np.frombuffer(buf, dtype=np.float32).reshape(self.height, self.width, 3)

    @property
    def results(self) -> np.ndarray:
        n = self.width * self.height
        buf = (C.c_char * (16 * n)).from_address(C.addressof(self.fb.results.contents))
        return np.frombuffer(buf, dtype=RESULT_DTYPE).reshape(self.height, self.width)

    def clear(self):
        self.lib.framebuffer_clear(C.byref(self.fb))

    def destroy(self):
        if self.fb.pixels:
            self.lib.framebuffer_destroy(C.byref(self.fb))
            self.fb.pixels = None
            self.fb.results = None

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass
