/*
 * Terra.h -- the scene / framebuffer / terra_render() C API, served by the
 * MI355X-native library libterra_amd.so.
 *
 * Drop-in boundary, part 2 of 3. Written from scratch; every type reproduces
 * the x86-64 SysV layout of its counterpart in the reference header
 * (reference include/Terra.h:36-198; sizes and offsets are pinned by the
 * TERRA_ABI_ASSERT block at the end of this file), and every entry point below
 * has the name, argument meaning and ownership rules of reference
 * include/Terra.h:205-245, so a client of the reference re-links against this
 * library without source changes.
 *
 * What differs, on purpose (documented in DESIGN.md):
 *   - terra_render() runs the tile loop, bounce integrator, BVH traversal,
 *     BSDF presets and RNG as HIP kernels on gfx950. There is no CPU fallback:
 *     when no device / kernel image is available, or a material uses function
 *     pointers the device cannot map (see terra_amd.h), the call renders
 *     nothing and records an error retrievable with terra_amd_last_error().
 *   - Randomness is two PCG32 streams per pixel keyed by
 *     (frame seed, pixel index, samples already accumulated) instead of the
 *     reference's wall-clock seed + process-global rand()
 *     (reference src/Terra.c:529-530, :115). See terra_amd.h.
 */
#ifndef TERRA_AMD_TERRA_H
#define TERRA_AMD_TERRA_H

#include <stdlib.h>
#include "TerraMath.h"

#ifdef __cplusplus
extern "C" {
#endif

#define terra_bsdf_importance_sample 0

#ifndef TERRA_MATERIAL_MAX_ATTRIBUTES
#define TERRA_MATERIAL_MAX_ATTRIBUTES 8
#endif
#ifndef TERRA_MATERIAL_MAX_LAYERS
#define TERRA_MATERIAL_MAX_LAYERS 4
#endif
#ifndef TERRA_MATERIAL_CONTEXT_SIZE
#define TERRA_MATERIAL_CONTEXT_SIZE 128
#endif

/* ---- shading ------------------------------------------------------------ */
/* Byte offsets (x86-64 SysV) are written next to the fields; the TERRA_ABI_ASSERT block at the end of the file
   checks them. Enumerators carry their values explicitly: they travel through the ABI as plain ints. */

/* Per-hit shading frame handed to the BSDF routines. 188 bytes. */
typedef struct TerraShadingSurface_s {
    TerraFloat4x4 transform;                                    /* +0    columns: tangent, normal, bitangent */
    TerraFloat3   normal;                                       /* +64   interpolated, normalised, not face-forwarded */
    TerraFloat3   emissive;                                     /* +76   */
    float         ior;                                          /* +88   */
    TerraFloat3   attributes[TERRA_MATERIAL_MAX_ATTRIBUTES];    /* +92   evaluated material attributes; presets also use slots as scratch */
} TerraShadingSurface;

/* sample(surface, e1, e2, e3, wo) -> wi;  pdf(surface, wi, wo);  eval(surface, wi, wo) -> f */
typedef TerraFloat3 ( TerraBSDFSampleRoutine ) ( const TerraShadingSurface* surface, float e1, float e2, float e3,
                                                 const TerraFloat3* wo );
typedef float       ( TerraBSDFPdfRoutine ) ( const TerraShadingSurface* surface, const TerraFloat3* wi,
                                              const TerraFloat3* wo );
typedef TerraFloat3 ( TerraBSDFEvalRoutine ) ( const TerraShadingSurface* surface, const TerraFloat3* wi,
                                               const TerraFloat3* wo );

/* A BSDF is three routines (24 bytes). On the device only the library's own presets
   (TerraPresets.h) are executable; they are recognised by pointer identity. */
typedef struct TerraBSDF_s {
    TerraBSDFSampleRoutine* sample;     /* +0  */
    TerraBSDFPdfRoutine*    pdf;        /* +8  */
    TerraBSDFEvalRoutine*   eval;       /* +16 */
} TerraBSDF;

typedef enum { kTerraFilterPoint = 0, kTerraFilterBilinear = 1, kTerraFilterTrilinear = 2, kTerraFilterAnisotropic = 3 } TerraFilter;

typedef enum { kTerraTextureAddressWrap = 0, kTerraTextureAddressMirror = 1, kTerraTextureAddressClamp = 2 } TerraTextureAddressMode;

/* 16 bytes; pixels == NULL marks an invalid texture */
typedef struct TerraTexture_s {
    void*    pixels;         /* +0   width * height * components elements, owned by the texture */
    uint16_t width;          /* +8   */
    uint16_t height;         /* +10  */
    uint8_t  components;     /* +12  */
    uint8_t  depth;          /* +13  bytes per component: 1 (unorm8) or 4 (float) */
    uint8_t  filter;         /* +14  TerraFilter */
    uint8_t  address_mode;   /* +15  TerraTextureAddressMode */
} TerraTexture;

typedef void        ( *TerraAttributeFinalize ) ( void* state );
typedef TerraFloat3 ( *TerraAttributeEval ) ( void* state, const void* texcoord_or_direction, const void* world_position );

/* 40 bytes. state == NULL: the constant `value`; otherwise eval(state, uv, xyz) */
typedef struct TerraAttribute_s {
    void*                  state;       /* +0   borrowed (a TerraTexture*) */
    TerraAttributeFinalize finalize;    /* +8   */
    TerraAttributeEval     eval;        /* +16  */
    TerraFloat3            value;       /* +24  */
} TerraAttribute;

/* 408 bytes */
typedef struct TerraMaterial_s {
    TerraBSDF      bsdf;                                        /* +0   */
    float          ior;                                         /* +24  */
    TerraAttribute emissive;                                    /* +32  */
    TerraAttribute attributes[TERRA_MATERIAL_MAX_ATTRIBUTES];   /* +72  slots named by the preset (TerraPresets.h) */
    size_t         attributes_count;                            /* +392 */
    bool           enable_bump_map_attr;                        /* +400 unused by the renderer */
    bool           enable_normal_map_attr;                      /* +401 unused by the renderer */
} TerraMaterial;

/* ---- geometry ----------------------------------------------------------- */

typedef struct TerraAABB {
    TerraFloat3 min;    /* +0  */
    TerraFloat3 max;    /* +12 */
} TerraAABB;

typedef struct TerraTriangle_s { TerraFloat3 a, b, c; } TerraTriangle;                          /* 36 bytes */

typedef struct TerraTriangleProperties_s {                                                      /* 60 bytes */
    TerraFloat3 normal_a, normal_b, normal_c;           /* +0  per-vertex normals */
    TerraFloat2 texcoord_a, texcoord_b, texcoord_c;     /* +36 per-vertex texture coordinates, in TEXELS */
} TerraTriangleProperties;

/* 432 bytes. Returned by terra_scene_add_object(); the caller fills triangles[],
   properties[] and material in place before terra_scene_commit(). */
typedef struct TerraObject_s {
    TerraTriangle*           triangles;         /* +0   triangles_count elements, owned by the scene */
    TerraTriangleProperties* properties;        /* +8   triangles_count elements, owned by the scene */
    size_t                   triangles_count;   /* +16  */
    TerraMaterial            material;          /* +24  */
} TerraObject;

/* ---- options ------------------------------------------------------------ */

typedef enum {
    kTerraTonemappingOperatorNone = 0, kTerraTonemappingOperatorLinear = 1, kTerraTonemappingOperatorReinhard = 2,
    kTerraTonemappingOperatorFilmic = 3, kTerraTonemappingOperatorUncharted2 = 4
} TerraTonemappingOperator;

typedef enum { kTerraAcceleratorBVH = 0 } TerraAccelerator;

typedef enum { kTerraSamplingMethodRandom = 0, kTerraSamplingMethodStratified = 1, kTerraSamplingMethodHalton = 2 } TerraSamplingMethod;

typedef enum {
    kTerraIntegratorSimple = 0, kTerraIntegratorDirect = 1, kTerraIntegratorDirectMis = 2, kTerraIntegratorDebugMono = 3,
    kTerraIntegratorDebugDepth = 4, kTerraIntegratorDebugNormals = 5, kTerraIntegratorDebugMisWeights = 6
} TerraIntegrator;

/* 96 bytes. Edited through terra_scene_get_options(); takes effect at the next
   terra_scene_commit(). */
typedef struct TerraSceneOptions_s {
    TerraAttribute           environment_map;           /* +0   see terra_amd_set_environment_lighting (terra_amd.h) */
    TerraTonemappingOperator tonemapping_operator;      /* +40  */
    TerraAccelerator         accelerator;               /* +44  */
    TerraSamplingMethod      sampling_method;           /* +48  only the stratified spp round-up has an effect, as in the reference */
    TerraIntegrator          integrator;                /* +52  */
    float                    subpixel_jitter;           /* +56  */
    size_t                   samples_per_pixel;         /* +64  */
    size_t                   bounces;                   /* +72  */
    size_t                   strata;                    /* +80  */
    float                    manual_exposure;           /* +88  */
    float                    gamma;                     /* +92  */
} TerraSceneOptions;

/* 40 bytes; left-handed: x right, y up, z forward */
typedef struct TerraCamera_s {
    TerraFloat3 position;     /* +0  */
    TerraFloat3 direction;    /* +12 */
    TerraFloat3 up;           /* +24 */
    float       fov;          /* +36 vertical, degrees */
} TerraCamera;

/* 16 bytes: running sum of radiance and the number of samples in it, per pixel. */
typedef struct TerraRawIntegrationResult_s {
    TerraFloat3 acc;        /* +0  */
    int         samples;    /* +12 */
} TerraRawIntegrationResult;

/* 32 bytes: host framebuffer, row-major, index = y * width + x. */
typedef struct TerraFramebuffer_s {
    TerraFloat3*               pixels;    /* +0  tonemapped running mean */
    TerraRawIntegrationResult* results;   /* +8  */
    size_t                     width;     /* +16 */
    size_t                     height;    /* +24 */
} TerraFramebuffer;

/* 4 bytes: what a leaf of the reference's tree stores (8 bits of object index: at most 256 objects) */
typedef struct TerraPrimitiveRef_s {
    uint32_t object_idx   : 8;
    uint32_t triangle_idx : 24;
} TerraPrimitiveRef;

/* ---- API ---------------------------------------------------------------- */

typedef void* HTerraScene;

HTerraScene        terra_scene_create ( void );
TerraObject*       terra_scene_add_object ( HTerraScene scene, size_t triangle_count );
size_t             terra_scene_count_objects ( HTerraScene scene );
/* Builds the BVH + light list on the host and uploads the flattened scene to
   the current device (reference src/Terra.c:162-236). */
void               terra_scene_commit ( HTerraScene scene );
void               terra_scene_clear ( HTerraScene scene );
TerraSceneOptions* terra_scene_get_options ( HTerraScene scene );
void               terra_scene_destroy ( HTerraScene scene );

bool               terra_framebuffer_create ( TerraFramebuffer* framebuffer, size_t width, size_t height );
void               terra_framebuffer_clear ( TerraFramebuffer* framebuffer );
void               terra_framebuffer_destroy ( TerraFramebuffer* framebuffer );

bool               terra_texture_init ( TerraTexture* texture, size_t width, size_t height, size_t components, const void* data );
bool               terra_texture_init_hdr ( TerraTexture* texture, size_t width, size_t height, size_t components, const float* data );
TerraFloat3        terra_texture_read ( TerraTexture* texture, size_t x, size_t y );
TerraFloat3        terra_texture_sample ( void* texture, const void* uv, const void* xyz );
TerraFloat3        terra_texture_sample_latlong ( void* texture, const void* dir, const void* xyz );
void               terra_texture_destroy ( TerraTexture* texture );
void               terra_texture_finalize ( void* texture );

void               terra_attribute_init_constant ( TerraAttribute* attr, const TerraFloat3* value );
void               terra_attribute_init_texture ( TerraAttribute* attr, TerraTexture* texture );
void               terra_attribute_init_cubemap ( TerraAttribute* attr, TerraTexture* texture );

/* Adds samples_per_pixel samples to every pixel of the tile [x,x+width) x
   [y,y+height) of `framebuffer` and rewrites the tile's tonemapped pixels.
   Re-entrant for disjoint tiles of one framebuffer (reference
   src/Terra.c:512-635; caller pattern satellite/src/Renderer.cpp:70-98). */
void               terra_render ( const TerraCamera* camera, HTerraScene scene, const TerraFramebuffer* framebuffer,
                                  size_t x, size_t y, size_t width, size_t height );

void*              terra_malloc ( size_t size );
void*              terra_realloc ( void* ptr, size_t size );
void               terra_free ( void* ptr );
void               terra_log ( const char* str, ... );

/* ---- layout pins (x86-64 SysV) ---------------------- */
#if defined(__cplusplus)
#define TERRA_ABI_ASSERT(c) static_assert ( c, #c )
#else
#define TERRA_ABI_ASSERT(c) _Static_assert ( c, #c )
#endif
TERRA_ABI_ASSERT ( sizeof ( TerraFloat3 ) == 12 );
TERRA_ABI_ASSERT ( sizeof ( TerraFloat4x4 ) == 64 );
TERRA_ABI_ASSERT ( sizeof ( TerraShadingSurface ) == 188 );
TERRA_ABI_ASSERT ( offsetof ( TerraShadingSurface, normal ) == 64 );
TERRA_ABI_ASSERT ( offsetof ( TerraShadingSurface, emissive ) == 76 );
TERRA_ABI_ASSERT ( offsetof ( TerraShadingSurface, ior ) == 88 );
TERRA_ABI_ASSERT ( offsetof ( TerraShadingSurface, attributes ) == 92 );
TERRA_ABI_ASSERT ( sizeof ( TerraBSDF ) == 24 );
TERRA_ABI_ASSERT ( sizeof ( TerraTexture ) == 16 );
TERRA_ABI_ASSERT ( sizeof ( TerraAttribute ) == 40 );
TERRA_ABI_ASSERT ( offsetof ( TerraAttribute, value ) == 24 );
TERRA_ABI_ASSERT ( sizeof ( TerraMaterial ) == 408 );
TERRA_ABI_ASSERT ( offsetof ( TerraMaterial, ior ) == 24 );
TERRA_ABI_ASSERT ( offsetof ( TerraMaterial, emissive ) == 32 );
TERRA_ABI_ASSERT ( offsetof ( TerraMaterial, attributes ) == 72 );
TERRA_ABI_ASSERT ( offsetof ( TerraMaterial, attributes_count ) == 392 );
TERRA_ABI_ASSERT ( offsetof ( TerraMaterial, enable_bump_map_attr ) == 400 );
TERRA_ABI_ASSERT ( sizeof ( TerraAABB ) == 24 );
TERRA_ABI_ASSERT ( sizeof ( TerraTriangle ) == 36 );
TERRA_ABI_ASSERT ( sizeof ( TerraTriangleProperties ) == 60 );
TERRA_ABI_ASSERT ( sizeof ( TerraObject ) == 432 );
TERRA_ABI_ASSERT ( offsetof ( TerraObject, material ) == 24 );
TERRA_ABI_ASSERT ( sizeof ( TerraSceneOptions ) == 96 );
TERRA_ABI_ASSERT ( offsetof ( TerraSceneOptions, tonemapping_operator ) == 40 );
TERRA_ABI_ASSERT ( offsetof ( TerraSceneOptions, integrator ) == 52 );
TERRA_ABI_ASSERT ( offsetof ( TerraSceneOptions, subpixel_jitter ) == 56 );
TERRA_ABI_ASSERT ( offsetof ( TerraSceneOptions, samples_per_pixel ) == 64 );
TERRA_ABI_ASSERT ( offsetof ( TerraSceneOptions, bounces ) == 72 );
TERRA_ABI_ASSERT ( offsetof ( TerraSceneOptions, strata ) == 80 );
TERRA_ABI_ASSERT ( offsetof ( TerraSceneOptions, manual_exposure ) == 88 );
TERRA_ABI_ASSERT ( offsetof ( TerraSceneOptions, gamma ) == 92 );
TERRA_ABI_ASSERT ( sizeof ( TerraCamera ) == 40 );
TERRA_ABI_ASSERT ( sizeof ( TerraRawIntegrationResult ) == 16 );
TERRA_ABI_ASSERT ( sizeof ( TerraFramebuffer ) == 32 );
TERRA_ABI_ASSERT ( sizeof ( TerraPrimitiveRef ) == 4 );

#ifdef __cplusplus
}
#endif
#endif /* TERRA_AMD_TERRA_H */
