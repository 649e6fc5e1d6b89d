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

/* Per-hit shading frame handed to the BSDF routines (188 bytes). */
typedef struct {
    TerraFloat4x4 transform;   /* columns: tangent, normal, bitangent */
    TerraFloat3   normal;
    TerraFloat3   emissive;
    float         ior;
    TerraFloat3   attributes[TERRA_MATERIAL_MAX_ATTRIBUTES];
} TerraShadingSurface;

typedef TerraFloat3 ( TerraBSDFSampleRoutine ) ( const TerraShadingSurface* surface, float e1, float e2, float e3, const TerraFloat3* wo );
typedef float       ( TerraBSDFPdfRoutine )    ( const TerraShadingSurface* surface, const TerraFloat3* wi, const TerraFloat3* wo );
typedef TerraFloat3 ( TerraBSDFEvalRoutine )   ( const TerraShadingSurface* surface, const TerraFloat3* wi, const TerraFloat3* wo );

/* A BSDF is three routines. On the device only the library's own presets
   (TerraPresets.h) are executable; they are recognised by pointer identity. */
typedef struct {
    TerraBSDFSampleRoutine* sample;
    TerraBSDFPdfRoutine*    pdf;
    TerraBSDFEvalRoutine*   eval;
} TerraBSDF;

typedef enum {
    kTerraFilterPoint,
    kTerraFilterBilinear,
    kTerraFilterTrilinear,
    kTerraFilterAnisotropic
} TerraFilter;

typedef enum {
    kTerraTextureAddressWrap,
    kTerraTextureAddressMirror,
    kTerraTextureAddressClamp
} TerraTextureAddressMode;

/* pixels == NULL marks an invalid texture */
typedef struct {
    void*    pixels;
    uint16_t width;
    uint16_t height;
    uint8_t  components;
    uint8_t  depth;          /* bytes per component: 1 (unorm8) or 4 (float) */
    uint8_t  filter;         /* TerraFilter */
    uint8_t  address_mode;   /* TerraTextureAddressMode */
} TerraTexture;

typedef void        ( *TerraAttributeFinalize ) ( void* attribute );
typedef TerraFloat3 ( *TerraAttributeEval )     ( void* attribute, const void* texcoord, const void* world_pos );

/* state == NULL: constant `value`; otherwise eval(state, uv, xyz) */
typedef struct {
    void*                  state;
    TerraAttributeFinalize finalize;
    TerraAttributeEval     eval;
    TerraFloat3            value;
} TerraAttribute;

typedef struct {
    TerraBSDF      bsdf;
    float          ior;
    TerraAttribute emissive;
    TerraAttribute attributes[TERRA_MATERIAL_MAX_ATTRIBUTES];
    size_t         attributes_count;
    bool           enable_bump_map_attr;
    bool           enable_normal_map_attr;
} TerraMaterial;

/* ---- geometry ----------------------------------------------------------- */

typedef struct TerraAABB {
    TerraFloat3 min;
    TerraFloat3 max;
} TerraAABB;

typedef struct {
    TerraFloat3 a, b, c;
} TerraTriangle;

typedef struct {
    TerraFloat3 normal_a, normal_b, normal_c;
    TerraFloat2 texcoord_a, texcoord_b, texcoord_c;
} TerraTriangleProperties;

/* Returned by terra_scene_add_object(); the caller fills triangles[],
   properties[] and material in place before terra_scene_commit(). */
typedef struct {
    TerraTriangle*           triangles;
    TerraTriangleProperties* properties;
    size_t                   triangles_count;
    TerraMaterial            material;
} TerraObject;

/* ---- options ------------------------------------------------------------ */

typedef enum {
    kTerraTonemappingOperatorNone,
    kTerraTonemappingOperatorLinear,
    kTerraTonemappingOperatorReinhard,
    kTerraTonemappingOperatorFilmic,
    kTerraTonemappingOperatorUncharted2
} TerraTonemappingOperator;

typedef enum {
    kTerraAcceleratorBVH
} TerraAccelerator;

typedef enum {
    kTerraSamplingMethodRandom,
    kTerraSamplingMethodStratified,
    kTerraSamplingMethodHalton
} TerraSamplingMethod;

typedef enum {
    kTerraIntegratorSimple,
    kTerraIntegratorDirect,
    kTerraIntegratorDirectMis,
    kTerraIntegratorDebugMono,
    kTerraIntegratorDebugDepth,
    kTerraIntegratorDebugNormals,
    kTerraIntegratorDebugMisWeights,
} TerraIntegrator;

/* Edited through terra_scene_get_options(); takes effect at the next
   terra_scene_commit(). */
typedef struct {
    TerraAttribute           environment_map;
    TerraTonemappingOperator tonemapping_operator;
    TerraAccelerator         accelerator;
    TerraSamplingMethod      sampling_method;
    TerraIntegrator          integrator;

    float  subpixel_jitter;
    size_t samples_per_pixel;
    size_t bounces;
    size_t strata;

    float  manual_exposure;
    float  gamma;
} TerraSceneOptions;

typedef struct {
    TerraFloat3 position;
    TerraFloat3 direction;
    TerraFloat3 up;
    float       fov;          /* vertical, degrees */
} TerraCamera;

/* Running sum of radiance and the number of samples in it, per pixel. */
typedef struct {
    TerraFloat3 acc;
    int         samples;
} TerraRawIntegrationResult;

/* Host framebuffer, row-major, index = y * width + x. */
typedef struct {
    TerraFloat3*               pixels;    /* tonemapped running mean */
    TerraRawIntegrationResult* results;
    size_t                     width;
    size_t                     height;
} TerraFramebuffer;

typedef struct {
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
