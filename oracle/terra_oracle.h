/*
 * TEST INFRASTRUCTURE (oracle/). Not part of the product.
 *
 * terra_oracle: a plain-C CPU restatement of the reference's hot path
 * terra_render -> terra_trace -> terra_scene_raycast -> terra_bvh_traverse
 * (+ watertight / Moeller-Trumbore tests, BVH build, surface init, diffuse and
 * Phong presets, all seven integrators, tonemapping), behind the same API
 * shape as include/Terra.h with the prefix orc_ instead of terra_.
 *
 * Who may use it: tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * leg -- as the checker / reported baseline only. The product
 * (terra_amd/libterra_amd.so) never links, loads or calls it.
 *
 * Pinning: tests/test_oracle_vs_reference.py compares it bit-for-bit with the
 * compiled reference (oracle/_ref) where /root/reference exists, and
 * tests/test_oracle_golden.py against the committed fixtures in tests/golden/
 * that were generated from the compiled reference (tests/golden/generate.py).
 */
#ifndef TERRA_ORACLE_H
#define TERRA_ORACLE_H

#include "Terra.h"
#include "TerraPresets.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- the Terra.h surface, prefix orc_ -------------------------------------- */
HTerraScene        orc_scene_create ( void );
TerraObject*       orc_scene_add_object ( HTerraScene scene, size_t triangle_count );
size_t             orc_scene_count_objects ( HTerraScene scene );
void               orc_scene_commit ( HTerraScene scene );
void               orc_scene_clear ( HTerraScene scene );
TerraSceneOptions* orc_scene_get_options ( HTerraScene scene );
void               orc_scene_destroy ( HTerraScene scene );
bool               orc_framebuffer_create ( TerraFramebuffer* fb, size_t width, size_t height );
void               orc_framebuffer_clear ( TerraFramebuffer* fb );
void               orc_framebuffer_destroy ( TerraFramebuffer* fb );
bool               orc_texture_init ( TerraTexture* texture, size_t width, size_t height, size_t components, const void* data );
bool               orc_texture_init_hdr ( TerraTexture* texture, size_t width, size_t height, size_t components, const float* data );
TerraFloat3        orc_texture_read ( TerraTexture* texture, size_t x, size_t y );
TerraFloat3        orc_texture_sample ( void* texture, const void* uv, const void* xyz );
TerraFloat3        orc_texture_sample_latlong ( void* texture, const void* dir, const void* xyz );
void               orc_texture_destroy ( TerraTexture* texture );
void               orc_texture_finalize ( void* texture );
void               orc_attribute_init_constant ( TerraAttribute* attr, const TerraFloat3* value );
void               orc_attribute_init_texture ( TerraAttribute* attr, TerraTexture* texture );
void               orc_attribute_init_cubemap ( TerraAttribute* attr, TerraTexture* texture );
/* per-pixel streams keyed by (scene frame seed, pixel, samples so far); see stream_key.h */
void               orc_render ( const TerraCamera* camera, HTerraScene scene, const TerraFramebuffer* fb, size_t x, size_t y, size_t width, size_t height );
void*              orc_malloc ( size_t size );
void*              orc_realloc ( void* ptr, size_t size );
void               orc_free ( void* ptr );
void               orc_log ( const char* str, ... );
void               orc_bsdf_diffuse_init ( TerraBSDF* bsdf );
void               orc_bsdf_phong_init ( TerraBSDF* bsdf );
void               orc_bsdf_ggx_init ( TerraBSDF* bsdf );      /* no live reference: defined by this repo, see TerraPresets.h */
void               orc_bsdf_glass_init ( TerraBSDF* bsdf );

/* ---- checker controls -------------------------------------------------------- */
enum { ORC_MATH_LIBM = 0, ORC_MATH_DEVMATH = 1 };
/* ORC_MATH_LIBM: sinf/cosf/powf/acosf from this host's libm (bit-identical to
   the compiled reference on the same host). ORC_MATH_DEVMATH: the portable
   restatement in oracle_devmath.h (bit-identical to the device). Process-global. */
void     orc_set_math_mode ( int mode );
int      orc_get_math_mode ( void );
void     orc_set_frame_seed ( HTerraScene scene, uint64_t seed );
/* extension mirrored from terra_amd_set_environment_lighting(): a ray that leaves the scene adds
   throughput * environment (the line the reference has commented out, src/Terra.c:1056). Off by default. */
void     orc_set_environment_lighting ( HTerraScene scene, int on );
/* Extension, off by default, UNPINNED (the reference never draws from the pixel sampler it constructs, src/Terra.c:535-548): with on = 1 and the Halton or
   stratified sampling method, camera sample n of a pixel takes element n of the pixel's sampler and uses it as the first two variates of the BSDF sample at
   bounce 0. The product's switch of the same name (terra_amd_set_sampler_integration) does the same, bit for bit. */
void     orc_set_sampler_integration ( HTerraScene scene, int on );
/* Extension, off by default, UNPINNED (nothing in the reference calls TerraDistribution2D, src/Terra.c:812-846): with environment lighting on and a lat-long
   environment texture, the Direct and Direct+MIS integrators take one environment sample per shaded hit through a TerraDistribution2D over the map's
   luminance x sin(theta) (two more draws of stream B, a shadow ray), and a path ray that leaves the scene after bounce 0 no longer adds the environment.
   Takes effect at the next orc_scene_commit (or at once on a committed scene). The product's terra_amd_set_environment_sampling does the same, bit for bit. */
void     orc_set_environment_sampling ( HTerraScene scene, int on );

/* work counters accumulated by every raycast since the last reset (thread-safe sums) */
typedef struct {
    uint64_t rays;          /* terra_scene_raycast-equivalents */
    uint64_t nodes;         /* BVH nodes popped */
    uint64_t box_tests;     /* slab tests */
    uint64_t tri_tests;     /* watertight queries */
    uint64_t hits;          /* rays that hit (surface init executed) */
    uint64_t samples;       /* camera samples traced */
    uint64_t rand_calls;    /* draws from stream B */
    uint64_t attr_fetches;  /* attributes_count + 1 summed over hits */
} OrcCounters;
void     orc_counters_reset ( void );
void     orc_counters_get ( OrcCounters* out );

/* rendering variants: explicit seed, optional per-pixel rand-call counts (indexed like the framebuffer) */
void     orc_render_pixels ( const TerraCamera* camera, HTerraScene scene, const TerraFramebuffer* fb,
                             size_t x, size_t y, size_t w, size_t h, uint64_t frame_seed, uint32_t* rand_calls );
void     orc_render_pixels_mt ( const TerraCamera* camera, HTerraScene scene, const TerraFramebuffer* fb,
                                size_t x, size_t y, size_t w, size_t h, uint64_t frame_seed, uint32_t* rand_calls, int nthreads );

/* ---- unit-level entry points (same shapes as oracle/ref_wrapper.c's ref_*) -- */
void        orc_pcg_floats ( uint32_t seed, int n, float* out );
/* SURVEY.md 8f N4, unit level (reference src/Terra.c:703-755, 760-846) */
void        orc_stratified_pairs ( uint32_t seed, int strata, int samples, int n, float* out2 );
void        orc_halton_pairs ( int first, int n, float* out2 );
float       orc_radical_inverse ( uint64_t base, uint64_t a );
void        orc_distribution_1d ( const float* f, size_t n, const float* e, int m, float* x, float* pdf, uint32_t* idx, float* cdf_out, float* integral_out );
void        orc_distribution_2d ( const float* f, size_t nx, size_t ny, const float* e12, int m, float* xy2, float* pdf, float* marginal_cdf_out );
int         orc_ray_aabb ( const TerraFloat3* origin, const TerraFloat3* dir, const TerraAABB* box, float* tmin, float* tmax );
int         orc_watertight ( const TerraFloat3* origin, const TerraFloat3* dir, const TerraTriangle* tri, float* out8 );
int         orc_moller_trumbore ( const TerraFloat3* origin, const TerraFloat3* dir, const TerraTriangle* tri, float* out4 );
int         orc_bvh_node_count ( HTerraScene scene );
const void* orc_bvh_nodes ( HTerraScene scene );
int         orc_bvh_max_stack ( HTerraScene scene );
int         orc_bvh_traverse ( HTerraScene scene, const TerraFloat3* origin, const TerraFloat3* dir, TerraFloat3* point, uint32_t* prim );
int         orc_raycast ( HTerraScene scene, const TerraFloat3* origin, const TerraFloat3* dir, TerraShadingSurface* surface, TerraFloat3* point, int* triangle );
TerraFloat3 orc_trace_one ( HTerraScene scene, const TerraFloat3* origin, const TerraFloat3* dir, uint64_t stateB, uint64_t incB, uint32_t* rand_calls );
TerraFloat3 orc_camera_sample ( const TerraCamera* camera, size_t fb_width, size_t fb_height, size_t x, size_t y, float jitter, float r1, float r2 );
TerraFloat4x4 orc_camera_frame ( const TerraCamera* camera );
void        orc_surface_init ( TerraShadingSurface* surface, const TerraTriangle* tri, const TerraMaterial* material, const TerraTriangleProperties* props, const TerraFloat3* point );
TerraFloat3 orc_tonemap ( const TerraFloat3* color, int op, float gamma );
size_t      orc_lights_count ( HTerraScene scene );
size_t      orc_lights_triangles_count ( HTerraScene scene );
int         orc_light_object_index ( HTerraScene scene, size_t i );
float       orc_light_area ( HTerraScene scene, size_t i );
const float* orc_light_triangle_areas ( HTerraScene scene, size_t i );
float       orc_math_sinf ( float x );
float       orc_math_cosf ( float x );
float       orc_math_powf ( float x, float y );
float       orc_math_acosf ( float x );
void        orc_devmath_sincos_domain_check ( uint64_t* sin_mismatch, uint64_t* cos_mismatch );
void        orc_math_eval ( int fn, int mode, int n, const float* x, const float* y, float* out );
void        orc_pixel_stream_key ( uint64_t frame_seed, uint64_t pix, uint64_t samples_so_far, uint64_t* out3 );

#ifdef __cplusplus
}
#endif
#endif
