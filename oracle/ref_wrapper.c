/*
 * TEST INFRASTRUCTURE (oracle/). Not part of the product.
 *
 * Wrapper translation unit that compiles the UNMODIFIED reference renderer
 * (/root/reference/src/Terra.c, passed in as REF_TERRA_C by oracle/Makefile)
 * with its three entropy sources pinned, per SURVEY.md section 8c:
 *
 *   time(NULL) ^ &exit   (reference src/Terra.c:679)  ->  ref_seed, because the
 *                         macro below pre-xors &exit so it cancels;
 *   rand()               (reference src/Terra.c:115)  ->  ref_rand(), a PCG32
 *                         stream returning 24 random bits scaled so that
 *                         (float)r / RAND_MAX == u24 * 2^-24 < 1.
 *
 * No reference source is copied: the file is #included from where it lies and
 * only exists on the build container. The output (oracle/_ref/libterra_ref.so)
 * is git-ignored. Everything after the #include is harness code owned by this
 * repo: per-pixel rendering with order-independent streams, and accessors for
 * private state that the golden generator dumps.
 */
#include <stdlib.h>
#include <stdint.h>
#include <time.h>
#include <pthread.h>
#include "stream_key.h"

static __thread uint64_t  ref_seed;
static __thread OrcPcg32  ref_stream;
static __thread uint64_t  ref_rand_count;

static int ref_rand ( void ) {
    ++ref_rand_count;
    return orc_rand_from_stream ( &ref_stream );
}

#define time(x) ( ( time_t ) ( ref_seed ^ ( uint64_t ) &exit ) )
#define rand    ref_rand

#include REF_TERRA_C

#undef time
#undef rand

/* ---- entropy control ------------------------------------------------------ */

void ref_set_streams ( uint64_t frame_seed, uint64_t pix, uint64_t samples_so_far ) {
    OrcPixelStreams s = orc_pixel_streams ( frame_seed, pix, samples_so_far );
    ref_seed = s.seedA;
    ref_stream = s.streamB;
}

void ref_set_raw_streams ( uint32_t seedA, uint64_t stateB, uint64_t incB ) {
    ref_seed = seedA;
    ref_stream.state = stateB;
    ref_stream.inc = incB;
}

uint64_t ref_rand_calls ( void ) { return ref_rand_count; }
void     ref_rand_calls_reset ( void ) { ref_rand_count = 0; }

/* first n outputs of the reference's camera PCG for a 32-bit seed */
void ref_pcg_floats ( uint32_t seed, int n, float* out ) {
    ref_seed = seed;
    TerraSamplerRandom s;
    terra_sampler_random_init ( &s );
    for ( int i = 0; i < n; ++i ) {
        out[i] = terra_sampler_random_next ( &s );
    }
}

/* ---- SURVEY.md 8f N4, unit level: the reference's stratified / Halton samplers and 1D / 2D distributions
   (src/Terra.c:703-755, 760-846), driven through their own entry points. Nothing in the reference calls them
   on the render path (the hemisphere sampler is constructed and ignored, src/Terra.c:535-548). */
void ref_stratified_pairs ( uint32_t seed, int strata, int samples, int n, float* out2 ) {
    ref_seed = seed;
    TerraSamplerRandom r;
    terra_sampler_random_init ( &r );
    TerraSamplerStratified s;
    terra_sampler_stratified_init ( &s, &r, strata, samples );
    for ( int i = 0; i < n; ++i ) terra_sampler_stratified_next_pair ( &s, out2 + 2 * i, out2 + 2 * i + 1 );
}
void ref_halton_pairs ( int first, int n, float* out2 ) {
    TerraSamplerHalton h;
    terra_sampler_halton_init ( &h );
    h.next = first;
    for ( int i = 0; i < n; ++i ) terra_sampler_halton_next_pair ( &h, out2 + 2 * i, out2 + 2 * i + 1 );
}
float ref_radical_inverse ( uint64_t base, uint64_t a ) { return terra_radical_inverse ( base, a ); }
/* builds the distribution over f[n], returns its cdf / integral, and samples it at e[m] (every e must be below the last cdf
   entry: the reference asserts otherwise) */
void ref_distribution_1d ( const float* f, size_t n, const float* e, int m, float* x, float* pdf, uint32_t* idx, float* cdf_out, float* integral_out ) {
    TerraDistribution1D d;
    terra_distribution_1d_init ( &d, f, n );
    if ( cdf_out ) memcpy ( cdf_out, d.cdf, n * sizeof ( float ) );
    if ( integral_out ) *integral_out = d.integral;
    for ( int i = 0; i < m; ++i ) {
        size_t k = 0; float p = 0.f;
        x[i] = terra_distribution_1d_sample ( &d, e[i], &p, &k );
        pdf[i] = p; idx[i] = ( uint32_t ) k;
    }
    terra_free ( d.cdf ); terra_free ( d.f );
}
void ref_distribution_2d ( const float* f, size_t nx, size_t ny, const float* e12, int m, float* xy2, float* pdf, float* marginal_cdf_out ) {
    TerraDistributon2D d;
    d.conditionals = ( TerraDistribution1D* ) terra_malloc ( sizeof ( TerraDistribution1D ) * ny );    /* the reference's init expects the rows allocated */
    terra_distribution_2d_init ( &d, f, nx, ny );
    if ( marginal_cdf_out ) memcpy ( marginal_cdf_out, d.marginal.cdf, ny * sizeof ( float ) );
    for ( int i = 0; i < m; ++i ) {
        float p = 0.f;
        TerraFloat2 s = terra_distribution_2d_sample ( &d, e12[2 * i], e12[2 * i + 1], &p );
        xy2[2 * i] = s.x; xy2[2 * i + 1] = s.y; pdf[i] = p;
    }
    for ( size_t i = 0; i < ny; ++i ) { terra_free ( d.conditionals[i].cdf ); terra_free ( d.conditionals[i].f ); }
    terra_free ( d.conditionals ); terra_free ( d.marginal.cdf ); terra_free ( d.marginal.f );
}

/* ---- per-pixel rendering ---------------------------------------------------
   One terra_render() call per pixel, streams re-keyed before each call, so the
   image does not depend on pixel visit order ("Per-pixel
   streams"). rand_calls (optional) is indexed like the framebuffer. */
void ref_render_pixels ( const TerraCamera* camera, HTerraScene scene, const TerraFramebuffer* fb,
                         size_t x, size_t y, size_t w, size_t h, uint64_t frame_seed, uint32_t* rand_calls ) {
    for ( size_t i = y; i < y + h; ++i ) {
        for ( size_t j = x; j < x + w; ++j ) {
            size_t pix = i * fb->width + j;
            ref_set_streams ( frame_seed, pix, ( uint64_t ) ( uint32_t ) fb->results[pix].samples );
            uint64_t before = ref_rand_count;
            terra_render ( camera, scene, fb, j, i, 1, 1 );

            if ( rand_calls ) {
                rand_calls[pix] = ( uint32_t ) ( ref_rand_count - before );
            }
        }
    }
}

typedef struct {
    const TerraCamera* camera; HTerraScene scene; const TerraFramebuffer* fb;
    size_t x, y, w, h; uint64_t frame_seed; uint32_t* rand_calls;
    volatile long* next_row; size_t rows_per_job;
} RefMtJob;

static void* ref_mt_worker ( void* p ) {
    RefMtJob* job = ( RefMtJob* ) p;
    for ( ;; ) {
        long r = __sync_fetch_and_add ( job->next_row, ( long ) job->rows_per_job );
        if ( ( size_t ) r >= job->h ) {
            break;
        }
        size_t rows = job->rows_per_job;
        if ( ( size_t ) r + rows > job->h ) {
            rows = job->h - ( size_t ) r;
        }
        ref_render_pixels ( job->camera, job->scene, job->fb, job->x, job->y + ( size_t ) r, job->w, rows, job->frame_seed, job->rand_calls );
    }
    return NULL;
}

/* Same result as ref_render_pixels (streams are per pixel), on nthreads
   threads with thread-local entropy: the fair multi-core timing of the
   reference (the unmodified global rand() anti-scales). */
void ref_render_pixels_mt ( const TerraCamera* camera, HTerraScene scene, const TerraFramebuffer* fb,
                            size_t x, size_t y, size_t w, size_t h, uint64_t frame_seed, uint32_t* rand_calls, int nthreads ) {
    volatile long next = 0;
    RefMtJob job = { camera, scene, fb, x, y, w, h, frame_seed, rand_calls, &next, 4 };
    if ( nthreads < 1 ) nthreads = 1;
    if ( nthreads > 256 ) nthreads = 256;
    pthread_t th[256];
    for ( int t = 0; t < nthreads; ++t ) pthread_create ( &th[t], NULL, ref_mt_worker, &job );
    for ( int t = 0; t < nthreads; ++t ) pthread_join ( th[t], NULL );
}

/* The unpinned-order path: ONE terra_render call over the tile with one
   stream pair (what the reference does natively). Used only to show the
   order dependence in tests. */
void ref_render_tile_shared_stream ( const TerraCamera* camera, HTerraScene scene, const TerraFramebuffer* fb,
                                     size_t x, size_t y, size_t w, size_t h, uint64_t frame_seed ) {
    ref_set_streams ( frame_seed, 0, 0 );
    terra_render ( camera, scene, fb, x, y, w, h );
}

/* ---- accessors for private state ------------------------------------------ */

int ref_bvh_node_count ( HTerraScene scene ) { return ( ( TerraScene* ) scene )->bvh.nodes_count; }
const void* ref_bvh_nodes ( HTerraScene scene ) { return ( ( TerraScene* ) scene )->bvh.nodes; }
size_t ref_lights_count ( HTerraScene scene ) { return ( ( TerraScene* ) scene )->lights_pop; }
size_t ref_lights_triangles_count ( HTerraScene scene ) { return ( ( TerraScene* ) scene )->lights_triangles_count; }
int ref_light_object_index ( HTerraScene scene, size_t i ) {
    TerraScene* s = ( TerraScene* ) scene;
    return ( int ) ( s->lights[i].object - s->objects );
}
float ref_light_area ( HTerraScene scene, size_t i ) { return ( ( TerraScene* ) scene )->lights[i].area; }
const float* ref_light_triangle_areas ( HTerraScene scene, size_t i ) { return ( ( TerraScene* ) scene )->lights[i].triangle_area; }
const TerraSceneOptions* ref_committed_options ( HTerraScene scene ) { return & ( ( TerraScene* ) scene )->opts; }

/* one primary sample through terra_trace with pinned streams; returns radiance
   and the number of rand() calls it made */
TerraFloat3 ref_trace_one ( HTerraScene scene, const TerraFloat3* origin, const TerraFloat3* dir,
                            uint64_t stateB, uint64_t incB, uint32_t* rand_calls ) {
    ref_set_raw_streams ( 0, stateB, incB );
    TerraRay ray = terra_ray ( origin, dir );
    uint64_t before = ref_rand_count;
    TerraFloat3 L = terra_trace ( ( TerraScene* ) scene, &ray );
    if ( rand_calls ) *rand_calls = ( uint32_t ) ( ref_rand_count - before );
    return L;
}

/* raycast wrapper returning plain indices: object index or -1 */
int ref_raycast ( HTerraScene scene, const TerraFloat3* origin, const TerraFloat3* dir,
                  TerraShadingSurface* surface, TerraFloat3* point, int* triangle ) {
    TerraScene* s = ( TerraScene* ) scene;
    TerraRay ray = terra_ray ( origin, dir );
    TerraRayState st;
    terra_ray_state_init ( &ray, &st );
    size_t tri = 0;
    TerraObject* o = terra_scene_raycast ( s, &ray, &st, surface, point, &tri );
    if ( !o ) return -1;
    *triangle = ( int ) tri;
    return ( int ) ( o - s->objects );
}

/* bvh traversal on the committed scene without the 0.001 origin push of raycast */
int ref_bvh_traverse ( HTerraScene scene, const TerraFloat3* origin, const TerraFloat3* dir, TerraFloat3* point, uint32_t* prim ) {
    TerraScene* s = ( TerraScene* ) scene;
    TerraRay ray = terra_ray ( origin, dir );
    TerraRayState st;
    terra_ray_state_init ( &ray, &st );
    TerraPrimitiveRef ref;
    memset ( &ref, 0, sizeof ref );
    bool found = terra_bvh_traverse ( &s->bvh, s->objects, &ray, &st, point, &ref );
    memcpy ( prim, &ref, 4 );
    return found ? 1 : 0;
}

/* watertight test for one (ray, triangle): returns hit and fills out[8] = u,v,w,depth,px,py,pz,0 */
int ref_watertight ( const TerraFloat3* origin, const TerraFloat3* dir, const TerraTriangle* tri, float* out ) {
    TerraRay ray = terra_ray ( origin, dir );
    TerraRayState st;
    terra_ray_triangle_intersection_init ( &ray, &st );
    TerraRayIntersectionQuery q;
    q.ray = &ray;
    q.state = &st;
    q.primitive.triangle = ( TerraTriangle* ) tri;
    TerraRayIntersectionResult r;
    memset ( &r, 0, sizeof r );
    int hit = terra_ray_triangle_intersection_query ( &q, &r );
    out[0] = r.u; out[1] = r.v; out[2] = r.w; out[3] = r.ray_depth;
    out[4] = r.point.x; out[5] = r.point.y; out[6] = r.point.z; out[7] = 0.f;
    return hit;
}

/* Moeller-Trumbore (exported-but-uncalled variant, reference src/Terra.c:880) */
int ref_moller_trumbore ( const TerraFloat3* origin, const TerraFloat3* dir, const TerraTriangle* tri, float* out ) {
    TerraRay ray = terra_ray ( origin, dir );
    TerraFloat3 p = { 0, 0, 0 };
    float t = 0;
    int hit = terra_ray_triangle_intersection ( &ray, tri, &p, &t ) ? 1 : 0;
    out[0] = t; out[1] = p.x; out[2] = p.y; out[3] = p.z;
    return hit;
}

int ref_ray_aabb ( const TerraFloat3* origin, const TerraFloat3* dir, const TerraAABB* box, float* tmin, float* tmax ) {
    TerraRay ray = terra_ray ( origin, dir );
    return terra_ray_aabb_intersection ( &ray, box, tmin, tmax ) ? 1 : 0;
}
