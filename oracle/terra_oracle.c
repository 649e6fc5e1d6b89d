/*
 * TEST INFRASTRUCTURE (oracle/). Not part of the product. See terra_oracle.h.
 *
 * CPU restatement of the reference hot path. Every function cites the reference
 * lines whose arithmetic it restates; the operation ORDER and the float/double
 * promotions are kept identical ("Mixed-precision traps"),
 * because one flipped branch changes a whole path. Build with
 * gcc -O2 -ffp-contract=off (oracle/Makefile).
 *
 * Deliberate differences from the reference (all UB / crashes there; SURVEY.md
 * section 7 "Reference UB to avoid"):
 *   - array growth passes byte counts to realloc (reference src/Terra.c:144,221 pass element counts);
 *   - lights_triangles_count is reset when lights are rebuilt (reference src/Terra.c:228 never resets it);
 *   - scenes with fewer than 2 triangles build a valid (degenerate) tree instead of
 *     running off the node array (reference src/TerraBVH.c:178-241);
 *   - a negative light-pick variate maps to light 0 (reference src/Terra.c:1618 casts a
 *     negative double to size_t);
 *   - traversal stack overflow (>64 entries, reference src/TerraBVH.c:252) is not reproduced:
 *     the stack here is sized from the tree;
 *   - surface->ior and attributes[i >= attributes_count] are zero/ior-initialised instead
 *     of left uninitialised (reference src/Terra.c:1758-1763).
 */
#include "terra_oracle.h"
#include "stream_key.h"
#include "oracle_devmath.h"

#include <stdio.h>
#include <stdarg.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <float.h>
#include <pthread.h>

typedef TerraFloat3 v3;

/* ------------------------------------------------------------------------- */
/* small vector helpers (value semantics; arithmetic = reference TerraMath.inl) */
/* ------------------------------------------------------------------------- */
static inline v3 v3_set ( float x, float y, float z ) { v3 r = { x, y, z }; return r; }
static inline v3 v3_add ( v3 a, v3 b ) { return v3_set ( a.x + b.x, a.y + b.y, a.z + b.z ); }
static inline v3 v3_sub ( v3 a, v3 b ) { return v3_set ( a.x - b.x, a.y - b.y, a.z - b.z ); }
static inline v3 v3_scale ( v3 a, float s ) { return v3_set ( a.x * s, a.y * s, a.z * s ); }
static inline v3 v3_mul ( v3 a, v3 b ) { return v3_set ( a.x * b.x, a.y * b.y, a.z * b.z ); }
static inline v3 v3_neg ( v3 a ) { return v3_set ( -a.x, -a.y, -a.z ); }
static inline float v3_dot ( v3 a, v3 b ) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static inline v3 v3_cross ( v3 a, v3 b ) {
    return v3_set ( a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x );
}
static inline float v3_len ( v3 a ) { return sqrtf ( a.x * a.x + a.y * a.y + a.z * a.z ); }
static inline v3 v3_norm ( v3 a ) { float l = v3_len ( a ); return v3_set ( a.x / l, a.y / l, a.z / l ); }
static inline float sel_min ( float a, float b ) { return a < b ? a : b; }   /* reference TerraMath.inl:175 */
static inline float sel_max ( float a, float b ) { return a > b ? a : b; }   /* reference TerraMath.inl:171 */
static inline float comp ( v3 a, int i ) { return i == 0 ? a.x : ( i == 1 ? a.y : a.z ); }
static inline v3 m3_apply ( const TerraFloat4x4* m, v3 v ) {                 /* reference TerraMath.inl:230 */
    return v3_set ( m->rows[0].x * v.x + m->rows[0].y * v.y + m->rows[0].z * v.z,
                    m->rows[1].x * v.x + m->rows[1].y * v.y + m->rows[1].z * v.z,
                    m->rows[2].x * v.x + m->rows[2].y * v.y + m->rows[2].z * v.z );
}

/* ------------------------------------------------------------------------- */
/* math mode                                                                  */
/* ------------------------------------------------------------------------- */
static int g_math_mode = ORC_MATH_LIBM;
void orc_set_math_mode ( int mode ) { g_math_mode = mode; }
int  orc_get_math_mode ( void ) { return g_math_mode; }
float orc_math_sinf ( float x )  { return g_math_mode == ORC_MATH_LIBM ? sinf ( x ) : orc_dm_sinf ( x ); }
float orc_math_cosf ( float x )  { return g_math_mode == ORC_MATH_LIBM ? cosf ( x ) : orc_dm_cosf ( x ); }
float orc_math_powf ( float x, float y ) { return g_math_mode == ORC_MATH_LIBM ? powf ( x, y ) : orc_dm_powf ( x, y ); }
float orc_math_acosf ( float x ) { return g_math_mode == ORC_MATH_LIBM ? acosf ( x ) : orc_dm_acosf ( x ); }
float orc_math_atan2f ( float y, float x ) { return g_math_mode == ORC_MATH_LIBM ? atan2f ( y, x ) : orc_dm_atan2f ( y, x ); }

/* ------------------------------------------------------------------------- */
/* private types                                                              */
/* ------------------------------------------------------------------------- */
typedef struct {            /* reference src/TerraBVH.h:13-17, 64 bytes */
    TerraAABB aabb[2];
    int32_t   index[2];
    int32_t   type[2];      /* -1 inner, 1 triangle leaf, 0 empty (oracle-only, <2 triangles) */
} OrcBVHNode;

typedef struct {            /* reference src/TerraPrivate.h:24-29 */
    v3     power;
    float  area;
    int    object;
    float* triangle_area;
} OrcLight;

typedef struct {
    TerraSceneOptions opts, new_opts;
    TerraObject* objects; size_t objects_pop, objects_cap;
    OrcLight*    lights;  size_t lights_pop, lights_cap;
    size_t       lights_triangles_count;
    v3           total_light_power;
    OrcBVHNode*  nodes;   int nodes_count; int max_stack;
    bool         dirty_objects, dirty_lights;
    uint64_t     frame_seed;
    int          env_lighting;   /* extension, off by default: see orc_set_environment_lighting */
    int          sampler_integration;   /* extension, off by default: see orc_set_sampler_integration */
    int          env_sampling;          /* extension, off by default: see orc_set_environment_sampling */
    /* the lat-long environment map as a TerraDistribution2D (src/Terra.c:812-846), rebuilt by orc_scene_commit while env_sampling is on */
    float*       env_f; float* env_cdf; float* env_row_f; float* env_row_cdf; float env_integral; size_t env_nx, env_ny;
} OrcScene;

typedef struct { v3 origin, direction, inv_direction; } OrcRay;            /* reference src/TerraPrivate.h:107-111 */
typedef struct { float shearx, sheary, scalez; int ix, iy, iz; } OrcRayState; /* reference src/TerraPrivate.h:114-120 */

/* per-thread entropy + counters */
typedef struct {
    OrcPcg32    streamB;
    OrcCounters c;
} OrcThread;
static __thread OrcThread tls;
static OrcCounters g_counters;
static pthread_mutex_t g_counters_lock = PTHREAD_MUTEX_INITIALIZER;

void orc_counters_reset ( void ) {
    pthread_mutex_lock ( &g_counters_lock );
    memset ( &g_counters, 0, sizeof g_counters );
    pthread_mutex_unlock ( &g_counters_lock );
    memset ( &tls.c, 0, sizeof tls.c );
}
static void counters_flush ( void ) {
    pthread_mutex_lock ( &g_counters_lock );
    uint64_t* d = ( uint64_t* ) &g_counters; const uint64_t* s = ( const uint64_t* ) &tls.c;
    for ( size_t i = 0; i < sizeof ( OrcCounters ) / 8; ++i ) d[i] += s[i];
    pthread_mutex_unlock ( &g_counters_lock );
    memset ( &tls.c, 0, sizeof tls.c );
}
void orc_counters_get ( OrcCounters* out ) {
    counters_flush();
    pthread_mutex_lock ( &g_counters_lock );
    *out = g_counters;
    pthread_mutex_unlock ( &g_counters_lock );
}

/* (float)rand() / RAND_MAX with rand() = 24 random bits << 7  (reference src/Terra.c:115; stream_key.h) */
static inline float randf ( void ) {
    ++tls.c.rand_calls;
    return ( float ) orc_rand_from_stream ( &tls.streamB ) / 2147483648.f;
}

/* ------------------------------------------------------------------------- */
/* system                                                                     */
/* ------------------------------------------------------------------------- */
void* orc_malloc ( size_t size ) { return malloc ( size ); }
void* orc_realloc ( void* p, size_t size ) { return realloc ( p, size ); }
void  orc_free ( void* p ) { free ( p ); }
void  orc_log ( const char* str, ... ) { va_list a; va_start ( a, str ); vfprintf ( stdout, str, a ); va_end ( a ); }

/* ------------------------------------------------------------------------- */
/* attributes, textures (reference src/Terra.c:287-304, :350-507, :1804-1810)  */
/* ------------------------------------------------------------------------- */
void orc_attribute_init_constant ( TerraAttribute* a, const TerraFloat3* value ) {
    a->state = NULL; a->finalize = NULL; a->eval = NULL; a->value = *value;
}
void orc_attribute_init_texture ( TerraAttribute* a, TerraTexture* t ) { a->state = t; a->eval = orc_texture_sample; a->finalize = orc_texture_finalize; }
void orc_attribute_init_cubemap ( TerraAttribute* a, TerraTexture* t ) { a->state = t; a->eval = orc_texture_sample_latlong; a->finalize = orc_texture_finalize; }

static v3 attribute_eval ( const TerraAttribute* a, const void* uv, const v3* xyz ) {
    if ( a->state != NULL ) return a->eval ( a->state, uv, xyz );
    return a->value;
}

bool orc_texture_init ( TerraTexture* t, size_t w, size_t h, size_t comps, const void* data ) {
    t->pixels = malloc ( w * h * comps );
    memcpy ( t->pixels, data, w * h * comps );
    t->width = ( uint16_t ) w; t->height = ( uint16_t ) h; t->components = ( uint8_t ) comps; t->depth = 1;
    return true;
}
bool orc_texture_init_hdr ( TerraTexture* t, size_t w, size_t h, size_t comps, const float* data ) {
    t->pixels = malloc ( sizeof ( float ) * w * h * comps );
    memcpy ( t->pixels, data, sizeof ( float ) * w * h * comps );
    t->width = ( uint16_t ) w; t->height = ( uint16_t ) h; t->components = ( uint8_t ) comps; t->depth = 4;
    return true;
}
TerraFloat3 orc_texture_read ( TerraTexture* t, size_t x, size_t y ) {      /* reference src/Terra.c:368-408 */
    size_t W = t->width, H = t->height;
    switch ( t->address_mode ) {
        case kTerraTextureAddressClamp: x = x < W - 1 ? x : W - 1; y = y < H - 1 ? y : H - 1; break;
        case kTerraTextureAddressWrap:  x %= W; y %= H; break;
        default: /* mirror */
            if ( ( x / W ) % 2 == 0 ) { x %= W; y %= H; }
            else {  /* the reference's mirror branch yields x == W / y == H (one past the end) whenever x%W == 0 / y%H == 0
                       (src/Terra.c:385-386): clamped here, as in the product */
                x = W - ( x % W ); y = H - ( y % H );
                if ( x > W - 1 ) x = W - 1;
                if ( y > H - 1 ) y = H - 1;
            }
            break;
    }
    if ( t->depth == 1 ) {
        const uint8_t* p = ( const uint8_t* ) t->pixels + ( y * W + x ) * t->components;
        return v3_set ( p[0] / 255.f, p[1] / 255.f, p[2] / 255.f );
    }
    const float* p = ( const float* ) t->pixels + ( y * W + x ) * t->components;
    return v3_set ( p[0], p[1], p[2] );
}
TerraFloat3 orc_texture_sample ( void* tex, const void* uvp, const void* xyz ) {   /* reference src/Terra.c:410-466 */
    TerraTexture* t = ( TerraTexture* ) tex; const TerraFloat2* uv = ( const TerraFloat2* ) uvp; ( void ) xyz;
    size_t ix = ( size_t ) uv->x, iy = ( size_t ) uv->y;
    v3 s = v3_set ( 0, 0, 0 );
    if ( t->filter == kTerraFilterPoint ) return orc_texture_read ( t, ix, iy );
    if ( t->filter == kTerraFilterBilinear ) {
        size_t x2 = ix + 1 < ( size_t ) t->width - 1 ? ix + 1 : ( size_t ) t->width - 1;
        size_t y3 = iy + 1 < ( size_t ) t->height - 1 ? iy + 1 : ( size_t ) t->height - 1;
        v3 n1 = orc_texture_read ( t, ix, iy ), n2 = orc_texture_read ( t, x2, iy );
        v3 n3 = orc_texture_read ( t, ix, y3 ), n4 = orc_texture_read ( t, x2, y3 );
        float wu = uv->x - ix, wv = uv->y - iy, wou = 1.f - wu, wov = 1.f - wv;
        s.x = ( n1.x * wou + n2.x * wu ) * wov + ( n3.x * wou + n4.x * wu ) * wv;
        s.y = ( n1.y * wou + n2.y * wu ) * wov + ( n3.y * wou + n4.y * wu ) * wv;
        s.z = ( n1.z * wou + n2.z * wu ) * wov + ( n3.z * wou + n4.z * wu ) * wv;
    }
    return s;
}
TerraFloat3 orc_texture_sample_latlong ( void* tex, const void* dirp, const void* xyz ) { /* reference src/Terra.c:468-477 */
    TerraTexture* t = ( TerraTexture* ) tex; ( void ) xyz;
    v3 d = v3_norm ( * ( const v3* ) dirp );
    float theta = orc_math_acosf ( d.y );
    float phi = orc_math_atan2f ( d.z, d.x ) + terra_PI;
    size_t u = ( size_t ) ( ( phi / ( 2 * terra_PI ) ) * t->width );      /* < width: terra_PI exceeds pi, so phi / (2 terra_PI) < 1 */
    size_t v = ( size_t ) ( ( theta / ( terra_PI ) ) * t->height );
    return orc_texture_read ( t, u, v );
}
void orc_texture_destroy ( TerraTexture* t ) { free ( t->pixels ); t->pixels = NULL; }
void orc_texture_finalize ( void* tex ) {                                      /* reference src/Terra.c:484-507 */
    TerraTexture* t = ( TerraTexture* ) tex;
    if ( !t || !t->pixels ) return;
    size_t n = ( size_t ) t->width * t->height * t->components;
    if ( t->depth == 1 ) { uint8_t* p = ( uint8_t* ) t->pixels; for ( size_t i = 0; i < n; ++i ) p[i] = ( uint8_t ) ( powf ( p[i] / 255.f, 2.2f ) * 255 ); }
    else if ( t->depth == 4 ) { float* p = ( float* ) t->pixels; for ( size_t i = 0; i < n; ++i ) p[i] = powf ( p[i], 2.2f ); }
}

/* ------------------------------------------------------------------------- */
/* framebuffer (reference src/Terra.c:309-345)                                 */
/* ------------------------------------------------------------------------- */
bool orc_framebuffer_create ( TerraFramebuffer* fb, size_t w, size_t h ) {
    if ( w == 0 || h == 0 ) return false;
    fb->width = w; fb->height = h;
    fb->pixels = ( TerraFloat3* ) calloc ( w * h, sizeof ( TerraFloat3 ) );
    fb->results = ( TerraRawIntegrationResult* ) calloc ( w * h, sizeof ( TerraRawIntegrationResult ) );
    return true;
}
void orc_framebuffer_clear ( TerraFramebuffer* fb ) {
    memset ( fb->pixels, 0, fb->width * fb->height * sizeof ( TerraFloat3 ) );
    memset ( fb->results, 0, fb->width * fb->height * sizeof ( TerraRawIntegrationResult ) );
}
void orc_framebuffer_destroy ( TerraFramebuffer* fb ) {
    if ( !fb ) return;
    free ( fb->results ); free ( fb->pixels ); fb->results = NULL; fb->pixels = NULL;
}

/* ------------------------------------------------------------------------- */
/* A2: camera-jitter PCG32 (reference src/Terra.c:678-701)                     */
/* ------------------------------------------------------------------------- */
static void pcgA_init ( OrcPcg32* g, uint32_t seed ) {
    g->state = 0; g->inc = 1;
    orc_pcg32_next ( g );
    g->state += seed;
    orc_pcg32_next ( g );
}
static inline float pcgA_nextf ( OrcPcg32* g ) {
    uint32_t r = orc_pcg32_next ( g );
    float resolution = 1.f / ( float ) ( ( uint64_t ) 1 << 32 );
    return r * resolution;      /* u32 -> float rounds to nearest; may return 1.0f */
}
void orc_pcg_floats ( uint32_t seed, int n, float* out ) {
    OrcPcg32 g; pcgA_init ( &g, seed );
    for ( int i = 0; i < n; ++i ) out[i] = pcgA_nextf ( &g );
}

/* ------------------------------------------------------------------------- */
/* N4, unit level: stratified / Halton samplers, 1D / 2D        */
/* distributions (reference src/Terra.c:703-755, 760-846). The reference        */
/* constructs a stratified or Halton sampler per pixel and never draws from it  */
/* (src/Terra.c:535-548); the distributions have no caller at all. Restated so  */
/* that the device versions can be pinned to the compiled reference.            */
/* ------------------------------------------------------------------------- */
/* terra_sampler_stratified_next_pair (src/Terra.c:714-723): stratum = next / samples, x = stratum % strata, y = stratum / strata;
   (size_t + float) is a float sum; the clamp is (float)(1.f - 1e-4) with the subtraction in double */
void orc_stratified_pairs ( uint32_t seed, int strata, int samples, int n, float* out2 ) {
    OrcPcg32 g; pcgA_init ( &g, seed );
    const float stratum_size = 1.f / strata;
    const float top = ( float ) ( 1.f - terra_Epsilon );
    for ( int next = 0; next < n; ++next ) {
        size_t stratum = ( size_t ) next / ( size_t ) samples;
        size_t x = stratum % ( size_t ) strata, y = stratum / ( size_t ) strata;
        float a = ( ( float ) x + pcgA_nextf ( &g ) ) * stratum_size;
        float b = ( ( float ) y + pcgA_nextf ( &g ) ) * stratum_size;
        out2[2 * next] = sel_min ( a, top );
        out2[2 * next + 1] = sel_min ( b, top );
    }
}
/* terra_radical_inverse (src/Terra.c:734-748): digits reversed in integer arithmetic, denominator as a float product */
float orc_radical_inverse ( uint64_t base, uint64_t a ) {
    float inv_base = 1.f / ( float ) base;
    uint64_t seq = 0;
    float denom = 1;
    while ( a ) {
        uint64_t next = a / base;
        uint64_t digit = a - next * base;
        seq = seq * base + digit;
        denom *= inv_base;
        a = next;
    }
    return sel_min ( ( float ) seq * denom, ( float ) ( 1.f - terra_Epsilon ) );
}
/* terra_sampler_halton_next_pair (src/Terra.c:750-755): bases 3 and 2 (src/Terra.c:725-729); `next` is an int */
void orc_halton_pairs ( int first, int n, float* out2 ) {
    for ( int i = 0; i < n; ++i ) {
        out2[2 * i] = orc_radical_inverse ( 3, ( uint64_t ) ( first + i ) );
        out2[2 * i + 1] = orc_radical_inverse ( 2, ( uint64_t ) ( first + i ) );
    }
}
/* terra_distribution_1d_init (src/Terra.c:760-780): running float sum, then every entry divided by the total */
static float dist1d_init ( const float* f, size_t n, float* cdf ) {
    float integral = 0;
    for ( size_t i = 0; i < n; ++i ) { integral += f[i]; cdf[i] = integral; }
    for ( size_t i = 0; i < n; ++i ) cdf[i] /= integral;
    return integral;
}
/* terra_distribution_1d_sample (src/Terra.c:781-810): first bucket with e < cdf[i], linear interpolation inside it;
   (size_t + float) / size_t in float. The reference asserts when no bucket is found; here: FLT_MAX, pdf and idx untouched */
static float dist1d_sample ( const float* f, const float* cdf, size_t n, float integral, float e, float* pdf, uint32_t* idx ) {
    float prev = 0;
    for ( size_t i = 0; i < n; ++i ) {
        float curr = cdf[i];
        if ( e < curr ) {
            if ( pdf ) *pdf = f[i] / integral;
            if ( idx ) *idx = ( uint32_t ) i;
            float d = e - prev;
            d /= curr - prev;
            return ( ( float ) i + d ) / ( float ) n;
        }
        prev = curr;
    }
    return FLT_MAX;
}
void orc_distribution_1d ( const float* f, size_t n, const float* e, int m, float* x, float* pdf, uint32_t* idx, float* cdf_out, float* integral_out ) {
    float* cdf = ( float* ) malloc ( sizeof ( float ) * ( n ? n : 1 ) );
    float integral = dist1d_init ( f, n, cdf );
    if ( cdf_out ) memcpy ( cdf_out, cdf, n * sizeof ( float ) );
    if ( integral_out ) *integral_out = integral;
    for ( int i = 0; i < m; ++i ) { pdf[i] = 0.f; idx[i] = 0; x[i] = dist1d_sample ( f, cdf, n, integral, e[i], &pdf[i], &idx[i] ); }
    free ( cdf );
}
/* terra_distribution_2d_init / _sample (src/Terra.c:812-846): one 1D distribution per row, the marginal over the rows' integrals;
   sample the row with e1, then the column inside that row with e2; pdf = product */
void orc_distribution_2d ( const float* f, size_t nx, size_t ny, const float* e12, int m, float* xy2, float* pdf, float* marginal_cdf_out ) {
    const size_t cells = nx * ny;
    float* cdf = ( float* ) malloc ( sizeof ( float ) * ( cells > 0 ? cells : 1 ) );
    float* integrals = ( float* ) malloc ( sizeof ( float ) * ( ny ? ny : 1 ) );
    float* mcdf = ( float* ) malloc ( sizeof ( float ) * ( ny ? ny : 1 ) );
    for ( size_t i = 0; i < ny; ++i ) integrals[i] = dist1d_init ( f + nx * i, nx, cdf + nx * i );
    float integral = 0;
    for ( size_t i = 0; i < ny; ++i ) { integral += integrals[i]; mcdf[i] = integral; }
    for ( size_t i = 0; i < ny; ++i ) mcdf[i] /= integral;
    if ( marginal_cdf_out ) memcpy ( marginal_cdf_out, mcdf, ny * sizeof ( float ) );
    for ( int k = 0; k < m; ++k ) {
        float p0 = 0.f, p1 = 0.f; uint32_t row = 0;
        float s1 = dist1d_sample ( integrals, mcdf, ny, integral, e12[2 * k], &p0, &row );
        if ( s1 == FLT_MAX ) { xy2[2 * k] = xy2[2 * k + 1] = FLT_MAX; pdf[k] = 0.f; continue; }       /* (the reference would read an uninitialised row index) */
        float s2 = dist1d_sample ( f + nx * row, cdf + nx * row, nx, integrals[row], e12[2 * k + 1], &p1, NULL );
        xy2[2 * k] = s1; xy2[2 * k + 1] = s2; pdf[k] = p0 * p1;
    }
    free ( cdf ); free ( integrals ); free ( mcdf );
}

/* ------------------------------------------------------------------------- */
/* A10/A11: rays and camera (reference src/Terra.c:1702-1724, :1770-1799)      */
/* ------------------------------------------------------------------------- */
static OrcRay make_ray ( v3 o, v3 d ) {
    OrcRay r; r.origin = o; r.direction = d;
    r.inv_direction = v3_set ( 1.f / d.x, 1.f / d.y, 1.f / d.z );
    return r;
}
static OrcRay surface_ray ( const TerraShadingSurface* s, v3 p, v3 d, float sign ) {
    v3 off = v3_scale ( s->normal, 0.0001f * sign );
    return make_ray ( v3_add ( p, off ), d );
}
TerraFloat4x4 orc_camera_frame ( const TerraCamera* c ) {
    v3 z = v3_norm ( c->direction );
    v3 x = v3_norm ( v3_cross ( c->up, z ) );
    v3 y = v3_cross ( z, x );
    TerraFloat4x4 m;
    m.rows[0] = terra_f4_set ( x.x, y.x, z.x, 0.f );
    m.rows[1] = terra_f4_set ( x.y, y.y, z.y, 0.f );
    m.rows[2] = terra_f4_set ( x.z, y.z, z.z, 0.f );
    m.rows[3] = terra_f4_set ( 0.f, 0.f, 0.f, 1.f );
    return m;
}
TerraFloat3 orc_camera_sample ( const TerraCamera* c, size_t W, size_t H, size_t x, size_t y, float jitter, float r1, float r2 ) {
    float dx = -jitter + 2 * r1 * jitter;
    float dy = -jitter + 2 * r2 * jitter;
    float ndc_x = ( x + 0.5f + dx ) / W;
    float ndc_y = ( y + 0.5f + dy ) / H;
    float sx = 2 * ndc_x - 1;
    float sy = 1 - 2 * ndc_y;
    float aspect = ( float ) W / ( float ) H;
    float t = ( float ) tan ( ( c->fov * 0.0174533f ) / 2 );      /* double tan of a float argument */
    float fx = sx * aspect * t;
    float fy = sy * t;
    return v3_norm ( v3_set ( fx, fy, 1.f ) );
}

/* ------------------------------------------------------------------------- */
/* A6: slab test (reference src/Terra.c:851-878)                               */
/* ------------------------------------------------------------------------- */
static inline bool ray_aabb ( const OrcRay* r, const TerraAABB* b, float* tmin_out, float* tmax_out ) {
    float t1 = ( b->min.x - r->origin.x ) * r->inv_direction.x;
    float t2 = ( b->max.x - r->origin.x ) * r->inv_direction.x;
    float tmin = sel_min ( t1, t2 ), tmax = sel_max ( t1, t2 );
    t1 = ( b->min.y - r->origin.y ) * r->inv_direction.y;
    t2 = ( b->max.y - r->origin.y ) * r->inv_direction.y;
    tmin = sel_max ( tmin, sel_min ( t1, t2 ) ); tmax = sel_min ( tmax, sel_max ( t1, t2 ) );
    t1 = ( b->min.z - r->origin.z ) * r->inv_direction.z;
    t2 = ( b->max.z - r->origin.z ) * r->inv_direction.z;
    tmin = sel_max ( tmin, sel_min ( t1, t2 ) ); tmax = sel_min ( tmax, sel_max ( t1, t2 ) );
    if ( tmax > sel_max ( tmin, 0.f ) ) {
        if ( tmin_out ) *tmin_out = tmin;
        if ( tmax_out ) *tmax_out = tmax;
        return true;
    }
    return false;
}
int orc_ray_aabb ( const TerraFloat3* o, const TerraFloat3* d, const TerraAABB* box, float* tmin, float* tmax ) {
    OrcRay r = make_ray ( *o, *d );
    return ray_aabb ( &r, box, tmin, tmax ) ? 1 : 0;
}

/* ------------------------------------------------------------------------- */
/* A7: watertight ray/triangle (reference src/TerraGeometry.c:98-138, :159-260) */
/* ------------------------------------------------------------------------- */
static void ray_state_init ( const OrcRay* r, OrcRayState* s ) {
    v3 a = v3_set ( fabsf ( r->direction.x ), fabsf ( r->direction.y ), fabsf ( r->direction.z ) );
    int iz = a.x > a.y ? ( a.x > a.z ? 0 : 2 ) : ( a.y > a.z ? 1 : 2 );   /* reference TerraMath.inl:155-169 */
    int ix = iz + 1 == 3 ? 0 : iz + 1;
    int iy = ix + 1 == 3 ? 0 : ix + 1;
    if ( comp ( r->direction, iz ) < 0.f ) { int t = ix; ix = iy; iy = t; }
    float scalez = 1.f / comp ( r->direction, iz );
    s->shearx = comp ( r->direction, ix ) * scalez;
    s->sheary = comp ( r->direction, iy ) * scalez;
    s->scalez = scalez;
    s->ix = ix; s->iy = iy; s->iz = iz;
}

typedef struct { float u, v, w, depth; v3 point; } OrcTriHit;

static inline bool watertight ( const OrcRay* r, const OrcRayState* s, const TerraTriangle* t, OrcTriHit* h ) {
    v3 A = v3_sub ( t->a, r->origin ), B = v3_sub ( t->b, r->origin ), C = v3_sub ( t->c, r->origin );
    float Aiz = comp ( A, s->iz ), Biz = comp ( B, s->iz ), Ciz = comp ( C, s->iz );
    float Ax = comp ( A, s->ix ) - s->shearx * Aiz, Ay = comp ( A, s->iy ) - s->sheary * Aiz;
    float Bx = comp ( B, s->ix ) - s->shearx * Biz, By = comp ( B, s->iy ) - s->sheary * Biz;
    float Cx = comp ( C, s->ix ) - s->shearx * Ciz, Cy = comp ( C, s->iy ) - s->sheary * Ciz;
    float U = Cx * By - Cy * Bx;
    float V = Ax * Cy - Ay * Cx;
    float W = Bx * Ay - By * Ax;
    if ( U == 0.f || V == 0.f || W == 0.f ) {        /* double fallback, :204-208 */
        U = ( float ) ( ( double ) Cx * ( double ) By - ( double ) Cy * ( double ) Bx );
        V = ( float ) ( ( double ) Ax * ( double ) Cy - ( double ) Ay * ( double ) Cx );
        W = ( float ) ( ( double ) Bx * ( double ) Ay - ( double ) By * ( double ) Ax );
    }
    uint32_t sign = orc_dm_bits ( U ) & 0x80000000u;       /* sign BITS must agree: -0 != +0 */
    if ( sign != ( orc_dm_bits ( V ) & 0x80000000u ) ) return false;
    if ( sign != ( orc_dm_bits ( W ) & 0x80000000u ) ) return false;
    float det = U + V + W;
    if ( det == 0.f ) return false;
    float Az = s->scalez * Aiz, Bz = s->scalez * Biz, Cz = s->scalez * Ciz;
    float depth = U * Az + V * Bz + W * Cz;
    if ( orc_dm_float ( orc_dm_bits ( depth ) ^ sign ) < 0.f ) return false;
    float inv_det = 1.f / det;
    h->u = U * inv_det; h->v = V * inv_det; h->w = W * inv_det;
    h->depth = depth * inv_det;
    h->point = v3_add ( r->origin, v3_scale ( r->direction, h->depth ) );      /* reference TerraGeometry.c:9-12 */
    return true;
}
int orc_watertight ( const TerraFloat3* o, const TerraFloat3* d, const TerraTriangle* tri, float* out ) {
    OrcRay r = make_ray ( *o, *d ); OrcRayState s; ray_state_init ( &r, &s );
    OrcTriHit h; memset ( &h, 0, sizeof h );
    int hit = watertight ( &r, &s, tri, &h ) ? 1 : 0;
    out[0] = h.u; out[1] = h.v; out[2] = h.w; out[3] = h.depth; out[4] = h.point.x; out[5] = h.point.y; out[6] = h.point.z; out[7] = 0.f;
    return hit;
}

/* A7': Moeller-Trumbore, the exported-but-uncalled variant (reference src/Terra.c:880-922) */
int orc_moller_trumbore ( const TerraFloat3* o, const TerraFloat3* d, const TerraTriangle* tri, float* out ) {
    v3 e1 = v3_sub ( tri->b, tri->a ), e2 = v3_sub ( tri->c, tri->a );
    v3 h = v3_cross ( *d, e2 );
    float a = v3_dot ( e1, h );
    out[0] = out[1] = out[2] = out[3] = 0.f;
    if ( a > -terra_Epsilon && a < terra_Epsilon ) return 0;        /* double compares */
    float f = 1 / a;
    v3 s = v3_sub ( *o, tri->a );
    float u = f * ( v3_dot ( s, h ) );
    if ( u < 0.f || u > 1.f ) return 0;
    v3 q = v3_cross ( s, e1 );
    float v = f * v3_dot ( *d, q );
    if ( v < 0.f || u + v > 1.f ) return 0;
    float t = f * v3_dot ( e2, q );
    if ( t > 0.00001f ) {
        v3 p = v3_add ( v3_scale ( *d, t ), *o );
        out[0] = t; out[1] = p.x; out[2] = p.y; out[3] = p.z;
        return 1;
    }
    return 0;
}

/* ------------------------------------------------------------------------- */
/* A16: BVH build (reference src/TerraBVH.c:24-35, :70-126, :128-244; Terra.c:972-997) */
/* ------------------------------------------------------------------------- */
typedef struct { TerraAABB aabb; uint32_t index; int type; } OrcVolume;

static void fit_triangle ( TerraAABB* b, const TerraTriangle* t ) {
    b->min.x = sel_min ( sel_min ( sel_min ( b->min.x, t->a.x ), t->b.x ), t->c.x );
    b->min.y = sel_min ( sel_min ( sel_min ( b->min.y, t->a.y ), t->b.y ), t->c.y );
    b->min.z = sel_min ( sel_min ( sel_min ( b->min.z, t->a.z ), t->b.z ), t->c.z );
    b->min.x -= terra_Epsilon; b->min.y -= terra_Epsilon; b->min.z -= terra_Epsilon;    /* double subtract, round */
    b->max.x = sel_max ( sel_max ( sel_max ( b->max.x, t->a.x ), t->b.x ), t->c.x );
    b->max.y = sel_max ( sel_max ( sel_max ( b->max.y, t->a.y ), t->b.y ), t->c.y );
    b->max.z = sel_max ( sel_max ( sel_max ( b->max.z, t->a.z ), t->b.z ), t->c.z );
    b->max.x += terra_Epsilon; b->max.y += terra_Epsilon; b->max.z += terra_Epsilon;
}
static void fit_aabb ( TerraAABB* b, const TerraAABB* o ) {
    b->min.x = sel_min ( b->min.x, o->min.x );
    b->min.y = sel_min ( b->min.y, o->min.y );
    b->min.z = sel_min ( b->min.z, o->min.z );
    b->max.x = sel_max ( b->max.x, o->max.x ) + terra_Epsilon;      /* grows by eps per merge */
    b->max.y = sel_max ( b->max.y, o->max.y ) + terra_Epsilon;
    b->max.z = sel_max ( b->max.z, o->max.z ) + terra_Epsilon;
}
static TerraAABB empty_aabb ( void ) {
    TerraAABB b; b.min = v3_set ( FLT_MAX, FLT_MAX, FLT_MAX ); b.max = v3_set ( -FLT_MAX, -FLT_MAX, -FLT_MAX ); return b;
}
static float surface_area ( const TerraAABB* b ) {
    float w = b->max.x - b->min.x, h = b->max.y - b->min.y, d = b->max.z - b->min.z;
    return 2 * ( w * d + w * h + d * h );
}
static inline float center_x ( const TerraAABB* b ) { return ( b->min.x + b->max.x ) / 2; }

/* The reference qsorts each range with a comparator returning (left.x < right.x)
   as an int (reference src/TerraBVH.c:37-46, :93). With glibc's merge sort
   ("take the left run's head when cmp <= 0") that is a STABLE sort by DESCENDING
   centre x. A bottom-up stable merge sort with the same take-rule is restated
   here; re-sorting an already sorted sub-range is the identity, so one global
   sort replaces the per-node qsort calls. */
static void sort_volumes_desc_x ( OrcVolume* v, int n ) {
    if ( n < 2 ) return;
    OrcVolume* tmp = ( OrcVolume* ) malloc ( sizeof ( OrcVolume ) * ( size_t ) n );
    OrcVolume* src = v; OrcVolume* dst = tmp;
    for ( int width = 1; width < n; width *= 2 ) {
        for ( int lo = 0; lo < n; lo += 2 * width ) {
            int mid = lo + width < n ? lo + width : n, hi = lo + 2 * width < n ? lo + 2 * width : n;
            int i = lo, j = mid, k = lo;
            while ( i < mid && j < hi ) {
                /* cmp(left,right) = left.x < right.x ; take left when cmp <= 0 */
                if ( ! ( center_x ( &src[i].aabb ) < center_x ( &src[j].aabb ) ) ) dst[k++] = src[i++];
                else dst[k++] = src[j++];
            }
            while ( i < mid ) dst[k++] = src[i++];
            while ( j < hi ) dst[k++] = src[j++];
        }
        OrcVolume* t = src; src = dst; dst = t;
    }
    if ( src != v ) memcpy ( v, src, sizeof ( OrcVolume ) * ( size_t ) n );
    free ( tmp );
}

/* sweep SAH over a sorted range; first minimum wins (reference src/TerraBVH.c:79-126) */
static int sah_split ( const OrcVolume* v, int n, const TerraAABB* container, float* left_area, float* right_area ) {
    float container_area = container ? surface_area ( container ) : FLT_MAX;
    TerraAABB b = empty_aabb();
    for ( int i = 0; i < n; ++i ) { fit_aabb ( &b, &v[i].aabb ); left_area[i] = surface_area ( &b ); }
    b = empty_aabb();
    for ( int i = n - 1; i >= 0; --i ) { fit_aabb ( &b, &v[i].aabb ); right_area[i] = surface_area ( &b ); }
    float min_cost = FLT_MAX; int best = -1;
    for ( int i = 0; i < n; ++i ) {
        int lc = i + 1, rc = n - lc;
        float cost = lc * left_area[i] / container_area + rc * right_area[i] / container_area;
        if ( cost < min_cost ) { min_cost = cost; best = i; }
    }
    return best;
}

static void bvh_destroy ( OrcScene* s ) { free ( s->nodes ); s->nodes = NULL; s->nodes_count = 0; }

static void bvh_build ( OrcScene* s ) {
    int n = 0;
    for ( size_t i = 0; i < s->objects_pop; ++i ) n += ( int ) s->objects[i].triangles_count;
    TerraAABB scene_aabb = empty_aabb();
    OrcVolume* vol = ( OrcVolume* ) malloc ( sizeof ( OrcVolume ) * ( size_t ) ( n > 0 ? n : 1 ) );
    int p = 0;
    for ( size_t j = 0; j < s->objects_pop; ++j )
        for ( size_t i = 0; i < s->objects[j].triangles_count; ++i, ++p ) {
            vol[p].aabb = empty_aabb();
            fit_triangle ( &scene_aabb, &s->objects[j].triangles[i] );     /* eps accumulates per triangle */
            fit_triangle ( &vol[p].aabb, &s->objects[j].triangles[i] );
            vol[p].type = 1;
            vol[p].index = ( uint32_t ) ( ( int ) j | ( ( int ) i << 8 ) );
        }
    s->nodes = ( OrcBVHNode* ) calloc ( ( size_t ) ( n > 0 ? 2 * n : 1 ), sizeof ( OrcBVHNode ) );
    s->nodes_count = 1;
    if ( n < 2 ) {      /* reference cannot build these (see header comment) */
        s->nodes[0].type[0] = n == 1 ? 1 : 0; s->nodes[0].type[1] = 0;
        if ( n == 1 ) { s->nodes[0].aabb[0] = vol[0].aabb; s->nodes[0].index[0] = ( int32_t ) vol[0].index; }
        s->max_stack = 1;
        free ( vol );
        return;
    }
    sort_volumes_desc_x ( vol, n );
    typedef struct { int start, end, node; const TerraAABB* aabb; } Task;
    Task* stack = ( Task* ) malloc ( sizeof ( Task ) * ( size_t ) ( 2 * n ) );
    float* la = ( float* ) malloc ( sizeof ( float ) * ( size_t ) n );
    float* ra = ( float* ) malloc ( sizeof ( float ) * ( size_t ) n );
    int sp = 0;
    Task root = { 0, n, 0, &scene_aabb };
    stack[sp++] = root;
    while ( sp > 0 ) {
        Task t = stack[--sp];
        int cnt = t.end - t.start;
        int best = sah_split ( vol + t.start, cnt, t.aabb, la, ra );
        if ( best < 0 ) best = 0;                    /* all-NaN costs: reference would index -1 */
        if ( best > cnt - 2 ) best = cnt - 2;        /* reference loops forever if the last index wins; it cannot without NaNs */
        int split = best + t.start;
        OrcBVHNode* node = &s->nodes[t.node];
        if ( split == t.start ) {
            node->type[0] = vol[t.start].type; node->aabb[0] = vol[t.start].aabb; node->index[0] = ( int32_t ) vol[t.start].index;
        } else {
            node->type[0] = -1;
            TerraAABB b = empty_aabb();
            for ( int i = t.start; i < split + 1; ++i ) fit_aabb ( &b, &vol[i].aabb );
            node->aabb[0] = b; node->index[0] = s->nodes_count;
            Task c = { t.start, split + 1, s->nodes_count, &node->aabb[0] };
            stack[sp++] = c; ++s->nodes_count;
        }
        if ( split == t.end - 2 ) {
            node->type[1] = vol[t.end - 1].type; node->aabb[1] = vol[t.end - 1].aabb; node->index[1] = ( int32_t ) vol[t.end - 1].index;
        } else {
            node->type[1] = -1;
            TerraAABB b = empty_aabb();
            for ( int i = split + 1; i < t.end; ++i ) fit_aabb ( &b, &vol[i].aabb );
            node->aabb[1] = b; node->index[1] = s->nodes_count;
            Task c = { split + 1, t.end, s->nodes_count, &node->aabb[1] };
            stack[sp++] = c; ++s->nodes_count;
        }
    }
    free ( stack ); free ( la ); free ( ra ); free ( vol );
    /* worst-case traversal stack: run traverse()'s push/pop order with every box test passing */
    {
        int* st = ( int* ) malloc ( sizeof ( int ) * ( size_t ) ( n + 2 ) );
        int top = 0, mx = 1;
        st[top++] = 0;
        while ( top > 0 ) {
            const OrcBVHNode* nd = &s->nodes[st[--top]];
            for ( int i = 0; i < 2; ++i ) if ( nd->type[i] == -1 ) { st[top++] = nd->index[i]; if ( top > mx ) mx = top; }
        }
        s->max_stack = mx;
        free ( st );
    }
}

int orc_bvh_node_count ( HTerraScene h ) { return ( ( OrcScene* ) h )->nodes_count; }
const void* orc_bvh_nodes ( HTerraScene h ) { return ( ( OrcScene* ) h )->nodes; }
int orc_bvh_max_stack ( HTerraScene h ) { return ( ( OrcScene* ) h )->max_stack; }

/* ------------------------------------------------------------------------- */
/* A5: traversal (reference src/TerraBVH.c:250-310)                            */
/* ------------------------------------------------------------------------- */
static bool bvh_traverse ( const OrcScene* s, const OrcRay* r, const OrcRayState* st, v3* point_out, int* obj_out, int* tri_out ) {
    int stack_local[256];
    int* stack = stack_local;
    if ( s->max_stack + 2 > 256 ) stack = ( int* ) malloc ( sizeof ( int ) * ( size_t ) ( s->max_stack + 2 ) );
    int top = 0; stack[top++] = 0;
    float min_d = FLT_MAX;
    v3 min_p = v3_set ( FLT_MAX, FLT_MAX, FLT_MAX );
    bool found = false;
    while ( top > 0 ) {
        const OrcBVHNode* nd = &s->nodes[stack[--top]];
        ++tls.c.nodes;
        for ( int i = 0; i < 2; ++i ) {
            if ( nd->type[i] == -1 ) {
                ++tls.c.box_tests;
                if ( ray_aabb ( r, &nd->aabb[i], NULL, NULL ) ) stack[top++] = nd->index[i];
            } else if ( nd->type[i] == 1 ) {
                int model = nd->index[i] & 0xff, tri = nd->index[i] >> 8;
                OrcTriHit h;
                ++tls.c.tri_tests;
                if ( watertight ( r, st, &s->objects[model].triangles[tri], &h ) && h.depth < min_d ) {
                    min_d = h.depth; min_p = h.point; *obj_out = model; *tri_out = tri; found = true;
                }
            }
        }
    }
    if ( stack != stack_local ) free ( stack );
    *point_out = min_p;
    return found;
}
int orc_bvh_traverse ( HTerraScene h, const TerraFloat3* o, const TerraFloat3* d, TerraFloat3* point, uint32_t* prim ) {
    OrcScene* s = ( OrcScene* ) h;
    OrcRay r = make_ray ( *o, *d ); OrcRayState st; ray_state_init ( &r, &st );
    int obj = 0, tri = 0;
    bool f = bvh_traverse ( s, &r, &st, point, &obj, &tri );
    *prim = f ? ( ( uint32_t ) obj & 0xffu ) | ( ( uint32_t ) tri << 8 ) : 0u;
    return f ? 1 : 0;
}

/* ------------------------------------------------------------------------- */
/* A8: surface init (reference src/Terra.c:1726-1764; basis TerraMath.inl:251-272) */
/* ------------------------------------------------------------------------- */
void orc_surface_init ( TerraShadingSurface* sf, const TerraTriangle* t, const TerraMaterial* m, const TerraTriangleProperties* pr, const TerraFloat3* point ) {
    v3 e0 = v3_sub ( t->b, t->a ), e1 = v3_sub ( t->c, t->a ), p = v3_sub ( *point, t->a );
    float d00 = v3_dot ( e0, e0 ), d11 = v3_dot ( e1, e1 ), d01 = v3_dot ( e0, e1 );
    float dp0 = v3_dot ( p, e0 ), dp1 = v3_dot ( p, e1 );
    float div = d00 * d11 - d01 * d01;
    float u = ( d11 * dp0 - d01 * dp1 ) / div;
    float v = ( d00 * dp1 - d01 * dp0 ) / div;
    float w = 1 - u - v;
    v3 n = v3_add ( v3_add ( v3_scale ( pr->normal_c, v ), v3_scale ( pr->normal_b, u ) ), v3_scale ( pr->normal_a, w ) );
    sf->normal = v3_norm ( n );
    TerraFloat2 tc;
    tc.x = ( pr->texcoord_c.x * v + pr->texcoord_b.x * u ) + pr->texcoord_a.x * w;
    tc.y = ( pr->texcoord_c.y * v + pr->texcoord_b.y * u ) + pr->texcoord_a.y * w;
    for ( int i = 0; i < TERRA_MATERIAL_MAX_ATTRIBUTES; ++i ) sf->attributes[i] = v3_set ( 0, 0, 0 );
    for ( size_t i = 0; i < m->attributes_count && i < TERRA_MATERIAL_MAX_ATTRIBUTES; ++i ) sf->attributes[i] = attribute_eval ( &m->attributes[i], &tc, point );
    sf->emissive = attribute_eval ( &m->emissive, &tc, point );
    sf->ior = m->ior;
    sf->transform = terra_f4x4_basis ( &sf->normal );
}

/* A4: raycast (reference src/Terra.c:1623-1657). Returns object index or -1. */
static int scene_raycast ( const OrcScene* s, const OrcRay* in, TerraShadingSurface* sf, v3* point, int* tri_out ) {
    OrcRay r = *in;
    r.origin = v3_add ( r.origin, v3_scale ( r.direction, 0.001f ) );
    OrcRayState st; ray_state_init ( &r, &st );
    int obj = -1, tri = 0;
    ++tls.c.rays;
    if ( !bvh_traverse ( s, &r, &st, point, &obj, &tri ) ) return -1;
    const TerraObject* o = &s->objects[obj];
    if ( tri_out ) *tri_out = tri;
    ++tls.c.hits; tls.c.attr_fetches += o->material.attributes_count + 1;
    orc_surface_init ( sf, &o->triangles[tri], &o->material, &o->properties[tri], point );
    return obj;
}
int orc_raycast ( HTerraScene h, const TerraFloat3* o, const TerraFloat3* d, TerraShadingSurface* sf, TerraFloat3* point, int* triangle ) {
    OrcRay r = make_ray ( *o, *d );
    return scene_raycast ( ( OrcScene* ) h, &r, sf, point, triangle );
}

/* ------------------------------------------------------------------------- */
/* A12/A13: BSDF presets (reference src/TerraPresets.c:34-146)                 */
/* ------------------------------------------------------------------------- */
static TerraFloat3 diffuse_sample ( const TerraShadingSurface* sf, float e1, float e2, float e3, const TerraFloat3* wo ) {
    ( void ) e3; ( void ) wo;
    float r = sqrtf ( e1 );
    float theta = 2 * terra_PI * e2;
    float x = r * orc_math_cosf ( theta );
    float z = r * orc_math_sinf ( theta );
    v3 wi = v3_set ( x, sqrtf ( sel_max ( 0.f, 1 - e1 ) ), z );
    return v3_norm ( m3_apply ( &sf->transform, wi ) );
}
static float diffuse_pdf ( const TerraShadingSurface* sf, const TerraFloat3* wi, const TerraFloat3* wo ) {
    ( void ) wo;
    return sel_max ( 0.f, v3_dot ( sf->normal, *wi ) ) / terra_PI;
}
static TerraFloat3 diffuse_eval ( const TerraShadingSurface* sf, const TerraFloat3* wi, const TerraFloat3* wo ) {
    ( void ) wi; ( void ) wo;
    return v3_scale ( sf->attributes[TERRA_DIFFUSE_ALBEDO], 1. / terra_PI );    /* double reciprocal, rounded at the call */
}
void orc_bsdf_diffuse_init ( TerraBSDF* b ) { b->sample = diffuse_sample; b->pdf = diffuse_pdf; b->eval = diffuse_eval; }

static void phong_kd_ks ( const TerraShadingSurface* sf, float* kd, float* ks ) {
    const v3 al = sf->attributes[TERRA_PHONG_ALBEDO], sp = sf->attributes[TERRA_PHONG_SPECULAR_COLOR];
    float diffuse = sel_max ( al.x + al.y + al.z, terra_Epsilon );
    float specular = sp.x + sp.y + sp.z;
    if ( specular > diffuse ) { *kd = 0.5f * diffuse / specular; *ks = 1.f - *kd; }
    else { *ks = 0.5f * specular / diffuse; *kd = 1.f - *ks; }
}
static v3 reflect_about_normal ( const TerraShadingSurface* sf, v3 wo ) {
    v3 wr = v3_scale ( sf->normal, 2.f * v3_dot ( wo, sf->normal ) );
    return v3_sub ( wr, wo );
}
static TerraFloat3 phong_sample ( const TerraShadingSurface* sf, float e1, float e2, float e3, const TerraFloat3* wo ) {
    float kd, ks; phong_kd_ks ( sf, &kd, &ks );
    TerraFloat3* pick = ( TerraFloat3* ) &sf->attributes[TERRA_PHONG_SAMPLE_PICK];   /* written through const, as the reference does */
    if ( e3 < kd ) { pick->x = 1.f; return diffuse_sample ( sf, e1, e2, e3, wo ); }
    pick->x = -1.f;
    v3 wr = reflect_about_normal ( sf, *wo );
    TerraFloat4x4 basis = terra_f4x4_basis ( &wr );
    float phi = 2 * terra_PI * e1;
    float theta = orc_math_acosf ( orc_math_powf ( 1.f - e2, 1.f / ( sf->attributes[TERRA_PHONG_SPECULAR_INTENSITY].x + 1 ) ) );
    float sin_theta = orc_math_sinf ( theta );
    v3 wi = v3_set ( sin_theta * orc_math_cosf ( phi ), orc_math_cosf ( theta ), sin_theta * orc_math_sinf ( phi ) );
    return v3_norm ( m3_apply ( &basis, wi ) );
}
static float phong_pdf ( const TerraShadingSurface* sf, const TerraFloat3* wi, const TerraFloat3* wo ) {
    float pick = sf->attributes[TERRA_PHONG_SAMPLE_PICK].x;
    if ( pick == 1.f ) return diffuse_pdf ( sf, wi, wo );
    /* pick == -1 (anything else asserts in the reference) */
    v3 wr = reflect_about_normal ( sf, *wo );
    float cos_alpha = v3_dot ( *wi, wr );
    float n = sf->attributes[TERRA_PHONG_SPECULAR_INTENSITY].x;
    return ( n + 1 ) / ( 2 * terra_PI ) * orc_math_powf ( cos_alpha, n );
}
static TerraFloat3 phong_eval ( const TerraShadingSurface* sf, const TerraFloat3* wi, const TerraFloat3* wo ) {
    float kd, ks; phong_kd_ks ( sf, &kd, &ks );
    float n = sf->attributes[TERRA_PHONG_SPECULAR_INTENSITY].x;
    v3 diffuse_term = v3_scale ( sf->attributes[TERRA_PHONG_ALBEDO], kd * 1.f / terra_PI );
    v3 wr = reflect_about_normal ( sf, *wo );
    float cos_alpha = v3_dot ( *wi, wr );
    float cos_n_alpha = orc_math_powf ( cos_alpha, n );
    v3 specular_term = v3_scale ( sf->attributes[TERRA_PHONG_SPECULAR_COLOR], ks * cos_n_alpha * ( n + 2 ) / ( 2 * terra_PI ) );
    return v3_add ( diffuse_term, specular_term );
}
void orc_bsdf_phong_init ( TerraBSDF* b ) { b->sample = phong_sample; b->pdf = phong_pdf; b->eval = phong_eval; }

/* ------------------------------------------------------------------------- */
/* A14: GGX conductor and dielectric glass -- NO LIVE REFERENCE ("parity unpinned").     */
/* Defined by this repo (include/TerraPresets.h) from the building blocks of the         */
/* reference's dead code (src/TerraPresets.c:303-320 D and G1, :333-343 half-vector       */
/* sampling, :399-449 Snell / TIR / Schlick). Only + - * / sqrt and the pinned sinf/cosf. */
/* ------------------------------------------------------------------------- */
static float ggx_D ( float NoH, float alpha2 ) {                 /* :316-320 */
    if ( NoH <= 0.f ) return 0.f;
    float NoH2 = NoH * NoH;
    float den = NoH2 * alpha2 + ( 1 - NoH2 );
    return alpha2 / ( terra_PI * den * den );
}
static float ggx_G1 ( v3 v, v3 n, v3 h, float alpha2 ) {         /* :307-314 */
    /* Smith G1 for GGX (Walter et al. 2007, eq. 34). The dead code takes the tangent of the angle to
       the HALF vector (:311-312), which over-weights grazing lobes (mean path weight > 1); the angle to
       the NORMAL is used here. */
    float VoH = v3_dot ( v, h ), VoN = v3_dot ( v, n );
    if ( VoH / VoN <= 0.f ) return 0.f;
    float VoN2 = VoN * VoN;
    float tan2 = ( 1.f - VoN2 ) / VoN2;
    return 2.f / ( sqrtf ( 1 + alpha2 * tan2 ) + 1 );
}
static TerraFloat3 ggx_sample ( const TerraShadingSurface* sf, float e1, float e2, float e3, const TerraFloat3* wo ) {
    ( void ) e3;
    float alpha = sf->attributes[TERRA_GGX_ROUGHNESS].x;
    float t2 = alpha * alpha * e1 / ( 1.f - e1 );                /* tan^2(theta_h), :337 without the atan */
    float cos_t = 1.f / sqrtf ( 1.f + t2 );
    float sin_t = sqrtf ( sel_max ( 0.f, 1.f - cos_t * cos_t ) );
    float phi = 2 * terra_PI * e2;
    v3 h = v3_set ( sin_t * orc_math_cosf ( phi ), cos_t, sin_t * orc_math_sinf ( phi ) );
    h = v3_norm ( m3_apply ( &sf->transform, h ) );
    float HoV = sel_max ( 0.f, v3_dot ( h, *wo ) );
    return v3_sub ( v3_scale ( h, 2 * HoV ), *wo );              /* :345-346 */
}
static float ggx_pdf ( const TerraShadingSurface* sf, const TerraFloat3* wi, const TerraFloat3* wo ) {
    float alpha = sf->attributes[TERRA_GGX_ROUGHNESS].x;
    v3 h = v3_norm ( v3_add ( *wi, *wo ) );
    float NoH = v3_dot ( sf->normal, h ), HoV = v3_dot ( h, *wo );
    if ( HoV <= 0.f ) return 0.f;
    return ggx_D ( NoH, alpha * alpha ) * NoH / ( 4.f * HoV );
}
static TerraFloat3 ggx_eval ( const TerraShadingSurface* sf, const TerraFloat3* wi, const TerraFloat3* wo ) {
    float alpha = sf->attributes[TERRA_GGX_ROUGHNESS].x, alpha2 = alpha * alpha;
    float NoL = v3_dot ( sf->normal, *wi ), NoV = v3_dot ( sf->normal, *wo );
    if ( NoL <= 0.f || NoV <= 0.f ) return v3_set ( 0, 0, 0 );
    v3 h = v3_norm ( v3_add ( *wi, *wo ) );
    float NoH = v3_dot ( sf->normal, h ), HoV = sel_max ( 0.f, v3_dot ( h, *wo ) );
    float m = 1.f - HoV, m2 = m * m, w5 = m2 * m2 * m;          /* Schlick weight */
    v3 F0 = sf->attributes[TERRA_GGX_F0];
    v3 F = v3_set ( F0.x + ( 1.f - F0.x ) * w5, F0.y + ( 1.f - F0.y ) * w5, F0.z + ( 1.f - F0.z ) * w5 );
    float G = ggx_G1 ( *wo, sf->normal, h, alpha2 ) * ggx_G1 ( *wi, sf->normal, h, alpha2 );
    float k = G * ggx_D ( NoH, alpha2 ) / ( 4.f * NoL * NoV );
    return v3_scale ( F, k );
}
void orc_bsdf_ggx_init ( TerraBSDF* b ) { b->sample = ggx_sample; b->pdf = ggx_pdf; b->eval = ggx_eval; }

static TerraFloat3 glass_sample ( const TerraShadingSurface* sf, float e1, float e2, float e3, const TerraFloat3* wo ) {
    ( void ) e1; ( void ) e2;
    TerraFloat3* sdir = ( TerraFloat3* ) &sf->attributes[TERRA_GLASS_SAMPLE_DIR];
    TerraFloat3* sprob = ( TerraFloat3* ) &sf->attributes[TERRA_GLASS_SAMPLE_PROB];
    v3 normal = sf->normal, incident = v3_neg ( *wo );
    float n1, n2, cos_i = v3_dot ( normal, incident );
    if ( cos_i > 0.f ) { n1 = sf->ior; n2 = terra_ior_air; normal = v3_neg ( normal ); }      /* leaving the medium, :409-413 */
    else { n1 = terra_ior_air; n2 = sf->ior; cos_i = -cos_i; }
    v3 refl = v3_sub ( incident, v3_scale ( normal, 2 * v3_dot ( normal, incident ) ) );
    float nni = n1 / n2;
    float cos_t2 = 1.f - nni * nni * ( 1.f - cos_i * cos_i );
    v3 dir; float prob;
    if ( cos_t2 < 0.f ) { dir = refl; prob = 1.f; }                                      /* total internal reflection, :424-427 */
    else {
        float cos_t = sqrtf ( cos_t2 );
        float t = 1.f - ( n1 <= n2 ? cos_i : cos_t );
        float R0 = ( n1 - n2 ) / ( n1 + n2 ); R0 *= R0;
        float R = R0 + ( 1 - R0 ) * ( t * t * t * t * t );
        if ( e3 < R ) { dir = refl; prob = R; }
        else {
            v3 tv = v3_scale ( normal, nni * cos_i - cos_t ), tn = v3_scale ( incident, nni );
            dir = v3_norm ( v3_add ( tv, tn ) ); prob = 1 - R;
        }
    }
    *sdir = dir; sprob->x = prob;
    return dir;
}
static bool glass_is_sampled ( const TerraShadingSurface* sf, const TerraFloat3* wi ) {
    const TerraFloat3* d = &sf->attributes[TERRA_GLASS_SAMPLE_DIR];
    return sf->attributes[TERRA_GLASS_SAMPLE_PROB].x > 0.f && wi->x == d->x && wi->y == d->y && wi->z == d->z;
}
static float glass_pdf ( const TerraShadingSurface* sf, const TerraFloat3* wi, const TerraFloat3* wo ) {
    ( void ) wo;
    return glass_is_sampled ( sf, wi ) ? sf->attributes[TERRA_GLASS_SAMPLE_PROB].x : 0.f;     /* a delta lobe: zero for any other direction */
}
static TerraFloat3 glass_eval ( const TerraShadingSurface* sf, const TerraFloat3* wi, const TerraFloat3* wo ) {
    ( void ) wo;
    if ( !glass_is_sampled ( sf, wi ) ) return v3_set ( 0, 0, 0 );
    /* terra_trace multiplies eval/pdf by the SIGNED dot(n, wi): divide it out so the path weight is the tint */
    float k = sf->attributes[TERRA_GLASS_SAMPLE_PROB].x / v3_dot ( sf->normal, *wi );
    return v3_scale ( sf->attributes[TERRA_GLASS_TINT], k );
}
void orc_bsdf_glass_init ( TerraBSDF* b ) { b->sample = glass_sample; b->pdf = glass_pdf; b->eval = glass_eval; }

/* ------------------------------------------------------------------------- */
/* lights (reference src/Terra.c:1592-1621, :1662-1697, :1833-1838)            */
/* ------------------------------------------------------------------------- */
static float triangle_area ( const TerraTriangle* t ) {
    v3 c = v3_cross ( v3_sub ( t->b, t->a ), v3_sub ( t->c, t->a ) );
    return v3_len ( c ) / 2;
}
static const OrcLight* pick_light ( const OrcScene* s, float e, float* pdf ) {
    double x = e * ( double ) s->lights_pop;
    size_t i = x < 0 ? 0 : ( size_t ) x;
    *pdf = 1.f / s->lights_triangles_count;
    return &s->lights[i];
}
static size_t light_pick_triangle ( const OrcScene* s, const OrcLight* l, float e, float* pdf ) {
    size_t n = s->objects[l->object].triangles_count;
    size_t i = ( size_t ) ( e * n );
    if ( i >= n ) i = n - 1;       /* e*n can round up to n; the reference would index past the end */
    *pdf = 1.f / n;
    return i;
}
static void light_sample_triangle ( const OrcScene* s, const OrcLight* l, size_t tri, float e1, float e2, v3* pos, v3* norm ) {
    const TerraTriangle* t = &s->objects[l->object].triangles[tri];
    const TerraTriangleProperties* pr = &s->objects[l->object].properties[tri];
    float sq = sqrtf ( e1 );
    float a = 1 - sq, b = e2 * sq, c = 1 - a - b;
    *pos = v3_add ( v3_add ( v3_scale ( t->a, a ), v3_scale ( t->b, b ) ), v3_scale ( t->c, c ) );
    v3 n = v3_add ( v3_add ( v3_scale ( pr->normal_a, a ), v3_scale ( pr->normal_b, b ) ), v3_scale ( pr->normal_c, c ) );
    *norm = v3_norm ( n );
}

/* ------------------------------------------------------------------------- */
/* A9: integrators (reference src/Terra.c:1099-1587)                           */
/* ------------------------------------------------------------------------- */

/* ------------------------------------------------------------------------- */
/* Environment importance sampling (extension, UNPINNED: nothing  */
/* in the reference calls TerraDistribution2D, src/Terra.c:812-846; this wiring */
/* is this repo's definition). Table: one value per texel of the lat-long map,  */
/* luminance x sin(theta of the texel row's centre), laid out row by row and    */
/* initialised exactly as terra_distribution_2d_init does (running float sums). */
/* ------------------------------------------------------------------------- */
static void env_table_free ( OrcScene* s ) {
    free ( s->env_f ); free ( s->env_cdf ); free ( s->env_row_f ); free ( s->env_row_cdf );
    s->env_f = s->env_cdf = s->env_row_f = s->env_row_cdf = NULL; s->env_nx = s->env_ny = 0; s->env_integral = 0;
}
static void env_table_build ( OrcScene* s ) {
    env_table_free ( s );
    const TerraAttribute* env = &s->opts.environment_map;
    if ( !s->env_sampling || !s->env_lighting || env->state == NULL || env->eval != orc_texture_sample_latlong ) return;
    TerraTexture* t = ( TerraTexture* ) env->state;
    const size_t nx = t->width, ny = t->height;
    if ( !t->pixels || nx == 0 || ny == 0 || t->components < 3 ) return;
    s->env_f = ( float* ) malloc ( sizeof ( float ) * nx * ny ); s->env_cdf = ( float* ) malloc ( sizeof ( float ) * nx * ny );
    s->env_row_f = ( float* ) malloc ( sizeof ( float ) * ny ); s->env_row_cdf = ( float* ) malloc ( sizeof ( float ) * ny );
    for ( size_t y = 0; y < ny; ++y ) {
        const float sin_row = sinf ( ( ( float ) y + 0.5f ) / ( float ) ny * terra_PI );          /* libm on the host, in the product too */
        for ( size_t x = 0; x < nx; ++x ) {
            v3 c = orc_texture_read ( t, x, y );
            float lum = 0.2126f * c.x; lum += 0.7152f * c.y; lum += 0.0722f * c.z;
            s->env_f[y * nx + x] = lum * sin_row;
        }
        s->env_row_f[y] = dist1d_init ( s->env_f + y * nx, nx, s->env_cdf + y * nx );
    }
    s->env_integral = dist1d_init ( s->env_row_f, ny, s->env_row_cdf );
    s->env_nx = nx; s->env_ny = ny;
}
/* One environment sample at a shaded point (Direct and Direct+MIS, after their own light samples): two draws of stream B pick a texel through the table
   (e1: the row, e2: the column inside it); the direction is the inverse of the lookup's mapping (src/Terra.c:468-477: theta = v terra_PI, phi = u 2 terra_PI -
   terra_PI); density over the sphere = texel probability x texels / (2 terra_PI^2 sin theta); the sample counts when the direction is in the upper
   hemisphere of the shading normal and its shadow ray leaves the scene; radiance = the chosen texel. Returns the term before the path throughput. */
static v3 environment_light_sample ( const OrcScene* s, const TerraObject* obj, const TerraShadingSurface* sf, v3 p, v3 wo ) {
    const v3 zero = v3_set ( 0, 0, 0 );
    float e1 = randf(), e2 = randf();
    float p_row = 0.f, p_col = 0.f; uint32_t row = 0, col = 0;
    float sv = dist1d_sample ( s->env_row_f, s->env_row_cdf, s->env_ny, s->env_integral, e1, &p_row, &row );
    if ( sv == FLT_MAX ) return zero;
    float su = dist1d_sample ( s->env_f + s->env_nx * row, s->env_cdf + s->env_nx * row, s->env_nx, s->env_row_f[row], e2, &p_col, &col );
    if ( su == FLT_MAX ) return zero;
    float theta = sv * terra_PI, phi = su * ( 2 * terra_PI ) - terra_PI;
    float st = orc_math_sinf ( theta ), ct = orc_math_cosf ( theta ), sp = orc_math_sinf ( phi ), cp = orc_math_cosf ( phi );
    if ( ! ( st > 0 ) ) return zero;
    v3 wi = v3_set ( st * cp, ct, st * sp );
    float cosine = v3_dot ( wi, sf->normal );
    if ( ! ( cosine > 0 ) ) return zero;
    float pdf = ( p_row * p_col ) * ( ( float ) s->env_nx * ( float ) s->env_ny ) / ( 2 * terra_PI * terra_PI * st );
    if ( ! ( pdf > 0 ) ) return zero;
    TerraShadingSurface lsf; v3 ip;
    OrcRay r = surface_ray ( sf, p, wi, 1 );
    if ( scene_raycast ( s, &r, &lsf, &ip, NULL ) >= 0 ) return zero;
    v3 L = orc_texture_read ( ( TerraTexture* ) s->opts.environment_map.state, col, row );
    v3 f = obj->material.bsdf.eval ( sf, &wi, &wo );
    return v3_scale ( v3_mul ( L, f ), cosine / pdf );
}
static inline bool env_sampling_active ( const OrcScene* s ) { return s->env_f != NULL; }

static v3 integrate_simple ( v3 throughput, const TerraShadingSurface* sf, v3 wo ) {
    if ( v3_dot ( wo, sf->normal ) > 0 ) return v3_mul ( sf->emissive, throughput );
    return v3_set ( 0, 0, 0 );
}

typedef struct { const OrcLight* light; float pick_pdf; size_t tri; v3 pos, norm; } LightSample;
static LightSample draw_light_sample ( const OrcScene* s ) {
    LightSample ls;
    { float e = randf() - terra_Epsilon; ls.light = pick_light ( s, e, &ls.pick_pdf ); }     /* double subtract, rounded to float */
    { float e = randf(); float tp; ls.tri = light_pick_triangle ( s, ls.light, e, &tp ); }
    { float e1 = randf(); float e2 = randf(); light_sample_triangle ( s, ls.light, ls.tri, e1, e2, &ls.pos, &ls.norm ); }
    return ls;
}

static v3 integrate_direct ( const OrcScene* s, const TerraObject* obj, const TerraShadingSurface* sf, v3 p, v3 wo, v3 throughput, size_t bounce ) {
    v3 Lo = v3_set ( 0, 0, 0 );
    if ( bounce == 0 && v3_dot ( wo, sf->normal ) > 0 ) Lo = v3_add ( Lo, sf->emissive );
    LightSample ls = draw_light_sample ( s );
    v3 p_to_light = v3_sub ( ls.pos, p );
    v3 wi = v3_norm ( p_to_light );
    TerraShadingSurface lsf; v3 ip; int ltri = 0;
    OrcRay r = surface_ray ( sf, p, wi, 1 );
    int hit = scene_raycast ( s, &r, &lsf, &ip, &ltri );
    if ( hit == ls.light->object && ( size_t ) ltri == ls.tri ) {
        float cosv = v3_dot ( v3_neg ( wi ), ls.norm );
        if ( cosv > 0 ) {
            v3 f = obj->material.bsdf.eval ( sf, &wi, &wo );
            float pdf = v3_dot ( p_to_light, p_to_light ) / fabsf ( cosv * ls.light->triangle_area[ltri] );
            v3 Ld = v3_mul ( lsf.emissive, f );
            Ld = v3_scale ( Ld, v3_dot ( wi, sf->normal ) / ( pdf * ls.pick_pdf ) );
            Lo = v3_add ( Lo, Ld );
        }
    }
    if ( env_sampling_active ( s ) ) Lo = v3_add ( Lo, environment_light_sample ( s, obj, sf, p, wo ) );
    return v3_mul ( Lo, throughput );
}

/* shared by DirectMis and DebugMisWeights: they differ only in what is accumulated */
static v3 integrate_mis ( const OrcScene* s, const TerraObject* obj, const TerraShadingSurface* sf, v3 p, v3 wo, v3 throughput, size_t bounce, bool debug_weights ) {
    v3 Lo = v3_set ( 0, 0, 0 );
    if ( debug_weights ) { if ( bounce != 0 ) return Lo; }
    else if ( bounce == 0 && v3_dot ( wo, sf->normal ) > 0 ) Lo = v3_add ( Lo, sf->emissive );
    v3 bsdf_dir;
    { float e1 = randf(); float e2 = randf(); float e3 = randf(); bsdf_dir = obj->material.bsdf.sample ( sf, e1, e2, e3, &wo ); }
    LightSample ls = draw_light_sample ( s );
    {   /* light-sampling strategy */
        v3 p_to_light = v3_sub ( ls.pos, p );
        v3 wi = v3_norm ( p_to_light );
        TerraShadingSurface lsf; v3 ip; int ltri = 0;
        OrcRay r = surface_ray ( sf, p, wi, 1 );
        int hit = scene_raycast ( s, &r, &lsf, &ip, &ltri );
        if ( hit == ls.light->object && ( size_t ) ltri == ls.tri ) {
            float cosv = v3_dot ( ls.norm, v3_neg ( wi ) );
            if ( cosv > 0 ) {
                float bsdf_pdf = obj->material.bsdf.pdf ( sf, &wi, &wo );
                float light_pdf = v3_dot ( p_to_light, p_to_light ) / fabsf ( cosv * ls.light->triangle_area[ltri] );
                if ( debug_weights ) {
                    float weight = ( bsdf_pdf * bsdf_pdf ) / ( light_pdf * light_pdf + bsdf_pdf * bsdf_pdf );
                    Lo = v3_add ( Lo, v3_set ( 0, 0, weight ) );
                } else {
                    float weight = ( light_pdf * light_pdf ) / ( light_pdf * light_pdf + bsdf_pdf * bsdf_pdf );
                    if ( light_pdf != 0 ) {
                        v3 f = obj->material.bsdf.eval ( sf, &wi, &wo );
                        v3 L = v3_mul ( lsf.emissive, f );
                        L = v3_scale ( L, v3_dot ( wi, sf->normal ) * weight / ( light_pdf * ls.pick_pdf ) );
                        Lo = v3_add ( Lo, L );
                    }
                }
            }
        }
    }
    {   /* BSDF-sampling strategy */
        v3 wi = bsdf_dir;
        v3 f = obj->material.bsdf.eval ( sf, &wi, &wo );
        float bsdf_pdf = obj->material.bsdf.pdf ( sf, &wi, &wo );
        v3 light_wo = v3_neg ( wi );
        TerraShadingSurface lsf; v3 ip; int ltri = 0;
        OrcRay r = surface_ray ( sf, p, wi, 1 );
        int hit = scene_raycast ( s, &r, &lsf, &ip, &ltri );
        if ( hit == ls.light->object ) {          /* hit is never -1 here: light->object >= 0 */
            float NoW = v3_dot ( lsf.normal, light_wo );
            if ( NoW > 0 ) {
                v3 dlt = v3_sub ( p, ip );
                float dist = v3_dot ( dlt, dlt );
                float light_pdf = dist / ( NoW * triangle_area ( &s->objects[hit].triangles[ltri] ) );
                float weight = ( bsdf_pdf * bsdf_pdf ) / ( light_pdf * light_pdf + bsdf_pdf * bsdf_pdf );
                if ( debug_weights ) {
                    Lo = v3_add ( Lo, v3_set ( weight, 0, 0 ) );
                } else if ( bsdf_pdf != 0 ) {
                    v3 L = v3_mul ( lsf.emissive, f );
                    L = v3_scale ( L, v3_dot ( wi, sf->normal ) * weight / bsdf_pdf );
                    Lo = v3_add ( Lo, L );
                }
            }
        }
    }
    if ( !debug_weights && env_sampling_active ( s ) ) Lo = v3_add ( Lo, environment_light_sample ( s, obj, sf, p, wo ) );
    return v3_mul ( Lo, throughput );
}

static v3 integrate_debug_normals ( const TerraShadingSurface* sf, size_t bounce ) {       /* reference src/Terra.c:1159-1197 */
    if ( bounce != 0 ) return v3_set ( 0, 0, 0 );
    v3 n = sf->normal;
    v3 pp = v3_set ( sel_min ( n.x > 0 ? n.x : 0.f, 1.f ), sel_min ( n.y > 0 ? n.y : 0.f, 1.f ), sel_min ( n.z > 0 ? n.z : 0.f, 1.f ) );
    v3 nn = v3_set ( sel_min ( n.x > -1 ? n.x : -1.f, 0.f ), sel_min ( n.y > -1 ? n.y : -1.f, 0.f ), sel_min ( n.z > -1 ? n.z : -1.f, 0.f ) );
    nn = v3_scale ( nn, -1.f );
    v3 c = v3_set ( 0, 0, 0 );
    c = v3_add ( c, v3_scale ( v3_set ( 1, 0, 0 ), pp.x ) );
    c = v3_add ( c, v3_scale ( v3_set ( 0, 1, 0 ), pp.y ) );
    c = v3_add ( c, v3_scale ( v3_set ( 0, 0, 1 ), pp.z ) );
    c = v3_add ( c, v3_scale ( v3_set ( 0, 1, 1 ), nn.x ) );
    c = v3_add ( c, v3_scale ( v3_set ( 1, 0, 1 ), nn.y ) );
    c = v3_add ( c, v3_scale ( v3_set ( 1, 1, 0 ), nn.z ) );
    return c;
}

static v3 integrate ( const OrcScene* s, const OrcRay* ray, const TerraObject* obj, const TerraShadingSurface* sf, v3 p, v3 wo, v3 throughput, size_t bounce ) {
    switch ( s->opts.integrator ) {
        case kTerraIntegratorSimple:    return integrate_simple ( throughput, sf, wo );
        case kTerraIntegratorDirect:    return integrate_direct ( s, obj, sf, p, wo, throughput, bounce );
        case kTerraIntegratorDirectMis: return integrate_mis ( s, obj, sf, p, wo, throughput, bounce, false );
        case kTerraIntegratorDebugMono: return bounce != 0 ? v3_set ( 0, 0, 0 ) : v3_set ( 1, 1, 1 );
        case kTerraIntegratorDebugDepth: {
            if ( bounce != 0 ) return v3_set ( 0, 0, 0 );
            float d = v3_len ( v3_sub ( ray->origin, p ) ) / 500.f;
            return v3_set ( d, d, d );
        }
        case kTerraIntegratorDebugNormals:    return integrate_debug_normals ( sf, bounce );
        case kTerraIntegratorDebugMisWeights: return integrate_mis ( s, obj, sf, p, wo, throughput, bounce, true );
        default: return v3_set ( 0, 0, 0 );
    }
}

/* ------------------------------------------------------------------------- */
/* A3: terra_trace (reference src/Terra.c:1039-1097)                           */
/* ------------------------------------------------------------------------- */
/* first_pair (extension, NULL = the reference's behaviour): the pixel sampler's pair for this camera sample; it replaces the first two variates handed to
   bsdf.sample at bounce 0 (orc_set_sampler_integration). Stream B is consumed exactly as without it. */
static v3 trace ( const OrcScene* s, const OrcRay* primary, const float* first_pair ) {
    v3 Lo = v3_set ( 0, 0, 0 ), throughput = v3_set ( 1, 1, 1 );
    OrcRay ray = *primary;
    for ( size_t bounce = 0; bounce <= s->opts.bounces; ++bounce ) {
        TerraShadingSurface sf; v3 p;
        int oi = scene_raycast ( s, &ray, &sf, &p, NULL );
        if ( oi < 0 ) {              /* the reference scales the throughput by the environment and then discards it (:1053-1058) */
            /* (with environment sampling in a light integrator the environment reaches the path through the samples taken at its hits: only the camera ray adds it here) */
            const bool env_by_samples = env_sampling_active ( s ) && ( s->opts.integrator == kTerraIntegratorDirect || s->opts.integrator == kTerraIntegratorDirectMis );
            if ( s->env_lighting && ( bounce == 0 || !env_by_samples ) ) { /* extension: what the commented-out "Lo += throughput" (:1056) would add */
                v3 env = attribute_eval ( &s->opts.environment_map, &ray.direction, &p );
                throughput = v3_mul ( throughput, env );
                Lo = v3_add ( Lo, throughput );
            }
            break;
        }
        const TerraObject* obj = &s->objects[oi];
        v3 wo = v3_neg ( ray.direction );
        Lo = v3_add ( Lo, integrate ( s, &ray, obj, &sf, p, wo, throughput, bounce ) );
        float e0 = randf(), e1 = randf(), e2 = randf();
        if ( bounce == 0 && first_pair ) { e0 = first_pair[0]; e1 = first_pair[1]; }
        v3 wi = obj->material.bsdf.sample ( &sf, e0, e1, e2, &wo );
        float pdf = sel_max ( obj->material.bsdf.pdf ( &sf, &wi, &wo ), terra_Epsilon );    /* eps -> 1e-4f at the call */
        v3 f = obj->material.bsdf.eval ( &sf, &wi, &wo );
        f = v3_scale ( f, 1.f / pdf );
        throughput = v3_mul ( throughput, f );
        throughput = v3_scale ( throughput, v3_dot ( sf.normal, wi ) );
        float pr = sel_max ( throughput.x, sel_max ( throughput.y, throughput.z ) );
        float e3 = randf();
        if ( e3 > pr ) break;
        throughput = v3_scale ( throughput, 1.f / ( pr + terra_Epsilon ) );                  /* double divide, rounded at the call */
        ray = surface_ray ( &sf, p, wi, 1.f );
    }
    return Lo;
}
TerraFloat3 orc_trace_one ( HTerraScene h, const TerraFloat3* o, const TerraFloat3* d, uint64_t stateB, uint64_t incB, uint32_t* rand_calls ) {
    tls.streamB.state = stateB; tls.streamB.inc = incB;
    uint64_t before = tls.c.rand_calls;
    OrcRay r = make_ray ( *o, *d );
    v3 L = trace ( ( OrcScene* ) h, &r, NULL );
    if ( rand_calls ) *rand_calls = ( uint32_t ) ( tls.c.rand_calls - before );
    return L;
}

/* ------------------------------------------------------------------------- */
/* A15: tonemap (reference src/Terra.c:578-627, :1815-1828)                    */
/* ------------------------------------------------------------------------- */
static v3 uncharted2 ( v3 x ) {
    const float A = 0.15f, B = 0.5f, C = 0.1f, D = 0.2f, E = 0.02f, F = 0.3f;
    v3 r;
    r.x = ( ( x.x * ( A * x.x + C * B ) + D * E ) / ( x.x * ( A * x.x + B ) + D * F ) ) - E / F;
    r.y = ( ( x.y * ( A * x.y + C * B ) + D * E ) / ( x.y * ( A * x.y + B ) + D * F ) ) - E / F;
    r.z = ( ( x.z * ( A * x.z + C * B ) + D * E ) / ( x.z * ( A * x.z + B ) + D * F ) ) - E / F;
    return r;
}
static v3 powv ( v3 c, float e ) { return v3_set ( orc_math_powf ( c.x, e ), orc_math_powf ( c.y, e ), orc_math_powf ( c.z, e ) ); }
TerraFloat3 orc_tonemap ( const TerraFloat3* in, int op, float gamma ) {
    v3 c = *in;
    switch ( op ) {
        case kTerraTonemappingOperatorLinear: c = powv ( c, 1.f / gamma ); break;
        case kTerraTonemappingOperatorReinhard:
            c.x = c.x / ( 1.f + c.x ); c.y = c.y / ( 1.f + c.y ); c.z = c.z / ( 1.f + c.z );
            c = powv ( c, 1.f / gamma ); break;
        case kTerraTonemappingOperatorFilmic: {
            v3 x = v3_set ( sel_max ( 0.f, c.x - 0.004f ), sel_max ( 0.f, c.y - 0.004f ), sel_max ( 0.f, c.z - 0.004f ) );
            c.x = ( x.x * ( 6.2f * x.x + 0.5f ) ) / ( x.x * ( 6.2f * x.x + 1.7f ) + 0.06f );
            c.y = ( x.y * ( 6.2f * x.y + 0.5f ) ) / ( x.y * ( 6.2f * x.y + 1.7f ) + 0.06f );
            c.x = ( x.z * ( 6.2f * x.z + 0.5f ) ) / ( x.z * ( 6.2f * x.z + 1.7f ) + 0.06f );   /* the .z result lands in .x (:604) */
            break;
        }
        case kTerraTonemappingOperatorUncharted2: {
            v3 ws = uncharted2 ( v3_set ( 11.2f, 11.2f, 11.2f ) );
            ws = v3_set ( 1.f / ws.x, 1.f / ws.y, 1.f / ws.z );
            v3 t = uncharted2 ( v3_scale ( c, 2.f ) );
            c = powv ( v3_mul ( t, ws ), 1.f / gamma );
            break;
        }
        default: break;
    }
    return c;
}

/* ------------------------------------------------------------------------- */
/* A1: terra_render (reference src/Terra.c:512-635) with per-pixel streams     */
/* ------------------------------------------------------------------------- */
static size_t effective_spp ( const TerraSceneOptions* o ) {
    size_t spp = o->samples_per_pixel;
    if ( o->sampling_method == kTerraSamplingMethodStratified ) {       /* :519-527 */
        size_t cur = o->strata * o->strata;
        while ( spp > cur && cur > 1 ) cur *= cur;          /* cur <= 1 never terminates in the reference */
        if ( cur >= spp ) spp = cur;
    }
    return spp;
}

void orc_render_pixels ( const TerraCamera* cam, HTerraScene h, const TerraFramebuffer* fb, size_t x, size_t y, size_t w, size_t hgt, uint64_t frame_seed, uint32_t* rand_calls ) {
    OrcScene* s = ( OrcScene* ) h;
    TerraFloat4x4 rot = orc_camera_frame ( cam );
    size_t spp = effective_spp ( &s->opts );
    for ( size_t i = y; i < y + hgt; ++i ) for ( size_t j = x; j < x + w; ++j ) {
        size_t pix = i * fb->width + j;
        TerraRawIntegrationResult* part = &fb->results[pix];
        OrcPixelStreams st = orc_pixel_streams ( frame_seed, pix, ( uint64_t ) ( uint32_t ) part->samples );
        OrcPcg32 A; pcgA_init ( &A, st.seedA );
        tls.streamB = st.streamB;
        uint64_t before = tls.c.rand_calls;
        v3 acc = v3_set ( 0, 0, 0 );
        for ( size_t k = 0; k < spp; ++k ) {
            float r1 = pcgA_nextf ( &A ), r2 = pcgA_nextf ( &A );
            v3 dir = orc_camera_sample ( cam, fb->width, fb->height, j, i, s->opts.subpixel_jitter, r1, r2 );
            dir = m3_apply ( &rot, dir );
            OrcRay ray = make_ray ( cam->position, dir );
            ++tls.c.samples;
            /* Sampler integration (extension; the reference constructs the pixel's stratified / Halton "hemisphere sampler" at src/Terra.c:535-548 and never
               draws from it -- this wiring is this repo's definition, UNPINNED): camera sample number n = (samples already in the pixel) + k takes element n of
               the pixel's sampler, right after r1, r2. Halton: the pair (radical inverse base 3, base 2) of n (src/Terra.c:734-755). Stratified: the sampler as
               :542 builds it -- `strata` strata per dimension, 16 samples per stratum, sharing the pixel's camera stream -- at element n mod (strata^2 * 16)
               (the reference asserts beyond that): stratum = m / 16, cell (stratum % strata, stratum / strata), offsets = the next two draws of stream A (:714-723). */
            float pair[2]; const float* first_pair = NULL;
            if ( s->sampler_integration && s->opts.sampling_method == kTerraSamplingMethodHalton ) {
                const uint64_t n = ( uint64_t ) ( uint32_t ) part->samples + k;
                pair[0] = orc_radical_inverse ( 3, n ); pair[1] = orc_radical_inverse ( 2, n ); first_pair = pair;
            } else if ( s->sampler_integration && s->opts.sampling_method == kTerraSamplingMethodStratified && s->opts.strata > 0 ) {
                const uint64_t strata = s->opts.strata, cap = strata * strata * 16;
                const uint64_t m = ( ( uint64_t ) ( uint32_t ) part->samples + k ) % cap, stratum = m / 16;
                const float stratum_size = 1.f / ( float ) strata, top = ( float ) ( 1.f - terra_Epsilon );
                const float a = ( ( float ) ( stratum % strata ) + pcgA_nextf ( &A ) ) * stratum_size;
                const float b = ( ( float ) ( stratum / strata ) + pcgA_nextf ( &A ) ) * stratum_size;
                pair[0] = sel_min ( a, top ); pair[1] = sel_min ( b, top ); first_pair = pair;
            }
            acc = v3_add ( acc, trace ( s, &ray, first_pair ) );
        }
        if ( rand_calls ) rand_calls[pix] = ( uint32_t ) ( tls.c.rand_calls - before );
        part->acc = v3_add ( acc, part->acc );
        part->samples += ( int ) spp;
        v3 color = v3_set ( part->acc.x / ( float ) part->samples, part->acc.y / ( float ) part->samples, part->acc.z / ( float ) part->samples );
        color = v3_scale ( color, s->opts.manual_exposure );
        fb->pixels[pix] = orc_tonemap ( &color, ( int ) s->opts.tonemapping_operator, s->opts.gamma );
    }
    counters_flush();
}

typedef struct { const TerraCamera* cam; HTerraScene h; const TerraFramebuffer* fb; size_t x, y, w, hgt; uint64_t seed; uint32_t* rc; volatile long* next; long step; } MtJob;
static void* mt_worker ( void* p ) {
    MtJob* j = ( MtJob* ) p;
    for ( ;; ) {
        long r = __sync_fetch_and_add ( j->next, j->step );
        if ( ( size_t ) r >= j->hgt ) break;
        size_t rows = ( size_t ) j->step; if ( ( size_t ) r + rows > j->hgt ) rows = j->hgt - ( size_t ) r;
        orc_render_pixels ( j->cam, j->h, j->fb, j->x, j->y + ( size_t ) r, j->w, rows, j->seed, j->rc );
    }
    return NULL;
}
void orc_render_pixels_mt ( const TerraCamera* cam, HTerraScene h, const TerraFramebuffer* fb, size_t x, size_t y, size_t w, size_t hgt, uint64_t seed, uint32_t* rc, int nthreads ) {
    volatile long next = 0;
    MtJob job = { cam, h, fb, x, y, w, hgt, seed, rc, &next, 2 };
    if ( nthreads < 1 ) nthreads = 1;
    if ( nthreads > 512 ) nthreads = 512;
    pthread_t* th = ( pthread_t* ) malloc ( sizeof ( pthread_t ) * ( size_t ) nthreads );
    for ( int t = 0; t < nthreads; ++t ) pthread_create ( &th[t], NULL, mt_worker, &job );
    for ( int t = 0; t < nthreads; ++t ) pthread_join ( th[t], NULL );
    free ( th );
}
void orc_render ( const TerraCamera* cam, HTerraScene h, const TerraFramebuffer* fb, size_t x, size_t y, size_t w, size_t hgt ) {
    orc_render_pixels ( cam, h, fb, x, y, w, hgt, ( ( OrcScene* ) h )->frame_seed, NULL );
}
void orc_set_frame_seed ( HTerraScene h, uint64_t seed ) { ( ( OrcScene* ) h )->frame_seed = seed; }
void orc_set_environment_lighting ( HTerraScene h, int on ) { OrcScene* s = ( OrcScene* ) h; s->env_lighting = on != 0; env_table_build ( s ); }
void orc_set_sampler_integration ( HTerraScene h, int on ) { ( ( OrcScene* ) h )->sampler_integration = on != 0; }
void orc_set_environment_sampling ( HTerraScene h, int on ) { OrcScene* s = ( OrcScene* ) h; s->env_sampling = on != 0; env_table_build ( s ); }

/* ------------------------------------------------------------------------- */
/* scene lifecycle (reference src/Terra.c:130-282)                             */
/* ------------------------------------------------------------------------- */
HTerraScene orc_scene_create ( void ) {
    OrcScene* s = ( OrcScene* ) calloc ( 1, sizeof ( OrcScene ) );
    s->objects_cap = 64; s->objects = ( TerraObject* ) malloc ( sizeof ( TerraObject ) * s->objects_cap );
    s->lights_cap = 16;  s->lights = ( OrcLight* ) malloc ( sizeof ( OrcLight ) * s->lights_cap );
    s->frame_seed = ORC_DEFAULT_FRAME_SEED;
    return s;
}
TerraObject* orc_scene_add_object ( HTerraScene h, size_t n ) {
    OrcScene* s = ( OrcScene* ) h;
    if ( s->objects_pop == s->objects_cap ) {
        s->objects_cap *= 2;
        s->objects = ( TerraObject* ) realloc ( s->objects, sizeof ( TerraObject ) * s->objects_cap );
    }
    TerraObject* o = &s->objects[s->objects_pop++];
    memset ( o, 0, sizeof *o );
    o->triangles = ( TerraTriangle* ) malloc ( sizeof ( TerraTriangle ) * ( n ? n : 1 ) );
    o->properties = ( TerraTriangleProperties* ) malloc ( sizeof ( TerraTriangleProperties ) * ( n ? n : 1 ) );
    o->triangles_count = n;
    s->dirty_objects = true; s->dirty_lights = true;
    return o;
}
size_t orc_scene_count_objects ( HTerraScene h ) { return ( ( OrcScene* ) h )->objects_pop; }
TerraSceneOptions* orc_scene_get_options ( HTerraScene h ) { return & ( ( OrcScene* ) h )->new_opts; }

static void free_lights ( OrcScene* s ) {
    for ( size_t i = 0; i < s->lights_pop; ++i ) free ( s->lights[i].triangle_area );
    s->lights_pop = 0;
}
void orc_scene_commit ( HTerraScene h ) {
    OrcScene* s = ( OrcScene* ) h;
    bool dirty_acc = s->dirty_objects || s->opts.accelerator != s->new_opts.accelerator || s->nodes == NULL;     /* a scene that never had an object has no tree in the reference (it would crash in traverse) */
    if ( dirty_acc ) bvh_destroy ( s );
    s->opts = s->new_opts;
    if ( dirty_acc ) bvh_build ( s );
    if ( s->dirty_lights ) {                            /* :194-231 */
        free_lights ( s );
        s->lights_triangles_count = 0;
        s->total_light_power = v3_set ( 0, 0, 0 );
        for ( size_t i = 0; i < s->objects_pop; ++i ) {
            TerraFloat2 uv = { 0.5f, 0.5f };
            v3 em = attribute_eval ( &s->objects[i].material.emissive, &uv, NULL );
            if ( em.x == 0 && em.y == 0 && em.z == 0 ) continue;
            if ( s->lights_pop == s->lights_cap ) { s->lights_cap *= 2; s->lights = ( OrcLight* ) realloc ( s->lights, sizeof ( OrcLight ) * s->lights_cap ); }
            OrcLight* l = &s->lights[s->lights_pop++];
            size_t n = s->objects[i].triangles_count;
            l->triangle_area = ( float* ) malloc ( sizeof ( float ) * ( n ? n : 1 ) );
            float area = 0;
            for ( size_t j = 0; j < n; ++j ) { float a = triangle_area ( &s->objects[i].triangles[j] ); l->triangle_area[j] = a; area += a; }
            l->power = v3_scale ( em, area * terra_PI );
            s->total_light_power = v3_add ( s->total_light_power, l->power );
            l->object = ( int ) i; l->area = area;
            s->lights_triangles_count += n;
        }
    }
    s->dirty_objects = false; s->dirty_lights = false;
    env_table_build ( s );
}
void orc_scene_clear ( HTerraScene h ) {
    OrcScene* s = ( OrcScene* ) h;
    for ( size_t i = 0; i < s->objects_pop; ++i ) { free ( s->objects[i].triangles ); free ( s->objects[i].properties ); }
    s->objects_pop = 0;
    free_lights ( s );
    s->dirty_objects = true; s->dirty_lights = true;
}
void orc_scene_destroy ( HTerraScene h ) {
    OrcScene* s = ( OrcScene* ) h;
    if ( !s ) return;
    orc_scene_clear ( h );
    env_table_free ( s );
    free ( s->objects ); free ( s->lights ); bvh_destroy ( s ); free ( s );
}
size_t orc_lights_count ( HTerraScene h ) { return ( ( OrcScene* ) h )->lights_pop; }
size_t orc_lights_triangles_count ( HTerraScene h ) { return ( ( OrcScene* ) h )->lights_triangles_count; }
int    orc_light_object_index ( HTerraScene h, size_t i ) { return ( ( OrcScene* ) h )->lights[i].object; }
float  orc_light_area ( HTerraScene h, size_t i ) { return ( ( OrcScene* ) h )->lights[i].area; }
const float* orc_light_triangle_areas ( HTerraScene h, size_t i ) { return ( ( OrcScene* ) h )->lights[i].triangle_area; }

/* ------------------------------------------------------------------------- */
/* devmath checks (tests/test_oracle_math.py)                                  */
/* ------------------------------------------------------------------------- */
/* every theta the diffuse sampler can produce: 2*terra_PI*(k*2^-24), k in [0,2^24) (reference src/TerraPresets.c:38) */
void orc_devmath_sincos_domain_check ( uint64_t* sin_mismatch, uint64_t* cos_mismatch ) {
    uint64_t bs = 0, bc = 0;
    for ( uint32_t k = 0; k < ( 1u << 24 ); ++k ) {
        float e2 = ( float ) k * 0x1p-24f;
        float theta = 2 * terra_PI * e2;
        if ( orc_dm_bits ( sinf ( theta ) ) != orc_dm_bits ( orc_dm_sinf ( theta ) ) ) ++bs;
        if ( orc_dm_bits ( cosf ( theta ) ) != orc_dm_bits ( orc_dm_cosf ( theta ) ) ) ++bc;
    }
    *sin_mismatch = bs; *cos_mismatch = bc;
}
/* fn: 0 sinf 1 cosf 2 powf 3 acosf; mode: ORC_MATH_LIBM / ORC_MATH_DEVMATH */
void orc_math_eval ( int fn, int mode, int n, const float* x, const float* y, float* out ) {
    for ( int i = 0; i < n; ++i ) {
        /* fn: 0 sinf, 1 cosf, 2 powf(x,y), 3 acosf, 4 atan2f(x,y) */
        if ( mode == ORC_MATH_LIBM ) out[i] = fn == 0 ? sinf ( x[i] ) : fn == 1 ? cosf ( x[i] ) : fn == 2 ? powf ( x[i], y[i] ) : fn == 3 ? acosf ( x[i] ) : atan2f ( x[i], y[i] );
        else out[i] = fn == 0 ? orc_dm_sinf ( x[i] ) : fn == 1 ? orc_dm_cosf ( x[i] ) : fn == 2 ? orc_dm_powf ( x[i], y[i] ) : fn == 3 ? orc_dm_acosf ( x[i] ) : orc_dm_atan2f ( x[i], y[i] );
    }
}

/* stream keys of one pixel: out3 = { seedA, stateB, incB } (stream_key.h) */
void orc_pixel_stream_key ( uint64_t frame_seed, uint64_t pix, uint64_t samples_so_far, uint64_t* out3 ) {
    OrcPixelStreams s = orc_pixel_streams ( frame_seed, pix, samples_so_far );
    out3[0] = s.seedA; out3[1] = s.streamB.state; out3[2] = s.streamB.inc;
}
