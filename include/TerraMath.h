/*
 * TerraMath.h -- vector/matrix value types and small helpers of the Terra C API.
 *
 * Drop-in boundary, part 1 of 3 (TerraMath.h, Terra.h, TerraPresets.h).
 * This header is written from scratch for terra_amd; it reproduces the *layout*
 * and the *names* a client of the reference compiles against
 * (reference: include/TerraMath.h:17-38 types/constants, :52-98 helper names,
 * include/TerraMath.inl for their semantics) so that existing client code keeps
 * compiling unchanged. Conventions (reference include/TerraMath.h:10-12):
 * row-major 4x4, column vectors (M*v), left-handed: x right, y up, z forward.
 *
 * Parity notes that clients can observe through these helpers:
 *   - terra_PI is 3.1416926535f, not pi, and terra_Epsilon is a *double* 1e-4
 *     (reference include/TerraMath.h:17-18). Both are kept bit-for-bit.
 *   - terra_minf/terra_maxf are compare-selects (NaN-order sensitive),
 *     terra_maxf3/terra_min3 are fmaxf/fminf (reference TerraMath.inl:171-203).
 *   - terra_f4x4_basis scales (does not normalise) the tangent
 *     (reference TerraMath.inl:258-264).
 */
#ifndef TERRA_AMD_TERRA_MATH_H
#define TERRA_AMD_TERRA_MATH_H

#include <stdbool.h>
#include <stddef.h>
#include <stdint.h>
#include <string.h>
#include <math.h>
#include <float.h>

#define terra_PI 3.1416926535f
#define terra_Epsilon 1e-4
#define terra_ior_air 1.f

typedef struct TerraInt4   { int   x, y, z, w; } TerraInt4;
typedef struct TerraFloat2 { float x, y;       } TerraFloat2;
typedef struct TerraFloat3 { float x, y, z;    } TerraFloat3;
typedef struct TerraFloat4 { float x, y, z, w; } TerraFloat4;
typedef struct TerraFloat4x4 { TerraFloat4 rows[4]; } TerraFloat4x4;

#ifdef __cplusplus
#define TERRA_LIT(T) T
#else
#define TERRA_LIT(T) (T)
#endif
#define terra_f2_zero (TERRA_LIT(TerraFloat2){0.f, 0.f})
#define terra_f3_zero (TERRA_LIT(TerraFloat3){0.f, 0.f, 0.f})
#define terra_f3_one  (TERRA_LIT(TerraFloat3){1.f, 1.f, 1.f})

#define TERRA_MATH_FN static inline

/* ---- constructors ------------------------------------------------------ */
TERRA_MATH_FN TerraFloat2 terra_f2_set ( float x, float y )                    { TerraFloat2 r = { x, y };       return r; }
TERRA_MATH_FN TerraFloat3 terra_f3_set ( float x, float y, float z )           { TerraFloat3 r = { x, y, z };    return r; }
TERRA_MATH_FN TerraFloat3 terra_f3_set1 ( float v )                            { TerraFloat3 r = { v, v, v };    return r; }
TERRA_MATH_FN TerraFloat3 terra_f3_setv ( const float* p )                     { TerraFloat3 r = { p[0], p[1], p[2] }; return r; }
TERRA_MATH_FN TerraFloat4 terra_f4_set ( float x, float y, float z, float w )  { TerraFloat4 r = { x, y, z, w }; return r; }
TERRA_MATH_FN TerraInt4   terra_i4_set ( int x, int y, int z, int w )          { TerraInt4 r = { x, y, z, w };   return r; }

/* ---- component-wise arithmetic ----------------------------------------- */
TERRA_MATH_FN bool        terra_equalf3 ( const TerraFloat3* a, const TerraFloat3* b ) { return a->x == b->x && a->y == b->y && a->z == b->z; }
TERRA_MATH_FN TerraFloat3 terra_addf3 ( const TerraFloat3* a, const TerraFloat3* b )   { return terra_f3_set ( a->x + b->x, a->y + b->y, a->z + b->z ); }
TERRA_MATH_FN TerraFloat2 terra_addf2 ( const TerraFloat2* a, const TerraFloat2* b )   { return terra_f2_set ( a->x + b->x, a->y + b->y ); }
TERRA_MATH_FN TerraFloat3 terra_subf3 ( const TerraFloat3* a, const TerraFloat3* b )   { return terra_f3_set ( a->x - b->x, a->y - b->y, a->z - b->z ); }
TERRA_MATH_FN TerraFloat2 terra_mulf2 ( const TerraFloat2* a, float s )                { return terra_f2_set ( a->x * s, a->y * s ); }
TERRA_MATH_FN TerraFloat3 terra_mulf3 ( const TerraFloat3* a, float s )                { return terra_f3_set ( a->x * s, a->y * s, a->z * s ); }
TERRA_MATH_FN TerraFloat3 terra_divf3 ( const TerraFloat3* a, float s )                { return terra_f3_set ( a->x / s, a->y / s, a->z / s ); }
TERRA_MATH_FN TerraFloat3 terra_powf3 ( const TerraFloat3* a, float e )                { return terra_f3_set ( powf ( a->x, e ), powf ( a->y, e ), powf ( a->z, e ) ); }
TERRA_MATH_FN TerraFloat3 terra_pointf3 ( const TerraFloat3* a, const TerraFloat3* b ) { return terra_f3_set ( a->x * b->x, a->y * b->y, a->z * b->z ); }
TERRA_MATH_FN TerraFloat3 terra_negf3 ( const TerraFloat3* a )                         { return terra_f3_set ( -a->x, -a->y, -a->z ); }
TERRA_MATH_FN TerraFloat3 terra_absf3 ( const TerraFloat3* a )                         { return terra_f3_set ( fabsf ( a->x ), fabsf ( a->y ), fabsf ( a->z ) ); }

/* ---- products, lengths -------------------------------------------------- */
TERRA_MATH_FN float terra_dotf3 ( const TerraFloat3* a, const TerraFloat3* b ) { return a->x * b->x + a->y * b->y + a->z * b->z; }
TERRA_MATH_FN TerraFloat3 terra_crossf3 ( const TerraFloat3* a, const TerraFloat3* b ) {
    return terra_f3_set ( a->y * b->z - a->z * b->y, a->z * b->x - a->x * b->z, a->x * b->y - a->y * b->x );
}
TERRA_MATH_FN float terra_sqlenf3 ( const TerraFloat3* a ) { return a->x * a->x + a->y * a->y + a->z * a->z; }
TERRA_MATH_FN float terra_lenf3 ( const TerraFloat3* a )   { return sqrtf ( a->x * a->x + a->y * a->y + a->z * a->z ); }
TERRA_MATH_FN float terra_distf3 ( const TerraFloat3* a, const TerraFloat3* b )   { TerraFloat3 d = terra_subf3 ( a, b ); return terra_lenf3 ( &d ); }
TERRA_MATH_FN float terra_sqdistf3 ( const TerraFloat3* a, const TerraFloat3* b ) { TerraFloat3 d = terra_subf3 ( b, a ); return terra_dotf3 ( &d, &d ); }
/* three IEEE divisions by the length, not a multiply by its reciprocal */
TERRA_MATH_FN TerraFloat3 terra_normf3 ( const TerraFloat3* a ) { float l = terra_lenf3 ( a ); return terra_f3_set ( a->x / l, a->y / l, a->z / l ); }

/* ---- scalar helpers ------------------------------------------------------ */
TERRA_MATH_FN float    terra_maxf ( float a, float b )   { return a > b ? a : b; }
TERRA_MATH_FN float    terra_minf ( float a, float b )   { return a < b ? a : b; }
TERRA_MATH_FN size_t   terra_maxi ( size_t a, size_t b ) { return a > b ? a : b; }
TERRA_MATH_FN size_t   terra_mini ( size_t a, size_t b ) { return a < b ? a : b; }
TERRA_MATH_FN int      terra_signf ( float v )           { return v == 0.f ? 0 : ( v > 0.f ? 1 : -1 ); }
TERRA_MATH_FN uint32_t terra_signf_mask ( float v )      { uint32_t u; memcpy ( &u, &v, 4 ); return u & 0x80000000u; }
TERRA_MATH_FN float    terra_xorf ( float a, float b )   { uint32_t x, y; memcpy ( &x, &a, 4 ); memcpy ( &y, &b, 4 ); x ^= y; memcpy ( &a, &x, 4 ); return a; }
TERRA_MATH_FN void     terra_swap_xorf ( float* a, float* b ) { float t = *a; *a = *b; *b = t; }
TERRA_MATH_FN void     terra_swap_xori ( int* a, int* b )     { int t = *a; *a = *b; *b = t; }
TERRA_MATH_FN float    terra_maxf3 ( const TerraFloat3* a ) { return fmaxf ( a->x, fmaxf ( a->y, a->z ) ); }
TERRA_MATH_FN float    terra_min3 ( const TerraFloat3* a )  { return fminf ( a->x, fminf ( a->y, a->z ) ); }
TERRA_MATH_FN float    terra_clamp ( float v, float lo, float hi ) { return v < lo ? lo : v > hi ? hi : v; }
TERRA_MATH_FN float    terra_sqr ( float v )             { return v * v; }
TERRA_MATH_FN float    terra_lerp ( float a, float b, float t ) { return a + ( b - a ) * t; }
TERRA_MATH_FN bool     terra_f3_is_zero ( const TerraFloat3* a ) { return a->x == 0 && a->y == 0 && a->z == 0; }
/* index of the largest component; ties resolve to the later axis */
TERRA_MATH_FN int terra_max_coefff3 ( const TerraFloat3* a ) {
    if ( a->x > a->y ) { return a->x > a->z ? 0 : 2; }
    return a->y > a->z ? 1 : 2;
}
TERRA_MATH_FN TerraFloat3 terra_lerpf3 ( const TerraFloat3* a, const TerraFloat3* b, float t ) {
    return terra_f3_set ( terra_lerp ( a->x, b->x, t ), terra_lerp ( a->y, b->y, t ), terra_lerp ( a->z, b->z, t ) );
}
TERRA_MATH_FN TerraFloat3 terra_clampf3 ( const TerraFloat3* v, const TerraFloat3* lo, const TerraFloat3* hi ) {
    TerraFloat3 r;
    r.x = v->x > lo->x ? v->x : lo->x;  r.y = v->y > lo->y ? v->y : lo->y;  r.z = v->z > lo->z ? v->z : lo->z;
    r.x = r.x < hi->x ? r.x : hi->x;    r.y = r.y < hi->y ? r.y : hi->y;    r.z = r.z < hi->z ? r.z : hi->z;
    return r;
}

/* ---- 4x4 ----------------------------------------------------------------- */
/* upper-left 3x3 of M applied to v */
TERRA_MATH_FN TerraFloat3 terra_transformf3 ( const TerraFloat4x4* m, const TerraFloat3* v ) {
    return terra_f3_set ( m->rows[0].x * v->x + m->rows[0].y * v->y + m->rows[0].z * v->z,
                          m->rows[1].x * v->x + m->rows[1].y * v->y + m->rows[1].z * v->z,
                          m->rows[2].x * v->x + m->rows[2].y * v->y + m->rows[2].z * v->z );
}
/* columns = (tangent, normal, bitangent); tangent is scaled by, not divided by, its would-be length */
TERRA_MATH_FN TerraFloat4x4 terra_f4x4_basis ( const TerraFloat3* n ) {
    TerraFloat3 t, b;
    if ( fabsf ( n->x ) > fabsf ( n->y ) ) {
        float k = sqrtf ( n->x * n->x + n->z * n->z );
        t = terra_f3_set ( n->z * k, 0.f * k, -n->x * k );
    } else {
        float k = sqrtf ( n->y * n->y + n->z * n->z );
        t = terra_f3_set ( 0.f * k, -n->z * k, n->y * k );
    }
    b = terra_crossf3 ( n, &t );
    TerraFloat4x4 m;
    m.rows[0] = terra_f4_set ( t.x, n->x, b.x, 0.f );
    m.rows[1] = terra_f4_set ( t.y, n->y, b.y, 0.f );
    m.rows[2] = terra_f4_set ( t.z, n->z, b.z, 0.f );
    m.rows[3] = terra_f4_set ( 0.f, 0.f, 0.f, 1.f );
    return m;
}
TERRA_MATH_FN TerraFloat3 terra_f4x4_get_tangent ( const TerraFloat4x4* m )   { return terra_f3_set ( m->rows[0].x, m->rows[1].x, m->rows[2].x ); }
TERRA_MATH_FN TerraFloat3 terra_f4x4_get_normal ( const TerraFloat4x4* m )    { return terra_f3_set ( m->rows[0].y, m->rows[1].y, m->rows[2].y ); }
TERRA_MATH_FN TerraFloat3 terra_f4x4_get_bitangent ( const TerraFloat4x4* m ) { return terra_f3_set ( m->rows[0].z, m->rows[1].z, m->rows[2].z ); }

#endif /* TERRA_AMD_TERRA_MATH_H */
