/*
 * TEST INFRASTRUCTURE (oracle/). Not part of the product.
 *
 * "devmath": transcendental functions restated as sequences of IEEE-754
 * binary64/binary32 basic operations (+ - * / sqrt, no FMA contraction), so
 * that the SAME sequence evaluated on the host (gcc -ffp-contract=off) and on
 * gfx950 (hipcc -ffp-contract=off, correctly rounded div/sqrt) gives the SAME
 * bits. The product carries its own copy (terra_amd/csrc/dev_math.h); the two
 * are compared bit-for-bit by tests/test_devmath.py through the C-ABI.
 *
 * Why not just call libm on the host and ocml on the device: the reference
 * calls glibc sinf/cosf in the diffuse sampler (reference src/TerraPresets.c:39-40)
 * and one flipped low bit there can flip a Russian-roulette or hit/miss branch
 * further down the path (SURVEY.md section 7 "Chaotic parity"). glibc 2.35's
 * sinf/cosf evaluate an odd/even minimax polynomial in double after a
 * 2/pi range reduction and round once (algorithm published with ARM's
 * optimized-routines, math/sinf.c, math/cosf.c, math/sincosf.h; glibc
 * sysdeps/ieee754/flt-32/s_sinf.c). orc_dm_sinf/orc_dm_cosf restate that
 * published algorithm; tests/test_oracle_math.py checks them against this
 * container's libm for EVERY input the diffuse sampler can produce
 * (theta = 2*terra_PI*e2 for the 2^24 values e2 = k*2^-24): 0 mismatches.
 */
#ifndef ORACLE_DEVMATH_H
#define ORACLE_DEVMATH_H
#include <stdint.h>
#include <string.h>
#include <math.h>

static inline uint32_t orc_dm_bits ( float f ) { uint32_t u; memcpy ( &u, &f, 4 ); return u; }
static inline float    orc_dm_float ( uint32_t u ) { float f; memcpy ( &f, &u, 4 ); return f; }
static inline uint32_t orc_dm_top12 ( float f ) { return ( orc_dm_bits ( f ) >> 20 ) & 0x7ffu; }

/* coefficients of the published sincosf tables (double) */
#define ORC_DM_HPI_INV 0x1.45F306DC9C883p+23   /* 2/pi * 2^24 */
#define ORC_DM_HPI     0x1.921FB54442D18p0     /* pi/2 */
#define ORC_DM_C0 0x1p0
#define ORC_DM_C1 -0x1.ffffffd0c621cp-2
#define ORC_DM_C2 0x1.55553e1068f19p-5
#define ORC_DM_C3 -0x1.6c087e89a359dp-10
#define ORC_DM_C4 0x1.99343027bf8c3p-16
#define ORC_DM_S1 -0x1.555545995a603p-3
#define ORC_DM_S2 0x1.1107605230bc4p-7
#define ORC_DM_S3 -0x1.994eb3774cf24p-13

/* polynomial on the reduced argument: quadrant parity n&1 picks sin or cos,
   flip negates the cosine coefficients (quadrants 2,3) */
static inline float orc_dm_sincos_poly ( double x, double x2, int n, int flip ) {
    if ( ( n & 1 ) == 0 ) {
        double x3 = x * x2;
        double t = ORC_DM_S2 + x2 * ORC_DM_S3;
        double x7 = x3 * x2;
        double s = x + x3 * ORC_DM_S1;
        return ( float ) ( s + x7 * t );
    } else {
        double k = flip ? -1.0 : 1.0;
        double x4 = x2 * x2;
        double t2 = ( k * ORC_DM_C3 ) + x2 * ( k * ORC_DM_C4 );
        double t1 = ( k * ORC_DM_C0 ) + x2 * ( k * ORC_DM_C1 );
        double x6 = x4 * x2;
        double c = t1 + x4 * ( k * ORC_DM_C2 );
        return ( float ) ( c + x6 * t2 );
    }
}

/* valid for |y| < 120 (the renderer only produces [0, 2*terra_PI] and [0, pi]) */
static inline float orc_dm_sinf ( float y ) {
    double x = y;
    if ( orc_dm_top12 ( y ) < orc_dm_top12 ( 0x1.921FB6p-1f ) ) {
        if ( orc_dm_top12 ( y ) < orc_dm_top12 ( 0x1p-12f ) ) {
            return y;
        }
        return orc_dm_sincos_poly ( x, x * x, 0, 0 );
    }
    double r = x * ORC_DM_HPI_INV;
    int n = ( ( int32_t ) r + 0x800000 ) >> 24;
    x = x - ( double ) n * ORC_DM_HPI;
    double sgn = ( ( n & 3 ) == 1 || ( n & 3 ) == 2 ) ? -1.0 : 1.0;
    return orc_dm_sincos_poly ( x * sgn, x * x, n, ( n & 2 ) != 0 );
}

static inline float orc_dm_cosf ( float y ) {
    double x = y;
    if ( orc_dm_top12 ( y ) < orc_dm_top12 ( 0x1.921FB6p-1f ) ) {
        if ( orc_dm_top12 ( y ) < orc_dm_top12 ( 0x1p-12f ) ) {
            return 1.0f;
        }
        return orc_dm_sincos_poly ( x, x * x, 1, 0 );
    }
    double r = x * ORC_DM_HPI_INV;
    int n = ( ( int32_t ) r + 0x800000 ) >> 24;
    x = x - ( double ) n * ORC_DM_HPI;
    int m = n + 1;
    double sgn = ( ( m & 3 ) == 1 || ( m & 3 ) == 2 ) ? -1.0 : 1.0;
    return orc_dm_sincos_poly ( x * sgn, x * x, m, ( m & 2 ) != 0 );
}

/* ---------------------------------------------------------------------------
 * powf / acosf (Phong lobe, tonemap gamma). Not bit-pinned to glibc: computed in
 * double from log2/exp2 built out of basic operations and rounded once, which
 * agrees with a correctly rounded powf except in rare double-rounding cases.
 * tests/test_oracle_math.py measures and states the mismatch rate vs libm.
 * ------------------------------------------------------------------------- */

/* log2(m) for m in [sqrt(1/2), sqrt(2)) via atanh series: log(m) = 2*(s + s^3/3 + ...), s = (m-1)/(m+1) */
static inline double orc_dm_log2_d ( double v ) {
    uint64_t u; memcpy ( &u, &v, 8 );
    int e = ( int ) ( ( u >> 52 ) & 0x7ff ) - 1023;
    u = ( u & 0x000fffffffffffffull ) | 0x3ff0000000000000ull;
    double m; memcpy ( &m, &u, 8 );
    if ( m > 1.4142135623730951 ) { m = m * 0.5; e += 1; }
    double s = ( m - 1.0 ) / ( m + 1.0 );
    double s2 = s * s;
    /* 2/ln2 * (s + s^3/3 + s^5/5 + ... + s^23/23): |s| <= 0.1716, truncation < 2^-60 */
    double p = 1.0 / 23.0;
    p = p * s2 + 1.0 / 21.0;
    p = p * s2 + 1.0 / 19.0;
    p = p * s2 + 1.0 / 17.0;
    p = p * s2 + 1.0 / 15.0;
    p = p * s2 + 1.0 / 13.0;
    p = p * s2 + 1.0 / 11.0;
    p = p * s2 + 1.0 / 9.0;
    p = p * s2 + 1.0 / 7.0;
    p = p * s2 + 1.0 / 5.0;
    p = p * s2 + 1.0 / 3.0;
    p = p * s2 + 1.0;
    return ( double ) e + ( s * p ) * 2.8853900817779268; /* 2/ln 2 */
}

/* 2^t for |t| < 1000: split integer part, Taylor in r*ln2 with |r| <= 0.5 */
static inline double orc_dm_exp2_d ( double t ) {
    double fl = floor ( t + 0.5 );
    double r = ( t - fl ) * 0.6931471805599453; /* ln 2 */
    double p = 1.0 / 6227020800.0;              /* 1/13! */
    p = p * r + 1.0 / 479001600.0;
    p = p * r + 1.0 / 39916800.0;
    p = p * r + 1.0 / 3628800.0;
    p = p * r + 1.0 / 362880.0;
    p = p * r + 1.0 / 40320.0;
    p = p * r + 1.0 / 5040.0;
    p = p * r + 1.0 / 720.0;
    p = p * r + 1.0 / 120.0;
    p = p * r + 1.0 / 24.0;
    p = p * r + 1.0 / 6.0;
    p = p * r + 0.5;
    p = p * r + 1.0;
    p = p * r + 1.0;
    int64_t k = ( int64_t ) fl;
    if ( k < -1000 ) return 0.0;
    if ( k > 1000 ) return INFINITY;
    uint64_t sb = ( uint64_t ) ( k + 1023 ) << 52;
    double scale; memcpy ( &scale, &sb, 8 );
    return p * scale;
}

/* C99 powf special cases that the renderer can reach, then exp2(y*log2(x)) */
static inline float orc_dm_powf ( float x, float y ) {
    if ( y == 0.0f || x == 1.0f ) return 1.0f;
    if ( x != x || y != y ) return NAN;
    if ( x == 0.0f ) return y > 0.0f ? 0.0f : INFINITY;
    if ( x < 0.0f ) {
        /* negative base: defined only for integral y */
        float yi = floorf ( y );
        if ( yi != y ) return NAN;
        float r = orc_dm_powf ( -x, y );
        int odd = fabsf ( y ) < 16777216.0f && ( ( ( int64_t ) yi ) & 1 );
        return odd ? -r : r;
    }
    if ( isinf ( x ) ) return y > 0.0f ? INFINITY : 0.0f;
    if ( isinf ( y ) ) {
        if ( x > 1.0f ) return y > 0.0f ? INFINITY : 0.0f;
        return y > 0.0f ? 0.0f : INFINITY;
    }
    double t = ( double ) y * orc_dm_log2_d ( ( double ) x );
    if ( t > 200.0 ) return INFINITY;
    if ( t < -200.0 ) return 0.0f;
    return ( float ) orc_dm_exp2_d ( t );
}

/* acos via atan-free identity in double: acos(x) = 2*asin(sqrt((1-x)/2)) for x>0.5 etc.
   asin by the fdlibm-style rational on [0,0.5], all in double, rounded once. */
static inline double orc_dm_asin_core_d ( double z ) {
    /* R(z) ~ (asin(sqrt z)/sqrt z - 1)/z on [0,0.25]; published fdlibm e_asin.c coefficients */
    const double pS0 = 1.66666666666666657415e-01, pS1 = -3.25565818622400915405e-01, pS2 = 2.01212532134862925881e-01,
                 pS3 = -4.00555345006794114027e-02, pS4 = 7.91534994289814532176e-04, pS5 = 3.47933107596021167570e-05,
                 qS1 = -2.40339491173441421878e+00, qS2 = 2.02094576023350569471e+00, qS3 = -6.88283971605453293030e-01,
                 qS4 = 7.70381505559019352791e-02;
    double p = z * ( pS0 + z * ( pS1 + z * ( pS2 + z * ( pS3 + z * ( pS4 + z * pS5 ) ) ) ) );
    double q = 1.0 + z * ( qS1 + z * ( qS2 + z * ( qS3 + z * qS4 ) ) );
    return p / q;
}

static inline float orc_dm_acosf ( float xf ) {
    double x = xf;
    const double pio2 = 1.57079632679489655800e+00, pi = 3.14159265358979311600e+00;
    if ( x != x || x > 1.0 || x < -1.0 ) return NAN;
    if ( x == 1.0 ) return 0.0f;
    if ( x == -1.0 ) return ( float ) pi;
    double ax = x < 0 ? -x : x;
    if ( ax < 0.5 ) {
        double r = orc_dm_asin_core_d ( x * x );
        return ( float ) ( pio2 - ( x + x * r ) );
    }
    double z = ( 1.0 - ax ) * 0.5;
    double s = sqrt ( z );
    double r = orc_dm_asin_core_d ( z );
    double a = 2.0 * ( s + s * r );      /* acos(|x|) */
    return ( float ) ( x < 0 ? pi - a : a );
}

#endif
