// dev_math.h -- transcendental functions used on the path, written as sequences of
// IEEE-754 basic operations so host and gfx950 produce the same bits
// (compile with -ffp-contract=off; HIP's default correctly rounded f32 div/sqrt).
//
// sinf/cosf: the published double-polynomial algorithm glibc >= 2.28 uses
// (ARM optimized-routines math/sinf.c, math/cosf.c, math/sincosf.h): reduce by
// pi/2 with a 2^24-scaled 2/pi, evaluate an odd or even minimax polynomial in
// double, round once. On every argument the diffuse sampler can produce
// (theta = 2*terra_PI*k*2^-24, reference src/TerraPresets.c:38-40) this gives
// exactly glibc 2.35's sinf/cosf (checked exhaustively on the CPU side).
// powf/acosf (Phong, gamma): double log2/exp2/asin kernels, rounded once.
#pragma once
#include <stdint.h>
#include <math.h>

#if defined(__HIPCC__)
#define TDM_FN __host__ __device__ inline
#else
#define TDM_FN static inline
#endif

TDM_FN uint32_t tdm_bits ( float f ) { return __builtin_bit_cast ( uint32_t, f ); }
TDM_FN float    tdm_float ( uint32_t u ) { return __builtin_bit_cast ( float, u ); }
TDM_FN uint32_t tdm_top12 ( float f ) { return ( tdm_bits ( f ) >> 20 ) & 0x7ffu; }

TDM_FN float tdm_sincos_poly ( double x, double x2, int n, bool flip ) {
    if ( ( n & 1 ) == 0 ) {
        const double S1 = -0x1.555545995a603p-3, S2 = 0x1.1107605230bc4p-7, S3 = -0x1.994eb3774cf24p-13;
        double x3 = x * x2;
        double t = S2 + x2 * S3;
        double x7 = x3 * x2;
        double s = x + x3 * S1;
        return ( float ) ( s + x7 * t );
    }
    double k = flip ? -1.0 : 1.0;
    const double C0 = 0x1p0, C1 = -0x1.ffffffd0c621cp-2, C2 = 0x1.55553e1068f19p-5, C3 = -0x1.6c087e89a359dp-10, C4 = 0x1.99343027bf8c3p-16;
    double x4 = x2 * x2;
    double t2 = ( k * C3 ) + x2 * ( k * C4 );
    double t1 = ( k * C0 ) + x2 * ( k * C1 );
    double x6 = x4 * x2;
    double c = t1 + x4 * ( k * C2 );
    return ( float ) ( c + x6 * t2 );
}

// shift = 0: sine, shift = 1: cosine. Valid for |y| < 120.
TDM_FN float tdm_sincosf ( float y, int shift ) {
    double x = y;
    if ( tdm_top12 ( y ) < tdm_top12 ( 0x1.921FB6p-1f ) ) {
        if ( tdm_top12 ( y ) < tdm_top12 ( 0x1p-12f ) ) {
            return shift ? 1.0f : y;
        }
        return tdm_sincos_poly ( x, x * x, shift, false );
    }
    double r = x * 0x1.45F306DC9C883p+23;
    int n = ( ( int32_t ) r + 0x800000 ) >> 24;
    x = x - ( double ) n * 0x1.921FB54442D18p0;
    int m = n + shift;
    int q = m & 3;
    double sgn = ( q == 1 || q == 2 ) ? -1.0 : 1.0;
    return tdm_sincos_poly ( x * sgn, x * x, m, ( m & 2 ) != 0 );
}
TDM_FN float tdm_sinf ( float y ) { return tdm_sincosf ( y, 0 ); }
TDM_FN float tdm_cosf ( float y ) { return tdm_sincosf ( y, 1 ); }

TDM_FN double tdm_log2_d ( double v ) {
    uint64_t u = __builtin_bit_cast ( uint64_t, v );
    int e = ( int ) ( ( u >> 52 ) & 0x7ff ) - 1023;
    u = ( u & 0x000fffffffffffffull ) | 0x3ff0000000000000ull;
    double m = __builtin_bit_cast ( double, u );
    if ( m > 1.4142135623730951 ) { m = m * 0.5; e += 1; }
    double s = ( m - 1.0 ) / ( m + 1.0 );
    double s2 = s * s;
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
    return ( double ) e + ( s * p ) * 2.8853900817779268;
}

TDM_FN double tdm_exp2_d ( double t ) {
    double fl = floor ( t + 0.5 );
    double r = ( t - fl ) * 0.6931471805599453;
    double p = 1.0 / 6227020800.0;
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
    if ( k > 1000 ) return __builtin_inf();
    uint64_t sb = ( uint64_t ) ( k + 1023 ) << 52;
    return p * __builtin_bit_cast ( double, sb );
}

TDM_FN float tdm_powf ( float x, float y ) {
    const float inf = __builtin_inff();
    if ( y == 0.0f || x == 1.0f ) return 1.0f;
    if ( x != x || y != y ) return __builtin_nanf ( "" );
    if ( x == 0.0f ) return y > 0.0f ? 0.0f : inf;
    bool negate = false;
    if ( x < 0.0f ) {
        float yi = floorf ( y );
        if ( yi != y ) return __builtin_nanf ( "" );
        negate = fabsf ( y ) < 16777216.0f && ( ( ( int64_t ) yi ) & 1 );
        x = -x;
        if ( x == 1.0f ) return negate ? -1.0f : 1.0f;
    }
    float r;
    if ( x == inf ) r = y > 0.0f ? inf : 0.0f;
    else if ( y == inf || y == -inf ) r = ( ( x > 1.0f ) == ( y > 0.0f ) ) ? inf : 0.0f;
    else {
        double t = ( double ) y * tdm_log2_d ( ( double ) x );
        if ( t > 200.0 ) r = inf;
        else if ( t < -200.0 ) r = 0.0f;
        else r = ( float ) tdm_exp2_d ( t );
    }
    return negate ? -r : r;
}

TDM_FN double tdm_asin_core_d ( double z ) {
    const double pS0 = 1.66666666666666657415e-01, pS1 = -3.25565818622400915405e-01, pS2 = 2.01212532134862925881e-01,
                 pS3 = -4.00555345006794114027e-02, pS4 = 7.91534994289814532176e-04, pS5 = 3.47933107596021167570e-05,
                 qS1 = -2.40339491173441421878e+00, qS2 = 2.02094576023350569471e+00, qS3 = -6.88283971605453293030e-01,
                 qS4 = 7.70381505559019352791e-02;
    double p = z * ( pS0 + z * ( pS1 + z * ( pS2 + z * ( pS3 + z * ( pS4 + z * pS5 ) ) ) ) );
    double q = 1.0 + z * ( qS1 + z * ( qS2 + z * ( qS3 + z * qS4 ) ) );
    return p / q;
}

TDM_FN float tdm_acosf ( float xf ) {
    double x = xf;
    const double pio2 = 1.57079632679489655800e+00, pi = 3.14159265358979311600e+00;
    if ( x != x || x > 1.0 || x < -1.0 ) return __builtin_nanf ( "" );
    if ( x == 1.0 ) return 0.0f;
    if ( x == -1.0 ) return ( float ) pi;
    double ax = x < 0 ? -x : x;
    if ( ax < 0.5 ) {
        double r = tdm_asin_core_d ( x * x );
        return ( float ) ( pio2 - ( x + x * r ) );
    }
    double z = ( 1.0 - ax ) * 0.5;
    double s = sqrt ( z );
    double r = tdm_asin_core_d ( z );
    double a = 2.0 * ( s + s * r );
    return ( float ) ( x < 0 ? pi - a : a );
}
