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
// powf: the published table+polynomial algorithm behind glibc 2.35's powf; acosf: the fdlibm
// single-precision kernel glibc 2.35 ships. Both reproduce this image's libm bit for bit on
// every argument tried (2*10^7 / 4*10^8; DESIGN.md "Bit-faithful arithmetic").
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
// sin and cos of one argument with ONE range reduction; each result is bit-identical to
// tdm_sincosf(y, 0/1) because the reduced argument and both polynomials are evaluated with
// the same operations in the same order.
TDM_FN void tdm_sincosf_pair ( float y, float& s, float& c ) {
    double x = y;
    if ( tdm_top12 ( y ) < tdm_top12 ( 0x1.921FB6p-1f ) ) {
        if ( tdm_top12 ( y ) < tdm_top12 ( 0x1p-12f ) ) { s = y; c = 1.0f; return; }
        double x2 = x * x;
        s = tdm_sincos_poly ( x, x2, 0, false );
        c = tdm_sincos_poly ( x, x2, 1, false );
        return;
    }
    double r = x * 0x1.45F306DC9C883p+23;
    int n = ( ( int32_t ) r + 0x800000 ) >> 24;
    x = x - ( double ) n * 0x1.921FB54442D18p0;
    double x2 = x * x;
    int qs = n & 3, m = n + 1, qc = m & 3;
    double sgn_s = ( qs == 1 || qs == 2 ) ? -1.0 : 1.0;
    double sgn_c = ( qc == 1 || qc == 2 ) ? -1.0 : 1.0;
    s = tdm_sincos_poly ( x * sgn_s, x2, n, ( n & 2 ) != 0 );
    c = tdm_sincos_poly ( x * sgn_c, x2, m, ( m & 2 ) != 0 );
}
TDM_FN float tdm_sinf ( float y ) { return tdm_sincosf ( y, 0 ); }
TDM_FN float tdm_cosf ( float y ) { return tdm_sincosf ( y, 1 ); }

TDM_FN uint64_t tdm_bits64 ( double f ) { return __builtin_bit_cast ( uint64_t, f ); }
TDM_FN double   tdm_double ( uint64_t u ) { return __builtin_bit_cast ( double, u ); }

/* ---------------------------------------------------------------------------
 * powf: restatement of the published algorithm behind glibc 2.35's powf
 * (ARM optimized-routines math/powf.c + powf_log2_data.c + exp2f_data.c; glibc
 * sysdeps/ieee754/flt-32/e_powf.c): log2(x) from a 16-entry (1/c, log2 c) table
 * and a degree-5 polynomial in double, y*log2(x) in double, 2^t from a 32-entry
 * table and a cubic, rounded to float once. Checked against this container's
 * libm on 2*10^7 arguments incl. random bit patterns: 0 mismatches
 * (the CPU twin of this file is checked in tests/test_oracle_math.py).
 * ------------------------------------------------------------------------- */
TDM_FN double tdm_powf_log2 ( uint32_t ix ) {
    static const double T[16][2] = {
        { 0x1.661ec79f8f3bep+0, -0x1.efec65b963019p-2 }, { 0x1.571ed4aaf883dp+0, -0x1.b0b6832d4fca4p-2 },
        { 0x1.49539f0f010bp+0, -0x1.7418b0a1fb77bp-2 }, { 0x1.3c995b0b80385p+0, -0x1.39de91a6dcf7bp-2 },
        { 0x1.30d190c8864a5p+0, -0x1.01d9bf3f2b631p-2 }, { 0x1.25e227b0b8eap+0, -0x1.97c1d1b3b7afp-3 },
        { 0x1.1bb4a4a1a343fp+0, -0x1.2f9e393af3c9fp-3 }, { 0x1.12358f08ae5bap+0, -0x1.960cbbf788d5cp-4 },
        { 0x1.0953f419900a7p+0, -0x1.a6f9db6475fcep-5 }, { 0x1p+0, 0x0p+0 },
        { 0x1.e608cfd9a47acp-1, 0x1.338ca9f24f53dp-4 }, { 0x1.ca4b31f026aap-1, 0x1.476a9543891bap-3 },
        { 0x1.b2036576afce6p-1, 0x1.e840b4ac4e4d2p-3 }, { 0x1.9c2d163a1aa2dp-1, 0x1.40645f0c6651cp-2 },
        { 0x1.886e6037841edp-1, 0x1.88e9c2c1b9ff8p-2 }, { 0x1.767dcf5534862p-1, 0x1.ce0a44eb17bccp-2 } };
    const double A0 = 0x1.27616c9496e0bp-2, A1 = -0x1.71969a075c67ap-2, A2 = 0x1.ec70a6ca7baddp-2, A3 = -0x1.7154748bef6c8p-1, A4 = 0x1.71547652ab82bp0;
    uint32_t tmp = ix - 0x3f330000u;
    int i = ( int ) ( ( tmp >> 19 ) % 16u );
    uint32_t top = tmp & 0xff800000u;
    uint32_t iz = ix - top;
    int k = ( int32_t ) top >> 23;
    double z = ( double ) tdm_float ( iz );
    double r = z * T[i][0] - 1.0;
    double y0 = T[i][1] + ( double ) k;
    double r2 = r * r;
    double y = A0 * r + A1;
    double p = A2 * r + A3;
    double r4 = r2 * r2;
    double q = A4 * r + y0;
    q = p * r2 + q;
    return y * r4 + q;
}
TDM_FN float tdm_powf_exp2 ( double xd, uint32_t sign_bias ) {
    static const uint64_t T[32] = {
        0x3ff0000000000000ull, 0x3fefd9b0d3158574ull, 0x3fefb5586cf9890full, 0x3fef9301d0125b51ull,
        0x3fef72b83c7d517bull, 0x3fef54873168b9aaull, 0x3fef387a6e756238ull, 0x3fef1e9df51fdee1ull,
        0x3fef06fe0a31b715ull, 0x3feef1a7373aa9cbull, 0x3feedea64c123422ull, 0x3feece086061892dull,
        0x3feebfdad5362a27ull, 0x3feeb42b569d4f82ull, 0x3feeab07dd485429ull, 0x3feea47eb03a5585ull,
        0x3feea09e667f3bcdull, 0x3fee9f75e8ec5f74ull, 0x3feea11473eb0187ull, 0x3feea589994cce13ull,
        0x3feeace5422aa0dbull, 0x3feeb737b0cdc5e5ull, 0x3feec49182a3f090ull, 0x3feed503b23e255dull,
        0x3feee89f995ad3adull, 0x3feeff76f2fb5e47ull, 0x3fef199bdd85529cull, 0x3fef3720dcef9069ull,
        0x3fef5818dcfba487ull, 0x3fef7c97337b9b5full, 0x3fefa4afa2a490daull, 0x3fefd0765b6e4540ull };
    const double C0 = 0x1.c6af84b912394p-5, C1 = 0x1.ebfce50fac4f3p-3, C2 = 0x1.62e42ff0c52d6p-1, SHIFT = 0x1.8p+47;
    double kd = xd + SHIFT;
    uint64_t ki = tdm_bits64 ( kd );
    kd = kd - SHIFT;
    double r = xd - kd;
    uint64_t t = T[ki % 32u];
    uint64_t ski = ki + sign_bias;
    t += ski << 47;
    double s = tdm_double ( t );
    double z = C0 * r + C1;
    double r2 = r * r;
    double y = C2 * r + 1.0;
    y = z * r2 + y;
    y = y * s;
    return ( float ) y;
}
/* 0: y is not an integer, 1: odd integer, 2: even integer */
TDM_FN int tdm_powf_checkint ( uint32_t iy ) {
    int e = ( int ) ( iy >> 23 & 0xff );
    if ( e < 0x7f ) return 0;
    if ( e > 0x7f + 23 ) return 2;
    if ( iy & ( ( 1u << ( 0x7f + 23 - e ) ) - 1 ) ) return 0;
    if ( iy & ( 1u << ( 0x7f + 23 - e ) ) ) return 1;
    return 2;
}
TDM_FN bool tdm_zeroinfnan ( uint32_t ix ) { return 2 * ix - 1 >= 2u * 0x7f800000u - 1; }

TDM_FN float tdm_powf ( float x, float y ) {
    uint32_t sign_bias = 0;
    uint32_t ix = tdm_bits ( x ), iy = tdm_bits ( y );
    if ( ix - 0x00800000u >= 0x7f800000u - 0x00800000u || tdm_zeroinfnan ( iy ) ) {
        if ( tdm_zeroinfnan ( iy ) ) {
            if ( 2 * iy == 0 ) return 1.0f;
            if ( ix == 0x3f800000u ) return 1.0f;
            if ( 2 * ix > 2u * 0x7f800000u || 2 * iy > 2u * 0x7f800000u ) return x + y;
            if ( 2 * ix == 2 * 0x3f800000u ) return 1.0f;
            if ( ( 2 * ix < 2 * 0x3f800000u ) == ! ( iy & 0x80000000u ) ) return 0.0f;
            return y * y;
        }
        if ( tdm_zeroinfnan ( ix ) ) {
            float x2 = x * x;
            if ( ( ix & 0x80000000u ) && tdm_powf_checkint ( iy ) == 1 ) x2 = -x2;
            return ( iy & 0x80000000u ) ? 1 / x2 : x2;
        }
        if ( ix & 0x80000000u ) {
            int yint = tdm_powf_checkint ( iy );
            if ( yint == 0 ) return ( x - x ) / ( x - x );
            if ( yint == 1 ) sign_bias = 1u << 16;
            ix &= 0x7fffffffu;
        }
        if ( ix < 0x00800000u ) {
            ix = tdm_bits ( x * 0x1p23f );
            ix &= 0x7fffffffu;
            ix -= 23u << 23;
        }
    }
    double logx = tdm_powf_log2 ( ix );
    double ylogx = ( double ) y * logx;
    if ( ( tdm_bits64 ( ylogx ) >> 47 & 0xffff ) >= ( tdm_bits64 ( 126.0 ) >> 47 ) ) {
        if ( ylogx > 0x1.fffffffd1d571p+6 ) return sign_bias ? -__builtin_inff() : __builtin_inff();
        if ( ylogx <= -150.0 ) return sign_bias ? -0.0f : 0.0f;
    }
    return tdm_powf_exp2 ( ylogx, sign_bias );
}

/* ---------------------------------------------------------------------------
 * acosf: restatement of the fdlibm single-precision kernel glibc 2.35 ships
 * (sysdeps/ieee754/flt-32/e_acosf.c, from Sun's e_acos.c): rational
 * approximation in FLOAT arithmetic with a split square root. Checked against
 * this container's libm on every 7th float in [-1,1]: 0 mismatches.
 * ------------------------------------------------------------------------- */
TDM_FN float tdm_acosf ( float x ) {
    const float one = 1.0000000000e+00f, pi = 3.1415925026e+00f, pio2_hi = 1.5707962513e+00f, pio2_lo = 7.5497894159e-08f,
                pS0 = 1.6666667163e-01f, pS1 = -3.2556581497e-01f, pS2 = 2.0121252537e-01f, pS3 = -4.0055535734e-02f,
                pS4 = 7.9153501429e-04f, pS5 = 3.4793309169e-05f,
                qS1 = -2.4033949375e+00f, qS2 = 2.0209457874e+00f, qS3 = -6.8828397989e-01f, qS4 = 7.7038154006e-02f;
    int32_t hx = ( int32_t ) tdm_bits ( x ), ix = hx & 0x7fffffff;
    if ( ix == 0x3f800000 ) {
        if ( hx > 0 ) return 0.0f;
        return pi + 2.0f * pio2_lo;
    } else if ( ix > 0x3f800000 ) {
        return ( x - x ) / ( x - x );
    }
    if ( ix < 0x3f000000 ) {
        if ( ix <= 0x23000000 ) return pio2_hi + pio2_lo;
        float z = x * x;
        float p = z * ( pS0 + z * ( pS1 + z * ( pS2 + z * ( pS3 + z * ( pS4 + z * pS5 ) ) ) ) );
        float q = one + z * ( qS1 + z * ( qS2 + z * ( qS3 + z * qS4 ) ) );
        float r = p / q;
        return pio2_hi - ( x - ( pio2_lo - x * r ) );
    } else if ( hx < 0 ) {
        float z = ( one + x ) * 0.5f;
        float p = z * ( pS0 + z * ( pS1 + z * ( pS2 + z * ( pS3 + z * ( pS4 + z * pS5 ) ) ) ) );
        float q = one + z * ( qS1 + z * ( qS2 + z * ( qS3 + z * qS4 ) ) );
        float s = sqrtf ( z );
        float r = p / q;
        float w = r * s - pio2_lo;
        return pi - 2.0f * ( s + w );
    }
    float z = ( one - x ) * 0.5f;
    float s = sqrtf ( z );
    float df = tdm_float ( tdm_bits ( s ) & 0xfffff000u );
    float c = ( z - df * df ) / ( s + df );
    float p = z * ( pS0 + z * ( pS1 + z * ( pS2 + z * ( pS3 + z * ( pS4 + z * pS5 ) ) ) ) );
    float q = one + z * ( qS1 + z * ( qS2 + z * ( qS3 + z * qS4 ) ) );
    float r = p / q;
    float w = r * s + c;
    return 2.0f * ( df + w );
}

/* ---------------------------------------------------------------------------
 * atanf / atan2f: restatement of the fdlibm single-precision kernels glibc 2.35
 * ships (sysdeps/ieee754/flt-32/s_atanf.c, e_atan2f.c): argument reduction to
 * four breakpoints + an odd degree-11-in-z polynomial, all in FLOAT arithmetic
 * (large-argument cut at 2^25, as this glibc has it). Checked against this
 * container's libm: atanf on every 7th float of the whole range, atan2f on
 * 3e8 random pairs (raw bit patterns and unit-square values): 0 mismatches.
 * ------------------------------------------------------------------------- */
TDM_FN float tdm_atanf ( float x ) {
    const float hi0 = 4.6364760399e-01f, hi1 = 7.8539812565e-01f, hi2 = 9.8279368877e-01f, hi3 = 1.5707962513e+00f,
                lo0 = 5.0121582440e-09f, lo1 = 3.7748947079e-08f, lo2 = 3.4473217170e-08f, lo3 = 7.5497894159e-08f,
                aT0 = 3.3333334327e-01f, aT1 = -2.0000000298e-01f, aT2 = 1.4285714924e-01f, aT3 = -1.1111110449e-01f,
                aT4 = 9.0908870101e-02f, aT5 = -7.6918758452e-02f, aT6 = 6.6610731184e-02f, aT7 = -5.8335702866e-02f,
                aT8 = 4.9768779427e-02f, aT9 = -3.6531571299e-02f, aT10 = 1.6285819933e-02f;
    int32_t hx = ( int32_t ) tdm_bits ( x ), ix = hx & 0x7fffffff;
    int id;
    if ( ix >= 0x4c000000 ) {
        if ( ix > 0x7f800000 ) return x + x;
        return hx > 0 ? hi3 + lo3 : -hi3 - lo3;
    }
    if ( ix < 0x3ee00000 ) {
        if ( ix < 0x31000000 ) return x;
        id = -1;
    } else {
        x = tdm_float ( ( uint32_t ) ix );
        if ( ix < 0x3f980000 ) {
            if ( ix < 0x3f300000 ) { id = 0; x = ( 2.0f * x - 1.0f ) / ( 2.0f + x ); }
            else { id = 1; x = ( x - 1.0f ) / ( x + 1.0f ); }
        } else {
            if ( ix < 0x401c0000 ) { id = 2; x = ( x - 1.5f ) / ( 1.0f + 1.5f * x ); }
            else { id = 3; x = -1.0f / x; }
        }
    }
    float z = x * x, w = z * z;
    float s1 = z * ( aT0 + w * ( aT2 + w * ( aT4 + w * ( aT6 + w * ( aT8 + w * aT10 ) ) ) ) );
    float s2 = w * ( aT1 + w * ( aT3 + w * ( aT5 + w * ( aT7 + w * aT9 ) ) ) );
    if ( id < 0 ) return x - x * ( s1 + s2 );
    const float hi = id == 0 ? hi0 : id == 1 ? hi1 : id == 2 ? hi2 : hi3, lo = id == 0 ? lo0 : id == 1 ? lo1 : id == 2 ? lo2 : lo3;
    z = hi - ( ( x * ( s1 + s2 ) - lo ) - x );
    return hx < 0 ? -z : z;
}
TDM_FN float tdm_atan2f ( float y, float x ) {
    const float tiny = 1.0e-30f, pi_o_4 = 7.8539818525e-01f, pi_o_2 = 1.5707963705e+00f, pi = 3.1415927410e+00f, pi_lo = -8.7422776573e-08f;
    int32_t hx = ( int32_t ) tdm_bits ( x ), hy = ( int32_t ) tdm_bits ( y ), ix = hx & 0x7fffffff, iy = hy & 0x7fffffff;
    if ( ix > 0x7f800000 || iy > 0x7f800000 ) return x + y;
    if ( hx == 0x3f800000 ) return tdm_atanf ( y );
    const int m = ( ( hy >> 31 ) & 1 ) | ( ( hx >> 30 ) & 2 );
    if ( iy == 0 ) return m < 2 ? y : ( m == 2 ? pi + tiny : -pi - tiny );
    if ( ix == 0 ) return hy < 0 ? -pi_o_2 - tiny : pi_o_2 + tiny;
    if ( ix == 0x7f800000 ) {
        if ( iy == 0x7f800000 ) return m == 0 ? pi_o_4 + tiny : m == 1 ? -pi_o_4 - tiny : m == 2 ? 3.0f * pi_o_4 + tiny : -3.0f * pi_o_4 - tiny;
        return m == 0 ? 0.0f : m == 1 ? -0.0f : m == 2 ? pi + tiny : -pi - tiny;
    }
    if ( iy == 0x7f800000 ) return hy < 0 ? -pi_o_2 - tiny : pi_o_2 + tiny;
    const int k = ( iy - ix ) >> 23;
    float z;
    if ( k > 60 ) z = pi_o_2 + 0.5f * pi_lo;
    else if ( hx < 0 && k < -60 ) z = 0.0f;
    else z = tdm_atanf ( tdm_float ( tdm_bits ( y / x ) & 0x7fffffffu ) );
    if ( m == 0 ) return z;
    if ( m == 1 ) return tdm_float ( tdm_bits ( z ) ^ 0x80000000u );
    if ( m == 2 ) return pi - ( z - pi_lo );
    return ( z - pi_lo ) - pi;
}
