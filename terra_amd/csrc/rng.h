// rng.h -- the two per-pixel PCG32 streams of the device path (DESIGN.md "Randomness").
//
// Stream A (camera jitter): the reference's TerraSamplerRandom (reference
// src/Terra.c:678-701): state = 0, inc = 1, step, state += seed32, step; output
// XSH-RR; float = u32 * 2^-32 (the u32->float conversion rounds, so 1.0f is possible).
// Stream B replaces libc rand() (reference src/Terra.c:115): a PCG32 whose output
// is cut to 24 bits, so that "(float)rand() / RAND_MAX" becomes u24 * 2^-24 < 1.
//
// Keying, for framebuffer index pix, frame seed F, K = samples already in the pixel:
//     b      = splitmix64( splitmix64(F + pix) ^ (K * 0x9E3779B97F4A7C15) )
//     seedA  = high 32 bits of b
//     B      = pcg32_srandom( initstate = splitmix64(b ^ 1), initseq = splitmix64(b ^ 2) )
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define TRNG_FN __host__ __device__ inline
#else
#define TRNG_FN static inline
#endif

struct Pcg32 { uint64_t state, inc; };

TRNG_FN uint64_t trng_splitmix64 ( uint64_t z ) {
    z += 0x9E3779B97F4A7C15ull;
    z = ( z ^ ( z >> 30 ) ) * 0xBF58476D1CE4E5B9ull;
    z = ( z ^ ( z >> 27 ) ) * 0x94D049BB133111EBull;
    return z ^ ( z >> 31 );
}

TRNG_FN uint32_t trng_next ( Pcg32& g ) {
    uint64_t old = g.state;
    g.state = old * 6364136223846793005ull + g.inc;
    uint32_t xs = ( uint32_t ) ( ( ( old >> 18 ) ^ old ) >> 27 );
    uint32_t rot = ( uint32_t ) ( old >> 59 );
    return ( xs >> rot ) | ( xs << ( ( 0u - rot ) & 31u ) );
}

TRNG_FN void trng_seed ( Pcg32& g, uint64_t initstate, uint64_t initseq ) {
    g.state = 0;
    g.inc = ( initseq << 1 ) | 1u;
    trng_next ( g );
    g.state += initstate;
    trng_next ( g );
}

struct PixelStreams { Pcg32 a, b; uint32_t seedA; };

TRNG_FN PixelStreams trng_pixel_streams ( uint64_t frame_seed, uint64_t pix, uint64_t samples_so_far ) {
    PixelStreams s;
    uint64_t k = trng_splitmix64 ( trng_splitmix64 ( frame_seed + pix ) ^ ( samples_so_far * 0x9E3779B97F4A7C15ull ) );
    s.seedA = ( uint32_t ) ( k >> 32 );
    s.a.state = 0; s.a.inc = 1;
    trng_next ( s.a );
    s.a.state += s.seedA;
    trng_next ( s.a );
    trng_seed ( s.b, trng_splitmix64 ( k ^ 1ull ), trng_splitmix64 ( k ^ 2ull ) );
    return s;
}

// camera jitter variate in [0,1]
TRNG_FN float trng_a_float ( Pcg32& a ) { return ( float ) trng_next ( a ) * 0x1p-32f; }
// the value "(float)rand() / RAND_MAX" takes: 24 random bits * 2^-24, in [0,1)
TRNG_FN float trng_b_float ( Pcg32& b ) { return ( float ) ( trng_next ( b ) >> 8 ) * 0x1p-24f; }
