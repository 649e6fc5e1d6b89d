/*
 * TEST INFRASTRUCTURE (oracle/). Not part of the product; see oracle/README.md.
 *
 * Per-pixel random streams used by BOTH CPU checkers (the compiled reference
 * in oracle/_ref and the restatement in oracle/terra_oracle.c). The product
 * keeps its own copy of the same definition in terra_amd/csrc/rng.h; DESIGN.md
 * section "Randomness" is the normative text.
 *
 * The reference draws camera jitter from one PCG32 per terra_render() call
 * seeded by time()^&exit (reference src/Terra.c:529-530, :678-701) and every
 * other random number from libc rand() (reference src/Terra.c:115). Neither is
 * reproducible or order-independent. Both are pinned *without editing the
 * reference* (SURVEY.md section 8c): for the pixel with framebuffer index
 * `pix`, frame seed `F` and `K` = samples already accumulated in that pixel,
 *
 *     b        = splitmix64( splitmix64(F + pix) ^ (K * 0x9E3779B97F4A7C15) )
 *     seedA    = (uint32_t)(b >> 32)          -> the value time()^&exit collapses to
 *     streamB  = pcg32_srandom( initstate = splitmix64(b ^ 1), initseq = splitmix64(b ^ 2) )
 *     rand()   = (pcg32_next(streamB) >> 8) << 7          (so (float)rand()/RAND_MAX == u24 * 2^-24 < 1)
 */
#ifndef ORACLE_STREAM_KEY_H
#define ORACLE_STREAM_KEY_H
#include <stdint.h>

#define ORC_DEFAULT_FRAME_SEED 0x5EED0001ull

typedef struct { uint64_t state, inc; } OrcPcg32;

static inline uint64_t orc_splitmix64 ( uint64_t z ) {
    z += 0x9E3779B97F4A7C15ull;
    z = ( z ^ ( z >> 30 ) ) * 0xBF58476D1CE4E5B9ull;
    z = ( z ^ ( z >> 27 ) ) * 0x94D049BB133111EBull;
    return z ^ ( z >> 31 );
}

static inline uint32_t orc_pcg32_next ( OrcPcg32* g ) {
    uint64_t old = g->state;
    g->state = old * 6364136223846793005ull + g->inc;
    uint32_t xs = ( uint32_t ) ( ( ( old >> 18 ) ^ old ) >> 27 );
    uint32_t rot = ( uint32_t ) ( old >> 59 );
    return ( xs >> rot ) | ( xs << ( ( 0u - rot ) & 31u ) );
}

static inline void orc_pcg32_seed ( OrcPcg32* g, uint64_t initstate, uint64_t initseq ) {
    g->state = 0;
    g->inc = ( initseq << 1 ) | 1u;
    orc_pcg32_next ( g );
    g->state += initstate;
    orc_pcg32_next ( g );
}

typedef struct {
    uint32_t seedA;    /* 32-bit seed of the camera-jitter PCG (inc = 1) */
    OrcPcg32 streamB;  /* stream behind rand() */
} OrcPixelStreams;

static inline OrcPixelStreams orc_pixel_streams ( uint64_t frame_seed, uint64_t pix, uint64_t samples_so_far ) {
    OrcPixelStreams s;
    uint64_t b = orc_splitmix64 ( orc_splitmix64 ( frame_seed + pix ) ^ ( samples_so_far * 0x9E3779B97F4A7C15ull ) );
    s.seedA = ( uint32_t ) ( b >> 32 );
    orc_pcg32_seed ( &s.streamB, orc_splitmix64 ( b ^ 1ull ), orc_splitmix64 ( b ^ 2ull ) );
    return s;
}

/* the integer rand() returns; (float)r / 2147483648.f is exact and < 1 */
static inline int orc_rand_from_stream ( OrcPcg32* g ) {
    return ( int ) ( ( orc_pcg32_next ( g ) >> 8 ) << 7 );
}

#endif
