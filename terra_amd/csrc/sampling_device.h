// sampling_device.h -- device forms of the reference's 2D samplers and tabulated distributions:
//
//   stratified sampler   reference src/Terra.c:703-723   (TerraSamplerStratified, src/TerraPrivate.h:41-47)
//   Halton sampler       reference src/Terra.c:725-755   (bases 3 and 2)
//   1D / 2D distribution reference src/Terra.c:760-846   (TerraDistribution1D / TerraDistributon2D, src/TerraPrivate.h:86-96)
//
// Unit level only: the reference constructs a stratified or Halton sampler per pixel and never draws from it
// (src/Terra.c:535-548) and nothing calls the distributions, so they are not on the render path here either; they are
// pinned to the compiled reference through the terra_amd_unit_* entry points (tests/golden/samplers.npz).
// Same arithmetic rules as trace_device.h: binary32, the reference's operation order and conversions.
#pragma once
#include <hip/hip_runtime.h>
#include <float.h>
#include "rng.h"

#define SD __device__ __forceinline__

// the clamp both samplers apply: terra_minf ( v, 1.f - terra_Epsilon ), the subtraction done in double (terra_Epsilon is a double literal)
SD float sd_below_one ( float v ) { const float top = ( float ) ( 1.0 - 1e-4 ); return v < top ? v : top; }

// The reference's sampler keeps a pointer to a shared TerraSamplerRandom; the device form owns its stream.
struct StratifiedSampler { Pcg32 rng; int samples, strata, next; float stratum_size; };
SD void stratified_init ( StratifiedSampler& s, uint32_t seed, int strata, int samples ) {
    s.rng.state = 0; s.rng.inc = 1; trng_next ( s.rng ); s.rng.state += seed; trng_next ( s.rng );      // terra_sampler_random_init, src/Terra.c:678-689
    s.strata = strata; s.samples = samples; s.next = 0;
    s.stratum_size = 1.f / ( float ) strata;
}
SD void stratified_next_pair ( StratifiedSampler& s, float& e1, float& e2 ) {
    const uint32_t stratum = ( uint32_t ) s.next / ( uint32_t ) s.samples;
    const uint32_t x = stratum % ( uint32_t ) s.strata, y = stratum / ( uint32_t ) s.strata;
    e1 = sd_below_one ( ( ( float ) x + trng_a_float ( s.rng ) ) * s.stratum_size );
    e2 = sd_below_one ( ( ( float ) y + trng_a_float ( s.rng ) ) * s.stratum_size );
    ++s.next;
}

// terra_radical_inverse: the digits of a in `base`, reversed in integer arithmetic; the denominator is a running float product
SD float radical_inverse ( uint64_t base, uint64_t a ) {
    const float inv_base = 1.f / ( float ) base;
    uint64_t seq = 0;
    float denom = 1.f;
    while ( a ) {
        const uint64_t next = a / base;
        const uint64_t digit = a - next * base;
        seq = seq * base + digit;
        denom *= inv_base;
        a = next;
    }
    return sd_below_one ( ( float ) seq * denom );
}
// element `index` of the Halton sequence: unlike the stratified sampler it has no running state beyond the index, so every lane
// can take its own element
SD void halton_pair ( int index, float& e1, float& e2 ) {
    e1 = radical_inverse ( 3, ( uint64_t ) index );
    e2 = radical_inverse ( 2, ( uint64_t ) index );
}

// A tabulated distribution in HBM: f[n], cdf[n] (normalised running sum), the total. `monotone` = every f >= 0 and the total is
// finite and positive: then the cdf is non-decreasing and "first bucket with e < cdf[i]" is found by bisection instead of the
// reference's linear scan (same bucket); otherwise the scan runs as written.
struct DevDistribution1D { const float* f; const float* cdf; uint32_t n; float integral; uint32_t monotone; };

// one row: running float sum in index order (it cannot be re-associated), then the division; returns the row's total
SD float distribution_row_init ( const float* f, uint32_t n, float* cdf, uint32_t* monotone ) {
    float integral = 0.f; bool mono = true;
    for ( uint32_t i = 0; i < n; ++i ) { const float v = f[i]; mono = mono && v >= 0.f; integral += v; cdf[i] = integral; }
    for ( uint32_t i = 0; i < n; ++i ) cdf[i] /= integral;
    *monotone = ( mono && integral > 0.f && integral <= FLT_MAX ) ? 1u : 0u;
    return integral;
}
// terra_distribution_1d_sample. Not found (the reference asserts): FLT_MAX, pdf / idx untouched.
SD float distribution_sample ( const DevDistribution1D& d, float e, float* pdf, uint32_t* idx ) {
    uint32_t i = d.n;
    if ( d.monotone ) {
        uint32_t lo = 0, hi = d.n;                      // first i with e < cdf[i]
        while ( lo < hi ) { const uint32_t mid = ( lo + hi ) >> 1; if ( e < d.cdf[mid] ) hi = mid; else lo = mid + 1; }
        i = lo;
    } else {
        for ( uint32_t k = 0; k < d.n; ++k ) if ( e < d.cdf[k] ) { i = k; break; }
    }
    if ( i >= d.n ) return FLT_MAX;
    const float curr = d.cdf[i], prev = i ? d.cdf[i - 1] : 0.f;
    if ( pdf ) *pdf = d.f[i] / d.integral;
    if ( idx ) *idx = i;
    float t = e - prev;
    t /= curr - prev;
    return ( ( float ) i + t ) / ( float ) d.n;
}
