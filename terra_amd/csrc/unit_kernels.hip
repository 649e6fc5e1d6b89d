// unit_kernels.hip -- one-function-per-kernel wrappers over the device functions
// of trace_device.h, so parity tests can pin every stage of the path against the
// oracle and the golden vectors (include/terra_amd.h, "unit-level device entry points").
#include <hip/hip_runtime.h>
#include "trace_device.h"
#include "sampling_device.h"
#include "kernels.h"

#define UNIT_GRID(n) dim3 ( ( ( n ) + 255 ) / 256 ), dim3 ( 256 )

__global__ void k_pcg ( const uint32_t* seeds, int nseeds, int n, float* out ) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if ( i >= nseeds ) return;
    Pcg32 a; a.state = 0; a.inc = 1;
    trng_next ( a ); a.state += seeds[i]; trng_next ( a );
    for ( int j = 0; j < n; ++j ) out[ ( size_t ) i * n + j] = trng_a_float ( a );
}
hipError_t terra_unit_pcg ( const uint32_t* seeds, int nseeds, int n, float* out ) {
    hipLaunchKernelGGL ( k_pcg, UNIT_GRID ( nseeds ), 0, 0, seeds, nseeds, n, out );
    return hipGetLastError();
}

__global__ void k_stream_keys ( uint64_t seed, const uint64_t* pix, const uint64_t* k, int n, uint64_t* out3 ) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if ( i >= n ) return;
    PixelStreams s = trng_pixel_streams ( seed, pix[i], k[i] );
    out3[3 * i] = s.seedA; out3[3 * i + 1] = s.b.state; out3[3 * i + 2] = s.b.inc;
}
hipError_t terra_unit_stream_keys ( uint64_t frame_seed, const uint64_t* pix, const uint64_t* k, int n, uint64_t* out3 ) {
    hipLaunchKernelGGL ( k_stream_keys, UNIT_GRID ( n ), 0, 0, frame_seed, pix, k, n, out3 );
    return hipGetLastError();
}

__global__ void k_ray_aabb ( int n, const float* o, const float* d, const float* boxes, int* hit, float* tmin, float* tmax ) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if ( i >= n ) return;
    Ray r = make_ray ( v3p ( o + 3 * i ), v3p ( d + 3 * i ) );
    float a, b;
    bool h = ray_aabb ( r, v3p ( boxes + 6 * i ), v3p ( boxes + 6 * i + 3 ), &a, &b );
    hit[i] = h ? 1 : 0;
    if ( h ) { tmin[i] = a; tmax[i] = b; }
}
hipError_t terra_unit_ray_aabb ( int n, const float* o, const float* d, const float* boxes, int* hit, float* tmin, float* tmax ) {
    hipLaunchKernelGGL ( k_ray_aabb, UNIT_GRID ( n ), 0, 0, n, o, d, boxes, hit, tmin, tmax );
    return hipGetLastError();
}

__global__ void k_watertight ( int n, const float* o, const float* d, const float* tris, int* hit, float* out8 ) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if ( i >= n ) return;
    Ray r = make_ray ( v3p ( o + 3 * i ), v3p ( d + 3 * i ) );
    RayState s = ray_state_init ( r );
    TriHit h; h.u = h.v = h.w = h.depth = 0.f; h.point = v3 ( 0, 0, 0 );
    bool ok = watertight ( r, s, v3p ( tris + 9 * i ), v3p ( tris + 9 * i + 3 ), v3p ( tris + 9 * i + 6 ), h );
    hit[i] = ok ? 1 : 0;
    float* q = out8 + 8 * i;
    if ( ok ) { q[0] = h.u; q[1] = h.v; q[2] = h.w; q[3] = h.depth; q[4] = h.point.x; q[5] = h.point.y; q[6] = h.point.z; q[7] = 0.f; }
    else { for ( int j = 0; j < 8; ++j ) q[j] = 0.f; }
}
hipError_t terra_unit_watertight ( int n, const float* o, const float* d, const float* tris, int* hit, float* out8 ) {
    hipLaunchKernelGGL ( k_watertight, UNIT_GRID ( n ), 0, 0, n, o, d, tris, hit, out8 );
    return hipGetLastError();
}

__global__ void k_mt ( int n, const float* o, const float* d, const float* tris, int* hit, float* out4 ) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if ( i >= n ) return;
    float t = 0.f; V3 p = v3 ( 0, 0, 0 );
    bool ok = moller_trumbore ( v3p ( o + 3 * i ), v3p ( d + 3 * i ), v3p ( tris + 9 * i ), v3p ( tris + 9 * i + 3 ), v3p ( tris + 9 * i + 6 ), t, p );
    hit[i] = ok ? 1 : 0;
    out4[4 * i] = ok ? t : 0.f; out4[4 * i + 1] = ok ? p.x : 0.f; out4[4 * i + 2] = ok ? p.y : 0.f; out4[4 * i + 3] = ok ? p.z : 0.f;
}
hipError_t terra_unit_moller_trumbore ( int n, const float* o, const float* d, const float* tris, int* hit, float* out4 ) {
    hipLaunchKernelGGL ( k_mt, UNIT_GRID ( n ), 0, 0, n, o, d, tris, hit, out4 );
    return hipGetLastError();
}

// unit kernels read the scene from global memory (MODE 0); LDS only holds stack + leaf list
__device__ __forceinline__ Tracer unit_tracer ( const DevScene& sc, int* lds ) {
    Tracer T; T.sc = sc; T.l_nodes = nullptr; T.l_tris = nullptr; T.l_props = nullptr; T.l_mats = sc.mats; T.l_lights = sc.lights; T.l_area = sc.tri_area; T.lds_nodes = 0; T.lds_tris = 0;
    T.stack = lds + threadIdx.x; T.leaves = lds + ( sc.max_stack < 1 ? 1 : sc.max_stack ) * 256 + threadIdx.x; T.leaf_cap = TERRA_LEAF_CAP_MAX;
    T.stack_lim = 0; T.spill = nullptr; T.spill_cap = 0;
    T.stack_cap = sc.max_stack < 1 ? 1 : sc.max_stack; T.faults = nullptr; T.cull = false; T.fused = false;      // unit level: the reference's traversal decision by decision       // (unit kernels are not built with TERRA_CHECK_BOUNDS)
    return T;
}
__global__ __launch_bounds__ ( 256 ) void k_bvh_traverse ( DevScene sc, int n, const float* o, const float* d, int* found, uint32_t* prim, float* point ) {
    extern __shared__ int lds_stack[];
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if ( i >= n ) return;
    Ray r = make_ray ( v3p ( o + 3 * i ), v3p ( d + 3 * i ) );
    RayState s = ray_state_init ( r );
    Counters c = counters_zero();
    Tracer T = unit_tracer ( sc, lds_stack );
    Closest b = bvh_traverse<0, 0> ( T, r, s, c );
    bool f = b.tri != 0xffffffffu;
    found[i] = f ? 1 : 0;
    prim[i] = f ? ( sc.tris[b.tri].object | ( sc.tris[b.tri].tri_in_object << 8 ) ) : 0u;
    V3 pt = f ? r.o + r.d * b.depth : v3 ( FLT_MAX, FLT_MAX, FLT_MAX );
    point[3 * i] = pt.x; point[3 * i + 1] = pt.y; point[3 * i + 2] = pt.z;
}
static size_t stack_lds ( const DevScene& sc ) { return ( size_t ) ( ( sc.max_stack < 1 ? 1 : sc.max_stack ) + TERRA_LEAF_CAP_MAX ) * 256 * sizeof ( int ); }
hipError_t terra_unit_bvh_traverse ( const DevScene& sc, int n, const float* o, const float* d, int* found, uint32_t* prim, float* point ) {
    hipLaunchKernelGGL ( k_bvh_traverse, UNIT_GRID ( n ), stack_lds ( sc ), 0, sc, n, o, d, found, prim, point );
    return hipGetLastError();
}

// the fast tree's traversal (MODE 2) ray by ray, with the nodes each ray visited: same answers as k_bvh_traverse on any ray, axis-parallel ones included
#define UNIT_FAST_STACK_LDS 32      // entries of the unit kernel's stack in LDS (32 KB per block); a deeper tree spills to `spill` like the render kernels' stacks do
__global__ __launch_bounds__ ( 256 ) void k_bvh_traverse_fast ( DevScene sc, int n, const float* o, const float* d, int* found, uint32_t* prim, float* point, uint32_t* nodes_visited, uint32_t* spill, uint32_t spill_cap, uint32_t lds_entries ) {
    extern __shared__ int lds_stack[];
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if ( i >= n ) return;
    Ray r = make_ray ( v3p ( o + 3 * i ), v3p ( d + 3 * i ) );
    RayState s = ray_state_init ( r );
    Counters c = counters_zero();
    Tracer T; T.sc = sc; T.l_nodes = nullptr; T.l_tris = nullptr; T.l_props = nullptr; T.l_mats = sc.mats; T.l_lights = sc.lights; T.l_area = sc.tri_area; T.lds_nodes = 0; T.lds_tris = 0;
    T.stack = lds_stack + threadIdx.x; T.leaves = T.stack; T.leaf_cap = 0; T.stack_cap = ( int ) lds_entries; T.faults = nullptr; T.cull = false; T.fused = false;
    T.stack_lim = ( uint32_t ) ( uintptr_t ) lds_stack + lds_entries * 1024u; T.spill = spill ? spill + ( size_t ) i * spill_cap : nullptr; T.spill_cap = spill_cap;
    ClosestRanked b = bvh_traverse_fast<1> ( T, r, s, c );
    bool f = b.tri != 0xffffffffu;
    found[i] = f ? 1 : 0;
    prim[i] = f ? ( sc.fast_tris[b.tri].object | ( sc.fast_tris[b.tri].tri_in_object << 8 ) ) : 0u;
    V3 pt = f ? r.o + r.d * b.depth : v3 ( FLT_MAX, FLT_MAX, FLT_MAX );
    point[3 * i] = pt.x; point[3 * i + 1] = pt.y; point[3 * i + 2] = pt.z;
    nodes_visited[i] = c.nodes;
}
hipError_t terra_unit_bvh_traverse_fast ( const DevScene& sc, int n, const float* o, const float* d, int* found, uint32_t* prim, float* point, uint32_t* nodes_visited ) {
    const uint32_t need = ( uint32_t ) ( sc.fast_max_stack < 1 ? 1 : sc.fast_max_stack ), lds_entries = need < UNIT_FAST_STACK_LDS ? need : UNIT_FAST_STACK_LDS, spill_cap = need - lds_entries;
    uint32_t* spill = nullptr;
    if ( spill_cap ) { const hipError_t e = hipMalloc ( ( void** ) &spill, ( size_t ) ( ( n + 255 ) / 256 ) * 256 * spill_cap * sizeof ( uint32_t ) ); if ( e != hipSuccess ) return e; }
    hipLaunchKernelGGL ( k_bvh_traverse_fast, UNIT_GRID ( n ), ( size_t ) lds_entries * 256 * sizeof ( int ), 0, sc, n, o, d, found, prim, point, nodes_visited, spill, spill_cap, lds_entries );
    hipError_t e = hipGetLastError();
    if ( spill ) { const hipError_t e2 = hipDeviceSynchronize(); if ( e == hipSuccess ) e = e2; ( void ) hipFree ( spill ); }
    return e;
}

__device__ void surface_to_floats ( const DevScene& sc, const Surface& sf, uint32_t object, float* q ) {
    Basis bs = make_basis ( sf.normal );
    q[0] = bs.r0[0]; q[1] = bs.r0[1]; q[2] = bs.r0[2]; q[3] = 0.f;
    q[4] = bs.r1[0]; q[5] = bs.r1[1]; q[6] = bs.r1[2]; q[7] = 0.f;
    q[8] = bs.r2[0]; q[9] = bs.r2[1]; q[10] = bs.r2[2]; q[11] = 0.f;
    q[12] = 0.f; q[13] = 0.f; q[14] = 0.f; q[15] = 1.f;
    q[16] = sf.normal.x; q[17] = sf.normal.y; q[18] = sf.normal.z;
    q[19] = sf.emissive.x; q[20] = sf.emissive.y; q[21] = sf.emissive.z;
    q[22] = sc.mats[object].ior;
    for ( int a = 0; a < 8; ++a ) for ( int k = 0; k < 3; ++k ) q[23 + 3 * a + k] = sc.mats[object].attributes[a][k];
}
__global__ __launch_bounds__ ( 256 ) void k_raycast ( DevScene sc, int n, const float* o, const float* d, int* obj, int* tri, float* point, float* surface47 ) {
    extern __shared__ int lds_stack[];
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if ( i >= n ) return;
    Ray r = make_ray ( v3p ( o + 3 * i ), v3p ( d + 3 * i ) );
    Counters c = counters_zero();
    Surface sf;
    Tracer T = unit_tracer ( sc, lds_stack );
    RaycastResult h = scene_raycast<0, 0, TERRA_KINDS_ALL> ( T, r, sf, c );
    obj[i] = h.hit ? ( int ) h.object : -1;
    tri[i] = h.hit ? ( int ) h.tri_in_object : 0;
    point[3 * i] = h.point.x; point[3 * i + 1] = h.point.y; point[3 * i + 2] = h.point.z;
    float* q = surface47 + 47 * ( size_t ) i;
    if ( h.hit ) surface_to_floats ( sc, sf, h.object, q );
    else for ( int j = 0; j < 47; ++j ) q[j] = 0.f;
}
hipError_t terra_unit_raycast ( const DevScene& sc, int n, const float* o, const float* d, int* obj, int* tri, float* point, float* surface47 ) {
    hipLaunchKernelGGL ( k_raycast, UNIT_GRID ( n ), stack_lds ( sc ), 0, sc, n, o, d, obj, tri, point, surface47 );
    return hipGetLastError();
}

template <int I>
__global__ __launch_bounds__ ( 256 ) void k_trace ( DevScene sc, uint32_t bounces, int n, const float* o, const float* d, const uint64_t* stateB, const uint64_t* incB, float* radiance, uint32_t* rand_calls ) {
    extern __shared__ int lds_stack[];
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if ( i >= n ) return;
    Ray r = make_ray ( v3p ( o + 3 * i ), v3p ( d + 3 * i ) );
    Pcg32 b; b.state = stateB[i]; b.inc = incB[i];
    Counters c = counters_zero();
    Tracer T = unit_tracer ( sc, lds_stack );
    V3 L = trace_path<I, 2, 0, TERRA_KINDS_ALL> ( T, r, bounces, b, c );
    radiance[3 * i] = L.x; radiance[3 * i + 1] = L.y; radiance[3 * i + 2] = L.z;
    rand_calls[i] = c.rand_calls;
}
hipError_t terra_unit_trace ( const DevScene& sc, int integrator, uint32_t bounces, int n, const float* o, const float* d,
                              const uint64_t* stateB, const uint64_t* incB, float* radiance, uint32_t* rand_calls ) {
#define TRACE_CASE(I) case I: hipLaunchKernelGGL ( k_trace<I>, UNIT_GRID ( n ), stack_lds ( sc ), 0, sc, bounces, n, o, d, stateB, incB, radiance, rand_calls ); break;
    switch ( integrator ) {
        TRACE_CASE ( 0 ) TRACE_CASE ( 1 ) TRACE_CASE ( 2 ) TRACE_CASE ( 3 ) TRACE_CASE ( 4 ) TRACE_CASE ( 5 ) TRACE_CASE ( 6 )
        default: return hipErrorInvalidValue;
    }
#undef TRACE_CASE
    return hipGetLastError();
}

__global__ void k_bsdf ( int kind, int n, float* surfaces47, const float* e3, const float* wo3, float* wi3, float* pdf, float* f3 ) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if ( i >= n ) return;
    float* q = surfaces47 + 47 * ( size_t ) i;
    Surface sf;
    // the tangent frame is a function of the normal (terra_f4x4_basis); q[0..15] is not read
    sf.normal = v3p ( q + 16 ); sf.emissive = v3p ( q + 19 );
    for ( int a = 0; a < 4; ++a ) sf.attr[a] = v3p ( q + 23 + 3 * a );
    sf.bsdf = kind; sf.ior = q[22];
    V3 wo = v3p ( wo3 + 3 * i );
    V3 wi = bsdf_sample<TERRA_KINDS_ALL> ( sf, e3[3 * i], e3[3 * i + 1], e3[3 * i + 2], wo, azimuth_none() );
    float p = bsdf_pdf<TERRA_KINDS_ALL> ( sf, wi, wo );
    V3 f = bsdf_eval<TERRA_KINDS_ALL> ( sf, wi, wo );
    wi3[3 * i] = wi.x; wi3[3 * i + 1] = wi.y; wi3[3 * i + 2] = wi.z;
    pdf[i] = p;
    f3[3 * i] = f.x; f3[3 * i + 1] = f.y; f3[3 * i + 2] = f.z;
    q[23 + 6] = sf.attr[2].x; q[23 + 7] = sf.attr[2].y; q[23 + 8] = sf.attr[2].z; q[23 + 9] = sf.attr[3].x;
}
hipError_t terra_unit_bsdf ( int kind, int n, float* surfaces47, const float* e3, const float* wo3, float* wi3, float* pdf, float* f3 ) {
    hipLaunchKernelGGL ( k_bsdf, UNIT_GRID ( n ), 0, 0, kind, n, surfaces47, e3, wo3, wi3, pdf, f3 );
    return hipGetLastError();
}

__global__ void k_camera ( DevRenderParams p, int n, const uint32_t* xy2, const float* r2, float* dirs3 ) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if ( i >= n ) return;
    V3 d = camera_sample ( p, xy2[2 * i], xy2[2 * i + 1], r2[2 * i], r2[2 * i + 1] );
    dirs3[3 * i] = d.x; dirs3[3 * i + 1] = d.y; dirs3[3 * i + 2] = d.z;
}
hipError_t terra_unit_camera ( const DevRenderParams& p, int n, const uint32_t* xy2, const float* r2, float* dirs3 ) {
    hipLaunchKernelGGL ( k_camera, UNIT_GRID ( n ), 0, 0, p, n, xy2, r2, dirs3 );
    return hipGetLastError();
}

__global__ void k_tonemap ( int op, float gamma, int n, float* c3 ) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if ( i >= n ) return;
    V3 c = tonemap ( v3p ( c3 + 3 * i ), op, gamma );
    c3[3 * i] = c.x; c3[3 * i + 1] = c.y; c3[3 * i + 2] = c.z;
}
hipError_t terra_unit_tonemap ( int op, float gamma, int n, float* colors3 ) {
    hipLaunchKernelGGL ( k_tonemap, UNIT_GRID ( n ), 0, 0, op, gamma, n, colors3 );
    return hipGetLastError();
}

__global__ void k_math ( int fn, int n, const float* x, const float* y, float* out ) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if ( i >= n ) return;
    float r;
    switch ( fn ) {
        case 0: r = tdm_sinf ( x[i] ); break;
        case 1: r = tdm_cosf ( x[i] ); break;
        case 2: r = tdm_powf ( x[i], y[i] ); break;
        case 3: r = tdm_acosf ( x[i] ); break;
        default: r = tdm_atan2f ( x[i], y[i] ); break;
    }
    out[i] = r;
}
hipError_t terra_unit_math ( int fn, int n, const float* x, const float* y, float* out ) {
    hipLaunchKernelGGL ( k_math, UNIT_GRID ( n ), 0, 0, fn, n, x, y, out );
    return hipGetLastError();
}


// ---- SURVEY.md 8f N4, unit level: samplers and distributions (sampling_device.h) ---------------------------------------------
// one sampler per seed; a stratified sampler's n pairs are inherently sequential (they share one random stream)
__global__ void k_stratified ( const uint32_t* seeds, int nseeds, int strata, int samples, int n, float* out2 ) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if ( i >= nseeds ) return;
    StratifiedSampler s; stratified_init ( s, seeds[i], strata, samples );
    for ( int j = 0; j < n; ++j ) { float a, b; stratified_next_pair ( s, a, b ); out2[ ( ( size_t ) i * n + j ) * 2] = a; out2[ ( ( size_t ) i * n + j ) * 2 + 1] = b; }
}
hipError_t terra_unit_stratified ( const uint32_t* seeds, int nseeds, int strata, int samples, int n, float* out2 ) {
    hipLaunchKernelGGL ( k_stratified, UNIT_GRID ( nseeds ), 0, 0, seeds, nseeds, strata, samples, n, out2 );
    return hipGetLastError();
}
__global__ void k_halton ( int first, int n, float* out2 ) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if ( i >= n ) return;
    float a, b; halton_pair ( first + i, a, b );
    out2[2 * i] = a; out2[2 * i + 1] = b;
}
hipError_t terra_unit_halton ( int first, int n, float* out2 ) {
    hipLaunchKernelGGL ( k_halton, UNIT_GRID ( n ), 0, 0, first, n, out2 );
    return hipGetLastError();
}
// rows of a 2D table (ny = 1: a 1D distribution): one lane per row, each a sequential float sum
__global__ void k_dist_rows ( const float* f, uint32_t nx, uint32_t ny, float* cdf, float* integrals, uint32_t* monotone ) {
    uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if ( r >= ny ) return;
    integrals[r] = distribution_row_init ( f + ( size_t ) nx * r, nx, cdf + ( size_t ) nx * r, monotone + r );
}
// the marginal over the rows' totals (reference src/Terra.c:817-833): one lane, sequential; slot ny of `integrals` / `monotone` receives its total / flag
__global__ void k_dist_marginal ( uint32_t ny, float* integrals, float* mcdf, uint32_t* monotone ) {
    if ( blockIdx.x || threadIdx.x ) return;
    integrals[ny] = distribution_row_init ( integrals, ny, mcdf, monotone + ny );
}
__global__ void k_dist_sample_1d ( const float* f, const float* cdf, uint32_t n, const float* integral, const uint32_t* monotone, const float* e, int m, float* x, float* pdf, uint32_t* idx ) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if ( i >= m ) return;
    DevDistribution1D d = { f, cdf, n, integral[0], monotone[0] };
    float p = 0.f; uint32_t k = 0;
    x[i] = distribution_sample ( d, e[i], &p, &k );
    pdf[i] = p; idx[i] = k;
}
__global__ void k_dist_sample_2d ( const float* f, const float* cdf, uint32_t nx, uint32_t ny, const float* integrals, const float* mcdf, const uint32_t* monotone, const float* e12, int m, float* xy2, float* pdf ) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if ( i >= m ) return;
    DevDistribution1D marginal = { integrals, mcdf, ny, integrals[ny], monotone[ny] };
    float p0 = 0.f, p1 = 0.f; uint32_t row = 0;
    const float s1 = distribution_sample ( marginal, e12[2 * i], &p0, &row );
    if ( s1 == FLT_MAX ) { xy2[2 * i] = xy2[2 * i + 1] = FLT_MAX; pdf[i] = 0.f; return; }
    DevDistribution1D cond = { f + ( size_t ) nx * row, cdf + ( size_t ) nx * row, nx, integrals[row], monotone[row] };
    const float s2 = distribution_sample ( cond, e12[2 * i + 1], &p1, nullptr );
    xy2[2 * i] = s1; xy2[2 * i + 1] = s2; pdf[i] = p0 * p1;
}
hipError_t terra_unit_distribution_1d ( const float* f, uint32_t n, float* cdf, float* integral, uint32_t* monotone, const float* e, int m, float* x, float* pdf, uint32_t* idx ) {
    hipLaunchKernelGGL ( k_dist_rows, dim3 ( 1 ), dim3 ( 64 ), 0, 0, f, n, 1u, cdf, integral, monotone );
    if ( m > 0 ) hipLaunchKernelGGL ( k_dist_sample_1d, UNIT_GRID ( m ), 0, 0, f, cdf, n, integral, monotone, e, m, x, pdf, idx );
    return hipGetLastError();
}
hipError_t terra_unit_distribution_2d ( const float* f, uint32_t nx, uint32_t ny, float* cdf, float* integrals, float* mcdf, uint32_t* monotone, const float* e12, int m, float* xy2, float* pdf ) {
    hipLaunchKernelGGL ( k_dist_rows, UNIT_GRID ( ny ), 0, 0, f, nx, ny, cdf, integrals, monotone );
    hipLaunchKernelGGL ( k_dist_marginal, dim3 ( 1 ), dim3 ( 64 ), 0, 0, ny, integrals, mcdf, monotone );
    if ( m > 0 ) hipLaunchKernelGGL ( k_dist_sample_2d, UNIT_GRID ( m ), 0, 0, f, cdf, nx, ny, integrals, mcdf, monotone, e12, m, xy2, pdf );
    return hipGetLastError();
}

// ---- DevScene::sincos24: (cos, sin) of 2 * terra_PI * (k * 2^-24) for every 24-bit k, each entry by tdm_sincosf_pair itself (trace_device.h azimuth_fetch) ----
__global__ __launch_bounds__ ( 256 ) void terra_sincos24_kernel ( float2* table ) {
    const uint32_t k = blockIdx.x * 256u + threadIdx.x;          // grid = 2^24 / 256 blocks
    float sn, cs;
    tdm_sincosf_pair ( 2 * TERRA_PI_F * ( ( float ) k * 0x1p-24f ), sn, cs );
    table[k] = make_float2 ( cs, sn );
}
hipError_t terra_fill_sincos24 ( float2* table, hipStream_t stream ) {
    hipLaunchKernelGGL ( terra_sincos24_kernel, dim3 ( ( 1u << 24 ) / 256u ), dim3 ( 256 ), 0, stream, table );
    return hipGetLastError();
}
