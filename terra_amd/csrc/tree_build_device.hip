// tree_build_device.hip -- the fast tree built on the GPU (SURVEY.md 8f N3 "GPU-side BVH build"; the traversal it feeds
// replaces reference src/TerraBVH.c:250-310, the build it replaces is src/TerraBVH.c:128-244).
//
// A linear BVH (Karras 2012, "Maximizing parallelism in the construction of BVHs, octrees and k-d trees"): 30-bit Morton
// codes of the box centres made unique by the triangle index, one radix sort (hipCUB), every inner node's range and split
// found independently from the sorted keys, boxes fitted bottom-up with one atomic counter per node, subtrees of at most
// TERRA_LBVH_LEAF_MAX triangles collapsed into leaves, the surviving nodes compacted by a prefix sum. The result has the
// layout the host builder (tree_build.cpp, binned SAH) produces -- DevNode array with the root at 0, leaf word =
// DEV_CHILD_LEAF | (count - 1) << 27 | first, triangle soup in leaf order with the reference visit rank in DevTri::pad --
// so terra_render_kernel<.., MODE 2, ..> traverses either. Boxes are the same +-1e-4 triangle boxes as the host's and
// the reference's (src/Terra.c:982-996), inner boxes plain unions: the containment the culling relies on holds by construction.
// An LBVH is built in milliseconds but traverses more nodes per ray than the SAH tree (DESIGN.md "Fast tree" has both numbers);
// terra_amd_set_tree_builder() chooses.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>
#include <cfloat>
#include "dev_types.h"
#include "kernels.h"

#ifndef TERRA_LBVH_LEAF_MAX
#define TERRA_LBVH_LEAF_MAX 2      // (hall at 97k / 289k / 650k triangles, 16 spp: 4 -> 1,064 / 970 / 897 Msamples/s, 2 -> 1,150 / 1,037 / 961, 1 -> 1,067 / 987 / 790)
#endif

namespace {
struct Box { float mn[3], mx[3]; };

__device__ __forceinline__ Box tri_box ( const DevTri& t, float extra ) {          // bvh::grow_by_triangle (tree_build.cpp): extent grown by 1e-4, the add in double; `extra` more on every side (reachability mode, scene_host.cpp)
    Box b;
    const float* a = t.a; const float* bb = t.b; const float* c = t.c;
    #pragma unroll
    for ( int k = 0; k < 3; ++k ) {
        const float lo = fminf ( a[k], fminf ( bb[k], c[k] ) ), hi = fmaxf ( a[k], fmaxf ( bb[k], c[k] ) );
        b.mn[k] = ( float ) ( ( double ) lo - 1e-4 ) - extra; b.mx[k] = ( float ) ( ( double ) hi + 1e-4 ) + extra;
    }
    return b;
}
__device__ __forceinline__ uint32_t ordered ( float f ) { uint32_t u = __float_as_uint ( f ); return ( u & 0x80000000u ) ? ~u : ( u | 0x80000000u ); }     // monotone float -> uint
__device__ __forceinline__ float unordered ( uint32_t u ) { return __uint_as_float ( ( u & 0x80000000u ) ? ( u & 0x7fffffffu ) : ~u ); }

__global__ void k_boxes ( const DevTri* tris, uint32_t n, float extra, Box* boxes, uint32_t* bounds6 ) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if ( i >= n ) return;
    Box b = tri_box ( tris[i], extra );
    boxes[i] = b;
    #pragma unroll
    for ( int k = 0; k < 3; ++k ) {          // bounds of the box CENTRES (what the Morton grid spans)
        const float c = 0.5f * ( b.mn[k] + b.mx[k] );
        atomicMin ( &bounds6[k], ordered ( c ) ); atomicMax ( &bounds6[3 + k], ordered ( c ) );
    }
}
__device__ __forceinline__ uint32_t spread10 ( uint32_t v ) {      // 10 bits -> every third bit
    v = ( v * 0x00010001u ) & 0xFF0000FFu; v = ( v * 0x00000101u ) & 0x0F00F00Fu; v = ( v * 0x00000011u ) & 0xC30C30C3u; v = ( v * 0x00000005u ) & 0x49249249u;
    return v;
}
__global__ void k_keys ( const Box* boxes, uint32_t n, const uint32_t* bounds6, unsigned long long* keys ) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if ( i >= n ) return;
    uint32_t q[3];
    #pragma unroll
    for ( int k = 0; k < 3; ++k ) {
        const float lo = unordered ( bounds6[k] ), hi = unordered ( bounds6[3 + k] ), c = 0.5f * ( boxes[i].mn[k] + boxes[i].mx[k] );
        const float ext = hi - lo;
        float t = ext > 0.f ? ( c - lo ) / ext : 0.f;
        t = fminf ( fmaxf ( t * 1024.f, 0.f ), 1023.f );
        q[k] = ( uint32_t ) t;
    }
    const uint32_t m = ( spread10 ( q[0] ) << 2 ) | ( spread10 ( q[1] ) << 1 ) | spread10 ( q[2] );
    keys[i] = ( ( unsigned long long ) m << 32 ) | i;              // the index makes every key unique
}
__device__ __forceinline__ int delta ( const unsigned long long* keys, int n, int i, int j ) {
    if ( j < 0 || j >= n ) return -1;
    return __clzll ( ( long long ) ( keys[i] ^ keys[j] ) );
}
// Karras 2012, section 4: inner node i covers [first, last] and splits after gamma. Children are encoded as inner index, or ~leaf.
__global__ void k_hierarchy ( const unsigned long long* keys, int n, int2* child, int2* range, int* parent_inner, int* parent_leaf ) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if ( i >= n - 1 ) return;
    const int d = delta ( keys, n, i, i + 1 ) - delta ( keys, n, i, i - 1 ) >= 0 ? 1 : -1;
    const int dmin = delta ( keys, n, i, i - d );
    int lmax = 2;
    while ( delta ( keys, n, i, i + lmax * d ) > dmin ) lmax <<= 1;
    int l = 0;
    for ( int t = lmax >> 1; t >= 1; t >>= 1 ) if ( delta ( keys, n, i, i + ( l + t ) * d ) > dmin ) l += t;
    const int j = i + l * d;
    const int dnode = delta ( keys, n, i, j );
    int s = 0, t = l;
    do { t = ( t + 1 ) >> 1; if ( delta ( keys, n, i, i + ( s + t ) * d ) > dnode ) s += t; } while ( t > 1 );
    const int gamma = i + s * d + ( d < 0 ? d : 0 );
    const int first = i < j ? i : j, last = i < j ? j : i;
    const int left = first == gamma ? ~gamma : gamma, right = last == gamma + 1 ? ~ ( gamma + 1 ) : gamma + 1;
    child[i] = make_int2 ( left, right ); range[i] = make_int2 ( first, last );
    if ( left < 0 ) parent_leaf[~left] = i; else parent_inner[left] = i;
    if ( right < 0 ) parent_leaf[~right] = i; else parent_inner[right] = i;
    if ( i == 0 ) parent_inner[0] = -1;
}
__device__ __forceinline__ Box unite ( const Box& a, const Box& b ) {
    Box r;
    #pragma unroll
    for ( int k = 0; k < 3; ++k ) { r.mn[k] = fminf ( a.mn[k], b.mn[k] ); r.mx[k] = fmaxf ( a.mx[k], b.mx[k] ); }
    return r;
}
// bottom-up fit: a leaf walks towards the root; at every inner node the first arrival stops, the second unites the children
__global__ void k_fit ( const unsigned long long* keys, const Box* prim_boxes, int n, const int2* child, const int* parent_inner, const int* parent_leaf,
                        Box* inner_boxes, Box* leaf_boxes, unsigned int* arrived ) {
    int k = blockIdx.x * blockDim.x + threadIdx.x;
    if ( k >= n ) return;
    leaf_boxes[k] = prim_boxes[ ( uint32_t ) keys[k]];
    __threadfence();
    int p = parent_leaf[k];
    while ( p >= 0 ) {
        if ( atomicAdd ( &arrived[p], 1u ) == 0u ) return;
        __threadfence();
        const int2 c = child[p];
        const Box a = c.x < 0 ? leaf_boxes[~c.x] : inner_boxes[c.x], b = c.y < 0 ? leaf_boxes[~c.y] : inner_boxes[c.y];
        inner_boxes[p] = unite ( a, b );
        __threadfence();
        p = parent_inner[p];
    }
}
// an inner node survives when its range holds more triangles than a leaf may (the root always survives)
__global__ void k_keep ( const int2* range, int n, uint32_t* keep ) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if ( i >= n - 1 ) return;
    keep[i] = ( i == 0 || range[i].y - range[i].x + 1 > TERRA_LBVH_LEAF_MAX ) ? 1u : 0u;
}
__device__ __forceinline__ void put_child ( DevNode& nd, int slot, const Box& b, uint32_t word ) {
    float* mn = slot == 0 ? nd.min0 : nd.min1; float* mx = slot == 0 ? nd.max0 : nd.max1;
    mn[0] = b.mn[0]; mn[1] = b.mn[1]; mn[2] = b.mn[2]; mx[0] = b.mx[0]; mx[1] = b.mx[1]; mx[2] = b.mx[2];
    nd.child[slot] = word; nd.prim[slot] = 0;
}
__global__ void k_emit ( int n, const int2* child, const int2* range, const int* parent_inner, const uint32_t* keep, const uint32_t* new_index,
                         const Box* inner_boxes, const Box* leaf_boxes, DevNode* out, int* max_depth ) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if ( i >= n - 1 || !keep[i] ) return;
    DevNode nd;
    const int2 c = child[i];
    const int cs[2] = { c.x, c.y };
    #pragma unroll
    for ( int slot = 0; slot < 2; ++slot ) {
        const int ch = cs[slot];
        if ( ch < 0 ) put_child ( nd, slot, leaf_boxes[~ch], DEV_CHILD_LEAF | ( uint32_t ) ~ch );                                   // a single triangle
        else if ( !keep[ch] ) put_child ( nd, slot, inner_boxes[ch], DEV_CHILD_LEAF | ( ( uint32_t ) ( range[ch].y - range[ch].x ) << 27 ) | ( uint32_t ) range[ch].x );      // collapsed subtree
        else put_child ( nd, slot, inner_boxes[ch], new_index[ch] );
    }
    out[new_index[i]] = nd;
    int depth = 1;                                           // kept ancestors are exactly the ancestors (a kept node's parent is kept)
    for ( int p = parent_inner[i]; p >= 0; p = parent_inner[p] ) ++depth;
    atomicMax ( max_depth, depth );
}
__global__ void k_soup ( const unsigned long long* keys, const DevTri* tris, const uint32_t* rank, uint32_t n, DevTri* out ) {
    uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if ( k >= n ) return;
    const uint32_t src = ( uint32_t ) keys[k];
    DevTri t = tris[src]; t.pad = rank[src];
    out[k] = t;
}
struct Scratch {
    void* p = nullptr;
    ~Scratch() { if ( p ) ( void ) hipFree ( p ); }
};
} // namespace

#define TB_TRY(expr) do { hipError_t e_ = ( expr ); if ( e_ != hipSuccess ) return e_; } while ( 0 )

// tris / rank / out_nodes (capacity n - 1) / out_tris (capacity n) are device pointers; n > TERRA_LBVH_LEAF_MAX.
hipError_t terra_build_fast_tree_device ( const DevTri* tris, const uint32_t* rank, uint32_t n, float extra_margin, DevNode* out_nodes, DevTri* out_tris, uint32_t* n_nodes_out, int* max_stack_out, hipStream_t stream ) {
    if ( n <= TERRA_LBVH_LEAF_MAX || n > 0x07ffffffu ) return hipErrorInvalidValue;
    const size_t N = n;
    auto al = [] ( size_t v ) { return ( v + 255 ) & ~size_t ( 255 ); };
    size_t off = 0;
    const size_t o_boxes = off; off = al ( off + N * sizeof ( Box ) );
    const size_t o_keys0 = off; off = al ( off + N * 8 );
    const size_t o_keys1 = off; off = al ( off + N * 8 );
    const size_t o_child = off; off = al ( off + N * sizeof ( int2 ) );
    const size_t o_range = off; off = al ( off + N * sizeof ( int2 ) );
    const size_t o_pin = off; off = al ( off + N * 4 );
    const size_t o_plf = off; off = al ( off + N * 4 );
    const size_t o_ibox = off; off = al ( off + N * sizeof ( Box ) );
    const size_t o_lbox = off; off = al ( off + N * sizeof ( Box ) );
    const size_t o_arr = off; off = al ( off + N * 4 );
    const size_t o_keep = off; off = al ( off + N * 4 );
    const size_t o_idx = off; off = al ( off + N * 4 );
    const size_t o_misc = off; off = al ( off + 64 );               // bounds6, max depth, kept count
    size_t sort_bytes = 0, scan_bytes = 0;
    TB_TRY ( hipcub::DeviceRadixSort::SortKeys ( nullptr, sort_bytes, ( const unsigned long long* ) nullptr, ( unsigned long long* ) nullptr, ( int ) n, 0, 62, stream ) );
    TB_TRY ( hipcub::DeviceScan::ExclusiveSum ( nullptr, scan_bytes, ( const uint32_t* ) nullptr, ( uint32_t* ) nullptr, ( int ) n, stream ) );
    const size_t o_tmp = off; off = al ( off + ( sort_bytes > scan_bytes ? sort_bytes : scan_bytes ) );
    Scratch sc;
    TB_TRY ( hipMalloc ( &sc.p, off ) );
    char* base = ( char* ) sc.p;
    Box* boxes = ( Box* ) ( base + o_boxes ); unsigned long long* keys0 = ( unsigned long long* ) ( base + o_keys0 ); unsigned long long* keys = ( unsigned long long* ) ( base + o_keys1 );
    int2* child = ( int2* ) ( base + o_child ); int2* range = ( int2* ) ( base + o_range ); int* pin = ( int* ) ( base + o_pin ); int* plf = ( int* ) ( base + o_plf );
    Box* ibox = ( Box* ) ( base + o_ibox ); Box* lbox = ( Box* ) ( base + o_lbox ); unsigned int* arrived = ( unsigned int* ) ( base + o_arr );
    uint32_t* keep = ( uint32_t* ) ( base + o_keep ); uint32_t* new_index = ( uint32_t* ) ( base + o_idx ); uint32_t* misc = ( uint32_t* ) ( base + o_misc );
    const uint32_t init[8] = { 0xffffffffu, 0xffffffffu, 0xffffffffu, 0u, 0u, 0u, 0u, 0u };        // min x3, max x3, max depth, (spare)
    TB_TRY ( hipMemcpyAsync ( misc, init, sizeof init, hipMemcpyHostToDevice, stream ) );
    TB_TRY ( hipMemsetAsync ( arrived, 0, N * 4, stream ) );
    TB_TRY ( hipMemsetAsync ( keep, 0, N * 4, stream ) );
    const dim3 blk ( 256 ), grd ( ( n + 255 ) / 256 );
    hipLaunchKernelGGL ( k_boxes, grd, blk, 0, stream, tris, n, extra_margin, boxes, misc );
    hipLaunchKernelGGL ( k_keys, grd, blk, 0, stream, boxes, n, misc, keys0 );
    TB_TRY ( hipcub::DeviceRadixSort::SortKeys ( base + o_tmp, sort_bytes, keys0, keys, ( int ) n, 0, 62, stream ) );
    hipLaunchKernelGGL ( k_hierarchy, grd, blk, 0, stream, keys, ( int ) n, child, range, pin, plf );
    hipLaunchKernelGGL ( k_fit, grd, blk, 0, stream, keys, boxes, ( int ) n, child, pin, plf, ibox, lbox, arrived );
    hipLaunchKernelGGL ( k_keep, grd, blk, 0, stream, range, ( int ) n, keep );
    TB_TRY ( hipcub::DeviceScan::ExclusiveSum ( base + o_tmp, scan_bytes, keep, new_index, ( int ) n, stream ) );
    hipLaunchKernelGGL ( k_emit, grd, blk, 0, stream, ( int ) n, child, range, pin, keep, new_index, ibox, lbox, out_nodes, ( int* ) ( misc + 6 ) );
    hipLaunchKernelGGL ( k_soup, grd, blk, 0, stream, keys, tris, rank, n, out_tris );
    TB_TRY ( hipGetLastError() );
    uint32_t tail[2] = { 0, 0 }; uint32_t depth = 0;
    TB_TRY ( hipMemcpyAsync ( &tail[0], new_index + ( n - 1 ), 4, hipMemcpyDeviceToHost, stream ) );      // exclusive sum at n - 1 (keep[n - 1] = 0) = number of kept nodes
    TB_TRY ( hipMemcpyAsync ( &depth, misc + 6, 4, hipMemcpyDeviceToHost, stream ) );
    TB_TRY ( hipStreamSynchronize ( stream ) );
    *n_nodes_out = tail[0];
    *max_stack_out = ( int ) depth + 1;             // one pending (far) child per level + the node in hand when a lane leaves the loop (as fastbvh::build)
    return hipSuccess;
}
