// trace_device.h -- device functions of the hot path (gfx950).
//
//   camera sample            reference src/Terra.c:1783-1799
//   slab test                reference src/Terra.c:851-878
//   watertight ray/triangle  reference src/TerraGeometry.c:98-138, 159-260
//   Moeller-Trumbore         reference src/Terra.c:880-922 (unit level only)
//   BVH stack traversal      reference src/TerraBVH.c:250-310
//   raycast + surface init   reference src/Terra.c:1623-1657, 1726-1764, TerraMath.inl:251-272
//   diffuse / Phong presets  reference src/TerraPresets.c:34-146
//   integrators              reference src/Terra.c:1099-1587
//   bounce loop              reference src/Terra.c:1039-1097
//   tonemap                  reference src/Terra.c:578-627, 1815-1828
//
// Arithmetic rules (DESIGN.md "Bit-faithful arithmetic"): binary32 everywhere the
// reference is binary32, the reference's double promotions kept, operation order
// kept, no FMA contraction (-ffp-contract=off), IEEE division and square root,
// compare-select min/max where a NaN could reach them.
#pragma once
#include <hip/hip_runtime.h>
#include <float.h>
#include "dev_types.h"
#include "dev_math.h"
#include "rng.h"
#include "sampling_device.h"

#define TD __device__ __forceinline__

struct V3 { float x, y, z; };

TD V3 v3 ( float x, float y, float z ) { V3 r; r.x = x; r.y = y; r.z = z; return r; }
TD V3 v3p ( const float* p ) { return v3 ( p[0], p[1], p[2] ); }
TD V3 operator+ ( V3 a, V3 b ) { return v3 ( a.x + b.x, a.y + b.y, a.z + b.z ); }
TD V3 operator- ( V3 a, V3 b ) { return v3 ( a.x - b.x, a.y - b.y, a.z - b.z ); }
TD V3 operator* ( V3 a, float s ) { return v3 ( a.x * s, a.y * s, a.z * s ); }
TD V3 had ( V3 a, V3 b ) { return v3 ( a.x * b.x, a.y * b.y, a.z * b.z ); }
TD V3 neg ( V3 a ) { return v3 ( -a.x, -a.y, -a.z ); }
TD float dot ( V3 a, V3 b ) { return a.x * b.x + a.y * b.y + a.z * b.z; }
TD V3 cross ( V3 a, V3 b ) { return v3 ( a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x ); }
TD float length ( V3 a ) { return sqrtf ( a.x * a.x + a.y * a.y + a.z * a.z ); }
TD V3 normalize ( V3 a ) { float l = length ( a ); return v3 ( a.x / l, a.y / l, a.z / l ); }
// compare-selects: exactly "a < b ? a : b" / "a > b ? a : b" (NaN-order sensitive)
TD float sel_min ( float a, float b ) { return a < b ? a : b; }
TD float sel_max ( float a, float b ) { return a > b ? a : b; }
TD float pick ( V3 a, int i ) { return i == 0 ? a.x : ( i == 1 ? a.y : a.z ); }

// columns of the shading basis: tangent, normal, bitangent. Stored by rows as the reference does.
struct Basis { float r0[3], r1[3], r2[3]; };
TD V3 basis_apply ( const Basis& m, V3 v ) {
    return v3 ( m.r0[0] * v.x + m.r0[1] * v.y + m.r0[2] * v.z,
                m.r1[0] * v.x + m.r1[1] * v.y + m.r1[2] * v.z,
                m.r2[0] * v.x + m.r2[1] * v.y + m.r2[2] * v.z );
}
TD Basis make_basis ( V3 n ) {
    V3 t;
    if ( fabsf ( n.x ) > fabsf ( n.y ) ) {
        float k = sqrtf ( n.x * n.x + n.z * n.z );
        t = v3 ( n.z * k, 0.f * k, -n.x * k );
    } else {
        float k = sqrtf ( n.y * n.y + n.z * n.z );
        t = v3 ( 0.f * k, -n.z * k, n.y * k );
    }
    V3 b = cross ( n, t );
    Basis m;
    m.r0[0] = t.x; m.r0[1] = n.x; m.r0[2] = b.x;
    m.r1[0] = t.y; m.r1[1] = n.y; m.r1[2] = b.y;
    m.r2[0] = t.z; m.r2[1] = n.z; m.r2[2] = b.z;
    return m;
}

struct Ray { V3 o, d, inv; };
TD Ray make_ray ( V3 o, V3 d ) { Ray r; r.o = o; r.d = d; r.inv = v3 ( 1.f / d.x, 1.f / d.y, 1.f / d.z ); return r; }

struct RayState { float shearx, sheary, scalez; int ix, iy, iz; };

// per-lane work counters (registers); flushed with one atomic per wave and counter.
// Not counted on the device because the host can derive them exactly: slab tests
// (= 2*nodes - tri_tests: every child of a popped node is either slab-tested or, if a
// leaf, triangle-tested), camera samples and pixels (tile geometry x spp).
#ifndef TERRA_PHASE_STATS          // lane-occupancy study builds (tools/phase_stats.py): per-phase wave iterations / active lanes
#define TERRA_PHASE_STATS 0
#endif
struct Counters {
    uint32_t rays, nodes, tri_tests, hits, rand_calls, attr_fetches;
    uint32_t tri_culled;     // leaves met whose triangle test was skipped (Tracer::cull); counted at COUNT level 2 only
#if TERRA_PHASE_STATS
    uint32_t ps[16];
#endif
};
TD Counters counters_zero() {
    Counters c; c.rays = c.nodes = c.tri_tests = c.hits = c.rand_calls = c.attr_fetches = c.tri_culled = 0;
#if TERRA_PHASE_STATS
    for ( int i = 0; i < 16; ++i ) c.ps[i] = 0;
#endif
    return c;
}
#if TERRA_PHASE_STATS
// PS_WAVE: +1 per wave (the first active lane counts); PS_LANE: +1 per active lane
#define PS_WAVE(c, k) do { if ( ( int ) ( threadIdx.x & 63 ) == __ffsll ( ( long long ) __ballot ( 1 ) ) - 1 ) ++( c ).ps[k]; } while ( 0 )
#define PS_LANE(c, k) do { ++( c ).ps[k]; } while ( 0 )
#else
#define PS_WAVE(c, k) do { } while ( 0 )
#define PS_LANE(c, k) do { } while ( 0 )
#endif
enum { kPsRayIter = 0, kPsNodeIter, kPsLeafIter, kPsShadeIter, kPsCamIter, kPsCamLanes, kPsRayLanes, kPsShadeLanes, kPsNodeLanes, kPsLeafLanes, kPsDrainIter,
       kPsTop64, kPsTop256, kPsTop1024, kPsTop4096 };      // node visits that fall into the first K nodes of the (breadth-first numbered) array: what an LDS-staged prefix would serve

// -----------------------------------------------------------------------------
// camera
// -----------------------------------------------------------------------------
TD V3 camera_sample ( const DevRenderParams& p, uint32_t px, uint32_t py, float r1, float r2 ) {
    float dx = -p.jitter + 2 * r1 * p.jitter;
    float dy = -p.jitter + 2 * r2 * p.jitter;
    float ndc_x = ( ( float ) px + 0.5f + dx ) / ( float ) p.fb_w;
    float ndc_y = ( ( float ) py + 0.5f + dy ) / ( float ) p.fb_h;
    float sx = 2 * ndc_x - 1;
    float sy = 1 - 2 * ndc_y;
    float fx = sx * p.aspect * p.tan_half_fov;
    float fy = sy * p.tan_half_fov;
    V3 d = normalize ( v3 ( fx, fy, 1.f ) );
    return v3 ( p.cam_rot[0] * d.x + p.cam_rot[1] * d.y + p.cam_rot[2] * d.z,
                p.cam_rot[3] * d.x + p.cam_rot[4] * d.y + p.cam_rot[5] * d.z,
                p.cam_rot[6] * d.x + p.cam_rot[7] * d.y + p.cam_rot[8] * d.z );
}

// -----------------------------------------------------------------------------
// slab test
// -----------------------------------------------------------------------------
TD bool ray_aabb ( const Ray& r, V3 bmin, V3 bmax, float* tmin_out, float* tmax_out ) {
    float t1 = ( bmin.x - r.o.x ) * r.inv.x;
    float t2 = ( bmax.x - r.o.x ) * r.inv.x;
    float tmin = sel_min ( t1, t2 ), tmax = sel_max ( t1, t2 );
    t1 = ( bmin.y - r.o.y ) * r.inv.y;
    t2 = ( bmax.y - r.o.y ) * r.inv.y;
    tmin = sel_max ( tmin, sel_min ( t1, t2 ) ); tmax = sel_min ( tmax, sel_max ( t1, t2 ) );
    t1 = ( bmin.z - r.o.z ) * r.inv.z;
    t2 = ( bmax.z - r.o.z ) * r.inv.z;
    tmin = sel_max ( tmin, sel_min ( t1, t2 ) ); tmax = sel_min ( tmax, sel_max ( t1, t2 ) );
    bool hit = tmax > sel_max ( tmin, 0.f );
    if ( tmin_out ) *tmin_out = tmin;
    if ( tmax_out ) *tmax_out = tmax;
    return hit;
}

// -----------------------------------------------------------------------------
// watertight ray/triangle
// -----------------------------------------------------------------------------
TD RayState ray_state_init ( const Ray& r ) {
    float ax = fabsf ( r.d.x ), ay = fabsf ( r.d.y ), az = fabsf ( r.d.z );
    int iz = ax > ay ? ( ax > az ? 0 : 2 ) : ( ay > az ? 1 : 2 );   // ties -> later axis
    int ix = iz + 1 == 3 ? 0 : iz + 1;
    int iy = ix + 1 == 3 ? 0 : ix + 1;
    if ( pick ( r.d, iz ) < 0.f ) { int t = ix; ix = iy; iy = t; }
    RayState s;
    s.scalez = pick ( r.inv, iz );          // 1.f / d[iz] (src/TerraGeometry.c:124): the quotient make_ray already holds, same operands, same rounding
    s.shearx = pick ( r.d, ix ) * s.scalez;
    s.sheary = pick ( r.d, iy ) * s.scalez;
    s.ix = ix; s.iy = iy; s.iz = iz;
    return s;
}

struct TriHit { float u, v, w, depth; V3 point; };

TD bool watertight ( const Ray& r, const RayState& s, V3 ta, V3 tb, V3 tc, TriHit& h ) {
    V3 A = ta - r.o, B = tb - r.o, C = tc - r.o;
    float Aiz = pick ( A, s.iz ), Biz = pick ( B, s.iz ), Ciz = pick ( C, s.iz );
    float Ax = pick ( A, s.ix ) - s.shearx * Aiz, Ay = pick ( A, s.iy ) - s.sheary * Aiz;
    float Bx = pick ( B, s.ix ) - s.shearx * Biz, By = pick ( B, s.iy ) - s.sheary * Biz;
    float Cx = pick ( C, s.ix ) - s.shearx * Ciz, Cy = pick ( C, s.iy ) - s.sheary * Ciz;
    float U = Cx * By - Cy * Bx;
    float V = Ax * Cy - Ay * Cx;
    float W = Bx * Ay - By * Ax;
    if ( U == 0.f || V == 0.f || W == 0.f ) {
        U = ( float ) ( ( double ) Cx * ( double ) By - ( double ) Cy * ( double ) Bx );
        V = ( float ) ( ( double ) Ax * ( double ) Cy - ( double ) Ay * ( double ) Cx );
        W = ( float ) ( ( double ) Bx * ( double ) Ay - ( double ) By * ( double ) Ax );
    }
    uint32_t sign = tdm_bits ( U ) & 0x80000000u;
    if ( ( ( tdm_bits ( V ) ^ tdm_bits ( U ) ) | ( tdm_bits ( W ) ^ tdm_bits ( U ) ) ) & 0x80000000u ) return false;
    float det = U + V + W;
    if ( det == 0.f ) return false;
    float Az = s.scalez * Aiz, Bz = s.scalez * Biz, Cz = s.scalez * Ciz;
    float depth = U * Az + V * Bz + W * Cz;
    if ( tdm_float ( tdm_bits ( depth ) ^ sign ) < 0.f ) return false;
    float inv_det = 1.f / det;
    h.u = U * inv_det; h.v = V * inv_det; h.w = W * inv_det;
    h.depth = depth * inv_det;
    h.point = r.o + r.d * h.depth;
    return true;
}

TD bool moller_trumbore ( V3 o, V3 d, V3 ta, V3 tb, V3 tc, float& t_out, V3& p_out ) {
    V3 e1 = tb - ta, e2 = tc - ta;
    V3 h = cross ( d, e2 );
    float a = dot ( e1, h );
    if ( ( double ) a > -1e-4 && ( double ) a < 1e-4 ) return false;
    float f = 1 / a;
    V3 s = o - ta;
    float u = f * dot ( s, h );
    if ( u < 0.f || u > 1.f ) return false;
    V3 q = cross ( s, e1 );
    float v = f * dot ( d, q );
    if ( v < 0.f || u + v > 1.f ) return false;
    float t = f * dot ( e2, q );
    if ( t > 0.00001f ) { t_out = t; p_out = d * t + o; return true; }
    return false;
}

// -----------------------------------------------------------------------------
// Tracer: where a thread finds the scene and its traversal scratch.
//
// LDS layout of a block (DESIGN.md "LDS"): [staged nodes: lds_nodes x 112 B] [staged triangles: lds_tris x 48 B]
// [staged vertex properties: lds_tris x 64 B] [node stack: stack_depth x 256 ints] [leaf list: leaf_cap x 256 ints]
// [per-thread parked words]. Stack and leaf list are indexed [entry][thread] so the 64 lanes of a wave touch
// 64 consecutive words (conflict free); a lane walks its column with a pointer (one add per push / pop).
// Nodes and triangles are staged only when the whole scene fits.
//
// Staged node (MODE 1), 7 x 16 B, "axis major, both signs":
//     [x+] min0.x max0.x min1.x max1.x     [x-] max0.x min0.x max1.x min1.x
//     [y+] ...                             [y-] ...
//     [z+] ...                             [z-] ...
//     [children] child0 child1 - -         (an inner child = the BYTE OFFSET of its staged node, a leaf = DEV_CHILD_LEAF | triangle)
// A ray whose inverse direction is finite and non-zero on every axis reads, per axis, the copy that matches the sign of its
// direction (SlabSel): the four floats are then (near plane, far plane) of child 0 and of child 1, so the slab test needs no
// per-axis min/max at all -- v_min/v_max_f32 issue at 0.57 G/s per SIMD on gfx950 against 0.96 for v_sub/v_mul_f32
// (profiles/r02_measurements/valu_rates.log). Picking the plane by the sign is exactly min(t1, t2) / max(t1, t2): for
// bmin <= bmax, (b - o) * inv is monotone in b (both roundings are), increasing for inv > 0 and decreasing for inv < 0.
// -----------------------------------------------------------------------------
#define TERRA_LEAF_CAP_MAX 16
#define TERRA_COL 256              // stride of a stack / leaf-list column: the block's thread count
#define TERRA_LDS_NODE_BYTES 112   // staged node (see above)

struct Tracer {
    DevScene      sc;
    const float4* l_nodes;     // LDS copies (valid for index < lds_nodes / lds_tris)
    const float*  l_tris;
    const float4* l_props;
    const DevMaterial* l_mats;  // materials, lights, per-triangle areas: the block's LDS copies in MODE 1, the arrays in HBM otherwise (make_tracer)
    const DevLight*    l_lights;
    const float*       l_area;
    uint32_t      lds_nodes, lds_tris;
    int*          stack;       // this thread's column
    int*          leaves;
    int           leaf_cap;    // entries in the leaf list (>= 2)
    int           stack_cap;   // entries in the stack column (TERRA_CHECK_BOUNDS builds verify every push against it)
    // fast-tree launches: entries beyond the LDS column live in HBM (DevRenderParams::stack_spill): spill = this lane's spill_cap words, nullptr when the column holds the whole stack
    uint32_t      stack_lim;   // 32-bit LDS address of the block's stack words + stack entries * 1024: wave-uniform (fast_push / fast_pop)
    uint32_t*     spill;
    uint32_t      spill_cap;
    unsigned long long* faults;
    // leaf-box cull (DESIGN.md "Leaf-box cull"): a leaf child's triangle is tested only if the ray passes the slab test of
    // that child's box -- the box the node already carries and the node step already tests. The reference tests the triangle
    // unconditionally (src/TerraBVH.c:284-300); the closest hit is the same whenever a triangle the ray hits lies inside its
    // own +-1e-4 box as the slab test sees it, which the host verifies numerically at commit (terra_cull_margin_ok).
    bool cull;
    // cull launches INSIDE the coordinate range may also decide the inner boxes with the fused slab arithmetic (slab_near_far_fused): the containment proof covers
    // every box there. Outside it (Scene::reach_cull) only the rebuilt leaf boxes carry a margin; the inner boxes must be tested exactly as the reference tests them.
    bool fused;
};

// -----------------------------------------------------------------------------
// slab test of one child box. FAST is legal when every component of the ray's
// inverse direction is finite and non-zero: then no NaN can appear (boxes and
// origins are finite) and "a<b?a:b" differs from v_min_f32 only in the sign of a
// zero, which the final comparison cannot see. Otherwise the compare-select form of
// the reference runs (NaN order matters there).
// -----------------------------------------------------------------------------
template <bool FAST>
TD bool slab ( V3 bmin, V3 bmax, const Ray& r ) {
    float t1x = ( bmin.x - r.o.x ) * r.inv.x, t2x = ( bmax.x - r.o.x ) * r.inv.x;
    float t1y = ( bmin.y - r.o.y ) * r.inv.y, t2y = ( bmax.y - r.o.y ) * r.inv.y;
    float t1z = ( bmin.z - r.o.z ) * r.inv.z, t2z = ( bmax.z - r.o.z ) * r.inv.z;
    if ( FAST ) {
        float tmin = __builtin_fmaxf ( __builtin_fmaxf ( __builtin_fminf ( t1x, t2x ), __builtin_fminf ( t1y, t2y ) ), __builtin_fminf ( t1z, t2z ) );
        float tmax = __builtin_fminf ( __builtin_fminf ( __builtin_fmaxf ( t1x, t2x ), __builtin_fmaxf ( t1y, t2y ) ), __builtin_fmaxf ( t1z, t2z ) );
        return tmax > __builtin_fmaxf ( tmin, 0.f );
    }
    float tmin = sel_min ( t1x, t2x ), tmax = sel_max ( t1x, t2x );
    tmin = sel_max ( tmin, sel_min ( t1y, t2y ) ); tmax = sel_min ( tmax, sel_max ( t1y, t2y ) );
    tmin = sel_max ( tmin, sel_min ( t1z, t2z ) ); tmax = sel_min ( tmax, sel_max ( t1z, t2z ) );
    return tmax > sel_max ( tmin, 0.f );
}

TD bool ray_is_regular ( const Ray& r ) {
    // finite and non-zero inverse direction components
    uint32_t ax = tdm_bits ( r.inv.x ) & 0x7fffffffu, ay = tdm_bits ( r.inv.y ) & 0x7fffffffu, az = tdm_bits ( r.inv.z ) & 0x7fffffffu;
    return ax - 1u < 0x7f7fffffu && ay - 1u < 0x7f7fffffu && az - 1u < 0x7f7fffffu;
}

// -----------------------------------------------------------------------------
// watertight test on components already gathered in the ray's permuted axes:
// p?[0..2] = vertex[ix], vertex[iy], vertex[iz]; o = origin permuted the same way.
// Same operations, in the same order, as watertight() above.
// -----------------------------------------------------------------------------
TD bool watertight_permuted ( const float pa[3], const float pb[3], const float pc[3], V3 o, const RayState& s, float& depth_out ) {
    float Aix = pa[0] - o.x, Aiy = pa[1] - o.y, Aiz = pa[2] - o.z;
    float Bix = pb[0] - o.x, Biy = pb[1] - o.y, Biz = pb[2] - o.z;
    float Cix = pc[0] - o.x, Ciy = pc[1] - o.y, Ciz = pc[2] - o.z;
    float Ax = Aix - s.shearx * Aiz, Ay = Aiy - s.sheary * Aiz;
    float Bx = Bix - s.shearx * Biz, By = Biy - s.sheary * Biz;
    float Cx = Cix - s.shearx * Ciz, Cy = Ciy - s.sheary * Ciz;
    float U = Cx * By - Cy * Bx;
    float V = Ax * Cy - Ay * Cx;
    float W = Bx * Ay - By * Ax;
    if ( U == 0.f || V == 0.f || W == 0.f ) {
        U = ( float ) ( ( double ) Cx * ( double ) By - ( double ) Cy * ( double ) Bx );
        V = ( float ) ( ( double ) Ax * ( double ) Cy - ( double ) Ay * ( double ) Cx );
        W = ( float ) ( ( double ) Bx * ( double ) Ay - ( double ) By * ( double ) Ax );
    }
    uint32_t sign = tdm_bits ( U ) & 0x80000000u;
    if ( ( ( tdm_bits ( V ) ^ tdm_bits ( U ) ) | ( tdm_bits ( W ) ^ tdm_bits ( U ) ) ) & 0x80000000u ) return false;
    float det = U + V + W;
    if ( det == 0.f ) return false;
    float Az = s.scalez * Aiz, Bz = s.scalez * Biz, Cz = s.scalez * Ciz;
    float depth = U * Az + V * Bz + W * Cz;
    if ( tdm_float ( tdm_bits ( depth ) ^ sign ) < 0.f ) return false;
    float inv_det = 1.f / det;
    depth_out = depth * inv_det;
    return true;
}

// -----------------------------------------------------------------------------
// BVH traversal (reference src/TerraBVH.c:250-310), restructured without changing
// what is computed:
//   * the node loop only does slab tests and stack traffic; leaves met on the way are
//     appended to a per-lane list and tested afterwards in the order they were met.
//     The reference never lets a hit influence the traversal (no culling against the
//     closest hit), so testing the leaves later, in the same order, with the same
//     strict "<" on depth, selects the same triangle;
//   * when a lane's list is full the lists are drained and the node loop resumes;
//   * the hit point is formed once, from the winning depth (same expression).
// MODE 0: nodes/triangles from global memory; 1: everything staged in LDS.
// -----------------------------------------------------------------------------
struct Closest { float depth; uint32_t tri; };

// Stack / leaf-list writes. A TERRA_CHECK_BOUNDS build (python -m terra_amd.build --variant chk -DTERRA_CHECK_BOUNDS=1)
// refuses (drops the entry, so the column is never left) and counts any write beyond the sizes the host planned; the shipped build trusts the plan
// (max_stack is the exact worst case of the tree, computed at commit).
#ifndef TERRA_CHECK_BOUNDS
#define TERRA_CHECK_BOUNDS 0
#endif
#define TERRA_PUSH(T, sp, v) do { if ( TERRA_CHECK_BOUNDS && ( sp ) >= ( T ).stack + ( T ).stack_cap * TERRA_COL ) { if ( ( T ).faults ) atomicAdd ( ( T ).faults, 1ull ); } else { *( sp ) = ( int ) ( v ); ( sp ) += TERRA_COL; } } while ( 0 )
#define TERRA_LEAF(T, lp, v) do { if ( TERRA_CHECK_BOUNDS && ( lp ) >= ( T ).leaves + ( T ).leaf_cap * TERRA_COL ) { if ( ( T ).faults ) atomicAdd ( ( T ).faults, 1ull ); } else { *( lp ) = ( int ) ( v ); ( lp ) += TERRA_COL; } } while ( 0 )

// which copy of each axis a lane reads from a staged node (byte offsets inside the node); regular rays only. oi = origin * inverse direction, for the
// fused form of the slab test (slab_near_far_fused)
struct SlabSel { uint32_t x, y, z; V3 oi; };
TD SlabSel slab_sel ( const Ray& r ) {
    SlabSel s;
    s.x = r.inv.x < 0.f ? 16u : 0u; s.y = r.inv.y < 0.f ? 48u : 32u; s.z = r.inv.z < 0.f ? 80u : 64u;
    s.oi = v3 ( r.o.x * r.inv.x, r.o.y * r.inv.y, r.o.z * r.inv.z );
    return s;
}
// regular AND every |inverse direction component| below 2^96: origin * inv cannot overflow for any origin the containment check admits
TD bool ray_is_tame ( const Ray& r ) {
    uint32_t ax = tdm_bits ( r.inv.x ) & 0x7fffffffu, ay = tdm_bits ( r.inv.y ) & 0x7fffffffu, az = tdm_bits ( r.inv.z ) & 0x7fffffffu;
    return ax - 1u < 0x6f7fffffu && ay - 1u < 0x6f7fffffu && az - 1u < 0x6f7fffffu;
}
// slab test from (near, far) planes per axis: what slab<true> computes, without the per-axis min / max
TD bool slab_near_far ( float nx, float fx, float ny, float fy, float nz, float fz, const Ray& r ) {
    float tnx = ( nx - r.o.x ) * r.inv.x, tfx = ( fx - r.o.x ) * r.inv.x;
    float tny = ( ny - r.o.y ) * r.inv.y, tfy = ( fy - r.o.y ) * r.inv.y;
    float tnz = ( nz - r.o.z ) * r.inv.z, tfz = ( fz - r.o.z ) * r.inv.z;
    float tmin = __builtin_fmaxf ( __builtin_fmaxf ( tnx, tny ), tnz );
    float tmax = __builtin_fminf ( __builtin_fminf ( tfx, tfy ), tfz );
    return tmax > __builtin_fmaxf ( tmin, 0.f );
}

// The same test with t = fma ( plane, inv, -(o * inv) ): one instruction per plane instead of two. NOT the reference's arithmetic -- (plane - o) * inv -- so only the
// launches that need not reproduce the reference's traversal decision by decision may use it: the leaf-box-cull launches (Tracer::cull), whose commit-time proof
// (scene_host.cpp "numeric containment check") only asks that every box test be CONSERVATIVE within the error budget: a triangle the ray hits must pass the test of
// every box built around it. Here t carries two roundings -- of o * inv and of the fma -- worth u |o| + u |plane - o| in position, less than the three roundings of
// the reference form the budget was drawn up for. Which nodes are visited beyond that may differ from the replica's by a few per billion (never the image).
TD bool slab_near_far_fused ( float nx, float fx, float ny, float fy, float nz, float fz, const Ray& r, V3 oi, float& t_enter ) {
    float tnx = __builtin_fmaf ( nx, r.inv.x, -oi.x ), tfx = __builtin_fmaf ( fx, r.inv.x, -oi.x );
    float tny = __builtin_fmaf ( ny, r.inv.y, -oi.y ), tfy = __builtin_fmaf ( fy, r.inv.y, -oi.y );
    float tnz = __builtin_fmaf ( nz, r.inv.z, -oi.z ), tfz = __builtin_fmaf ( fz, r.inv.z, -oi.z );
    float tmin = __builtin_fmaxf ( __builtin_fmaxf ( tnx, tny ), tnz );
    float tmax = __builtin_fminf ( __builtin_fminf ( tfx, tfy ), tfz );
    t_enter = __builtin_fmaxf ( tmin, 0.f );
    return tmax > t_enter;
}

// one node of the reference traversal (src/TerraBVH.c:262-303): pop, slab-test both child boxes, push the inner children
// that are hit, append the leaf children to the lane's list (all of them; with Tracer::cull only those whose box is hit).
// An empty child slot (scenes with < 2 triangles) travels as a leaf and is dropped by leaf_step.
template <int COUNT, int MODE, bool FAST, bool FUSED = false>
TD void node_step ( const Tracer& T, const Ray& r, const SlabSel& sel, int*& sp, int*& lp, Counters& c ) {
    PS_WAVE ( c, kPsNodeIter ); PS_LANE ( c, kPsNodeLanes );
    sp -= TERRA_COL;
    const uint32_t w = ( uint32_t ) * sp;
    uint32_t child0, child1; bool hit0, hit1;
    float te0 = 0.f, te1 = 0.f;          // (FUSED) entry distance of each child box (unused: kept out of registers by the optimiser)
    if ( MODE == 1 ) {
        const char* node = reinterpret_cast<const char*> ( T.l_nodes ) + w;          // w = byte offset of the staged node
        const uint2 cw = *reinterpret_cast<const uint2*> ( node + 96 );
        child0 = cw.x; child1 = cw.y;
        if ( FAST ) {
            const float4 ax = *reinterpret_cast<const float4*> ( node + sel.x ), ay = *reinterpret_cast<const float4*> ( node + sel.y ), az = *reinterpret_cast<const float4*> ( node + sel.z );
            if ( FUSED ) {
                hit0 = slab_near_far_fused ( ax.x, ax.y, ay.x, ay.y, az.x, az.y, r, sel.oi, te0 );
                hit1 = slab_near_far_fused ( ax.z, ax.w, ay.z, ay.w, az.z, az.w, r, sel.oi, te1 );
            } else {
                hit0 = slab_near_far ( ax.x, ax.y, ay.x, ay.y, az.x, az.y, r );
                hit1 = slab_near_far ( ax.z, ax.w, ay.z, ay.w, az.z, az.w, r );
            }
        } else {
            const float4 ax = *reinterpret_cast<const float4*> ( node ), ay = *reinterpret_cast<const float4*> ( node + 32 ), az = *reinterpret_cast<const float4*> ( node + 64 );
            hit0 = slab<false> ( v3 ( ax.x, ay.x, az.x ), v3 ( ax.y, ay.y, az.y ), r );
            hit1 = slab<false> ( v3 ( ax.z, ay.z, az.z ), v3 ( ax.w, ay.w, az.w ), r );
        }
    } else {
        const float4* g_nodes = reinterpret_cast<const float4*> ( T.sc.nodes );
#if TERRA_PHASE_STATS
        c.ps[kPsTop64] += w < 64u; c.ps[kPsTop256] += w < 256u; c.ps[kPsTop1024] += w < 1024u; c.ps[kPsTop4096] += w < 4096u;
#endif
        const float4 q0 = g_nodes[4 * w], q1 = g_nodes[4 * w + 1], q2 = g_nodes[4 * w + 2], q3 = g_nodes[4 * w + 3];
        child0 = __float_as_uint ( q3.x ); child1 = __float_as_uint ( q3.y );
        hit0 = slab<FAST> ( v3 ( q0.x, q0.y, q0.z ), v3 ( q0.w, q1.x, q1.y ), r );
        hit1 = slab<FAST> ( v3 ( q1.z, q1.w, q2.x ), v3 ( q2.y, q2.z, q2.w ), r );
    }
    if ( COUNT ) ++c.nodes;
    const bool leaf0 = ( child0 & DEV_CHILD_LEAF ) != 0, leaf1 = ( child1 & DEV_CHILD_LEAF ) != 0;
    if ( !leaf0 && hit0 ) { TERRA_PUSH ( T, sp, child0 ); }
    if ( !leaf1 && hit1 ) { TERRA_PUSH ( T, sp, child1 ); }
    if ( leaf0 && ( hit0 || !T.cull ) ) { TERRA_LEAF ( T, lp, ( child0 & 0x7fffffffu ) ); }
    if ( leaf1 && ( hit1 || !T.cull ) ) { TERRA_LEAF ( T, lp, ( child1 & 0x7fffffffu ) ); }
    if ( COUNT == 2 && T.cull ) c.tri_culled += ( uint32_t ) ( leaf0 && !hit0 ) + ( uint32_t ) ( leaf1 && !hit1 );
}

// triangle test of one entry of the lane's leaf list, in the order the leaves were met: strict "<" keeps the first of equal depths
// (expected != none: the ray only asks whether its closest hit is triangle `expected` -- scene_raycast_triangle -- and `stop` is set by the first other triangle that comes first)
template <int COUNT, int MODE, bool ANYHIT = false>
TD void leaf_step ( const Tracer& T, const int* entry, const RayState& st, V3 o_perm, Closest& best, Counters& c, uint32_t expected = 0xffffffffu, bool* stop = nullptr ) {
    const float4* g_tris = reinterpret_cast<const float4*> ( T.sc.tris );
    const int kx = st.ix, ky = st.iy, kz = st.iz;
    PS_WAVE ( c, kPsLeafIter ); PS_LANE ( c, kPsLeafLanes );
    const uint32_t ti = ( uint32_t ) * entry;
    if ( ti == ( DEV_CHILD_EMPTY & 0x7fffffffu ) ) return;           // the empty slot of a degenerate tree
    float pa[3], pb[3], pc[3];
    if ( MODE == 1 ) {
        const float* t = T.l_tris + 12 * ti;
        pa[0] = t[kx]; pa[1] = t[ky]; pa[2] = t[kz];
        pb[0] = t[4 + kx]; pb[1] = t[4 + ky]; pb[2] = t[4 + kz];
        pc[0] = t[8 + kx]; pc[1] = t[8 + ky]; pc[2] = t[8 + kz];
    } else {
        float4 a = g_tris[3 * ti], b = g_tris[3 * ti + 1], cc = g_tris[3 * ti + 2];
        V3 va = v3 ( a.x, a.y, a.z ), vb = v3 ( b.x, b.y, b.z ), vc = v3 ( cc.x, cc.y, cc.z );
        pa[0] = pick ( va, kx ); pa[1] = pick ( va, ky ); pa[2] = pick ( va, kz );
        pb[0] = pick ( vb, kx ); pb[1] = pick ( vb, ky ); pb[2] = pick ( vb, kz );
        pc[0] = pick ( vc, kx ); pc[1] = pick ( vc, ky ); pc[2] = pick ( vc, kz );
    }
    if ( COUNT ) ++c.tri_tests;
    float depth;
    if ( watertight_permuted ( pa, pb, pc, o_perm, st, depth ) && depth < best.depth ) { best.depth = depth; best.tri = ti; if ( ANYHIT && ti != expected ) *stop = true; }
}

template <int COUNT, int MODE, bool FAST, bool FUSED = false, bool ANYHIT = false>
TD void traverse_loops ( const Tracer& T, const Ray& r, const RayState& st, V3 o_perm, Closest& best, Counters& c, uint32_t expected = 0xffffffffu ) {
    const SlabSel sel = slab_sel ( r );
    int* sp = T.stack; int* lp = T.leaves;
    int* const lp_full = T.leaves + ( T.leaf_cap - 2 ) * TERRA_COL;       // a node adds at most two leaves
    *sp = 0; sp += TERRA_COL;                                          // the root: node 0 = byte offset 0
    for ( ;; ) {
        PS_WAVE ( c, kPsDrainIter );
        while ( sp != T.stack && lp <= lp_full ) node_step<COUNT, MODE, FAST, FUSED> ( T, r, sel, sp, lp, c );
        if constexpr ( ANYHIT ) {
            bool stop = false;
            for ( const int* e = T.leaves; e != lp && !stop; e += TERRA_COL ) leaf_step<COUNT, MODE, true> ( T, e, st, o_perm, best, c, expected, &stop );
            if ( stop ) sp = T.stack;                                  // another triangle comes first: nothing further can change the answer
        } else
        for ( const int* e = T.leaves; e != lp; e += TERRA_COL ) leaf_step<COUNT, MODE> ( T, e, st, o_perm, best, c );
        lp = T.leaves;
        if ( sp == T.stack ) break;
    }
}

#ifndef TERRA_FUSED_SLAB
#define TERRA_FUSED_SLAB 1
#endif
template <int COUNT, int MODE>
TD Closest bvh_traverse ( const Tracer& T, const Ray& r, const RayState& st, Counters& c ) {
    Closest best; best.depth = FLT_MAX; best.tri = 0xffffffffu;
    V3 o_perm = v3 ( pick ( r.o, st.ix ), pick ( r.o, st.iy ), pick ( r.o, st.iz ) );
    // the slab variant is chosen per WAVE: one irregular ray sends its whole wave down the exact path
    if ( TERRA_FUSED_SLAB && MODE == 1 && T.fused && __all ( ray_is_tame ( r ) ) ) traverse_loops<COUNT, MODE, true, true> ( T, r, st, o_perm, best, c );
    else if ( __all ( ray_is_regular ( r ) ) ) traverse_loops<COUNT, MODE, true> ( T, r, st, o_perm, best, c );
    else traverse_loops<COUNT, MODE, false> ( T, r, st, o_perm, best, c );
    return best;
}

// -----------------------------------------------------------------------------
// MODE 2: traversal of the optional fast tree (DESIGN.md "Fast tree"). Not the reference's
// traversal: near child first, subtrees whose entry distance exceeds the closest hit are
// culled, leaves hold up to 4 triangles. It returns the reference's answer because the
// reference's closest hit is order independent once ties are resolved the way its fixed
// leaf visit order resolves them: smallest depth, then smallest reference visit rank
// (DevTri::pad of the fast soup). The triangle test itself is the same arithmetic.
// Child word of a fast node: bit 31 leaf; leaf = (count-1) << 27 | first triangle.
//
// Node format in HBM (DevFastNode, 128 B = one cache line, FOUR children; made on the host by tree_build.cpp fastbvh::widen): the planes of the child boxes as binary16,
// rounded OUTWARD, one 32-bit word per child and axis, the four children's words of one axis side by side -- and each such 16-byte group TWICE: as (min | max << 16)
// for rays that travel in the axis' positive direction and as (max | min << 16) for the others --, then the four child words. A ray loads, per axis, the group whose
// low half is ITS near plane (three offsets it computes once), plus the child words: four 16-byte loads fetch what it needs of a node, 64 of the 128 bytes.
// What binds the render kernels of scenes read from global memory (profiles/r04_measurements/ab_fast_node_formats.log): (1) the texture addresser -- ~21-27 of its
// cycles per wave-level load instruction whatever the instruction's width, 99 % busy when a node step issues 4 loads for 2 boxes -- so what counts is the NUMBER of
// load instructions per ray, not their bytes; (2) the dependent chain: a ray's node fetches follow one another, and the waves spend 54 % of their cycles waiting;
// (3) VALU issue. A 4-wide node of binary16 planes answers (1) and (2): 4 loads bring 4 boxes (the 64-byte (min, max) node of rounds 2-3 brought 2), and a ray needs
// half as many round trips. Each plane goes straight into t = fma ( plane, inv, -(o * inv) ) as the binary16 operand of v_fma_mix_f32 (op_sel picks the half): a box
// costs six fused multiply-adds, a v_max3, a v_min3 and the comparisons. (Until round 4's third session a node was 64 bytes with one (min | max << 16) word per child
// and axis, and the ray swapped the halves with a v_perm_b32 per box and axis: twelve more VALU instructions per node step on a kernel bound by VALU issue; choosing
// the group by ADDRESS costs three adds. Hall 198.9 -> 193.3 ms, sphere scene 302.2 -> 293.8: profiles/r04_measurements/ab_fast_tree_builder.log.) Unlike the
// reference tree's test this one only has to be CONSERVATIVE (never
// reject a box that holds a triangle the ray hits; DESIGN.md "Traversal policy"): rounding the planes outward only widens the box, and t carries two roundings
// (o * inv, the fma) where the commit-time error budget (scene_host.cpp "numeric containment check") allows four. Planes are stored times DevScene::fast_scale (a
// power of two: exact) so that every scene fits binary16's range; the ray's inverse direction is divided by it (exact too).
// An empty child slot is an inverted box (min = +max_half, max = -max_half): never entered, so no lane ever holds it.
// -----------------------------------------------------------------------------
typedef _Float16 terra_half2 __attribute__ (( ext_vector_type ( 2 ) ));
TD terra_half2 as_half2 ( uint32_t u ) { return __builtin_bit_cast ( terra_half2, u ); }
// what a ray needs of itself for the fast tree's box test: the inverse direction -- clamped (a ray parallel to an axis has an infinite inverse direction there, and inf - inf
// would drop that axis from the test: correct but ruinous, such a ray then visits every box along its line; clamped to +-2^100 the axis keeps its meaning: (plane - o) * 2^100
// has the sign of "outside the slab") and divided by the planes' scale --, origin x clamped inverse direction, and per axis the byte offset (inside a node) of the plane group that has the ray's near plane in the low half
struct FastRay { V3 inv, oi; uint32_t px, py, pz; };
TD FastRay fast_ray ( const Ray& ray, float inv_scale ) {
    FastRay f;
    const float cx = __builtin_fminf ( __builtin_fmaxf ( ray.inv.x, -0x1p100f ), 0x1p100f ), cy = __builtin_fminf ( __builtin_fmaxf ( ray.inv.y, -0x1p100f ), 0x1p100f ), cz = __builtin_fminf ( __builtin_fmaxf ( ray.inv.z, -0x1p100f ), 0x1p100f );
    f.oi = v3 ( ray.o.x * cx, ray.o.y * cy, ray.o.z * cz );
    f.inv = v3 ( cx * inv_scale, cy * inv_scale, cz * inv_scale );
    f.px = cx < 0.f ? 16u : 0u; f.py = cy < 0.f ? 48u : 32u; f.pz = cz < 0.f ? 80u : 64u;      // byte offsets of the ray's plane groups inside a node
    return f;
}
// entry distance of one child box from its three plane words (near | far << 16 per axis: the ray loaded the group that has them this way round); hit = the ray's interval inside the box is not empty and starts no later
// than the closest hit so far. `limit_up` = that hit's depth plus one ulp: t_enter <= depth is t_enter < limit_up, which folds into the min3 of the far planes.
TD bool slab_half ( uint32_t wx, uint32_t wy, uint32_t wz, const FastRay& f, float limit_up, float& t_enter ) {
    const terra_half2 x = as_half2 ( wx ), y = as_half2 ( wy ), z = as_half2 ( wz );      // (near, far): the ray loaded the group that has them this way round
    const float tnx = __builtin_fmaf ( ( float ) x.x, f.inv.x, -f.oi.x ), tfx = __builtin_fmaf ( ( float ) x.y, f.inv.x, -f.oi.x );
    const float tny = __builtin_fmaf ( ( float ) y.x, f.inv.y, -f.oi.y ), tfy = __builtin_fmaf ( ( float ) y.y, f.inv.y, -f.oi.y );
    const float tnz = __builtin_fmaf ( ( float ) z.x, f.inv.z, -f.oi.z ), tfz = __builtin_fmaf ( ( float ) z.y, f.inv.z, -f.oi.z );
    t_enter = __builtin_fmaxf ( __builtin_fmaxf ( __builtin_fmaxf ( tnx, tny ), tnz ), 0.f );
    return __builtin_fminf ( __builtin_fminf ( __builtin_fminf ( tfx, tfy ), tfz ), limit_up ) > t_enter;
}

struct ClosestRanked { float depth; uint32_t rank; uint32_t tri; };

// Would the REFERENCE traversal (src/TerraBVH.c:250-310) have tested fast triangle `ti` for this ray? It tests a leaf child whenever it visits the
// leaf's node, and it visits a node when the slab test of that node's box -- stored in its parent -- passed, for every inner node on the way down from
// the root. So: replay that slab test (the reference's compare-select form, unclamped inverse direction) up the parent links. Only scenes outside the
// coordinate range of the containment proof need this (DevScene::reach): inside it, a triangle the ray hits implies all of these tests pass.
#ifndef TERRA_REACH_SELFCHECK      // check builds: replay every level and count (terra_amd_debug_faults) the ones the mask had cleared that fail -- must stay 0
#define TERRA_REACH_SELFCHECK 0
#endif
TD bool reference_reaches ( const Tracer& T, uint32_t ti, const Ray& ray ) {
    const float4* tab = reinterpret_cast<const float4*> ( T.sc.ref_replay );
    // levels whose test can fail at all (DevScene::fast_leaf_mask): the walk ends above the highest of them. The mask's "contains the level below" shortcut
    // needs a regular ray (monotone slab arithmetic, no NaN); any other ray replays every level
    uint32_t mask = ray_is_regular ( ray ) ? T.sc.fast_leaf_mask[ti] : 0xffffffffu;
    uint32_t q = T.sc.fast_leaf_parent[ti];
    bool ok = true;
    while ( q != 0u && ( TERRA_REACH_SELFCHECK || mask != 0u ) ) {
        const float4 a = tab[2 * q], b = tab[2 * q + 1];          // {min, parent} {max, -}
        if ( TERRA_REACH_SELFCHECK || ( mask & 1u ) ) {
            if ( !slab<false> ( v3 ( a.x, a.y, a.z ), v3 ( b.x, b.y, b.z ), ray ) ) {
                if ( !TERRA_REACH_SELFCHECK ) return false;
                // a cleared level that fails while every tested level below it passed: the mask is wrong. (After a tested level has failed, the levels above may
                // fail too -- "contains the level below" only promises that a pass propagates upwards -- and mean nothing.)
                if ( mask & 1u ) ok = false; else if ( ok && T.faults ) atomicAdd ( T.faults, 1ull );
            }
        }
        q = __float_as_uint ( a.w );
        mask = ( mask & 0x80000000u ) | ( mask >> 1 );      // next level (bit 31 stands for every level from 31 up)
    }
    return ok;
}

// A lane's traversal state is its stack and TWO registers. `held`: the leaf whose triangles it is testing (0 = none). `hand`: DEV_CHILD_EMPTY = nothing; a node
// index = the node it descends into next; a leaf word = the next leaf, waiting for `held` to be free. Of the children of a node whose boxes the ray enters, the
// nearest goes into `hand` (so a descent step does not wait for an LDS write + read of its own) -- or straight into `held` when it is a leaf and `held` is free --
// and the others wait on the lane's stack, farthest first. The stack's first entries are an LDS column, the rest -- which a ray almost never reaches: the column
// covers the depths rays actually see, the bound is the tree's worst case -- a few words of HBM per lane (fast_push / fast_pop). A ray starts with the root
// (node 0) in hand and an empty stack.
// Each iteration the wave votes: while fewer than TERRA_FAST_LEAF_16THS / 16 of its busy lanes hold a leaf (and some lane can still descend) the lanes that can
// descend take a node step, otherwise the holders test one triangle each. A lane that holds a leaf does NOT wait for the triangle step: it goes on descending
// towards its next leaf (speculatively: had its held leaf been tested first, the closer hit might have culled some of those nodes) and only stops when that one is
// in hand too. Without this, node steps ran 61 % full and triangle steps 40 % (hall); the few extra node visits cost less than the fuller steps save
// (profiles/r04_measurements/ab_fast_tree_knobs.log). (16/16 would be the classic "while-while" loop: descend until every lane holds a leaf.)
// The traversal is resumable (stack in LDS / HBM; top, hand, held, closest hit in registers): it returns as soon as the number of busy lanes has dropped to
// `exit_active`, so the render loop can shade the finished lanes and hand them their next ray (exit_active = 0: run every lane's ray to the end). A lane is done
// when it holds nothing and its stack is empty (fast_traversing). WHICH nodes a lane visits depends on the votes (on when its held leaf is tested), the closest hit
// it returns does not: that is the minimum over (depth, reference visit rank) of the triangles the ray hits, and no box that holds it is ever culled.
#ifndef TERRA_FAST_LEAF_16THS
#define TERRA_FAST_LEAF_16THS 12
#endif
#ifndef TERRA_FAST_SORT            // 1: the entered children of a node are visited nearest first (sorting network); 0: nearest first, the rest in slot order (A/B)
#define TERRA_FAST_SORT 1
#endif
// (the test "is this entry in the LDS column" compares the entry's 32-bit LDS address with ONE wave-uniform limit: entry e of thread t sits at column base + e * 1024 + t * 4,
//  and t * 4 < 1024, so address < base + cap * 1024 exactly when e < cap; the HBM index is computed on the cold side only)
// (Entries that carry their box's entry distance, so that one the closest hit has overtaken is dropped when it comes off the stack, were measured: 8 % fewer node
//  steps on the hall, 5 % on the sphere scene, and no time gained -- the second word and the pop loop cost what they save. profiles/r04_measurements/ab_fast_tree_knobs.log)
TD void fast_push ( const Tracer& T, int*& top, uint32_t v ) {
    const uint32_t a = ( uint32_t ) ( uintptr_t ) top;
    // (bounds-checking builds: an entry beyond what the host planned -- LDS column + HBM part, the positive control's shrink taken off the column -- is refused and counted)
    if ( TERRA_CHECK_BOUNDS && ( int ) ( top - T.stack ) / TERRA_COL >= T.stack_cap + ( int ) T.spill_cap ) { if ( T.faults ) atomicAdd ( T.faults, 1ull ); return; }
    if ( __builtin_expect ( a < T.stack_lim, 1 ) ) *top = ( int ) v;
    else {
        const uint32_t k = ( a - T.stack_lim ) >> 10;
        if ( TERRA_CHECK_BOUNDS && ( !T.spill || k >= T.spill_cap ) ) { if ( T.faults ) atomicAdd ( T.faults, 1ull ); return; }
        T.spill[k] = v;
    }
    top += TERRA_COL;
}
TD uint32_t fast_pop ( const Tracer& T, int*& top ) {
    top -= TERRA_COL;
    const uint32_t a = ( uint32_t ) ( uintptr_t ) top;
    // (the LDS side is read through an LDS-typed pointer: left as two loads of generic pointers, the compiler merges them into ONE flat load of a selected address -- and a
    //  flat load goes through the texture addresser, the unit these kernels are short of, instead of the LDS pipeline)
    typedef const __attribute__ (( address_space ( 3 ) )) uint32_t* LdsPtr;
    if ( __builtin_expect ( a < T.stack_lim, 1 ) ) return * ( LdsPtr ) ( uintptr_t ) a;
    return T.spill[ ( a - T.stack_lim ) >> 10];
}
// compare-exchange of two (key, child word) pairs: afterwards a holds the smaller key
TD void order_pair ( uint32_t& ka, uint32_t& ca, uint32_t& kb, uint32_t& cb ) {
    const bool swap = kb < ka;
    const uint32_t k0 = swap ? kb : ka, k1 = swap ? ka : kb, c0 = swap ? cb : ca, c1 = swap ? ca : cb;
    ka = k0; kb = k1; ca = c0; cb = c1;
}
#define TERRA_FAST_ROOT_IN_HAND 0u
// (a leaf word in `hand` is recognised by ( int ) hand < -1: DEV_CHILD_EMPTY is -1 and no leaf word is -- a leaf has at most 4 triangles, so bits 29-30 of its count field are clear)
TD bool fast_traversing ( const Tracer& T, uint32_t hand, uint32_t held, const int* top ) { return ( hand != DEV_CHILD_EMPTY ) | ( top != T.stack ) | ( held != 0u ); }
template <int COUNT>
TD void traverse_fast_resume ( const Tracer& T, const Ray& ray, const RayState& st, V3 o_perm, ClosestRanked& best, int*& top, uint32_t& hand, uint32_t& held, int exit_active, Counters& c, bool checked = false, bool anyhit = false ) {
    const FastRay f = fast_ray ( ray, T.sc.fast_inv_scale );
    const char* nodes = reinterpret_cast<const char*> ( T.sc.fast_nodes_h );
    const float4* tris = reinterpret_cast<const float4*> ( T.sc.fast_tris );
    for ( ;; ) {
        // (the votes are taken on plain compares, whose results ARE wave masks; a vote on a combined bool costs a select + a compare to rebuild the mask)
        // a lane TESTS the leaf in `held` and may meanwhile descend on towards its next one (which then waits in hand): busy = can descend or holds
        const uint64_t m_hold = __builtin_amdgcn_ballot_w64 ( held != 0u ), m_can = __builtin_amdgcn_ballot_w64 ( ( int ) hand >= 0 ) | ( __builtin_amdgcn_ballot_w64 ( hand == DEV_CHILD_EMPTY ) & __builtin_amdgcn_ballot_w64 ( top != T.stack ) );
        const int n_can = __popcll ( m_can ), n_hold = __popcll ( m_hold ), n_busy = __popcll ( m_can | m_hold );
        if ( n_busy <= exit_active ) break;
        if ( n_can != 0 && n_hold * 16 < n_busy * TERRA_FAST_LEAF_16THS ) {
            if ( ( ( int ) hand >= 0 ) | ( ( hand == DEV_CHILD_EMPTY ) & ( top != T.stack ) ) ) {
                PS_WAVE ( c, kPsNodeIter ); PS_LANE ( c, kPsNodeLanes );
                uint32_t w = hand;
                if ( ( int ) w < 0 ) w = fast_pop ( T, top );
                uint32_t nw = w;                                     // (a leaf that waited on the stack stays in hand)
                if ( ( int ) w >= 0 ) {
                    const uint32_t off = w << 7;
                    const uint4 gx = *reinterpret_cast<const uint4*> ( nodes + ( off + f.px ) ), gy = *reinterpret_cast<const uint4*> ( nodes + ( off + f.py ) ),
                                gz = *reinterpret_cast<const uint4*> ( nodes + ( off + f.pz ) ), ch = *reinterpret_cast<const uint4*> ( nodes + ( off + 96u ) );      // {x0 x1 x2 x3} {y0 ..} {z0 ..} {children}
                    if ( COUNT ) ++c.nodes;
#if TERRA_PHASE_STATS
                    c.ps[kPsTop64] += w < 64u; c.ps[kPsTop256] += w < 256u; c.ps[kPsTop1024] += w < 1024u; c.ps[kPsTop4096] += w < 4096u;
#endif
                    const float limit_up = __uint_as_float ( __float_as_uint ( best.depth + 0.f ) + 1u );      // the next float up (FLT_MAX -> inf); + 0.f: a hit at depth -0 counts as +0
                    float te0, te1, te2, te3;
                    const bool hit0 = slab_half ( gx.x, gy.x, gz.x, f, limit_up, te0 );
                    const bool hit1 = slab_half ( gx.y, gy.y, gz.y, f, limit_up, te1 );
                    const bool hit2 = slab_half ( gx.z, gy.z, gz.z, f, limit_up, te2 );
                    const bool hit3 = slab_half ( gx.w, gy.w, gz.w, f, limit_up, te3 );
                    // nearest first: the entry distances (>= 0, so their bit patterns order like the floats) sorted with their child words; a box not entered sorts last
                    uint32_t k0 = hit0 ? __float_as_uint ( te0 ) : 0xffffffffu, k1 = hit1 ? __float_as_uint ( te1 ) : 0xffffffffu, k2 = hit2 ? __float_as_uint ( te2 ) : 0xffffffffu, k3 = hit3 ? __float_as_uint ( te3 ) : 0xffffffffu;
                    uint32_t c0 = ch.x, c1 = ch.y, c2 = ch.z, c3 = ch.w;
#if TERRA_FAST_SORT
                    order_pair ( k0, c0, k1, c1 ); order_pair ( k2, c2, k3, c3 ); order_pair ( k0, c0, k2, c2 ); order_pair ( k1, c1, k3, c3 ); order_pair ( k1, c1, k2, c2 );
#else               // (A/B) only the nearest is found; the others go on the stack in slot order
                    order_pair ( k0, c0, k1, c1 ); order_pair ( k0, c0, k2, c2 ); order_pair ( k0, c0, k3, c3 );
#endif
                    // the farthest goes in first, so the nearer ones come off first. (Branch-free pushes -- every child word stored at the top, the top moved only for the
                    // entered ones -- measured no faster on the hall and 2 % slower on the sphere scene: profiles/r04_measurements/ab_fast_tree_knobs.log)
                    if ( k3 != 0xffffffffu ) fast_push ( T, top, c3 );
                    if ( k2 != 0xffffffffu ) fast_push ( T, top, c2 );
                    if ( k1 != 0xffffffffu ) fast_push ( T, top, c1 );
#if TERRA_PHASE_STATS
                    if ( k1 != 0xffffffffu ) { const int dpt = ( int ) ( top - T.stack ) / TERRA_COL; ++c.ps[kPsCamLanes]; c.ps[kPsShadeIter] += dpt >= 4; c.ps[kPsRayLanes] += dpt >= 6; c.ps[kPsCamIter] += dpt >= 8; c.ps[kPsDrainIter] += dpt >= 12; c.ps[kPsShadeLanes] += dpt >= 16; }
#endif
                    nw = k0 != 0xffffffffu ? c0 : DEV_CHILD_EMPTY;
                }
                { const bool take = ( ( int ) nw < -1 ) & ( held == 0u ); held = take ? nw : held; nw = take ? DEV_CHILD_EMPTY : nw; }      // a leaf goes to the testing slot if that is free
                hand = nw;
            }
        } else if ( held != 0u ) {
            PS_WAVE ( c, kPsLeafIter ); PS_LANE ( c, kPsLeafLanes );
            const uint32_t ti = held & 0x07ffffffu;
            held = ( held & 0x78000000u ) ? held + 1u - 0x08000000u : 0u;        // next triangle of the leaf, one fewer to go
            if ( ( held == 0u ) & ( ( int ) hand < -1 ) ) { held = hand; hand = DEV_CHILD_EMPTY; }      // the leaf that waited in hand moves up
            const float4 a = tris[3 * ti], b = tris[3 * ti + 1], cc = tris[3 * ti + 2];          // three loads: every wave-level load instruction costs the texture addresser the same ~21 cycles
            const V3 va = v3 ( a.x, a.y, a.z ), vb = v3 ( b.x, b.y, b.z ), vc = v3 ( cc.x, cc.y, cc.z );
            const float pa[3] = { pick ( va, st.ix ), pick ( va, st.iy ), pick ( va, st.iz ) };
            const float pb[3] = { pick ( vb, st.ix ), pick ( vb, st.iy ), pick ( vb, st.iz ) };
            const float pc[3] = { pick ( vc, st.ix ), pick ( vc, st.iy ), pick ( vc, st.iz ) };
            const uint32_t rank = __float_as_uint ( cc.w );
            if ( COUNT ) ++c.tri_tests;
            float depth;
            if ( watertight_permuted ( pa, pb, pc, o_perm, st, depth ) ) {
                if ( depth < best.depth || ( depth == best.depth && rank < best.rank ) ) {
                    if ( !checked || reference_reaches ( T, ti, ray ) ) { best.depth = depth; best.rank = rank; best.tri = ti; }      // (checked: DevScene::reach, second pass)
                    // a shadow ray that knows the triangle it expects (fast_expect) only asks whether ANY triangle comes first: this one does, the lane is done
                    if ( anyhit ) { top = T.stack; hand = DEV_CHILD_EMPTY; held = 0u; }
                }
            }
        }
    }
}

// A ray of which only "is triangle E the closest hit" matters -- the light-sample ray of the Direct and MIS integrators (src/Terra.c:1349-1426: the sample counts when
// the ray's closest hit is the sampled light triangle) -- need not search for its closest hit. E is tested first, with the arithmetic the traversal would use on it; if
// the ray misses E the answer is no, whatever else it hits (false: the caller traces the ray the ordinary way, for the hit count). Otherwise the traversal starts from
// the closest hit (depth of E, rank of E): every box beyond E is culled from the first node on, and the first triangle that beats E -- nearer, or as near with a smaller
// reference visit rank: exactly the triangles the reference's traversal would have preferred -- ends it (traverse_fast_resume `anyhit`). best.tri stays
// TERRA_TRI_EXPECTED if none does. Scenes inside the coordinate range only (MODE 2: what the reference reaches needs no replay), kernels without work counters only
// (the attribute-fetch counter is defined by the CLOSEST hit's material). DevTri::pad of the soup holds the rank when the scene has a fast tree.
#define TERRA_TRI_EXPECTED 0xfffffffeu
#ifndef TERRA_SHADOW_ANYHIT
#define TERRA_SHADOW_ANYHIT 1
#endif
TD bool fast_expect ( const Tracer& T, const RayState& st, V3 o_perm, uint32_t expected_soup, ClosestRanked& best ) {
    const float4* tris = reinterpret_cast<const float4*> ( T.sc.tris );
    const float4 a = tris[3 * expected_soup], b = tris[3 * expected_soup + 1], cc = tris[3 * expected_soup + 2];
    const V3 va = v3 ( a.x, a.y, a.z ), vb = v3 ( b.x, b.y, b.z ), vc = v3 ( cc.x, cc.y, cc.z );
    const float pa[3] = { pick ( va, st.ix ), pick ( va, st.iy ), pick ( va, st.iz ) };
    const float pb[3] = { pick ( vb, st.ix ), pick ( vb, st.iy ), pick ( vb, st.iz ) };
    const float pc[3] = { pick ( vc, st.ix ), pick ( vc, st.iy ), pick ( vc, st.iz ) };
    float depth;
    if ( !watertight_permuted ( pa, pb, pc, o_perm, st, depth ) ) return false;
    best.depth = depth; best.rank = __float_as_uint ( cc.w ); best.tri = TERRA_TRI_EXPECTED;
    return true;
}

// REACH = false: the kernels launched for scenes inside the coordinate range (template MODE 2) carry none of the replay code; MODE 3 = the same loops with it
template <int COUNT, bool REACH = true>
TD ClosestRanked bvh_traverse_fast ( const Tracer& T, const Ray& r, const RayState& st, Counters& c ) {
    V3 o_perm = v3 ( pick ( r.o, st.ix ), pick ( r.o, st.iy ), pick ( r.o, st.iz ) );
    ClosestRanked best;
    // DevScene::reach: the closest of ALL hits is the answer if the reference would have reached it (then it is also the closest of the reachable ones);
    // only if not -- float rounding makes that very rare -- the ray is traced again with every candidate checked
    for ( int pass = 0; pass < 2; ++pass ) {
        best.depth = FLT_MAX; best.rank = 0xffffffffu; best.tri = 0xffffffffu;
        int* top = T.stack;
        uint32_t hand = TERRA_FAST_ROOT_IN_HAND, held = 0u;
        traverse_fast_resume<COUNT> ( T, r, st, o_perm, best, top, hand, held, 0, c, REACH && pass == 1 );
        if ( !REACH || pass == 1 || !T.sc.reach || best.tri == 0xffffffffu || reference_reaches ( T, best.tri, r ) ) break;
    }
    return best;
}

// -----------------------------------------------------------------------------
// Resumable traversal for the decoupled render loop (large scenes; render_kernels.hip).
// The 64 lanes of a wave hold DIFFERENT rays at different stages; a lane's traversal state
// (stack column and leaf list in LDS; top, nleaf, closest hit in registers) survives leaving
// and re-entering these functions. They return as soon as the number of lanes that still
// have nodes to visit has dropped to `exit_active`, so that the finished lanes can be shaded
// and given their next ray instead of idling until the slowest ray of the wave is done
// (on the 97k-triangle hall a ray visits 474 nodes on average with a long tail: waiting for
// the slowest of 64 left 17 % of the lanes busy). What is computed per ray, and in which
// order, is exactly what traverse_loops computes.
// `traversing` is cleared for lanes whose traversal completed.
// -----------------------------------------------------------------------------
template <int COUNT, int MODE, bool FAST>
TD void traverse_resume ( const Tracer& T, const Ray& r, const SlabSel& sel, const RayState& st, V3 o_perm, Closest& best, int*& sp, bool& traversing, int exit_active, Counters& c ) {
    int* const lp_full = T.leaves + ( T.leaf_cap - 2 ) * TERRA_COL;
    int* lp = T.leaves;                                      // every lane's list is empty on entry and on exit
    for ( ;; ) {
        for ( ;; ) {
            const bool can = traversing && sp != T.stack && lp <= lp_full;
            const int n_can = __popcll ( __ballot ( can ) ), n_nodes = __popcll ( __ballot ( traversing && sp != T.stack ) );
            if ( n_can == 0 || n_nodes <= exit_active ) break;
            if ( can ) node_step<COUNT, MODE, FAST> ( T, r, sel, sp, lp, c );
        }
        for ( const int* e = T.leaves; e != lp; e += TERRA_COL ) leaf_step<COUNT, MODE> ( T, e, st, o_perm, best, c );        // lanes that are not traversing hold an empty list
        lp = T.leaves;
        if ( traversing && sp == T.stack ) traversing = false;
        if ( __popcll ( __ballot ( traversing ) ) <= exit_active ) break;
    }
}

// -----------------------------------------------------------------------------
// textures (reference src/Terra.c:368-466). uv is in TEXEL units, as the reference uses it
// ((size_t)uv->x); three consecutive components are read whatever `components` says, as the
// reference does; negative coordinates are undefined there and clamp to 0 here.
// -----------------------------------------------------------------------------
TD V3 texture_read ( const DevTexture& t, uint32_t x, uint32_t y ) {
    const uint32_t W = t.width, H = t.height;
    if ( t.address_mode == 2 ) { x = x < W - 1 ? x : W - 1; y = y < H - 1 ? y : H - 1; }
    else if ( t.address_mode == 0 ) { x %= W; y %= H; }
    else if ( ( x / W ) % 2 == 0 ) { x %= W; y %= H; }
    else { x = W - ( x % W ); y = H - ( y % H ); x = x < W - 1 ? x : W - 1; y = y < H - 1 ? y : H - 1; }
    const size_t e = ( ( size_t ) y * W + x ) * t.components;
    if ( t.depth == 1 ) {
        const uint8_t* p = reinterpret_cast<const uint8_t*> ( t.data ) + e;
        return v3 ( p[0] / 255.f, p[1] / 255.f, p[2] / 255.f );
    }
    const float* p = reinterpret_cast<const float*> ( t.data ) + e;
    return v3 ( p[0], p[1], p[2] );
}
TD V3 texture_sample ( const DevTexture& t, float u, float v ) {
    uint32_t ix = u > 0.f ? ( uint32_t ) u : 0u, iy = v > 0.f ? ( uint32_t ) v : 0u;
    if ( t.filter == 0 ) return texture_read ( t, ix, iy );
    if ( t.filter != 1 ) return v3 ( 0, 0, 0 );          // trilinear / anisotropic: unimplemented in the reference too (returns zero)
    uint32_t x2 = ix + 1 < t.width - 1 ? ix + 1 : t.width - 1, y2 = iy + 1 < t.height - 1 ? iy + 1 : t.height - 1;
    V3 n1 = texture_read ( t, ix, iy ), n2 = texture_read ( t, x2, iy ), n3 = texture_read ( t, ix, y2 ), n4 = texture_read ( t, x2, y2 );
    float wu = u - ( float ) ix, wv = v - ( float ) iy, wou = 1.f - wu, wov = 1.f - wv;
    return v3 ( ( n1.x * wou + n2.x * wu ) * wov + ( n3.x * wou + n4.x * wu ) * wv,
                ( n1.y * wou + n2.y * wu ) * wov + ( n3.y * wou + n4.y * wu ) * wv,
                ( n1.z * wou + n2.z * wu ) * wov + ( n3.z * wou + n4.z * wu ) * wv );
}

#define TERRA_PI_F 3.1416926535f        // the reference's terra_PI (include/TerraMath.h), not pi
// lat-long environment lookup by direction (reference src/Terra.c:468-477): nearest texel, no filtering.
// terra_PI exceeds pi, so phi / (2 terra_PI) and theta / terra_PI stay below 1 and the texel is in range.
TD V3 environment_eval ( const DevScene& sc, V3 dir ) {
    if ( sc.env_mode == 1 ) return v3 ( sc.env_color[0], sc.env_color[1], sc.env_color[2] );
    const DevTexture& t = sc.textures[sc.env_tex];
    V3 d = normalize ( dir );
    float theta = tdm_acosf ( d.y );
    float phi = tdm_atan2f ( d.z, d.x ) + TERRA_PI_F;
    uint32_t u = ( uint32_t ) ( ( phi / ( 2 * TERRA_PI_F ) ) * ( float ) t.width );
    uint32_t v = ( uint32_t ) ( ( theta / TERRA_PI_F ) * ( float ) t.height );
    return texture_read ( t, u, v );
}

// -----------------------------------------------------------------------------
// surface
// -----------------------------------------------------------------------------
// The reference also stores the tangent frame (terra_f4x4_basis of the normal) in the
// surface; it is a pure function of the normal, so it is rebuilt where it is consumed
// (diffuse sampling) instead of being carried in 9 registers.
struct Surface {
    V3    normal;
    V3    emissive;
    V3    attr[4];      // the presets use at most 4 slots; Phong slot 3.x and glass slots 2, 3.x are scratch written by sample()
    float ior;
    int   bsdf;
};

template <int MODE, int KINDS>
TD void surface_init ( const Tracer& T, uint32_t ti, V3 point, Surface& sf, uint32_t& object_out, uint32_t& tri_in_object_out, uint32_t& nattr_out ) {
    float4 t0, t1, t2, p0, p1, p2, p3x;
    if ( MODE == 1 ) {
        const float4* lt = reinterpret_cast<const float4*> ( T.l_tris );
        t0 = lt[3 * ti]; t1 = lt[3 * ti + 1]; t2 = lt[3 * ti + 2];
        p0 = T.l_props[4 * ti]; p1 = T.l_props[4 * ti + 1]; p2 = T.l_props[4 * ti + 2]; p3x = T.l_props[4 * ti + 3];
    } else {
        const float4* tris = reinterpret_cast<const float4*> ( MODE >= 2 ? T.sc.fast_tris : T.sc.tris );
        const float4* props = reinterpret_cast<const float4*> ( T.sc.props );
        t0 = tris[3 * ti]; t1 = tris[3 * ti + 1]; t2 = tris[3 * ti + 2];
        uint32_t pi = ti;
        if ( MODE >= 2 ) pi = T.sc.mats[__float_as_uint ( t0.w )].first_tri + __float_as_uint ( t1.w );
        p0 = props[4 * pi]; p1 = props[4 * pi + 1]; p2 = props[4 * pi + 2]; p3x = props[4 * pi + 3];
    }
    V3 ta = v3 ( t0.x, t0.y, t0.z ), tb = v3 ( t1.x, t1.y, t1.z ), tc = v3 ( t2.x, t2.y, t2.z );
    uint32_t object = __float_as_uint ( t0.w );
    object_out = object; tri_in_object_out = __float_as_uint ( t1.w );
    V3 e0 = tb - ta, e1 = tc - ta, p = point - ta;
    float d00 = dot ( e0, e0 ), d11 = dot ( e1, e1 ), d01 = dot ( e0, e1 );
    float dp0 = dot ( p, e0 ), dp1 = dot ( p, e1 );
    float div = d00 * d11 - d01 * d01;
    float u = ( d11 * dp0 - d01 * dp1 ) / div;
    float v = ( d00 * dp1 - d01 * dp0 ) / div;
    float w = 1 - u - v;
    V3 na = v3 ( p0.x, p0.y, p0.z ), nb = v3 ( p0.w, p1.x, p1.y ), nc = v3 ( p1.z, p1.w, p2.x );
    sf.normal = normalize ( ( nc * v + nb * u ) + na * w );
    const DevMaterial& m = T.l_mats[object];
    sf.emissive = v3p ( m.emissive );
    #pragma unroll
    for ( int i = 0; i < 4; ++i ) sf.attr[i] = v3p ( m.attributes[i] );
    if ( ( KINDS & TERRA_KIND_TEX ) && m.any_texture ) {       // textured attributes: interpolate the texcoord as the reference does (src/Terra.c:1748-1752) and sample
        V3 pa2 = v3 ( p2.y, p2.z, 0.f );            // texcoord_a
        V3 pb2 = v3 ( p2.w, p3x.x, 0.f );           // texcoord_b
        V3 pc2 = v3 ( p3x.y, p3x.z, 0.f );          // texcoord_c
        float tx = ( pc2.x * v + pb2.x * u ) + pa2.x * w;
        float ty = ( pc2.y * v + pb2.y * u ) + pa2.y * w;
        #pragma unroll
        for ( int i = 0; i < 4; ++i ) if ( m.tex[i] >= 0 ) sf.attr[i] = texture_sample ( T.sc.textures[m.tex[i]], tx, ty );
        if ( m.tex[TERRA_DEV_MAX_ATTR] >= 0 ) sf.emissive = texture_sample ( T.sc.textures[m.tex[TERRA_DEV_MAX_ATTR]], tx, ty );
    }
    sf.bsdf = m.bsdf;
    sf.ior = m.ior;
    nattr_out = m.attributes_count;
}

struct RaycastResult { bool hit; uint32_t object, tri_in_object, tri; V3 point; };

struct Azimuth { float sn, cs; bool have; };
struct PathDraws { float e0, e1, e2, e3; Azimuth az; };
template <int COUNT> TD PathDraws path_draw ( const float2* sincos24, Pcg32& rb, Counters& c );

// pre / rb (optional): a hit draws the path's four continuation variates (path_draw) BEFORE the surface is set up, so that the azimuth table load they issue
// is in flight during terra_surface_init's arithmetic -- only for integrators that draw nothing of their own between the hit and the BSDF sample
template <int COUNT, int MODE, int KINDS>
TD RaycastResult scene_raycast ( const Tracer& T, const Ray& in, Surface& sf, Counters& c, PathDraws* pre = nullptr, Pcg32* rb = nullptr ) {
    Ray r = in;
    r.o = r.o + r.d * 0.001f;
    RayState st = ray_state_init ( r );
    if ( COUNT ) ++c.rays;
    Closest best;
    if ( MODE >= 2 ) { ClosestRanked b2 = bvh_traverse_fast<COUNT, MODE == 3> ( T, r, st, c ); best.depth = b2.depth; best.tri = b2.tri; }
    else best = bvh_traverse<COUNT, MODE> ( T, r, st, c );
    RaycastResult res; res.hit = best.tri != 0xffffffffu; res.tri = best.tri; res.object = 0; res.tri_in_object = 0;
    res.point = res.hit ? r.o + r.d * best.depth : v3 ( FLT_MAX, FLT_MAX, FLT_MAX );
    if ( res.hit ) {
        uint32_t nattr;
        if ( pre ) *pre = path_draw<COUNT> ( T.sc.sincos24, *rb, c );
        surface_init<MODE, KINDS> ( T, best.tri, res.point, sf, res.object, res.tri_in_object, nattr );
        if ( MODE >= 2 ) res.tri = T.sc.mats[res.object].first_tri + res.tri_in_object;      // back to the soup index (lights, areas)
        if ( COUNT ) ++c.hits;
        if ( COUNT == 2 ) c.attr_fetches += nattr + 1;
    }
    return res;
}

// terra_scene_raycast for a ray of which only "which triangle is hit first" matters (the shadow ray of the Direct integrator, src/Terra.c:1349-1426, on scenes whose
// emissive attributes are constants): same traversal, same counts -- a hit is a surface initialisation in the reference -- without setting the surface up.
// Returns the triangle's index in the soup (the index the light tables use), 0xffffffff for a miss.
// `expected` (soup index): the only answer the caller distinguishes from the others; kernels without counters then take the shortcut of fast_expect on MODE 2
template <int COUNT, int MODE>
TD uint32_t scene_raycast_triangle ( const Tracer& T, const Ray& in, Counters& c, uint32_t expected ) {
    Ray r = in;
    r.o = r.o + r.d * 0.001f;
    RayState st = ray_state_init ( r );
    if ( COUNT ) ++c.rays;
    if constexpr ( TERRA_SHADOW_ANYHIT && MODE == 2 && COUNT == 0 ) {
        const V3 o_perm = v3 ( pick ( r.o, st.ix ), pick ( r.o, st.iy ), pick ( r.o, st.iz ) );
        ClosestRanked best;
        if ( !fast_expect ( T, st, o_perm, expected, best ) ) return 0xffffffffu;          // (not the expected triangle; without counters nothing else is asked)
        int* top = T.stack; uint32_t hand = TERRA_FAST_ROOT_IN_HAND, held = 0u;
        traverse_fast_resume<COUNT> ( T, r, st, o_perm, best, top, hand, held, 0, c, false, true );
        return best.tri == TERRA_TRI_EXPECTED ? expected : 0xffffffffu;
    }
    // LDS-resident scenes, every tree mode: the reference's traversal order is kept, so "comes first" is its own strict "<" -- with the closest hit preset to ONE ULP
    // BEYOND the expected triangle's depth, a triangle as near as the expected one wins exactly when the reference meets it earlier; the expected triangle itself, when
    // its leaf is reached, takes the record as it would; and the traversal ends at the first OTHER triangle that takes it (traverse_loops ANYHIT): up to there it has
    // made the reference's decisions one by one, after that none can change the answer. A ray that misses its triangle is not traced at all.
    if constexpr ( TERRA_SHADOW_ANYHIT && MODE == 1 && COUNT == 0 ) {
        const V3 o_perm = v3 ( pick ( r.o, st.ix ), pick ( r.o, st.iy ), pick ( r.o, st.iz ) );
        const float* t = T.l_tris + 12 * expected;
        const float pa[3] = { t[st.ix], t[st.iy], t[st.iz] }, pb[3] = { t[4 + st.ix], t[4 + st.iy], t[4 + st.iz] }, pc[3] = { t[8 + st.ix], t[8 + st.iy], t[8 + st.iz] };
        float depth;
        if ( !watertight_permuted ( pa, pb, pc, o_perm, st, depth ) ) return 0xffffffffu;
        Closest best; best.depth = __uint_as_float ( __float_as_uint ( depth + 0.f ) + 1u ); best.tri = 0xffffffffu;      // (depth >= 0; + 0.f: -0 -> +0)
        if ( TERRA_FUSED_SLAB && T.fused && __all ( ray_is_tame ( r ) ) ) traverse_loops<COUNT, MODE, true, true, true> ( T, r, st, o_perm, best, c, expected );
        else if ( __all ( ray_is_regular ( r ) ) ) traverse_loops<COUNT, MODE, true, false, true> ( T, r, st, o_perm, best, c, expected );
        else traverse_loops<COUNT, MODE, false, false, true> ( T, r, st, o_perm, best, c, expected );
        return best.tri;
    }
    uint32_t tri;
    if ( MODE >= 2 ) { const ClosestRanked b2 = bvh_traverse_fast<COUNT, MODE == 3> ( T, r, st, c ); tri = b2.tri; }
    else tri = bvh_traverse<COUNT, MODE> ( T, r, st, c ).tri;
    if ( tri == 0xffffffffu ) return tri;
    uint32_t object;
    if ( MODE == 1 ) object = __float_as_uint ( T.l_tris[12 * tri + 3] );
    else {
        const float4* tris = reinterpret_cast<const float4*> ( MODE >= 2 ? T.sc.fast_tris : T.sc.tris );
        const float4 t0 = tris[3 * tri];
        object = __float_as_uint ( t0.w );
        if ( MODE >= 2 ) tri = T.sc.mats[object].first_tri + __float_as_uint ( tris[3 * tri + 1].w );      // back to the soup index
    }
    if ( COUNT ) ++c.hits;
    if ( COUNT == 2 ) c.attr_fetches += T.l_mats[object].attributes_count + 1;
    return tri;
}

TD Ray surface_ray ( const Surface& sf, V3 p, V3 d, float sign ) {
    V3 off = sf.normal * ( 0.0001f * sign );
    return make_ray ( p + off, d );
}

// -----------------------------------------------------------------------------
// BSDF presets
// -----------------------------------------------------------------------------

// sin / cos of the azimuth 2 * terra_PI * e that the samplers make of one variate (src/TerraPresets.c:39-40). A stream-B variate is u24 * 2^-24
// (rng.h), so over the render path this is a pure function of 24 bits: DevScene::sincos24 tabulates it -- every entry computed by tdm_sincosf_pair itself, at
// library start-up (unit_kernels.hip terra_fill_sincos24) -- and one 8-byte load replaces ~100 double-precision instructions (the glibc algorithm restated in
// dev_math.h). Any other argument (unit-level calls with arbitrary variates, a sampler-driven first bounce) takes the computation.
TD Azimuth azimuth_none() { Azimuth a; a.sn = 0.f; a.cs = 1.f; a.have = false; return a; }
// the table entry of variate e, if e is one of the 2^24 stream-B values (the load is issued here; the caller uses it as late as it can)
TD Azimuth azimuth_fetch ( const float2* tab, float e ) {
    Azimuth a = azimuth_none();
    const float x = e * 16777216.f;
    if ( tab && e >= 0.f && x < 16777216.f ) {
        const uint32_t k = ( uint32_t ) x;
        if ( ( float ) k == x ) { const float2 v = tab[k]; a.cs = v.x; a.sn = v.y; a.have = true; }
    }
    return a;
}
TD void azimuth_sincos ( const Azimuth& az, float e, float& sn, float& cs ) {
    if ( az.have ) { sn = az.sn; cs = az.cs; }
    else tdm_sincosf_pair ( 2 * TERRA_PI_F * e, sn, cs );
}

TD V3 diffuse_sample ( const Surface& sf, float e1, float e2, const Azimuth& az ) {
    float r = sqrtf ( e1 );
    float sn, cs;
    azimuth_sincos ( az, e2, sn, cs );
    float x = r * cs;
    float z = r * sn;
    V3 wi = v3 ( x, sqrtf ( sel_max ( 0.f, 1 - e1 ) ), z );
    return normalize ( basis_apply ( make_basis ( sf.normal ), wi ) );
}
TD float diffuse_pdf ( const Surface& sf, V3 wi ) { return sel_max ( 0.f, dot ( sf.normal, wi ) ) / TERRA_PI_F; }
TD V3 diffuse_eval ( const Surface& sf ) { return sf.attr[0] * ( float ) ( 1. / ( double ) TERRA_PI_F ); }

TD void phong_kd_ks ( const Surface& sf, float& kd, float& ks ) {
    V3 al = sf.attr[1], sp = sf.attr[0];
    float diffuse = sel_max ( al.x + al.y + al.z, ( float ) 1e-4 );
    float specular = sp.x + sp.y + sp.z;
    if ( specular > diffuse ) { kd = 0.5f * diffuse / specular; ks = 1.f - kd; }
    else { ks = 0.5f * specular / diffuse; kd = 1.f - ks; }
}
TD V3 phong_reflect ( const Surface& sf, V3 wo ) { return sf.normal * ( 2.f * dot ( wo, sf.normal ) ) - wo; }

TD V3 phong_sample ( Surface& sf, float e1, float e2, float e3, V3 wo, const Azimuth& az ) {
    float kd, ks; phong_kd_ks ( sf, kd, ks );
    if ( e3 < kd ) { sf.attr[3].x = 1.f; return diffuse_sample ( sf, e1, e2, az ); }
    sf.attr[3].x = -1.f;
    V3 wr = phong_reflect ( sf, wo );
    Basis b = make_basis ( wr );
    float phi = 2 * TERRA_PI_F * e1;
    float theta = tdm_acosf ( tdm_powf ( 1.f - e2, 1.f / ( sf.attr[2].x + 1 ) ) );
    float sin_theta = tdm_sinf ( theta );
    V3 wi = v3 ( sin_theta * tdm_cosf ( phi ), tdm_cosf ( theta ), sin_theta * tdm_sinf ( phi ) );
    return normalize ( basis_apply ( b, wi ) );
}
TD float phong_pdf ( const Surface& sf, V3 wi, V3 wo ) {
    if ( sf.attr[3].x == 1.f ) return diffuse_pdf ( sf, wi );
    V3 wr = phong_reflect ( sf, wo );
    float cos_alpha = dot ( wi, wr );
    float n = sf.attr[2].x;
    return ( n + 1 ) / ( 2 * TERRA_PI_F ) * tdm_powf ( cos_alpha, n );
}
TD V3 phong_eval ( const Surface& sf, V3 wi, V3 wo ) {
    float kd, ks; phong_kd_ks ( sf, kd, ks );
    float n = sf.attr[2].x;
    V3 diffuse_term = sf.attr[1] * ( kd * 1.f / TERRA_PI_F );
    V3 wr = phong_reflect ( sf, wo );
    float cos_alpha = dot ( wi, wr );
    float cos_n_alpha = tdm_powf ( cos_alpha, n );
    V3 specular_term = sf.attr[0] * ( ks * cos_n_alpha * ( n + 2 ) / ( 2 * TERRA_PI_F ) );
    return diffuse_term + specular_term;
}

// ---- GGX conductor and dielectric glass: defined by this repo (include/TerraPresets.h), no live
// reference; building blocks from the reference's dead code (src/TerraPresets.c:303-320, 333-343, 399-449)
TD float ggx_D ( float NoH, float alpha2 ) {
    if ( NoH <= 0.f ) return 0.f;
    float NoH2 = NoH * NoH;
    float den = NoH2 * alpha2 + ( 1 - NoH2 );
    return alpha2 / ( TERRA_PI_F * den * den );
}
TD float ggx_G1 ( V3 v, V3 n, V3 h, float alpha2 ) {
    float VoH = dot ( v, h ), VoN = dot ( v, n );
    if ( VoH / VoN <= 0.f ) return 0.f;
    float VoN2 = VoN * VoN;
    float tan2 = ( 1.f - VoN2 ) / VoN2;        // Smith G1 w.r.t. the normal (Walter 2007 eq. 34), see oracle note
    return 2.f / ( sqrtf ( 1 + alpha2 * tan2 ) + 1 );
}
TD V3 ggx_sample ( const Surface& sf, float e1, float e2, V3 wo, const Azimuth& az ) {
    float alpha = sf.attr[1].x;
    float t2 = alpha * alpha * e1 / ( 1.f - e1 );
    float cos_t = 1.f / sqrtf ( 1.f + t2 );
    float sin_t = sqrtf ( sel_max ( 0.f, 1.f - cos_t * cos_t ) );
    float sn, cs;
    azimuth_sincos ( az, e2, sn, cs );
    V3 h = v3 ( sin_t * cs, cos_t, sin_t * sn );
    h = normalize ( basis_apply ( make_basis ( sf.normal ), h ) );
    float HoV = sel_max ( 0.f, dot ( h, wo ) );
    return h * ( 2 * HoV ) - wo;
}
TD float ggx_pdf ( const Surface& sf, V3 wi, V3 wo ) {
    float alpha = sf.attr[1].x;
    V3 h = normalize ( wi + wo );
    float NoH = dot ( sf.normal, h ), HoV = dot ( h, wo );
    if ( HoV <= 0.f ) return 0.f;
    return ggx_D ( NoH, alpha * alpha ) * NoH / ( 4.f * HoV );
}
TD V3 ggx_eval ( const Surface& sf, V3 wi, V3 wo ) {
    float alpha = sf.attr[1].x, alpha2 = alpha * alpha;
    float NoL = dot ( sf.normal, wi ), NoV = dot ( sf.normal, wo );
    if ( NoL <= 0.f || NoV <= 0.f ) return v3 ( 0, 0, 0 );
    V3 h = normalize ( wi + wo );
    float NoH = dot ( sf.normal, h ), HoV = sel_max ( 0.f, dot ( h, wo ) );
    float m = 1.f - HoV, m2 = m * m, w5 = m2 * m2 * m;
    V3 F0 = sf.attr[0];
    V3 F = v3 ( F0.x + ( 1.f - F0.x ) * w5, F0.y + ( 1.f - F0.y ) * w5, F0.z + ( 1.f - F0.z ) * w5 );
    float G = ggx_G1 ( wo, sf.normal, h, alpha2 ) * ggx_G1 ( wi, sf.normal, h, alpha2 );
    float k = G * ggx_D ( NoH, alpha2 ) / ( 4.f * NoL * NoV );
    return F * k;
}
TD V3 glass_sample ( Surface& sf, float e3, V3 wo ) {
    V3 normal = sf.normal, incident = neg ( wo );
    float n1, n2, cos_i = dot ( normal, incident );
    if ( cos_i > 0.f ) { n1 = sf.ior; n2 = 1.f; normal = neg ( normal ); }
    else { n1 = 1.f; n2 = sf.ior; cos_i = -cos_i; }
    V3 refl = incident - normal * ( 2 * dot ( normal, incident ) );
    float nni = n1 / n2;
    float cos_t2 = 1.f - nni * nni * ( 1.f - cos_i * cos_i );
    V3 dir; float prob;
    if ( cos_t2 < 0.f ) { dir = refl; prob = 1.f; }
    else {
        float cos_t = sqrtf ( cos_t2 );
        float t = 1.f - ( n1 <= n2 ? cos_i : cos_t );
        float R0 = ( n1 - n2 ) / ( n1 + n2 ); R0 *= R0;
        float R = R0 + ( 1 - R0 ) * ( t * t * t * t * t );
        if ( e3 < R ) { dir = refl; prob = R; }
        else {
            V3 tv = normal * ( nni * cos_i - cos_t ), tn = incident * nni;
            dir = normalize ( tv + tn ); prob = 1 - R;
        }
    }
    sf.attr[2] = dir; sf.attr[3].x = prob;
    return dir;
}
TD bool glass_is_sampled ( const Surface& sf, V3 wi ) {
    return sf.attr[3].x > 0.f && wi.x == sf.attr[2].x && wi.y == sf.attr[2].y && wi.z == sf.attr[2].z;
}
TD float glass_pdf ( const Surface& sf, V3 wi ) { return glass_is_sampled ( sf, wi ) ? sf.attr[3].x : 0.f; }
TD V3 glass_eval ( const Surface& sf, V3 wi ) {
    if ( !glass_is_sampled ( sf, wi ) ) return v3 ( 0, 0, 0 );
    float k = sf.attr[3].x / dot ( sf.normal, wi );
    return sf.attr[0] * k;
}

// BSDF dispatch. KINDS is a compile-time mask of the preset kinds present in the committed scene
// (bit k = DevBsdfKind k): a diffuse-only scene compiles to straight-line diffuse code, which is
// what keeps the Simple kernel inside 96 VGPRs (5 waves/SIMD) without scratch.
#define TERRA_KINDS_ALL 127
// az: the azimuth of the SECOND variate (e2), if its table entry was fetched (azimuth_fetch) -- what the diffuse and GGX samplers and Phong's diffuse branch use
template <int KINDS>
TD V3 bsdf_sample ( Surface& sf, float e1, float e2, float e3, V3 wo, const Azimuth& az ) {
    if ( ( KINDS & 2 ) && ( KINDS == 2 || sf.bsdf == kDevBsdfPhong ) ) return phong_sample ( sf, e1, e2, e3, wo, az );
    if ( ( KINDS & 4 ) && ( KINDS == 4 || sf.bsdf == kDevBsdfGGX ) ) return ggx_sample ( sf, e1, e2, wo, az );
    if ( ( KINDS & 8 ) && ( KINDS == 8 || sf.bsdf == kDevBsdfGlass ) ) return glass_sample ( sf, e3, wo );
    return diffuse_sample ( sf, e1, e2, az );
}
template <int KINDS>
TD float bsdf_pdf ( const Surface& sf, V3 wi, V3 wo ) {
    if ( ( KINDS & 2 ) && ( KINDS == 2 || sf.bsdf == kDevBsdfPhong ) ) return phong_pdf ( sf, wi, wo );
    if ( ( KINDS & 4 ) && ( KINDS == 4 || sf.bsdf == kDevBsdfGGX ) ) return ggx_pdf ( sf, wi, wo );
    if ( ( KINDS & 8 ) && ( KINDS == 8 || sf.bsdf == kDevBsdfGlass ) ) return glass_pdf ( sf, wi );
    return diffuse_pdf ( sf, wi );
}
template <int KINDS>
TD V3 bsdf_eval ( const Surface& sf, V3 wi, V3 wo ) {
    if ( ( KINDS & 2 ) && ( KINDS == 2 || sf.bsdf == kDevBsdfPhong ) ) return phong_eval ( sf, wi, wo );
    if ( ( KINDS & 4 ) && ( KINDS == 4 || sf.bsdf == kDevBsdfGGX ) ) return ggx_eval ( sf, wi, wo );
    if ( ( KINDS & 8 ) && ( KINDS == 8 || sf.bsdf == kDevBsdfGlass ) ) return glass_eval ( sf, wi );
    return diffuse_eval ( sf );
}

// -----------------------------------------------------------------------------
// lights and integrators
// -----------------------------------------------------------------------------
// COUNT levels: 0 none (the default launch); 2 = rays, nodes, triangle tests, hits, stream-B draws, attribute fetches per lane
TD float randf ( Pcg32& b, Counters& c, int count ) { if ( count == 2 ) ++c.rand_calls; return trng_b_float ( b ); }

TD float triangle_area ( V3 a, V3 b, V3 cc ) { return length ( cross ( b - a, cc - a ) ) / 2; }

struct LightSample { uint32_t light_object; uint32_t tri_in_object; uint32_t tri; float pick_pdf; V3 pos, norm; };

// (MODE 1: the light's triangle and vertex normals come from the block's LDS copy of the scene)
template <int COUNT, int MODE = 0>
TD LightSample draw_light_sample ( const DevScene& sc, Pcg32& rb, Counters& c, const Tracer* T = nullptr ) {
    LightSample ls;
    float e = ( float ) ( ( double ) randf ( rb, c, COUNT ) - 1e-4 );
    double xl = ( double ) e * ( double ) sc.n_lights;
    uint32_t li = xl < 0 ? 0u : ( uint32_t ) xl;
    ls.pick_pdf = 1.f / ( float ) sc.lights_triangles_count;
    DevLight l = ( MODE == 1 && T ) ? T->l_lights[li] : sc.lights[li];
    float e_t = randf ( rb, c, COUNT );
    uint32_t k = ( uint32_t ) ( e_t * ( float ) l.tri_count );
    if ( k >= l.tri_count ) k = l.tri_count - 1;
    ls.light_object = l.object; ls.tri_in_object = k; ls.tri = l.first_tri + k;
    float e1 = randf ( rb, c, COUNT ), e2 = randf ( rb, c, COUNT );
    const float4* tris = ( MODE == 1 && T ) ? reinterpret_cast<const float4*> ( T->l_tris ) : reinterpret_cast<const float4*> ( sc.tris );
    const float4* props = ( MODE == 1 && T ) ? T->l_props : reinterpret_cast<const float4*> ( sc.props );
    float4 t0 = tris[3 * ls.tri + 0], t1 = tris[3 * ls.tri + 1], t2 = tris[3 * ls.tri + 2];
    float4 p0 = props[4 * ls.tri + 0], p1 = props[4 * ls.tri + 1], p2 = props[4 * ls.tri + 2];
    float s = sqrtf ( e1 );
    float a = 1 - s, b = e2 * s, cw = 1 - a - b;
    ls.pos = ( v3 ( t0.x, t0.y, t0.z ) * a + v3 ( t1.x, t1.y, t1.z ) * b ) + v3 ( t2.x, t2.y, t2.z ) * cw;
    V3 n = ( v3 ( p0.x, p0.y, p0.z ) * a + v3 ( p0.w, p1.x, p1.y ) * b ) + v3 ( p1.z, p1.w, p2.x ) * cw;
    ls.norm = normalize ( n );
    return ls;
}

// -----------------------------------------------------------------------------
// Environment importance sampling (extension, UNPINNED: nothing in the reference calls its TerraDistribution2D, src/Terra.c:812-846 -- the wiring is this
// repo's definition, restated by the oracle's environment_light_sample). One sample per shaded hit of Direct / Direct+MIS, after their own light samples: two draws of
// stream B pick a texel of the lat-long map through the table (e1: the row, e2: the column inside it; terra_distribution_2d_sample's arithmetic); the direction is the
// inverse of the lookup's mapping (src/Terra.c:468-477: theta = v terra_PI, phi = u 2 terra_PI - terra_PI); density over the sphere = texel probability x texels /
// (2 terra_PI^2 sin theta); the sample counts when the direction is in the upper hemisphere of the shading normal and its shadow ray leaves the scene; radiance = the
// chosen texel. Returns the term before the path throughput. Compiled into the KINDS & TERRA_KIND_SAMPLER kernels only.
// -----------------------------------------------------------------------------
TD bool env_sampling_active ( const DevScene& sc ) { return sc.env_nx != 0u; }
// with environment sampling in a light integrator the environment reaches a path through the samples taken at its hits: only the camera ray adds it on leaving the scene
template <int INTEGRATOR, int KINDS>
TD bool env_reaches_by_samples ( const DevScene& sc, uint32_t bounce ) {
    if constexpr ( ( KINDS & TERRA_KIND_SAMPLER ) != 0 && ( INTEGRATOR == 1 || INTEGRATOR == 2 ) ) return bounce != 0u && env_sampling_active ( sc );
    return false;
}
template <int COUNT, int MODE, int KINDS>
TD V3 environment_light_sample ( const Tracer& T, Surface& sf, V3 p, V3 wo, Pcg32& rb, Counters& c ) {
    const DevScene& sc = T.sc;
    const V3 zero = v3 ( 0, 0, 0 );
    const float e1 = randf ( rb, c, COUNT ), e2 = randf ( rb, c, COUNT );
    float p_row = 0.f, p_col = 0.f; uint32_t row = 0, col = 0;
    DevDistribution1D rows = { sc.env_row_f, sc.env_row_cdf, sc.env_ny, sc.env_integral, sc.env_monotone };
    const float sv = distribution_sample ( rows, e1, &p_row, &row );
    if ( sv == FLT_MAX ) return zero;
    DevDistribution1D cols = { sc.env_f + ( size_t ) sc.env_nx * row, sc.env_cdf + ( size_t ) sc.env_nx * row, sc.env_nx, sc.env_row_f[row], sc.env_monotone };
    const float su = distribution_sample ( cols, e2, &p_col, &col );
    if ( su == FLT_MAX ) return zero;
    const float theta = sv * TERRA_PI_F, phi = su * ( 2 * TERRA_PI_F ) - TERRA_PI_F;
    const float st = tdm_sinf ( theta ), ct = tdm_cosf ( theta ), sp = tdm_sinf ( phi ), cp = tdm_cosf ( phi );
    if ( ! ( st > 0.f ) ) return zero;
    const V3 wi = v3 ( st * cp, ct, st * sp );
    const float cosine = dot ( wi, sf.normal );
    if ( ! ( cosine > 0.f ) ) return zero;
    const float pdf = ( p_row * p_col ) * ( ( float ) sc.env_nx * ( float ) sc.env_ny ) / ( 2 * TERRA_PI_F * TERRA_PI_F * st );
    if ( ! ( pdf > 0.f ) ) return zero;
    Surface lsf;
    Ray r = surface_ray ( sf, p, wi, 1.f );
    RaycastResult h = scene_raycast<COUNT, MODE, KINDS> ( T, r, lsf, c );
    if ( h.hit ) return zero;
    const V3 L = texture_read ( sc.textures[sc.env_tex], col, row );
    const V3 f = bsdf_eval<KINDS> ( sf, wi, wo );
    return had ( L, f ) * ( cosine / pdf );
}

template <int COUNT, int MODE, int KINDS>
TD V3 integrate_direct ( const Tracer& T, Surface& sf, V3 p, V3 wo, V3 throughput, uint32_t bounce, Pcg32& rb, Counters& c ) {
    const DevScene& sc = T.sc;
    V3 Lo = v3 ( 0, 0, 0 );
    if ( bounce == 0 && dot ( wo, sf.normal ) > 0 ) Lo = Lo + sf.emissive;
    LightSample ls = draw_light_sample<COUNT, MODE> ( sc, rb, c, &T );
    V3 p_to_light = ls.pos - p;
    V3 wi = normalize ( p_to_light );
    Surface lsf;
    Ray r = surface_ray ( sf, p, wi, 1.f );
    RaycastResult h = scene_raycast<COUNT, MODE, KINDS> ( T, r, lsf, c );
    if ( h.hit && h.object == ls.light_object && h.tri_in_object == ls.tri_in_object ) {
        float cosv = dot ( neg ( wi ), ls.norm );
        if ( cosv > 0 ) {
            V3 f = bsdf_eval<KINDS> ( sf, wi, wo );
            float pdf = dot ( p_to_light, p_to_light ) / fabsf ( cosv * T.l_area[h.tri] );
            V3 Ld = had ( lsf.emissive, f );
            Ld = Ld * ( dot ( wi, sf.normal ) / ( pdf * ls.pick_pdf ) );
            Lo = Lo + Ld;
        }
    }
    if constexpr ( ( KINDS & TERRA_KIND_SAMPLER ) != 0 ) { if ( env_sampling_active ( sc ) ) Lo = Lo + environment_light_sample<COUNT, MODE, KINDS> ( T, sf, p, wo, rb, c ); }
    return had ( Lo, throughput );
}

// integrate_direct split at its shadow ray, for the decoupled loop (render_kernels.hip): everything that does not depend
// on the shadow ray's outcome is done up front -- same operations in the same order -- and both possible return values
// are kept: `hid` (light sample not visible) and `vis` (visible). Valid for scenes without textured attributes, where the
// emissive the shadow ray's surface_init would read is the light material's constant.
struct DirectPending { V3 vis, hid; uint32_t expected; };
// MODE: where the light's triangle, the materials and the areas are read from (1: the block's LDS copies, through T.l_*)
template <int COUNT, int KINDS, int MODE = 0>
TD DirectPending direct_prepare ( const Tracer& T, Surface& sf, V3 p, V3 wo, V3 throughput, uint32_t bounce, Pcg32& rb, Counters& c, Ray& shadow_ray ) {
    const DevScene& sc = T.sc;
    V3 Lo = v3 ( 0, 0, 0 );
    if ( bounce == 0 && dot ( wo, sf.normal ) > 0 ) Lo = Lo + sf.emissive;
    LightSample ls = draw_light_sample<COUNT, MODE> ( sc, rb, c, &T );
    V3 p_to_light = ls.pos - p;
    V3 wi = normalize ( p_to_light );
    shadow_ray = surface_ray ( sf, p, wi, 1.f );
    DirectPending d;
    d.hid = had ( Lo, throughput ); d.vis = d.hid; d.expected = ls.tri;
    float cosv = dot ( neg ( wi ), ls.norm );
    if ( cosv > 0 ) {
        V3 f = bsdf_eval<KINDS> ( sf, wi, wo );
        float pdf = dot ( p_to_light, p_to_light ) / fabsf ( cosv * T.l_area[ls.tri] );
        V3 Ld = had ( v3p ( T.l_mats[ls.light_object].emissive ), f );
        Ld = Ld * ( dot ( wi, sf.normal ) / ( pdf * ls.pick_pdf ) );
        d.vis = had ( Lo + Ld, throughput );
    }
    return d;
}

// integrate_mis (DEBUG_WEIGHTS = false) split at its two rays, for the decoupled loop. mis_prepare does everything that
// precedes the light-sample shadow ray (job A) and prepares both of its outcomes as the integrator's running sum
// (a_hid: emissive term only; a_vis: + the light-sample term); it also evaluates what the BSDF-sample ray (job B) will
// need from the shaded surface. mis_finish_b applies job B's hit to the running sum exactly as integrate_mis does.
// Job A's visible term uses the light material's constant emissive: valid for scenes without textured attributes.
struct MisPending { V3 a_vis, a_hid; uint32_t expected; V3 f2; float bpdf2, cos2; V3 p; uint32_t light_object; V3 t_before; };
template <int COUNT, int KINDS, int MODE = 0>
TD MisPending mis_prepare ( const Tracer& T, Surface& sf, V3 p, V3 wo, V3 throughput, uint32_t bounce, Pcg32& rb, Counters& c, Ray& ray_a, V3& dir_b ) {
    const DevScene& sc = T.sc;
    V3 Lo = v3 ( 0, 0, 0 );
    if ( bounce == 0 && dot ( wo, sf.normal ) > 0 ) Lo = Lo + sf.emissive;
    float e1 = randf ( rb, c, COUNT ), e2 = randf ( rb, c, COUNT ), e3 = randf ( rb, c, COUNT );
    V3 bsdf_dir = bsdf_sample<KINDS> ( sf, e1, e2, e3, wo, azimuth_fetch ( sc.sincos24, e2 ) );
    LightSample ls = draw_light_sample<COUNT, MODE> ( sc, rb, c, &T );
    MisPending m;
    m.a_hid = Lo; m.a_vis = Lo; m.expected = ls.tri; m.p = p; m.light_object = ls.light_object; m.t_before = throughput;
    {
        V3 p_to_light = ls.pos - p;
        V3 wi = normalize ( p_to_light );
        ray_a = surface_ray ( sf, p, wi, 1.f );
        float cosv = dot ( ls.norm, neg ( wi ) );
        if ( cosv > 0 ) {
            float bpdf = bsdf_pdf<KINDS> ( sf, wi, wo );
            float lpdf = dot ( p_to_light, p_to_light ) / fabsf ( cosv * T.l_area[ls.tri] );
            float weight = ( lpdf * lpdf ) / ( lpdf * lpdf + bpdf * bpdf );
            if ( lpdf != 0 ) {
                V3 f = bsdf_eval<KINDS> ( sf, wi, wo );
                V3 L = had ( v3p ( T.l_mats[ls.light_object].emissive ), f );
                L = L * ( dot ( wi, sf.normal ) * weight / ( lpdf * ls.pick_pdf ) );
                m.a_vis = Lo + L;
            }
        }
    }
    dir_b = bsdf_dir;
    m.f2 = bsdf_eval<KINDS> ( sf, bsdf_dir, wo );
    m.bpdf2 = bsdf_pdf<KINDS> ( sf, bsdf_dir, wo );
    m.cos2 = dot ( bsdf_dir, sf.normal );
    return m;
}
// job B came back with closest hit (tri, point, shaded surface lsf of the hit): returns the integrator's value
template <int MODE>
TD V3 mis_finish_b ( const Tracer& T, const MisPending& m, V3 Lo, bool hit, uint32_t hit_object, uint32_t hit_tri, V3 hit_point, const Surface& lsf, V3 wi ) {
    if ( hit && hit_object == m.light_object ) {
        float NoW = dot ( lsf.normal, neg ( wi ) );
        if ( NoW > 0 ) {
            V3 dl = m.p - hit_point;
            float dist = dot ( dl, dl );
            const float4* tris = MODE == 1 ? reinterpret_cast<const float4*> ( T.l_tris ) : reinterpret_cast<const float4*> ( T.sc.tris );      // (hit_tri: index in the soup)
            float4 t0 = tris[3 * hit_tri + 0], t1 = tris[3 * hit_tri + 1], t2 = tris[3 * hit_tri + 2];
            float area = triangle_area ( v3 ( t0.x, t0.y, t0.z ), v3 ( t1.x, t1.y, t1.z ), v3 ( t2.x, t2.y, t2.z ) );
            float lpdf = dist / ( NoW * area );
            float weight = ( m.bpdf2 * m.bpdf2 ) / ( lpdf * lpdf + m.bpdf2 * m.bpdf2 );
            if ( m.bpdf2 != 0 ) {
                V3 L = had ( lsf.emissive, m.f2 );
                L = L * ( m.cos2 * weight / m.bpdf2 );
                Lo = Lo + L;
            }
        }
    }
    return had ( Lo, m.t_before );
}

template <int COUNT, int MODE, int KINDS, bool DEBUG_WEIGHTS>
TD V3 integrate_mis ( const Tracer& T, Surface& sf, V3 p, V3 wo, V3 throughput, uint32_t bounce, Pcg32& rb, Counters& c ) {
    const DevScene& sc = T.sc;
    V3 Lo = v3 ( 0, 0, 0 );
    if ( DEBUG_WEIGHTS ) { if ( bounce != 0 ) return Lo; }
    else if ( bounce == 0 && dot ( wo, sf.normal ) > 0 ) Lo = Lo + sf.emissive;
    float e1 = randf ( rb, c, COUNT ), e2 = randf ( rb, c, COUNT ), e3 = randf ( rb, c, COUNT );
    V3 bsdf_dir = bsdf_sample<KINDS> ( sf, e1, e2, e3, wo, azimuth_fetch ( sc.sincos24, e2 ) );
    LightSample ls = draw_light_sample<COUNT, MODE> ( sc, rb, c, &T );
    {
        V3 p_to_light = ls.pos - p;
        V3 wi = normalize ( p_to_light );
        Surface lsf;
        Ray r = surface_ray ( sf, p, wi, 1.f );
        RaycastResult h = scene_raycast<COUNT, MODE, KINDS> ( T, r, lsf, c );
        if ( h.hit && h.object == ls.light_object && h.tri_in_object == ls.tri_in_object ) {
            float cosv = dot ( ls.norm, neg ( wi ) );
            if ( cosv > 0 ) {
                float bpdf = bsdf_pdf<KINDS> ( sf, wi, wo );
                float lpdf = dot ( p_to_light, p_to_light ) / fabsf ( cosv * T.l_area[h.tri] );
                if ( DEBUG_WEIGHTS ) {
                    float weight = ( bpdf * bpdf ) / ( lpdf * lpdf + bpdf * bpdf );
                    Lo = Lo + v3 ( 0, 0, weight );
                } else {
                    float weight = ( lpdf * lpdf ) / ( lpdf * lpdf + bpdf * bpdf );
                    if ( lpdf != 0 ) {
                        V3 f = bsdf_eval<KINDS> ( sf, wi, wo );
                        V3 L = had ( lsf.emissive, f );
                        L = L * ( dot ( wi, sf.normal ) * weight / ( lpdf * ls.pick_pdf ) );
                        Lo = Lo + L;
                    }
                }
            }
        }
    }
    {
        V3 wi = bsdf_dir;
        V3 f = bsdf_eval<KINDS> ( sf, wi, wo );
        float bpdf = bsdf_pdf<KINDS> ( sf, wi, wo );
        V3 light_wo = neg ( wi );
        Surface lsf;
        Ray r = surface_ray ( sf, p, wi, 1.f );
        RaycastResult h = scene_raycast<COUNT, MODE, KINDS> ( T, r, lsf, c );
        if ( h.hit && h.object == ls.light_object ) {
            float NoW = dot ( lsf.normal, light_wo );
            if ( NoW > 0 ) {
                V3 dl = p - h.point;
                float dist = dot ( dl, dl );
                const float4* tris = reinterpret_cast<const float4*> ( sc.tris );
                float4 t0 = tris[3 * h.tri + 0], t1 = tris[3 * h.tri + 1], t2 = tris[3 * h.tri + 2];
                float area = triangle_area ( v3 ( t0.x, t0.y, t0.z ), v3 ( t1.x, t1.y, t1.z ), v3 ( t2.x, t2.y, t2.z ) );
                float lpdf = dist / ( NoW * area );
                float weight = ( bpdf * bpdf ) / ( lpdf * lpdf + bpdf * bpdf );
                if ( DEBUG_WEIGHTS ) {
                    Lo = Lo + v3 ( weight, 0, 0 );
                } else if ( bpdf != 0 ) {
                    V3 L = had ( lsf.emissive, f );
                    L = L * ( dot ( wi, sf.normal ) * weight / bpdf );
                    Lo = Lo + L;
                }
            }
        }
    }
    if constexpr ( ( KINDS & TERRA_KIND_SAMPLER ) != 0 && !DEBUG_WEIGHTS ) { if ( env_sampling_active ( sc ) ) Lo = Lo + environment_light_sample<COUNT, MODE, KINDS> ( T, sf, p, wo, rb, c ); }
    return had ( Lo, throughput );
}

TD V3 integrate_debug_normals ( const Surface& sf, uint32_t bounce ) {
    if ( bounce != 0 ) return v3 ( 0, 0, 0 );
    V3 n = sf.normal;
    V3 pp = v3 ( sel_min ( n.x > 0 ? n.x : 0.f, 1.f ), sel_min ( n.y > 0 ? n.y : 0.f, 1.f ), sel_min ( n.z > 0 ? n.z : 0.f, 1.f ) );
    V3 nn = v3 ( sel_min ( n.x > -1 ? n.x : -1.f, 0.f ), sel_min ( n.y > -1 ? n.y : -1.f, 0.f ), sel_min ( n.z > -1 ? n.z : -1.f, 0.f ) );
    nn = nn * -1.f;
    V3 col = v3 ( 0, 0, 0 );
    col = col + v3 ( 1, 0, 0 ) * pp.x;
    col = col + v3 ( 0, 1, 0 ) * pp.y;
    col = col + v3 ( 0, 0, 1 ) * pp.z;
    col = col + v3 ( 0, 1, 1 ) * nn.x;
    col = col + v3 ( 1, 0, 1 ) * nn.y;
    col = col + v3 ( 1, 1, 0 ) * nn.z;
    return col;
}

// integrator ids = TerraIntegrator (reference include/Terra.h:149-157)
template <int INTEGRATOR, int COUNT, int MODE, int KINDS>
TD V3 integrate ( const Tracer& T, const Ray& ray, Surface& sf, V3 p, V3 wo, V3 throughput, uint32_t bounce, Pcg32& rb, Counters& c ) {
    if ( INTEGRATOR == 0 ) {
        if ( dot ( wo, sf.normal ) > 0 ) return had ( sf.emissive, throughput );
        return v3 ( 0, 0, 0 );
    } else if ( INTEGRATOR == 1 ) {
        return integrate_direct<COUNT, MODE, KINDS> ( T, sf, p, wo, throughput, bounce, rb, c );
    } else if ( INTEGRATOR == 2 ) {
        return integrate_mis<COUNT, MODE, KINDS, false> ( T, sf, p, wo, throughput, bounce, rb, c );
    } else if ( INTEGRATOR == 3 ) {
        return bounce != 0 ? v3 ( 0, 0, 0 ) : v3 ( 1, 1, 1 );
    } else if ( INTEGRATOR == 4 ) {
        if ( bounce != 0 ) return v3 ( 0, 0, 0 );
        float d = length ( ray.o - p ) / 500.f;
        return v3 ( d, d, d );
    } else if ( INTEGRATOR == 5 ) {
        return integrate_debug_normals ( sf, bounce );
    } else {
        return integrate_mis<COUNT, MODE, KINDS, true> ( T, sf, p, wo, throughput, bounce, rb, c );
    }
}

// The tail of one terra_trace iteration after the integrator's term (reference src/Terra.c:1066-1094): sample the BSDF, weight the
// throughput, play Russian roulette. Returns true when the path goes on (then `bounce` was advanced and wi is the next direction; the
// caller forms the next ray from the hit point). Same operations, draws and order in all four loops of the kernel.
// The four variates are consecutive draws of stream B whatever the surface is, so they can be drawn -- and the azimuth table entry requested -- BEFORE the
// surface is set up (path_draw), which hides the load behind terra_surface_init's work; integrators that draw from the stream themselves (Direct, MIS) call
// path_draw after their own draws, as the reference's order demands.
template <int COUNT>
TD PathDraws path_draw ( const float2* sincos24, Pcg32& rb, Counters& c ) {
    PathDraws d;
    d.e0 = randf ( rb, c, COUNT ); d.e1 = randf ( rb, c, COUNT ); d.e2 = randf ( rb, c, COUNT );
    d.az = azimuth_fetch ( sincos24, d.e1 );
    d.e3 = randf ( rb, c, COUNT );
    return d;
}
template <int KINDS>
TD bool path_continue ( Surface& sf, V3 wo, V3& throughput, uint32_t& bounce, uint32_t max_bounces, const PathDraws& d, V3& wi ) {
    wi = bsdf_sample<KINDS> ( sf, d.e0, d.e1, d.e2, wo, d.az );
    float pdf = sel_max ( bsdf_pdf<KINDS> ( sf, wi, wo ), ( float ) 1e-4 );
    V3 f = bsdf_eval<KINDS> ( sf, wi, wo ) * ( 1.f / pdf );
    throughput = had ( throughput, f );
    throughput = throughput * dot ( sf.normal, wi );
    float pr = sel_max ( throughput.x, sel_max ( throughput.y, throughput.z ) );
    if ( d.e3 > pr ) return false;
    throughput = throughput * ( float ) ( 1.0 / ( ( double ) pr + 1e-4 ) );
    ++bounce;
    return bounce <= max_bounces;
}
// Sampler integration (terra_amd_set_sampler_integration, UNPINNED extension): at bounce 0 the pixel sampler's pair replaces the first two variates handed to
// the BSDF's sampler; stream B has been consumed as always
struct SamplerPair { float u0, u1; bool on; };
TD SamplerPair sampler_pair_none() { SamplerPair s; s.u0 = s.u1 = 0.f; s.on = false; return s; }
TD void path_apply_sampler ( PathDraws& d, const SamplerPair& sp, uint32_t bounce ) {
    if ( sp.on && bounce == 0 ) { d.e0 = sp.u0; d.e1 = sp.u1; d.az = azimuth_none(); }
}
// element n of the pixel's sampler (n = camera samples the pixel has received before this one), as the oracle's orc_render_pixels takes it: Halton = the
// radical-inverse pair of n (src/Terra.c:734-755); stratified = the sampler of src/Terra.c:542 at element n mod (strata^2 * 16), its two offsets the next draws of
// the pixel's camera stream (src/Terra.c:714-723)
TD SamplerPair sampler_pair_draw ( uint32_t mode, uint32_t strata, uint64_t n, Pcg32& stream_a ) {
    SamplerPair s = sampler_pair_none();
    if ( mode == 1 ) { s.u0 = radical_inverse ( 3, n ); s.u1 = radical_inverse ( 2, n ); s.on = true; }
    else if ( mode == 2 && strata > 0 ) {
        const uint64_t cap = ( uint64_t ) strata * strata * 16ull, m = n % cap, stratum = m / 16ull;
        const float stratum_size = 1.f / ( float ) strata;
        s.u0 = sd_below_one ( ( ( float ) ( uint32_t ) ( stratum % strata ) + trng_a_float ( stream_a ) ) * stratum_size );
        s.u1 = sd_below_one ( ( ( float ) ( uint32_t ) ( stratum / strata ) + trng_a_float ( stream_a ) ) * stratum_size );
        s.on = true;
    }
    return s;
}
template <int COUNT, int KINDS>
TD bool path_continue ( const DevScene& sc, Surface& sf, V3 wo, V3& throughput, uint32_t& bounce, uint32_t max_bounces, Pcg32& rb, Counters& c, V3& wi, const SamplerPair& sp = sampler_pair_none() ) {
    PathDraws d = path_draw<COUNT> ( sc.sincos24, rb, c );
    if ( KINDS & TERRA_KIND_SAMPLER ) path_apply_sampler ( d, sp, bounce );
    return path_continue<KINDS> ( sf, wo, throughput, bounce, max_bounces, d, wi );
}

// -----------------------------------------------------------------------------
// one full path (the reference's terra_trace), used by the unit entry point and,
// restructured with path regeneration, by the render kernel
// -----------------------------------------------------------------------------
template <int INTEGRATOR, int COUNT, int MODE, int KINDS>
TD V3 trace_path ( const Tracer& T, Ray ray, uint32_t bounces, Pcg32& rb, Counters& c ) {
    V3 Lo = v3 ( 0, 0, 0 ), throughput = v3 ( 1, 1, 1 );
    for ( uint32_t bounce = 0; bounce <= bounces; ++bounce ) {
        Surface sf;
        RaycastResult h = scene_raycast<COUNT, MODE, KINDS> ( T, ray, sf, c );
        if ( !h.hit ) {
            if ( ( KINDS & TERRA_KIND_ENV ) && T.sc.env_mode && !env_reaches_by_samples<INTEGRATOR, KINDS> ( T.sc, bounce ) ) { throughput = had ( throughput, environment_eval ( T.sc, ray.d ) ); Lo = Lo + throughput; }
            break;
        }
        V3 wo = neg ( ray.d );
        Lo = Lo + integrate<INTEGRATOR, COUNT, MODE, KINDS> ( T, ray, sf, h.point, wo, throughput, bounce, rb, c );
        V3 wi;
        uint32_t next_bounce = bounce;
        if ( !path_continue<COUNT, KINDS> ( T.sc, sf, wo, throughput, next_bounce, bounces, rb, c, wi ) ) break;
        ray = surface_ray ( sf, h.point, wi, 1.f );
    }
    return Lo;
}

// -----------------------------------------------------------------------------
// tonemap
// -----------------------------------------------------------------------------
TD V3 uncharted2 ( V3 x ) {
    const float A = 0.15f, B = 0.5f, C = 0.1f, D = 0.2f, E = 0.02f, F = 0.3f;
    V3 r;
    r.x = ( ( x.x * ( A * x.x + C * B ) + D * E ) / ( x.x * ( A * x.x + B ) + D * F ) ) - E / F;
    r.y = ( ( x.y * ( A * x.y + C * B ) + D * E ) / ( x.y * ( A * x.y + B ) + D * F ) ) - E / F;
    r.z = ( ( x.z * ( A * x.z + C * B ) + D * E ) / ( x.z * ( A * x.z + B ) + D * F ) ) - E / F;
    return r;
}
TD V3 powv ( V3 c, float e ) { return v3 ( tdm_powf ( c.x, e ), tdm_powf ( c.y, e ), tdm_powf ( c.z, e ) ); }
TD V3 tonemap ( V3 c, int op, float gamma ) {
    switch ( op ) {
        case 1: c = powv ( c, 1.f / gamma ); break;
        case 2:
            c.x = c.x / ( 1.f + c.x ); c.y = c.y / ( 1.f + c.y ); c.z = c.z / ( 1.f + c.z );
            c = powv ( c, 1.f / gamma ); break;
        case 3: {
            V3 x = v3 ( sel_max ( 0.f, c.x - 0.004f ), sel_max ( 0.f, c.y - 0.004f ), sel_max ( 0.f, c.z - 0.004f ) );
            c.x = ( x.x * ( 6.2f * x.x + 0.5f ) ) / ( x.x * ( 6.2f * x.x + 1.7f ) + 0.06f );
            c.y = ( x.y * ( 6.2f * x.y + 0.5f ) ) / ( x.y * ( 6.2f * x.y + 1.7f ) + 0.06f );
            c.x = ( x.z * ( 6.2f * x.z + 0.5f ) ) / ( x.z * ( 6.2f * x.z + 1.7f ) + 0.06f );   // the reference stores the .z curve in .x
            break;
        }
        case 4: {
            V3 ws = uncharted2 ( v3 ( 11.2f, 11.2f, 11.2f ) );
            ws = v3 ( 1.f / ws.x, 1.f / ws.y, 1.f / ws.z );
            V3 t = uncharted2 ( c * 2.f );
            c = powv ( had ( t, ws ), 1.f / gamma );
            break;
        }
        default: break;
    }
    return c;
}
