// render_kernels.hip -- the terra_render tile loop as a gfx950 kernel.
//
// Mapping (DESIGN.md "Kernel: terra_render_kernel"):
//   * one LANE owns one pixel and walks its samples in order, so the per-pixel
//     random streams are consumed in the reference's order and the radiance sum
//     is accumulated in the reference's order (reference src/Terra.c:551-567);
//   * one WAVE owns an 8x8 pixel packet, one 256-thread block a 16x16 region;
//   * blocks are numbered tile-major (tile_size x tile_size tiles, row-major in the
//     rectangle) so that the tile-sharded multi-GPU form is the same kernel with
//     (rank, world) set (the reference shards tiles over threads the same way,
//     satellite/src/Renderer.cpp:316-350);
//   * a lane whose path ends starts its pixel's next sample in the same loop
//     iteration ("path regeneration"), so the traversal loop always runs with
//     every lane that still has samples;
//   * the traversal stack lives in LDS, one column per lane (conflict free),
//     sized from the tree (max_stack entries, computed at commit);
//   * with a sample split (DevRenderParams::split) a pixel's samples are cut into chunks that run on
//     separate lanes (consecutive blocks), and terra_resolve_kernel folds the chunk sums into the pixel
//     in order -- the framebuffer of `split` successive calls;
//   * scenes that do not fit in LDS run the decoupled loop: a lane is either traversing its ray or
//     waiting to be shaded, and the resumable traversal hands finished lanes back early.
#include <hip/hip_runtime.h>
#include "trace_device.h"
#include "kernels.h"

// One source, several translation units (terra_amd/build.py compiles this file once per TERRA_TU value, in parallel: a single unit takes 3 minutes):
//   TERRA_TU 0  everything that is not a render-kernel instance (resolve / tile kernels, LDS planning, the launch dispatch) + the kernels of template MODE 0
//   TERRA_TU 1 / 2 / 3  the kernels of template MODE 1 (LDS-resident scenes) / 2 (fast tree) / 3 (fast tree + reachability replay) and their launcher
//   undefined   all of it in one unit (tools/kernel_resources.sh, tools/isa_cost.py)
#ifndef TERRA_JOB_STREAM_TABLE      // the jobs' random streams keyed by a kernel of their own ahead of the render (DevRenderParams::job_streams): 0 = by the lane that takes the job
#define TERRA_JOB_STREAM_TABLE 1
#endif
#ifdef TERRA_TU
#define TERRA_TU_HAS(k) ( TERRA_TU == ( k ) )
#else
#define TERRA_TU_HAS(k) 1
#endif

struct DevResult { float acc[3]; int samples; };

TD void wave_flush_counters ( const Counters& c, unsigned long long* g ) {
    const uint32_t v[6] = { c.rays, c.nodes, c.tri_tests, c.hits, c.rand_calls, c.attr_fetches };
    const int slot[6] = { kCtrRays, kCtrNodes, kCtrTriTests, kCtrHits, kCtrRandCalls, kCtrAttrFetches };
    #pragma unroll
    for ( int k = 0; k < 6; ++k ) {
        unsigned long long x = v[k];
        #pragma unroll
        for ( int off = 32; off > 0; off >>= 1 ) x += __shfl_xor ( x, off, 64 );
        if ( ( threadIdx.x & 63 ) == 0 && x ) atomicAdd ( &g[slot[k]], x );
    }
}

// Builds the block's Tracer: carves the dynamic LDS, stages the scene prefix the host
// planned (DevRenderParams.lds_*), and leaves every thread with its own stack / leaf
// list column. Called by all 256 threads (it contains the block barrier).
// per-thread words parked in LDS between uses (indexed [word][thread] like the stack): the radiance sum of the lane's
// current job (touched once per path), the job's number and the lane's draw count when the job started (read at its end), and a row
// that holds each wave's pool of claimed jobs (render_kernels.hip "jobs")
#define TERRA_AUX_WORDS 6        // rows every launch has (acc x 3, job, draw count, wave pools)
#define TERRA_AUX_WORDS_LDS 9    // ... plus, on LDS-resident scenes (lds_mode 1), the path radiance Lo x 3 (TERRA_LO_IN_LDS)
#ifndef TERRA_LO_IN_LDS          // the coupled loop of LDS-resident scenes keeps the current path's radiance (three words that change on few hits) in rows 6-8 instead of registers
#define TERRA_LO_IN_LDS 1
#endif
#ifndef TERRA_CHECK_SHRINK
#define TERRA_CHECK_SHRINK 0
#endif
template <int MODE>
TD Tracer make_tracer ( const DevScene& sc, float4* lds, uint32_t stack_depth, uint32_t leaf_cap, uint32_t lds_nodes, uint32_t lds_tris, bool cull, bool fused ) {
    const int tid = threadIdx.x;
    Tracer T;
    T.sc = sc;
    // [staged nodes][staged triangles][staged properties][stack][leaf list][parked words]  (sizes: terra_lds_bytes)
    float4* ln = lds;                                                // byte offset 0: a staged node's address is its stack word
    float4* lt = ln + ( TERRA_LDS_NODE_BYTES / 16 ) * lds_nodes;      // (fast-tree launches stage nothing: lds_nodes == lds_tris == 0)
    float4* lp = lt + 3 * lds_tris;
    // MODE 1 also stages what shading reads per hit: the materials, the light list and the per-triangle areas (scene_extra_lds_bytes: three 16-byte aligned sections)
    uint32_t* lm = reinterpret_cast<uint32_t*> ( lp + 4 * lds_tris );
    const uint32_t m_words = MODE == 1 ? ( ( sc.n_objects * ( uint32_t ) ( sizeof ( DevMaterial ) / 4 ) + 3u ) & ~3u ) : 0u, l_words = MODE == 1 ? sc.n_lights * 4u : 0u, a_words = MODE == 1 ? ( ( sc.n_tris + 3u ) & ~3u ) : 0u;
    uint32_t* ll = lm + m_words; uint32_t* la = ll + l_words;
    int* words = reinterpret_cast<int*> ( la + a_words );
    T.stack = words + tid;
    T.leaves = words + stack_depth * TERRA_COL + tid;
    T.leaf_cap = ( int ) leaf_cap;
    T.stack_cap = ( int ) stack_depth - TERRA_CHECK_SHRINK;      // TERRA_CHECK_SHRINK > 0: positive control of the bounds check
    T.stack_lim = ( uint32_t ) ( uintptr_t ) words + stack_depth * 1024u; T.spill = nullptr; T.spill_cap = 0;      // (fast-tree launches: the kernel sets the spill area)
    T.faults = nullptr;
    T.cull = cull; T.fused = cull && fused;
    const float4* gn = reinterpret_cast<const float4*> ( sc.nodes );
    const float4* gt = reinterpret_cast<const float4*> ( sc.tris );
    const float4* gp = reinterpret_cast<const float4*> ( sc.props );
    for ( uint32_t i = tid; i < lds_nodes; i += TERRA_COL ) {       // node i -> the axis-major, both-signs layout (trace_device.h "Staged node")
        const float4 q0 = gn[4 * i], q1 = gn[4 * i + 1], q2 = gn[4 * i + 2], q3 = gn[4 * i + 3];
        const float mn0[3] = { q0.x, q0.y, q0.z }, mx0[3] = { q0.w, q1.x, q1.y }, mn1[3] = { q1.z, q1.w, q2.x }, mx1[3] = { q2.y, q2.z, q2.w };
        float4* o = ln + ( TERRA_LDS_NODE_BYTES / 16 ) * i;
        #pragma unroll
        for ( int a = 0; a < 3; ++a ) { o[2 * a] = make_float4 ( mn0[a], mx0[a], mn1[a], mx1[a] ); o[2 * a + 1] = make_float4 ( mx0[a], mn0[a], mx1[a], mn1[a] ); }
        uint32_t c0 = __float_as_uint ( q3.x ), c1 = __float_as_uint ( q3.y );
        if ( ! ( c0 & DEV_CHILD_LEAF ) ) c0 *= TERRA_LDS_NODE_BYTES;       // inner child: byte offset of its staged node
        if ( ! ( c1 & DEV_CHILD_LEAF ) ) c1 *= TERRA_LDS_NODE_BYTES;
        o[6] = make_float4 ( __uint_as_float ( c0 ), __uint_as_float ( c1 ), 0.f, 0.f );
    }
    for ( uint32_t i = tid; i < 3 * lds_tris; i += TERRA_COL ) lt[i] = gt[i];
    for ( uint32_t i = tid; i < 4 * lds_tris; i += TERRA_COL ) lp[i] = gp[i];
    if ( MODE == 1 ) {
        const uint32_t* gm = reinterpret_cast<const uint32_t*> ( sc.mats ); const uint32_t* gl = reinterpret_cast<const uint32_t*> ( sc.lights ); const uint32_t* ga = reinterpret_cast<const uint32_t*> ( sc.tri_area );
        for ( uint32_t i = tid; i < sc.n_objects * ( uint32_t ) ( sizeof ( DevMaterial ) / 4 ); i += TERRA_COL ) lm[i] = gm[i];
        for ( uint32_t i = tid; i < sc.n_lights * 4u; i += TERRA_COL ) ll[i] = gl[i];
        for ( uint32_t i = tid; i < sc.n_tris; i += TERRA_COL ) la[i] = ga[i];
    }
    T.l_mats = MODE == 1 ? reinterpret_cast<const DevMaterial*> ( lm ) : sc.mats; T.l_lights = MODE == 1 ? reinterpret_cast<const DevLight*> ( ll ) : sc.lights; T.l_area = MODE == 1 ? reinterpret_cast<const float*> ( la ) : sc.tri_area;
    T.l_nodes = ln; T.l_tris = reinterpret_cast<const float*> ( lt ); T.l_props = lp;
    T.lds_nodes = lds_nodes; T.lds_tris = lds_tris;
    __syncthreads();
    return T;
}

// block (in launch order: own tile k, 16x16 block b inside it) and thread -> pixel; a wave covers 8x8 pixels
TD bool block_pixel ( const DevRenderParams& p, uint32_t blk, uint32_t tid, uint32_t& px, uint32_t& py ) {
    const uint32_t bpt = p.tile_size >> 4, bpt2 = bpt * bpt;
    const uint32_t k = blk / bpt2, b = blk - k * bpt2;
    const uint32_t tiles_x = ( p.w + p.tile_size - 1 ) / p.tile_size;
    const uint32_t t = p.rank + k * p.world;
    const uint32_t tx = t % tiles_x, ty = t / tiles_x;
    const uint32_t bx = b % bpt, by = b / bpt;
    const uint32_t wave = tid >> 6, lane = tid & 63;
    const uint32_t lx = tx * p.tile_size + bx * 16 + ( wave & 1 ) * 8 + ( lane & 7 );
    const uint32_t ly = ty * p.tile_size + by * 16 + ( wave >> 1 ) * 8 + ( lane >> 3 );
    px = p.x + lx; py = p.y + ly;
    return lx < p.w && ly < p.h;
}

#if TERRA_TU_HAS ( 0 )
// Second kernel of every render: pixel += chunk sums, in chunk order (float adds in the order `split` successive calls
// would make them; split == 1: the one sum of the call, reference src/Terra.c:570-572), then exposure / tonemap / store
// (src/Terra.c:574-630). gridDim.x = the job space's 16x16 pixel blocks.
__global__ __launch_bounds__ ( 256 ) void terra_resolve_kernel ( DevRenderParams p ) {
    if ( p.job_queue && blockIdx.x == 0 && threadIdx.x == 0 ) *p.job_queue = 0u;       // the render kernel is done with its queue: leave the word ready for the next launch on this scratch (scene_host.cpp launch_render)
    uint32_t px, py;
    if ( !block_pixel ( p, blockIdx.x, threadIdx.x, px, py ) ) return;
    const uint32_t vb = p.block_order ? p.block_order[gridDim.x + blockIdx.x] : blockIdx.x;      // where the job order put this pixel block: its sums are stored under that number
    const size_t pix = ( size_t ) ( py - p.st_y ) * p.st_pitch + ( px - p.st_x );
    DevResult* results = reinterpret_cast<DevResult*> ( p.results );
    DevResult out = results[pix];
    uint32_t calls = 0;
    for ( uint32_t j = 0; j < p.split; ++j ) {
        const float4 q = p.partials[ ( ( size_t ) j * gridDim.x + vb ) * 256 + threadIdx.x];
        out.acc[0] = out.acc[0] + q.x; out.acc[1] = out.acc[1] + q.y; out.acc[2] = out.acc[2] + q.z;
        calls += __float_as_uint ( q.w );
    }
    out.samples = out.samples + ( int ) p.spp;
    results[pix] = out;
    float n = ( float ) out.samples;
    V3 color = v3 ( out.acc[0] / n, out.acc[1] / n, out.acc[2] / n ) * p.exposure;
    color = tonemap ( color, p.tonemap, p.gamma );
    p.pixels[3 * pix + 0] = color.x; p.pixels[3 * pix + 1] = color.y; p.pixels[3 * pix + 2] = color.z;
    if ( p.rand_calls ) p.rand_calls[pix] = calls;
}

#endif

// Occupancy target per integrator (second __launch_bounds__ argument = waves per SIMD = 256-thread
// blocks per CU). Measured on MI355X (profiles/): the Simple/debug kernels run fastest at 5 even
// with a small spill; Direct/MIS carry a second surface and more live state.
#ifndef TERRA_WAVES_SIMPLE
#define TERRA_WAVES_SIMPLE 5
#endif
#ifndef TERRA_WAVES_LIGHT
#define TERRA_WAVES_LIGHT 4
#endif
#ifndef TERRA_WAVES_MIS          // Direct + MIS on LDS-resident diffuse scenes, split at its two rays (TERRA_COUPLED_MIS_SPLIT): 5 -> 158.1 ms with 116 B of scratch, 4 -> 165.4 with none
#define TERRA_WAVES_MIS 5        // (Cornell 512 spp; profiles/r04_measurements/ab_light_split.log; unsplit, round 3: 5 -> 191.9, 4 -> 178.9)
#endif
#ifndef TERRA_WAVES_DIRECT_PHONG // ... also the diffuse + Phong variant (A/B)
#define TERRA_WAVES_DIRECT_PHONG 0
#endif
#ifndef TERRA_WAVES_DIRECT       // Direct on LDS-resident scenes, without work counters: 5 -> 117.7 ms, 4 -> 120.4 (Direct + MIS: 5 -> 191.9, 4 -> 178.9; profiles/r03_measurements/ab_waves_light.log)
#define TERRA_WAVES_DIRECT 5
#endif
#ifndef TERRA_WAVES_GENERIC      // Simple/debug kernels compiled for every preset, textures and the environment term (KINDS != diffuse-only)
#define TERRA_WAVES_GENERIC TERRA_WAVES_SIMPLE
#endif
// Decoupled traversal (see the kernel): scenes that are not LDS-resident, integrators whose shading casts no rays of its own
#ifndef TERRA_DECOUPLED_ENABLE
#define TERRA_DECOUPLED_ENABLE 1
#endif
#ifndef TERRA_DECOUPLED_EXIT_SHIFT      // leave the traversal when n >> shift of the n lanes that entered it have finished
#define TERRA_DECOUPLED_EXIT_SHIFT 4
#endif
#ifndef TERRA_DECOUPLED_FAST     // ... and the fast-tree kernels (MODE 2)
#define TERRA_DECOUPLED_FAST 1
#endif
#ifndef TERRA_FAST_EXIT_16THS    // MODE 2 leaves the traversal when this many 16ths of the lanes that entered it have finished (a ray is cheap there, so shading wants fuller waves)
#define TERRA_FAST_EXIT_16THS 12
#endif
// (LDS-resident scenes, MODE 1, keep the coupled loop: decoupled they measure 20 % slower -- shading outweighs traversal there; CHANGELOG.md round 1 / 2)
#define TERRA_DECOUPLED(I, M) ( TERRA_DECOUPLED_ENABLE && ( ( M ) == 0 || ( TERRA_DECOUPLED_FAST && ( M ) >= 2 ) ) && ( ( I ) == 0 || ( I ) == 3 || ( I ) == 4 || ( I ) == 5 ) )
// ... and Direct, whose one shadow ray per hit becomes a traversal job of its own (scenes without textured attributes)
#ifndef TERRA_DECOUPLED_DIRECT_ENABLE
#define TERRA_DECOUPLED_DIRECT_ENABLE 1
#endif
#ifndef TERRA_DECOUPLED_FAST_DIRECT  // ... also on the fast tree (MODE 2): hall, 32 spp, 112.0 -> 108.2 ms
#define TERRA_DECOUPLED_FAST_DIRECT 1
#endif
#ifndef TERRA_DECOUPLED_FAST_MIS     // Direct + MIS on the fast tree is faster coupled (hall, 32 spp: 168.6 ms against 187.7 decoupled, profiles/r02_measurements/ab_fast_light.log; round 4, with the
                                     // light-sample rays' shortcut in both: 100.5 against 117.5, sphere scene 128 spp 157.8 against 162.4, profiles/r04_measurements/ab_fast_tree_knobs.log)
#define TERRA_DECOUPLED_FAST_MIS 0
#endif
#define TERRA_DECOUPLED_DIRECT(I, M, K) ( TERRA_DECOUPLED_DIRECT_ENABLE && ( ( M ) == 0 || ( TERRA_DECOUPLED_FAST_DIRECT && ( M ) >= 2 ) ) && ( I ) == 1 && ( ( K ) & TERRA_KIND_TEX ) == 0 )
#ifndef TERRA_DECOUPLED_MIS_ENABLE
#define TERRA_DECOUPLED_MIS_ENABLE 1
#endif
#define TERRA_DECOUPLED_MIS(I, M, K) ( TERRA_DECOUPLED_MIS_ENABLE && ( ( M ) == 0 || ( TERRA_DECOUPLED_FAST_MIS && ( M ) >= 2 ) ) && ( I ) == 2 && ( ( K ) & TERRA_KIND_TEX ) == 0 )
#ifndef TERRA_WAVES_DECOUPLED
#define TERRA_WAVES_DECOUPLED TERRA_WAVES_SIMPLE
#endif
#ifndef TERRA_WAVES_FAST_TREE      // fast-tree (MODE 2) kernels are latency bound: more resident waves pay for the extra scratch
#define TERRA_WAVES_FAST_TREE 5     // (round 1, coupled loop: 4 -> 81.2 ms, 5 -> 71.0, 6 -> 67.2; with the decoupled loop, hall 64 spp: 5 -> 82.0 ms, 6 -> 83.6, 7 -> 89.0)
#endif
#ifndef TERRA_WAVES_FAST_TREE_LIGHT  // (round 2, hall, 16 spp, Direct / MIS: 4 -> 85.0 / 135.3 ms, 5 -> 79.3 / 122.2 ms, 6 -> 74.7 / 114.5 ms; ab_fl*.log. Round 3, with the job queue and
#define TERRA_WAVES_FAST_TREE_LIGHT 5 // without work counters, hall 64 spp: 5 -> 140.3 / 246.6 ms, 6 -> 163.5 / 267.5 ms; profiles/r03_measurements/ab_waves_light.log)
#endif
#ifndef TERRA_WAVES_GLOBAL_LIGHT     // reference tree read from global memory (MODE 0), Direct/MIS (sphere scene, Direct: 4 -> 864 ms, 5 -> 815 ms, 6 -> 841 ms; ab_gl*.log)
#define TERRA_WAVES_GLOBAL_LIGHT 5
#endif
#ifndef TERRA_WAVES_FAST_TREE_GENERIC        // ... but the generic-kinds kernels (Phong/GGX/glass/textures compiled in) already spill at 5:
#define TERRA_WAVES_FAST_TREE_GENERIC 5      // sphere scene, Simple, 64 spp: 5 -> 63.5 ms, 6 -> 70.1 ms, 7 -> 71.5 ms, 8 -> 81 ms (ab_ww2.log, ab_f78.log)
#endif
#ifndef TERRA_WAVES_FAST_TREE_GENERIC_LIGHT
#define TERRA_WAVES_FAST_TREE_GENERIC_LIGHT 4
#endif
#define TERRA_IS_LIGHT(I) ( ( I ) == 1 || ( I ) == 2 || ( I ) == 6 )
#define TERRA_WAVES_FOR(I, K, M) ( ( M ) >= 2 ? ( ( K ) == 1 ? ( TERRA_IS_LIGHT ( I ) ? TERRA_WAVES_FAST_TREE_LIGHT : TERRA_WAVES_FAST_TREE ) : ( TERRA_IS_LIGHT ( I ) ? TERRA_WAVES_FAST_TREE_GENERIC_LIGHT : TERRA_WAVES_FAST_TREE_GENERIC ) ) \
                                 : TERRA_IS_LIGHT ( I ) ? ( ( M ) == 0 ? TERRA_WAVES_GLOBAL_LIGHT : ( ( I ) == 1 && ( ( K ) == 1 || ( TERRA_WAVES_DIRECT_PHONG && ( K ) == 3 ) ) ) ? TERRA_WAVES_DIRECT : ( ( I ) == 2 && ( K ) == 1 ) ? TERRA_WAVES_MIS : TERRA_WAVES_LIGHT ) \
                                 : TERRA_DECOUPLED ( I, M ) ? TERRA_WAVES_DECOUPLED : ( ( K ) == 1 ? TERRA_WAVES_SIMPLE : TERRA_WAVES_GENERIC ) )
// ---- pieces shared by the decoupled loops of the kernel below -----------------------------------------
// Per-lane traversal state that survives leaving the resumable traversal (the stack column and the leaf list are in LDS).
struct LaneTraversal { RayState st; SlabSel sel; Closest best; uint32_t rank, hand, held; int top; bool traversing, regular, anyhit; };     // anyhit: a light-sample ray that knows its triangle (trace_device.h fast_expect)     // hand: MODE >= 2, what the lane holds between calls (trace_device.h traverse_fast_resume)     // regular: MODE 0 / 1 the ray's slab variant; MODE 2 (which has no use for that) "second, checked pass" of the reachability mode     // top: entries on the lane's stack (its leaf list is always empty between calls)
TD LaneTraversal lane_traversal_idle ( const Tracer& T, const Ray& any_ray ) {
    LaneTraversal t; t.st = ray_state_init ( any_ray ); t.sel = slab_sel ( any_ray ); t.best.depth = FLT_MAX; t.best.tri = 0xffffffffu; t.rank = 0xffffffffu; t.hand = DEV_CHILD_EMPTY; t.held = 0u; t.top = 0; t.anyhit = false; t.traversing = false; t.regular = true;
    return t;
}
// puts `ray` in flight: the origin offset terra_scene_raycast applies (src/Terra.c:1629-1630), ray state, empty closest hit, root on the stack
template <int COUNT, int MODE>
TD void lane_traversal_start ( const Tracer& T, const Ray& ray, LaneTraversal& t, Counters& c ) {
    Ray r = ray; r.o = r.o + r.d * 0.001f;
    t.st = ray_state_init ( r );
    t.sel = slab_sel ( r );
    t.regular = ray_is_regular ( r );
    t.best.depth = FLT_MAX; t.best.tri = 0xffffffffu; t.rank = 0xffffffffu;
    if ( MODE >= 2 ) { t.hand = TERRA_FAST_ROOT_IN_HAND; t.held = 0u; t.top = 0; t.anyhit = false; }      // the fast tree's traversal starts with the root in hand
    else { *T.stack = 0; t.top = 1; }
    t.traversing = true;
    if ( COUNT ) ++c.rays;
}
// the ray just started is a light-sample ray towards triangle `expected` (soup index): trace_device.h fast_expect. (A ray that misses its triangle stays an ordinary ray.)
template <int COUNT, int MODE>
TD void lane_traversal_expect ( const Tracer& T, const Ray& ray, LaneTraversal& t, uint32_t expected ) {
    if constexpr ( TERRA_SHADOW_ANYHIT && MODE == 2 && COUNT == 0 ) {
        Ray r = ray; r.o = r.o + r.d * 0.001f;
        const V3 o_perm = v3 ( pick ( r.o, t.st.ix ), pick ( r.o, t.st.iy ), pick ( r.o, t.st.iz ) );
        ClosestRanked b2; b2.depth = t.best.depth; b2.rank = t.rank; b2.tri = t.best.tri;
        t.anyhit = fast_expect ( T, t.st, o_perm, expected, b2 );
        t.best.depth = b2.depth; t.best.tri = b2.tri; t.rank = b2.rank;
        if ( !t.anyhit ) { t.hand = DEV_CHILD_EMPTY; t.traversing = false; }      // the ray misses its triangle: whatever it hits instead, the answer is "not the expected one" (best.tri: none)
    }
}
// DevScene::reach (scenes outside the coordinate range of the containment proof, MODE 2): a returned closest hit stands if the reference traversal would have
// reached it; if not -- very rare -- the same ray goes back in flight with every candidate checked (trace_device.h bvh_traverse_fast). True = the lane is traversing again.
template <int MODE>
TD bool lane_traversal_recheck ( const Tracer& T, const Ray& ray, LaneTraversal& t ) {
    if ( MODE != 3 || !T.sc.reach || !t.regular || t.best.tri == 0xffffffffu ) return false;      // (MODE 2: regular == false marks the checked pass)
    Ray r = ray; r.o = r.o + r.d * 0.001f;
    if ( reference_reaches ( T, t.best.tri, r ) ) return false;
    t.best.depth = FLT_MAX; t.best.tri = 0xffffffffu; t.rank = 0xffffffffu;
    t.hand = TERRA_FAST_ROOT_IN_HAND; t.held = 0u; t.top = 0;
    t.traversing = true; t.regular = false;
    return true;
}
// advances every traversing lane until 1 / 2^TERRA_DECOUPLED_EXIT_SHIFT of them have finished; false when no lane is traversing
template <int COUNT, int MODE>
TD bool lane_traversal_run ( const Tracer& T, const Ray& ray, LaneTraversal& t, Counters& c ) {
    const int n_trav = __popcll ( __ballot ( t.traversing ) );
    if ( n_trav == 0 ) return false;
    int quota = MODE >= 2 ? ( n_trav * TERRA_FAST_EXIT_16THS ) >> 4 : n_trav >> TERRA_DECOUPLED_EXIT_SHIFT; if ( quota < 1 ) quota = 1;
    const int exit_active = n_trav - quota;
    Ray r = ray; r.o = r.o + r.d * 0.001f;
    V3 o_perm = v3 ( pick ( r.o, t.st.ix ), pick ( r.o, t.st.iy ), pick ( r.o, t.st.iz ) );
    int* sp = T.stack + t.top * TERRA_COL;
    if constexpr ( MODE >= 2 ) {
        ClosestRanked b2; b2.depth = t.best.depth; b2.rank = t.rank; b2.tri = t.best.tri;
        traverse_fast_resume<COUNT> ( T, r, t.st, o_perm, b2, sp, t.hand, t.held, exit_active, c, MODE == 3 && T.sc.reach && !t.regular, TERRA_SHADOW_ANYHIT && MODE == 2 && COUNT == 0 && t.anyhit );      // (a ray with a zero direction component starts in the checked pass: harmless)
        t.traversing = fast_traversing ( T, t.hand, t.held, sp );
        t.best.depth = b2.depth; t.best.tri = b2.tri; t.rank = b2.rank;
    } else {
        if ( __all ( !t.traversing || t.regular ) ) traverse_resume<COUNT, MODE, true> ( T, r, t.sel, t.st, o_perm, t.best, sp, t.traversing, exit_active, c );
        else traverse_resume<COUNT, MODE, false> ( T, r, t.sel, t.st, o_perm, t.best, sp, t.traversing, exit_active, c );
    }
    t.top = ( int ) ( sp - T.stack ) / TERRA_COL;
    return true;
}

// the closest hit as the index the light tables use (the soup): the fast tree's triangles are in leaf order and carry (object, triangle in object)
template <int MODE>
TD uint32_t hit_soup_index ( const Tracer& T, uint32_t tri ) {
    if ( MODE < 2 || tri == 0xffffffffu ) return tri;
    const float4* ft = reinterpret_cast<const float4*> ( T.sc.fast_tris );
    return T.sc.mats[__float_as_uint ( ft[3 * tri].w )].first_tri + __float_as_uint ( ft[3 * tri + 1].w );
}
// a returned path segment's hit: the surface the reference's terra_scene_raycast hands to terra_trace (src/Terra.c:1640-1655)
template <int COUNT, int MODE, int KINDS>
TD V3 shade_surface ( const Tracer& T, const Ray& ray, const Closest& best, Surface& sf, Counters& c ) {
    Ray r = ray; r.o = r.o + r.d * 0.001f;           // the offset scene_raycast applies (src/Terra.c:1629-1630)
    V3 point = r.o + r.d * best.depth;
    uint32_t object, tri_in_object, nattr;
    surface_init<MODE, KINDS> ( T, best.tri, point, sf, object, tri_in_object, nattr );
    if ( COUNT ) ++c.hits;
    if ( COUNT == 2 ) c.attr_fetches += nattr + 1;
    return point;
}
// a path ended: its radiance joins the pixel's sum of this call (parked in LDS, see TERRA_AUX_WORDS)
TD void deposit ( float* acc_lds, V3 Lo ) { acc_lds[0] = acc_lds[0] + Lo.x; acc_lds[256] = acc_lds[256] + Lo.y; acc_lds[512] = acc_lds[512] + Lo.z; }

// ---- jobs ------------------------------------------------------------------------------------------------------------
// A JOB is one (pixel, chunk) pair: chunk_spp camera samples of one pixel, traced in order with the streams keyed (pixel,
// samples already in the pixel + chunk * chunk_spp) and summed from zero into partials[] (DevRenderParams::split; split == 1: the
// call's one sum per pixel). Jobs are numbered like the threads of a plain launch would be -- job = virtual block * 256 + virtual
// thread, virtual block = (16x16 pixel block of the shard) * split + chunk -- and the grid is PERSISTENT: at most as many blocks
// as the GPU holds at once (terra_launch_render). A wave claims the next unclaimed batch of 64 jobs from a queue word in HBM (one atomic per
// 64 jobs) whenever its pool (two words of LDS) is empty; its lanes take jobs from the pool whenever their own is finished. What a job computes does not depend on which lane runs it, so the frame
// is the plain launch's bit for bit; what changes is that a wave's 64 lanes no longer wait for the slowest of 64 fixed pixels
// (paths have random lengths: with 64-sample chunks 13 % of the lane time of the Cornell frame was spent in that ramp-down,
// profiles/r02_measurements/phase_cornell.log) and that the launch has no tail of half-empty rounds.
// Exit: the queue only grows; a lane that finds the pool empty and the queue beyond the job space leaves the loop for good.
// What it buys depends on the loop. The decoupled loops (scenes traversed from global memory), whose traversal hands finished lanes back early, gain 9 % (hall,
// sphere scene). The coupled loops (LDS-resident scenes) gain 1.5 % (Simple) to 3.5 % (Direct) -- far less than the 13 % of lane time that the ramp-down of a plain
// launch leaves idle, because a wave's cost is the sum over its iterations of the LONGEST lane's node and leaf loops, and a wave that ramps down with few live lanes
// runs short loops: with the queue the lanes of the Cornell frame are 97.5 % alive instead of 87 %, yet the frame needs as many node-loop iterations (4.5e8),
// leaf-loop iterations and shading executions as before (profiles/r03_measurements/phase_cornell_queue.log, ab_job_queue.log).
struct Jobs { uint32_t px, py, s; bool exhausted; uint32_t base; };      // base: camera samples the pixel has received before this job (sampler integration only)

#ifndef TERRA_JOB_FETCH_MIN      // (coupled loop) lanes at a job boundary switch jobs together once this many wait there -- or no lane of the wave is tracing:
#define TERRA_JOB_FETCH_MIN 1    // the switch (pixel decode, stream keys: ~250 instructions) then runs for several lanes at once. 1 measured best (Cornell: 1 -> 71.5 ms,
#endif                           // 2 -> 72.8, 4 -> 73.6, 8 -> 75.9 at 4 blocks per CU; 65.8 / 66.2 / 66.7 at 5; profiles/r03_measurements/ab_job_queue.log)
#define TERRA_JOB_BATCH 64u
#define TERRA_JOB_NONE 0xffffffffu

// n / d for the launch-constant divisors of the job decode: q = (n * magic) >> 32 with magic = ceil(2^32 / d), exact while n * d < 2^32 (the host checks);
// magic == 0 stands for d == 1
TD uint32_t magic_div ( uint32_t n, uint32_t magic ) { return magic ? __umulhi ( n, magic ) : n; }
// block_pixel() for a job, with the divisions by launch constants done by multiplication (p.job_div_*; the host refuses launches whose dividends are too large
// for that -- no plain-division fallback here: its hoisted reciprocals would sit in registers through the whole render loop)
TD bool job_pixel_of_block ( const DevRenderParams& p, uint32_t blk, uint32_t tid, uint32_t& px, uint32_t& py );
TD bool job_pixel ( const DevRenderParams& p, uint32_t job, uint32_t& px, uint32_t& py, uint32_t& chunk ) {
    const uint32_t vblock = job >> 8;
    chunk = vblock & ( p.split - 1 );
    return job_pixel_of_block ( p, vblock >> p.split_log2, job & 255u, px, py );
}
// thread tid's pixel of 16x16 pixel block blk of the launch
TD bool job_pixel_of_block ( const DevRenderParams& p, uint32_t blk, uint32_t tid, uint32_t& px, uint32_t& py ) {
    const uint32_t bpt = p.tile_size >> 4, bpt2 = bpt * bpt;
    const uint32_t k = magic_div ( blk, p.job_div_bpt2 ), b = blk - k * bpt2;
    const uint32_t t = p.rank + k * p.world;
    const uint32_t ty = magic_div ( t, p.job_div_tiles_x ), tx = t - ty * p.job_tiles_x;
    const uint32_t by = magic_div ( b, p.job_div_bpt ), bx = b - by * bpt;
    const uint32_t wave = tid >> 6, lane = tid & 63;
    const uint32_t lx = tx * p.tile_size + bx * 16 + ( wave & 1 ) * 8 + ( lane & 7 );
    const uint32_t ly = ty * p.tile_size + by * 16 + ( wave >> 1 ) * 8 + ( lane >> 3 );
    px = p.x + lx; py = p.y + ly;
    return lx < p.w && ly < p.h;
}
// aux: the lane's parked words (TERRA_AUX_WORDS rows of 256): [0], [256], [512] the job's radiance sum; [768] the job; [1024] the lane's draw count at the job's
// start; row 5 holds, per wave, the pool {next job, end} at [1280 + 64 * wave + 0 / 1] relative to thread 0's column
// (the pool is how the lanes of a wave tell each other which jobs are taken -- lane 0 of one group of callers writes it, another group reads it on a later call -- so its
//  words are read and written with relaxed wave-scope atomics: plain ds_read / ds_write instructions that the compiler may neither keep in a register across the render
//  loop's back edge nor merge. `volatile` says the same and costs the Simple kernel 12 bytes of scratch, the Direct one 20 more.)
typedef uint32_t PoolWord;
TD uint32_t pool_load ( const PoolWord* w ) { return __hip_atomic_load ( w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT ); }
TD void pool_store ( PoolWord* w, uint32_t v ) { __hip_atomic_store ( w, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT ); }
TD PoolWord* job_pool_of_wave ( float* aux_of_thread ) { return reinterpret_cast<PoolWord*> ( aux_of_thread - threadIdx.x ) + 1280 + ( threadIdx.x & ~63u ); }
TD void job_init_lane ( const DevRenderParams& p, float* aux ) {
    reinterpret_cast<uint32_t*> ( aux ) [768] = TERRA_JOB_NONE;
    if ( ( threadIdx.x & 63u ) == 0 ) {       // the wave's first pool
        PoolWord* pool = job_pool_of_wave ( aux );
        if ( p.job_queue ) { pool_store ( pool, 0 ); pool_store ( pool + 1, 0 ); }      // empty: the first ask claims a batch from the queue like every later one. (No job is RESERVED for a block by its index:
                                                              // a block that the GPU only makes resident late -- when others have drained the queue -- finds nothing and leaves.)
        else {                                                // plain launch (no queue): the 64 jobs of the wave's own index
            const uint32_t total = p.job_blocks * 256u, first = blockIdx.x * 256u + threadIdx.x;
            pool_store ( pool, first < total ? first : total ); pool_store ( pool + 1, first + TERRA_JOB_BATCH < total ? first + TERRA_JOB_BATCH : total );
        }
    }
}
// Job boundary, for whichever lanes of the wave call it together: store the finished job's sum, take the next job from the wave's pool (claiming a new batch from
// the queue when it runs dry), set up its pixel and streams. Leaves j.s == chunk_spp when the job is a pixel outside the rectangle (the job space is made of whole
// 16x16 blocks): the lane simply asks again. Only the calling lanes' registers change; the pool lives in LDS because the lanes that are not here must see it move too.
// TABLE: the job's streams come keyed from DevRenderParams::job_streams (LDS-resident scenes, whose kernels are bound by instruction issue: Cornell 56.7 -> 55.1 ms; the
// kernels that wait on memory lose a little to the table's 64 bytes of traffic per job -- hall 244.1 -> 245.7 ms -- and key their streams here)
template <int COUNT, bool TABLE>
TD void job_next ( const DevRenderParams& p, float* aux, Jobs& j, PixelStreams& rs, const Counters& c ) {
    uint32_t* auxu = reinterpret_cast<uint32_t*> ( aux );
    const uint32_t fin = auxu[768];
    if ( fin != TERRA_JOB_NONE ) {
        const uint32_t vblock = fin >> 8, chunk = vblock & ( p.split - 1 ), blk = vblock >> p.split_log2;
        p.partials[ ( ( size_t ) chunk * ( p.job_blocks >> p.split_log2 ) + blk ) * 256 + ( fin & 255u )] = make_float4 ( aux[0], aux[256], aux[512], __uint_as_float ( COUNT == 2 ? c.rand_calls - auxu[1024] : 0u ) );
        auxu[768] = TERRA_JOB_NONE;
    }
    PoolWord* pool = job_pool_of_wave ( aux );
    const unsigned long long m = __ballot ( 1 );                 // the lanes here
    const uint32_t lane = threadIdx.x & 63u, n = ( uint32_t ) __popcll ( m ), ahead = ( uint32_t ) __popcll ( m & ( ( 1ull << lane ) - 1ull ) );
    const uint32_t total = p.job_blocks * 256u;
    uint32_t next = pool_load ( pool ), end = pool_load ( pool + 1 );
    const uint32_t take = n < end - next ? n : end - next;
    uint32_t job = next + ahead;
    bool got = ahead < take;
    next += take;
    if ( take < n && next < total && p.job_queue ) {             // the pool ran dry: the next batch of the queue (one atomic by the first lane here), for the lanes still without a job
                                                                 // (next == total: an earlier batch already came back beyond the job space -- nothing is left, no need to ask again;
                                                                 //  no queue: a plain launch, every lane has the one job of its own index)
        uint32_t base = 0;
        if ( ahead == 0 ) base = atomicAdd ( p.job_queue, TERRA_JOB_BATCH );
        base = ( uint32_t ) __builtin_amdgcn_readfirstlane ( ( int ) base );
        next = base < total ? base : total; end = base + TERRA_JOB_BATCH < total ? base + TERRA_JOB_BATCH : total;
        const uint32_t take2 = n - take < end - next ? n - take : end - next;
        if ( !got ) { job = next + ( ahead - take ); got = ahead - take < take2; }
        next += take2;
    }
    if ( ahead == 0 ) { pool_store ( pool, next ); pool_store ( pool + 1, end ); }
    if ( !got ) { j.exhausted = true; return; }                  // the queue is monotone: a batch that does not cover the askers means nothing is left, ever
    uint32_t chunk;
    int prior_samples;
    if constexpr ( TERRA_JOB_STREAM_TABLE && TABLE ) {
        const uint4 e1 = p.job_streams[2 * ( size_t ) job + 1];       // keyed by terra_job_streams_kernel (below): the job's pixel comes with it, the decode's divisions are not repeated here
        const uint4 e0 = p.job_streams[2 * ( size_t ) job];
        // both halves of the entry are asked for BEFORE the first is looked at: the compiler otherwise fetches e1.z, waits, branches, and only then fetches the rest -- two
        // memory round trips per job switch with the whole wave parked at the wait (Cornell 512 spp, split 32: TERRA_JOB_ENTRY_ONE_TRIP 0 / 1, profiles/r04_measurements/ab_job_entry.log)
        asm volatile ( "" : : "v" ( e0.x ), "v" ( e0.y ), "v" ( e0.z ), "v" ( e0.w ), "v" ( e1.x ), "v" ( e1.y ), "v" ( e1.z ), "v" ( e1.w ) );
        if ( e1.z == 0xffffffffu ) return;                                 // (a pixel outside the rectangle)
        j.px = e1.z & 0xffffu; j.py = e1.z >> 16;
        chunk = ( job >> 8 ) & ( p.split - 1 );
        rs.a.state = ( uint64_t ) e0.x | ( ( uint64_t ) e0.y << 32 ); rs.a.inc = 1;
        rs.b.state = ( uint64_t ) e0.z | ( ( uint64_t ) e0.w << 32 ); rs.b.inc = ( uint64_t ) e1.x | ( ( uint64_t ) e1.y << 32 );
        rs.seedA = 0;                                                      // (only trng_pixel_streams itself uses it)
        prior_samples = ( int ) e1.w;
    } else {
        if ( !job_pixel ( p, job, j.px, j.py, chunk ) ) return;
        prior_samples = reinterpret_cast<const DevResult*> ( p.results ) [ ( size_t ) ( j.py - p.st_y ) * p.st_pitch + ( j.px - p.st_x )].samples;      // keys the streams; the sum itself is the resolve kernel's business
        rs = trng_pixel_streams ( p.frame_seed, ( uint64_t ) j.py * p.fb_w + j.px, ( uint64_t ) ( uint32_t ) prior_samples + ( uint64_t ) chunk * p.chunk_spp );
    }
    aux[0] = 0.f; aux[256] = 0.f; aux[512] = 0.f; auxu[768] = job;
    if ( COUNT == 2 ) auxu[1024] = c.rand_calls;
    j.s = 0; j.base = ( uint32_t ) prior_samples + chunk * p.chunk_spp;
}

template <int INTEGRATOR, int COUNT, int MODE, int KINDS>
__global__ __launch_bounds__ ( 256, TERRA_WAVES_FOR ( INTEGRATOR, KINDS, MODE ) ) void terra_render_kernel ( DevRenderParams p ) {
    extern __shared__ float4 lds_f4[];
    const int tid = threadIdx.x;
    Tracer T0 = make_tracer<MODE> ( p.scene, lds_f4, p.stack_depth, p.leaf_cap, p.lds_nodes, p.lds_tris, p.leaf_cull != 0, p.fused_slab != 0 );
    T0.faults = p.counters + kCtrFaults;
    if ( MODE >= 2 && p.stack_spill ) { T0.spill = p.stack_spill + ( size_t ) ( blockIdx.x * 256u + threadIdx.x ) * p.spill_cap; T0.spill_cap = p.spill_cap; }
    const Tracer T = T0;

    PixelStreams rs = trng_pixel_streams ( 0, 0, 0 );
    Counters c = counters_zero();
    float* acc_lds = reinterpret_cast<float*> ( T.stack - tid ) + ( p.stack_depth + p.leaf_cap ) * TERRA_COL + tid;     // the parked words follow the leaf list
    job_init_lane ( p, acc_lds );       // (make_tracer's barrier came before; a wave only ever touches its own pool words)
    Jobs jb; jb.px = jb.py = 0; jb.s = p.chunk_spp; jb.exhausted = false; jb.base = 0;
    SamplerPair sp = sampler_pair_none();       // (lives only in the KINDS & TERRA_KIND_SAMPLER variants)
    V3 Lo = v3 ( 0, 0, 0 ), throughput = v3 ( 1, 1, 1 );
    Ray ray = make_ray ( v3 ( 0, 0, 0 ), v3 ( 0, 0, 1 ) );
    uint32_t bounce = 0;
    const V3 cam_pos = v3p ( p.cam_pos );

    if constexpr ( TERRA_DECOUPLED_MIS ( INTEGRATOR, MODE, KINDS ) ) {
        // Decoupled loop for Direct + MIS: like the Direct one below with two shadow jobs per shaded hit, in the reference's
        // order -- A: the ray to the light sample, B: the BSDF-sampled ray (mis_prepare / mis_finish_b).
        LaneTraversal lt = lane_traversal_idle ( T, ray );
        int job = 0;                                             // 0 path segment, 1 shadow ray A, 2 shadow ray B
        bool done = false, have_ray = false, cont = false;
        MisPending pend; pend.a_vis = pend.a_hid = pend.f2 = pend.p = pend.t_before = v3 ( 0, 0, 0 ); pend.expected = 0; pend.bpdf2 = pend.cos2 = 0.f; pend.light_object = 0;
        V3 cont_o = v3 ( 0, 0, 0 ), cont_d = v3 ( 0, 0, 1 ), b_o = v3 ( 0, 0, 0 ), b_d = v3 ( 0, 0, 1 ), lo_i = v3 ( 0, 0, 0 );
        for ( ;; ) {
            if ( !lt.traversing && !done && ! ( have_ray && lane_traversal_recheck<MODE> ( T, ray, lt ) ) ) {
                bool start = false;
                V3 ro = v3 ( 0, 0, 0 ), rd = v3 ( 0, 0, 1 );              // the ray that starts (its reciprocals are taken once, below)
                if ( have_ray && job != 0 ) {
                    Ray r = ray; r.o = r.o + r.d * 0.001f;
                    const bool hit = lt.best.tri != 0xffffffffu;
                    Surface lsf; uint32_t object = 0, tri_in_object = 0, nattr = 0, tri_s = 0xffffffffu;
                    V3 point = hit ? r.o + r.d * lt.best.depth : v3 ( FLT_MAX, FLT_MAX, FLT_MAX );
                    if ( hit ) {
                        if ( job == 2 ) { surface_init<MODE, KINDS> ( T, lt.best.tri, point, lsf, object, tri_in_object, nattr ); tri_s = MODE >= 2 ? T.sc.mats[object].first_tri + tri_in_object : lt.best.tri; }
                        else { tri_s = hit_soup_index<MODE> ( T, lt.best.tri ); const float4 t0 = reinterpret_cast<const float4*> ( T.sc.tris ) [3 * tri_s]; object = __float_as_uint ( t0.w ); nattr = T.sc.mats[object].attributes_count; }
                        if ( COUNT ) ++c.hits;
                        if ( COUNT == 2 ) c.attr_fetches += nattr + 1;
                    }
                    if ( job == 1 ) {                            // A came back: pick its outcome, send B
                        lo_i = tri_s == pend.expected ? pend.a_vis : pend.a_hid;
                        ro = b_o; rd = b_d; job = 2; start = true;
                    } else {                                     // B came back: the integrator's value is complete
                        Lo = Lo + mis_finish_b<MODE> ( T, pend, lo_i, hit, object, tri_s, point, lsf, ray.d );
                        job = 0;
                        if ( cont ) { ro = cont_o; rd = cont_d; start = true; }
                        else { deposit ( acc_lds, Lo ); have_ray = false; }
                    }
                } else if ( have_ray ) {                         // a path segment came back
                    if ( lt.best.tri != 0xffffffffu ) {
                        Surface sf;
                        V3 point = shade_surface<COUNT, MODE, KINDS> ( T, ray, lt.best, sf, c );
                        V3 wo = neg ( ray.d );
                        Ray ray_a;
                        pend = mis_prepare<COUNT, KINDS, MODE> ( T, sf, point, wo, throughput, bounce, rs.b, c, ray_a, b_d );
                        b_o = point + sf.normal * 0.0001f;        // surface_ray ( sf, point, bsdf_dir, 1.f ) without the divisions
                        cont = path_continue<COUNT, KINDS> ( T.sc, sf, wo, throughput, bounce, p.bounces, rs.b, c, cont_d, sp );
                        cont_o = point + sf.normal * 0.0001f;     // (the divisions of surface_ray are redone when the ray starts)
                        ro = ray_a.o; rd = ray_a.d; job = 1; start = true;
                    } else {
                        if ( ( KINDS & TERRA_KIND_ENV ) && T.sc.env_mode ) { throughput = had ( throughput, environment_eval ( T.sc, ray.d ) ); Lo = Lo + throughput; }
                        deposit ( acc_lds, Lo ); have_ray = false;
                    }
                }
                if ( !start ) {
                    if ( jb.s == p.chunk_spp ) { job_next<COUNT, MODE == 1> ( p, acc_lds, jb, rs, c ); if ( jb.exhausted ) done = true; }
                    if ( jb.s != p.chunk_spp ) {
                        float r1 = trng_a_float ( rs.a ), r2 = trng_a_float ( rs.a );
                        ro = cam_pos; rd = camera_sample ( p, jb.px, jb.py, r1, r2 );
                        if constexpr ( ( KINDS & TERRA_KIND_SAMPLER ) != 0 ) sp = sampler_pair_draw ( p.sampler_mode, p.sampler_strata, ( uint64_t ) jb.base + jb.s, rs.a );
                        Lo = v3 ( 0, 0, 0 ); throughput = v3 ( 1, 1, 1 ); bounce = 0; ++jb.s; job = 0; start = true;
                    }
                }
                if ( start ) { ray = make_ray ( ro, rd ); lane_traversal_start<COUNT, MODE> ( T, ray, lt, c ); have_ray = true; }
            }
            if ( !lane_traversal_run<COUNT, MODE> ( T, ray, lt, c ) && __all ( done ) ) break;      // nobody traversing and no job left
        }
    } else if constexpr ( TERRA_DECOUPLED_DIRECT ( INTEGRATOR, MODE, KINDS ) ) {
        // Decoupled loop for the Direct integrator. A lane's ray in flight is either a path segment (MAIN) or the shadow
        // ray of the hit it just shaded (SHADOW). Shading a MAIN hit draws the light sample, prepares both outcomes of the
        // shadow test (direct_prepare), samples the BSDF and plays Russian roulette -- all stream draws in the reference's
        // order -- parks the continuation ray and sends the shadow ray; when that returns, the matching outcome is added
        // and the continuation (or the pixel's next sample) starts. Same rays, same draws, same sums as integrate_direct.
        LaneTraversal lt = lane_traversal_idle ( T, ray );
        bool done = false, have_ray = false, shadow = false, cont = false;
        DirectPending pend; pend.vis = pend.hid = v3 ( 0, 0, 0 ); pend.expected = 0;
        V3 cont_o = v3 ( 0, 0, 0 ), cont_d = v3 ( 0, 0, 1 );
        for ( ;; ) {
            if ( !lt.traversing && !done && ! ( have_ray && lane_traversal_recheck<MODE> ( T, ray, lt ) ) ) {
                bool start = false;
                V3 ro = v3 ( 0, 0, 0 ), rd = v3 ( 0, 0, 1 );
                if ( have_ray && shadow ) {                      // the shadow ray came back
                    const uint32_t tri_s = ( TERRA_SHADOW_ANYHIT && MODE == 2 && COUNT == 0 && lt.best.tri == TERRA_TRI_EXPECTED ) ? pend.expected : hit_soup_index<MODE> ( T, lt.best.tri );
                    if ( tri_s != 0xffffffffu ) {                   // (its hit counts as a surface init, as in the coupled form)
                        if ( COUNT ) ++c.hits;
                        if ( COUNT == 2 ) { const float4 t0 = reinterpret_cast<const float4*> ( T.sc.tris ) [3 * tri_s]; c.attr_fetches += T.sc.mats[__float_as_uint ( t0.w )].attributes_count + 1; }
                    }
                    Lo = Lo + ( tri_s == pend.expected ? pend.vis : pend.hid );
                    shadow = false;
                    if ( cont ) { ro = cont_o; rd = cont_d; start = true; }
                    else { deposit ( acc_lds, Lo ); have_ray = false; }
                } else if ( have_ray ) {                         // a path segment came back
                    if ( lt.best.tri != 0xffffffffu ) {
                        Surface sf;
                        V3 point = shade_surface<COUNT, MODE, KINDS> ( T, ray, lt.best, sf, c );
                        V3 wo = neg ( ray.d );
                        Ray shadow_ray;
                        pend = direct_prepare<COUNT, KINDS, MODE> ( T, sf, point, wo, throughput, bounce, rs.b, c, shadow_ray );
                        cont = path_continue<COUNT, KINDS> ( T.sc, sf, wo, throughput, bounce, p.bounces, rs.b, c, cont_d, sp );
                        cont_o = point + sf.normal * 0.0001f;     // surface_ray ( sf, point, wi, 1.f ) without the divisions: they are taken when the ray starts
                        ro = shadow_ray.o; rd = shadow_ray.d; shadow = true; start = true;
                    } else {
                        if ( ( KINDS & TERRA_KIND_ENV ) && T.sc.env_mode ) { throughput = had ( throughput, environment_eval ( T.sc, ray.d ) ); Lo = Lo + throughput; }
                        deposit ( acc_lds, Lo ); have_ray = false;
                    }
                }
                if ( !start ) {                                  // the path ended (or none was started yet): the pixel's next sample, or the lane's next job
                    if ( jb.s == p.chunk_spp ) { job_next<COUNT, MODE == 1> ( p, acc_lds, jb, rs, c ); if ( jb.exhausted ) done = true; }
                    if ( jb.s != p.chunk_spp ) {
                        float r1 = trng_a_float ( rs.a ), r2 = trng_a_float ( rs.a );
                        ro = cam_pos; rd = camera_sample ( p, jb.px, jb.py, r1, r2 );
                        if constexpr ( ( KINDS & TERRA_KIND_SAMPLER ) != 0 ) sp = sampler_pair_draw ( p.sampler_mode, p.sampler_strata, ( uint64_t ) jb.base + jb.s, rs.a );
                        Lo = v3 ( 0, 0, 0 ); throughput = v3 ( 1, 1, 1 ); bounce = 0; ++jb.s; start = true;
                    }
                }
                if ( start ) {
                    ray = make_ray ( ro, rd ); lane_traversal_start<COUNT, MODE> ( T, ray, lt, c ); have_ray = true;
                    if ( shadow ) lane_traversal_expect<COUNT, MODE> ( T, ray, lt, pend.expected );
                }
            }
            if ( !lane_traversal_run<COUNT, MODE> ( T, ray, lt, c ) && __all ( done ) ) break;      // nobody traversing and no job left
        }
    } else if constexpr ( TERRA_DECOUPLED ( INTEGRATOR, MODE ) ) {
        // Decoupled loop (scenes read from global memory, integrators without nested raycasts): a lane is either
        // traversing its current ray or waiting to be shaded. The resumable traversal returns as soon as 1/16 of
        // the lanes that entered it have finished; those are shaded and handed their next ray (continuation or the
        // pixel's next camera sample) while the others keep their traversal state. Per pixel nothing changes: same
        // rays, same stream draws, same accumulation order.
        LaneTraversal lt = lane_traversal_idle ( T, ray );
        bool done = false, have_ray = false;
        for ( ;; ) {
            if ( !lt.traversing && !done && ! ( have_ray && lane_traversal_recheck<MODE> ( T, ray, lt ) ) ) {
                bool next = false;
                V3 ro = v3 ( 0, 0, 0 ), rd = v3 ( 0, 0, 1 );
                if ( have_ray ) {
                    if ( lt.best.tri != 0xffffffffu ) {
                        Surface sf;
                        PathDraws pd = path_draw<COUNT> ( T.sc.sincos24, rs.b, c );       // (these integrators draw nothing themselves: the variates, and the table load, come first)
                        if constexpr ( ( KINDS & TERRA_KIND_SAMPLER ) != 0 ) path_apply_sampler ( pd, sp, bounce );
                        V3 point = shade_surface<COUNT, MODE, KINDS> ( T, ray, lt.best, sf, c );
                        V3 wo = neg ( ray.d ), wi;
                        Lo = Lo + integrate<INTEGRATOR, COUNT, MODE, KINDS> ( T, ray, sf, point, wo, throughput, bounce, rs.b, c );
                        next = path_continue<KINDS> ( sf, wo, throughput, bounce, p.bounces, pd, wi );
                        if ( next ) { ro = point + sf.normal * 0.0001f; rd = wi; }       // surface_ray ( sf, point, wi, 1.f ); its reciprocals are taken below
                    } else if ( ( KINDS & TERRA_KIND_ENV ) && T.sc.env_mode ) {
                        throughput = had ( throughput, environment_eval ( T.sc, ray.d ) );
                        Lo = Lo + throughput;
                    }
                    if ( !next ) { deposit ( acc_lds, Lo ); have_ray = false; }
                }
                if ( !next ) {
                    if ( jb.s == p.chunk_spp ) { job_next<COUNT, MODE == 1> ( p, acc_lds, jb, rs, c ); if ( jb.exhausted ) done = true; }
                    if ( jb.s != p.chunk_spp ) {
                        float r1 = trng_a_float ( rs.a ), r2 = trng_a_float ( rs.a );
                        ro = cam_pos; rd = camera_sample ( p, jb.px, jb.py, r1, r2 );
                        if constexpr ( ( KINDS & TERRA_KIND_SAMPLER ) != 0 ) sp = sampler_pair_draw ( p.sampler_mode, p.sampler_strata, ( uint64_t ) jb.base + jb.s, rs.a );
                        Lo = v3 ( 0, 0, 0 ); throughput = v3 ( 1, 1, 1 ); bounce = 0; ++jb.s; next = true;
                    }
                }
                if ( next ) { ray = make_ray ( ro, rd ); lane_traversal_start<COUNT, MODE> ( T, ray, lt, c ); have_ray = true; }
            }
            if ( !lane_traversal_run<COUNT, MODE> ( T, ray, lt, c ) && __all ( done ) ) break;      // nobody traversing and no job left
        }
    } else {
    // Coupled loop (LDS-resident scenes; Direct / MIS everywhere their decoupled forms are off): every lane that holds a live path traces one
    // ray per iteration; a lane whose path ended starts its pixel's next sample in the same iteration, or its next job.
    bool alive = false;
    V3 ro = v3 ( 0, 0, 0 ), rd = v3 ( 0, 0, 1 );                     // origin / direction of the lane's next ray; the reciprocals are taken in ONE place for camera and continuation rays
    // The path's radiance Lo: a term is added on the few hits that emit (or, Direct / MIS, are lit) and the sum is needed when the path ends -- in between it only
    // occupies three registers of a kernel that has none to spare. LDS-resident scenes park it next to the job's sum. Adding a term of +-0 would leave every
    // component as it is (Lo is never -0: it starts at +0 and x + (-x) = +0), so skipping such terms is exact.
#ifndef TERRA_LO_IN_LDS_LIGHT     // ... also for Direct / MIS, whose Lo changes on most hits (A/B)
#define TERRA_LO_IN_LDS_LIGHT 1
#endif
    constexpr bool LO_LDS = TERRA_LO_IN_LDS && MODE == 1 && ( TERRA_LO_IN_LDS_LIGHT || !TERRA_IS_LIGHT ( INTEGRATOR ) );
    float* lo_lds = acc_lds + 6 * 256;
    auto lo_reset = [&] () { if ( LO_LDS ) { lo_lds[0] = 0.f; lo_lds[256] = 0.f; lo_lds[512] = 0.f; } else Lo = v3 ( 0, 0, 0 ); };
    auto lo_add = [&] ( V3 t ) {
        if ( LO_LDS ) { if ( t.x != 0.f || t.y != 0.f || t.z != 0.f ) { lo_lds[0] = lo_lds[0] + t.x; lo_lds[256] = lo_lds[256] + t.y; lo_lds[512] = lo_lds[512] + t.z; } }
        else Lo = Lo + t;
    };
    auto lo_deposit = [&] () { if ( LO_LDS ) deposit ( acc_lds, v3 ( lo_lds[0], lo_lds[256], lo_lds[512] ) ); else deposit ( acc_lds, Lo ); };
#ifndef TERRA_COUPLED_DIRECT_SPLIT     // the coupled loop's Direct integrator split at its shadow ray (below); 0: integrate_direct as one piece (A/B)
#define TERRA_COUPLED_DIRECT_SPLIT 1
#endif
#ifndef TERRA_COUPLED_MIS_SPLIT        // ... and Direct + MIS at its two rays
#define TERRA_COUPLED_MIS_SPLIT 1
#endif
#ifndef TERRA_REGEN_MIN           // (A/B) lanes whose path ended wait until this many of the wave's lanes have, then start their next camera rays TOGETHER (same bounce depth afterwards)
#define TERRA_REGEN_MIN 1
#endif
    while ( true ) {
        const bool any_alive = __any ( alive );
        if ( !alive && ( TERRA_REGEN_MIN <= 1 || !any_alive || __popcll ( __ballot ( !alive ) ) >= TERRA_REGEN_MIN ) ) {
            // (a lane waits at the boundary until TERRA_JOB_FETCH_MIN lanes do, or nobody is tracing: the switch then serves several lanes per execution)
            if ( jb.s == p.chunk_spp && ( TERRA_JOB_FETCH_MIN <= 1 || !any_alive || __popcll ( __ballot ( jb.s == p.chunk_spp ) ) >= TERRA_JOB_FETCH_MIN ) ) job_next<COUNT, MODE == 1> ( p, acc_lds, jb, rs, c );
            if ( jb.exhausted ) break;
            if ( jb.s != p.chunk_spp ) {
                PS_WAVE ( c, kPsCamIter ); PS_LANE ( c, kPsCamLanes );
                float r1 = trng_a_float ( rs.a ), r2 = trng_a_float ( rs.a );
                ro = cam_pos; rd = camera_sample ( p, jb.px, jb.py, r1, r2 );
                if constexpr ( ( KINDS & TERRA_KIND_SAMPLER ) != 0 ) sp = sampler_pair_draw ( p.sampler_mode, p.sampler_strata, ( uint64_t ) jb.base + jb.s, rs.a );
                lo_reset(); throughput = v3 ( 1, 1, 1 ); bounce = 0; alive = true; ++jb.s;
            }
        }
        if ( alive ) {
            ray = make_ray ( ro, rd );
            Surface sf;
            PS_WAVE ( c, kPsRayIter ); PS_LANE ( c, kPsRayLanes );
            constexpr bool pre_draw = !TERRA_IS_LIGHT ( INTEGRATOR );      // integrators that draw nothing themselves: the continuation variates come before the surface set-up
            PathDraws pd;
            RaycastResult h = scene_raycast<COUNT, MODE, KINDS> ( T, ray, sf, c, pre_draw ? &pd : nullptr, pre_draw ? &rs.b : nullptr );
            bool end = !h.hit;
            if ( h.hit ) {
                PS_WAVE ( c, kPsShadeIter ); PS_LANE ( c, kPsShadeLanes );
                V3 wo = neg ( ray.d ), wi;
                // Direct on scenes whose emissive attributes are constants: the integrator is split at its shadow ray. Everything that does not depend on the ray's outcome --
                // the light sample, both possible values of the integrator, then the continuation's draws, in the reference's order (src/Terra.c:1068-1079 after :1349-1426) --
                // comes BEFORE the shadow traversal, so that the traversal runs with a pending pair and a continuation ray in registers instead of the whole shaded surface
                constexpr bool SPLIT_DIRECT = TERRA_COUPLED_DIRECT_SPLIT && INTEGRATOR == 1 && ( KINDS & ( TERRA_KIND_TEX | TERRA_KIND_SAMPLER ) ) == 0;
                if constexpr ( SPLIT_DIRECT ) {
                    Ray shadow_ray;
                    const DirectPending pend = direct_prepare<COUNT, KINDS, MODE> ( T, sf, h.point, wo, throughput, bounce, rs.b, c, shadow_ray );
                    pd = path_draw<COUNT> ( T.sc.sincos24, rs.b, c );
                    end = !path_continue<KINDS> ( sf, wo, throughput, bounce, p.bounces, pd, wi );
                    if ( !end ) { ro = h.point + sf.normal * 0.0001f; rd = wi; }
                    const uint32_t tri = scene_raycast_triangle<COUNT, MODE> ( T, shadow_ray, c, pend.expected );
                    lo_add ( tri == pend.expected ? pend.vis : pend.hid );
                } else if constexpr ( TERRA_COUPLED_MIS_SPLIT && INTEGRATOR == 2 && ( KINDS & ( TERRA_KIND_TEX | TERRA_KIND_SAMPLER ) ) == 0 ) {
                    // Direct + MIS split the same way at its two rays (mis_prepare / mis_finish_b, src/Terra.c:1428-1587): A, the ray to the light sample, only has to
                    // name the triangle it hits; B, the BSDF-sampled ray, needs the surface it hits
                    Ray ray_a; V3 b_d;
                    const MisPending pend = mis_prepare<COUNT, KINDS, MODE> ( T, sf, h.point, wo, throughput, bounce, rs.b, c, ray_a, b_d );
                    const V3 next_o = h.point + sf.normal * 0.0001f;          // surface_ray ( sf, h.point, direction, 1.f ) for B and for the continuation alike
                    end = !path_continue<COUNT, KINDS> ( T.sc, sf, wo, throughput, bounce, p.bounces, rs.b, c, wi, sp );
                    if ( !end ) { ro = next_o; rd = wi; }
                    const uint32_t tri_a = scene_raycast_triangle<COUNT, MODE> ( T, ray_a, c, pend.expected );
                    const V3 lo_a = tri_a == pend.expected ? pend.a_vis : pend.a_hid;
                    Surface lsf;
                    const RaycastResult hb = scene_raycast<COUNT, MODE, KINDS> ( T, make_ray ( next_o, b_d ), lsf, c );
                    lo_add ( mis_finish_b<MODE> ( T, pend, lo_a, hb.hit, hb.object, hb.tri, hb.point, lsf, b_d ) );
                } else {
                lo_add ( integrate<INTEGRATOR, COUNT, MODE, KINDS> ( T, ray, sf, h.point, wo, throughput, bounce, rs.b, c ) );
                if ( !pre_draw ) pd = path_draw<COUNT> ( T.sc.sincos24, rs.b, c );
                if constexpr ( ( KINDS & TERRA_KIND_SAMPLER ) != 0 ) path_apply_sampler ( pd, sp, bounce );
                end = !path_continue<KINDS> ( sf, wo, throughput, bounce, p.bounces, pd, wi );
                if ( !end ) { ro = h.point + sf.normal * 0.0001f; rd = wi; }      // surface_ray ( sf, h.point, wi, 1.f ): its make_ray is the one at the top of this block
                }
            } else if ( ( KINDS & TERRA_KIND_ENV ) && T.sc.env_mode && !env_reaches_by_samples<INTEGRATOR, KINDS> ( T.sc, bounce ) ) {     // extension: the reference's commented-out "Lo += throughput" (src/Terra.c:1056)
                throughput = had ( throughput, environment_eval ( T.sc, ray.d ) );
                lo_add ( throughput );
            }
            if ( end ) { lo_deposit(); alive = false; }
        }
    }
    }

    if ( COUNT ) wave_flush_counters ( c, p.counters );
    if ( COUNT == 2 && p.leaf_cull ) {
        unsigned long long x = c.tri_culled;
        for ( int off = 32; off > 0; off >>= 1 ) x += __shfl_xor ( x, off, 64 );
        if ( ( threadIdx.x & 63 ) == 0 && x ) atomicAdd ( &p.counters[kCtrTriCulled], x );
    }
#if TERRA_PHASE_STATS
    for ( int k = 0; k < 16; ++k ) {
        unsigned long long x = c.ps[k];
        for ( int off = 32; off > 0; off >>= 1 ) x += __shfl_xor ( x, off, 64 );
        if ( ( threadIdx.x & 63 ) == 0 && x ) atomicAdd ( &p.counters[kCtrDbg0 + k], x );
    }
#endif
}

// ---- launch -------------------------------------------------------------------

#if TERRA_TU_HAS ( 0 )
static uint32_t own_tiles ( uint32_t w, uint32_t h, uint32_t tile, uint32_t rank, uint32_t world ) {
    uint32_t tiles = ( ( w + tile - 1 ) / tile ) * ( ( h + tile - 1 ) / tile );
    return tiles > rank ? ( tiles - rank + world - 1 ) / world : 0;
}

// LDS a MODE-1 block spends on materials, lights and triangle areas (make_tracer): three sections, each a multiple of 16 bytes
static size_t scene_extra_lds_bytes ( uint32_t n_objects, uint32_t n_lights, uint32_t n_tris ) {
    return ( ( ( size_t ) n_objects * sizeof ( DevMaterial ) + 15 ) & ~size_t ( 15 ) ) + ( size_t ) n_lights * sizeof ( DevLight ) + ( ( ( size_t ) n_tris * 4 + 15 ) & ~size_t ( 15 ) );
}
size_t terra_lds_bytes ( const DevRenderParams& p ) {
    return ( size_t ) ( p.stack_depth + p.leaf_cap + ( p.lds_mode == 1 ? TERRA_AUX_WORDS_LDS : TERRA_AUX_WORDS ) ) * 1024 + ( size_t ) p.lds_nodes * TERRA_LDS_NODE_BYTES + ( size_t ) p.lds_tris * ( 48 + 64 )
           + ( p.lds_mode == 1 ? scene_extra_lds_bytes ( p.scene.n_objects, p.scene.n_lights, p.scene.n_tris ) : 0 );
}
// fast tree (MODE 2 / 3): nothing is staged. A lane holds at most two leaves (in registers: the one it tests, the next one), so there is no leaf list. The stack: its first TERRA_FAST_STACK_LDS entries
// in LDS (1 KB per entry and block), the rest -- up to the tree's worst case, which a ray almost never reaches -- in HBM, 4 bytes per entry and resident lane
// (DevRenderParams::stack_spill, part of the launch's scratch: trace_device.h fast_push / fast_pop). Depth no longer decides whether a tree can be launched.
// (Rounds 2-3 staged the first 64 nodes as plain 64-byte nodes read through a flat load: +3.7 % then. Flat loads go through the texture addresser like global ones,
// and that unit is what binds these kernels: nothing is gained by it now.)
#ifndef TERRA_FAST_STACK_LDS
#define TERRA_FAST_STACK_LDS 16
#endif
void terra_plan_fast_tree ( DevRenderParams& p ) {
    const uint32_t need = ( uint32_t ) ( p.scene.fast_max_stack < 1 ? 1 : p.scene.fast_max_stack );
    p.lds_mode = 2; p.lds_tris = 0; p.lds_nodes = 0; p.leaf_cap = 0; p.stack_depth = need < ( uint32_t ) TERRA_FAST_STACK_LDS ? need : ( uint32_t ) TERRA_FAST_STACK_LDS;
    p.spill_cap = need - p.stack_depth; p.stack_spill = nullptr;
}
// resident lanes a fast-tree launch can have at most (8 blocks of 256 threads per CU): what the spill area is sized for
size_t terra_fast_spill_bytes ( const DevRenderParams& p ) {
    if ( p.lds_mode != 2 || p.spill_cap == 0 ) return 0;
    int cus = 0, dev = 0; ( void ) hipGetDevice ( &dev );
    if ( hipDeviceGetAttribute ( &cus, hipDeviceAttributeMultiprocessorCount, dev ) != hipSuccess || cus < 1 ) { ( void ) hipGetLastError(); cus = 256; }
    const size_t blocks = ( size_t ) p.job_blocks < ( size_t ) cus * 8 ? ( size_t ) p.job_blocks : ( size_t ) cus * 8;
    return blocks * 256 * ( size_t ) p.spill_cap * sizeof ( uint32_t );
}

// LDS plan. Small scenes (whole scene + stack + a leaf list of at least TERRA_LEAF_CAP_RESIDENT_MIN entries <= budget): stage
// everything; with the Cornell box that is 31.9 KB per block (112-B staged nodes, 14-entry leaf list), so the 5 blocks/CU the
// Simple kernel's registers allow stay resident. The leaf list takes what the budget leaves, up to 16 entries: a list that
// fills is drained and the node loop resumes, so its length only decides how often that happens.
// Large scenes: nothing is staged -- their node fetches are bound by the L1 tag rate of divergent
// 16-byte loads (each lane its own 64-B node) and by latency, so resident blocks matter most: the
// leaf list takes what is left of the CU's 160 KB after fitting as many blocks as possible while
// keeping at least 8 entries (profiles/r01_measurements/ab4.log, ab5.log: 4 blocks x 14 entries 219 ms vs
// 3 blocks x 16 entries 305 ms vs 4-entry lists 261 ms on the 97k-triangle hall).
#ifndef TERRA_LDS_CU_KB
#define TERRA_LDS_CU_KB 158
#endif
#ifndef TERRA_LEAF_CAP_MIN       // smallest leaf list worth an extra resident block (with the decoupled loop, hall: 5 blocks x 6 entries
#define TERRA_LEAF_CAP_MIN 6     // 391 ms vs 4 blocks x 14 entries 400 ms, Direct 473 vs 487 ms; profiles/r01_measurements/ab_lc*.log)
#endif
#ifndef TERRA_LDS_BUDGET          // per block, so that FIVE blocks stay resident per CU: a CU does not hand out all of its 160 KB -- 5 x 31,632 B fit, 5 x 32,656 B
#define TERRA_LDS_BUDGET ( TERRA_LDS_CU_KB * 1024 / 5 )      // do not (measured: 4.57 -> 3.67 waves per SIMD and 65.6 -> 74.4 ms on the Cornell frame, profiles/r03_measurements/lds_cliff.log)
#endif
#ifndef TERRA_LEAF_CAP_RESIDENT_MIN
#define TERRA_LEAF_CAP_RESIDENT_MIN 8
#endif
// leaf-list entries an LDS-resident plan can afford (0 = the scene does not fit)
static uint32_t resident_leaf_cap ( uint32_t n_nodes, uint32_t n_tris, int max_stack, uint32_t n_objects, uint32_t n_lights ) {
    const uint32_t depth = max_stack < 1 ? 1u : ( uint32_t ) max_stack;
    const size_t fixed = ( size_t ) ( depth + TERRA_AUX_WORDS_LDS ) * 1024 + ( size_t ) n_nodes * TERRA_LDS_NODE_BYTES + ( size_t ) n_tris * 112 + scene_extra_lds_bytes ( n_objects, n_lights, n_tris );
    if ( fixed + ( size_t ) TERRA_LEAF_CAP_RESIDENT_MIN * 1024 > ( size_t ) TERRA_LDS_BUDGET ) return 0;
    const uint32_t cap = ( uint32_t ) ( ( ( size_t ) TERRA_LDS_BUDGET - fixed ) / 1024 );
    return cap > TERRA_LEAF_CAP_MAX ? TERRA_LEAF_CAP_MAX : cap;
}
bool terra_scene_fits_lds ( uint32_t n_nodes, uint32_t n_tris, int max_stack, uint32_t n_objects, uint32_t n_lights ) { return resident_leaf_cap ( n_nodes, n_tris, max_stack, n_objects, n_lights ) != 0; }
void terra_plan_lds ( DevRenderParams& p ) {
    uint32_t depth = p.scene.max_stack < 1 ? 1u : ( uint32_t ) p.scene.max_stack;
    p.stack_depth = depth;
    p.leaf_cap = TERRA_LEAF_CAP_MAX;
    if ( const uint32_t cap = resident_leaf_cap ( p.scene.n_nodes, p.scene.n_tris, p.scene.max_stack, p.scene.n_objects, p.scene.n_lights ) ) {
        p.lds_mode = 1; p.lds_nodes = p.scene.n_nodes; p.lds_tris = p.scene.n_tris; p.leaf_cap = cap;
        return;
    }
    p.lds_mode = 0; p.lds_nodes = 0; p.lds_tris = 0;
    for ( int blocks = 5; blocks >= 1; --blocks ) {
        int room = TERRA_LDS_CU_KB / blocks - ( int ) depth - TERRA_AUX_WORDS;       // KB per block left for the leaf list (2 KB of slack per CU)
        if ( room >= TERRA_LEAF_CAP_MIN || blocks == 1 ) { p.leaf_cap = ( uint32_t ) ( room > TERRA_LEAF_CAP_MAX ? TERRA_LEAF_CAP_MAX : ( room < 4 ? 4 : room ) ); break; }
    }
    // a deep tree: keep the block within the 64 KB a launch may ask for without an opt-in while the leaf list keeps at least 4 entries; deeper still, the launch opts in
    // (launch_instance: hipFuncAttributeMaxDynamicSharedMemorySize, one block per CU) up to TERRA_LDS_BLOCK_MAX_KB, beyond which terra_launch_render refuses with a message
    while ( p.leaf_cap > 4 && terra_lds_bytes ( p ) > ( size_t ) 64 * 1024 ) --p.leaf_cap;
}

#endif

#ifndef TERRA_LDS_BLOCK_MAX_KB      // the most dynamic LDS one block may opt in to (a CU's 160 KB less what the runtime keeps)
#define TERRA_LDS_BLOCK_MAX_KB 156
#endif
#if TERRA_TU_HAS ( 0 )
size_t terra_lds_block_limit ( void ) { return ( size_t ) TERRA_LDS_BLOCK_MAX_KB * 1024; }
#endif
// blocks of one kernel instance the GPU holds at once (occupancy x CUs), cached per (kernel, LDS size): the persistent grid
static uint32_t resident_blocks ( const void* fn, size_t lds ) {
    struct Key { const void* fn; size_t lds; int dev; uint32_t blocks; };
    static thread_local Key cache[8]; static thread_local int used = 0;
    int dev = 0; ( void ) hipGetDevice ( &dev );
    for ( int i = 0; i < used; ++i ) if ( cache[i].fn == fn && cache[i].lds == lds && cache[i].dev == dev ) return cache[i].blocks;
    int per_cu = 0, cus = 0;
    if ( hipOccupancyMaxActiveBlocksPerMultiprocessor ( &per_cu, fn, 256, lds ) != hipSuccess || per_cu < 1 ) { ( void ) hipGetLastError(); per_cu = 1; }
    if ( hipDeviceGetAttribute ( &cus, hipDeviceAttributeMultiprocessorCount, dev ) != hipSuccess || cus < 1 ) { ( void ) hipGetLastError(); cus = 256; }
    const uint32_t blocks = ( uint32_t ) per_cu * ( uint32_t ) cus;
    Key& k = cache[used < 8 ? used++ : 7]; k.fn = fn; k.lds = lds; k.dev = dev; k.blocks = blocks;
    return blocks;
}
template <int I, int COUNT, int MODE, int KINDS>
static hipError_t launch_instance ( const DevRenderParams& p, size_t lds, hipStream_t stream ) {
    auto fn = terra_render_kernel<I, COUNT, MODE, KINDS>;
    if ( lds > ( size_t ) 64 * 1024 ) {          // a traversal stack deeper than ~55 entries (deep reference tree, Morton-ordered tree over clustered geometry): opt in, once per kernel and size
        if ( lds > ( size_t ) TERRA_LDS_BLOCK_MAX_KB * 1024 ) return hipErrorInvalidValue;      // (terra_launch_render checks first and says why)
        static thread_local size_t opted[8] = { 0 }; int dev = 0; ( void ) hipGetDevice ( &dev );
        if ( opted[dev & 7] < lds ) {
            const hipError_t e = hipFuncSetAttribute ( reinterpret_cast<const void*> ( fn ), hipFuncAttributeMaxDynamicSharedMemorySize, ( int ) lds );
            if ( e != hipSuccess ) return e;
            opted[dev & 7] = lds;
        }
    }
    uint32_t grid = p.job_blocks;
    if ( p.job_queue ) { const uint32_t cap = resident_blocks ( reinterpret_cast<const void*> ( fn ), lds ); if ( grid > cap ) grid = cap; }
    if ( MODE >= 2 && p.spill_cap && ( !p.stack_spill || ( size_t ) grid * 256 * p.spill_cap * sizeof ( uint32_t ) > terra_fast_spill_bytes ( p ) ) ) return hipErrorInvalidValue;      // (the spill area is sized for 8 blocks per CU)
    hipLaunchKernelGGL ( fn, dim3 ( grid ), dim3 ( 256 ), lds, stream, p );
    return hipGetLastError();
}
template <int I, int MODE, int KINDS>
static hipError_t launch_kinds ( const DevRenderParams& p, size_t lds, hipStream_t stream ) {
    // work counters are instrumentation, off unless asked for (terra_amd_set_work_counters / a per-pixel draw-count buffer): the counting kernels carry
    // 4-7 more live registers per lane and cost the Cornell frame 6 % (59.3 vs 62.9 ms; profiles/r03_measurements/ab_counters.log)
    if ( p.count_level == 0 ) return launch_instance<I, 0, MODE, KINDS> ( p, lds, stream );
    return launch_instance<I, 2, MODE, KINDS> ( p, lds, stream );
}
// kinds present in the scene -> the leanest compiled variant that covers them: diffuse only (1), diffuse + Phong (3: what
// OBJ/MTL scenes map to, satellite/src/Scene.cpp:193-230), everything (GGX, glass, textures, environment term), or everything + the sampler integration
#ifndef TERRA_KINDS_PRESETS_VARIANT
#define TERRA_KINDS_PRESETS_VARIANT 1
#endif
template <int I, int MODE>
static hipError_t launch_mode ( const DevRenderParams& p, size_t lds, hipStream_t stream ) {
    if ( p.bsdf_kinds == 1 ) return launch_kinds<I, MODE, 1> ( p, lds, stream );
    if ( ( p.bsdf_kinds & ~3u ) == 0 ) return launch_kinds<I, MODE, 3> ( p, lds, stream );
    if constexpr ( TERRA_KINDS_PRESETS_VARIANT && I <= 2 && MODE != 0 ) {       // the four presets on constant attributes, no environment term (BASELINE config 4's sphere scene): the usual integrators, off the reference tree
        if ( ( p.bsdf_kinds & ~15u ) == 0 ) return launch_kinds<I, MODE, 15> ( p, lds, stream );
    }
    if ( ( p.bsdf_kinds & TERRA_KIND_SAMPLER ) == 0 ) return launch_kinds<I, MODE, TERRA_KINDS_ALL & ~TERRA_KIND_SAMPLER> ( p, lds, stream );      // (the sampler integration costs the generic
    return launch_kinds<I, MODE, TERRA_KINDS_ALL> ( p, lds, stream );                                                                            //  kernel 11 % when merely compiled in: its own variant)
}
template <int MODE>
static hipError_t launch_integrator ( const DevRenderParams& p, size_t lds, hipStream_t stream ) {
    switch ( p.integrator ) {
        case 0: return launch_mode<0, MODE> ( p, lds, stream );
        case 1: return launch_mode<1, MODE> ( p, lds, stream );
        case 2: return launch_mode<2, MODE> ( p, lds, stream );
        case 3: return launch_mode<3, MODE> ( p, lds, stream );
        case 4: return launch_mode<4, MODE> ( p, lds, stream );
        case 5: return launch_mode<5, MODE> ( p, lds, stream );
        case 6: return launch_mode<6, MODE> ( p, lds, stream );
        default: return hipErrorInvalidValue;
    }
}
// one launcher per template MODE, each in its own translation unit (see the top of the file)
#if TERRA_TU_HAS ( 0 )
hipError_t terra_launch_render_mode0 ( const DevRenderParams& p, size_t lds, hipStream_t stream ) { return launch_integrator<0> ( p, lds, stream ); }
#endif
#if TERRA_TU_HAS ( 1 )
hipError_t terra_launch_render_mode1 ( const DevRenderParams& p, size_t lds, hipStream_t stream ) { return launch_integrator<1> ( p, lds, stream ); }
#endif
#if TERRA_TU_HAS ( 2 )
hipError_t terra_launch_render_mode2 ( const DevRenderParams& p, size_t lds, hipStream_t stream ) { return launch_integrator<2> ( p, lds, stream ); }
#endif
#if TERRA_TU_HAS ( 3 )
hipError_t terra_launch_render_mode3 ( const DevRenderParams& p, size_t lds, hipStream_t stream ) { return launch_integrator<3> ( p, lds, stream ); }
#endif

#if TERRA_TU_HAS ( 0 )
// p.job_blocks and p.partials must be set (scene_host.cpp launch_render); p.job_queue (a zeroed word) = persistent grid fed by the queue, nullptr = plain launch
bool terra_render_wants_queue ( const DevRenderParams& p ) {
#ifdef TERRA_QUEUE_NEVER         // A/B builds: every loop launched plainly
    ( void ) p; return false;
#else
    ( void ) p; return true;
#endif
}
hipError_t terra_launch_render ( const DevRenderParams& p, hipStream_t stream ) {
    if ( p.job_blocks == 0 ) return hipSuccess;
    if ( !p.partials ) return hipErrorInvalidValue;
    size_t lds = terra_lds_bytes ( p );
    // DevRenderParams::lds_mode 2 = the fast tree; scenes outside the coordinate range of the containment proof (DevScene::reach) run the kernels that carry the replay
    if ( p.lds_mode == 1 ) return terra_launch_render_mode1 ( p, lds, stream );
    if ( p.lds_mode == 2 ) return p.scene.reach ? terra_launch_render_mode3 ( p, lds, stream ) : terra_launch_render_mode2 ( p, lds, stream );
    return terra_launch_render_mode0 ( p, lds, stream );
}

uint32_t terra_render_blocks ( const DevRenderParams& p ) {
    uint32_t bpt = p.tile_size / 16;
    return own_tiles ( p.w, p.h, p.tile_size, p.rank, p.world ) * bpt * bpt;
}
// First kernel of every render: the random streams of every job of the launch (what job_next would otherwise compute when a lane takes the job). One thread per job,
// numbered like the render kernel's jobs; a job whose pixel lies outside the rectangle has no entry (nobody reads it).
__global__ __launch_bounds__ ( 256 ) void terra_job_streams_kernel ( DevRenderParams p ) {
    const uint32_t job = blockIdx.x * 256u + threadIdx.x;
    uint32_t px, py;
    const uint32_t vblock = job >> 8, chunk = vblock & ( p.split - 1 ), v = vblock >> p.split_log2;
    if ( !job_pixel_of_block ( p, p.block_order ? p.block_order[v] : v, job & 255u, px, py ) ) { p.job_streams[2 * ( size_t ) job + 1] = make_uint4 ( 0u, 0u, 0xffffffffu, 0u ); return; }
    const int prior_samples = reinterpret_cast<const DevResult*> ( p.results ) [ ( size_t ) ( py - p.st_y ) * p.st_pitch + ( px - p.st_x )].samples;      // keys the streams; the sum itself is the resolve kernel's business
    const PixelStreams rs = trng_pixel_streams ( p.frame_seed, ( uint64_t ) py * p.fb_w + px, ( uint64_t ) ( uint32_t ) prior_samples + ( uint64_t ) chunk * p.chunk_spp );
    p.job_streams[2 * ( size_t ) job] = make_uint4 ( ( uint32_t ) rs.a.state, ( uint32_t ) ( rs.a.state >> 32 ), ( uint32_t ) rs.b.state, ( uint32_t ) ( rs.b.state >> 32 ) );
    p.job_streams[2 * ( size_t ) job + 1] = make_uint4 ( ( uint32_t ) rs.b.inc, ( uint32_t ) ( rs.b.inc >> 32 ), px | ( py << 16 ), ( uint32_t ) prior_samples );      // (frames of up to 65,535 x 65,535: scene_host.cpp launch_render refuses larger ones)
}
// ---- job order ------------------------------------------------------------------------------------------------------------
// A persistent launch ends with its TAIL: when the queue runs dry every lane is inside a job, and the kernel lasts until the longest of those is done -- about 1.1 ms on the
// Cornell frame whatever the launch's share of it (profiles/r04_measurements/launch_fixed_cost.log), a seventh of a 1/8 shard. What a job costs is mostly decided by whether
// its pixel's camera rays hit anything: a sample whose camera ray leaves the scene is one short traversal, nothing to shade, no bounce. So the launches that key their
// streams ahead (LDS-resident scenes: a few dozen triangles) first CLASSIFY their 16x16 pixel blocks -- five camera rays per block (centre and corners, no jitter) against
// every triangle, the test the traversal applies -- and hand the blocks out hit ones first, empty ones last: the long jobs are under way while there is still plenty of
// short work to fill the lanes beside them, and the launch ends on jobs of a few ray iterations. Only the ORDER in which jobs are taken changes: which job a lane runs
// never mattered (render_kernels.hip "jobs"), a block's sums are stored under its place in the order and the resolve kernel looks them up there (DevRenderParams::block_order).
#define TERRA_CLASS_LDS_TRIS 256          // triangles the classifier stages in LDS (12 KB); larger scenes are read from global memory
__global__ __launch_bounds__ ( 256 ) void terra_block_class_kernel ( DevRenderParams p, uint32_t nblocks, uint32_t* cls ) {
    __shared__ float4 staged[3 * TERRA_CLASS_LDS_TRIS];
    const float4* tris = reinterpret_cast<const float4*> ( p.scene.tris );
    if ( p.scene.n_tris <= TERRA_CLASS_LDS_TRIS ) {          // (every thread of the block takes part, also those beyond the last pixel block)
        for ( uint32_t i = threadIdx.x; i < 3 * p.scene.n_tris; i += 256u ) staged[i] = tris[i];
        __syncthreads();
        tris = staged;
    }
    // eight lanes per pixel block, five of them with a probe each (centre and corners): the block's class is the vote of its lanes
    const uint32_t id = blockIdx.x * 256u + threadIdx.x, b = id >> 3, k = id & 7u;
    bool hit = false;
    if ( b < nblocks && k < 5u ) {
        const uint32_t lx = k == 0 ? 8u : ( ( k - 1 ) & 1u ) * 15u, ly = k == 0 ? 8u : ( ( k - 1 ) >> 1 ) * 15u;
        const uint32_t tid = ( ( ( lx >> 3 ) | ( ( ly >> 3 ) << 1 ) ) << 6 ) | ( lx & 7u ) | ( ( ly & 7u ) << 3 );      // (block_pixel's thread -> pixel map, inverted)
        uint32_t px, py;
        if ( job_pixel_of_block ( p, b, tid, px, py ) ) {                    // (else: a pixel outside the rectangle)
            Ray r = make_ray ( v3p ( p.cam_pos ), camera_sample ( p, px, py, 0.5f, 0.5f ) );
            r.o = r.o + r.d * 0.001f;
            const RayState st = ray_state_init ( r );
            for ( uint32_t t = 0; t < p.scene.n_tris && !hit; ++t ) {
                const float4 a = tris[3 * t], bb = tris[3 * t + 1], cc = tris[3 * t + 2];
                TriHit h;
                hit = watertight ( r, st, v3 ( a.x, a.y, a.z ), v3 ( bb.x, bb.y, bb.z ), v3 ( cc.x, cc.y, cc.z ), h );
            }
        }
    }
    const unsigned long long votes = __ballot ( hit );
    if ( b < nblocks && k == 0 ) cls[b] = ( ( votes >> ( threadIdx.x & 56u ) ) & 0xffull ) ? 0u : 1u;
}
// order[0 .. n): the blocks of class 0 in their own order, then those of class 1; order[n + b]: block b's place. One block of 256 threads.
__global__ __launch_bounds__ ( 256 ) void terra_block_order_kernel ( uint32_t n, const uint32_t* cls, uint32_t* order ) {
    __shared__ uint32_t before0[256];
    __shared__ uint32_t total0;
    const uint32_t t = threadIdx.x, per = ( n + 255u ) / 256u;
    const uint32_t lo = t * per < n ? t * per : n, hi = lo + per < n ? lo + per : n;
    uint32_t c0 = 0;
    for ( uint32_t i = lo; i < hi; ++i ) c0 += cls[i] == 0u;
    before0[t] = c0;
    __syncthreads();
    if ( t == 0 ) { uint32_t run = 0; for ( uint32_t k = 0; k < 256u; ++k ) { const uint32_t c = before0[k]; before0[k] = run; run += c; } total0 = run; }
    __syncthreads();
    uint32_t at0 = before0[t], at1 = total0 + ( lo - before0[t] );
    for ( uint32_t i = lo; i < hi; ++i ) {
        const uint32_t v = cls[i] == 0u ? at0++ : at1++;
        order[v] = i; order[n + i] = v;
    }
}
#ifndef TERRA_JOB_ORDER_MIN_BLOCKS
#define TERRA_JOB_ORDER_MIN_BLOCKS 256
#endif
uint32_t terra_job_order_min_blocks ( void ) { return TERRA_JOB_ORDER_MIN_BLOCKS; }
size_t terra_block_order_bytes ( const DevRenderParams& p, bool small_too ) {          // class word + the two halves of the order per pixel block, or 0: this launch keeps the order of the numbering
    if ( terra_job_streams_bytes ( p ) == 0 || p.scene.n_tris == 0 || p.scene.n_tris > 4096 ) return 0;
    const size_t blocks = terra_render_blocks ( p );
    // (not for small launches -- below TERRA_JOB_ORDER_MIN_BLOCKS pixel blocks, a 256 x 256 rectangle: a tile-sized call is one of several in flight, whose work hides its tail,
    //  and its two extra small kernels would queue behind the other callers' render grids: the reference client's tile loop 66.5 -> 72.6 ms with them)
    return blocks >= ( small_too ? 1u : ( unsigned ) TERRA_JOB_ORDER_MIN_BLOCKS ) ? ( blocks * 3 * sizeof ( uint32_t ) + 255 ) & ~size_t ( 255 ) : 0;
}
hipError_t terra_launch_block_order ( const DevRenderParams& p, uint32_t* cls, hipStream_t stream ) {      // p.block_order = cls + blocks
    const uint32_t blocks = terra_render_blocks ( p );
    if ( !p.block_order || !cls || blocks == 0 ) return hipErrorInvalidValue;
    hipLaunchKernelGGL ( terra_block_class_kernel, dim3 ( ( blocks * 8 + 255 ) / 256 ), dim3 ( 256 ), 0, stream, p, blocks, cls );      // eight lanes per pixel block
    hipLaunchKernelGGL ( terra_block_order_kernel, dim3 ( 1 ), dim3 ( 256 ), 0, stream, blocks, ( const uint32_t* ) cls, const_cast<uint32_t*> ( p.block_order ) );
    return hipGetLastError();
}
hipError_t terra_launch_job_streams ( const DevRenderParams& p, hipStream_t stream ) {
    if ( terra_job_streams_bytes ( p ) == 0 ) return hipSuccess;
    if ( !p.job_streams ) return hipErrorInvalidValue;
    hipLaunchKernelGGL ( terra_job_streams_kernel, dim3 ( p.job_blocks ), dim3 ( 256 ), 0, stream, p );
    return hipGetLastError();
}
size_t terra_job_streams_bytes ( const DevRenderParams& p ) { return ( TERRA_JOB_STREAM_TABLE && p.lds_mode == 1 ) ? ( size_t ) p.job_blocks * 256 * 32 : 0; }      // (p.job_blocks set)
hipError_t terra_launch_resolve ( const DevRenderParams& p, hipStream_t stream ) {
    uint32_t blocks = terra_render_blocks ( p );
    if ( blocks == 0 ) return hipSuccess;
    hipLaunchKernelGGL ( terra_resolve_kernel, dim3 ( blocks ), dim3 ( 256 ), 0, stream, p );
    return hipGetLastError();
}

// ---- tile pack / unpack for the multi-GPU gather -------------------------------
// packed layout per tile: tile_size^2 pixels (12 B each, rows contiguous) then tile_size^2 results (16 B each)

template <bool PACK>
__global__ __launch_bounds__ ( 256 ) void terra_tiles_kernel ( float* pixels, DevResult* results, uint32_t fb_w, uint32_t x, uint32_t y, uint32_t w, uint32_t h,
                                                                  uint32_t tile, uint32_t rank, uint32_t world, float* packed ) {
    const uint32_t tiles_x = ( w + tile - 1 ) / tile;
    const uint32_t per_tile = tile * tile;
    const uint32_t k = blockIdx.y;
    const uint32_t t = rank + k * world;
    const uint32_t tx = t % tiles_x, ty = t / tiles_x;
    float* tile_base = packed + ( size_t ) k * per_tile * 7;      // 3 + 4 floats per pixel
    float* ppix = tile_base;
    DevResult* pres = reinterpret_cast<DevResult*> ( tile_base + ( size_t ) per_tile * 3 );
    for ( uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < per_tile; i += gridDim.x * blockDim.x ) {
        uint32_t lx = tx * tile + i % tile, ly = ty * tile + i / tile;
        if ( lx >= w || ly >= h ) {
            if ( PACK ) { ppix[3 * i] = ppix[3 * i + 1] = ppix[3 * i + 2] = 0.f; DevResult z = { { 0.f, 0.f, 0.f }, 0 }; pres[i] = z; }
            continue;
        }
        size_t pix = ( size_t ) ( y + ly ) * fb_w + ( x + lx );
        if ( PACK ) {
            ppix[3 * i] = pixels[3 * pix]; ppix[3 * i + 1] = pixels[3 * pix + 1]; ppix[3 * i + 2] = pixels[3 * pix + 2];
            pres[i] = results[pix];
        } else {
            pixels[3 * pix] = ppix[3 * i]; pixels[3 * pix + 1] = ppix[3 * i + 1]; pixels[3 * pix + 2] = ppix[3 * i + 2];
            results[pix] = pres[i];
        }
    }
}

hipError_t terra_launch_tiles ( bool pack, float* pixels, void* results, uint32_t fb_w, uint32_t x, uint32_t y, uint32_t w, uint32_t h,
                                uint32_t tile, uint32_t rank, uint32_t world, float* packed, hipStream_t stream ) {
    uint32_t n = own_tiles ( w, h, tile, rank, world );
    if ( n == 0 ) return hipSuccess;
    uint32_t per_tile = tile * tile;
    dim3 grid ( ( per_tile + 255 ) / 256 < 64 ? ( per_tile + 255 ) / 256 : 64, n );
    if ( pack ) hipLaunchKernelGGL ( terra_tiles_kernel<true>, grid, dim3 ( 256 ), 0, stream, pixels, ( DevResult* ) results, fb_w, x, y, w, h, tile, rank, world, packed );
    else hipLaunchKernelGGL ( terra_tiles_kernel<false>, grid, dim3 ( 256 ), 0, stream, pixels, ( DevResult* ) results, fb_w, x, y, w, h, tile, rank, world, packed );
    return hipGetLastError();
}
#endif      // TERRA_TU_HAS ( 0 )
