// tree_build.cpp -- the two host-side tree builders (tree_build.h). Plain C++: no device code, no HIP calls.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <system_error>
#include <sched.h>
#include <cstdio>
#include <cstdlib>
#include <cfloat>
#include <cmath>
#include <cstring>
#include "tree_build.h"

// ---- BVH build: the reference's tree ------------------------
// Per-triangle boxes inflated by 1e-4 (double add, rounded), one stable sort by
// DESCENDING box-centre x (what the reference's bool comparator produces under
// glibc's merge sort), sweep SAH per range with the first minimum winning, inner
// boxes growing their max by 1e-4 per merge, children numbered left-then-right when
// the parent is expanded and the right range expanded first (LIFO task stack).
namespace bvh {
struct Volume { TerraAABB box; uint32_t index; };

static inline float fmin_sel ( float a, float b ) { return a < b ? a : b; }
static inline float fmax_sel ( float a, float b ) { return a > b ? a : b; }
TerraAABB empty_box() { TerraAABB b; b.min = { FLT_MAX, FLT_MAX, FLT_MAX }; b.max = { -FLT_MAX, -FLT_MAX, -FLT_MAX }; return b; }
void grow_by_triangle ( TerraAABB& b, const TerraTriangle& t ) {
    const double eps = 1e-4;
    b.min.x = ( float ) ( ( double ) fmin_sel ( fmin_sel ( fmin_sel ( b.min.x, t.a.x ), t.b.x ), t.c.x ) - eps );
    b.min.y = ( float ) ( ( double ) fmin_sel ( fmin_sel ( fmin_sel ( b.min.y, t.a.y ), t.b.y ), t.c.y ) - eps );
    b.min.z = ( float ) ( ( double ) fmin_sel ( fmin_sel ( fmin_sel ( b.min.z, t.a.z ), t.b.z ), t.c.z ) - eps );
    b.max.x = ( float ) ( ( double ) fmax_sel ( fmax_sel ( fmax_sel ( b.max.x, t.a.x ), t.b.x ), t.c.x ) + eps );
    b.max.y = ( float ) ( ( double ) fmax_sel ( fmax_sel ( fmax_sel ( b.max.y, t.a.y ), t.b.y ), t.c.y ) + eps );
    b.max.z = ( float ) ( ( double ) fmax_sel ( fmax_sel ( fmax_sel ( b.max.z, t.a.z ), t.b.z ), t.c.z ) + eps );
}
static void grow_by_box ( TerraAABB& b, const TerraAABB& o ) {
    const double eps = 1e-4;
    b.min.x = fmin_sel ( b.min.x, o.min.x ); b.min.y = fmin_sel ( b.min.y, o.min.y ); b.min.z = fmin_sel ( b.min.z, o.min.z );
    b.max.x = ( float ) ( ( double ) fmax_sel ( b.max.x, o.max.x ) + eps );
    b.max.y = ( float ) ( ( double ) fmax_sel ( b.max.y, o.max.y ) + eps );
    b.max.z = ( float ) ( ( double ) fmax_sel ( b.max.z, o.max.z ) + eps );
}
static float area ( const TerraAABB& b ) {
    float w = b.max.x - b.min.x, h = b.max.y - b.min.y, d = b.max.z - b.min.z;
    return 2 * ( w * d + w * h + d * h );
}
static float centre_x ( const TerraAABB& b ) { return ( b.min.x + b.max.x ) / 2; }

void build ( const TerraObject* objects, size_t nobj, std::vector<HostNode>& nodes, int& max_stack ) {
    size_t n = 0;
    for ( size_t j = 0; j < nobj; ++j ) n += objects[j].triangles_count;
    std::vector<Volume> vol ( n );
    TerraAABB scene_box = empty_box();
    size_t p = 0;
    for ( size_t j = 0; j < nobj; ++j ) for ( size_t i = 0; i < objects[j].triangles_count; ++i, ++p ) {
        vol[p].box = empty_box();
        grow_by_triangle ( scene_box, objects[j].triangles[i] );     // the scene box shrinks/grows by eps per triangle, as in the reference
        grow_by_triangle ( vol[p].box, objects[j].triangles[i] );
        vol[p].index = ( uint32_t ) ( ( int ) j | ( ( int ) i << 8 ) );
    }
    nodes.assign ( n > 1 ? n - 1 : 1, HostNode() );
    memset ( nodes.data(), 0, nodes.size() * sizeof ( HostNode ) );
    max_stack = 1;
    if ( n < 2 ) {      // the reference cannot build these; emit a root with one (or no) leaf
        nodes[0].type[0] = n == 1 ? 1 : 0; nodes[0].type[1] = 0;
        if ( n == 1 ) { nodes[0].aabb[0] = vol[0].box; nodes[0].index[0] = ( int32_t ) vol[0].index; }
        return;
    }
    std::stable_sort ( vol.begin(), vol.end(), [] ( const Volume & l, const Volume & r ) { return centre_x ( l.box ) > centre_x ( r.box ); } );
    struct Task { int start, end, node; TerraAABB container; };
    std::vector<Task> todo;
    std::vector<float> la ( n ), ra ( n );
    todo.push_back ( { 0, ( int ) n, 0, scene_box } );
    int next_node = 1;
    while ( !todo.empty() ) {
        Task t = todo.back(); todo.pop_back();
        const int cnt = t.end - t.start;
        const Volume* v = vol.data() + t.start;
        const float container_area = area ( t.container );
        TerraAABB acc = empty_box();
        for ( int i = 0; i < cnt; ++i ) { grow_by_box ( acc, v[i].box ); la[i] = area ( acc ); }
        acc = empty_box();
        for ( int i = cnt - 1; i >= 0; --i ) { grow_by_box ( acc, v[i].box ); ra[i] = area ( acc ); }
        float best_cost = FLT_MAX; int best = -1;
        for ( int i = 0; i < cnt; ++i ) {
            const int lc = i + 1, rc = cnt - lc;
            float cost = lc * la[i] / container_area + rc * ra[i] / container_area;
            if ( cost < best_cost ) { best_cost = cost; best = i; }
        }
        if ( best < 0 ) best = 0;
        if ( best > cnt - 2 ) best = cnt - 2;
        const int split = best + t.start;
        HostNode& nd = nodes[t.node];
        if ( split == t.start ) {
            nd.type[0] = 1; nd.aabb[0] = vol[t.start].box; nd.index[0] = ( int32_t ) vol[t.start].index;
        } else {
            TerraAABB b = empty_box();
            for ( int i = t.start; i <= split; ++i ) grow_by_box ( b, vol[i].box );
            nd.type[0] = -1; nd.aabb[0] = b; nd.index[0] = next_node;
            todo.push_back ( { t.start, split + 1, next_node, b } );
            ++next_node;
        }
        if ( split == t.end - 2 ) {
            nd.type[1] = 1; nd.aabb[1] = vol[t.end - 1].box; nd.index[1] = ( int32_t ) vol[t.end - 1].index;
        } else {
            TerraAABB b = empty_box();
            for ( int i = split + 1; i < t.end; ++i ) grow_by_box ( b, vol[i].box );
            nd.type[1] = -1; nd.aabb[1] = b; nd.index[1] = next_node;
            todo.push_back ( { split + 1, t.end, next_node, b } );
            ++next_node;
        }
    }
    nodes.resize ( ( size_t ) next_node );
    // stack entries a ray can need: replay the traversal's push/pop order with every box hit
    std::vector<int> st; st.reserve ( 64 ); st.push_back ( 0 );
    while ( !st.empty() ) {
        const HostNode& nd = nodes[ ( size_t ) st.back()]; st.pop_back();
        for ( int i = 0; i < 2; ++i ) if ( nd.type[i] == -1 ) { st.push_back ( nd.index[i] ); max_stack = std::max ( max_stack, ( int ) st.size() ); }
    }
}
} // namespace bvh

// ---- fast tree: 3-axis binned SAH, BVH2, leaves of <= 4 triangles ----------
// Built over the same per-triangle boxes as the reference (triangle bounds +- 1e-4) so every
// triangle a ray can hit lies inside its ancestors' boxes; inner boxes are plain unions.
#ifndef TERRA_FAST_PREFIX_NODES  // nodes of the fast tree's top levels numbered first: the levels every ray visits share a few cache lines
#define TERRA_FAST_PREFIX_NODES 64
#endif
#ifndef TERRA_FAST_LEAF_MAX      // triangles per leaf of the fast tree (the leaf word holds count-1 in 4 bits)
#define TERRA_FAST_LEAF_MAX 4
#endif
#ifndef TERRA_FAST_SAH_TRI_COST
#define TERRA_FAST_SAH_TRI_COST 1.0f      // measured, not derived: hall / sphere scene render 81.4 / 65.5 ms at 1.0, 82.1 / 66.7 at 3.0 (profiles/r02_measurements/ab_fast_tree_loop.log)
#endif
#ifndef TERRA_FAST_BINS          // bins of the binned surface-area split (ranges above TERRA_FAST_SWEEP_MAX triangles)
#define TERRA_FAST_BINS 32
#endif
#ifndef TERRA_FAST_SWEEP_MAX     // ranges of at most this many triangles are split by the exact sweep (0: bins all the way down)
#define TERRA_FAST_SWEEP_MAX 1024
#endif
namespace fastbvh {

static inline void grow ( TerraAABB& b, const TerraAABB& o ) {
    b.min.x = std::min ( b.min.x, o.min.x ); b.min.y = std::min ( b.min.y, o.min.y ); b.min.z = std::min ( b.min.z, o.min.z );
    b.max.x = std::max ( b.max.x, o.max.x ); b.max.y = std::max ( b.max.y, o.max.y ); b.max.z = std::max ( b.max.z, o.max.z );
}
static inline float half_area ( const TerraAABB& b ) {
    float w = b.max.x - b.min.x, h = b.max.y - b.min.y, d = b.max.z - b.min.z;
    return w * h + h * d + d * w;
}
static TerraAABB empty() { return bvh::empty_box(); }

// returns the child word for the range [lo, hi) of prims, appending nodes as needed
static uint32_t build_range ( std::vector<Prim>& prims, int lo, int hi, Built& out, int depth, int& max_depth );

static uint32_t make_leaf ( int lo, int hi ) { return DEV_CHILD_LEAF | ( ( uint32_t ) ( hi - lo - 1 ) << 27 ) | ( uint32_t ) lo; }

static void set_child ( DevNode& n, int k, const TerraAABB& b, uint32_t word ) {
    float* mn = k == 0 ? n.min0 : n.min1; float* mx = k == 0 ? n.max0 : n.max1;
    mn[0] = b.min.x; mn[1] = b.min.y; mn[2] = b.min.z; mx[0] = b.max.x; mx[1] = b.max.y; mx[2] = b.max.z;
    n.child[k] = word; n.prim[k] = 0;
}

Built build ( std::vector<Prim>& prims ) {
    Built out;
    const int n = ( int ) prims.size();
    out.nodes.reserve ( ( size_t ) std::max ( 1, n ) );
    out.nodes.push_back ( DevNode() );
    memset ( &out.nodes[0], 0, sizeof ( DevNode ) );
    struct Task { int lo, hi, node, slot, depth; };
    // root node holds the whole scene as (child0 = everything, child1 = empty) unless it splits
    std::vector<Task> todo;
    int max_depth = 1;
    auto bounds = [&] ( int lo, int hi ) { TerraAABB b = empty(); for ( int i = lo; i < hi; ++i ) grow ( b, prims[i].box ); return b; };
    auto split = [&] ( int lo, int hi, int& mid, bool refine ) -> bool {
        const int cnt = hi - lo;
        if ( cnt <= 1 ) return false;
        if ( cnt <= TERRA_FAST_LEAF_MAX ) {
            if ( !refine ) return false;
            // a range that may become a leaf: exact sweep over the three axes, split only if the surface-area estimate says the extra
            // node step is cheaper than the triangle steps it saves (TERRA_FAST_SAH_TRI_COST = cost of a triangle step in node steps)
            const float pa = half_area ( bounds ( lo, hi ) );
            float best = ( float ) cnt * TERRA_FAST_SAH_TRI_COST; int best_axis = -1, best_k = 0;
            for ( int a = 0; a < 3 && pa > 0.f; ++a ) {
                int idx[TERRA_FAST_LEAF_MAX];
                for ( int i = 0; i < cnt; ++i ) idx[i] = lo + i;
                std::sort ( idx, idx + cnt, [&] ( int x, int y ) { return prims[x].c[a] < prims[y].c[a] || ( prims[x].c[a] == prims[y].c[a] && prims[x].soup < prims[y].soup ); } );
                for ( int k = 1; k < cnt; ++k ) {
                    TerraAABB l = empty(), r = empty();
                    for ( int i = 0; i < k; ++i ) grow ( l, prims[idx[i]].box );
                    for ( int i = k; i < cnt; ++i ) grow ( r, prims[idx[i]].box );
                    const float cost = 1.f + ( half_area ( l ) * ( float ) k + half_area ( r ) * ( float ) ( cnt - k ) ) / pa * TERRA_FAST_SAH_TRI_COST;
                    if ( cost < best ) { best = cost; best_axis = a; best_k = k; }
                }
            }
            if ( best_axis < 0 ) return false;
            const int a = best_axis;
            std::sort ( prims.begin() + lo, prims.begin() + hi, [&] ( const Prim & x, const Prim & y ) { return x.c[a] < y.c[a] || ( x.c[a] == y.c[a] && x.soup < y.soup ); } );
            mid = lo + best_k;
            return true;
        }
        if ( cnt <= TERRA_FAST_SWEEP_MAX ) {
            // a small range: the exact sweep over the three axes (every split position between two neighbours in centroid order) instead of the bins
            std::vector<int> idx ( ( size_t ) cnt ); std::vector<float> ra ( ( size_t ) cnt );
            float best_cost = FLT_MAX; int best_axis = -1, best_k = 0;
            for ( int a = 0; a < 3; ++a ) {
                for ( int i = 0; i < cnt; ++i ) idx[ ( size_t ) i] = lo + i;
                std::sort ( idx.begin(), idx.end(), [&] ( int x, int y ) { return prims[x].c[a] < prims[y].c[a] || ( prims[x].c[a] == prims[y].c[a] && prims[x].soup < prims[y].soup ); } );
                TerraAABB acc = empty();
                for ( int i = cnt - 1; i > 0; --i ) { grow ( acc, prims[idx[ ( size_t ) i]].box ); ra[ ( size_t ) i] = half_area ( acc ); }
                acc = empty();
                for ( int k = 1; k < cnt; ++k ) {
                    grow ( acc, prims[idx[ ( size_t ) k - 1]].box );
                    const float cost = half_area ( acc ) * ( float ) k + ra[ ( size_t ) k] * ( float ) ( cnt - k );
                    if ( cost < best_cost ) { best_cost = cost; best_axis = a; best_k = k; }
                }
            }
            if ( best_axis < 0 ) { mid = lo + cnt / 2; return true; }
            const int a = best_axis;
            std::sort ( prims.begin() + lo, prims.begin() + hi, [&] ( const Prim & x, const Prim & y ) { return x.c[a] < y.c[a] || ( x.c[a] == y.c[a] && x.soup < y.soup ); } );
            mid = lo + best_k;
            return true;
        }
        float cmin[3] = { FLT_MAX, FLT_MAX, FLT_MAX }, cmax[3] = { -FLT_MAX, -FLT_MAX, -FLT_MAX };
        for ( int i = lo; i < hi; ++i ) for ( int a = 0; a < 3; ++a ) { cmin[a] = std::min ( cmin[a], prims[i].c[a] ); cmax[a] = std::max ( cmax[a], prims[i].c[a] ); }
        const int B = TERRA_FAST_BINS;
        float best_cost = FLT_MAX; int best_axis = -1, best_bin = -1;
        for ( int a = 0; a < 3; ++a ) {
            float ext = cmax[a] - cmin[a];
            if ( ! ( ext > 0.f ) ) continue;
            TerraAABB bb[B]; int bc[B];
            for ( int b = 0; b < B; ++b ) { bb[b] = empty(); bc[b] = 0; }
            const float scale = ( float ) B / ext;
            for ( int i = lo; i < hi; ++i ) { int b = std::min ( B - 1, std::max ( 0, ( int ) ( ( prims[i].c[a] - cmin[a] ) * scale ) ) ); grow ( bb[b], prims[i].box ); ++bc[b]; }
            float la[B], ra[B]; int lc[B], rc[B];
            TerraAABB acc = empty(); int cacc = 0;
            for ( int b = 0; b < B; ++b ) { grow ( acc, bb[b] ); cacc += bc[b]; la[b] = cacc ? half_area ( acc ) : 0.f; lc[b] = cacc; }
            acc = empty(); cacc = 0;
            for ( int b = B - 1; b >= 0; --b ) { grow ( acc, bb[b] ); cacc += bc[b]; ra[b] = cacc ? half_area ( acc ) : 0.f; rc[b] = cacc; }
            for ( int b = 0; b < B - 1; ++b ) {
                if ( lc[b] == 0 || rc[b + 1] == 0 ) continue;
                float cost = la[b] * ( float ) lc[b] + ra[b + 1] * ( float ) rc[b + 1];
                if ( cost < best_cost ) { best_cost = cost; best_axis = a; best_bin = b; }
            }
        }
        if ( best_axis < 0 ) {      // all centroids coincide: split in the middle
            mid = lo + cnt / 2;
            return true;
        }
        const float ext = cmax[best_axis] - cmin[best_axis], scale = ( float ) B / ext, c0 = cmin[best_axis];
        const int a = best_axis, bsel = best_bin;
        auto it = std::partition ( prims.begin() + lo, prims.begin() + hi, [&] ( const Prim & p ) {
            int b = std::min ( B - 1, std::max ( 0, ( int ) ( ( p.c[a] - c0 ) * scale ) ) ); return b <= bsel; } );
        mid = ( int ) ( it - prims.begin() );
        if ( mid == lo || mid == hi ) mid = lo + cnt / 2;
        return true;
    };
    if ( n == 0 ) { out.nodes[0].child[0] = DEV_CHILD_EMPTY; out.nodes[0].child[1] = DEV_CHILD_EMPTY; out.max_stack = 1; return out; }
    int mid = 0;
    if ( !split ( 0, n, mid, false ) ) {
        set_child ( out.nodes[0], 0, bounds ( 0, n ), make_leaf ( 0, n ) );
        out.nodes[0].child[1] = DEV_CHILD_EMPTY;
        out.max_stack = 1;
    } else {
        // pass 1: ranges of more than TERRA_FAST_LEAF_MAX triangles are split, smaller ones become leaves; pass 2 splits those leaves
        // further where the surface-area estimate pays -- but never below the depth pass 1 reached, because the traversal stack
        // (one KB of LDS per entry and block) is sized by the depth and the kernel's occupancy hangs on it
        std::vector<Task> leaves;
        // Subtrees are independent (disjoint triangle ranges, disjoint child slots), so large scenes are built by several threads:
        // a shared list hands out tasks, a thread keeps a subtree to itself once its range is below 8192 triangles. Nodes come from
        // a preallocated array through an atomic counter (a binary tree over n triangles has fewer than n inner nodes); which thread
        // built what does not show in the result, because the array is renumbered depth first below.
        out.nodes.resize ( ( size_t ) n + 1 + 2 * 64 * 16 );
        std::atomic<uint32_t> next_node ( 1 );
        std::atomic<int> depth_seen ( max_depth );
        int n_threads = 1;
        if ( n >= 20000 ) {
            cpu_set_t set; CPU_ZERO ( &set );
            int cpus = sched_getaffinity ( 0, sizeof set, &set ) == 0 ? CPU_COUNT ( &set ) : ( int ) std::thread::hardware_concurrency();
            if ( const int asked = terra_build_threads() ) cpus = asked;          // terra_amd_set_build_threads
            n_threads = std::max ( 1, std::min ( cpus, 16 ) );
        }
        auto run = [&] ( bool refine, int depth_cap ) {
            std::mutex mu; std::condition_variable cv; int busy = 0;
            std::vector<std::vector<Task>> kept ( ( size_t ) n_threads );
            auto worker = [&] ( int tid ) {
                std::vector<Task> local, mine;
                uint32_t chunk_next = 0, chunk_end = 0;          // node indices are taken 64 at a time, so threads do not write to neighbouring cache lines
                for ( ;; ) {
                    {
                        std::unique_lock<std::mutex> lk ( mu );
                        cv.wait ( lk, [&] { return !todo.empty() || busy == 0; } );
                        if ( todo.empty() ) { kept[ ( size_t ) tid].swap ( mine ); return; }
                        // one big range at a time, or a batch of small ones (pass 2 hands out tens of thousands of 2-4 triangle ranges)
                        do { local.push_back ( todo.back() ); todo.pop_back(); } while ( !todo.empty() && local.size() < 512 && local.back().hi - local.back().lo <= 64 && todo.back().hi - todo.back().lo <= 64 );
                        ++busy;
                    }
                    int deepest = 0;
                    while ( !local.empty() ) {
                        Task t = local.back(); local.pop_back();
                        deepest = std::max ( deepest, t.depth );
                        TerraAABB b = bounds ( t.lo, t.hi );
                        int m = 0;
                        if ( ( refine && t.depth >= depth_cap ) || !split ( t.lo, t.hi, m, refine ) ) {
                            set_child ( out.nodes[ ( size_t ) t.node], t.slot, b, make_leaf ( t.lo, t.hi ) );
                            if ( !refine && t.hi - t.lo > 1 ) mine.push_back ( t );
                            continue;
                        }
                        if ( chunk_next == chunk_end ) { chunk_next = next_node.fetch_add ( 64 ); chunk_end = chunk_next + 64; }
                        const uint32_t idx = chunk_next++;
                        memset ( &out.nodes[idx], 0, sizeof ( DevNode ) );
                        set_child ( out.nodes[ ( size_t ) t.node], t.slot, b, idx );
                        const Task l = { t.lo, m, ( int ) idx, 0, t.depth + 1 }, r = { m, t.hi, ( int ) idx, 1, t.depth + 1 };
                        if ( n_threads > 1 && t.hi - t.lo > 8192 ) {
                            { std::lock_guard<std::mutex> lk ( mu ); todo.push_back ( l ); todo.push_back ( r ); }
                            cv.notify_all();
                        } else { local.push_back ( l ); local.push_back ( r ); }
                    }
                    int seen = depth_seen.load();
                    while ( deepest > seen && !depth_seen.compare_exchange_weak ( seen, deepest ) ) {}
                    { std::lock_guard<std::mutex> lk ( mu ); --busy; }
                    cv.notify_all();
                }
            };
            std::vector<std::thread> pool;
            for ( int k = 1; k < n_threads; ++k ) {
                try { pool.emplace_back ( worker, k ); }
                catch ( ... ) { break; }                      // no more threads to be had: the ones running (at least this one) share the list
            }
            worker ( 0 );
            for ( std::thread& th : pool ) th.join();
            for ( const std::vector<Task>& v : kept ) leaves.insert ( leaves.end(), v.begin(), v.end() );
            max_depth = depth_seen.load();
        };
        todo.push_back ( { 0, mid, 0, 0, 1 } );
        todo.push_back ( { mid, n, 0, 1, 1 } );
        const bool timing = terra_commit_timing_on();
        auto now = [] { return std::chrono::duration<double> ( std::chrono::steady_clock::now().time_since_epoch() ).count(); };
        double t0 = now();
        run ( false, 0 );
        const int depth_cap = max_depth;
        if ( timing ) { fprintf ( stderr, "[terra_amd timing]   fast tree pass 1 (binned splits)  %8.2f ms on %d threads, %zu leaves to refine\n", ( now() - t0 ) * 1e3, n_threads, leaves.size() ); t0 = now(); }
        todo.swap ( leaves );
        run ( true, depth_cap );
        if ( timing ) fprintf ( stderr, "[terra_amd timing]   fast tree pass 2 (leaf refinement) %8.2f ms\n", ( now() - t0 ) * 1e3 );
        out.nodes.resize ( next_node.load() );
        // ordered traversal with the near child kept in a register: one pending (far) child per level, plus the node in hand when a lane leaves the loop
        out.max_stack = max_depth + 1;
    }
    // Numbering: the first TERRA_FAST_PREFIX_NODES nodes in breadth-first order (the levels every ray visits, packed into 4 KB), all others depth first (a node is followed by one child's whole subtree: a descent finds its next
    // nodes close by; measured on the 97k-triangle hall, a fully breadth-first array renders 3.7 % slower).
    {
        const size_t K = std::min ( out.nodes.size(), ( size_t ) TERRA_FAST_PREFIX_NODES );
        std::vector<uint32_t> order, newidx ( out.nodes.size(), 0xffffffffu );
        order.reserve ( out.nodes.size() ); order.push_back ( 0 ); newidx[0] = 0;
        for ( size_t head = 0; head < order.size() && order.size() < K; ++head ) {
            const DevNode& nd = out.nodes[order[head]];
            for ( int k = 0; k < 2 && order.size() < K; ++k ) if ( nd.child[k] != DEV_CHILD_EMPTY && ! ( nd.child[k] & DEV_CHILD_LEAF ) ) { newidx[nd.child[k]] = ( uint32_t ) order.size(); order.push_back ( nd.child[k] ); }
        }
        {   // the rest depth first (a node is followed by one child's whole subtree), whichever pass created it
            std::vector<uint32_t> st; st.push_back ( 0 );
            while ( !st.empty() ) {
                const uint32_t i = st.back(); st.pop_back();
                if ( newidx[i] == 0xffffffffu ) { newidx[i] = ( uint32_t ) order.size(); order.push_back ( i ); }
                const DevNode& nd = out.nodes[i];
                for ( int k = 0; k < 2; ++k ) if ( nd.child[k] != DEV_CHILD_EMPTY && ! ( nd.child[k] & DEV_CHILD_LEAF ) ) st.push_back ( nd.child[k] );
            }
        }
        std::vector<DevNode> renum ( order.size() );
        for ( size_t k = 0; k < order.size(); ++k ) {
            renum[k] = out.nodes[order[k]];
            for ( int c = 0; c < 2; ++c ) if ( renum[k].child[c] != DEV_CHILD_EMPTY && ! ( renum[k].child[c] & DEV_CHILD_LEAF ) ) renum[k].child[c] = newidx[renum[k].child[c]];
        }
        out.nodes.swap ( renum );
    }
    out.order.resize ( ( size_t ) n );
    for ( int i = 0; i < n; ++i ) out.order[ ( size_t ) i] = prims[ ( size_t ) i].soup;
    return out;
}

// ---- binary16 with directed rounding ---------------------------------------------------------------------------------------------
// x -> binary16 bits, rounded towards -inf (up = false) or +inf (up = true): integer arithmetic on the double's bits, no dependence on the host's half support.
// Magnitudes beyond 65504 give +-65504 when rounding towards zero and +-inf when rounding away (a wider box, never a narrower one); NaN gives the widest value too.
uint16_t half_outward ( double x, bool up ) {
    uint64_t u; memcpy ( &u, &x, 8 );
    const bool neg = ( u >> 63 ) != 0;
    const uint16_t sign = neg ? 0x8000u : 0u;
    const int e11 = ( int ) ( ( u >> 52 ) & 0x7ff );
    const uint64_t m52 = u & ( ( 1ull << 52 ) - 1 );
    const bool away = neg ? !up : up;                                       // the magnitude is rounded away from zero
    if ( e11 == 0x7ff ) return m52 ? ( up ? 0x7c00u : 0xfc00u ) : ( uint16_t ) ( sign | 0x7c00u );      // NaN: +inf as an upper bound, -inf as a lower one; inf: itself
    if ( e11 == 0 ) return ( m52 && away ) ? ( uint16_t ) ( sign | 1u ) : sign;                          // zero / double subnormal (below every binary16 subnormal)
    const int e = e11 - 1023 + 15;                                          // binary16 biased exponent
    if ( e >= 31 ) return ( uint16_t ) ( sign | ( away ? 0x7c00u : 0x7bffu ) );
    uint32_t h; bool inexact;
    if ( e >= 1 ) { h = ( ( uint32_t ) e << 10 ) | ( uint32_t ) ( m52 >> 42 ); inexact = ( m52 & ( ( 1ull << 42 ) - 1 ) ) != 0; }
    else {
        const int sh = 42 + ( 1 - e );                                      // binary16 subnormal: shift the significand (with its leading 1) further right
        const uint64_t full = ( 1ull << 52 ) | m52;
        if ( sh >= 64 ) { h = 0; inexact = true; }
        else { h = ( uint32_t ) ( full >> sh ); inexact = ( full & ( ( 1ull << sh ) - 1 ) ) != 0; }
    }
    if ( inexact && away ) ++h;                                             // (a carry out of the mantissa moves into the exponent: the next power of two, or inf)
    return ( uint16_t ) ( sign | h );
}

// ---- the binary tree as 4-wide binary16 nodes -------------------------------------------------------------------------------------
// child k's planes on axis a, for both ray directions (dev_types.h DevFastNode)
static inline void set_planes ( DevFastNode& nd, int k, int a, uint32_t lo, uint32_t hi ) { nd.q[a][0][k] = lo | ( hi << 16 ); nd.q[a][1][k] = hi | ( lo << 16 ); }
Wide widen ( const std::vector<DevNode>& n2, float scale ) {
    Wide out;
    struct Slot { float mn[3], mx[3]; uint32_t word; };
    auto slot_of = [&] ( const DevNode & nd, int k ) { Slot s; memcpy ( s.mn, k ? nd.min1 : nd.min0, 12 ); memcpy ( s.mx, k ? nd.max1 : nd.max0, 12 ); s.word = nd.child[k]; return s; };
    auto inner = [] ( uint32_t w ) { return w != DEV_CHILD_EMPTY && ! ( w & DEV_CHILD_LEAF ); };
    auto area = [] ( const Slot & s ) { const float w = s.mx[0] - s.mn[0], h = s.mx[1] - s.mn[1], d = s.mx[2] - s.mn[2]; return w * h + h * d + d * w; };
    if ( n2.empty() ) return out;
    // wide node i is made from binary node src[i]; children get consecutive indices when their parent is made (siblings share cache lines), subtrees follow depth first
    std::vector<uint32_t> src; src.push_back ( 0 );
    std::vector<uint32_t> todo; todo.push_back ( 0 );
    std::vector<int> kids;                       // per wide node: children in use (for the stack bound below)
    out.nodes.push_back ( DevFastNode() ); kids.push_back ( 0 );
    while ( !todo.empty() ) {
        const uint32_t wi = todo.back(); todo.pop_back();
        Slot slots[4]; int n = 0;
        for ( int k = 0; k < 2; ++k ) if ( n2[src[wi]].child[k] != DEV_CHILD_EMPTY ) slots[n++] = slot_of ( n2[src[wi]], k );
        while ( n < 4 ) {
            int pick = -1; float best = -1.f;
            for ( int k = 0; k < n; ++k ) if ( inner ( slots[k].word ) ) { const DevNode& c = n2[slots[k].word]; if ( c.child[0] == DEV_CHILD_EMPTY || c.child[1] == DEV_CHILD_EMPTY ) continue; const float a = area ( slots[k] ); if ( a > best ) { best = a; pick = k; } }
            if ( pick < 0 ) break;
            const DevNode& c = n2[slots[pick].word];
            slots[pick] = slot_of ( c, 0 ); slots[n++] = slot_of ( c, 1 );
        }
        DevFastNode nd; memset ( &nd, 0, sizeof nd );
        for ( int k = 0; k < 4; ++k ) {
            if ( k >= n ) { for ( int a = 0; a < 3; ++a ) set_planes ( nd, k, a, 0x7bffu, 0xfbffu ); nd.child[k] = DEV_CHILD_EMPTY; continue; }      // an empty slot: min = +65504, max = -65504 -- no ray enters it
            for ( int a = 0; a < 3; ++a ) {
                const bool empty = ! ( slots[k].mn[a] <= slots[k].mx[a] );
                const uint32_t lo = empty ? 0x7bffu : half_outward ( ( double ) slots[k].mn[a] * ( double ) scale, false ), hi = empty ? 0xfbffu : half_outward ( ( double ) slots[k].mx[a] * ( double ) scale, true );
                set_planes ( nd, k, a, lo, hi );
            }
            if ( inner ( slots[k].word ) ) { nd.child[k] = ( uint32_t ) src.size(); src.push_back ( slots[k].word ); out.nodes.push_back ( DevFastNode() ); kids.push_back ( 0 ); }
            else nd.child[k] = slots[k].word;
        }
        kids[wi] = n;
        out.nodes[wi] = nd;
        for ( int k = n - 1; k >= 0; --k ) if ( ! ( nd.child[k] & DEV_CHILD_LEAF ) ) todo.push_back ( nd.child[k] );
    }
    // stack entries a ray can need: a node with k children in use leaves up to k - 1 of them pending while the traversal is below one of them; + the root's own entry
    // and the node a lane has in hand when it leaves the loop. Children have larger indices than their parent, so one backward sweep does it.
    std::vector<int> need ( out.nodes.size(), 0 );
    for ( size_t i = out.nodes.size(); i-- > 0; ) {
        int below = 0;
        for ( int k = 0; k < 4; ++k ) { const uint32_t w = out.nodes[i].child[k]; if ( w != DEV_CHILD_EMPTY && ! ( w & DEV_CHILD_LEAF ) ) below = std::max ( below, need[w] ); }
        need[i] = std::max ( 0, kids[i] - 1 ) + below;
    }
    out.max_stack = need[0] + 2;
    return out;
}
} // namespace fastbvh
