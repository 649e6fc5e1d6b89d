// dev_types.h -- plain-data layouts shared by the host side (scene_host.cpp) and
// the HIP kernels. Everything here is what actually sits in HBM / kernarg.
//
// HBM layout of a committed scene (one replica per GPU):
//   nodes   : DevNode[n_nodes]     64 B each, same size as the reference node
//             (reference src/TerraBVH.h:13-17) but with the two children packed as
//             4 x 16-byte quads so a lane fetches a node with four dwordx4 loads.
//   tris    : DevTri[n_tris]       48 B each: 36 B of vertex data (reference
//             include/Terra.h:109-113) + the (object, triangle) reference in the
//             pad lanes, so a leaf test is three aligned dwordx4 loads.
//   props   : DevProps[n_tris]     64 B each: 60 B of vertex normals/texcoords
//             (reference include/Terra.h:115-122) + 4 B pad.
//   mats    : DevMaterial[n_objects]
//   lights  : DevLight[n_lights], tri_area : float[n_tris] (only light triangles are read)
// The "algorithmic bytes" of the roofline use the reference sizes (64/36/60),
// not the padded ones (DESIGN.md "Roofline").
#pragma once
#include <stdint.h>
#include <hip/hip_vector_types.h>     // float4 (DevRenderParams::partials)

#define TERRA_DEV_MAX_ATTR 8

struct DevF3 { float x, y, z; };
struct DevF4 { float x, y, z, w; };

// child word: bit 31 clear -> inner node index; bit 31 set -> leaf, low 31 bits = global triangle index;
// 0xFFFFFFFF -> empty slot (only in scenes with < 2 triangles)
#define DEV_CHILD_LEAF  0x80000000u
#define DEV_CHILD_EMPTY 0xFFFFFFFFu

struct DevNode {
    // q0 = min0.xyz, max0.x   q1 = max0.yz, min1.xy   q2 = min1.z, max1.xyz   q3 = child0, child1, prim0, prim1
    // prim = reference leaf index (object | triangle << 8), kept so results carry the reference's TerraPrimitiveRef
    float    min0[3], max0[3];
    float    min1[3], max1[3];
    uint32_t child[2];
    uint32_t prim[2];
};
static_assert ( sizeof ( DevNode ) == 64, "DevNode must be 64 bytes" );

// A fast-tree node as the kernels read it (trace_device.h "MODE 2"), 128 bytes = one cache line, FOUR children: per child and axis the two planes of the child's box
// as binary16 rounded outward (min down, max up), times DevScene's power-of-two scale, as one 32-bit word -- the four children's words of an axis side by side and
// every such 16-byte group twice: q[axis][0][child] = min | max << 16 for rays travelling in the axis' positive direction, q[axis][1][child] = max | min << 16 for the
// others, so that a ray LOADS the group whose low half is its near plane (no per-box swap) --, then the four child words (inner: index of a wide node; leaf:
// DEV_CHILD_LEAF | (count - 1) << 27 | first triangle; DEV_CHILD_EMPTY with an inverted box -- min = +65504, max = -65504 -- for a slot not in use). A ray reads four
// 16-byte pieces of a node (its three groups and the child words). Made on the host from the builders' binary tree (tree_build.cpp fastbvh::widen).
struct DevFastNode {
    uint32_t q[3][2][4];
    uint32_t child[4];
    uint32_t pad[4];
};
static_assert ( sizeof ( DevFastNode ) == 128, "DevFastNode must be 128 bytes" );

struct DevTri {
    float    a[3]; uint32_t object;
    float    b[3]; uint32_t tri_in_object;
    float    c[3]; uint32_t pad;
};
static_assert ( sizeof ( DevTri ) == 48, "DevTri must be 48 bytes" );

struct DevProps {
    float na[3], nb[3], nc[3];
    float ta[2], tb[2], tc[2];
    float pad;
};
static_assert ( sizeof ( DevProps ) == 64, "DevProps must be 64 bytes" );

enum DevBsdfKind { kDevBsdfDiffuse = 0, kDevBsdfPhong = 1, kDevBsdfGGX = 2, kDevBsdfGlass = 3 };

// A texture as the reference samples it (reference src/Terra.c:368-466): texel-space lookups,
// 1-byte (x/255) or float components, point or bilinear filter, wrap / mirror / clamp addressing.
struct DevTexture {
    const void* data;              // HBM copy of TerraTexture::pixels (+ 2 elements of padding)
    uint32_t width, height;
    uint32_t components;
    uint32_t depth;                // 1 or 4 bytes per component
    uint32_t filter;               // TerraFilter
    uint32_t address_mode;         // TerraTextureAddressMode
};

struct DevMaterial {
    int32_t  bsdf;                 // DevBsdfKind
    uint32_t attributes_count;
    float    ior;
    uint32_t first_tri;            // offset of the object's triangles in the soup
    float    emissive[3];
    uint32_t tri_count;
    float    attributes[TERRA_DEV_MAX_ATTR][3];
    // texture index per attribute slot (slot TERRA_DEV_MAX_ATTR = emissive), -1 = the constant above
    int32_t  tex[TERRA_DEV_MAX_ATTR + 1];
    int32_t  any_texture;
};

struct DevReplay {                 // one level of the reachability replay (DevScene::ref_replay)
    float    bmin[3];
    uint32_t parent;
    float    bmax[3];
    uint32_t pad;
};

struct DevLight {
    uint32_t object;
    uint32_t first_tri;
    uint32_t tri_count;
    float    area;
};

struct DevScene {
    const DevNode*     nodes;
    const DevTri*      tris;
    const DevProps*    props;
    const DevMaterial* mats;
    const DevLight*    lights;
    const float*       tri_area;
    const DevTexture*  textures;
    uint32_t n_nodes, n_tris, n_objects, n_lights;
    uint32_t lights_triangles_count;
    int32_t  max_stack;
    // optional second accelerator over the same triangles (terra_amd_set_tree_mode, DESIGN.md "Fast tree"):
    // 3-axis binned-SAH BVH2, leaves of up to 4 triangles; fast_tris is the soup in leaf order with
    // DevTri::pad = the triangle's rank in the REFERENCE tree's leaf visit order (the tie-break key)
    const DevNode*     fast_nodes;      // the device builder's output: a binary tree with (min, max) boxes (read back, checked and widened by the host; nullptr for host-built trees)
    const DevFastNode* fast_nodes_h;    // the tree as traversed: 4-wide nodes of binary16 planes (tree_build.cpp fastbvh::widen)
    const DevTri*      fast_tris;
    uint32_t n_fast_nodes;              // wide nodes
    int32_t  fast_max_stack;            // stack entries a ray can need in the wide tree
    float    fast_inv_scale;            // 1 / (the power of two DevFastNode's planes are stored times): the factor of the ray's inverse direction
    // scenes outside the coordinate range of the containment proof (DESIGN.md "Reachability"): the fast tree (boxes inflated to the rounding bound) finds the
    // candidates, and one is accepted only if the REFERENCE traversal would have reached it, i.e. if the slab tests of its ancestors in the reference tree pass.
    // ref_replay[q] (DevReplay, 32 B) = the box the reference tests before it visits inner node q -- stored in q's parent -- and that parent's index, so that one
    // 32-byte fetch gives a level's test AND the way up (node 0 = the root: never tested); fast_leaf_parent[i] = the reference node whose leaf child fast triangle i is
    const struct DevReplay* ref_replay;
    const uint32_t*    fast_leaf_parent;
    // fast_leaf_mask[i]: bit L set = the slab test of the L-th node on the way up from fast triangle i's leaf (L = 0: the leaf's node) has to be replayed for a ray that
    // hits the triangle. A clear bit = it cannot fail: either the box contains the triangle's extent with a clearance above the rounding bound (the containment
    // argument, at the scene's scale), or it contains the box of level L - 1 component by component -- then, for a ray whose inverse direction is finite and non-zero,
    // (b - o) * inv being monotone in b makes "level L - 1 passes" imply "level L passes" in the very same float arithmetic. Rays that are not regular replay every
    // level. Bit 31 stands for every level from 31 up.
    const uint32_t*    fast_leaf_mask;
    uint32_t           reach;
    // (cos, sin) of 2 * terra_PI * (k * 2^-24) for k = 0 .. 2^24 - 1: the azimuth the BSDF samplers make of a stream-B variate, tabulated once per device
    // (128 MB of HBM; trace_device.h azimuth_fetch). nullptr = compute.
    const float2*      sincos24;
    // environment lighting (terra_amd_set_environment_lighting; off = the reference's behaviour):
    // 0 off, 1 constant env_color, 2 lat-long lookup of textures[env_tex] by ray direction
    int32_t  env_mode;
    int32_t  env_tex;
    float    env_color[3];
    // environment importance sampling (terra_amd_set_environment_sampling; compiled into the KINDS & TERRA_KIND_SAMPLER variants only): the lat-long map textures[env_tex]
    // as a TerraDistribution2D (reference src/Terra.c:812-846) over luminance x sin(theta of the row): env_f / env_cdf = the env_nx x env_ny table and each row's
    // normalised running sum, env_row_f / env_row_cdf = the rows' totals and their normalised running sum, env_integral = the total. env_nx == 0: off.
    const float* env_f; const float* env_cdf; const float* env_row_f; const float* env_row_cdf;
    uint32_t env_nx, env_ny; float env_integral; uint32_t env_monotone;
};
// bit of the kernels' KINDS mask (bits 0-3 = DevBsdfKind present) that compiles the environment term in
#define TERRA_KIND_ENV 16
// ... and the bit that compiles textured-attribute sampling into terra_surface_init's counterpart
#define TERRA_KIND_TEX 32
// ... and the bit that compiles the sampler integration in (terra_amd_set_sampler_integration: the pixel's Halton / stratified sampler feeds the first bounce)
#define TERRA_KIND_SAMPLER 64

// indices into the device counter array (uint64 each); mirrors TerraAmdStats
enum { kCtrRays = 0, kCtrNodes, kCtrBoxTests, kCtrTriTests, kCtrHits, kCtrSamples, kCtrRandCalls, kCtrAttrFetches, kCtrPixels, kCtrLaunches, kCtrFaults, kCtrTriCulled,
       kCtrDbg0, kCtrDbgLast = kCtrDbg0 + 15, kCtrCount };   // kCtrFaults: only written by TERRA_CHECK_BOUNDS builds; kCtrDbg*: only by TERRA_PHASE_STATS builds (lane-occupancy study)

struct DevRenderParams {
    DevScene scene;
    // camera (reference src/Terra.c:1770-1799): rot rows, position, tan(fov/2) (host double tan, rounded), aspect
    float    cam_rot[9];
    float    cam_pos[3];
    float    tan_half_fov;
    float    aspect;
    float    jitter;
    float    exposure;
    float    gamma;
    uint32_t fb_w, fb_h;
    uint32_t x, y, w, h;            // rectangle to render
    // where pixel (px, py) of the frame lives in `pixels` / `results` / `rand_calls`: element (py - st_y) * st_pitch + (px - st_x).
    // Device-resident frames: st_x = st_y = 0, st_pitch = fb_w; terra_render() on a host frame stages the rectangle only.
    uint32_t st_x, st_y, st_pitch;
    uint32_t tile_size, rank, world; // sharding: tiles t with t % world == rank (world == 1: everything)
    uint32_t spp;                   // effective samples per pixel (after the stratified round-up)
    // sample split (terra_amd_set_sample_split): the call's spp samples are cut into `split` = 2^split_log2 chunks
    // of chunk_spp, one lane per (pixel, chunk); chunk j draws from the streams keyed (pixel, samples_so_far + j*chunk_spp)
    // and its radiance sum goes to partials[(j * blocks + block) * 256 + thread] = {sum.xyz, rand calls};
    // terra_resolve_kernel then adds the chunk sums to the pixel IN CHUNK ORDER -- exactly what `split`
    // successive calls of chunk_spp samples produce. split == 1: one chunk, the call's own sum (src/Terra.c:551-572).
    uint32_t split, split_log2, chunk_spp;
    float4*  partials;
    // job_streams[2 * job], [2 * job + 1]: the random streams of job `job`, keyed ahead of the render kernel by terra_job_streams_kernel with every lane busy -- {A.state, B.state},
    // {B.inc, px | py << 16 (all ones: a pixel outside the rectangle), samples already in the pixel} -- so that a lane at a job boundary loads 32 bytes instead of hashing its keys (ten 64-bit multiplications) nearly alone
    uint4*   job_streams;
    // job space of the persistent render grid (render_kernels.hip "jobs"): job_blocks virtual 256-thread blocks = (16x16 pixel blocks of the
    // shard) * split; *job_queue (zeroed before the launch) hands out the jobs beyond the ones the launched lanes start with
    uint32_t job_blocks;
    uint32_t* job_queue;
    // job order (launches that key their streams ahead, render_kernels.hip "job order"): the 16x16 pixel blocks of the launch are handed out in the order
    // block_order[0 .. blocks) -- the blocks some camera ray hits first, the blocks whose camera rays all leave the scene last, so that the launch ends on short jobs --
    // and block_order[blocks + b] is where pixel block b went (the resolve kernel finds the block's sums there). nullptr: the order of the numbering.
    const uint32_t* block_order;
    // the job decode's divisions by launch constants as multiplications: ceil(2^32 / d) (0 for d == 1) for d = (blocks per tile)^2, tiles per row of the
    // rectangle, blocks per tile row (the host refuses a launch in which some product n * d could reach 2^32)
    uint32_t job_div_bpt2, job_div_tiles_x, job_div_bpt, job_tiles_x;
    uint32_t bounces;
    int32_t  integrator;
    int32_t  tonemap;
    uint64_t frame_seed;
    float*   pixels;                // 3 floats per pixel
    void*    results;               // {float acc[3]; int samples} per pixel
    uint32_t* rand_calls;           // optional
    unsigned long long* counters;   // kCtrCount entries
    // LDS plan of a block (host decides, kernel obeys): stack_depth stack entries per thread,
    // lds_nodes nodes (breadth-first prefix) and lds_tris triangles (+ their vertex properties)
    // staged; leaf_cap deferred-leaf entries per thread; lds_mode 0 = nothing staged, 1 = whole scene
    uint32_t stack_depth, leaf_cap, lds_nodes, lds_tris;
    // fast-tree launches: entries of a lane's traversal stack beyond the stack_depth kept in LDS live in HBM -- stack_spill[(resident lane) * spill_cap + (entry - stack_depth)]
    // (the launch's scratch; nullptr when the whole stack fits in LDS). Rarely touched: the LDS part covers the depth a ray actually reaches, the bound is the tree's worst case.
    uint32_t* stack_spill; uint32_t spill_cap;
    int32_t  lds_mode;
    int32_t  count_level;           // work counters: 0 none (the default), 2 all six per lane (terra_amd_set_work_counters, or a per-pixel draw-count buffer was passed)
    uint32_t bsdf_kinds;            // mask of DevBsdfKind present in the scene (bit k = kind k)
    uint32_t leaf_cull;             // 1: skip the triangle test of a leaf child whose box the ray misses (Tracer::cull); host decides per call
    uint32_t fused_slab;            // 1: a cull launch inside the coordinate range: inner boxes may be decided with the fused slab arithmetic (Tracer::fused)
    // sampler integration (terra_amd_set_sampler_integration; compiled into the KINDS & TERRA_KIND_SAMPLER variants only): 0 off, 1 Halton, 2 stratified
    // (`sampler_strata` strata per dimension, 16 samples per stratum: the sampler the reference constructs at src/Terra.c:542)
    uint32_t sampler_mode, sampler_strata;
};
