/*
 * terra_amd.h -- C-ABI of libterra_amd.so beyond the drop-in Terra.h surface.
 *
 * Plain pointers and sizes only; no C++ or framework types cross this boundary.
 * Each entry point names the reference interface it replaces or serves
 * (paths relative to the reference tree). INTEGRATION.md shows the binding a
 * maintainer of the reference adds on their side.
 *
 * Conventions: functions returning int return 0 on success and a negative
 * TerraAmdStatus on failure; the message is kept per thread and read with
 * terra_amd_last_error(). Device pointers are raw HIP device addresses
 * (e.g. hipMalloc() results or a tensor's data pointer); `stream` is a
 * hipStream_t passed as void* (NULL = the null stream).
 */
#ifndef TERRA_AMD_H
#define TERRA_AMD_H

#include "Terra.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
    kTerraAmdOk               = 0,
    kTerraAmdErrNoDevice      = -1,  /* no gfx950 device / HIP runtime failure */
    kTerraAmdErrNotCommitted  = -2,  /* terra_scene_commit() has not run since the last change */
    kTerraAmdErrUnsupported   = -3,  /* material/attribute uses host function pointers the device cannot run */
    kTerraAmdErrBadArgument   = -4,
    kTerraAmdErrLaunch        = -5
} TerraAmdStatus;

/* Last error recorded on the calling thread ("" if none). terra_render() is
   void in the reference API (include/Terra.h:229), so this is its side channel. */
const char* terra_amd_last_error ( void );
void        terra_amd_clear_error ( void );
/* Process-wide companion: the FIRST error any thread has recorded since the last terra_amd_clear_first_error(), copied into buf
   (capacity bytes, always terminated); returns its TerraAmdStatus, 0 if there is none. For clients whose worker threads call
   terra_render() while another thread polls for completion, as the reference's does (satellite/src/Renderer.cpp:70-98,118-157):
   the per-thread channel above never shows a worker's failure to the polling thread. */
int         terra_amd_first_error ( char* buf, size_t capacity );
void        terra_amd_clear_first_error ( void );
/* Bytes of device staging memory the calling thread holds for terra_render() on host framebuffers: 28 B per pixel of the largest
   tile it has rendered (not of the frame). Released, with the thread's stream, when the thread exits. */
size_t      terra_amd_thread_staging_bytes ( void );

/* Process set-up a client may ask for, first thing in main() (before anything touches the GPU): ROCm maps a process's streams onto GPU_MAX_HW_QUEUES hardware
   queues -- 4 by default -- and kernels of streams that share a queue run one after the other; a client that calls terra_render() from 8 worker threads, as
   the reference's does (satellite/src/Renderer.cpp:70-98), wants 8. terra_amd_init() sets that variable unless the user already has; it has no effect once the
   HIP runtime is up. Optional: a client that does not call it gets ROCm's default. (The library never changes the environment on its own.) */
int  terra_amd_init ( void );
/* Process-wide switches of the commit path (all optional):
   terra_amd_set_commit_timing(1)  prints the phases of terra_scene_commit() on stderr;
   terra_amd_set_build_threads(n)  host threads of the fast tree's builder (0 = as many as the process may use, at most 16; the tree does not depend on it);
   terra_amd_set_azimuth_table(0)  scenes committed from now on compute the samplers' sin / cos instead of reading the 128 MB per-device table (same bits). */
void terra_amd_set_commit_timing ( int on );
int  terra_amd_set_build_threads ( int threads );
void terra_amd_set_azimuth_table ( int on );

/* Device selection for subsequent commits/renders issued by this thread's
   scenes: one device (one process per GPU is bench.py's layout, DESIGN.md "Multi-GPU"), or a set of devices driven from this process (below). */
int  terra_amd_device_count ( void );
int  terra_amd_set_device ( int device );
int  terra_amd_get_device ( void );

/* Several GPUs from ONE process (the layout of the reference's client: one process, tiles dealt to workers, satellite/src/Renderer.cpp:316-350).
   terra_amd_set_devices() names the devices every scene committed from now on is replicated on (no duplicates; count 0 = back to the one device
   of terra_amd_set_device); devices[0] is the primary device: it holds the staging frame and receives the gather. With more than one device
   in the set, terra_render() on a host framebuffer
     * shards a call that covers at least 256 x 256 pixels per device: device k renders the 64-pixel tiles t with terra_amd_shard_owner(t, N) == k,
       and one gather -- RCCL over xGMI, loaded at run time (librccl.so.1) and issued by this library -- brings the packed tiles to the primary
       device, which copies the rectangle to the host once;
     * sends a smaller call (the reference client's 128-pixel tiles) whole to one device: the calling thread's -- threads are dealt to the
       devices round-robin when they first call, so the client's eight workers drive eight GPUs.
   The framebuffer does not depend on the device set (same pixels, same sums, bit for bit). terra_amd_render_multi() is the sharded form
   called explicitly, for any rectangle and tile size (0 = 64) and any number of devices -- with ONE device in the set the same calls
   run (a communicator of one rank, the gather a copy inside the device), which is how a one-GPU box tests it. */
int  terra_amd_set_devices ( const int* devices, int count );
int  terra_amd_get_devices ( int* out, int capacity );              /* returns the number of devices in the set (>= 1) */
int  terra_amd_shard_owner ( size_t tile_index, int world );        /* the rank that renders tile `tile_index` (row-major in the rectangle) of `world` ranks */
int  terra_amd_render_multi ( const TerraCamera* camera, HTerraScene scene, const TerraFramebuffer* framebuffer,
                              size_t x, size_t y, size_t width, size_t height, size_t tile_size );
typedef struct {
    int      devices;               /* size of the set the scene was committed for */
    int      device[16];            /* ... its first 16 members, primary first */
    int      replicas;              /* device copies of the scene that exist (0: not committed / commit failed) */
    uint64_t gathers;               /* gathers this scene's multi-device renders have issued */
    uint64_t last_gather_bytes;     /* bytes the last one moved to the primary device (28 B per pixel of whole tiles, every rank's share) */
    uint64_t process_collectives;   /* RCCL group calls issued by this process */
    int      rccl_version;          /* ncclGetVersion() of the loaded library, 0 = not loaded (no multi-device render yet) */
    int      communicator_ranks;    /* ranks of the cached communicator */
    char     rccl_library[64];      /* the name it was loaded by */
    uint64_t rehearsed_gathers;     /* gathers of this process that did NOT go through RCCL: the stand-in of terra_amd_debug_replicas_share_device */
} TerraAmdMultiInfo;
int  terra_amd_multi_info ( HTerraScene scene, TerraAmdMultiInfo* out );
/* TEST HOOK, off by default (0): terra_amd_set_devices accepts a device listed more than once, so that a box with ONE GPU can run a scene with 2, 3, ... replicas --
   each with its own copy of the scene (pointers rebased), its own stream, staging frame and share of the tiles -- through terra_amd_render_multi / terra_render.
   RCCL admits one communicator rank per device, so the gather of such a set is a stand-in (one device-to-device copy per replica; TerraAmdMultiInfo::rehearsed_gathers
   counts them, process_collectives does not): what runs is everything around the transport, what does not is the transport between distinct devices. */
int  terra_amd_debug_replicas_share_device ( int on );

/* Can this library render the scene as it stands (objects added, options set; before or after terra_scene_commit)? 0 = yes; otherwise the TerraAmdStatus the commit
   would record, with the reason in `why` (capacity bytes, always terminated; may be NULL). The reference runs any host callback a material carries
   (TerraBSDF::sample / pdf / eval, TerraAttribute::eval: src/Terra.c:1071-1075, 1804-1810); the device runs the presets of TerraPresets.h and texture lookups only, and
   this library has no CPU path: a client that supports custom callbacks asks here first and keeps such scenes on the reference renderer. A pure query: nothing is
   recorded in the error channels. */
int  terra_amd_scene_supported ( HTerraScene scene, char* why, size_t capacity );

/* Frame seed F of the per-pixel random streams (DESIGN.md "Randomness"):
   replaces the reference's time(NULL)^&exit seed (src/Terra.c:679) and libc
   rand() (src/Terra.c:115). Default 0x5EED0001. */
void     terra_amd_set_frame_seed ( HTerraScene scene, uint64_t seed );
uint64_t terra_amd_get_frame_seed ( HTerraScene scene );

/* Traversal policy of subsequent commits (replaces src/TerraBVH.c:128-310 on the device).
   0 = replica: the reference's own tree (X-only sweep SAH, src/TerraBVH.c:79-244) traversed decision by decision as
       terra_bvh_traverse does (every leaf met is triangle-tested, nothing is culled).
   1 = fast tree, unconditionally: a 3-axis binned-SAH BVH2 over the same triangles with ordered, culled traversal; ties in
       depth are resolved by the reference tree's leaf visit order, so it selects the same triangle as mode 0 (DESIGN.md
       "Fast tree").
   2 = automatic (DEFAULT): at commit the scene is checked numerically (every coordinate within +-13 units, where the reference's
       1e-4 box margin provably exceeds the rounding error of the slab and triangle tests; every leaf / fast-tree box contains what
       it was built around). Scenes that pass run the reference tree with the leaf-box cull (a leaf's triangle is tested only
       if the ray passes that leaf's own box) when they fit in LDS, and the fast tree otherwise. Scenes beyond that range that do not fit
       in LDS keep the fast tree (host- or device-built) with its boxes inflated to the rounding bound, and a closest hit stands only if the
       reference traversal would have reached it (the slab tests of its ancestors in the reference tree are replayed; DESIGN.md "Reachability
       mode"); those that do fit keep the reference tree with the leaf-box cull on leaf boxes rebuilt with that same bound. The rest
       (non-finite or > 1e6 coordinates, trees too deep for the LDS stack) -- and calls whose camera lies outside
       TerraAmdTraversalInfo::camera_limit -- run as mode 0. All produce the reference's image bit for bit; only
       mode 0 also reproduces its work counters. terra_amd_traversal_info() reports the decision and the reason.
   Takes effect at the next terra_scene_commit(). */
int  terra_amd_set_tree_mode ( HTerraScene scene, int mode );
int  terra_amd_get_tree_mode ( HTerraScene scene );
typedef struct {
    int   tree_mode;                /* as set */
    int   fast_tree;                /* 1: the fast tree is used */
    int   fast_tree_built_on_device;/* 1: ... and was built on the GPU (terra_amd_set_tree_builder) */
    int   leaf_cull;                /* 1: reference tree with the leaf-box cull */
    int   lds_resident;             /* 1: the whole scene (nodes, triangles, vertex properties) is staged in LDS by every block */
    float max_coordinate;           /* largest |vertex coordinate| of the committed scene */
    float max_coordinate_allowed;   /* limit of the numeric containment check */
    char  note[256];                /* the reason, in words */
    int   last_call;                /* TerraAmdCallTraversal of the most recent render call of this scene (0: none yet). The commit-time decision above can be
                                       overridden per call: a camera outside camera_limit sends that call down the replica traversal */
    float camera_limit;             /* largest |camera position coordinate| for which a call keeps the commit-time decision: 13 inside the coordinate range,
                                       8 x the scene's largest coordinate in the reachability mode */
} TerraAmdTraversalInfo;
typedef enum {
    kTerraAmdCallNone = 0,
    kTerraAmdCallReplica = 1,        /* the reference's tree, every decision of terra_bvh_traverse reproduced */
    kTerraAmdCallLeafCull = 2,       /* ... with the leaf-box cull */
    kTerraAmdCallFastTree = 3,
    kTerraAmdCallFastTreeReach = 4   /* fast tree + reachability replay (scenes outside the coordinate range) */
} TerraAmdCallTraversal;
int  terra_amd_traversal_info ( HTerraScene scene, TerraAmdTraversalInfo* out );
/* Who builds the fast tree at commit (replaces src/TerraBVH.c:128-244 for that tree): 0 (default) = the host, 3-axis binned SAH;
   1 = the GPU, a linear BVH (Morton sort + Karras hierarchy + bottom-up fit) built from the triangle soup already in HBM in a few
   milliseconds. Same node format, same traversal kernel, same image; the LBVH visits more nodes per ray than the SAH tree. The
   reference tree -- needed for the visit ranks that break depth ties, and for replica traversal -- is built on the host either way. */
int  terra_amd_set_tree_builder ( HTerraScene scene, int builder );
int  terra_amd_get_tree_builder ( HTerraScene scene );
/* TEST HOOK, off by default (amount 0): at the next commit the DEVICE copy of the reference tree's boxes is shrunk by `amount` on every side, so that the
   reference traversal -- as the device replays it -- misses triangles the watertight test would hit: the situation the reachability replay exists for and
   that float rounding alone produces too rarely to test. While it is on, the shortcuts that rest on the commit-time containment proof (leaf-box cull, fast
   tree inside the coordinate range) are not taken and terra_amd_traversal_info() says so. */
int  terra_amd_debug_shrink_reference_boxes ( HTerraScene scene, float amount );
/* TEST HOOK, off by default (0): every launch of this scene plans `entries` more traversal-stack entries per lane than the tree needs (1 KB of LDS per entry and block):
   lets a test drive the launch path of trees deeper than 64 KB of LDS per block -- which the reference's own builder only produces on inputs of pathological size -- with an
   ordinary scene. The image does not change. A plan beyond what a block can hold fails the call with a message, as a real tree of that depth would. */
int  terra_amd_debug_pad_stack ( HTerraScene scene, int entries );
/* TEST HOOK, off by default (0): fast-tree launches of this scene keep only `entries` entries of a lane's traversal stack in LDS (default: 16) and the rest in HBM, so that
   an ordinary scene exercises the HBM part, which rays of real scenes almost never reach. The image does not change. */
int  terra_amd_debug_fast_stack_lds ( HTerraScene scene, int entries );

/* Sample split: how many lanes share one pixel. With split = S (a power of two up to 64; default 1) a render call of
   spp samples per pixel runs as S chunks of spp/S samples on S lanes, chunk j drawing from the random
   streams keyed (pixel, samples already in the pixel + j * spp/S), and the chunk sums are added to the
   pixel in chunk order: bit for bit the framebuffer that S successive calls of spp/S samples produce
   (each call of the reference sums its own samples and then adds them to the running sum,
   src/Terra.c:551-572). It exists for small tiles and shards: one GPU has more lanes than a 1/8 share
   of a 1080p frame has pixels. If spp is not a multiple of S the largest power of two dividing it is
   used. split = 0 picks S per call from the call's own size (about 200 jobs per lane the
   GPU holds at once -- about 50 for the launches that use the job order, below --, chunks of at least 16 samples, at most 32 lanes per pixel: 32 for a 128-pixel tile
   or a 1/8 shard of a 1080p frame at 512 spp, 8 for the whole frame of a small scene): for clients that
   render in small tiles, as the reference's does (satellite/include/Config.hpp:25), and for whole frames alike -- the render grid
   is persistent and hands (pixel, chunk) jobs to its lanes from a queue, and a launch with few jobs per lane ends with its last jobs
   ramping down alone. A launch parameter: no commit needed. */
int  terra_amd_set_sample_split ( HTerraScene scene, int split );
int  terra_amd_get_sample_split ( HTerraScene scene );
/* What split = 0 chooses for rank 0's share (tiles t with t % world == 0; world = 1: the whole rectangle) of a width x height rectangle at spp samples per pixel:
   job_ordered = 1 for scenes whose launches use the job order (TerraAmdTraversalInfo::lds_resident scenes with terra_amd_set_job_order on, rectangles of at least
   256 pixel blocks) -- they end on short jobs and want about 50 jobs per resident lane --, 0 for the others, which want about 200. For clients that must know the
   chunking of a call (bench.py renders its sharded and unsharded frames with the same number so that they can be compared bit for bit). -1 on bad arguments. */
int  terra_amd_auto_sample_split ( size_t width, size_t height, size_t tile, int world, size_t spp, int job_ordered );

/* Job order, on by default. A render launch on a scene that is staged whole in the compute units' local memory hands its (pixel, chunk) jobs to the lanes of a persistent
   grid, and ends when the longest of the jobs in flight at the end is done: about a millisecond on a Cornell-box frame whatever the launch's share of it -- a seventh of the
   time of a 1/8 shard. Such a launch therefore first classifies its 16x16 pixel blocks with five camera rays each (centre and corners) and hands out the blocks some camera
   ray hits first, the blocks whose camera rays all leave the scene -- short jobs: one traversal per sample, nothing to shade -- last. Which lane runs a job, and when,
   has no influence on what the job computes: the framebuffer is the same bit for bit (tests/test_job_order.py). Launches of fewer than 256 pixel blocks (a 256 x 256
   rectangle) keep the order of the numbering: a tile-sized call is usually one of several in flight, whose work hides its tail, and the two small kernels that make the
   order would queue behind the other callers' grids. terra_amd_set_job_order(scene, 0) keeps the order of the numbering everywhere, 2 orders launches of any size
   (what the tests use on small frames); 1 is the default. A launch parameter: no commit needed. */
int  terra_amd_set_job_order ( HTerraScene scene, int on );
int  terra_amd_get_job_order ( HTerraScene scene );

/* Environment lighting, off by default. The reference evaluates scene options' environment_map for a ray
   that leaves the scene, multiplies the throughput by it and then drops the result: the line that would
   add it is commented out (src/Terra.c:1053-1058), so the environment never reaches the image. With
   on = 1 such a ray adds throughput * environment, where the attribute is a constant
   (terra_attribute_init_constant) or a lat-long lookup by ray direction (terra_attribute_init_cubemap ->
   terra_texture_sample_latlong, src/Terra.c:468-477). Takes effect at the next terra_scene_commit().
   This changes images relative to the reference by design ("behind a flag"). */
int  terra_amd_set_environment_lighting ( HTerraScene scene, int on );
int  terra_amd_get_environment_lighting ( HTerraScene scene );

/* Sampler integration, off by default, PARITY UNPINNED. The reference constructs a stratified or Halton "hemisphere sampler" per pixel
   (src/Terra.c:535-548) and never draws from it; only the stratified method's spp round-up has an effect. With on = 1 and one of those two sampling methods,
   camera sample n of a pixel (n counts the samples the pixel has received, across calls) takes element n of that sampler -- Halton: the radical-inverse pair
   (base 3, base 2) of n (src/Terra.c:734-755); stratified: `strata` x `strata` strata, 16 samples per stratum, element n mod (strata^2 * 16), the two offsets
   drawn from the pixel's camera stream as terra_sampler_stratified_next_pair does (src/Terra.c:714-723) -- and uses it as the first two variates of the BSDF
   sample at bounce 0 (src/Terra.c:1068-1071). Stream B is consumed as always; later bounces, light samples and MIS's own BSDF sample are untouched. The wiring
   is this library's definition (the oracle's orc_set_sampler_integration restates it; device == oracle bit for bit); there is nothing in the reference to pin
   it to. A launch parameter: no commit needed. With the switch off every image is the reference's. */
int  terra_amd_set_sampler_integration ( HTerraScene scene, int on );
int  terra_amd_get_sampler_integration ( HTerraScene scene );

/* Environment importance sampling, off by default, PARITY UNPINNED ("TerraDistribution1D/2D for env-map importance sampling"). The reference
   implements TerraDistribution2D (src/Terra.c:812-846) and nothing calls it. With on = 1, environment lighting on (above) and a lat-long environment TEXTURE of at
   least three components, terra_scene_commit() tabulates the map as that distribution -- one value per texel, luminance (0.2126, 0.7152, 0.0722) x sin(theta of the
   row's centre), initialised exactly as terra_distribution_2d_init does -- and the Direct and Direct+MIS integrators take ONE environment sample per shaded hit, after
   their own light samples: two more draws of stream B choose a texel (the row from the first, the column from the second; terra_distribution_2d_sample's
   arithmetic), the direction is the inverse of the lookup's mapping (src/Terra.c:468-477), the sample counts when it lies in the upper hemisphere of the shading
   normal and its shadow ray leaves the scene, and it adds texel x BSDF x cosine / density (density = texel probability x texels / (2 terra_PI^2 sin theta)). A path
   ray that leaves the scene after bounce 0 then no longer adds the environment (the camera ray still does): the same integral, far less noise for maps with small
   bright regions. Other integrators, constant environments and maps with fewer than three components are untouched. This library's definition (the oracle's
   orc_set_environment_sampling restates it; device == oracle bit for bit; tests/test_environment_sampling.py also checks the two estimators' means against each
   other). Takes effect at the next terra_scene_commit(). */
int  terra_amd_set_environment_sampling ( HTerraScene scene, int on );
int  terra_amd_get_environment_sampling ( HTerraScene scene );

/* Work counters of the device path, summed over all launches since the last
   reset. They define the algorithmic bytes of the roofline:
   bytes = 64*nodes + 36*tri_tests + hits*(36+60) + 12*attr_fetches + 44*pixels.
   The device-side counters (rays, nodes, box_tests, tri_tests, hits, rand_calls, attr_fetches, tri_culled) are INSTRUMENTATION and off by default, like the
   reference's TERRA_PROFILE (src/Terra.c:564,634,1643: compiled out unless defined): terra_amd_set_work_counters(scene, 1) makes subsequent render calls count
   (a launch parameter, no commit needed), at 4-6 % of the render time; a call that is handed a per-pixel draw-count buffer counts regardless. samples, pixels
   and launches are kept by the host either way. rays, hits, rand_calls and attr_fetches are functions of the image (equal in every tree mode, run after run); nodes,
   box_tests and tri_tests count the work the chosen traversal did, and on the fast tree -- where a lane that holds a leaf descends on speculatively while it waits
   for its wave's triangle step -- they vary by a fraction of a percent with how the frame's jobs happened to be dealt to the waves. */
int terra_amd_set_work_counters ( HTerraScene scene, int on );
int terra_amd_get_work_counters ( HTerraScene scene );
typedef struct {
    uint64_t rays;          /* terra_scene_raycast equivalents (src/Terra.c:1623) */
    uint64_t nodes;         /* BVH nodes popped (src/TerraBVH.c:267) */
    uint64_t box_tests;     /* slab tests (src/Terra.c:851) */
    uint64_t tri_tests;     /* watertight queries (src/TerraGeometry.c:159) */
    uint64_t hits;          /* rays that hit: one terra_surface_init each (src/Terra.c:1726) */
    uint64_t samples;       /* camera samples traced */
    uint64_t rand_calls;    /* stream-B draws == the reference's rand() calls */
    uint64_t attr_fetches;  /* (attributes_count + 1) summed over hits */
    uint64_t pixels;        /* pixels written */
    uint64_t launches;      /* render kernel launches */
    uint64_t tri_culled;    /* leaves met whose triangle test the leaf-box cull skipped (counted by fully counting launches only: rand_calls requested) */
} TerraAmdStats;
int terra_amd_get_stats ( HTerraScene scene, TerraAmdStats* out );
int terra_amd_reset_stats ( HTerraScene scene );
/* Debug builds only (-DTERRA_CHECK_BOUNDS=1 verifies every traversal-stack and leaf-list write against the sizes
   the host planned): number of writes refused since the last reset; the shipped build always reports 0. -1 = no device. */
long long terra_amd_debug_faults ( HTerraScene scene );
/* Debug builds only (-DTERRA_PHASE_STATS=1, tools/phase_stats.py): 16 lane-occupancy counters of the render kernel's
   phases since the last reset (wave-level iterations and lanes active in them); the shipped build reports zeros. */
int terra_amd_debug_counters ( HTerraScene scene, unsigned long long* out16 );

/* Flattened-scene facts after commit (for tests and the roofline model). */
typedef struct {
    uint32_t triangles, nodes, objects, lights;
    uint32_t lights_triangles_count;
    int32_t  max_stack;      /* traversal stack entries a ray can need */
    uint64_t device_bytes;   /* HBM bytes held by the scene replica */
} TerraAmdSceneInfo;
int terra_amd_scene_info ( HTerraScene scene, TerraAmdSceneInfo* out );
/* Copies the host BVH (reference node layout, src/TerraBVH.h:13-17, 64 B/node)
   into `out` (capacity in nodes); returns the node count or a negative status. */
int terra_amd_scene_bvh_nodes ( HTerraScene scene, void* out, int capacity );

/* terra_render() (include/Terra.h:229, src/Terra.c:512-635) on a framebuffer
   that already lives in HBM: d_pixels = float[3]*fb_width*fb_height,
   d_results = {float acc[3]; int samples}*fb_width*fb_height, both row-major
   like TerraFramebuffer. Asynchronous on `stream`; nothing is copied to the
   host. d_rand_calls (optional, uint32 per pixel, same indexing) receives the
   number of stream-B draws of this call per pixel. */
int terra_amd_render_device ( const TerraCamera* camera, HTerraScene scene,
                              void* d_pixels, void* d_results, size_t fb_width, size_t fb_height,
                              size_t x, size_t y, size_t width, size_t height,
                              void* d_rand_calls, void* stream );

/* Tile-sharded form for one-process-per-GPU rendering (the reference shards
   the same way over CPU threads: satellite/src/Renderer.cpp:316-350): the
   rectangle is cut into tile_size x tile_size tiles numbered row-major and this
   call renders the tiles t with t % world == rank. */
int terra_amd_render_device_sharded ( const TerraCamera* camera, HTerraScene scene,
                                      void* d_pixels, void* d_results, size_t fb_width, size_t fb_height,
                                      size_t x, size_t y, size_t width, size_t height,
                                      size_t tile_size, int rank, int world,
                                      void* d_rand_calls, void* stream );

/* Gather support for the sharded form: copy this rank's tiles of (pixels,
   results) into / out of a packed buffer of tiles_of_rank * tile_size^2 * 28
   bytes (12 B pixel + 16 B result per pixel, tile-major, rows inside a tile
   contiguous). The packed buffers of all ranks are what the single RCCL gather
   moves. Both return the number of tiles handled or a negative status. */
int    terra_amd_shard_tile_count ( size_t width, size_t height, size_t tile_size, int rank, int world );
size_t terra_amd_shard_packed_bytes ( size_t width, size_t height, size_t tile_size, int world );
int terra_amd_pack_tiles ( const void* d_pixels, const void* d_results, size_t fb_width, size_t fb_height,
                           size_t x, size_t y, size_t width, size_t height, size_t tile_size, int rank, int world,
                           void* d_packed, void* stream );
int terra_amd_unpack_tiles ( void* d_pixels, void* d_results, size_t fb_width, size_t fb_height,
                             size_t x, size_t y, size_t width, size_t height, size_t tile_size, int rank, int world,
                             const void* d_packed, void* stream );

/* Blocks until all work this library queued on `stream` has finished. */
int terra_amd_synchronize ( void* stream );

/* Times `launches` back-to-back terra_amd_render_device() calls with HIP events
   recorded on `stream` and returns the average kernel milliseconds in *ms_avg. */
int terra_amd_time_render_device ( const TerraCamera* camera, HTerraScene scene,
                                   void* d_pixels, void* d_results, size_t fb_width, size_t fb_height,
                                   size_t x, size_t y, size_t width, size_t height,
                                   int launches, void* stream, float* ms_avg );

/* ---- unit-level device entry points -----------------------------------------
   Each runs the DEVICE implementation of one reference function over arrays of
   independent inputs (host pointers; copied in and out). They exist so parity
   tests can pin every stage of the path, not only whole images. */

/* first n floats of the camera PCG for each seed (src/Terra.c:678-701); out[nseeds*n] */
int terra_amd_unit_pcg ( const uint32_t* seeds, int nseeds, int n, float* out );
/* stream keys (DESIGN.md "Randomness"): out3[i] = {seedA, stateB, incB} as uint64 */
int terra_amd_unit_stream_keys ( uint64_t frame_seed, const uint64_t* pix, const uint64_t* samples_so_far, int n, uint64_t* out3 );
/* terra_ray_aabb_intersection (src/Terra.c:851-878): hit[n], tmin[n], tmax[n] (t* written only on hit) */
int terra_amd_unit_ray_aabb ( int n, const float* origins3, const float* dirs3, const float* boxes6, int* hit, float* tmin, float* tmax );
/* watertight init+query (src/TerraGeometry.c:98-138,159-260): out8 = u,v,w,depth,px,py,pz,0 */
int terra_amd_unit_watertight ( int n, const float* origins3, const float* dirs3, const float* tris9, int* hit, float* out8 );
/* Moeller-Trumbore (src/Terra.c:880-922): out4 = t,px,py,pz */
int terra_amd_unit_moller_trumbore ( int n, const float* origins3, const float* dirs3, const float* tris9, int* hit, float* out4 );
/* terra_bvh_traverse on the committed scene (src/TerraBVH.c:250-310): prim = obj | tri<<8, point3 */
int terra_amd_unit_bvh_traverse ( HTerraScene scene, int n, const float* origins3, const float* dirs3, int* found, uint32_t* prim, float* point3 );
/* The same query through the fast tree (terra_amd_set_tree_mode 1 / 2; ordered, culled traversal, ties resolved by the
   reference's leaf visit order): same found / prim / point3 as terra_amd_unit_bvh_traverse for any ray, plus the number of
   nodes each ray visited. Fails if the committed scene has no fast tree. */
int terra_amd_unit_bvh_traverse_fast ( HTerraScene scene, int n, const float* origins3, const float* dirs3, int* found, uint32_t* prim, float* point3, uint32_t* nodes_visited );
/* terra_scene_raycast + terra_surface_init (src/Terra.c:1623-1657,1726-1764):
   obj[n] (-1 miss), tri[n], point3, surface47 = TerraShadingSurface as 47 floats */
int terra_amd_unit_raycast ( HTerraScene scene, int n, const float* origins3, const float* dirs3, int* obj, int* tri, float* point3, float* surface47 );
/* terra_trace (src/Terra.c:1039-1097) for n primary rays with explicit stream-B state: radiance3, rand_calls */
int terra_amd_unit_trace ( HTerraScene scene, int n, const float* origins3, const float* dirs3, const uint64_t* stateB, const uint64_t* incB, float* radiance3, uint32_t* rand_calls );
/* BSDF presets (src/TerraPresets.c:34-146). kind: 0 diffuse, 1 phong, 2 GGX, 3 glass (the last two are
   this library's own definitions, TerraPresets.h). surfaces47 in/out (Phong and glass write scratch slots). Per item: e[3], wo[3] -> wi[3], pdf, f[3] (pdf/eval at the sampled wi) */
int terra_amd_unit_bsdf ( int kind, int n, float* surfaces47, const float* e3, const float* wo3, float* wi3, float* pdf, float* f3 );
/* camera (src/Terra.c:1770-1799): dirs3 in world space for pixel (x,y), jitter, r1, r2 */
int terra_amd_unit_camera ( const TerraCamera* camera, size_t fb_width, size_t fb_height, int n, const uint32_t* xy2, float jitter, const float* r2, float* dirs3 );
/* tonemap (src/Terra.c:578-627): colors3 in place */
int terra_amd_unit_tonemap ( int op, float gamma, int n, float* colors3 );
/* device math: fn 0 sinf, 1 cosf, 2 powf(x,y), 3 acosf, 4 atan2f(x,y) */
int terra_amd_unit_math ( int fn, int n, const float* x, const float* y, float* out );
/* The plane values of the fast tree's nodes: x[i] rounded to binary16 towards -inf (up = 0) or +inf (up != 0), as bits -- so that a box of such planes
   contains the box it was made from (host arithmetic; no device needed) */
int terra_amd_unit_half_outward ( const double* x, int n, int up, uint16_t* out );
/* SURVEY.md 8f N4, unit level. The reference constructs these samplers and never draws from them on the render path
   (src/Terra.c:535-548) and nothing calls its distributions; they are provided and pinned as stand-alone device functions.
   terra_sampler_stratified_next_pair (src/Terra.c:714-723) for one sampler per seed: out2[nseeds][n][2]; n <= strata^2 * samples_per_stratum */
int terra_amd_unit_stratified ( const uint32_t* seeds, int nseeds, int strata, int samples_per_stratum, int n, float* out2 );
/* terra_sampler_halton_next_pair (src/Terra.c:734-755), elements first .. first+n-1 of the (base 3, base 2) sequence: out2[n][2] */
int terra_amd_unit_halton ( int first, int n, float* out2 );
/* terra_distribution_1d_init + _sample (src/Terra.c:760-810): builds the distribution over f[n] on the device and samples it at e[m]:
   x[m] (FLT_MAX when no bucket holds e: the reference asserts), pdf[m], idx[m]; cdf_out[n] / integral_out optional */
int terra_amd_unit_distribution_1d ( const float* f, size_t n, const float* e, int m, float* x, float* pdf, uint32_t* idx, float* cdf_out, float* integral_out );
/* terra_distribution_2d_init + _sample (src/Terra.c:812-846) over f[ny][nx] at (e1,e2)[m]: xy2[m][2] = (row coordinate, column
   coordinate) as the reference returns them, pdf[m]; marginal_cdf_out[ny] optional */
int terra_amd_unit_distribution_2d ( const float* f, size_t nx, size_t ny, const float* e12, int m, float* xy2, float* pdf, float* marginal_cdf_out );

#ifdef __cplusplus
}
#endif
#endif /* TERRA_AMD_H */
