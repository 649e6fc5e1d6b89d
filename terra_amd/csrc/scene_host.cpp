// scene_host.cpp -- host side of libterra_amd.so: the Terra.h scene/framebuffer API,
// terra_scene_commit (host BVH build in the reference's tree layout + upload of the
// flattened scene to HBM) and terra_render / terra_amd_render_device (kernel launch).
//
// Reference counterparts: scene lifecycle src/Terra.c:130-282, framebuffer :309-345,
// attributes :287-304, render :512-635, BVH build src/TerraBVH.c:70-244 and
// src/Terra.c:972-997, light list src/Terra.c:194-231.
//
// There is no CPU rendering path in this library. If the device is missing or a
// material cannot run on it, the call fails loudly (terra_amd_last_error()).
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <unordered_set>
#include <atomic>
#include <chrono>
#include <vector>

#include "../../include/terra_amd.h"
#include "../../include/TerraPresets.h"
#include "dev_types.h"
#include "kernels.h"
#include "tree_build.h"
#include "multi_gpu.h"

// ------------------------------------------------------------------------------
// error channel
// ------------------------------------------------------------------------------
static thread_local std::string g_last_error;
// process-wide: the FIRST error any thread recorded since the last terra_amd_clear_first_error(). terra_render() is void and the
// reference's client calls it from worker threads while the main thread polls (satellite/src/Renderer.cpp:70-98,118-157):
// the per-thread channel alone would never show a worker's failure to the thread that looks.
static std::mutex g_first_error_mutex;
static std::string g_first_error;
static int g_first_error_status = 0;

static int fail ( TerraAmdStatus st, const char* fmt, ... ) {
    char buf[512];
    va_list a; va_start ( a, fmt ); vsnprintf ( buf, sizeof buf, fmt, a ); va_end ( a );
    g_last_error = buf;
    {
        std::lock_guard<std::mutex> lock ( g_first_error_mutex );
        if ( g_first_error.empty() ) { g_first_error = buf; g_first_error_status = ( int ) st; }
    }
    fprintf ( stderr, "[terra_amd] error: %s\n", buf );
    return ( int ) st;
}
#define HIP_TRY(expr, st) do { hipError_t e_ = ( expr ); if ( e_ != hipSuccess ) return fail ( st, "%s: %s", #expr, hipGetErrorString ( e_ ) ); } while ( 0 )

extern "C" const char* terra_amd_last_error ( void ) { return g_last_error.c_str(); }
extern "C" void terra_amd_clear_error ( void ) { g_last_error.clear(); }
extern "C" int terra_amd_first_error ( char* buf, size_t capacity ) {
    std::lock_guard<std::mutex> lock ( g_first_error_mutex );
    if ( buf && capacity ) snprintf ( buf, capacity, "%s", g_first_error.c_str() );
    return g_first_error.empty() ? 0 : g_first_error_status;
}
extern "C" void terra_amd_clear_first_error ( void ) {
    std::lock_guard<std::mutex> lock ( g_first_error_mutex );
    g_first_error.clear(); g_first_error_status = 0;
}

// ------------------------------------------------------------------------------
// device selection
// ------------------------------------------------------------------------------
static int g_device = 0;
extern "C" int terra_amd_device_count ( void ) {
    int n = 0;
    if ( hipGetDeviceCount ( &n ) != hipSuccess ) return 0;
    return n;
}
extern "C" int terra_amd_set_device ( int device ) {
    int n = terra_amd_device_count();
    if ( device < 0 || device >= n ) return fail ( kTerraAmdErrNoDevice, "device %d not available (%d visible)", device, n );
    HIP_TRY ( hipSetDevice ( device ), kTerraAmdErrNoDevice );
    g_device = device;
    return 0;
}
extern "C" int terra_amd_get_device ( void ) { return g_device; }
// The devices a scene committed from now on is replicated on (terra_amd_set_devices): empty = the one device above. devices[0] is the scene's primary device: it holds
// the staging frame of terra_render() / terra_amd_render_multi() and receives the gather.
static std::mutex g_devices_lock;
static std::vector<int> g_devices;
static std::atomic<int> g_replicas_share_device { 0 };       // TEST HOOK (terra_amd_debug_replicas_share_device)
extern "C" int terra_amd_debug_replicas_share_device ( int on ) { g_replicas_share_device.store ( on ? 1 : 0 ); return 0; }
extern "C" int terra_amd_set_devices ( const int* devices, int count ) {
    if ( count < 0 || count > 64 || ( count > 0 && !devices ) ) return fail ( kTerraAmdErrBadArgument, "terra_amd_set_devices: %d devices", count );
    const int n = terra_amd_device_count();
    for ( int i = 0; i < count; ++i ) {
        if ( devices[i] < 0 || devices[i] >= n ) return fail ( kTerraAmdErrNoDevice, "device %d not available (%d visible)", devices[i], n );
        for ( int j = 0; j < i && !g_replicas_share_device.load(); ++j ) if ( devices[j] == devices[i] ) return fail ( kTerraAmdErrBadArgument, "device %d is listed twice", devices[i] );
    }
    std::lock_guard<std::mutex> g ( g_devices_lock );
    const std::vector<int> now ( devices, devices + count );
    if ( now != g_devices ) multigpu::forget_communicators();
    g_devices = now;
    if ( count > 0 ) g_device = devices[0];
    return 0;
}
extern "C" int terra_amd_get_devices ( int* out, int capacity ) {
    std::lock_guard<std::mutex> g ( g_devices_lock );
    const int n = g_devices.empty() ? 1 : ( int ) g_devices.size();
    for ( int i = 0; i < n && i < capacity && out; ++i ) out[i] = g_devices.empty() ? g_device : g_devices[ ( size_t ) i];
    return n;
}
// which device of `world` renders tile t of a frame: the shard rule of the render kernels (DevRenderParams::rank / world), of bench.py's ranks and of
// terra_amd_render_multi -- the reference deals tiles to its worker threads the same way (satellite/src/Renderer.cpp:316-350)
extern "C" int terra_amd_shard_owner ( size_t tile_index, int world ) {
    if ( world < 1 ) return fail ( kTerraAmdErrBadArgument, "bad shard arguments" );
    return ( int ) ( tile_index % ( size_t ) world );
}

// ------------------------------------------------------------------------------
// system
// ------------------------------------------------------------------------------
extern "C" void* terra_malloc ( size_t size ) { return malloc ( size ); }
extern "C" void* terra_realloc ( void* p, size_t size ) { return realloc ( p, size ); }
extern "C" void  terra_free ( void* p ) { free ( p ); }
extern "C" void  terra_log ( const char* str, ... ) { va_list a; va_start ( a, str ); vfprintf ( stdout, str, a ); va_end ( a ); }

// ------------------------------------------------------------------------------
// preset markers. The device BSDFs live in trace_device.h; these host symbols only
// identify a preset. Calling them is an error: no CPU shading here.
// ------------------------------------------------------------------------------
static TerraFloat3 host_call_refused ( const char* what ) {
    fail ( kTerraAmdErrUnsupported, "%s called on the host: BSDF presets execute on the device only", what );
    TerraFloat3 z = { 0.f, 0.f, 0.f };
    return z;
}
extern "C" {
TerraFloat3 terra_bsdf_diffuse_sample ( const TerraShadingSurface*, float, float, float, const TerraFloat3* ) { return host_call_refused ( "terra_bsdf_diffuse_sample" ); }
float       terra_bsdf_diffuse_pdf ( const TerraShadingSurface*, const TerraFloat3*, const TerraFloat3* ) { host_call_refused ( "terra_bsdf_diffuse_pdf" ); return 0.f; }
TerraFloat3 terra_bsdf_diffuse_eval ( const TerraShadingSurface*, const TerraFloat3*, const TerraFloat3* ) { return host_call_refused ( "terra_bsdf_diffuse_eval" ); }
TerraFloat3 terra_bsdf_phong_sample ( const TerraShadingSurface*, float, float, float, const TerraFloat3* ) { return host_call_refused ( "terra_bsdf_phong_sample" ); }
float       terra_bsdf_phong_pdf ( const TerraShadingSurface*, const TerraFloat3*, const TerraFloat3* ) { host_call_refused ( "terra_bsdf_phong_pdf" ); return 0.f; }
TerraFloat3 terra_bsdf_phong_eval ( const TerraShadingSurface*, const TerraFloat3*, const TerraFloat3* ) { return host_call_refused ( "terra_bsdf_phong_eval" ); }
TerraFloat3 terra_bsdf_ggx_sample ( const TerraShadingSurface*, float, float, float, const TerraFloat3* ) { return host_call_refused ( "terra_bsdf_ggx_sample" ); }
float       terra_bsdf_ggx_pdf ( const TerraShadingSurface*, const TerraFloat3*, const TerraFloat3* ) { host_call_refused ( "terra_bsdf_ggx_pdf" ); return 0.f; }
TerraFloat3 terra_bsdf_ggx_eval ( const TerraShadingSurface*, const TerraFloat3*, const TerraFloat3* ) { return host_call_refused ( "terra_bsdf_ggx_eval" ); }
TerraFloat3 terra_bsdf_glass_sample ( const TerraShadingSurface*, float, float, float, const TerraFloat3* ) { return host_call_refused ( "terra_bsdf_glass_sample" ); }
float       terra_bsdf_glass_pdf ( const TerraShadingSurface*, const TerraFloat3*, const TerraFloat3* ) { host_call_refused ( "terra_bsdf_glass_pdf" ); return 0.f; }
TerraFloat3 terra_bsdf_glass_eval ( const TerraShadingSurface*, const TerraFloat3*, const TerraFloat3* ) { return host_call_refused ( "terra_bsdf_glass_eval" ); }
void terra_bsdf_ggx_init ( TerraBSDF* b ) { b->sample = terra_bsdf_ggx_sample; b->pdf = terra_bsdf_ggx_pdf; b->eval = terra_bsdf_ggx_eval; }
void terra_bsdf_glass_init ( TerraBSDF* b ) { b->sample = terra_bsdf_glass_sample; b->pdf = terra_bsdf_glass_pdf; b->eval = terra_bsdf_glass_eval; }
void terra_bsdf_diffuse_init ( TerraBSDF* b ) { b->sample = terra_bsdf_diffuse_sample; b->pdf = terra_bsdf_diffuse_pdf; b->eval = terra_bsdf_diffuse_eval; }
void terra_bsdf_phong_init ( TerraBSDF* b ) { b->sample = terra_bsdf_phong_sample; b->pdf = terra_bsdf_phong_pdf; b->eval = terra_bsdf_phong_eval; }
}

// ------------------------------------------------------------------------------
// attributes and textures (host data structures of the API; reference src/Terra.c:287-507)
// ------------------------------------------------------------------------------
extern "C" void terra_attribute_init_constant ( TerraAttribute* a, const TerraFloat3* v ) { a->state = nullptr; a->finalize = nullptr; a->eval = nullptr; a->value = *v; }
extern "C" void terra_attribute_init_texture ( TerraAttribute* a, TerraTexture* t ) { a->state = t; a->eval = terra_texture_sample; a->finalize = terra_texture_finalize; }
extern "C" void terra_attribute_init_cubemap ( TerraAttribute* a, TerraTexture* t ) { a->state = t; a->eval = terra_texture_sample_latlong; a->finalize = terra_texture_finalize; }

extern "C" bool terra_texture_init ( TerraTexture* t, size_t w, size_t h, size_t comps, const void* data ) {
    size_t bytes = w * h * comps;
    t->pixels = malloc ( bytes ? bytes : 1 );
    if ( !t->pixels ) return false;
    memcpy ( t->pixels, data, bytes );
    t->width = ( uint16_t ) w; t->height = ( uint16_t ) h; t->components = ( uint8_t ) comps; t->depth = 1;
    return true;
}
extern "C" bool terra_texture_init_hdr ( TerraTexture* t, size_t w, size_t h, size_t comps, const float* data ) {
    size_t bytes = sizeof ( float ) * w * h * comps;
    t->pixels = malloc ( bytes ? bytes : 1 );
    if ( !t->pixels ) return false;
    memcpy ( t->pixels, data, bytes );
    t->width = ( uint16_t ) w; t->height = ( uint16_t ) h; t->components = ( uint8_t ) comps; t->depth = 4;
    return true;
}
extern "C" TerraFloat3 terra_texture_read ( TerraTexture* t, size_t x, size_t y ) {
    TerraFloat3 out = { 0.f, 0.f, 0.f };
    if ( !t || !t->pixels || !t->width || !t->height ) return out;
    const size_t W = t->width, H = t->height;
    if ( t->address_mode == kTerraTextureAddressClamp ) { x = std::min ( x, W - 1 ); y = std::min ( y, H - 1 ); }
    else if ( t->address_mode == kTerraTextureAddressWrap ) { x %= W; y %= H; }
    else if ( ( x / W ) % 2 == 0 ) { x %= W; y %= H; }
    else { x = W - ( x % W ); y = H - ( y % H ); x = std::min ( x, W - 1 ); y = std::min ( y, H - 1 ); }
    const size_t texel = ( y * W + x ) * t->components;
    if ( t->depth == 1 ) {
        const uint8_t* p = ( const uint8_t* ) t->pixels + texel;
        out.x = p[0] / 255.f; out.y = t->components > 1 ? p[1] / 255.f : 0.f; out.z = t->components > 2 ? p[2] / 255.f : 0.f;
    } else if ( t->depth == 4 ) {
        const float* p = ( const float* ) t->pixels + texel;
        out.x = p[0]; out.y = t->components > 1 ? p[1] : 0.f; out.z = t->components > 2 ? p[2] : 0.f;
    }
    return out;
}
extern "C" TerraFloat3 terra_texture_sample ( void* tex, const void* uvp, const void* ) {
    TerraTexture* t = ( TerraTexture* ) tex; const TerraFloat2* uv = ( const TerraFloat2* ) uvp;
    size_t ix = ( size_t ) uv->x, iy = ( size_t ) uv->y;
    if ( t->filter == kTerraFilterPoint ) return terra_texture_read ( t, ix, iy );
    TerraFloat3 s = { 0.f, 0.f, 0.f };
    if ( t->filter == kTerraFilterBilinear ) {
        size_t x2 = std::min<size_t> ( ix + 1, ( size_t ) t->width - 1 ), y2 = std::min<size_t> ( iy + 1, ( size_t ) t->height - 1 );
        TerraFloat3 n1 = terra_texture_read ( t, ix, iy ), n2 = terra_texture_read ( t, x2, iy );
        TerraFloat3 n3 = terra_texture_read ( t, ix, y2 ), n4 = terra_texture_read ( t, x2, y2 );
        float wu = uv->x - ix, wv = uv->y - iy, wou = 1.f - wu, wov = 1.f - wv;
        s.x = ( n1.x * wou + n2.x * wu ) * wov + ( n3.x * wou + n4.x * wu ) * wv;
        s.y = ( n1.y * wou + n2.y * wu ) * wov + ( n3.y * wou + n4.y * wu ) * wv;
        s.z = ( n1.z * wou + n2.z * wu ) * wov + ( n3.z * wou + n4.z * wu ) * wv;
    }
    return s;
}
extern "C" TerraFloat3 terra_texture_sample_latlong ( void* tex, const void* dirp, const void* ) {
    TerraTexture* t = ( TerraTexture* ) tex;
    TerraFloat3 d = terra_normf3 ( ( const TerraFloat3* ) dirp );
    float theta = acosf ( d.y );
    float phi = atan2f ( d.z, d.x ) + terra_PI;
    size_t u = ( size_t ) ( ( phi / ( 2 * terra_PI ) ) * t->width );
    size_t v = ( size_t ) ( ( theta / ( terra_PI ) ) * t->height );
    return terra_texture_read ( t, u, v );
}
extern "C" void terra_texture_destroy ( TerraTexture* t ) { if ( t ) { free ( t->pixels ); t->pixels = nullptr; } }
extern "C" void terra_texture_finalize ( void* tex ) {
    TerraTexture* t = ( TerraTexture* ) tex;
    if ( !t || !t->pixels ) return;
    size_t n = ( size_t ) t->width * t->height * t->components;
    if ( t->depth == 1 ) { uint8_t* p = ( uint8_t* ) t->pixels; for ( size_t i = 0; i < n; ++i ) p[i] = ( uint8_t ) ( powf ( p[i] / 255.f, 2.2f ) * 255 ); }
    else if ( t->depth == 4 ) { float* p = ( float* ) t->pixels; for ( size_t i = 0; i < n; ++i ) p[i] = powf ( p[i], 2.2f ); }
}

// ------------------------------------------------------------------------------
// framebuffer: pinned host memory when a device is present (faster tile copies)
// ------------------------------------------------------------------------------
static std::mutex g_pinned_lock;
static std::unordered_set<void*> g_pinned;

static void* fb_alloc ( size_t bytes ) {
    void* p = nullptr;
    if ( terra_amd_device_count() > 0 && hipHostMalloc ( &p, bytes, hipHostMallocDefault ) == hipSuccess && p ) {
        std::lock_guard<std::mutex> g ( g_pinned_lock );
        g_pinned.insert ( p );
        return p;
    }
    ( void ) hipGetLastError();
    return malloc ( bytes );
}
static void fb_free ( void* p ) {
    if ( !p ) return;
    bool pinned;
    { std::lock_guard<std::mutex> g ( g_pinned_lock ); pinned = g_pinned.erase ( p ) > 0; }
    if ( pinned ) ( void ) hipHostFree ( p ); else free ( p );
}
extern "C" bool terra_framebuffer_create ( TerraFramebuffer* fb, size_t w, size_t h ) {
    if ( !fb || w == 0 || h == 0 ) return false;
    fb->width = w; fb->height = h;
    fb->pixels = ( TerraFloat3* ) fb_alloc ( sizeof ( TerraFloat3 ) * w * h );
    fb->results = ( TerraRawIntegrationResult* ) fb_alloc ( sizeof ( TerraRawIntegrationResult ) * w * h );
    if ( !fb->pixels || !fb->results ) return false;
    terra_framebuffer_clear ( fb );
    return true;
}
extern "C" void terra_framebuffer_clear ( TerraFramebuffer* fb ) {
    memset ( fb->pixels, 0, sizeof ( TerraFloat3 ) * fb->width * fb->height );
    memset ( fb->results, 0, sizeof ( TerraRawIntegrationResult ) * fb->width * fb->height );
}
extern "C" void terra_framebuffer_destroy ( TerraFramebuffer* fb ) {
    if ( !fb ) return;
    fb_free ( fb->results ); fb_free ( fb->pixels );
    fb->results = nullptr; fb->pixels = nullptr;
}

// ------------------------------------------------------------------------------
// scene
// ------------------------------------------------------------------------------
struct HostLight { uint32_t object; float area; TerraFloat3 power; };

#ifndef TERRA_REACH_CAMERA_FACTOR
#define TERRA_REACH_CAMERA_FACTOR 8.f
#endif
#define TERRA_REACH_MAX_COORD 1e6f       // beyond it (c - o) x 2^100 (the fast tree's clamped slab test) approaches the binary32 range: replica
#define TERRA_CULL_MAX_COORD 13.0f       // limit of the numeric containment check (derivation above verify_reference_leaf_boxes)
#ifndef TERRA_SCRATCH_MAX_GB
#define TERRA_SCRATCH_MAX_GB 64                     // most scratch one launch may take from the device's pool (launch_render)
#endif
#define TERRA_FAST_STACK_MAX 2048                   // stack entries per ray beyond which the fast tree is not used (its spill space: 4 B x entries x 327,680 resident lanes)
struct Scene {
    TerraSceneOptions opts, new_opts;
    TerraObject* objects = nullptr; size_t objects_pop = 0, objects_cap = 0;
    bool dirty_objects = true, dirty_lights = true, committed = false, device_ok = false;
    uint64_t frame_seed = 0x5EED0001ull;
    int device = 0;
    // host mirrors
    std::vector<HostNode> nodes; int max_stack = 1;
    std::vector<HostLight> lights; size_t lights_triangles_count = 0;
    std::vector<uint32_t> first_tri;     // per object
    // device replica (of the primary device, `device`)
    DevScene dev; void* d_blob = nullptr; size_t d_bytes = 0;
    unsigned long long* d_counters = nullptr;
    // ... and of the other devices of the set the scene was committed for (terra_amd_set_devices): byte copies of the blob with the pointers rebased
    struct Replica { int device = -1; DevScene dev; void* d_blob = nullptr; unsigned long long* d_counters = nullptr; float* d_env_dist = nullptr; };
    std::vector<Replica> extra;
    std::vector<int> devices;           // the set, primary first (size 1: single-device scene)
    std::vector<DevTexture> tdesc_host; size_t o_tdesc = 0, blob_bytes = 0, env_dist_floats = 0;      // what replicate() needs of the upload's layout
    struct MultiCtx* multi = nullptr;   // streams / staging frames / packed buffers of terra_amd_render_multi, one set per device (made on first use)
    std::mutex multi_lock;              // a multi-device render owns every device of the set: one at a time per scene
    std::atomic<uint64_t> launches { 0 }, stat_pixels { 0 }, stat_samples { 0 };     // terra_render() is called from several threads at once
    int uniform_attr_count = -1;        // attributes_count shared by every material, or -1
    uint32_t bsdf_kinds = 0;            // mask of preset kinds in the committed scene
    // traversal policy (terra_amd_set_tree_mode): 0 = replica: the reference's tree, every traversal decision reproduced; 1 = fast tree, unconditionally;
    // 2 = automatic (default): scenes that pass the numeric containment check of verify_containment() run the reference tree with the
    // leaf-box cull when they are LDS-resident and the fast tree otherwise; scenes that fail it run as mode 0 (the reason is kept in tree_note)
    int tree_mode = 2;
    bool use_fast = false;              // what the last upload decided
    int tree_builder = 0;               // terra_amd_set_tree_builder: 0 = host (binned SAH), 1 = device (LBVH, tree_build_device.hip)
    bool fast_on_device = false;        // the fast tree of the last upload was built on the device
    bool cull_ok = false;               // leaf-box cull allowed for this scene (subject to the per-call camera check)
    bool reach = false;                 // outside the coordinate range: fast tree + reference reachability check (camera within reach_limit, checked per call)
    bool reach_cull = false;            // outside the coordinate range, LDS-resident: reference tree, leaf-box cull on leaf boxes inflated to the rounding bound (same camera limit)
    float reach_limit = 0.f;
    std::atomic<int> last_call { 0 };   // TerraAmdTraversalInfo::last_call: the traversal the most recent render call actually ran
    float coord_max = 0.f;              // largest |coordinate| of any vertex
    std::string tree_note;              // why the automatic mode chose what it chose
    uint32_t sample_split = 1;          // terra_amd_set_sample_split: chunks a call's samples are cut into (lanes per pixel)
    bool env_lighting = false;          // terra_amd_set_environment_lighting: escaping rays add throughput * environment
    bool work_counters = false;         // terra_amd_set_work_counters: the render kernels count rays / nodes / tests / hits / draws (instrumentation, off by default)
    bool env_sampling = false;          // terra_amd_set_environment_sampling: Direct / Direct+MIS sample a lat-long environment through a TerraDistribution2D (built at commit)
    float* d_env_dist = nullptr;        // its tables on the device (own allocation)
    int job_order = 1;                  // terra_amd_set_job_order (0 off, 1 on, 2 on for launches of any size): launches that key their streams ahead hand out the pixel blocks no camera ray hits last (launch_render)
    bool sampler_integration = false;   // terra_amd_set_sampler_integration: the pixel's Halton / stratified sampler feeds the first bounce (a launch parameter)
    int fast_max_stack = 1; uint32_t fast_nodes = 0;
    std::string commit_error;
    std::atomic<bool> warned_camera { false };      // the per-call fallback (camera outside camera_limit) has been reported on stderr once
    int test_pad_stack = 0;                         // terra_amd_debug_pad_stack (tests only): extra stack entries every launch plans
    int test_fast_stack_lds = 0;                    // terra_amd_debug_fast_stack_lds (tests only): entries of a fast-tree launch's stack kept in LDS (0: the default), the rest spills to HBM
    float test_shrink_reference_boxes = 0.f;       // terra_amd_debug_shrink_reference_boxes (tests only): the device copy of the reference tree's boxes is shrunk by this much
};

static Scene* S ( HTerraScene h ) { return ( Scene* ) h; }

// terra_amd_set_commit_timing(1): commit phases on stderr (tools/scale_triangles.py reads them)
static std::atomic<bool> g_commit_timing { false };
static bool timing_on() { return g_commit_timing.load ( std::memory_order_relaxed ); }
bool terra_commit_timing_on() { return timing_on(); }          // (tree_build.cpp prints its passes too)
extern "C" void terra_amd_set_commit_timing ( int on ) { g_commit_timing.store ( on != 0, std::memory_order_relaxed ); }
// host threads of the fast tree's builder (tree_build.cpp): 0 = as many as the process may use, at most 16
static std::atomic<int> g_build_threads { 0 };
int terra_build_threads() { return g_build_threads.load ( std::memory_order_relaxed ); }
extern "C" int terra_amd_set_build_threads ( int threads ) {
    if ( threads < 0 || threads > 256 ) return fail ( kTerraAmdErrBadArgument, "build threads must be 0 (automatic) .. 256" );
    g_build_threads.store ( threads, std::memory_order_relaxed );
    return 0;
}
static double now_s() { return std::chrono::duration<double> ( std::chrono::steady_clock::now().time_since_epoch() ).count(); }
static void phase ( const char* what, double& t0 ) { if ( !timing_on() ) return; double t = now_s(); fprintf ( stderr, "[terra_amd timing] %-28s %8.2f ms\n", what, ( t - t0 ) * 1e3 ); t0 = t; }

extern "C" HTerraScene terra_scene_create ( void ) {
    Scene* s = new Scene();
    memset ( &s->opts, 0, sizeof s->opts ); memset ( &s->new_opts, 0, sizeof s->new_opts ); memset ( &s->dev, 0, sizeof s->dev );
    s->objects_cap = 64;
    s->objects = ( TerraObject* ) malloc ( sizeof ( TerraObject ) * s->objects_cap );
    s->device = g_device;
    return s;
}
extern "C" TerraObject* terra_scene_add_object ( HTerraScene h, size_t n ) {
    Scene* s = S ( h );
    if ( s->objects_pop == s->objects_cap ) {
        s->objects_cap *= 2;
        s->objects = ( TerraObject* ) realloc ( s->objects, sizeof ( TerraObject ) * s->objects_cap );
    }
    TerraObject* o = &s->objects[s->objects_pop++];
    memset ( o, 0, sizeof *o );
    o->triangles = ( TerraTriangle* ) malloc ( sizeof ( TerraTriangle ) * ( n ? n : 1 ) );
    o->properties = ( TerraTriangleProperties* ) malloc ( sizeof ( TerraTriangleProperties ) * ( n ? n : 1 ) );
    o->triangles_count = n;
    s->dirty_objects = true; s->dirty_lights = true; s->committed = false;
    return o;
}
extern "C" size_t terra_scene_count_objects ( HTerraScene h ) { return S ( h )->objects_pop; }
extern "C" TerraSceneOptions* terra_scene_get_options ( HTerraScene h ) { return &S ( h )->new_opts; }
extern "C" int terra_amd_set_tree_mode ( HTerraScene h, int mode ) {
    if ( mode != 0 && mode != 1 && mode != 2 ) return fail ( kTerraAmdErrBadArgument, "tree mode %d (0 = reference tree, 1 = fast tree, 2 = automatic)", mode );
    Scene* s = S ( h );
    if ( s->tree_mode != mode ) { s->tree_mode = mode; s->dirty_objects = true; s->committed = false; }
    return 0;
}
extern "C" int terra_amd_get_tree_mode ( HTerraScene h ) { return S ( h )->tree_mode; }
extern "C" int terra_amd_set_tree_builder ( HTerraScene h, int builder ) {
    if ( builder != 0 && builder != 1 ) return fail ( kTerraAmdErrBadArgument, "tree builder %d (0 = host binned SAH, 1 = device LBVH)", builder );
    Scene* s = S ( h );
    if ( s->tree_builder != builder ) { s->tree_builder = builder; s->dirty_objects = true; s->committed = false; }
    return 0;
}
extern "C" int terra_amd_get_tree_builder ( HTerraScene h ) { return S ( h )->tree_builder; }
extern "C" int terra_amd_debug_shrink_reference_boxes ( HTerraScene h, float amount ) {
    if ( ! ( amount >= 0.f ) ) return fail ( kTerraAmdErrBadArgument, "shrink amount must be >= 0" );
    Scene* s = S ( h );
    if ( s->test_shrink_reference_boxes != amount ) { s->test_shrink_reference_boxes = amount; s->dirty_objects = true; s->committed = false; }
    return 0;
}
extern "C" int terra_amd_debug_pad_stack ( HTerraScene h, int entries ) {
    if ( entries < 0 || entries > 4096 ) return fail ( kTerraAmdErrBadArgument, "stack padding must be 0 .. 4096 entries" );
    S ( h )->test_pad_stack = entries;
    return 0;
}
extern "C" int terra_amd_debug_fast_stack_lds ( HTerraScene h, int entries ) {
    if ( entries < 0 || entries > 4096 ) return fail ( kTerraAmdErrBadArgument, "LDS stack entries must be 0 (default) .. 4096" );
    S ( h )->test_fast_stack_lds = entries;
    return 0;
}
extern "C" int terra_amd_traversal_info ( HTerraScene h, TerraAmdTraversalInfo* out ) {
    Scene* s = S ( h );
    if ( !out ) return fail ( kTerraAmdErrBadArgument, "null output" );
    memset ( out, 0, sizeof *out );
    if ( !s->committed ) return fail ( kTerraAmdErrNotCommitted, "scene not committed" );
    out->tree_mode = s->tree_mode; out->fast_tree = s->use_fast ? 1 : 0; out->fast_tree_built_on_device = s->fast_on_device ? 1 : 0; out->leaf_cull = ( s->cull_ok && !s->use_fast ) ? 1 : 0;
    out->lds_resident = ( !s->use_fast && terra_scene_fits_lds ( ( uint32_t ) s->nodes.size(), s->dev.n_tris, s->max_stack, ( uint32_t ) s->objects_pop, ( uint32_t ) s->lights.size() ) ) ? 1 : 0;
    out->max_coordinate = s->coord_max; out->max_coordinate_allowed = TERRA_CULL_MAX_COORD;
    out->last_call = s->last_call.load ( std::memory_order_relaxed ); out->camera_limit = ( s->reach || s->reach_cull ) ? s->reach_limit : TERRA_CULL_MAX_COORD;
    snprintf ( out->note, sizeof out->note, "%s", s->tree_note.c_str() );
    return 0;
}
extern "C" int terra_amd_set_sample_split ( HTerraScene h, int split ) {
    if ( split < 0 || split > 64 || ( split & ( split - 1 ) ) != 0 ) return fail ( kTerraAmdErrBadArgument, "sample split %d: must be 0 (automatic) or a power of two up to 64", split );
    S ( h )->sample_split = ( uint32_t ) split;
    return 0;
}
extern "C" int terra_amd_get_sample_split ( HTerraScene h ) { return ( int ) S ( h )->sample_split; }
static uint32_t auto_sample_split ( uint32_t blocks, uint32_t spp, bool ordered );
extern "C" int terra_amd_auto_sample_split ( size_t width, size_t height, size_t tile, int world, size_t spp, int job_ordered ) {
    if ( width == 0 || height == 0 || tile < 16 || tile % 16 || world < 1 || spp == 0 ) return fail ( kTerraAmdErrBadArgument, "terra_amd_auto_sample_split: empty rectangle, tile not a multiple of 16, or no rank" ) , -1;
    const uint64_t tiles = ( ( width + tile - 1 ) / tile ) * ( ( height + tile - 1 ) / tile ), own = ( tiles + ( uint64_t ) world - 1 ) / ( uint64_t ) world, bpt = tile / 16;
    const uint32_t blocks = ( uint32_t ) ( own * bpt * bpt );
    uint32_t split = auto_sample_split ( blocks, ( uint32_t ) spp, job_ordered != 0 && blocks >= terra_job_order_min_blocks() );
    while ( split > 1 && spp % split ) split >>= 1;
    return ( int ) split;
}
extern "C" int terra_amd_set_environment_lighting ( HTerraScene h, int on ) {
    Scene* s = S ( h );
    if ( s->env_lighting != ( on != 0 ) ) { s->env_lighting = on != 0; s->dirty_lights = true; s->committed = false; }
    return 0;
}
extern "C" int terra_amd_get_environment_lighting ( HTerraScene h ) { return S ( h )->env_lighting ? 1 : 0; }
extern "C" int terra_amd_set_work_counters ( HTerraScene h, int on ) { S ( h )->work_counters = on != 0; return 0; }
extern "C" int terra_amd_get_work_counters ( HTerraScene h ) { return S ( h )->work_counters ? 1 : 0; }
extern "C" int terra_amd_set_environment_sampling ( HTerraScene h, int on ) {
    Scene* s = S ( h ); if ( !s ) return fail ( kTerraAmdErrBadArgument, "null scene" );
    if ( s->env_sampling != ( on != 0 ) ) { s->env_sampling = on != 0; s->dirty_lights = true; s->committed = false; }
    return 0;
}
extern "C" int terra_amd_get_environment_sampling ( HTerraScene h ) { return S ( h )->env_sampling ? 1 : 0; }
extern "C" int terra_amd_set_job_order ( HTerraScene h, int on ) {
    if ( on < 0 || on > 2 ) return fail ( kTerraAmdErrBadArgument, "terra_amd_set_job_order: 0 (off), 1 (on) or 2 (on for launches of any size)" );
    S ( h )->job_order = on; return 0;
}
extern "C" int terra_amd_get_job_order ( HTerraScene h ) { return S ( h )->job_order; }
extern "C" int terra_amd_set_sampler_integration ( HTerraScene h, int on ) { S ( h )->sampler_integration = on != 0; return 0; }
extern "C" int terra_amd_get_sampler_integration ( HTerraScene h ) { return S ( h )->sampler_integration ? 1 : 0; }
extern "C" void terra_amd_set_frame_seed ( HTerraScene h, uint64_t seed ) { S ( h )->frame_seed = seed; }
extern "C" uint64_t terra_amd_get_frame_seed ( HTerraScene h ) { return S ( h )->frame_seed; }

// per device of a multi-device scene: its stream, its staging frame (the rectangle of the call, 28 B per pixel), its packed tiles; on the primary device also
// the receive buffer of the gather (every rank's packed tiles, one after the other)
struct MultiCtx {
    struct PerDevice { int device = -1; hipStream_t stream = nullptr; void* d_pixels = nullptr; void* d_results = nullptr; size_t cap_px = 0; float* d_packed = nullptr; size_t packed_floats = 0; };
    std::vector<PerDevice> dev; float* d_recv = nullptr; size_t recv_floats = 0;
    uint64_t gathers = 0, last_gather_bytes = 0;
    void release() {
        for ( PerDevice& q : dev ) {
            if ( q.device < 0 || hipSetDevice ( q.device ) != hipSuccess ) continue;
            if ( q.stream ) { ( void ) hipStreamSynchronize ( q.stream ); ( void ) hipStreamDestroy ( q.stream ); }
            if ( q.d_pixels ) ( void ) hipFree ( q.d_pixels );
            if ( q.d_results ) ( void ) hipFree ( q.d_results );
            if ( q.d_packed ) ( void ) hipFree ( q.d_packed );
            if ( &q == &dev[0] && d_recv ) ( void ) hipFree ( d_recv );
        }
        dev.clear(); d_recv = nullptr; recv_floats = 0;
    }
};
static void release_device ( Scene* s ) {
    if ( s->multi ) { s->multi->release(); delete s->multi; s->multi = nullptr; }
    for ( Scene::Replica& r : s->extra ) {
        if ( r.device < 0 || hipSetDevice ( r.device ) != hipSuccess ) continue;
        if ( r.d_env_dist ) ( void ) hipFree ( r.d_env_dist );
        if ( r.d_blob ) ( void ) hipFree ( r.d_blob );
        if ( r.d_counters ) ( void ) hipFree ( r.d_counters );
    }
    s->extra.clear();
    if ( s->d_blob || s->d_counters || s->d_env_dist ) {
        ( void ) hipSetDevice ( s->device );
        if ( s->d_env_dist ) ( void ) hipFree ( s->d_env_dist );
        if ( s->d_blob ) ( void ) hipFree ( s->d_blob );
        if ( s->d_counters ) ( void ) hipFree ( s->d_counters );
    }
    s->d_blob = nullptr; s->d_counters = nullptr; s->d_env_dist = nullptr; s->d_bytes = 0; s->device_ok = false;
    memset ( &s->dev, 0, sizeof s->dev );
}
extern "C" void terra_scene_clear ( HTerraScene h ) {
    Scene* s = S ( h );
    for ( size_t i = 0; i < s->objects_pop; ++i ) { free ( s->objects[i].triangles ); free ( s->objects[i].properties ); }
    s->objects_pop = 0;
    s->lights.clear();
    s->dirty_objects = true; s->dirty_lights = true; s->committed = false;
}
extern "C" void terra_scene_destroy ( HTerraScene h ) {
    Scene* s = S ( h );
    if ( !s ) return;
    terra_scene_clear ( h );
    release_device ( s );
    free ( s->objects );
    delete s;
}


static float triangle_area ( const TerraTriangle& t ) {
    TerraFloat3 ab = terra_subf3 ( &t.b, &t.a ), ac = terra_subf3 ( &t.c, &t.a );
    TerraFloat3 c = terra_crossf3 ( &ab, &ac );
    return terra_lenf3 ( &c ) / 2;
}

static bool is_diffuse ( const TerraBSDF& b ) { return b.sample == terra_bsdf_diffuse_sample && b.pdf == terra_bsdf_diffuse_pdf && b.eval == terra_bsdf_diffuse_eval; }
static bool is_ggx ( const TerraBSDF& b ) { return b.sample == terra_bsdf_ggx_sample && b.pdf == terra_bsdf_ggx_pdf && b.eval == terra_bsdf_ggx_eval; }
static bool is_glass ( const TerraBSDF& b ) { return b.sample == terra_bsdf_glass_sample && b.pdf == terra_bsdf_glass_pdf && b.eval == terra_bsdf_glass_eval; }
static bool is_phong ( const TerraBSDF& b ) { return b.sample == terra_bsdf_phong_sample && b.pdf == terra_bsdf_phong_pdf && b.eval == terra_bsdf_phong_eval; }

// ---- numeric containment check behind the automatic traversal mode ---------------------------------------------------------
// Both shortcuts of mode 2 -- skipping the triangle test of a leaf whose box the ray misses, and the fast tree's ordered, culled
// traversal -- return the reference's closest hit provided that every triangle a ray HITS (watertight test, src/TerraGeometry.c:159-260)
// lies inside each box that was built around it AS THE SLAB TEST SEES IT (src/Terra.c:851-878). The boxes are the triangle's extent
// grown by 1e-4 on every side (src/Terra.c:982-996) or unions of such boxes, so geometrically that is always true; numerically it
// needs the rounding errors of both tests to stay below the 1e-4 margin. With u = 2^-24 and D = the largest distance between a ray
// origin and a vertex: translating and shearing the vertices perturbs them by <= 4 u D per coordinate, the sign of an edge function
// can be wrong only within ~2 u D of the edge, the slab test's t values carry <= 3 roundings (<= ~6 u D in position), the box itself
// is rounded by <= u R: together < 16 u D, D <= 2 sqrt(3) R for origins and vertices inside [-R, R]^3, i.e. < 56 u R.
// The fast tree's boxes are traversed as binary16 planes rounded outward (tree_build_device.hip tb_half_planes_kernel): the new box contains the old one, and
// t = fma ( plane, inv, -(o * inv) ) carries 2 roundings (the product o * inv, worth u |o| in position, and the fma's), fewer than the 3 of the reference form.
// The check demands 128 u R <= 1e-4 (a factor 2 beyond those estimates): R <= 13.1 scene units, for the vertices (at commit) and for
// the camera position (per call). Scenes or cameras outside that range run in replica mode. tools/fuzz_vs_oracle.py scales scenes
// through and beyond the limit (FUZZ_SCALE) to exercise both sides.
static bool coords_within_margin ( const float* v, size_t n ) {
    for ( size_t i = 0; i < n; ++i ) if ( ! ( fabsf ( v[i] ) <= TERRA_CULL_MAX_COORD ) ) return false;      // also false for NaN / inf
    return true;
}
// every leaf child box of the reference tree contains its triangle's extent grown by (almost) 1e-4
static bool verify_reference_leaf_boxes ( const Scene* s, std::string& why ) {
    for ( size_t k = 0; k < s->nodes.size(); ++k ) for ( int c = 0; c < 2; ++c ) {
        const HostNode& h = s->nodes[k];
        if ( h.type[c] != 1 ) continue;
        const uint32_t obj = ( uint32_t ) h.index[c] & 0xffu, tri = ( uint32_t ) h.index[c] >> 8;
        if ( obj >= s->objects_pop || tri >= s->objects[obj].triangles_count ) { why = "leaf references a missing triangle"; return false; }
        const TerraTriangle& t = s->objects[obj].triangles[tri];
        const float* a = &t.a.x; const float* b = &t.b.x; const float* cc = &t.c.x;
        const float* lo = &h.aabb[c].min.x; const float* hi = &h.aabb[c].max.x;
        for ( int ax = 0; ax < 3; ++ax ) {
            const float mn = std::min ( a[ax], std::min ( b[ax], cc[ax] ) ), mx = std::max ( a[ax], std::max ( b[ax], cc[ax] ) );
            if ( ! ( ( double ) lo[ax] <= ( double ) mn - 0.9e-4 && ( double ) hi[ax] >= ( double ) mx + 0.9e-4 ) ) { why = "a leaf box of the reference tree does not contain its triangle's 1e-4 margin"; return false; }
        }
    }
    return true;
}
// every child box of the fast tree contains the boxes of all triangles below it (what its culling relies on)
static bool verify_fast_tree ( const std::vector<DevNode>& nodes, const std::vector<TerraAABB>& boxes_in_leaf_order, std::string& why ) {
    if ( nodes.empty() ) return true;
    struct Item { uint32_t node; TerraAABB bound[2]; int stage; };
    // iterative post-order: compute the union of triangle boxes below each child and compare with the stored box
    std::vector<TerraAABB> below ( nodes.size() * 2 );
    std::vector<std::pair<uint32_t, int>> st; st.push_back ( { 0u, 0 } );
    auto contains = [] ( const float* lo, const float* hi, const TerraAABB & b ) {
        return lo[0] <= b.min.x && lo[1] <= b.min.y && lo[2] <= b.min.z && hi[0] >= b.max.x && hi[1] >= b.max.y && hi[2] >= b.max.z;
    };
    auto unite = [] ( TerraAABB & a, const TerraAABB & b ) {
        a.min.x = std::min ( a.min.x, b.min.x ); a.min.y = std::min ( a.min.y, b.min.y ); a.min.z = std::min ( a.min.z, b.min.z );
        a.max.x = std::max ( a.max.x, b.max.x ); a.max.y = std::max ( a.max.y, b.max.y ); a.max.z = std::max ( a.max.z, b.max.z );
    };
    while ( !st.empty() ) {
        auto [ni, stage] = st.back(); st.pop_back();
        const DevNode& n = nodes[ni];
        if ( stage == 0 ) {
            st.push_back ( { ni, 1 } );
            for ( int c = 0; c < 2; ++c ) if ( n.child[c] != DEV_CHILD_EMPTY && ! ( n.child[c] & DEV_CHILD_LEAF ) ) {
                if ( n.child[c] >= nodes.size() ) { why = "fast tree: child index out of range"; return false; }
                st.push_back ( { n.child[c], 0 } );
            }
            continue;
        }
        for ( int c = 0; c < 2; ++c ) {
            TerraAABB u = bvh::empty_box();
            if ( n.child[c] == DEV_CHILD_EMPTY ) { below[2 * ni + c] = u; continue; }
            if ( n.child[c] & DEV_CHILD_LEAF ) {
                const uint32_t first = n.child[c] & 0x07ffffffu, cnt = ( ( n.child[c] >> 27 ) & 0xfu ) + 1;
                if ( ( size_t ) first + cnt > boxes_in_leaf_order.size() ) { why = "fast tree: leaf range out of range"; return false; }
                for ( uint32_t j = 0; j < cnt; ++j ) unite ( u, boxes_in_leaf_order[first + j] );
            } else {
                const uint32_t ch = n.child[c];
                unite ( u, below[2 * ch] ); unite ( u, below[2 * ch + 1] );
            }
            below[2 * ni + c] = u;
            const float* lo = c == 0 ? n.min0 : n.min1; const float* hi = c == 0 ? n.max0 : n.max1;
            if ( !contains ( lo, hi, u ) ) { why = "fast tree: a child box does not contain the triangle boxes below it"; return false; }
        }
    }
    return true;
}


// ---- reachability tables (DevScene::ref_replay / fast_leaf_parent / fast_leaf_mask) -------------------------------------------------------------
// nodes: the reference tree as the DEVICE holds it (breadth-first numbering, test hook applied); soup_of_fast[k] = soup index of fast triangle k.
// Level L of fast triangle k = the L-th reference node on the way up from the triangle's leaf; its test is the slab test of the box that node's parent
// stores for it. mask bit L is CLEAR when that test cannot fail for a ray that hits the triangle:
//   (1) the box contains the triangle's extent with a clearance of `margin` (the containment argument at the scene's scale), or
//   (2) L >= 1 and the box contains the box of level L - 1 component by component: for a regular ray (finite, non-zero inverse direction) fl((b - o) * inv) is
//       monotone in b, so per axis the outer box's [near, far] contains the inner box's in the very same float arithmetic, hence tmin_outer <= tmin_inner,
//       tmax_outer >= tmax_inner and "inner passes" implies "outer passes" exactly -- no error bound involved. Level L - 1 itself passes by induction (tested, or
//       cleared). Irregular rays ignore the mask and replay every level (trace_device.h reference_reaches).
struct ReachTables { std::vector<DevReplay> replay; std::vector<uint32_t> leaf_parent, leaf_mask; uint64_t levels = 0, replayed = 0; };
static void build_reach_tables ( const std::vector<DevNode>& nodes, const std::vector<DevTri>& tris, const std::vector<uint32_t>& soup_of_fast, float margin, ReachTables& out ) {
    const size_t nn = nodes.size(), nt = soup_of_fast.size();
    out.replay.assign ( nn ? nn : 1, DevReplay{} );
    std::vector<uint32_t> leaf_parent_of_soup ( tris.size(), 0u );
    for ( size_t k = 0; k < nn; ++k ) for ( int c = 0; c < 2; ++c ) {
        const uint32_t w = nodes[k].child[c];
        if ( w == DEV_CHILD_EMPTY ) continue;
        if ( w & DEV_CHILD_LEAF ) { leaf_parent_of_soup[w & 0x7fffffffu] = ( uint32_t ) k; continue; }
        DevReplay& r = out.replay[w];
        memcpy ( r.bmin, c ? nodes[k].min1 : nodes[k].min0, 12 ); memcpy ( r.bmax, c ? nodes[k].max1 : nodes[k].max0, 12 );
        r.parent = ( uint32_t ) k; r.pad = 0;
    }
    // does node q's own box (the one its parent tests) contain the box of its child node w, component by component?  (false for NaN)
    auto contains = [&] ( const DevReplay & outer, const DevReplay & inner ) {
        bool ok = true;
        for ( int a = 0; a < 3; ++a ) ok = ok && outer.bmin[a] <= inner.bmin[a] && outer.bmax[a] >= inner.bmax[a];
        return ok;
    };
    out.leaf_parent.resize ( nt ); out.leaf_mask.resize ( nt );
    for ( size_t k = 0; k < nt; ++k ) {
        const uint32_t soup = soup_of_fast[k];
        out.leaf_parent[k] = leaf_parent_of_soup[soup];
        float lo[3], hi[3];
        for ( int a = 0; a < 3; ++a ) { lo[a] = std::min ( tris[soup].a[a], std::min ( tris[soup].b[a], tris[soup].c[a] ) ); hi[a] = std::max ( tris[soup].a[a], std::max ( tris[soup].b[a], tris[soup].c[a] ) ); }
        uint32_t mask = 0, level = 0, below = 0;
        for ( uint32_t q = leaf_parent_of_soup[soup]; q != 0u; ++level ) {
            const DevReplay& r = out.replay[q];
            bool clear = true;
            for ( int a = 0; a < 3; ++a ) clear = clear && ( lo[a] - r.bmin[a] >= margin ) && ( r.bmax[a] - hi[a] >= margin );      // (false for NaN)
            if ( !clear && level > 0 ) clear = contains ( r, out.replay[below] );
            if ( !clear ) { mask |= 1u << ( level < 31u ? level : 31u ); ++out.replayed; }
            ++out.levels;
            below = q; q = r.parent;
        }
        out.leaf_mask[k] = mask;
    }
}

// DevScene::sincos24, one table per device for the life of the process (128 MB of its 288 GB; filled by tdm_sincosf_pair itself, ~1 ms). nullptr when it cannot
// be had (or TERRA_AMD_NO_SINCOS_TABLE is set: A/B runs) -- the kernels then compute.
static std::atomic<bool> g_azimuth_table { true };
extern "C" void terra_amd_set_azimuth_table ( int on ) { g_azimuth_table.store ( on != 0, std::memory_order_relaxed ); }      // (takes effect at the next commit)
static const float2* sincos_table_of ( int device ) {
    static std::mutex lock; static const float2* tables[64]; static bool tried[64];
    if ( device < 0 || device >= 64 || !g_azimuth_table.load ( std::memory_order_relaxed ) ) return nullptr;
    std::lock_guard<std::mutex> g ( lock );
    if ( tried[device] ) return tables[device];
    tried[device] = true;
    float2* t = nullptr;
    if ( hipMalloc ( ( void** ) &t, sizeof ( float2 ) << 24 ) != hipSuccess ) { ( void ) hipGetLastError(); return nullptr; }
    if ( terra_fill_sincos24 ( t, nullptr ) != hipSuccess || hipDeviceSynchronize() != hipSuccess ) { ( void ) hipGetLastError(); ( void ) hipFree ( t ); return nullptr; }
    tables[device] = t;
    return t;
}

// Can this scene, as it stands, be rendered by this library? The reference runs ANY host callback (TerraBSDF::sample / pdf / eval, TerraAttribute::eval:
// src/Terra.c:1071-1075, 1804-1810); the device runs the presets of TerraPresets.h and texture lookups only, and there is no CPU path here. A client that
// supports custom callbacks asks before it commits and keeps such scenes on the reference renderer. (upload_scene applies the same rules and fails the commit.)
static int scene_support ( const Scene* s, std::string& why ) {
    char b[256];
    if ( s->objects_pop > 256 ) { snprintf ( b, sizeof b, "%zu objects: the primitive reference holds 8 bits of object index (include/Terra.h:195-198)", s->objects_pop ); why = b; return kTerraAmdErrUnsupported; }
    auto attr_ok = [&] ( const TerraAttribute & at, size_t j, const char* what ) {
        if ( at.state == nullptr ) return true;
        if ( at.eval != terra_texture_sample ) { snprintf ( b, sizeof b, "object %zu %s: an attribute callback other than terra_texture_sample (host code cannot run on the device)", j, what ); why = b; return false; }
        const TerraTexture* t = ( const TerraTexture* ) at.state;
        if ( !t->pixels || !t->width || !t->height || ( t->depth != 1 && t->depth != 4 ) || t->components == 0 ) { snprintf ( b, sizeof b, "object %zu %s: invalid texture", j, what ); why = b; return false; }
        return true;
    };
    for ( size_t j = 0; j < s->objects_pop; ++j ) {
        const TerraMaterial& m = s->objects[j].material;
        if ( ! ( is_diffuse ( m.bsdf ) || is_phong ( m.bsdf ) || is_ggx ( m.bsdf ) || is_glass ( m.bsdf ) ) ) {
            snprintf ( b, sizeof b, "object %zu: its BSDF's sample / pdf / eval are not a terra_bsdf_*_init preset of this library (host callbacks cannot run on the device)", j ); why = b; return kTerraAmdErrUnsupported;
        }
        if ( m.attributes_count > TERRA_MATERIAL_MAX_ATTRIBUTES ) { snprintf ( b, sizeof b, "object %zu: attributes_count %zu > %d", j, m.attributes_count, TERRA_MATERIAL_MAX_ATTRIBUTES ); why = b; return kTerraAmdErrBadArgument; }
        if ( !attr_ok ( m.emissive, j, "emissive" ) ) return kTerraAmdErrUnsupported;
        for ( size_t a = 0; a < m.attributes_count; ++a ) { char w[32]; snprintf ( w, sizeof w, "attribute %zu", a ); if ( !attr_ok ( m.attributes[a], j, w ) ) return kTerraAmdErrUnsupported; }
    }
    if ( s->env_lighting ) {
        const TerraAttribute& env = s->new_opts.environment_map;
        if ( env.state != nullptr && env.eval != terra_texture_sample_latlong ) { why = "environment: with environment lighting on, the attribute must be a constant or terra_attribute_init_cubemap"; return kTerraAmdErrUnsupported; }
    }
    if ( terra_amd_device_count() <= 0 ) { why = "no HIP device visible"; return kTerraAmdErrNoDevice; }
    return 0;
}
extern "C" int terra_amd_scene_supported ( HTerraScene h, char* why, size_t capacity ) {
    std::string w;
    const int st = scene_support ( S ( h ), w );
    if ( why && capacity ) snprintf ( why, capacity, "%s", w.c_str() );
    return st;          // (a query: nothing is recorded in the error channels)
}

// validates that every material can run on the device and uploads the flattened scene
static int upload_scene ( Scene* s ) {
    const size_t nobj = s->objects_pop;
    if ( nobj > 256 ) return fail ( kTerraAmdErrUnsupported, "%zu objects: the primitive reference holds 8 bits of object index (include/Terra.h:195-198)", nobj );
    size_t ntri = 0;
    s->first_tri.assign ( nobj, 0 );
    for ( size_t j = 0; j < nobj; ++j ) { s->first_tri[j] = ( uint32_t ) ntri; ntri += s->objects[j].triangles_count; }
    if ( ntri >= 0x7fffffffu ) return fail ( kTerraAmdErrUnsupported, "too many triangles" );
    std::vector<const TerraTexture*> textures;
    std::vector<DevMaterial> mats ( nobj ? nobj : 1 );
    memset ( mats.data(), 0, mats.size() * sizeof ( DevMaterial ) );
    for ( size_t j = 0; j < nobj; ++j ) {
        const TerraMaterial& m = s->objects[j].material;
        DevMaterial& d = mats[j];
        if ( is_diffuse ( m.bsdf ) ) d.bsdf = kDevBsdfDiffuse;
        else if ( is_phong ( m.bsdf ) ) d.bsdf = kDevBsdfPhong;
        else if ( is_ggx ( m.bsdf ) ) d.bsdf = kDevBsdfGGX;
        else if ( is_glass ( m.bsdf ) ) d.bsdf = kDevBsdfGlass;
        else return fail ( kTerraAmdErrUnsupported, "object %zu: BSDF function pointers are not a terra_bsdf_*_init preset of this library; host callbacks cannot run on the device", j );
        if ( m.attributes_count > TERRA_MATERIAL_MAX_ATTRIBUTES ) return fail ( kTerraAmdErrBadArgument, "object %zu: attributes_count %zu > %d", j, m.attributes_count, TERRA_MATERIAL_MAX_ATTRIBUTES );
        for ( int a = 0; a <= TERRA_DEV_MAX_ATTR; ++a ) d.tex[a] = -1;
        d.any_texture = 0;
        // an attribute is a constant (state == NULL) or a texture sampled with this library's terra_texture_sample
        // (reference src/Terra.c:294-298, 1804-1810); anything else is a host callback the device cannot run
        auto bind = [&] ( const TerraAttribute & at, int slot, const char* what ) -> int {
            if ( at.state == nullptr ) return 0;
            if ( at.eval != terra_texture_sample ) return fail ( kTerraAmdErrUnsupported, "object %zu %s: attribute callbacks other than terra_texture_sample cannot run on the device (lat-long lookups of a material attribute read past the texcoord in the reference, src/Terra.c:468-471)", j, what );
            const TerraTexture* t = ( const TerraTexture* ) at.state;
            if ( !t->pixels || !t->width || !t->height || ( t->depth != 1 && t->depth != 4 ) || t->components == 0 ) return fail ( kTerraAmdErrBadArgument, "object %zu %s: invalid texture", j, what );
            size_t k = 0;
            for ( ; k < textures.size(); ++k ) if ( textures[k] == t ) break;
            if ( k == textures.size() ) textures.push_back ( t );
            d.tex[slot] = ( int32_t ) k; d.any_texture = 1;
            return 0;
        };
        if ( int rc = bind ( m.emissive, TERRA_DEV_MAX_ATTR, "emissive" ) ) return rc;
        for ( size_t a = 0; a < m.attributes_count; ++a ) {
            char what[32]; snprintf ( what, sizeof what, "attribute %zu", a );
            if ( int rc = bind ( m.attributes[a], ( int ) a, what ) ) return rc;
            d.attributes[a][0] = m.attributes[a].value.x; d.attributes[a][1] = m.attributes[a].value.y; d.attributes[a][2] = m.attributes[a].value.z;
        }
        d.attributes_count = ( uint32_t ) m.attributes_count;
        d.ior = m.ior;
        d.first_tri = s->first_tri[j]; d.tri_count = ( uint32_t ) s->objects[j].triangles_count;
        d.emissive[0] = m.emissive.value.x; d.emissive[1] = m.emissive.value.y; d.emissive[2] = m.emissive.value.z;
    }
    s->uniform_attr_count = nobj ? ( int ) mats[0].attributes_count : -1;
    for ( size_t j = 1; j < nobj; ++j ) if ( ( int ) mats[j].attributes_count != s->uniform_attr_count ) s->uniform_attr_count = -1;
    s->bsdf_kinds = 0;
    for ( size_t j = 0; j < nobj; ++j ) { s->bsdf_kinds |= 1u << mats[j].bsdf; if ( mats[j].any_texture ) s->bsdf_kinds |= TERRA_KIND_TEX; }
    // environment (extension, off by default): constant colour or a lat-long lookup (reference src/Terra.c:468-477)
    int32_t env_mode = 0, env_tex = -1; float env_color[3] = { 0.f, 0.f, 0.f };
    if ( s->env_lighting ) {
        const TerraAttribute& env = s->opts.environment_map;
        if ( env.state == nullptr ) {
            env_mode = 1; env_color[0] = env.value.x; env_color[1] = env.value.y; env_color[2] = env.value.z;
        } else {
            if ( env.eval != terra_texture_sample_latlong ) return fail ( kTerraAmdErrUnsupported, "environment: with environment lighting on, the attribute must be a constant or terra_attribute_init_cubemap (a lat-long lookup by direction)" );
            const TerraTexture* t = ( const TerraTexture* ) env.state;
            if ( !t->pixels || !t->width || !t->height || ( t->depth != 1 && t->depth != 4 ) || t->components == 0 ) return fail ( kTerraAmdErrBadArgument, "environment: invalid texture" );
            size_t k = 0;
            for ( ; k < textures.size(); ++k ) if ( textures[k] == t ) break;
            if ( k == textures.size() ) textures.push_back ( t );
            env_mode = 2; env_tex = ( int32_t ) k;
        }
        s->bsdf_kinds |= TERRA_KIND_ENV;
    }
    // flatten
    std::vector<DevTri> tris ( ntri ? ntri : 1 );
    std::vector<DevProps> props ( ntri ? ntri : 1 );
    std::vector<float> tri_area ( ntri ? ntri : 1, 0.f );
    for ( size_t j = 0; j < nobj; ++j ) for ( size_t i = 0; i < s->objects[j].triangles_count; ++i ) {
        const TerraTriangle& t = s->objects[j].triangles[i]; const TerraTriangleProperties& q = s->objects[j].properties[i];
        DevTri& d = tris[s->first_tri[j] + i];
        d.a[0] = t.a.x; d.a[1] = t.a.y; d.a[2] = t.a.z; d.object = ( uint32_t ) j;
        d.b[0] = t.b.x; d.b[1] = t.b.y; d.b[2] = t.b.z; d.tri_in_object = ( uint32_t ) i;
        d.c[0] = t.c.x; d.c[1] = t.c.y; d.c[2] = t.c.z; d.pad = 0;
        DevProps& e = props[s->first_tri[j] + i];
        memcpy ( e.na, &q.normal_a, 12 ); memcpy ( e.nb, &q.normal_b, 12 ); memcpy ( e.nc, &q.normal_c, 12 );
        memcpy ( e.ta, &q.texcoord_a, 8 ); memcpy ( e.tb, &q.texcoord_b, 8 ); memcpy ( e.tc, &q.texcoord_c, 8 );
        e.pad = 0.f;
    }
    std::vector<DevLight> lights ( s->lights.size() ? s->lights.size() : 1 );
    for ( size_t l = 0; l < s->lights.size(); ++l ) {
        const uint32_t o = s->lights[l].object;
        lights[l].object = o; lights[l].first_tri = s->first_tri[o]; lights[l].tri_count = ( uint32_t ) s->objects[o].triangles_count; lights[l].area = s->lights[l].area;
        for ( size_t i = 0; i < s->objects[o].triangles_count; ++i ) tri_area[s->first_tri[o] + i] = triangle_area ( s->objects[o].triangles[i] );
    }
    // Device numbering is breadth first (child 0 before child 1), so the top of the tree is the
    // contiguous prefix the kernel stages in LDS. Traversal order depends on the tree's shape and
    // child order only, never on node numbers, so results are unchanged.
    std::vector<uint32_t> bfs_of ( s->nodes.size(), 0 ), order;
    order.reserve ( s->nodes.size() );
    if ( !s->nodes.empty() ) {
        order.push_back ( 0 );
        for ( size_t head = 0; head < order.size(); ++head ) {
            const HostNode& h = s->nodes[order[head]];
            for ( int c = 0; c < 2; ++c ) if ( h.type[c] == -1 ) { bfs_of[ ( size_t ) h.index[c]] = ( uint32_t ) order.size(); order.push_back ( ( uint32_t ) h.index[c] ); }
        }
    }
    std::vector<DevNode> nodes ( s->nodes.size() );
    for ( size_t k = 0; k < order.size(); ++k ) {
        const HostNode& h = s->nodes[order[k]]; DevNode& d = nodes[k];
        memcpy ( d.min0, &h.aabb[0].min, 12 ); memcpy ( d.max0, &h.aabb[0].max, 12 );
        memcpy ( d.min1, &h.aabb[1].min, 12 ); memcpy ( d.max1, &h.aabb[1].max, 12 );
        for ( int c = 0; c < 2; ++c ) {
            if ( h.type[c] == -1 ) { d.child[c] = bfs_of[ ( size_t ) h.index[c]]; d.prim[c] = 0; }
            else if ( h.type[c] == 1 ) {
                uint32_t obj = ( uint32_t ) h.index[c] & 0xffu, tri = ( uint32_t ) h.index[c] >> 8;
                d.child[c] = DEV_CHILD_LEAF | ( s->first_tri[obj] + tri ); d.prim[c] = ( uint32_t ) h.index[c];
            } else { d.child[c] = DEV_CHILD_EMPTY; d.prim[c] = 0; }
        }
    }
    // test hook (tests/test_gpu_render.py "reachability"): shrink the DEVICE copy of the reference tree's boxes, so that the reference traversal -- as the device replays
    // it -- misses triangles the watertight test would hit, the situation the reachability replay exists for and that float rounding alone produces too rarely to test
    if ( s->test_shrink_reference_boxes > 0.f ) {
        const float g = s->test_shrink_reference_boxes;
        for ( DevNode& d : nodes ) for ( int a = 0; a < 3; ++a ) {
            if ( d.max0[a] - d.min0[a] > 2.f * g ) { d.min0[a] += g; d.max0[a] -= g; }
            if ( d.max1[a] - d.min1[a] > 2.f * g ) { d.min1[a] += g; d.max1[a] -= g; }
        }
    }
    // optional fast tree: same triangles, own node array and leaf-ordered soup with reference visit ranks
    std::vector<DevNode> fnodes; std::vector<DevTri> ftris; std::vector<uint32_t> rank_for_device;
    fastbvh::Wide fwide;                // the fast tree as traversed: 4-wide nodes of binary16 planes
    ReachTables reach_tabs;
    s->fast_nodes = 0; s->fast_max_stack = 1; s->fast_on_device = false;
    // traversal policy (see Scene::tree_mode and the containment check above)
    s->coord_max = 0.f; s->cull_ok = false; s->tree_note.clear();
    bool margin_ok = true;
    for ( size_t j = 0; j < nobj && margin_ok; ++j ) margin_ok = coords_within_margin ( &s->objects[j].triangles[0].a.x, s->objects[j].triangles_count * 9 );
    bool coords_finite = true;
    for ( size_t j = 0; j < nobj; ++j ) for ( size_t i = 0; i < s->objects[j].triangles_count * 9; ++i ) { float v = fabsf ( ( &s->objects[j].triangles[0].a.x ) [i] ); if ( v > s->coord_max ) s->coord_max = v; if ( !std::isfinite ( v ) ) coords_finite = false; }
    const bool resident = terra_scene_fits_lds ( ( uint32_t ) s->nodes.size(), ( uint32_t ) ntri, s->max_stack, ( uint32_t ) nobj, ( uint32_t ) s->lights.size() );
    const bool hooked = s->test_shrink_reference_boxes > 0.f;      // the containment proof below ran on the unshrunk boxes: no shortcut that rests on it
    bool auto_ok = false;
    if ( s->tree_mode == 2 ) {
        std::string why;
        if ( ntri < 2 ) s->tree_note = "fewer than 2 triangles: replica traversal";
        else if ( !margin_ok ) { char b[160]; snprintf ( b, sizeof b, "a vertex coordinate exceeds %.1f (largest %.6g): the 1e-4 box margin is not provably above rounding error, replica traversal", ( double ) TERRA_CULL_MAX_COORD, ( double ) s->coord_max ); s->tree_note = b; }
        else if ( hooked ) s->tree_note = "TEST HOOK terra_amd_debug_shrink_reference_boxes: replica traversal";
        else if ( !verify_reference_leaf_boxes ( s, why ) ) s->tree_note = why + ": replica traversal";
        else auto_ok = true;
    } else s->tree_note = s->tree_mode == 0 ? "replica traversal requested" : "fast tree requested";
    s->cull_ok = auto_ok;
    // Outside the range the reference's own slab test can numerically miss a box whose triangle the watertight test would hit, so the containment
    // shortcut is gone -- but not the fast tree: with its boxes inflated to the rounding bound it still finds every triangle the watertight test accepts,
    // and a hit stands only if the reference traversal would have reached it (its inner ancestors' slab tests, replayed exactly: for the closest of all hits first, and only
    // if that one fails -- float rounding makes it very rare -- for every candidate of a second pass; trace_device.h bvh_traverse_fast). Rays may start
    // up to TERRA_REACH_CAMERA_FACTOR x the scene's largest coordinate from the origin (the camera, checked per call); margin = 128 u x that limit.
    // LDS-resident scenes out of range keep the reference tree and its exact traversal of the inner nodes; their LEAF boxes -- which the reference never tests -- are rebuilt
    // around the triangle's extent with that same margin, so the leaf-box cull only skips triangle tests that cannot succeed: no replay needed (reach_cull).
    s->reach = false; s->reach_cull = false; s->reach_limit = 0.f;
    float reach_margin = 0.f;
    if ( s->tree_mode == 2 && !auto_ok && !margin_ok && ntri >= 2 && coords_finite && s->coord_max <= TERRA_REACH_MAX_COORD ) {
        s->reach_limit = TERRA_REACH_CAMERA_FACTOR * s->coord_max;
        reach_margin = 128.f * 5.9604645e-8f * s->reach_limit;
        char b[256];
        if ( resident ) {
            s->reach_cull = true; s->cull_ok = true;
            snprintf ( b, sizeof b, "largest coordinate %.6g exceeds %.1f: reference tree with the leaf-box cull on leaf boxes rebuilt with a margin of %.3g (scene is LDS-resident)", ( double ) s->coord_max, ( double ) TERRA_CULL_MAX_COORD, ( double ) reach_margin );
        } else {
            s->reach = true;
            snprintf ( b, sizeof b, "largest coordinate %.6g exceeds %.1f: fast tree with boxes inflated by %.3g, the reference's reachability replayed for the closest hit", ( double ) s->coord_max, ( double ) TERRA_CULL_MAX_COORD, ( double ) reach_margin );
        }
        s->tree_note = b;
    }
    if ( s->reach_cull ) {
        // every leaf child's box := its triangle's extent +- the margin (united with the stored box): a ray that hits the triangle passes this box's slab test by the
        // error bound above, at the scene's scale; inner boxes stay as they are (their tests ARE the reference's traversal)
        for ( DevNode& d : nodes ) for ( int c = 0; c < 2; ++c ) {
            if ( d.child[c] == DEV_CHILD_EMPTY || ! ( d.child[c] & DEV_CHILD_LEAF ) ) continue;
            const DevTri& t = tris[d.child[c] & 0x7fffffffu];
            float* mn = c ? d.min1 : d.min0; float* mx = c ? d.max1 : d.max0;
            for ( int a = 0; a < 3; ++a ) {
                const float lo = std::min ( t.a[a], std::min ( t.b[a], t.c[a] ) ) - reach_margin, hi = std::max ( t.a[a], std::max ( t.b[a], t.c[a] ) ) + reach_margin;
                mn[a] = std::min ( mn[a], lo ); mx[a] = std::max ( mx[a], hi );
            }
        }
    }
    s->use_fast = s->tree_mode == 1 || ( auto_ok && !resident ) || s->reach;
    const float fast_extra = ( s->reach && reach_margin > 1e-4f ) ? reach_margin - 1e-4f : 0.f;      // on top of the +-1e-4 of every triangle box
    // the fast tree's planes are stored as binary16 times a power of two that brings the largest of them (|coordinate| + box margin) below 2^14: binary16 reaches 65,504,
    // and a power of two changes no bit of a plane or a t value -- the kernel divides the ray's inverse direction by the same factor (DevScene::fast_inv_scale)
    float fast_scale = 1.f;
    { const float span = s->coord_max + 1e-4f + fast_extra; if ( std::isfinite ( span ) ) while ( span * fast_scale >= 16384.f && fast_scale > 0x1p-100f ) fast_scale *= 0.5f; }
    std::vector<uint32_t> soup_of_fast;                                                             // reach: soup index of every fast triangle (host-built: known here)
    if ( s->use_fast ) {
        // rank of every soup triangle in the reference traversal's leaf visit order (all boxes hit)
        std::vector<uint32_t> rank ( ntri ? ntri : 1, 0 );
        {
            uint32_t next = 0; std::vector<int> st; st.push_back ( 0 );
            while ( !st.empty() ) {
                const HostNode& h = s->nodes[ ( size_t ) st.back()]; st.pop_back();
                for ( int c = 0; c < 2; ++c ) {
                    if ( h.type[c] == 1 ) { uint32_t obj = ( uint32_t ) h.index[c] & 0xffu, tri = ( uint32_t ) h.index[c] >> 8; rank[s->first_tri[obj] + tri] = next++; }
                    else if ( h.type[c] == -1 ) st.push_back ( h.index[c] );
                }
            }
        }
        for ( size_t k = 0; k < ntri; ++k ) tris[k].pad = rank[k];      // the soup carries the ranks too: a light-sample ray tests its triangle before it traverses (trace_device.h fast_expect)
        s->fast_on_device = s->tree_builder == 1 && ntri > 64 && terra_amd_device_count() > 0;
        if ( s->fast_on_device ) {
            // built after the upload, from the soup already in HBM; the host only supplies the reference visit ranks
            rank_for_device.swap ( rank );
        } else {
            std::vector<fastbvh::Prim> prims ( ntri );
            for ( size_t j = 0; j < nobj; ++j ) for ( size_t i = 0; i < s->objects[j].triangles_count; ++i ) {
                fastbvh::Prim& q = prims[s->first_tri[j] + i];
                q.box = bvh::empty_box(); bvh::grow_by_triangle ( q.box, s->objects[j].triangles[i] );
                if ( fast_extra > 0.f ) { const float g = fast_extra; q.box.min.x -= g; q.box.min.y -= g; q.box.min.z -= g; q.box.max.x += g; q.box.max.y += g; q.box.max.z += g; }
                q.c[0] = 0.5f * ( q.box.min.x + q.box.max.x ); q.c[1] = 0.5f * ( q.box.min.y + q.box.max.y ); q.c[2] = 0.5f * ( q.box.min.z + q.box.max.z );
                q.soup = ( uint32_t ) ( s->first_tri[j] + i );
            }
            double t_phase = now_s();
            fastbvh::Built built = fastbvh::build ( prims );        // (reorders prims into leaf order)
            phase ( "fast tree (host)", t_phase );
            if ( timing_on() ) fprintf ( stderr, "[terra_amd timing]   fast tree: %zu nodes, stack %d entries\n", built.nodes.size(), built.max_stack );
            std::string why;
            std::vector<TerraAABB> leaf_boxes ( prims.size() );
            for ( size_t k = 0; k < prims.size(); ++k ) leaf_boxes[k] = prims[k].box;
            if ( s->tree_mode == 2 && !verify_fast_tree ( built.nodes, leaf_boxes, why ) ) {       // cannot happen with this builder (plain unions); checked because the culling relies on it
                s->use_fast = false; s->tree_note = why + ": reference tree";
                built.nodes.clear(); built.order.clear();
            }
            fnodes.swap ( built.nodes );
            if ( s->use_fast ) {
                double t_w = now_s();
                fwide = fastbvh::widen ( fnodes, fast_scale );
                phase ( "  4-wide binary16 nodes", t_w );
            }
            ftris.resize ( ntri ? ntri : 1 );
            for ( size_t k = 0; k < built.order.size(); ++k ) { ftris[k] = tris[built.order[k]]; ftris[k].pad = rank[built.order[k]]; }
            if ( s->reach && s->use_fast ) soup_of_fast = built.order;
            s->fast_nodes = ( uint32_t ) fwide.nodes.size(); s->fast_max_stack = fwide.max_stack;
        }
    }
    if ( s->tree_mode == 2 && s->reach && !s->use_fast ) { s->reach = false; }
    // the stack of a fast-tree launch keeps its first entries in LDS and the rest in HBM (terra_plan_fast_tree: 4 bytes per entry and resident lane), so depth is no longer
    // a reason to give the fast tree up -- short of a degenerate tree whose worst case would ask for gigabytes of spill space
    auto fast_stack_fits = [] ( int depth ) { return depth <= TERRA_FAST_STACK_MAX; };
    if ( s->use_fast && !s->fast_on_device && !fast_stack_fits ( s->fast_max_stack ) ) {
        char b[200]; snprintf ( b, sizeof b, "fast tree needs a traversal stack of %d entries (limit %d): %s", s->fast_max_stack, TERRA_FAST_STACK_MAX, s->cull_ok ? "reference tree with the leaf-box cull (global memory)" : "reference tree, replica traversal" );
        s->use_fast = false; s->reach = false; s->tree_note = b; fnodes.clear(); fwide.nodes.clear(); ftris.clear(); soup_of_fast.clear(); s->fast_nodes = 0; s->fast_max_stack = 1;
    }
    if ( s->tree_mode == 2 && auto_ok ) s->tree_note = s->use_fast ? ( s->fast_on_device ? "containment verified: fast tree built on the device (LBVH; scene is not LDS-resident)" : "containment verified: fast tree (scene is not LDS-resident)" ) : ( resident ? "containment verified: reference tree with the leaf-box cull (scene is LDS-resident)" : s->tree_note );
    if ( s->reach && !s->fast_on_device ) build_reach_tables ( nodes, tris, soup_of_fast, reach_margin, reach_tabs );
    // The automatic mode's fallbacks are correct and 10-20 x slower on scenes of this size (hall: 110 against 2,400 Msamples/s): say so where a client looks, once per commit.
    if ( s->tree_mode == 2 && !s->use_fast && !resident && ntri >= 2 )
        fprintf ( stderr, "[terra_amd] warning: this scene (%zu triangles) is traversed through the reference tree%s, typically 10-20 x slower than the fast tree -- %s\n", ntri,
                  s->cull_ok ? " with the leaf-box cull" : " decision by decision", s->tree_note.c_str() );
    // one blob, 256-byte aligned sections
    auto align = [] ( size_t v ) { return ( v + 255 ) & ~size_t ( 255 ); };
    size_t o_nodes = 0, o_tris = align ( o_nodes + nodes.size() * sizeof ( DevNode ) ), o_props = align ( o_tris + tris.size() * sizeof ( DevTri ) );
    size_t o_mats = align ( o_props + props.size() * sizeof ( DevProps ) ), o_lights = align ( o_mats + mats.size() * sizeof ( DevMaterial ) );
    size_t o_area = align ( o_lights + lights.size() * sizeof ( DevLight ) ), o_fn = align ( o_area + tri_area.size() * sizeof ( float ) );
    // o_fn: the device builder's output (a binary tree of at most n - 1 nodes; host-built trees do not need it on the device), o_fh: the wide nodes the kernels traverse
    const size_t fn_cap = s->fast_on_device ? ntri : 0, fh_cap = s->fast_on_device ? ntri : fwide.nodes.size(), ft_cap = s->fast_on_device ? ntri : ftris.size();
    size_t o_fh = align ( o_fn + fn_cap * sizeof ( DevNode ) ), o_ft = align ( o_fh + fh_cap * sizeof ( DevFastNode ) ), o_td = align ( o_ft + ft_cap * sizeof ( DevTri ) );
    const size_t n_replay = s->reach ? ( nodes.size() ? nodes.size() : 1 ) : 0, n_reach_tri = s->reach ? ntri : 0;
    const size_t o_rp = o_td, o_lp = align ( o_rp + n_replay * sizeof ( DevReplay ) ), o_lm = align ( o_lp + n_reach_tri * 4 ); o_td = align ( o_lm + n_reach_tri * 4 );
    std::vector<DevTexture> tdesc ( textures.size() );
    std::vector<size_t> tex_off ( textures.size() );
    size_t total = align ( o_td + tdesc.size() * sizeof ( DevTexture ) );
    for ( size_t k = 0; k < textures.size(); ++k ) {
        const TerraTexture* t = textures[k];
        tex_off[k] = total;
        total = align ( total + ( ( size_t ) t->width * t->height * t->components + 2 ) * t->depth );      // +2 elements: the 3-component read of the last texel
    }

    release_device ( s );
    s->device = g_device;
    if ( terra_amd_device_count() <= 0 ) return fail ( kTerraAmdErrNoDevice, "no HIP device visible: terra_scene_commit built the host tree but cannot upload; terra_render will fail" );
    HIP_TRY ( hipSetDevice ( s->device ), kTerraAmdErrNoDevice );
    HIP_TRY ( hipMalloc ( &s->d_blob, total ), kTerraAmdErrNoDevice );
    HIP_TRY ( hipMemset ( s->d_blob, 0, total ), kTerraAmdErrNoDevice );
    HIP_TRY ( hipMalloc ( ( void** ) &s->d_counters, kCtrCount * sizeof ( unsigned long long ) ), kTerraAmdErrNoDevice );
    HIP_TRY ( hipMemset ( s->d_counters, 0, kCtrCount * sizeof ( unsigned long long ) ), kTerraAmdErrNoDevice );
    char* base = ( char* ) s->d_blob;
    HIP_TRY ( hipMemcpy ( base + o_nodes, nodes.data(), nodes.size() * sizeof ( DevNode ), hipMemcpyHostToDevice ), kTerraAmdErrNoDevice );
    HIP_TRY ( hipMemcpy ( base + o_tris, tris.data(), tris.size() * sizeof ( DevTri ), hipMemcpyHostToDevice ), kTerraAmdErrNoDevice );
    HIP_TRY ( hipMemcpy ( base + o_props, props.data(), props.size() * sizeof ( DevProps ), hipMemcpyHostToDevice ), kTerraAmdErrNoDevice );
    HIP_TRY ( hipMemcpy ( base + o_mats, mats.data(), mats.size() * sizeof ( DevMaterial ), hipMemcpyHostToDevice ), kTerraAmdErrNoDevice );
    HIP_TRY ( hipMemcpy ( base + o_lights, lights.data(), lights.size() * sizeof ( DevLight ), hipMemcpyHostToDevice ), kTerraAmdErrNoDevice );
    HIP_TRY ( hipMemcpy ( base + o_area, tri_area.data(), tri_area.size() * sizeof ( float ), hipMemcpyHostToDevice ), kTerraAmdErrNoDevice );
    if ( !fwide.nodes.empty() ) {
        HIP_TRY ( hipMemcpy ( base + o_fh, fwide.nodes.data(), fwide.nodes.size() * sizeof ( DevFastNode ), hipMemcpyHostToDevice ), kTerraAmdErrNoDevice );
        HIP_TRY ( hipMemcpy ( base + o_ft, ftris.data(), ftris.size() * sizeof ( DevTri ), hipMemcpyHostToDevice ), kTerraAmdErrNoDevice );
    }
    auto upload_reach_tables = [&] () -> int {
        HIP_TRY ( hipMemcpy ( base + o_rp, reach_tabs.replay.data(), n_replay * sizeof ( DevReplay ), hipMemcpyHostToDevice ), kTerraAmdErrNoDevice );
        HIP_TRY ( hipMemcpy ( base + o_lp, reach_tabs.leaf_parent.data(), n_reach_tri * 4, hipMemcpyHostToDevice ), kTerraAmdErrNoDevice );
        HIP_TRY ( hipMemcpy ( base + o_lm, reach_tabs.leaf_mask.data(), n_reach_tri * 4, hipMemcpyHostToDevice ), kTerraAmdErrNoDevice );
        if ( timing_on() ) fprintf ( stderr, "[terra_amd timing]   reachability: %llu ancestor levels, %llu to replay (%.2f per triangle)\n", ( unsigned long long ) reach_tabs.levels, ( unsigned long long ) reach_tabs.replayed, ntri ? ( double ) reach_tabs.replayed / ( double ) ntri : 0. );
        return 0;
    };
    if ( s->reach && !s->fast_on_device ) { if ( int rc = upload_reach_tables() ) return rc; }
    bool have_fast = !fwide.nodes.empty();
    if ( s->fast_on_device ) {
        double t_phase = now_s();
        uint32_t* d_rank = nullptr;
        HIP_TRY ( hipMalloc ( ( void** ) &d_rank, ntri * sizeof ( uint32_t ) ), kTerraAmdErrNoDevice );
        hipError_t e = hipMemcpy ( d_rank, rank_for_device.data(), ntri * sizeof ( uint32_t ), hipMemcpyHostToDevice );
        uint32_t built_nodes = 0; int built_stack = 1;
        if ( e == hipSuccess ) e = terra_build_fast_tree_device ( ( const DevTri* ) ( base + o_tris ), d_rank, ( uint32_t ) ntri, fast_extra, ( DevNode* ) ( base + o_fn ), ( DevTri* ) ( base + o_ft ), &built_nodes, &built_stack, nullptr );
        ( void ) hipFree ( d_rank );
        if ( e != hipSuccess ) return fail ( kTerraAmdErrLaunch, "device tree build: %s", hipGetErrorString ( e ) );
        phase ( "fast tree (device LBVH)", t_phase );
        // the binary tree comes back to the host: it is checked like a host-built one (automatic mode: the culling relies on containment) and made 4-wide there
        std::vector<DevNode> rn ( built_nodes );
        HIP_TRY ( hipMemcpy ( rn.data(), base + o_fn, rn.size() * sizeof ( DevNode ), hipMemcpyDeviceToHost ), kTerraAmdErrNoDevice );
        {   // (every device-built tree is checked, whatever the tree mode: the read-back is needed for the wide nodes anyway)
            std::vector<DevTri> rt ( ntri );
            HIP_TRY ( hipMemcpy ( rt.data(), base + o_ft, rt.size() * sizeof ( DevTri ), hipMemcpyDeviceToHost ), kTerraAmdErrNoDevice );
            std::vector<TerraAABB> leaf_boxes ( ntri );
            for ( size_t k = 0; k < ntri; ++k ) {
                TerraTriangle t; memcpy ( &t.a, rt[k].a, 12 ); memcpy ( &t.b, rt[k].b, 12 ); memcpy ( &t.c, rt[k].c, 12 );
                leaf_boxes[k] = bvh::empty_box(); bvh::grow_by_triangle ( leaf_boxes[k], t );
                if ( fast_extra > 0.f ) { const float g = fast_extra; TerraAABB& q = leaf_boxes[k]; q.min.x -= g; q.min.y -= g; q.min.z -= g; q.max.x += g; q.max.y += g; q.max.z += g; }
            }
            std::string why;
            if ( !verify_fast_tree ( rn, leaf_boxes, why ) ) return fail ( kTerraAmdErrLaunch, "device tree build: %s", why.c_str() );
            phase ( "  read back + containment check", t_phase );
            if ( s->reach ) {          // the replay tables need the order the device put the triangles in
                soup_of_fast.resize ( ntri );
                for ( size_t k = 0; k < ntri; ++k ) soup_of_fast[k] = s->first_tri[rt[k].object] + rt[k].tri_in_object;
                build_reach_tables ( nodes, tris, soup_of_fast, reach_margin, reach_tabs );
                if ( int rc = upload_reach_tables() ) return rc;
                phase ( "  reachability tables", t_phase );
            }
        }
        fwide = fastbvh::widen ( rn, fast_scale );
        phase ( "  4-wide binary16 nodes", t_phase );
        s->fast_nodes = ( uint32_t ) fwide.nodes.size(); s->fast_max_stack = fwide.max_stack; have_fast = true;
        if ( !fast_stack_fits ( fwide.max_stack ) ) {          // (a Morton-ordered tree over coincident geometry can be thousands of levels deep)
            char b[240]; snprintf ( b, sizeof b, "device-built fast tree needs a traversal stack of %d entries (limit %d): %s", fwide.max_stack, TERRA_FAST_STACK_MAX, s->cull_ok ? "reference tree with the leaf-box cull (global memory)" : "reference tree, replica traversal" );
            s->use_fast = false; s->reach = false; s->fast_on_device = false; s->tree_note = b; s->fast_nodes = 0; s->fast_max_stack = 1; have_fast = false;
            if ( s->tree_mode == 2 && auto_ok && resident ) s->tree_note = "containment verified: reference tree with the leaf-box cull (scene is LDS-resident)";
        } else HIP_TRY ( hipMemcpy ( base + o_fh, fwide.nodes.data(), fwide.nodes.size() * sizeof ( DevFastNode ), hipMemcpyHostToDevice ), kTerraAmdErrNoDevice );
    }
    if ( have_fast && s->fast_nodes >= ( 1u << 25 ) ) return fail ( kTerraAmdErrUnsupported, "fast tree of %u nodes: the kernels address the 128-byte nodes by a 32-bit byte offset (at most 2^25 nodes)", s->fast_nodes );
    for ( size_t k = 0; k < textures.size(); ++k ) {
        const TerraTexture* t = textures[k];
        HIP_TRY ( hipMemcpy ( base + tex_off[k], t->pixels, ( size_t ) t->width * t->height * t->components * t->depth, hipMemcpyHostToDevice ), kTerraAmdErrNoDevice );
        tdesc[k].data = base + tex_off[k]; tdesc[k].width = t->width; tdesc[k].height = t->height; tdesc[k].components = t->components;
        tdesc[k].depth = t->depth; tdesc[k].filter = t->filter; tdesc[k].address_mode = t->address_mode;
    }
    if ( !tdesc.empty() ) HIP_TRY ( hipMemcpy ( base + o_td, tdesc.data(), tdesc.size() * sizeof ( DevTexture ), hipMemcpyHostToDevice ), kTerraAmdErrNoDevice );
    s->dev.textures = tdesc.empty() ? nullptr : ( const DevTexture* ) ( base + o_td );
    s->d_bytes = total; s->blob_bytes = total; s->o_tdesc = o_td; s->tdesc_host = tdesc; s->env_dist_floats = 0;
    s->dev.nodes = ( const DevNode* ) ( base + o_nodes ); s->dev.tris = ( const DevTri* ) ( base + o_tris ); s->dev.props = ( const DevProps* ) ( base + o_props );
    s->dev.mats = ( const DevMaterial* ) ( base + o_mats ); s->dev.lights = ( const DevLight* ) ( base + o_lights ); s->dev.tri_area = ( const float* ) ( base + o_area );
    s->dev.n_nodes = ( uint32_t ) nodes.size(); s->dev.n_tris = ( uint32_t ) ntri; s->dev.n_objects = ( uint32_t ) nobj; s->dev.n_lights = ( uint32_t ) s->lights.size();
    s->dev.lights_triangles_count = ( uint32_t ) s->lights_triangles_count; s->dev.max_stack = s->max_stack;
    s->dev.fast_nodes = ( have_fast && s->fast_on_device ) ? ( const DevNode* ) ( base + o_fn ) : nullptr; s->dev.fast_nodes_h = have_fast ? ( const DevFastNode* ) ( base + o_fh ) : nullptr; s->dev.fast_tris = have_fast ? ( const DevTri* ) ( base + o_ft ) : nullptr;
    s->dev.n_fast_nodes = s->fast_nodes; s->dev.fast_max_stack = s->fast_max_stack; s->dev.fast_inv_scale = 1.f / fast_scale;
    s->dev.reach = ( s->reach && have_fast && !reach_tabs.leaf_parent.empty() ) ? 1u : 0u;
    s->dev.ref_replay = s->dev.reach ? ( const DevReplay* ) ( base + o_rp ) : nullptr; s->dev.fast_leaf_parent = s->dev.reach ? ( const uint32_t* ) ( base + o_lp ) : nullptr; s->dev.fast_leaf_mask = s->dev.reach ? ( const uint32_t* ) ( base + o_lm ) : nullptr;
    s->dev.env_mode = env_mode; s->dev.env_tex = env_tex; memcpy ( s->dev.env_color, env_color, sizeof env_color );
    s->dev.sincos24 = sincos_table_of ( s->device );
    // environment importance sampling (extension, UNPINNED; the oracle's env_table_build restates it): the map as the reference's TerraDistribution2D would hold it
    // (terra_distribution_2d_init, src/Terra.c:812-829: per row a running float sum in index order divided by its total, then the same over the rows' totals)
    if ( s->env_sampling && env_mode == 2 && textures[env_tex]->components >= 3 ) {        // (the table reads three components per texel)
        TerraTexture* t = const_cast<TerraTexture*> ( textures[env_tex] );
        const size_t nx = t->width, ny = t->height, cells = nx * ny;
        std::vector<float> tab ( 2 * cells + 2 * ny );
        float* f = tab.data(), * cdf = f + cells, * row_f = cdf + cells, * row_cdf = row_f + ny;
        bool mono = true;
        auto row_init = [&mono] ( const float* v, size_t n, float* c ) {
            float integral = 0.f;
            for ( size_t i = 0; i < n; ++i ) { mono = mono && v[i] >= 0.f; integral += v[i]; c[i] = integral; }
            for ( size_t i = 0; i < n; ++i ) c[i] /= integral;
            return integral;
        };
        for ( size_t y = 0; y < ny; ++y ) {
            const float sin_row = sinf ( ( ( float ) y + 0.5f ) / ( float ) ny * terra_PI );
            for ( size_t x = 0; x < nx; ++x ) {
                const TerraFloat3 c = terra_texture_read ( t, x, y );
                float lum = 0.2126f * c.x; lum += 0.7152f * c.y; lum += 0.0722f * c.z;
                f[y * nx + x] = lum * sin_row;
            }
            row_f[y] = row_init ( f + y * nx, nx, cdf + y * nx );
        }
        const float integral = row_init ( row_f, ny, row_cdf );
        HIP_TRY ( hipMalloc ( ( void** ) &s->d_env_dist, tab.size() * sizeof ( float ) ), kTerraAmdErrNoDevice );
        HIP_TRY ( hipMemcpy ( s->d_env_dist, tab.data(), tab.size() * sizeof ( float ), hipMemcpyHostToDevice ), kTerraAmdErrNoDevice );
        s->d_bytes += tab.size() * sizeof ( float ); s->env_dist_floats = tab.size();
        s->dev.env_f = s->d_env_dist; s->dev.env_cdf = s->d_env_dist + cells; s->dev.env_row_f = s->d_env_dist + 2 * cells; s->dev.env_row_cdf = s->d_env_dist + 2 * cells + ny;
        s->dev.env_nx = ( uint32_t ) nx; s->dev.env_ny = ( uint32_t ) ny; s->dev.env_integral = integral;
        s->dev.env_monotone = ( mono && integral > 0.f && integral <= FLT_MAX ) ? 1u : 0u;      // then every running sum is non-decreasing: bisection finds the scan's bucket
    }
    s->device_ok = true;
    return 0;
}

// The committed scene on the other devices of the set: the primary device's finished blob (trees built, boxes converted, tables filled) copied device to device,
// every pointer into it moved by the difference of the two base addresses, the texture descriptors -- which hold absolute addresses -- written again.
static int replicate_scene ( Scene* s ) {
    const char* base0 = ( const char* ) s->d_blob;
    for ( size_t k = 1; k < s->devices.size(); ++k ) {
        Scene::Replica r; r.device = s->devices[k];
        HIP_TRY ( hipSetDevice ( r.device ), kTerraAmdErrNoDevice );
        HIP_TRY ( hipMalloc ( &r.d_blob, s->blob_bytes ), kTerraAmdErrNoDevice );
        s->extra.push_back ( r );                                   // (owned from here on: release_device frees it whatever fails below)
        Scene::Replica& q = s->extra.back();
        HIP_TRY ( hipMemcpyPeer ( q.d_blob, q.device, s->d_blob, s->device, s->blob_bytes ), kTerraAmdErrNoDevice );
        HIP_TRY ( hipMalloc ( ( void** ) &q.d_counters, kCtrCount * sizeof ( unsigned long long ) ), kTerraAmdErrNoDevice );
        HIP_TRY ( hipMemset ( q.d_counters, 0, kCtrCount * sizeof ( unsigned long long ) ), kTerraAmdErrNoDevice );
        char* base = ( char* ) q.d_blob;
        auto move = [&] ( const void* p ) -> const void* { return p ? ( const void* ) ( base + ( ( const char* ) p - base0 ) ) : nullptr; };
        q.dev = s->dev;
        q.dev.nodes = ( const DevNode* ) move ( s->dev.nodes ); q.dev.tris = ( const DevTri* ) move ( s->dev.tris ); q.dev.props = ( const DevProps* ) move ( s->dev.props );
        q.dev.mats = ( const DevMaterial* ) move ( s->dev.mats ); q.dev.lights = ( const DevLight* ) move ( s->dev.lights ); q.dev.tri_area = ( const float* ) move ( s->dev.tri_area );
        q.dev.textures = ( const DevTexture* ) move ( s->dev.textures ); q.dev.fast_nodes = ( const DevNode* ) move ( s->dev.fast_nodes ); q.dev.fast_nodes_h = ( const DevFastNode* ) move ( s->dev.fast_nodes_h ); q.dev.fast_tris = ( const DevTri* ) move ( s->dev.fast_tris );
        q.dev.ref_replay = ( const DevReplay* ) move ( s->dev.ref_replay ); q.dev.fast_leaf_parent = ( const uint32_t* ) move ( s->dev.fast_leaf_parent ); q.dev.fast_leaf_mask = ( const uint32_t* ) move ( s->dev.fast_leaf_mask );
        if ( !s->tdesc_host.empty() ) {
            std::vector<DevTexture> td = s->tdesc_host;
            for ( DevTexture& t : td ) t.data = move ( t.data );
            HIP_TRY ( hipMemcpy ( base + s->o_tdesc, td.data(), td.size() * sizeof ( DevTexture ), hipMemcpyHostToDevice ), kTerraAmdErrNoDevice );
        }
        if ( s->d_env_dist && s->env_dist_floats ) {
            HIP_TRY ( hipMalloc ( ( void** ) &q.d_env_dist, s->env_dist_floats * sizeof ( float ) ), kTerraAmdErrNoDevice );
            HIP_TRY ( hipMemcpyPeer ( q.d_env_dist, q.device, s->d_env_dist, s->device, s->env_dist_floats * sizeof ( float ) ), kTerraAmdErrNoDevice );
            auto emove = [&] ( const float* p ) { return p ? q.d_env_dist + ( p - s->d_env_dist ) : nullptr; };
            q.dev.env_f = emove ( s->dev.env_f ); q.dev.env_cdf = emove ( s->dev.env_cdf ); q.dev.env_row_f = emove ( s->dev.env_row_f ); q.dev.env_row_cdf = emove ( s->dev.env_row_cdf );
        }
        q.dev.sincos24 = sincos_table_of ( q.device );
        // self-check, independent of the list above: no 8-byte word of the replica's scene record may still be an address inside the PRIMARY's blob (or its
        // environment tables) -- a pointer member added to DevScene and forgotten here would be exactly that, and on a single box it would even keep working
        {
            uint64_t words[ ( sizeof ( DevScene ) + 7 ) / 8] = { 0 }; memcpy ( words, &q.dev, sizeof ( DevScene ) );
            const uint64_t b0 = ( uint64_t ) ( uintptr_t ) base0, b1 = b0 + s->blob_bytes, e0 = ( uint64_t ) ( uintptr_t ) s->d_env_dist, e1 = e0 + s->env_dist_floats * sizeof ( float );
            for ( size_t i = 0; i < sizeof ( DevScene ) / 8; ++i )
                if ( ( words[i] >= b0 && words[i] < b1 ) || ( e0 && words[i] >= e0 && words[i] < e1 ) )
                    return fail ( kTerraAmdErrLaunch, "scene replica for device %d: byte %zu of its scene record still points into the primary device's copy (a pointer that replicate_scene does not rebase)", q.device, i * 8 );
        }
    }
    HIP_TRY ( hipSetDevice ( s->device ), kTerraAmdErrNoDevice );
    return 0;
}

extern "C" void terra_scene_commit ( HTerraScene h ) {
    Scene* s = S ( h );
    const bool rebuild = s->dirty_objects || s->opts.accelerator != s->new_opts.accelerator || s->nodes.empty();
    const bool env_changed = s->env_lighting && memcmp ( &s->opts.environment_map, &s->new_opts.environment_map, sizeof ( TerraAttribute ) ) != 0;
    s->opts = s->new_opts;
    double t_phase = now_s();
    if ( rebuild ) { bvh::build ( s->objects, s->objects_pop, s->nodes, s->max_stack ); phase ( "reference tree (host)", t_phase ); }
    const bool relight = s->dirty_lights || rebuild;
    if ( relight ) {
        s->lights.clear();
        s->lights_triangles_count = 0;      // the reference never resets this (src/Terra.c:228); see DESIGN.md deviations
        for ( size_t i = 0; i < s->objects_pop; ++i ) {
            const TerraAttribute& em = s->objects[i].material.emissive;
            TerraFloat3 e = em.value;
            if ( em.state != nullptr ) {            // reference src/Terra.c:199-200: evaluated at uv (0.5, 0.5)
                if ( em.eval != terra_texture_sample ) continue;       // rejected in upload_scene with a message
                TerraFloat2 uv = { 0.5f, 0.5f };
                e = terra_texture_sample ( em.state, &uv, nullptr );
            }
            if ( e.x == 0 && e.y == 0 && e.z == 0 ) continue;
            float area = 0;
            for ( size_t j = 0; j < s->objects[i].triangles_count; ++j ) area += triangle_area ( s->objects[i].triangles[j] );
            HostLight l; l.object = ( uint32_t ) i; l.area = area; l.power = terra_mulf3 ( &e, area * terra_PI );
            s->lights.push_back ( l );
            s->lights_triangles_count += s->objects[i].triangles_count;
        }
    }
    s->dirty_objects = false; s->dirty_lights = false;
    s->committed = true;
    s->commit_error.clear();
    // The environment attribute only ever scales a throughput that is then discarded (reference src/Terra.c:1053-1058:
    // the "Lo +=" is commented out), so by default neither a constant nor a textured environment reaches the image and
    // nothing is uploaded for it; terra_amd_set_environment_lighting(scene, 1) turns that line on (upload_scene binds it).
    // options travel as kernel arguments; geometry/material/light changes need a new replica
    std::vector<int> set;
    { std::lock_guard<std::mutex> g ( g_devices_lock ); set = g_devices; }
    if ( set.empty() ) set.push_back ( g_device );
    const bool set_changed = set != s->devices;          // (terra_amd_set_devices / terra_amd_set_device since the last commit: the replicas move)
    if ( rebuild || relight || env_changed || !s->device_ok || set_changed ) {
        if ( upload_scene ( s ) != 0 ) { s->commit_error = g_last_error; s->device_ok = false; return; }
        s->devices = set;
        if ( set.size() > 1 && replicate_scene ( s ) != 0 ) { s->commit_error = g_last_error; release_device ( s ); }
    }
}

extern "C" int terra_amd_scene_info ( HTerraScene h, TerraAmdSceneInfo* out ) {
    Scene* s = S ( h );
    if ( !s->committed ) return fail ( kTerraAmdErrNotCommitted, "scene not committed" );
    size_t ntri = 0; for ( size_t j = 0; j < s->objects_pop; ++j ) ntri += s->objects[j].triangles_count;
    out->triangles = ( uint32_t ) ntri; out->nodes = ( uint32_t ) s->nodes.size(); out->objects = ( uint32_t ) s->objects_pop; out->lights = ( uint32_t ) s->lights.size();
    out->lights_triangles_count = ( uint32_t ) s->lights_triangles_count; out->max_stack = s->max_stack; out->device_bytes = s->d_bytes;
    return 0;
}
extern "C" int terra_amd_scene_bvh_nodes ( HTerraScene h, void* out, int capacity ) {
    Scene* s = S ( h );
    if ( !s->committed ) return fail ( kTerraAmdErrNotCommitted, "scene not committed" );
    int n = ( int ) s->nodes.size();
    if ( out && capacity >= n ) memcpy ( out, s->nodes.data(), ( size_t ) n * sizeof ( HostNode ) );
    return n;
}
extern "C" int terra_amd_get_stats ( HTerraScene h, TerraAmdStats* out ) {
    Scene* s = S ( h );
    memset ( out, 0, sizeof *out );
    if ( !s->device_ok ) return fail ( kTerraAmdErrNotCommitted, "scene has no device replica" );
    unsigned long long c[kCtrCount];
    HIP_TRY ( hipSetDevice ( s->device ), kTerraAmdErrNoDevice );
    HIP_TRY ( hipMemcpy ( c, s->d_counters, sizeof c, hipMemcpyDeviceToHost ), kTerraAmdErrNoDevice );
    out->rays = c[kCtrRays]; out->nodes = c[kCtrNodes]; out->tri_tests = c[kCtrTriTests]; out->hits = c[kCtrHits];
    out->rand_calls = c[kCtrRandCalls]; out->attr_fetches = c[kCtrAttrFetches]; out->tri_culled = c[kCtrTriCulled];
    // derived exactly on the host (see Counters in trace_device.h)
    out->box_tests = s->dev.n_tris >= 2 ? ( s->use_fast ? 4 * out->nodes : s->cull_ok ? 2 * out->nodes : 2 * out->nodes - out->tri_tests ) : 0;      // the fast tree's nodes hold four boxes; with the leaf-box cull every child's slab test is used
    out->samples = s->stat_samples; out->pixels = s->stat_pixels; out->launches = s->launches;
    return 0;
}
extern "C" long long terra_amd_debug_faults ( HTerraScene h ) {
    // out-of-plan stack / leaf-list writes refused by a TERRA_CHECK_BOUNDS build since the last reset (always 0 in the shipped build)
    Scene* s = S ( h );
    if ( !s->device_ok ) return -1;
    unsigned long long v = 0;
    if ( hipSetDevice ( s->device ) != hipSuccess || hipMemcpy ( &v, s->d_counters + kCtrFaults, sizeof v, hipMemcpyDeviceToHost ) != hipSuccess ) return -1;
    return ( long long ) v;
}
extern "C" int terra_amd_debug_counters ( HTerraScene h, unsigned long long* out16 ) {
    // phase occupancy counters of a TERRA_PHASE_STATS build (tools/phase_stats.py); all zero in the shipped build
    Scene* s = S ( h );
    if ( !s->device_ok || !out16 ) return fail ( kTerraAmdErrNotCommitted, "scene has no device replica" );
    HIP_TRY ( hipSetDevice ( s->device ), kTerraAmdErrNoDevice );
    HIP_TRY ( hipMemcpy ( out16, s->d_counters + kCtrDbg0, 16 * sizeof ( unsigned long long ), hipMemcpyDeviceToHost ), kTerraAmdErrNoDevice );
    return 0;
}
extern "C" int terra_amd_reset_stats ( HTerraScene h ) {
    Scene* s = S ( h );
    if ( !s->device_ok ) return fail ( kTerraAmdErrNotCommitted, "scene has no device replica" );
    HIP_TRY ( hipSetDevice ( s->device ), kTerraAmdErrNoDevice );
    HIP_TRY ( hipMemset ( s->d_counters, 0, kCtrCount * sizeof ( unsigned long long ) ), kTerraAmdErrNoDevice );
    s->launches = 0; s->stat_pixels = 0; s->stat_samples = 0;
    return 0;
}

// ------------------------------------------------------------------------------
// render
// ------------------------------------------------------------------------------
static uint32_t effective_spp ( const TerraSceneOptions& o ) {
    size_t spp = o.samples_per_pixel;
    if ( o.sampling_method == kTerraSamplingMethodStratified ) {      // reference src/Terra.c:519-527
        size_t cur = o.strata * o.strata;
        while ( spp > cur && cur > 1 ) cur *= cur;
        if ( cur >= spp ) spp = cur;
    }
    return ( uint32_t ) spp;
}

// replica: which device's copy of the scene the launch reads (nullptr: the primary device's)
static int fill_params ( Scene* s, const TerraCamera* cam, size_t fb_w, size_t fb_h, size_t x, size_t y, size_t w, size_t h,
                         size_t tile, int rank, int world, DevRenderParams& p, const Scene::Replica* replica = nullptr ) {
    if ( !s->committed ) return fail ( kTerraAmdErrNotCommitted, "terra_scene_commit has not run since the scene changed" );
    if ( !s->device_ok ) return fail ( kTerraAmdErrNoDevice, "scene has no device replica: %s", s->commit_error.c_str() );
    if ( !cam || w == 0 || h == 0 || x + w > fb_w || y + h > fb_h ) return fail ( kTerraAmdErrBadArgument, "bad tile rectangle %zu,%zu %zux%zu in %zux%zu", x, y, w, h, fb_w, fb_h );
    if ( tile == 0 || tile % 16 != 0 ) return fail ( kTerraAmdErrBadArgument, "tile_size %zu must be a positive multiple of 16", tile );
    if ( world < 1 || rank < 0 || rank >= world ) return fail ( kTerraAmdErrBadArgument, "bad shard %d/%d", rank, world );
    memset ( &p, 0, sizeof p );
    p.scene = replica ? replica->dev : s->dev;
    // camera frame, reference src/Terra.c:1770-1781
    TerraFloat3 z = terra_normf3 ( &cam->direction );
    TerraFloat3 xa = terra_crossf3 ( &cam->up, &z ); xa = terra_normf3 ( &xa );
    TerraFloat3 ya = terra_crossf3 ( &z, &xa );
    p.cam_rot[0] = xa.x; p.cam_rot[1] = ya.x; p.cam_rot[2] = z.x;
    p.cam_rot[3] = xa.y; p.cam_rot[4] = ya.y; p.cam_rot[5] = z.y;
    p.cam_rot[6] = xa.z; p.cam_rot[7] = ya.z; p.cam_rot[8] = z.z;
    p.cam_pos[0] = cam->position.x; p.cam_pos[1] = cam->position.y; p.cam_pos[2] = cam->position.z;
    p.tan_half_fov = ( float ) tan ( ( double ) ( ( cam->fov * 0.0174533f ) / 2 ) );     // double tan of a float argument, src/Terra.c:1794
    p.aspect = ( float ) fb_w / ( float ) fb_h;
    p.jitter = s->opts.subpixel_jitter; p.exposure = s->opts.manual_exposure; p.gamma = s->opts.gamma;
    p.fb_w = ( uint32_t ) fb_w; p.fb_h = ( uint32_t ) fb_h;
    p.x = ( uint32_t ) x; p.y = ( uint32_t ) y; p.w = ( uint32_t ) w; p.h = ( uint32_t ) h;
    p.tile_size = ( uint32_t ) tile; p.rank = ( uint32_t ) rank; p.world = ( uint32_t ) world;
    p.st_x = 0; p.st_y = 0; p.st_pitch = ( uint32_t ) fb_w;          // framebuffer arrays indexed like the frame (render_host stages a rectangle instead)
    p.spp = effective_spp ( s->opts );
    p.split = 1; p.split_log2 = 0; p.chunk_spp = p.spp; p.partials = nullptr; p.job_blocks = 0; p.job_queue = nullptr;
    p.bounces = ( uint32_t ) s->opts.bounces;
    p.integrator = ( int32_t ) s->opts.integrator; p.tonemap = ( int32_t ) s->opts.tonemapping_operator;
    if ( p.integrator < 0 || p.integrator > 6 ) return fail ( kTerraAmdErrBadArgument, "unknown integrator %d", p.integrator );
    if ( ( p.integrator == kTerraIntegratorDirect || p.integrator == kTerraIntegratorDirectMis || p.integrator == kTerraIntegratorDebugMisWeights ) && s->lights.empty() )
        return fail ( kTerraAmdErrBadArgument, "integrator %d needs at least one emissive object (the reference asserts, src/Terra.c:1617)", p.integrator );
    p.frame_seed = s->frame_seed;
    p.counters = replica ? replica->d_counters : s->d_counters;
    terra_plan_lds ( p );
    // automatic mode: the containment argument also needs the ray origins (the camera) inside the verified coordinate range
    const bool cam_ok = ( s->reach || s->reach_cull ) ? ( fabsf ( p.cam_pos[0] ) <= s->reach_limit && fabsf ( p.cam_pos[1] ) <= s->reach_limit && fabsf ( p.cam_pos[2] ) <= s->reach_limit ) : coords_within_margin ( p.cam_pos, 3 );
    if ( s->use_fast && s->dev.fast_nodes_h && ( s->tree_mode == 1 || cam_ok ) ) terra_plan_fast_tree ( p );
    else if ( ( s->use_fast || s->cull_ok ) && !cam_ok && !s->warned_camera.exchange ( true ) )          // (once per scene)
        fprintf ( stderr, "[terra_amd] warning: camera at (%g, %g, %g) lies outside the range (+-%g) for which this scene's traversal shortcut is proven: this call runs the reference "
                  "tree's replica traversal (same image, typically 10-20 x slower on large scenes); terra_amd_traversal_info() reports camera_limit and last_call\n",
                  ( double ) p.cam_pos[0], ( double ) p.cam_pos[1], ( double ) p.cam_pos[2], ( double ) ( ( s->reach || s->reach_cull ) ? s->reach_limit : TERRA_CULL_MAX_COORD ) );
    if ( s->test_fast_stack_lds > 0 && p.lds_mode == 2 ) {      // TEST HOOK: a short LDS column, so that ordinary scenes exercise the HBM part of the stack
        const uint32_t need = p.stack_depth + p.spill_cap;
        p.stack_depth = need < ( uint32_t ) s->test_fast_stack_lds ? need : ( uint32_t ) s->test_fast_stack_lds; p.spill_cap = need - p.stack_depth;
    }
    if ( s->test_pad_stack > 0 && p.lds_mode != 1 ) {          // TEST HOOK: a deeper stack than the tree needs (the LDS-resident plan is sized to the byte and stays as it is)
        p.stack_depth += ( uint32_t ) s->test_pad_stack;
        while ( p.leaf_cap > 4 && terra_lds_bytes ( p ) > ( size_t ) 64 * 1024 ) --p.leaf_cap;      // (what terra_plan_lds does for a deep tree)
    }
    p.leaf_cull = ( s->cull_ok && cam_ok && p.lds_mode != 2 ) ? 1u : 0u;
    p.fused_slab = ( p.leaf_cull && !s->reach_cull ) ? 1u : 0u;      // (out of range only the rebuilt LEAF boxes carry a margin: the inner boxes are tested exactly as the reference tests them)
    // the azimuth table pays where VALU issue binds (LDS-resident scenes: Cornell Simple 65.8 -> 64.2 ms, Direct 145.2 -> 142.5); the kernels that wait on memory anyway
    // lose by one more dependent load per shaded hit (sphere scene 395 -> 419 ms, hall 282 -> 284; profiles/r03_measurements/ab_sincos_table.log)
#ifndef TERRA_SINCOS_TABLE_FAST_TREE       // (A/B) the azimuth table for fast-tree launches too
#define TERRA_SINCOS_TABLE_FAST_TREE 0
#endif
    if ( p.lds_mode != 1 && ! ( TERRA_SINCOS_TABLE_FAST_TREE && p.lds_mode == 2 ) ) p.scene.sincos24 = nullptr;
    // what this call runs (TerraAmdTraversalInfo::last_call): the commit-time decision can be overridden per call by the camera position
    s->last_call.store ( p.lds_mode == 2 ? ( s->dev.reach ? kTerraAmdCallFastTreeReach : kTerraAmdCallFastTree ) : ( p.leaf_cull ? kTerraAmdCallLeafCull : kTerraAmdCallReplica ), std::memory_order_relaxed );
    p.bsdf_kinds = s->bsdf_kinds;
    p.sampler_mode = 0; p.sampler_strata = ( uint32_t ) s->opts.strata;
    if ( s->sampler_integration && s->opts.sampling_method == kTerraSamplingMethodHalton ) p.sampler_mode = 1;
    if ( s->sampler_integration && s->opts.sampling_method == kTerraSamplingMethodStratified && s->opts.strata > 0 ) p.sampler_mode = 2;
    if ( p.sampler_mode ) p.bsdf_kinds |= TERRA_KIND_SAMPLER;
    // environment sampling lives in the same kernel variant; it only acts in the two integrators that sample lights
    if ( s->dev.env_nx && ( p.integrator == kTerraIntegratorDirect || p.integrator == kTerraIntegratorDirectMis ) ) p.bsdf_kinds |= TERRA_KIND_SAMPLER;
    else { p.scene.env_nx = 0; p.scene.env_ny = 0; }
    p.count_level = s->work_counters ? 2 : 0;
    return 0;
}

// pixels a launch covers: own tiles (t % world == rank) clipped to the rectangle
static uint64_t shard_pixels ( const DevRenderParams& p ) {
    uint64_t n = 0;
    const uint32_t tx = ( p.w + p.tile_size - 1 ) / p.tile_size, ty = ( p.h + p.tile_size - 1 ) / p.tile_size;
    for ( uint32_t t = p.rank; t < tx * ty; t += p.world ) {
        uint32_t x0 = ( t % tx ) * p.tile_size, y0 = ( t / tx ) * p.tile_size;
        n += ( uint64_t ) std::min ( p.tile_size, p.w - x0 ) * std::min ( p.tile_size, p.h - y0 );
    }
    return n;
}
static void account_launch ( Scene* s, const DevRenderParams& p ) {
    uint64_t px = shard_pixels ( p );
    s->launches.fetch_add ( 1, std::memory_order_relaxed ); s->stat_pixels.fetch_add ( px, std::memory_order_relaxed ); s->stat_samples.fetch_add ( px * p.spp, std::memory_order_relaxed );
}

// One render of p on `stream`: the render kernel sums every (pixel, chunk) job into a stream-ordered scratch buffer (persistent grid,
// jobs handed out through a queue word at the head of that buffer; render_kernels.hip "jobs"), then the resolve kernel folds the chunk
// sums into the pixels in chunk order and tonemaps (DevRenderParams::split; split == 1: one chunk per pixel).
struct ThreadSlot;
static void* slot_scratch ( ThreadSlot* slot, size_t bytes );
// The automatic sample split (terra_amd_set_sample_split(scene, 0); terra_amd_auto_sample_split): a function of the launch's size -- its 16x16 pixel blocks --, its spp and
// whether its jobs are handed out in the job order, so the same calls always give the same framebuffer. Chunks of at least 16 samples, at most 32 lanes per pixel (the
// reference client's 128-pixel tiles at 512 spp, called from 8 threads, 69.1 -> 66.7 ms per frame with 32 instead of 16).
//   * launches WITHOUT the job order end with their last jobs ramping down alone, so they want many short jobs: about 200 per lane the GPU holds at once (256 CUs x 5 blocks x
//     256 lanes) -- hall 1080p 256 spp: split 1 / 4 / 8 -> 276.6 / 255.1 / 251.0 ms (profiles/r03_measurements/ab_job_queue.log);
//   * launches WITH it (LDS-resident scenes, render_kernels.hip "job order") end on short jobs whatever the split, and every job switch costs them (its code runs for one
//     or two lanes of a wave; the jobs' stream table and sums are traffic): about 50 jobs per lane is where the slowest of N shards of the Cornell frame is fastest --
//     split 8 / 16 / 32 / 32 for 1 / 2 / 4 / 8 shards: 50.2 / 25.6 / 13.5 / 7.0 ms, against 53.5 / 26.9 / 13.5 / 7.0 at 32 everywhere
//     (tools/split_matrix.py, profiles/r04_measurements/split_matrix.log).
static uint32_t auto_sample_split ( uint32_t blocks, uint32_t spp, bool ordered ) {
    const uint64_t enough = ordered ? 61440u : 245760u;          // blocks x split: x 256 jobs each, over 327,680 resident lanes = 48 / 192 jobs per lane
    uint32_t split = 1;
    while ( split < 32 && ( uint64_t ) blocks * split < enough && spp / ( split * 2 ) >= 16 ) split *= 2;
    return split;
}
static int launch_render ( Scene* s, DevRenderParams& p, hipStream_t stream, ThreadSlot* slot = nullptr, int device = -1 ) {      // device: the (current) device of the launch, -1 = the scene's primary
    uint32_t split = s->sample_split;
    const uint32_t blocks = terra_render_blocks ( p );
    if ( blocks == 0 ) return 0;
    p.job_blocks = blocks;                                            // (what terra_block_order_bytes looks at is the launch's size and layout, not the split)
    if ( split == 0 ) split = auto_sample_split ( blocks, p.spp, s->job_order && terra_block_order_bytes ( p, s->job_order == 2 ) != 0 );
    while ( split > 1 && p.spp % split ) split >>= 1;              // chunks must be equal: fall back to the largest power of two dividing spp
    if ( split < 1 ) split = 1;
    if ( device < 0 ) device = s->device;
    static thread_local uint64_t pool_kept = 0;          // (bit d: done for device d)
    if ( device < 64 && ! ( pool_kept >> device & 1ull ) ) {        // keep freed scratch cached in the device's default pool instead of returning it to the OS at every sync
        hipMemPool_t pool;
        if ( hipDeviceGetDefaultMemPool ( &pool, device ) == hipSuccess ) { uint64_t keep = ~0ull; ( void ) hipMemPoolSetAttribute ( pool, hipMemPoolAttrReleaseThreshold, &keep ); }
        pool_kept |= 1ull << device;
    }
    const size_t header = 256;                                     // the job queue word (+ padding that keeps the partials 256-byte aligned)
    const size_t partial_bytes = ( size_t ) split * blocks * 256 * sizeof ( float4 );
    p.job_blocks = blocks * split;
    const size_t stream_bytes = terra_job_streams_bytes ( p );
    if ( stream_bytes && ( p.fb_w > 65535u || p.fb_h > 65535u ) ) return fail ( kTerraAmdErrBadArgument, "framebuffer of %u x %u: at most 65,535 pixels per side (the job table packs a pixel into 32 bits)", p.fb_w, p.fb_h );
    const size_t spill_bytes = terra_fast_spill_bytes ( p );                 // fast-tree launches: the part of the lanes' traversal stacks that does not live in LDS
    const size_t order_bytes = s->job_order ? terra_block_order_bytes ( p, s->job_order == 2 ) : 0;           // LDS-resident launches: the order the pixel blocks are handed out in (render_kernels.hip "job order")
    const size_t scratch_bytes = header + partial_bytes + stream_bytes + spill_bytes + order_bytes;      // [queue word][job sums][job streams (LDS-resident scenes)][stack spill (fast tree)][job order]
    // (a thread's slot keeps scratch for tile-sized calls only: a full-frame call's gigabytes come from, and go back to, the device's pool)
    // a launch's scratch is bounded: 48 bytes per (pixel, lane-per-pixel) job on LDS-resident scenes -- a 4K frame at 64 lanes per pixel asks for 25 GB per concurrent stream.
    // Beyond TERRA_SCRATCH_MAX_GB the call is refused with the size in the message (fewer lanes per pixel, or the frame in several calls, give the same framebuffer)
    if ( scratch_bytes > ( size_t ) TERRA_SCRATCH_MAX_GB << 30 )
        return fail ( kTerraAmdErrBadArgument, "this launch needs %.1f GB of scratch (%u blocks x sample split %u: 16 B of job sums%s per job), more than the %d GB a launch may take: lower terra_amd_set_sample_split or render the frame in several calls",
                      ( double ) scratch_bytes / ( 1 << 30 ), blocks, split, stream_bytes ? " + 32 B of job streams" : "", TERRA_SCRATCH_MAX_GB );
    void* scratch = ( slot && scratch_bytes <= ( size_t ( 256 ) << 20 ) ) ? slot_scratch ( slot, scratch_bytes ) : nullptr;
    const bool pooled = scratch == nullptr;
    if ( pooled ) { const hipError_t ea = hipMallocAsync ( &scratch, scratch_bytes, stream ); if ( ea != hipSuccess ) { ( void ) hipGetLastError(); return fail ( kTerraAmdErrNoDevice, "launch scratch of %.2f GB: %s", ( double ) scratch_bytes / ( 1 << 30 ), hipGetErrorString ( ea ) ); } }
    // the queue word must be zero when the render kernel starts: a slot's scratch is zeroed when it is allocated and again by every resolve kernel (one kernel less per
    // call on the host path, whose small kernels wait behind the other callers' render grids); memory from the pool is fresh each time
    hipError_t e = pooled ? hipMemsetAsync ( scratch, 0, header, stream ) : hipSuccess;
    p.split = split; p.split_log2 = 0; while ( ( 1u << p.split_log2 ) < split ) ++p.split_log2;
    p.chunk_spp = p.spp / split; p.partials = ( float4* ) ( ( char* ) scratch + header );
    p.job_blocks = blocks * split; p.job_queue = terra_render_wants_queue ( p ) ? ( uint32_t* ) scratch : nullptr;
    p.job_streams = stream_bytes ? ( uint4* ) ( ( char* ) scratch + header + partial_bytes ) : nullptr;
    p.stack_spill = spill_bytes ? ( uint32_t* ) ( ( char* ) scratch + header + partial_bytes + stream_bytes ) : nullptr;
    uint32_t* const order_cls = order_bytes ? ( uint32_t* ) ( ( char* ) scratch + header + partial_bytes + stream_bytes + spill_bytes ) : nullptr;
    p.block_order = order_cls ? order_cls + blocks : nullptr;
    {   // the job decode divides block numbers by launch constants: as multiplications by ceil(2^32 / d), exact while (largest dividend) * divisor < 2^32
        const uint64_t bpt = p.tile_size / 16, bpt2 = bpt * bpt, tiles_x = ( p.w + p.tile_size - 1 ) / p.tile_size, tiles_y = ( p.h + p.tile_size - 1 ) / p.tile_size;
        auto magic = [] ( uint64_t d ) { return d <= 1 ? 0u : ( uint32_t ) ( ( ( 1ull << 32 ) + d - 1 ) / d ); };
        if ( ! ( ( uint64_t ) blocks * split * 256 < ( 1ull << 32 ) && ( uint64_t ) blocks * bpt2 < ( 1ull << 32 ) && ( tiles_x * tiles_y + p.world ) * tiles_x < ( 1ull << 32 ) && bpt2 * bpt < ( 1ull << 32 ) ) ) {
            if ( pooled ) ( void ) hipFreeAsync ( scratch, stream );
            return fail ( kTerraAmdErrBadArgument, "render rectangle too large for one launch (%u blocks x split %u): render it in several calls", blocks, split );
        }
        p.job_div_bpt2 = magic ( bpt2 ); p.job_div_tiles_x = magic ( tiles_x ); p.job_div_bpt = magic ( bpt ); p.job_tiles_x = ( uint32_t ) tiles_x;
    }
    if ( terra_lds_bytes ( p ) > terra_lds_block_limit() ) {          // (a reference tree hundreds of levels deep: nothing this library can launch)
        if ( pooled ) ( void ) hipFreeAsync ( scratch, stream );
        return fail ( kTerraAmdErrUnsupported, "the traversal stack of this scene's tree (%u entries) needs %zu KB of LDS per block, more than the %zu KB a block can have", p.stack_depth, terra_lds_bytes ( p ) / 1024, terra_lds_block_limit() / 1024 );
    }
    if ( e == hipSuccess && order_cls ) e = terra_launch_block_order ( p, order_cls, stream );
    if ( e == hipSuccess ) e = terra_launch_job_streams ( p, stream );
    if ( e == hipSuccess ) e = terra_launch_render ( p, stream );
    if ( e == hipSuccess ) e = terra_launch_resolve ( p, stream );
    if ( pooled ) ( void ) hipFreeAsync ( scratch, stream );
    else if ( e != hipSuccess ) ( void ) hipMemsetAsync ( scratch, 0, header, stream );       // (a launch that failed half way must not leave a used queue word behind)
    if ( e != hipSuccess ) return fail ( kTerraAmdErrLaunch, "render launch: %s", hipGetErrorString ( e ) );
    return 0;
}

extern "C" int terra_amd_render_device_sharded ( const TerraCamera* cam, HTerraScene h, void* d_pixels, void* d_results, size_t fb_w, size_t fb_h,
                                                 size_t x, size_t y, size_t w, size_t hgt, size_t tile, int rank, int world, void* d_rand_calls, void* stream ) {
    Scene* s = S ( h );
    DevRenderParams p;
    int rc = fill_params ( s, cam, fb_w, fb_h, x, y, w, hgt, tile, rank, world, p );
    if ( rc ) return rc;
    if ( !d_pixels || !d_results ) return fail ( kTerraAmdErrBadArgument, "null framebuffer pointer" );
    p.pixels = ( float* ) d_pixels; p.results = d_results; p.rand_calls = ( uint32_t* ) d_rand_calls;
    if ( d_rand_calls ) p.count_level = 2;
    HIP_TRY ( hipSetDevice ( s->device ), kTerraAmdErrNoDevice );
    if ( int lrc = launch_render ( s, p, ( hipStream_t ) stream ) ) return lrc;
    account_launch ( s, p );
    return 0;
}
extern "C" int terra_amd_render_device ( const TerraCamera* cam, HTerraScene h, void* d_pixels, void* d_results, size_t fb_w, size_t fb_h,
                                         size_t x, size_t y, size_t w, size_t hgt, void* d_rand_calls, void* stream ) {
    return terra_amd_render_device_sharded ( cam, h, d_pixels, d_results, fb_w, fb_h, x, y, w, hgt, 64, 0, 1, d_rand_calls, stream );
}
extern "C" int terra_amd_synchronize ( void* stream ) {
    HIP_TRY ( hipStreamSynchronize ( ( hipStream_t ) stream ), kTerraAmdErrLaunch );
    return 0;
}
extern "C" int terra_amd_time_render_device ( const TerraCamera* cam, HTerraScene h, void* d_pixels, void* d_results, size_t fb_w, size_t fb_h,
                                              size_t x, size_t y, size_t w, size_t hgt, int launches, void* stream, float* ms_avg ) {
    if ( launches < 1 || !ms_avg ) return fail ( kTerraAmdErrBadArgument, "launches < 1" );
    hipEvent_t e0, e1;
    HIP_TRY ( hipSetDevice ( S ( h )->device ), kTerraAmdErrNoDevice );
    HIP_TRY ( hipEventCreate ( &e0 ), kTerraAmdErrLaunch ); HIP_TRY ( hipEventCreate ( &e1 ), kTerraAmdErrLaunch );
    HIP_TRY ( hipEventRecord ( e0, ( hipStream_t ) stream ), kTerraAmdErrLaunch );
    for ( int i = 0; i < launches; ++i ) {
        int rc = terra_amd_render_device ( cam, h, d_pixels, d_results, fb_w, fb_h, x, y, w, hgt, nullptr, stream );
        if ( rc ) { ( void ) hipEventDestroy ( e0 ); ( void ) hipEventDestroy ( e1 ); return rc; }
    }
    HIP_TRY ( hipEventRecord ( e1, ( hipStream_t ) stream ), kTerraAmdErrLaunch );
    HIP_TRY ( hipEventSynchronize ( e1 ), kTerraAmdErrLaunch );
    float ms = 0.f;
    HIP_TRY ( hipEventElapsedTime ( &ms, e0, e1 ), kTerraAmdErrLaunch );
    ( void ) hipEventDestroy ( e0 ); ( void ) hipEventDestroy ( e1 );
    *ms_avg = ms / ( float ) launches;
    return 0;
}

// ---- tile pack/unpack ------------------------------------------------------------
static uint32_t tiles_of_rank ( size_t w, size_t h, size_t tile, int rank, int world ) {
    size_t tiles = ( ( w + tile - 1 ) / tile ) * ( ( h + tile - 1 ) / tile );
    return tiles > ( size_t ) rank ? ( uint32_t ) ( ( tiles - ( size_t ) rank + ( size_t ) world - 1 ) / ( size_t ) world ) : 0u;
}
extern "C" int terra_amd_shard_tile_count ( size_t w, size_t h, size_t tile, int rank, int world ) {
    if ( tile == 0 || world < 1 || rank < 0 || rank >= world ) return fail ( kTerraAmdErrBadArgument, "bad shard arguments" );
    return ( int ) tiles_of_rank ( w, h, tile, rank, world );
}
extern "C" size_t terra_amd_shard_packed_bytes ( size_t w, size_t h, size_t tile, int world ) {
    if ( tile == 0 || world < 1 ) return 0;
    return ( size_t ) tiles_of_rank ( w, h, tile, 0, world ) * tile * tile * 28;     // rank 0 holds the most tiles
}
static int tiles_call ( bool pack, void* d_pixels, void* d_results, size_t fb_w, size_t fb_h, size_t x, size_t y, size_t w, size_t h, size_t tile, int rank, int world, void* d_packed, void* stream ) {
    if ( !d_pixels || !d_results || !d_packed || tile == 0 || tile % 16 != 0 || world < 1 || rank < 0 || rank >= world || x + w > fb_w || y + h > fb_h )
        return fail ( kTerraAmdErrBadArgument, "bad pack/unpack arguments" );
    HIP_TRY ( terra_launch_tiles ( pack, ( float* ) d_pixels, d_results, ( uint32_t ) fb_w, ( uint32_t ) x, ( uint32_t ) y, ( uint32_t ) w, ( uint32_t ) h,
                                   ( uint32_t ) tile, ( uint32_t ) rank, ( uint32_t ) world, ( float* ) d_packed, ( hipStream_t ) stream ), kTerraAmdErrLaunch );
    return ( int ) tiles_of_rank ( w, h, tile, rank, world );
}
extern "C" int terra_amd_pack_tiles ( const void* d_pixels, const void* d_results, size_t fb_w, size_t fb_h, size_t x, size_t y, size_t w, size_t h, size_t tile, int rank, int world, void* d_packed, void* stream ) {
    return tiles_call ( true, ( void* ) d_pixels, ( void* ) d_results, fb_w, fb_h, x, y, w, h, tile, rank, world, d_packed, stream );
}
extern "C" int terra_amd_unpack_tiles ( void* d_pixels, void* d_results, size_t fb_w, size_t fb_h, size_t x, size_t y, size_t w, size_t h, size_t tile, int rank, int world, const void* d_packed, void* stream ) {
    return tiles_call ( false, d_pixels, d_results, fb_w, fb_h, x, y, w, h, tile, rank, world, ( void* ) d_packed, stream );
}

// ---- terra_render on a HOST framebuffer (the drop-in entry point) -------------------
// Per calling thread: one stream and a device staging buffer the size of the largest TILE the thread has rendered (not of the
// frame: a client that renders 128-pixel tiles from 8 workers stages 8 x 0.46 MB, not 8 frames). The tile's running sums go up
// (they key the random streams and are accumulated on), the kernel renders into the staging rectangle, the tile comes back.
// The slot is released when its thread exits.
struct ThreadSlot {
    int device = -1; hipStream_t stream = nullptr; void* d_pixels = nullptr; void* d_results = nullptr; size_t cap_px = 0;
    void* d_scratch = nullptr; size_t scratch_bytes = 0;      // the launch's job sums + queue word (launch_render): kept per thread so that a tile-sized call does not go through the pool
    void release() {
        if ( device < 0 ) return;
        if ( hipSetDevice ( device ) == hipSuccess ) {
            if ( stream ) { ( void ) hipStreamSynchronize ( stream ); ( void ) hipStreamDestroy ( stream ); }
            if ( d_pixels ) ( void ) hipFree ( d_pixels );
            if ( d_results ) ( void ) hipFree ( d_results );
            if ( d_scratch ) ( void ) hipFree ( d_scratch );
        }
        device = -1; stream = nullptr; d_pixels = d_results = d_scratch = nullptr; cap_px = 0; scratch_bytes = 0;
    }
    ~ThreadSlot() { release(); }
};
// A thread keeps its slot (stream, staging buffers, scratch) for its lifetime; a thread that ends hands the slot to the next new thread instead of freeing it:
// clients that start fresh worker threads for every frame would otherwise pay a stream creation, three allocations and -- worse -- three hipFree, each of which
// waits for the whole device, per thread and frame. The pool is emptied when the library is unloaded.
// Hardware queues: ROCm maps a process's streams onto GPU_MAX_HW_QUEUES (default 4) hardware queues, and kernels of streams that share one run one after the other.
// The reference's client calls terra_render() from 8 worker threads (satellite/src/Renderer.cpp:70-98), each with its own stream here, so such a client wants 8 -- which
// only has an effect when the variable is set before the process's first HIP call (tile loop of the Cornell frame, 8 threads: 69.7 -> 65.5 ms;
// profiles/r03_measurements/host_tile_loop.log). A library does not change its host's environment behind its back: terra_amd_init() does it when the CLIENT asks, first
// thing in main(); it never overrides a value the user has set.
extern "C" int terra_amd_init ( void ) {
    if ( setenv ( "GPU_MAX_HW_QUEUES", "8", 0 ) != 0 ) return fail ( kTerraAmdErrBadArgument, "setenv failed" );
    return 0;
}
struct SlotPool {
    std::mutex lock; std::vector<ThreadSlot*> idle;
    ThreadSlot* take() { std::lock_guard<std::mutex> g ( lock ); if ( idle.empty() ) return new ThreadSlot(); ThreadSlot* t = idle.back(); idle.pop_back(); return t; }
    void give ( ThreadSlot* t ) { std::lock_guard<std::mutex> g ( lock ); idle.push_back ( t ); }
    ~SlotPool() { for ( ThreadSlot* t : idle ) delete t; }
};
static SlotPool g_slots;
struct SlotHolder { ThreadSlot* slot = nullptr; ThreadSlot& get() { if ( !slot ) slot = g_slots.take(); return *slot; } ~SlotHolder() { if ( slot ) g_slots.give ( slot ); } };
static thread_local SlotHolder t_slot_holder;
#define t_slot ( t_slot_holder.get() )
// (the slot's stream orders its launches: a call's sums are folded before the next call of the thread overwrites them)
static void* slot_scratch ( ThreadSlot* slot, size_t bytes ) {
    if ( slot->scratch_bytes < bytes ) {
        if ( slot->d_scratch ) { ( void ) hipStreamSynchronize ( slot->stream ); ( void ) hipFree ( slot->d_scratch ); slot->d_scratch = nullptr; slot->scratch_bytes = 0; }
        if ( hipMalloc ( &slot->d_scratch, bytes ) != hipSuccess ) { ( void ) hipGetLastError(); slot->d_scratch = nullptr; return nullptr; }      // (the pool then)
        if ( hipMemset ( slot->d_scratch, 0, 256 ) != hipSuccess ) { ( void ) hipGetLastError(); ( void ) hipFree ( slot->d_scratch ); slot->d_scratch = nullptr; return nullptr; }
        slot->scratch_bytes = bytes;
    }
    return slot->d_scratch;
}

static int slot_prepare ( int device, size_t npx ) {
    ThreadSlot& t = t_slot;
    int current = -1;
    if ( hipGetDevice ( &current ) != hipSuccess || current != device ) HIP_TRY ( hipSetDevice ( device ), kTerraAmdErrNoDevice );
    if ( t.device != device ) {
        const bool had_other = t.device >= 0;
        t.release(); t.device = device;
        if ( had_other ) HIP_TRY ( hipSetDevice ( device ), kTerraAmdErrNoDevice );      // (release() made the slot's old device current)
        HIP_TRY ( hipStreamCreateWithFlags ( &t.stream, hipStreamNonBlocking ), kTerraAmdErrNoDevice );
    }
    if ( t.cap_px < npx ) {
        if ( t.d_pixels ) ( void ) hipFree ( t.d_pixels );
        if ( t.d_results ) ( void ) hipFree ( t.d_results );
        t.d_pixels = t.d_results = nullptr; t.cap_px = 0;
        HIP_TRY ( hipMalloc ( &t.d_pixels, npx * 12 ), kTerraAmdErrNoDevice );
        HIP_TRY ( hipMalloc ( &t.d_results, npx * 16 ), kTerraAmdErrNoDevice );
        t.cap_px = npx;
    }
    return 0;
}
extern "C" size_t terra_amd_thread_staging_bytes ( void ) { return t_slot.cap_px * 28; }

static int render_host_multi ( const TerraCamera* cam, Scene* s, const TerraFramebuffer* fb, size_t x, size_t y, size_t w, size_t h, size_t tile );
static std::atomic<int> g_thread_ordinals { 0 };
static int render_host ( const TerraCamera* cam, Scene* s, const TerraFramebuffer* fb, size_t x, size_t y, size_t w, size_t h ) {
    if ( !fb || !fb->pixels || !fb->results ) return fail ( kTerraAmdErrBadArgument, "null framebuffer" );
    // A scene committed for several devices (terra_amd_set_devices). A call that covers enough of a frame to give every device a fair share of 64-pixel tiles is
    // sharded over them and gathered (render_host_multi); a small one -- the reference client's 128-pixel tiles, called from its worker threads
    // (satellite/src/Renderer.cpp:70-98,316-350) -- goes whole to ONE device, the calling thread's: threads are dealt to the devices round-robin when they first call, so
    // eight workers drive eight GPUs and a thread's stream and staging buffers stay on one device.
    const Scene::Replica* replica = nullptr;
    if ( s->devices.size() > 1 && s->device_ok ) {
        if ( w * h >= ( size_t ) 256 * 256 * s->devices.size() ) return render_host_multi ( cam, s, fb, x, y, w, h, 64 );
        static thread_local int ordinal = -1;
        if ( ordinal < 0 ) ordinal = g_thread_ordinals.fetch_add ( 1, std::memory_order_relaxed );
        const size_t k = ( size_t ) ordinal % s->devices.size();
        if ( k > 0 && k - 1 < s->extra.size() ) replica = &s->extra[k - 1];
    }
    const int device = replica ? replica->device : s->device;
    DevRenderParams p;
    int rc = fill_params ( s, cam, fb->width, fb->height, x, y, w, h, 64, 0, 1, p, replica );
    if ( rc ) return rc;
    rc = slot_prepare ( device, w * h );
    if ( rc ) return rc;
    ThreadSlot& t = t_slot;
    // the staging buffer holds the rectangle only (rows of w pixels); the kernel addresses it through st_x / st_y / st_pitch while
    // the camera and the random streams keep using the frame's geometry
    p.st_x = ( uint32_t ) x; p.st_y = ( uint32_t ) y; p.st_pitch = ( uint32_t ) w;
    const size_t rpitch = fb->width * 16, ppitch = fb->width * 12;
    const char* hres = ( const char* ) fb->results + ( y * fb->width + x ) * 16;
    char* hpix = ( char* ) fb->pixels + ( y * fb->width + x ) * 12;
    HIP_TRY ( hipMemcpy2DAsync ( t.d_results, w * 16, hres, rpitch, w * 16, h, hipMemcpyHostToDevice, t.stream ), kTerraAmdErrLaunch );
    p.pixels = ( float* ) t.d_pixels; p.results = t.d_results; p.rand_calls = nullptr;
    if ( int lrc = launch_render ( s, p, t.stream, &t, device ) ) return lrc;
    HIP_TRY ( hipMemcpy2DAsync ( ( void* ) hres, rpitch, t.d_results, w * 16, w * 16, h, hipMemcpyDeviceToHost, t.stream ), kTerraAmdErrLaunch );
    HIP_TRY ( hipMemcpy2DAsync ( hpix, ppitch, t.d_pixels, w * 12, w * 12, h, hipMemcpyDeviceToHost, t.stream ), kTerraAmdErrLaunch );
    HIP_TRY ( hipStreamSynchronize ( t.stream ), kTerraAmdErrLaunch );
    account_launch ( s, p );
    return 0;
}

// terra_render() over the scene's device set from one process: device k renders the tiles t with t % N == k of the rectangle (terra_amd_shard_owner; the same kernels
// with DevRenderParams::rank / world set) into its own staging frame, packs them, and ONE gather -- RCCL over xGMI, issued from here (multi_gpu.cpp) -- brings every
// rank's packed tiles to the primary device, which unpacks them into its staging frame and copies the rectangle to the host once. The reference's client shards tiles over
// worker threads of one process the same way (satellite/src/Renderer.cpp:316-350). The running sums of the rectangle go up to every device first (they key the random
// streams and are accumulated on, src/Terra.c:570-574): 16 B per pixel and device over each device's own PCIe link.
// With ONE device in the set the same calls run: a communicator of one rank, the gather a copy inside the device.
static int render_host_multi ( const TerraCamera* cam, Scene* s, const TerraFramebuffer* fb, size_t x, size_t y, size_t w, size_t h, size_t tile ) {
    if ( !fb || !fb->pixels || !fb->results ) return fail ( kTerraAmdErrBadArgument, "null framebuffer" );
    if ( !s->committed ) return fail ( kTerraAmdErrNotCommitted, "terra_scene_commit has not run since the scene changed" );
    if ( !s->device_ok ) return fail ( kTerraAmdErrNoDevice, "scene has no device replica: %s", s->commit_error.c_str() );
    if ( tile == 0 || tile % 16 != 0 ) return fail ( kTerraAmdErrBadArgument, "tile_size %zu must be a positive multiple of 16", tile );
    std::lock_guard<std::mutex> lock ( s->multi_lock );
    const int world = ( int ) s->devices.size();
    if ( world < 1 || s->extra.size() + 1 != ( size_t ) world ) return fail ( kTerraAmdErrNotCommitted, "the scene's device set changed: commit again" );
    if ( !s->multi ) { s->multi = new MultiCtx(); s->multi->dev.resize ( ( size_t ) world ); for ( int k = 0; k < world; ++k ) s->multi->dev[ ( size_t ) k].device = s->devices[ ( size_t ) k]; }
    MultiCtx& m = *s->multi;
    const size_t npx = w * h;
    std::vector<size_t> counts ( ( size_t ) world ); size_t total = 0;
    for ( int k = 0; k < world; ++k ) { counts[ ( size_t ) k] = ( size_t ) tiles_of_rank ( w, h, tile, k, world ) * tile * tile * 7; total += counts[ ( size_t ) k]; }
    for ( int k = 0; k < world; ++k ) {
        MultiCtx::PerDevice& q = m.dev[ ( size_t ) k];
        HIP_TRY ( hipSetDevice ( q.device ), kTerraAmdErrNoDevice );
        if ( !q.stream ) HIP_TRY ( hipStreamCreateWithFlags ( &q.stream, hipStreamNonBlocking ), kTerraAmdErrNoDevice );
        if ( q.cap_px < npx ) {
            if ( q.d_pixels ) ( void ) hipFree ( q.d_pixels );
            if ( q.d_results ) ( void ) hipFree ( q.d_results );
            q.d_pixels = q.d_results = nullptr; q.cap_px = 0;
            HIP_TRY ( hipMalloc ( &q.d_pixels, npx * 12 ), kTerraAmdErrNoDevice ); HIP_TRY ( hipMalloc ( &q.d_results, npx * 16 ), kTerraAmdErrNoDevice );
            q.cap_px = npx;
        }
        if ( q.packed_floats < counts[ ( size_t ) k] ) {
            if ( q.d_packed ) ( void ) hipFree ( q.d_packed );
            q.d_packed = nullptr; q.packed_floats = 0;
            HIP_TRY ( hipMalloc ( ( void** ) &q.d_packed, counts[ ( size_t ) k] * sizeof ( float ) ), kTerraAmdErrNoDevice );
            q.packed_floats = counts[ ( size_t ) k];
        }
        if ( k == 0 && m.recv_floats < total ) {
            if ( m.d_recv ) ( void ) hipFree ( m.d_recv );
            m.d_recv = nullptr; m.recv_floats = 0;
            HIP_TRY ( hipMalloc ( ( void** ) &m.d_recv, total * sizeof ( float ) ), kTerraAmdErrNoDevice );
            m.recv_floats = total;
        }
    }
    const size_t rpitch = fb->width * 16, ppitch = fb->width * 12;
    const char* hres = ( const char* ) fb->results + ( y * fb->width + x ) * 16;
    char* hpix = ( char* ) fb->pixels + ( y * fb->width + x ) * 12;
    std::vector<DevRenderParams> params ( ( size_t ) world );
    for ( int k = 0; k < world; ++k ) {          // every device: sums up, its tiles rendered, its tiles packed -- queued on its own stream, nothing waits here
        MultiCtx::PerDevice& q = m.dev[ ( size_t ) k];
        HIP_TRY ( hipSetDevice ( q.device ), kTerraAmdErrNoDevice );
        DevRenderParams& p = params[ ( size_t ) k];
        if ( int rc = fill_params ( s, cam, fb->width, fb->height, x, y, w, h, tile, k, world, p, k ? &s->extra[ ( size_t ) k - 1] : nullptr ) ) return rc;
        p.st_x = ( uint32_t ) x; p.st_y = ( uint32_t ) y; p.st_pitch = ( uint32_t ) w;
        p.pixels = ( float* ) q.d_pixels; p.results = q.d_results; p.rand_calls = nullptr;
        HIP_TRY ( hipMemcpy2DAsync ( q.d_results, w * 16, hres, rpitch, w * 16, h, hipMemcpyHostToDevice, q.stream ), kTerraAmdErrLaunch );
        if ( int rc = launch_render ( s, p, q.stream, nullptr, q.device ) ) return rc;
        HIP_TRY ( terra_launch_tiles ( true, ( float* ) q.d_pixels, q.d_results, ( uint32_t ) w, 0, 0, ( uint32_t ) w, ( uint32_t ) h, ( uint32_t ) tile, ( uint32_t ) k, ( uint32_t ) world, q.d_packed, q.stream ), kTerraAmdErrLaunch );
    }
    {   // the one collective
        std::vector<const float*> send ( ( size_t ) world ); std::vector<hipStream_t> streams ( ( size_t ) world );
        for ( int k = 0; k < world; ++k ) { send[ ( size_t ) k] = m.dev[ ( size_t ) k].d_packed; streams[ ( size_t ) k] = m.dev[ ( size_t ) k].stream; }
        std::string err;
        if ( !multigpu::gather_to_first ( s->devices, send, counts, m.d_recv, streams, err ) ) return fail ( kTerraAmdErrLaunch, "%s", err.c_str() );
        ++m.gathers; m.last_gather_bytes = total * sizeof ( float );
    }
    MultiCtx::PerDevice& q0 = m.dev[0];
    HIP_TRY ( hipSetDevice ( q0.device ), kTerraAmdErrNoDevice );
    size_t off = 0;
    for ( int k = 0; k < world; ++k ) {
        HIP_TRY ( terra_launch_tiles ( false, ( float* ) q0.d_pixels, q0.d_results, ( uint32_t ) w, 0, 0, ( uint32_t ) w, ( uint32_t ) h, ( uint32_t ) tile, ( uint32_t ) k, ( uint32_t ) world, m.d_recv + off, q0.stream ), kTerraAmdErrLaunch );
        off += counts[ ( size_t ) k];
    }
    HIP_TRY ( hipMemcpy2DAsync ( ( void* ) hres, rpitch, q0.d_results, w * 16, w * 16, h, hipMemcpyDeviceToHost, q0.stream ), kTerraAmdErrLaunch );
    HIP_TRY ( hipMemcpy2DAsync ( hpix, ppitch, q0.d_pixels, w * 12, w * 12, h, hipMemcpyDeviceToHost, q0.stream ), kTerraAmdErrLaunch );
    HIP_TRY ( hipStreamSynchronize ( q0.stream ), kTerraAmdErrLaunch );
    for ( int k = 1; k < world; ++k ) { HIP_TRY ( hipSetDevice ( m.dev[ ( size_t ) k].device ), kTerraAmdErrNoDevice ); HIP_TRY ( hipStreamSynchronize ( m.dev[ ( size_t ) k].stream ), kTerraAmdErrLaunch ); }
    HIP_TRY ( hipSetDevice ( s->device ), kTerraAmdErrNoDevice );
    for ( int k = 0; k < world; ++k ) account_launch ( s, params[ ( size_t ) k] );
    return 0;
}
extern "C" int terra_amd_render_multi ( const TerraCamera* cam, HTerraScene h, const TerraFramebuffer* fb, size_t x, size_t y, size_t w, size_t hgt, size_t tile ) {
    return render_host_multi ( cam, S ( h ), fb, x, y, w, hgt, tile ? tile : 64 );
}
extern "C" int terra_amd_multi_info ( HTerraScene h, TerraAmdMultiInfo* out ) {
    Scene* s = S ( h );
    if ( !out ) return fail ( kTerraAmdErrBadArgument, "null output" );
    memset ( out, 0, sizeof *out );
    out->devices = ( int ) ( s->devices.empty() ? 1 : s->devices.size() );
    for ( size_t k = 0; k < s->devices.size() && k < 16; ++k ) out->device[k] = s->devices[k];
    out->replicas = s->device_ok ? ( int ) s->extra.size() + 1 : 0;
    std::lock_guard<std::mutex> lock ( s->multi_lock );
    out->gathers = s->multi ? s->multi->gathers : 0; out->last_gather_bytes = s->multi ? s->multi->last_gather_bytes : 0;
    out->process_collectives = multigpu::collectives_issued(); out->rehearsed_gathers = multigpu::gathers_rehearsed(); out->rccl_version = multigpu::rccl_version(); out->communicator_ranks = multigpu::communicator_ranks();
    snprintf ( out->rccl_library, sizeof out->rccl_library, "%s", multigpu::rccl_path().c_str() );
    return 0;
}

extern "C" void terra_render ( const TerraCamera* cam, HTerraScene h, const TerraFramebuffer* fb, size_t x, size_t y, size_t w, size_t hgt ) {
    ( void ) render_host ( cam, S ( h ), fb, x, y, w, hgt );      // failures are recorded in terra_amd_last_error()
}

// ------------------------------------------------------------------------------
// unit-level entry points: host arrays in, device function, host arrays out
// ------------------------------------------------------------------------------
namespace {
struct DevBuf {
    void* p = nullptr; size_t bytes = 0; void* host = nullptr; bool out = false;
    ~DevBuf() { if ( p ) ( void ) hipFree ( p ); }
};
struct Unit {
    std::vector<DevBuf*> bufs; bool ok = true;
    ~Unit() { for ( DevBuf* b : bufs ) delete b; }
    template <class T> T* in ( const T* host, size_t count ) { return ( T* ) make ( ( void* ) host, count * sizeof ( T ), true, false ); }
    template <class T> T* out ( T* host, size_t count ) { return ( T* ) make ( host, count * sizeof ( T ), false, true ); }
    template <class T> T* inout ( T* host, size_t count ) { return ( T* ) make ( host, count * sizeof ( T ), true, true ); }
    void* make ( void* host, size_t bytes, bool copy_in, bool copy_out ) {
        DevBuf* b = new DevBuf(); bufs.push_back ( b );
        b->bytes = bytes; b->host = host; b->out = copy_out;
        if ( hipMalloc ( &b->p, bytes ? bytes : 4 ) != hipSuccess ) { ok = false; return nullptr; }
        if ( copy_in && bytes && hipMemcpy ( b->p, host, bytes, hipMemcpyHostToDevice ) != hipSuccess ) ok = false;
        if ( !copy_in && bytes && hipMemset ( b->p, 0, bytes ) != hipSuccess ) ok = false;
        return b->p;
    }
    int finish ( hipError_t launch ) {
        if ( !ok ) return fail ( kTerraAmdErrNoDevice, "unit call: device allocation/copy failed" );
        if ( launch != hipSuccess ) return fail ( kTerraAmdErrLaunch, "unit kernel launch: %s", hipGetErrorString ( launch ) );
        hipError_t e = hipDeviceSynchronize();
        if ( e != hipSuccess ) return fail ( kTerraAmdErrLaunch, "unit kernel: %s", hipGetErrorString ( e ) );
        for ( DevBuf* b : bufs ) if ( b->out && b->bytes && hipMemcpy ( b->host, b->p, b->bytes, hipMemcpyDeviceToHost ) != hipSuccess ) return fail ( kTerraAmdErrLaunch, "unit copy back failed" );
        return 0;
    }
};
int need_device() { return terra_amd_device_count() > 0 ? 0 : fail ( kTerraAmdErrNoDevice, "no HIP device visible" ); }
int need_scene ( Scene* s ) {
    if ( !s->committed ) return fail ( kTerraAmdErrNotCommitted, "scene not committed" );
    if ( !s->device_ok ) return fail ( kTerraAmdErrNoDevice, "scene has no device replica: %s", s->commit_error.c_str() );
    ( void ) hipSetDevice ( s->device );
    return 0;
}
}

extern "C" int terra_amd_unit_pcg ( const uint32_t* seeds, int nseeds, int n, float* out ) {
    if ( need_device() ) return kTerraAmdErrNoDevice;
    Unit u; auto ds = u.in ( seeds, nseeds ); auto dout = u.out ( out, ( size_t ) nseeds * n );
    return u.finish ( u.ok ? terra_unit_pcg ( ds, nseeds, n, dout ) : hipSuccess );
}
extern "C" int terra_amd_unit_stream_keys ( uint64_t seed, const uint64_t* pix, const uint64_t* k, int n, uint64_t* out3 ) {
    if ( need_device() ) return kTerraAmdErrNoDevice;
    Unit u; auto a = u.in ( pix, n ); auto b = u.in ( k, n ); auto o = u.out ( out3, ( size_t ) 3 * n );
    return u.finish ( u.ok ? terra_unit_stream_keys ( seed, a, b, n, o ) : hipSuccess );
}
extern "C" int terra_amd_unit_ray_aabb ( int n, const float* o, const float* d, const float* boxes, int* hit, float* tmin, float* tmax ) {
    if ( need_device() ) return kTerraAmdErrNoDevice;
    Unit u; auto a = u.in ( o, 3 * ( size_t ) n ); auto b = u.in ( d, 3 * ( size_t ) n ); auto c = u.in ( boxes, 6 * ( size_t ) n );
    auto h = u.out ( hit, n ); auto t0 = u.out ( tmin, n ); auto t1 = u.out ( tmax, n );
    return u.finish ( u.ok ? terra_unit_ray_aabb ( n, a, b, c, h, t0, t1 ) : hipSuccess );
}
extern "C" int terra_amd_unit_watertight ( int n, const float* o, const float* d, const float* tris, int* hit, float* out8 ) {
    if ( need_device() ) return kTerraAmdErrNoDevice;
    Unit u; auto a = u.in ( o, 3 * ( size_t ) n ); auto b = u.in ( d, 3 * ( size_t ) n ); auto c = u.in ( tris, 9 * ( size_t ) n );
    auto h = u.out ( hit, n ); auto q = u.out ( out8, 8 * ( size_t ) n );
    return u.finish ( u.ok ? terra_unit_watertight ( n, a, b, c, h, q ) : hipSuccess );
}
extern "C" int terra_amd_unit_moller_trumbore ( int n, const float* o, const float* d, const float* tris, int* hit, float* out4 ) {
    if ( need_device() ) return kTerraAmdErrNoDevice;
    Unit u; auto a = u.in ( o, 3 * ( size_t ) n ); auto b = u.in ( d, 3 * ( size_t ) n ); auto c = u.in ( tris, 9 * ( size_t ) n );
    auto h = u.out ( hit, n ); auto q = u.out ( out4, 4 * ( size_t ) n );
    return u.finish ( u.ok ? terra_unit_moller_trumbore ( n, a, b, c, h, q ) : hipSuccess );
}
extern "C" int terra_amd_unit_bvh_traverse ( HTerraScene hs, int n, const float* o, const float* d, int* found, uint32_t* prim, float* point3 ) {
    Scene* s = S ( hs ); int rc = need_scene ( s ); if ( rc ) return rc;
    Unit u; auto a = u.in ( o, 3 * ( size_t ) n ); auto b = u.in ( d, 3 * ( size_t ) n );
    auto f = u.out ( found, n ); auto pr = u.out ( prim, n ); auto pt = u.out ( point3, 3 * ( size_t ) n );
    return u.finish ( u.ok ? terra_unit_bvh_traverse ( s->dev, n, a, b, f, pr, pt ) : hipSuccess );
}
extern "C" int terra_amd_unit_bvh_traverse_fast ( HTerraScene hs, int n, const float* o, const float* d, int* found, uint32_t* prim, float* point3, uint32_t* nodes_visited ) {
    Scene* s = S ( hs ); int rc = need_scene ( s ); if ( rc ) return rc;
    if ( !s->dev.fast_nodes_h ) return fail ( kTerraAmdErrBadArgument, "the committed scene has no fast tree (terra_amd_set_tree_mode 1, or 2 on a scene that is not LDS-resident)" );
    Unit u; auto a = u.in ( o, 3 * ( size_t ) n ); auto b = u.in ( d, 3 * ( size_t ) n );
    auto f = u.out ( found, n ); auto pr = u.out ( prim, n ); auto pt = u.out ( point3, 3 * ( size_t ) n ); auto nv = u.out ( nodes_visited, n );
    return u.finish ( u.ok ? terra_unit_bvh_traverse_fast ( s->dev, n, a, b, f, pr, pt, nv ) : hipSuccess );
}
extern "C" int terra_amd_unit_raycast ( HTerraScene hs, int n, const float* o, const float* d, int* obj, int* tri, float* point3, float* surface47 ) {
    Scene* s = S ( hs ); int rc = need_scene ( s ); if ( rc ) return rc;
    Unit u; auto a = u.in ( o, 3 * ( size_t ) n ); auto b = u.in ( d, 3 * ( size_t ) n );
    auto ob = u.out ( obj, n ); auto tr = u.out ( tri, n ); auto pt = u.out ( point3, 3 * ( size_t ) n ); auto sf = u.out ( surface47, 47 * ( size_t ) n );
    return u.finish ( u.ok ? terra_unit_raycast ( s->dev, n, a, b, ob, tr, pt, sf ) : hipSuccess );
}
extern "C" int terra_amd_unit_trace ( HTerraScene hs, int n, const float* o, const float* d, const uint64_t* stateB, const uint64_t* incB, float* radiance3, uint32_t* rand_calls ) {
    Scene* s = S ( hs ); int rc = need_scene ( s ); if ( rc ) return rc;
    if ( ( s->opts.integrator == kTerraIntegratorDirect || s->opts.integrator == kTerraIntegratorDirectMis || s->opts.integrator == kTerraIntegratorDebugMisWeights ) && s->lights.empty() )
        return fail ( kTerraAmdErrBadArgument, "integrator needs a light" );
    Unit u; auto a = u.in ( o, 3 * ( size_t ) n ); auto b = u.in ( d, 3 * ( size_t ) n ); auto sb = u.in ( stateB, n ); auto ib = u.in ( incB, n );
    auto L = u.out ( radiance3, 3 * ( size_t ) n ); auto rcalls = u.out ( rand_calls, n );
    return u.finish ( u.ok ? terra_unit_trace ( s->dev, ( int ) s->opts.integrator, ( uint32_t ) s->opts.bounces, n, a, b, sb, ib, L, rcalls ) : hipSuccess );
}
extern "C" int terra_amd_unit_bsdf ( int kind, int n, float* surfaces47, const float* e3, const float* wo3, float* wi3, float* pdf, float* f3 ) {
    if ( need_device() ) return kTerraAmdErrNoDevice;
    if ( kind < kDevBsdfDiffuse || kind > kDevBsdfGlass ) return fail ( kTerraAmdErrBadArgument, "unknown bsdf kind %d", kind );
    Unit u; auto sf = u.inout ( surfaces47, 47 * ( size_t ) n ); auto e = u.in ( e3, 3 * ( size_t ) n ); auto wo = u.in ( wo3, 3 * ( size_t ) n );
    auto wi = u.out ( wi3, 3 * ( size_t ) n ); auto pd = u.out ( pdf, n ); auto f = u.out ( f3, 3 * ( size_t ) n );
    return u.finish ( u.ok ? terra_unit_bsdf ( kind, n, sf, e, wo, wi, pd, f ) : hipSuccess );
}
extern "C" int terra_amd_unit_camera ( const TerraCamera* cam, size_t fb_w, size_t fb_h, int n, const uint32_t* xy2, float jitter, const float* r2, float* dirs3 ) {
    if ( need_device() ) return kTerraAmdErrNoDevice;
    DevRenderParams p; memset ( &p, 0, sizeof p );
    TerraFloat3 z = terra_normf3 ( &cam->direction );
    TerraFloat3 xa = terra_crossf3 ( &cam->up, &z ); xa = terra_normf3 ( &xa );
    TerraFloat3 ya = terra_crossf3 ( &z, &xa );
    p.cam_rot[0] = xa.x; p.cam_rot[1] = ya.x; p.cam_rot[2] = z.x; p.cam_rot[3] = xa.y; p.cam_rot[4] = ya.y; p.cam_rot[5] = z.y; p.cam_rot[6] = xa.z; p.cam_rot[7] = ya.z; p.cam_rot[8] = z.z;
    p.tan_half_fov = ( float ) tan ( ( double ) ( ( cam->fov * 0.0174533f ) / 2 ) );
    p.aspect = ( float ) fb_w / ( float ) fb_h; p.jitter = jitter; p.fb_w = ( uint32_t ) fb_w; p.fb_h = ( uint32_t ) fb_h;
    Unit u; auto a = u.in ( xy2, 2 * ( size_t ) n ); auto b = u.in ( r2, 2 * ( size_t ) n ); auto o = u.out ( dirs3, 3 * ( size_t ) n );
    return u.finish ( u.ok ? terra_unit_camera ( p, n, a, b, o ) : hipSuccess );
}
extern "C" int terra_amd_unit_tonemap ( int op, float gamma, int n, float* colors3 ) {
    if ( need_device() ) return kTerraAmdErrNoDevice;
    Unit u; auto c = u.inout ( colors3, 3 * ( size_t ) n );
    return u.finish ( u.ok ? terra_unit_tonemap ( op, gamma, n, c ) : hipSuccess );
}
extern "C" int terra_amd_unit_math ( int fn, int n, const float* x, const float* y, float* out ) {
    if ( need_device() ) return kTerraAmdErrNoDevice;
    Unit u; auto a = u.in ( x, n ); auto b = u.in ( y ? y : x, n ); auto o = u.out ( out, n );
    return u.finish ( u.ok ? terra_unit_math ( fn, n, a, b, o ) : hipSuccess );
}

// binary16 planes of the fast tree's nodes (tree_build.cpp fastbvh::half_outward): host arithmetic, no device needed
extern "C" int terra_amd_unit_half_outward ( const double* x, int n, int up, uint16_t* out ) {
    if ( n < 0 || ( n && ( !x || !out ) ) ) return fail ( kTerraAmdErrBadArgument, "null array" );
    for ( int i = 0; i < n; ++i ) out[i] = fastbvh::half_outward ( x[i], up != 0 );
    return 0;
}

// ---- SURVEY.md 8f N4, unit level (reference src/Terra.c:703-755, 760-846) ------------------------------------------------------
extern "C" int terra_amd_unit_stratified ( const uint32_t* seeds, int nseeds, int strata, int samples_per_stratum, int n, float* out2 ) {
    if ( need_device() ) return kTerraAmdErrNoDevice;
    if ( nseeds < 0 || n < 0 || strata < 1 || samples_per_stratum < 1 ) return fail ( kTerraAmdErrBadArgument, "stratified sampler: strata and samples per stratum must be positive" );
    if ( ( long long ) n > ( long long ) strata * strata * samples_per_stratum ) return fail ( kTerraAmdErrBadArgument, "stratified sampler: %d pairs requested but strata^2 * samples = %lld (the reference asserts, src/Terra.c:716)", n, ( long long ) strata * strata * samples_per_stratum );
    Unit u; auto ds = u.in ( seeds, nseeds ); auto o = u.out ( out2, ( size_t ) nseeds * n * 2 );
    return u.finish ( u.ok ? terra_unit_stratified ( ds, nseeds, strata, samples_per_stratum, n, o ) : hipSuccess );
}
extern "C" int terra_amd_unit_halton ( int first, int n, float* out2 ) {
    if ( need_device() ) return kTerraAmdErrNoDevice;
    if ( first < 0 || n < 0 ) return fail ( kTerraAmdErrBadArgument, "halton: negative index" );
    Unit u; auto o = u.out ( out2, ( size_t ) n * 2 );
    return u.finish ( u.ok ? terra_unit_halton ( first, n, o ) : hipSuccess );
}
extern "C" int terra_amd_unit_distribution_1d ( const float* f, size_t n, const float* e, int m, float* x, float* pdf, uint32_t* idx, float* cdf_out, float* integral_out ) {
    if ( need_device() ) return kTerraAmdErrNoDevice;
    if ( n == 0 || n > 0x7fffffffu || m < 0 ) return fail ( kTerraAmdErrBadArgument, "distribution: empty table" );
    std::vector<float> cdf_tmp ( cdf_out ? 0 : n ); float integral_tmp = 0.f; uint32_t mono = 0;
    Unit u; auto df = u.in ( f, n ); auto de = u.in ( e, ( size_t ) m );
    auto dc = u.out ( cdf_out ? cdf_out : cdf_tmp.data(), n ); auto di = u.out ( integral_out ? integral_out : &integral_tmp, 1 ); auto dm = u.out ( &mono, 1 );
    auto dx = u.out ( x, ( size_t ) m ); auto dp = u.out ( pdf, ( size_t ) m ); auto dk = u.out ( idx, ( size_t ) m );
    return u.finish ( u.ok ? terra_unit_distribution_1d ( df, ( uint32_t ) n, dc, di, dm, de, m, dx, dp, dk ) : hipSuccess );
}
extern "C" int terra_amd_unit_distribution_2d ( const float* f, size_t nx, size_t ny, const float* e12, int m, float* xy2, float* pdf, float* marginal_cdf_out ) {
    if ( need_device() ) return kTerraAmdErrNoDevice;
    if ( nx == 0 || ny == 0 || nx * ny > 0x7fffffffu || m < 0 ) return fail ( kTerraAmdErrBadArgument, "distribution: empty table" );
    std::vector<float> cdf ( nx * ny ), integrals ( ny + 1 ), mcdf_tmp ( marginal_cdf_out ? 0 : ny ); std::vector<uint32_t> mono ( ny + 1 );
    Unit u; auto df = u.in ( f, nx * ny ); auto de = u.in ( e12, ( size_t ) m * 2 );
    auto dc = u.out ( cdf.data(), nx * ny ); auto di = u.out ( integrals.data(), ny + 1 ); auto dmc = u.out ( marginal_cdf_out ? marginal_cdf_out : mcdf_tmp.data(), ny ); auto dm = u.out ( mono.data(), ny + 1 );
    auto dxy = u.out ( xy2, ( size_t ) m * 2 ); auto dp = u.out ( pdf, ( size_t ) m );
    return u.finish ( u.ok ? terra_unit_distribution_2d ( df, ( uint32_t ) nx, ( uint32_t ) ny, dc, di, dmc, dm, de, m, dxy, dp ) : hipSuccess );
}
