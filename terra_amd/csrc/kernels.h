// kernels.h -- launchers implemented in the .hip files, called by the host side.
#pragma once
#include <hip/hip_runtime.h>
#include "dev_types.h"

hipError_t terra_launch_render ( const DevRenderParams& p, hipStream_t stream );
hipError_t terra_launch_job_streams ( const DevRenderParams& p, hipStream_t stream );      // before terra_launch_render: DevRenderParams::job_streams
size_t terra_block_order_bytes ( const DevRenderParams& p, bool small_too );        // scratch of the job order (class + order words per pixel block); 0: the launch keeps the order of the numbering (p.job_blocks, p.lds_mode set); small_too: also below TERRA_JOB_ORDER_MIN_BLOCKS
uint32_t   terra_job_order_min_blocks ( void );                     // launches of fewer pixel blocks keep the numbering's order (unless terra_amd_set_job_order(scene, 2))
hipError_t terra_launch_block_order ( const DevRenderParams& p, uint32_t* cls, hipStream_t stream );      // before terra_launch_job_streams: fills p.block_order (= cls + blocks)
size_t terra_job_streams_bytes ( const DevRenderParams& p );        // 0: this launch keys its streams in the render kernel (p.lds_mode, p.job_blocks set)
// (render_kernels.hip is compiled as several translation units, one per template MODE: 0 reference tree from global memory, 1 LDS-resident, 2 fast tree, 3 fast tree + reachability replay)
hipError_t terra_launch_render_mode0 ( const DevRenderParams& p, size_t lds, hipStream_t stream );
hipError_t terra_launch_render_mode1 ( const DevRenderParams& p, size_t lds, hipStream_t stream );
hipError_t terra_launch_render_mode2 ( const DevRenderParams& p, size_t lds, hipStream_t stream );
hipError_t terra_launch_render_mode3 ( const DevRenderParams& p, size_t lds, hipStream_t stream );
bool       terra_render_wants_queue ( const DevRenderParams& p );   // the launch's loop gains from the persistent grid + job queue (render_kernels.hip "jobs")
uint32_t   terra_render_blocks ( const DevRenderParams& p );   // 256-thread blocks of one chunk (own tiles x blocks per tile)
hipError_t terra_launch_resolve ( const DevRenderParams& p, hipStream_t stream );   // second kernel of a split render (p.split > 1)
bool       terra_scene_fits_lds ( uint32_t n_nodes, uint32_t n_tris, int max_stack, uint32_t n_objects, uint32_t n_lights );   // whole scene staged per block (the small-scene kernels)
void       terra_plan_lds ( DevRenderParams& p );     // fills stack_depth / lds_nodes / lds_tris / lds_mode
size_t     terra_lds_bytes ( const DevRenderParams& p );   // dynamic LDS per block of the planned launch
size_t     terra_lds_block_limit ( void );                // the most a block may ask for (launch_instance opts in above 64 KB)
size_t     terra_fast_spill_bytes ( const DevRenderParams& p );   // bytes of DevRenderParams::stack_spill a fast-tree launch needs (p.job_blocks set; 0: the stack fits in LDS)
void       terra_plan_fast_tree ( DevRenderParams& p );     // the plan of a fast-tree (MODE 2 / 3) launch: stack from the tree's depth, nothing staged
hipError_t terra_launch_tiles ( bool pack, float* pixels, void* results, uint32_t fb_w, uint32_t x, uint32_t y, uint32_t w, uint32_t h,
                                uint32_t tile, uint32_t rank, uint32_t world, float* packed, hipStream_t stream );

hipError_t terra_fill_sincos24 ( float2* table, hipStream_t stream );    // DevScene::sincos24: 2^24 entries (128 MB), device pointer

// unit-level launchers: all pointers are DEVICE pointers, n items, synchronous semantics left to the caller
hipError_t terra_unit_pcg ( const uint32_t* seeds, int nseeds, int n, float* out );
hipError_t terra_unit_stream_keys ( uint64_t frame_seed, const uint64_t* pix, const uint64_t* k, int n, uint64_t* out3 );
hipError_t terra_unit_ray_aabb ( int n, const float* o, const float* d, const float* boxes, int* hit, float* tmin, float* tmax );
hipError_t terra_unit_watertight ( int n, const float* o, const float* d, const float* tris, int* hit, float* out8 );
hipError_t terra_unit_moller_trumbore ( int n, const float* o, const float* d, const float* tris, int* hit, float* out4 );
hipError_t terra_unit_bvh_traverse ( const DevScene& sc, int n, const float* o, const float* d, int* found, uint32_t* prim, float* point );
hipError_t terra_unit_bvh_traverse_fast ( const DevScene& sc, int n, const float* o, const float* d, int* found, uint32_t* prim, float* point, uint32_t* nodes_visited );
hipError_t terra_unit_raycast ( const DevScene& sc, int n, const float* o, const float* d, int* obj, int* tri, float* point, float* surface47 );
hipError_t terra_unit_trace ( const DevScene& sc, int integrator, uint32_t bounces, int n, const float* o, const float* d,
                              const uint64_t* stateB, const uint64_t* incB, float* radiance, uint32_t* rand_calls );
hipError_t terra_unit_bsdf ( int kind, int n, float* surfaces47, const float* e3, const float* wo3, float* wi3, float* pdf, float* f3 );
hipError_t terra_unit_camera ( const DevRenderParams& p, int n, const uint32_t* xy2, const float* r2, float* dirs3 );
hipError_t terra_unit_tonemap ( int op, float gamma, int n, float* colors3 );
hipError_t terra_unit_math ( int fn, int n, const float* x, const float* y, float* out );
// SURVEY.md 8f N4 (sampling_device.h): cdf / integrals / monotone are device scratch the caller provides (sizes in scene_host.cpp)
hipError_t terra_unit_stratified ( const uint32_t* seeds, int nseeds, int strata, int samples, int n, float* out2 );
hipError_t terra_unit_halton ( int first, int n, float* out2 );
hipError_t terra_unit_distribution_1d ( const float* f, uint32_t n, float* cdf, float* integral, uint32_t* monotone, const float* e, int m, float* x, float* pdf, uint32_t* idx );
hipError_t terra_unit_distribution_2d ( const float* f, uint32_t nx, uint32_t ny, float* cdf, float* integrals, float* mcdf, uint32_t* monotone, const float* e12, int m, float* xy2, float* pdf );
// the fast tree built on the GPU (tree_build_device.hip): device pointers; out_nodes holds up to n - 1 nodes, out_tris n triangles
// extra_margin: added to the +-1e-4 triangle boxes on every side (0 inside the coordinate range; the rounding bound of the reachability mode outside it)
hipError_t terra_build_fast_tree_device ( const DevTri* tris, const uint32_t* rank, uint32_t n, float extra_margin, DevNode* out_nodes, DevTri* out_tris, uint32_t* n_nodes_out, int* max_stack_out, hipStream_t stream );
