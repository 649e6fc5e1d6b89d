// sanitizer harness only: host-side stand-ins for this repo's own kernel launchers (no device here)
#include <hip/hip_runtime.h>
#include "kernels.h"
hipError_t terra_launch_render ( const DevRenderParams&, hipStream_t ) { return hipErrorNoDevice; }
hipError_t terra_launch_job_streams ( const DevRenderParams&, hipStream_t ) { return hipErrorNoDevice; }
size_t terra_job_streams_bytes ( const DevRenderParams& ) { return 0; }
hipError_t terra_fill_sincos24 ( float2*, hipStream_t ) { return hipErrorNoDevice; }
bool       terra_render_wants_queue ( const DevRenderParams& ) { return false; }
uint32_t   terra_render_blocks ( const DevRenderParams& ) { return 0; }
hipError_t terra_launch_resolve ( const DevRenderParams&, hipStream_t ) { return hipErrorNoDevice; }
bool       terra_scene_fits_lds ( uint32_t n_nodes, uint32_t n_tris, int, uint32_t, uint32_t ) { return n_nodes * 64 + n_tris * 112 < 8192; }
void       terra_plan_lds ( DevRenderParams& ) {}
hipError_t terra_launch_tiles ( bool, float*, void*, uint32_t, uint32_t, uint32_t, uint32_t, uint32_t, uint32_t, uint32_t, uint32_t, float*, hipStream_t ) { return hipErrorNoDevice; }
hipError_t terra_unit_pcg ( const uint32_t*, int, int, float* ) { return hipErrorNoDevice; }
hipError_t terra_unit_stream_keys ( uint64_t, const uint64_t*, const uint64_t*, int, uint64_t* ) { return hipErrorNoDevice; }
hipError_t terra_unit_ray_aabb ( int, const float*, const float*, const float*, int*, float*, float* ) { return hipErrorNoDevice; }
hipError_t terra_unit_watertight ( int, const float*, const float*, const float*, int*, float* ) { return hipErrorNoDevice; }
hipError_t terra_unit_moller_trumbore ( int, const float*, const float*, const float*, int*, float* ) { return hipErrorNoDevice; }
hipError_t terra_unit_bvh_traverse ( const DevScene&, int, const float*, const float*, int*, uint32_t*, float* ) { return hipErrorNoDevice; }
hipError_t terra_unit_raycast ( const DevScene&, int, const float*, const float*, int*, int*, float*, float* ) { return hipErrorNoDevice; }
hipError_t terra_unit_trace ( const DevScene&, int, uint32_t, int, const float*, const float*, const uint64_t*, const uint64_t*, float*, uint32_t* ) { return hipErrorNoDevice; }
hipError_t terra_unit_bsdf ( int, int, float*, const float*, const float*, float*, float*, float* ) { return hipErrorNoDevice; }
hipError_t terra_unit_camera ( const DevRenderParams&, int, const uint32_t*, const float*, float* ) { return hipErrorNoDevice; }
hipError_t terra_unit_tonemap ( int, float, int, float* ) { return hipErrorNoDevice; }
hipError_t terra_unit_math ( int, int, const float*, const float*, float* ) { return hipErrorNoDevice; }
hipError_t terra_unit_bvh_traverse_fast ( const DevScene&, int, const float*, const float*, int*, uint32_t*, float*, uint32_t* ) { return hipErrorNoDevice; }
hipError_t terra_unit_stratified ( const uint32_t*, int, int, int, int, float* ) { return hipErrorNoDevice; }
hipError_t terra_unit_halton ( int, int, float* ) { return hipErrorNoDevice; }
hipError_t terra_unit_distribution_1d ( const float*, uint32_t, float*, float*, uint32_t*, const float*, int, float*, float*, uint32_t* ) { return hipErrorNoDevice; }
hipError_t terra_unit_distribution_2d ( const float*, uint32_t, uint32_t, float*, float*, float*, uint32_t*, const float*, int, float*, float* ) { return hipErrorNoDevice; }
hipError_t terra_build_fast_tree_device ( const DevTri*, const uint32_t*, uint32_t, float, DevNode*, DevTri*, uint32_t*, int*, hipStream_t ) { return hipErrorNoDevice; }
void terra_plan_fast_tree ( DevRenderParams& p ) { p.lds_mode = 2; p.lds_tris = 0; p.leaf_cap = 0; p.stack_depth = 1; p.lds_nodes = 0; p.spill_cap = 0; p.stack_spill = nullptr; }
size_t terra_fast_spill_bytes ( const DevRenderParams& ) { return 0; }
size_t terra_lds_bytes ( const DevRenderParams& ) { return 0; }
size_t terra_lds_block_limit ( void ) { return 156 * 1024; }
