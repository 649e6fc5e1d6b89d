// sanitizer harness only: host-side stand-ins for this repo's own kernel launchers (no device here)
#include <hip/hip_runtime.h>
#include "kernels.h"
hipError_t terra_launch_render ( const DevRenderParams&, hipStream_t ) { return hipErrorNoDevice; }
uint32_t   terra_render_blocks ( const DevRenderParams& ) { return 0; }
hipError_t terra_launch_resolve ( const DevRenderParams&, hipStream_t ) { return hipErrorNoDevice; }
bool       terra_scene_fits_lds ( uint32_t n_nodes, uint32_t n_tris, int ) { return n_nodes * 64 + n_tris * 112 < 8192; }
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
