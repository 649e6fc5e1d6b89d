// multi_gpu.h -- the collective behind terra_amd_render_multi (scene_host.cpp): RCCL, loaded at run time (multi_gpu.cpp).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <string>
#include <vector>

namespace multigpu {
// Rank k (= devices[k]) sends counts[k] floats from send[k] (memory of devices[k]) on streams[k]; devices[0] receives them one after the other,
// in rank order, into recv_on_first (its memory) on streams[0] -- its own share too. One ncclGroupStart / ncclGroupEnd around all of it; returns
// once the operations are queued on the streams. false + err when RCCL cannot be loaded or reports an error.
bool gather_to_first ( const std::vector<int>& devices, const std::vector<const float*>& send, const std::vector<size_t>& counts, float* recv_on_first,
                       const std::vector<hipStream_t>& streams, std::string& err );
void forget_communicators();          // destroys the cached communicators (the device set changed)
uint64_t collectives_issued();        // gathers issued by this process so far
uint64_t gathers_rehearsed();         // ... and gathers that went through the rehearsal's stand-in instead (a device listed twice: terra_amd_debug_replicas_share_device)
int rccl_version();                   // ncclGetVersion of the loaded library (0: not loaded yet)
std::string rccl_path();              // the name it was loaded by
int communicator_ranks();             // ranks of the cached communicator (0: none)
}
