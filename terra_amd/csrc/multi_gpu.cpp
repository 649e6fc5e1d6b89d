// multi_gpu.cpp -- the one collective of the in-process multi-GPU render (multi_gpu.h): every device's packed tiles to the first device.
//
// RCCL is loaded at run time, on the first multi-device call (dlopen of librccl.so.1): a client that renders on one GPU never
// needs it, and libterra_amd.so carries no link-time dependency on it. One communicator per device of the selected set, made
// with ncclCommInitAll (one process drives all devices -- the layout the reference's client has: one process, tiles dealt to
// workers, satellite/src/Renderer.cpp:316-350) and kept until the set changes.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>          // types and enumerators only: every entry point is resolved with dlsym
#include <dlfcn.h>
#include <atomic>
#include <cstdio>
#include <mutex>
#include "multi_gpu.h"

namespace multigpu {
namespace {
struct Api {
    void* handle = nullptr; std::string path;
    ncclResult_t ( *CommInitAll ) ( ncclComm_t*, int, const int* ) = nullptr;
    ncclResult_t ( *CommDestroy ) ( ncclComm_t ) = nullptr;
    ncclResult_t ( *GroupStart ) () = nullptr;
    ncclResult_t ( *GroupEnd ) () = nullptr;
    ncclResult_t ( *Send ) ( const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t ) = nullptr;
    ncclResult_t ( *Recv ) ( void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t ) = nullptr;
    const char* ( *GetErrorString ) ( ncclResult_t ) = nullptr;
    ncclResult_t ( *GetVersion ) ( int* ) = nullptr;
};
std::mutex g_lock;
Api g_api;
std::vector<int> g_comm_devices;          // the set the cached communicators were made for
std::vector<ncclComm_t> g_comms;
std::atomic<uint64_t> g_collectives { 0 }, g_rehearsed { 0 };

bool load_api ( std::string& err ) {
    if ( g_api.handle ) return true;
    const char* names[] = { "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1" };
    for ( const char* n : names ) { g_api.handle = dlopen ( n, RTLD_NOW | RTLD_LOCAL ); if ( g_api.handle ) { g_api.path = n; break; } }
    if ( !g_api.handle ) { err = std::string ( "RCCL is not loadable (librccl.so.1): " ) + ( dlerror() ? dlerror() : "?" ); return false; }
    auto sym = [&] ( const char* s ) { void* p = dlsym ( g_api.handle, s ); if ( !p && err.empty() ) err = std::string ( "RCCL lacks " ) + s; return p; };
    g_api.CommInitAll = ( decltype ( g_api.CommInitAll ) ) sym ( "ncclCommInitAll" ); g_api.CommDestroy = ( decltype ( g_api.CommDestroy ) ) sym ( "ncclCommDestroy" );
    g_api.GroupStart = ( decltype ( g_api.GroupStart ) ) sym ( "ncclGroupStart" ); g_api.GroupEnd = ( decltype ( g_api.GroupEnd ) ) sym ( "ncclGroupEnd" );
    g_api.Send = ( decltype ( g_api.Send ) ) sym ( "ncclSend" ); g_api.Recv = ( decltype ( g_api.Recv ) ) sym ( "ncclRecv" );
    g_api.GetErrorString = ( decltype ( g_api.GetErrorString ) ) sym ( "ncclGetErrorString" ); g_api.GetVersion = ( decltype ( g_api.GetVersion ) ) sym ( "ncclGetVersion" );
    if ( !err.empty() ) { dlclose ( g_api.handle ); g_api = Api(); return false; }
    return true;
}
void drop_comms_locked() {
    if ( g_api.CommDestroy ) for ( ncclComm_t c : g_comms ) if ( c ) ( void ) g_api.CommDestroy ( c );
    g_comms.clear(); g_comm_devices.clear();
}
bool comms_for ( const std::vector<int>& devices, std::string& err ) {
    if ( g_comm_devices == devices && g_comms.size() == devices.size() ) return true;
    drop_comms_locked();
    g_comms.assign ( devices.size(), nullptr );
    const ncclResult_t r = g_api.CommInitAll ( g_comms.data(), ( int ) devices.size(), devices.data() );
    if ( r != ncclSuccess ) { err = std::string ( "ncclCommInitAll: " ) + g_api.GetErrorString ( r ); g_comms.clear(); return false; }
    g_comm_devices = devices;
    return true;
}
} // namespace

bool gather_to_first ( const std::vector<int>& devices, const std::vector<const float*>& send, const std::vector<size_t>& counts, float* recv_on_first,
                       const std::vector<hipStream_t>& streams, std::string& err ) {
    const size_t n = devices.size();
    if ( n == 0 || send.size() != n || counts.size() != n || streams.size() != n ) { err = "gather: inconsistent arguments"; return false; }
    std::lock_guard<std::mutex> g ( g_lock );
    // REHEARSAL (terra_amd_debug_replicas_share_device: a device listed more than once, which RCCL refuses -- one communicator rank per device): the collective's
    // stand-in is one device-to-device copy per replica on that replica's stream, and the first stream waits for the others. Everything around the transport -- replicas,
    // tile dealing, pack, offsets, unpack -- is the code several devices run; the transport itself is NOT RCCL here and is not counted as a collective.
    bool shared = false;
    for ( size_t i = 0; i < n && !shared; ++i ) for ( size_t j = 0; j < i; ++j ) if ( devices[i] == devices[j] ) { shared = true; break; }
    if ( shared ) {
        size_t off = 0;
        for ( size_t k = 0; k < n; ++k ) {
            if ( counts[k] ) {
                hipError_t e = hipMemcpyAsync ( recv_on_first + off, send[k], counts[k] * sizeof ( float ), hipMemcpyDeviceToDevice, streams[k] );
                hipEvent_t ev = nullptr;
                if ( e == hipSuccess && k != 0 ) { e = hipEventCreateWithFlags ( &ev, hipEventDisableTiming ); if ( e == hipSuccess ) e = hipEventRecord ( ev, streams[k] ); if ( e == hipSuccess ) e = hipStreamWaitEvent ( streams[0], ev, 0 ); if ( ev ) ( void ) hipEventDestroy ( ev ); }
                if ( e != hipSuccess ) { err = std::string ( "rehearsal gather: " ) + hipGetErrorString ( e ); return false; }
            }
            off += counts[k];
        }
        g_rehearsed.fetch_add ( 1, std::memory_order_relaxed );
        return true;
    }
    if ( !load_api ( err ) || !comms_for ( devices, err ) ) return false;
    // one group: rank r sends its packed tiles to rank 0, rank 0 receives every rank's -- its own included, a copy inside device 0, so that one rank
    // runs exactly the calls that eight run
    ncclResult_t r = g_api.GroupStart();
    size_t off = 0;
    for ( size_t k = 0; k < n && r == ncclSuccess; ++k ) if ( counts[k] ) r = g_api.Send ( send[k], counts[k], ncclFloat, 0, g_comms[k], streams[k] );
    for ( size_t k = 0; k < n && r == ncclSuccess; ++k ) { if ( counts[k] ) r = g_api.Recv ( recv_on_first + off, counts[k], ncclFloat, ( int ) k, g_comms[0], streams[0] ); off += counts[k]; }
    const ncclResult_t e = g_api.GroupEnd();
    if ( r == ncclSuccess ) r = e;
    if ( r != ncclSuccess ) { err = std::string ( "RCCL gather: " ) + g_api.GetErrorString ( r ); return false; }
    g_collectives.fetch_add ( 1, std::memory_order_relaxed );
    return true;
}

void forget_communicators() { std::lock_guard<std::mutex> g ( g_lock ); drop_comms_locked(); }
uint64_t collectives_issued() { return g_collectives.load ( std::memory_order_relaxed ); }
uint64_t gathers_rehearsed() { return g_rehearsed.load ( std::memory_order_relaxed ); }
int rccl_version() { std::lock_guard<std::mutex> g ( g_lock ); int v = 0; if ( g_api.GetVersion ) ( void ) g_api.GetVersion ( &v ); return v; }
std::string rccl_path() { std::lock_guard<std::mutex> g ( g_lock ); return g_api.path; }
int communicator_ranks() { std::lock_guard<std::mutex> g ( g_lock ); return ( int ) g_comms.size(); }
} // namespace multigpu
