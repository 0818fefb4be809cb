// adsp_rccl.hip - the one collective of the multi-GPU path: a broadcast of the filter spectrum over RCCL.
//
// libadsp.so does not link librccl: it is opened on first use (dlopen), so the single-GPU product has no RCCL dependency
// and a process that already carries an RCCL (PyTorch-ROCm bundles one) shares it.  The communicator is built with
// ncclCommInitAll - ONE process driving n devices needs no rendezvous - and cached per device list.
// A one-process-per-GPU host (torchrun, MPI, anything that can hand 128 bytes from rank 0 to the others) uses
// rccl_broadcast_rank instead: rank 0 draws an id (ncclGetUniqueId), every rank joins with ncclCommInitRank - RCCL's own
// bootstrap over the loopback / host network, no torch.distributed - and the communicator is cached per (id, rank, world).
// The reference has no counterpart: its devices are independent Python objects that each design their own filter
// (Example2.py:13-14).
#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <mutex>
#include <vector>

#include "../../include/adsp.h"
#include "capi_common.hpp"

namespace {

// the slice of rccl.h this file needs (rccl.h: ncclResult_t / ncclDataType_t are plain enums, ncclComm_t an opaque pointer)
typedef void* comm_t;
constexpr int kNcclSuccess = 0;
constexpr int kNcclFloat32 = 7;
struct UniqueId {  // rccl.h: typedef struct { char internal[NCCL_UNIQUE_ID_BYTES = 128]; } ncclUniqueId - passed BY VALUE
    char internal[ADSP_RCCL_UNIQUE_ID_BYTES];
};
struct Api {
    void* handle = nullptr;
    int (*CommInitAll)(comm_t*, int, const int*) = nullptr;
    int (*GetUniqueId)(UniqueId*) = nullptr;
    int (*CommInitRank)(comm_t*, int, UniqueId, int) = nullptr;
    int (*CommDestroy)(comm_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    int (*Broadcast)(const void*, void*, size_t, int, int, comm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    int (*GetVersion)(int*) = nullptr;
};

std::mutex g_mu;
Api g_api;
std::map<std::vector<int>, std::vector<comm_t>> g_comms;  // device list -> one communicator per device
std::map<std::string, comm_t> g_rank_comms;               // (unique id, rank, world, device) -> this process's communicator

int load_api() {
    if (g_api.handle) return ADSP_OK;
    // an RCCL that is already in the process wins (one RCCL per process, like the one HIP runtime of _capi.py), then
    // ADSP_RCCL_LIB, then the loader's search path
    const char* names[] = {"librccl.so.1", "librccl.so"};
    void* h = nullptr;
    for (const char* n : names)
        if (!h) h = dlopen(n, RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL);
    if (!h)
        if (const char* p = getenv("ADSP_RCCL_LIB")) h = dlopen(p, RTLD_NOW | RTLD_GLOBAL);
    for (const char* n : names)
        if (!h) h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
    if (!h) return adsp::fail(ADSP_ERR_STATE, "cannot open librccl.so (set ADSP_RCCL_LIB): %s", dlerror());
    Api a;
    a.handle = h;
    a.CommInitAll = reinterpret_cast<decltype(a.CommInitAll)>(dlsym(h, "ncclCommInitAll"));
    a.GetUniqueId = reinterpret_cast<decltype(a.GetUniqueId)>(dlsym(h, "ncclGetUniqueId"));
    a.CommInitRank = reinterpret_cast<decltype(a.CommInitRank)>(dlsym(h, "ncclCommInitRank"));
    a.CommDestroy = reinterpret_cast<decltype(a.CommDestroy)>(dlsym(h, "ncclCommDestroy"));
    a.GroupStart = reinterpret_cast<decltype(a.GroupStart)>(dlsym(h, "ncclGroupStart"));
    a.GroupEnd = reinterpret_cast<decltype(a.GroupEnd)>(dlsym(h, "ncclGroupEnd"));
    a.Broadcast = reinterpret_cast<decltype(a.Broadcast)>(dlsym(h, "ncclBroadcast"));
    a.GetErrorString = reinterpret_cast<decltype(a.GetErrorString)>(dlsym(h, "ncclGetErrorString"));
    a.GetVersion = reinterpret_cast<decltype(a.GetVersion)>(dlsym(h, "ncclGetVersion"));
    if (!a.CommInitAll || !a.GetUniqueId || !a.CommInitRank || !a.CommDestroy || !a.GroupStart || !a.GroupEnd || !a.Broadcast || !a.GetErrorString)
        return adsp::fail(ADSP_ERR_STATE, "librccl.so lacks a symbol this library needs");
    g_api = a;
    return ADSP_OK;
}

#define NCCL_TRY(expr)                                                                                  \
    do {                                                                                                \
        const int r__ = (expr);                                                                         \
        if (r__ != kNcclSuccess)                                                                        \
            return adsp::fail(ADSP_ERR_HIP, "%s failed: %s (%s:%d)", #expr, g_api.GetErrorString(r__), __FILE__, __LINE__); \
    } while (0)

}  // namespace

namespace adsp {

// d_buf[i]: `count` floats on device devs[i]; after the call (stream-ordered on streams[i]) every buffer holds root's.
int rccl_broadcast(float* const* d_buf, const int* devs, const hipStream_t* streams, int n, size_t count, int root) {
    std::lock_guard<std::mutex> lock(g_mu);
    int rc = load_api();
    if (rc) return rc;
    std::vector<int> key(devs, devs + n);
    auto it = g_comms.find(key);
    if (it == g_comms.end()) {
        std::vector<comm_t> comms(n, nullptr);
        NCCL_TRY(g_api.CommInitAll(comms.data(), n, devs));
        it = g_comms.emplace(key, comms).first;
    }
    const std::vector<comm_t>& comms = it->second;
    NCCL_TRY(g_api.GroupStart());
    for (int i = 0; i < n; ++i) {
        const int r = g_api.Broadcast(d_buf[i], d_buf[i], count, kNcclFloat32, root, comms[i], streams[i]);
        if (r != kNcclSuccess) {
            (void)g_api.GroupEnd();
            return adsp::fail(ADSP_ERR_HIP, "ncclBroadcast (rank %d of %d) failed: %s", i, n, g_api.GetErrorString(r));
        }
    }
    NCCL_TRY(g_api.GroupEnd());
    return ADSP_OK;
}

// rank 0 of a one-process-per-GPU job: 128 bytes to hand to every other rank (file, environment, MPI, a socket ...)
int rccl_unique_id(char* out) {
    std::lock_guard<std::mutex> lock(g_mu);
    int rc = load_api();
    if (rc) return rc;
    UniqueId id;
    memset(&id, 0, sizeof id);
    NCCL_TRY(g_api.GetUniqueId(&id));
    memcpy(out, id.internal, sizeof id.internal);
    return ADSP_OK;
}

// d_buf: `count` floats on device `dev` of THIS process (rank `rank` of `world`); stream-ordered on `stream`, every rank's
// buffer then holds root's.  The first call with a given id builds the communicator (collective: every rank must call).
// g_mu guards the tables only: ncclCommInitRank and the broadcast BLOCK until every rank has joined, and two ranks driven from
// two threads of one process would otherwise wait for each other for ever (one inside the init, one for the lock).
int rccl_broadcast_rank(const char* unique_id, int rank, int world, int root, int dev, float* d_buf, size_t count, hipStream_t stream) {
    std::string key(unique_id, ADSP_RCCL_UNIQUE_ID_BYTES);
    key += ":" + std::to_string(rank) + "/" + std::to_string(world) + "@" + std::to_string(dev);
    comm_t comm = nullptr;
    {
        std::lock_guard<std::mutex> lock(g_mu);
        int rc = load_api();
        if (rc) return rc;
        auto it = g_rank_comms.find(key);
        if (it != g_rank_comms.end()) comm = it->second;
    }
    if (!comm) {
        UniqueId id;
        memcpy(id.internal, unique_id, sizeof id.internal);
        NCCL_TRY(g_api.CommInitRank(&comm, world, id, rank));  // (g_api is immutable once loaded)
        std::lock_guard<std::mutex> lock(g_mu);
        auto ins = g_rank_comms.emplace(key, comm);
        if (!ins.second) {  // another thread joined as the very same rank meanwhile (a caller error that RCCL let through): keep the first
            (void)g_api.CommDestroy(comm);
            comm = ins.first->second;
        }
    }
    NCCL_TRY(g_api.Broadcast(d_buf, d_buf, count, kNcclFloat32, root, comm, stream));
    return ADSP_OK;
}

// Destroys every communicator this library has built (process exit, or before a job re-forms with new ids).  The caller
// guarantees that no broadcast is in flight.
int rccl_finalize() {
    std::lock_guard<std::mutex> lock(g_mu);
    if (!g_api.handle) return ADSP_OK;
    int bad = kNcclSuccess;
    for (auto& kv : g_rank_comms) {
        const int r = g_api.CommDestroy(kv.second);
        if (r != kNcclSuccess) bad = r;
    }
    g_rank_comms.clear();
    for (auto& kv : g_comms)
        for (comm_t c : kv.second) {
            const int r = g_api.CommDestroy(c);
            if (r != kNcclSuccess) bad = r;
        }
    g_comms.clear();
    if (bad != kNcclSuccess) return adsp::fail(ADSP_ERR_HIP, "ncclCommDestroy failed: %s", g_api.GetErrorString(bad));
    return ADSP_OK;
}

int rccl_version(int* version) {
    std::lock_guard<std::mutex> lock(g_mu);
    int rc = load_api();
    if (rc) return rc;
    *version = 0;
    if (g_api.GetVersion) NCCL_TRY(g_api.GetVersion(version));
    return ADSP_OK;
}

}  // namespace adsp
