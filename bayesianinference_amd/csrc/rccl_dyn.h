// rccl_dyn.h -- RCCL bound at run time (dlopen), so libgphip.so carries no link-time dependency on it.
//
// Why not -lrccl: a host process may already hold a copy of RCCL (PyTorch-ROCm wheels bundle their own
// librccl.so next to their own HIP runtime).  Two RCCLs -- or an RCCL linked against a second
// libamdhip64 -- in one process do not work.  So the library first looks for an RCCL that is ALREADY
// mapped (RTLD_NOLOAD) -- unless $GPHIP_RCCL_PATH names one, which wins --, then for the system one.  A single-GPU handle never
// touches this file's code; a multi-device handle without any usable RCCL falls back to peer copies
// (CopyComm in gphip_multi.inc) when all its ranks live in this process, and fails loudly otherwise.
#pragma once
#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <mutex>
#include <string>

namespace gphip {

// the handful of declarations used, with RCCL's ABI (rccl.h: ncclUniqueId is 128 opaque bytes; ncclChar = 0,
// ncclFloat64 = 8, ncclSum = 0, ncclMax = 2; every call returns ncclResult_t, 0 = ncclSuccess)
struct NcclUniqueId { char internal[128]; };
typedef struct ncclComm* nccl_comm_t;
constexpr int NCCL_CHAR = 0, NCCL_FLOAT64 = 8, NCCL_SUM = 0, NCCL_MAX = 2;

struct RcclApi {
    void* so = nullptr;
    std::string origin;          // which library was bound (for gphip_last_error / diagnostics)
    int (*GetUniqueId)(NcclUniqueId*) = nullptr;
    int (*CommInitRank)(nccl_comm_t*, int, NcclUniqueId, int) = nullptr;
    int (*CommInitAll)(nccl_comm_t*, int, const int*) = nullptr;
    int (*CommDestroy)(nccl_comm_t) = nullptr;
    int (*Broadcast)(const void*, void*, size_t, int, int, nccl_comm_t, hipStream_t) = nullptr;
    int (*AllReduce)(const void*, void*, size_t, int, int, nccl_comm_t, hipStream_t) = nullptr;
    // optional (the two-hop panel broadcast: scatter by grouped send / recv, then an in-place all-gather)
    int (*Send)(const void*, size_t, int, int, nccl_comm_t, hipStream_t) = nullptr;
    int (*Recv)(void*, size_t, int, int, nccl_comm_t, hipStream_t) = nullptr;
    int (*AllGather)(const void*, void*, size_t, int, nccl_comm_t, hipStream_t) = nullptr;
    bool two_hop_ok() const { return Send && Recv && AllGather; }
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    bool ok() const { return so != nullptr; }
};

inline const RcclApi& rccl() {
    static RcclApi api;
    static std::once_flag once;
    std::call_once(once, [] {
        struct Cand { const char* name; int flags; };
        const char* env = getenv("GPHIP_RCCL_PATH");
        const Cand cands[] = {
            {env, RTLD_NOW},                             // an explicit $GPHIP_RCCL_PATH wins
            {"librccl.so", RTLD_NOW | RTLD_NOLOAD},      // PyTorch's bundled copy, if the host process mapped it
            {"librccl.so.1", RTLD_NOW | RTLD_NOLOAD},
            {"librccl.so.1", RTLD_NOW},
            {"librccl.so", RTLD_NOW},
            {"/opt/rocm/lib/librccl.so.1", RTLD_NOW},
        };
        for (const Cand& c : cands) {
            if (!c.name || !*c.name) continue;
            void* so = dlopen(c.name, c.flags);
            if (!so) continue;
            RcclApi a;
            a.so = so;
            a.origin = c.name;
#define GP_SYM(field, sym) a.field = reinterpret_cast<decltype(a.field)>(dlsym(so, sym))
            GP_SYM(GetUniqueId, "ncclGetUniqueId");
            GP_SYM(CommInitRank, "ncclCommInitRank");
            GP_SYM(CommInitAll, "ncclCommInitAll");
            GP_SYM(CommDestroy, "ncclCommDestroy");
            GP_SYM(Broadcast, "ncclBroadcast");
            GP_SYM(AllReduce, "ncclAllReduce");
            GP_SYM(Send, "ncclSend");
            GP_SYM(Recv, "ncclRecv");
            GP_SYM(AllGather, "ncclAllGather");
            GP_SYM(GroupStart, "ncclGroupStart");
            GP_SYM(GroupEnd, "ncclGroupEnd");
            GP_SYM(GetErrorString, "ncclGetErrorString");
#undef GP_SYM
            if (a.GetUniqueId && a.CommInitRank && a.CommInitAll && a.CommDestroy && a.Broadcast && a.AllReduce &&
                a.GroupStart && a.GroupEnd && a.GetErrorString) {
                api = a;
                return;
            }
            dlclose(so);
        }
    });
    return api;
}

}  // namespace gphip
