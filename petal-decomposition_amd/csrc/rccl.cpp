// rccl.cpp -- the built-in collective: RCCL all-reduce on the ctx stream, bound at run time.
//
// The sample-sharded path only ever all-reduces small replicated fp64 buffers.  A host that does
// not want a callback in that loop hands the library a ncclUniqueId instead: every rank calls petal_ctx_init_rccl()
// collectively and the library issues ncclAllReduce itself, in place, on the ctx stream (no host hop, no Python).
// RCCL is dlopen()ed -- first the copy already mapped into the process (PyTorch-ROCm ships its own librccl next to its
// own HIP runtime, and two RCCL/HIP runtimes must not be mixed), then the system one -- so libpetal_hip.so carries no
// link-time dependency on it and the host-simulation build of the same sources loads on a CPU-only machine.
#include <dlfcn.h>
#include <link.h>

#include <cstdio>
#include <mutex>

#include "ctx.h"

namespace petal {
namespace {

struct NcclUniqueId { char internal[128]; };  // ncclUniqueId (nccl.h: NCCL_UNIQUE_ID_BYTES = 128)
typedef void* NcclComm;
enum { kNcclFloat32 = 7, kNcclFloat64 = 8 };                 // ncclDataType_t
enum { kNcclSum = 0, kNcclMax = 2, kNcclMin = 3 };           // ncclRedOp_t

struct RcclApi {
    void* handle = nullptr;
    int (*GetUniqueId)(NcclUniqueId*) = nullptr;
    int (*CommInitRank)(NcclComm*, int, NcclUniqueId, int) = nullptr;
    int (*AllReduce)(const void*, void*, size_t, int, int, NcclComm, void*) = nullptr;
    int (*CommDestroy)(NcclComm) = nullptr;
    int (*CommCount)(NcclComm, int*) = nullptr;        // optional: what the communicator itself says about its size, device, rank
    int (*CommCuDevice)(NcclComm, int*) = nullptr;
    int (*CommUserRank)(NcclComm, int*) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    std::string why, path;
};

// the librccl the process has ALREADY mapped (e.g. the one PyTorch-ROCm bundles), by walking the loaded objects
int find_loaded_rccl(struct dl_phdr_info* info, size_t, void* out) {
    if (info->dlpi_name && std::strstr(info->dlpi_name, "librccl.so")) {
        *static_cast<std::string*>(out) = info->dlpi_name;
        return 1;
    }
    return 0;
}

RcclApi& api() {
    static RcclApi a;
    static std::once_flag once;
    std::call_once(once, [] {
        std::string loaded;
        dl_iterate_phdr(find_loaded_rccl, &loaded);
        if (!loaded.empty()) a.handle = dlopen(loaded.c_str(), RTLD_NOW | RTLD_GLOBAL);
        if (a.handle) a.path = loaded;
        const char* paths[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
        for (const char* n : paths)
            if (!a.handle) { a.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL); if (a.handle) a.path = n; }
        if (!a.handle) { a.why = "librccl.so not found"; return; }
        a.GetUniqueId = reinterpret_cast<decltype(a.GetUniqueId)>(dlsym(a.handle, "ncclGetUniqueId"));
        a.CommInitRank = reinterpret_cast<decltype(a.CommInitRank)>(dlsym(a.handle, "ncclCommInitRank"));
        a.AllReduce = reinterpret_cast<decltype(a.AllReduce)>(dlsym(a.handle, "ncclAllReduce"));
        a.CommDestroy = reinterpret_cast<decltype(a.CommDestroy)>(dlsym(a.handle, "ncclCommDestroy"));
        a.GetErrorString = reinterpret_cast<decltype(a.GetErrorString)>(dlsym(a.handle, "ncclGetErrorString"));
        a.CommCount = reinterpret_cast<decltype(a.CommCount)>(dlsym(a.handle, "ncclCommCount"));
        a.CommCuDevice = reinterpret_cast<decltype(a.CommCuDevice)>(dlsym(a.handle, "ncclCommCuDevice"));
        a.CommUserRank = reinterpret_cast<decltype(a.CommUserRank)>(dlsym(a.handle, "ncclCommUserRank"));
        if (!a.GetUniqueId || !a.CommInitRank || !a.AllReduce || !a.CommDestroy) a.why = "librccl.so lacks the nccl* entry points";
    });
    return a;
}

std::string nccl_error(int rc) {
    RcclApi& a = api();
    return std::string("RCCL error ") + std::to_string(rc) + (a.GetErrorString ? std::string(": ") + a.GetErrorString(rc) : "");
}

struct RcclComm { NcclComm comm = nullptr; };

// petal_allreduce_fn: in place on the ctx stream
int rccl_allreduce(void* user, void* buf, int64_t count, int dtype, int op, void* stream) {
    RcclComm* c = static_cast<RcclComm*>(user);
    const int dt = dtype == F32 ? kNcclFloat32 : kNcclFloat64;
    const int rop = op == PETAL_MAX ? kNcclMax : (op == PETAL_MIN ? kNcclMin : kNcclSum);
    return api().AllReduce(buf, buf, size_t(count), dt, rop, c->comm, stream);
}

}  // namespace

void rccl_unique_id(void* out128) {
    RcclApi& a = api();
    if (!a.why.empty()) device_error("built-in collective unavailable: " + a.why);
    NcclUniqueId id;
    const int rc = a.GetUniqueId(&id);
    if (rc != 0) device_error(nccl_error(rc));
    std::memcpy(out128, id.internal, sizeof(id.internal));
}

// what the built-in communicator reports about itself (ncclCommCount / ncclCommCuDevice / ncclCommUserRank): -1 where there is
// no built-in communicator or the library lacks the query.  A scaling record can then show that RCCL really spanned N ranks.
void rccl_info(const petal_ctx& c, int* count, int* device, int* rank) {
    *count = *device = *rank = -1;
    if (!c.rccl) return;
    const RcclComm* rc = static_cast<const RcclComm*>(c.rccl);
    RcclApi& a = api();
    int v = -1;
    if (a.CommCount && a.CommCount(rc->comm, &v) == 0) *count = v;
    if (a.CommCuDevice && a.CommCuDevice(rc->comm, &v) == 0) *device = v;
    if (a.CommUserRank && a.CommUserRank(rc->comm, &v) == 0) *rank = v;
}

void rccl_release(petal_ctx& c) {
    if (!c.rccl) return;
    RcclComm* rc = static_cast<RcclComm*>(c.rccl);
    if (rc->comm && api().CommDestroy) (void)api().CommDestroy(rc->comm);
    delete rc;
    c.rccl = nullptr;
    if (c.allreduce == &rccl_allreduce) { c.allreduce = nullptr; c.allreduce_user = nullptr; c.rank = 0; c.world = 1; }
}

void rccl_init(petal_ctx& c, const void* unique_id128, int rank, int world) {
    if (world < 1 || rank < 0 || rank >= world) invalid_input("bad rank / world_size");
    if (!unique_id128) invalid_input("unique id must not be null");
    RcclApi& a = api();
    if (!a.why.empty()) device_error("built-in collective unavailable: " + a.why);
    rccl_release(c);
    dev_make_current(c.dev);  // the communicator binds to the calling thread's current device
    NcclUniqueId id;
    std::memcpy(id.internal, unique_id128, sizeof(id.internal));
    RcclComm* rc = new RcclComm();
    const int e = a.CommInitRank(&rc->comm, world, id, rank);  // collective: every rank is in here together
    if (e != 0) { delete rc; device_error(nccl_error(e)); }
    c.rccl = rc;
    c.allreduce = &rccl_allreduce;
    c.allreduce_user = rc;
    c.rank = rank;
    c.world = world;
    if (std::getenv("PETAL_DEBUG")) std::fprintf(stderr, "[petal] rank %d/%d: built-in collective on %s\n", rank, world, a.path.c_str());
}

}  // namespace petal
