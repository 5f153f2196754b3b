// RCCL bound with dlopen (see hcv_rccl.h).

#include "hcv_rccl.h"

#include <dlfcn.h>

#include <cstring>

#include <mutex>

namespace hcv
{

namespace
{
    struct UniqueId { char internal[kRcclIdBytes]; };   // ncclUniqueId
    typedef void *Comm;                                 // ncclComm_t
    enum { kFloat32 = 7, kSum = 0, kSuccess = 0 };      // ncclFloat, ncclSum, ncclSuccess (rccl.h)

    struct Api
    {
        void *lib = nullptr;
        int (*GetUniqueId)(UniqueId *) = nullptr;
        int (*CommInitRank)(Comm *, int, UniqueId, int) = nullptr;
        int (*CommDestroy)(Comm) = nullptr;
        int (*AllReduce)(const void *, void *, size_t, int, int, Comm, hipStream_t) = nullptr;
        int (*GroupStart)() = nullptr;
        int (*GroupEnd)() = nullptr;
        const char *(*GetErrorString)(int) = nullptr;
        std::string error;
    };

    Api &api()
    {
        static Api a;
        static std::once_flag once;
        std::call_once(once, []()
        {
            // a copy the process already holds first (PyTorch's), then the system's
            const char *names[] = { "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1" };
            for (const char *n : names)
                if ((a.lib = dlopen(n, RTLD_NOW | RTLD_NOLOAD))) break;
            if (!a.lib)
                for (const char *n : names)
                    if ((a.lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL))) break;
            if (!a.lib)
            {
                const char *why = dlerror();                  // (cleared by the call: read once)
                a.error = std::string("librccl could not be loaded: ") + (why ? why : "not found");
                return;
            }
            auto sym = [&](const char *name) { return dlsym(a.lib, name); };
            a.GetUniqueId = reinterpret_cast<int (*)(UniqueId *)>(sym("ncclGetUniqueId"));
            a.CommInitRank = reinterpret_cast<int (*)(Comm *, int, UniqueId, int)>(sym("ncclCommInitRank"));
            a.CommDestroy = reinterpret_cast<int (*)(Comm)>(sym("ncclCommDestroy"));
            a.AllReduce = reinterpret_cast<int (*)(const void *, void *, size_t, int, int, Comm, hipStream_t)>(sym("ncclAllReduce"));
            a.GroupStart = reinterpret_cast<int (*)()>(sym("ncclGroupStart"));
            a.GroupEnd = reinterpret_cast<int (*)()>(sym("ncclGroupEnd"));
            a.GetErrorString = reinterpret_cast<const char *(*)(int)>(sym("ncclGetErrorString"));
            if (!a.GetUniqueId || !a.CommInitRank || !a.CommDestroy || !a.AllReduce || !a.GroupStart || !a.GroupEnd)
            {
                a.error = "librccl lacks an expected symbol";
                a.lib = nullptr;
            }
        });
        return a;
    }

    bool ok(int rc, const char *what, std::string *err)
    {
        if (rc == kSuccess) return true;
        Api &a = api();
        if (err) *err = std::string(what) + ": " + (a.GetErrorString ? a.GetErrorString(rc) : "rccl error");
        return false;
    }
}

struct RcclComm
{
    Comm comm = nullptr;
    int rank = 0, nranks = 1, device = 0;
};

bool rccl_available(std::string *err)
{
    Api &a = api();
    if (!a.lib && err) *err = a.error;
    return a.lib != nullptr;
}

bool rccl_unique_id(void *out128, std::string *err)
{
    if (!rccl_available(err)) return false;
    UniqueId id;
    if (!ok(api().GetUniqueId(&id), "ncclGetUniqueId", err)) return false;
    std::memcpy(out128, id.internal, kRcclIdBytes);
    return true;
}

RcclComm *rccl_comm_create(const void *id128, int rank, int nranks, int device, std::string *err)
{
    if (!rccl_available(err)) return nullptr;
    int prev = -1;
    (void) hipGetDevice(&prev);
    (void) hipSetDevice(device);
    UniqueId id;
    std::memcpy(id.internal, id128, kRcclIdBytes);
    RcclComm *c = new RcclComm();
    c->rank = rank;
    c->nranks = nranks;
    c->device = device;
    const bool good = ok(api().CommInitRank(&c->comm, nranks, id, rank), "ncclCommInitRank", err);
    if (prev >= 0 && prev != device) (void) hipSetDevice(prev);
    if (!good)
    {
        delete c;
        return nullptr;
    }
    return c;
}

void rccl_comm_destroy(RcclComm *c)
{
    if (!c) return;
    if (c->comm && api().lib) (void) api().CommDestroy(c->comm);
    delete c;
}

int rccl_comm_size(const RcclComm *c) { return c ? c->nranks : 0; }

bool rccl_all_reduce_sum(RcclComm *c, float *buf, size_t rows, size_t n, size_t stride, hipStream_t stream, std::string *err)
{
    if (!c || !rows || !n) return true;
    Api &a = api();
    if (stride == n || rows == 1) return ok(a.AllReduce(buf, buf, rows * n, kFloat32, kSum, c->comm, stream), "ncclAllReduce", err);
    if (!ok(a.GroupStart(), "ncclGroupStart", err)) return false;
    bool good = true;
    for (size_t r = 0; r < rows && good; r++) good = ok(a.AllReduce(buf + r * stride, buf + r * stride, n, kFloat32, kSum, c->comm, stream), "ncclAllReduce", err);
    return ok(a.GroupEnd(), "ncclGroupEnd", err) && good;
}

} // namespace hcv
