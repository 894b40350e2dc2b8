// Counter all-reduce over the GPUs of one node through RCCL, for callers of the C ABI that run every GPU from one
// process (the Python driver uses torch.distributed, one process per GPU).  RCCL is loaded when the first communicator
// is made (dlopen), so the library carries no link-time dependency on it.
#include <dlfcn.h>
#include <stdio.h>

#include <new>
#include <vector>

#include "polar_internal.h"

namespace {

// the few RCCL entry points used, with the NCCL ABI (rccl.h)
typedef struct ncclComm *ncclComm_t;
typedef int ncclResult_t;
enum { NCCL_UINT64 = 5, NCCL_SUM = 0 };  // ncclUint64, ncclSum
struct Rccl {
    void *handle = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
};
Rccl g_rccl;

int load_rccl()
{
    if (g_rccl.handle) return POLAR_OK;
    void *h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!h) {
        polar_set_error("RCCL is not available: %s", dlerror());
        return POLAR_EUNSUPPORTED;
    }
    Rccl r;
    r.handle = h;
    r.CommInitAll = (decltype(r.CommInitAll))dlsym(h, "ncclCommInitAll");
    r.CommDestroy = (decltype(r.CommDestroy))dlsym(h, "ncclCommDestroy");
    r.AllReduce = (decltype(r.AllReduce))dlsym(h, "ncclAllReduce");
    r.GroupStart = (decltype(r.GroupStart))dlsym(h, "ncclGroupStart");
    r.GroupEnd = (decltype(r.GroupEnd))dlsym(h, "ncclGroupEnd");
    r.GetErrorString = (decltype(r.GetErrorString))dlsym(h, "ncclGetErrorString");
    if (!r.CommInitAll || !r.CommDestroy || !r.AllReduce || !r.GroupStart || !r.GroupEnd || !r.GetErrorString) {
        polar_set_error("RCCL library lacks an expected entry point");
        dlclose(h);
        return POLAR_EUNSUPPORTED;
    }
    g_rccl = r;
    return POLAR_OK;
}

#define RCCL_TRY(expr)                                                                         \
    do {                                                                                       \
        ncclResult_t r__ = (expr);                                                             \
        if (r__ != 0) {                                                                        \
            polar_set_error("%s failed: %s", #expr, g_rccl.GetErrorString(r__));               \
            return POLAR_EHIP;                                                                 \
        }                                                                                      \
    } while (0)

}  // namespace

struct polar_comm {
    std::vector<int> devices;
    std::vector<ncclComm_t> comms;
    std::vector<hipStream_t> streams;
};

extern "C" int polar_comm_init_all(int ndev, const int *devices, polar_comm **out)
{
    ARG_TRY(out != nullptr, "out is NULL");
    int visible = 0;
    HIP_TRY(hipGetDeviceCount(&visible));
    ARG_TRY(ndev >= 1 && ndev <= visible, "ndev must be between 1 and the number of visible GPUs");
    polar_comm *c = new (std::nothrow) polar_comm();
    if (!c) return POLAR_ENOMEM;
    for (int i = 0; i < ndev; i++) {
        const int d = devices ? devices[i] : i;
        if (d < 0 || d >= visible) {
            delete c;
            polar_set_error("invalid argument: device %d out of range (%d visible)", d, visible);
            return POLAR_EINVAL;
        }
        for (int j = 0; j < i; j++)
            if (c->devices[j] == d) {
                delete c;
                polar_set_error("invalid argument: device %d listed twice", d);
                return POLAR_EINVAL;
            }
        c->devices.push_back(d);
    }
    int rc = load_rccl();
    if (rc) {
        delete c;
        return rc;
    }
    c->comms.resize(ndev);
    ncclResult_t r = g_rccl.CommInitAll(c->comms.data(), ndev, c->devices.data());
    if (r != 0) {
        polar_set_error("ncclCommInitAll failed: %s", g_rccl.GetErrorString(r));
        delete c;
        return POLAR_EHIP;
    }
    c->streams.assign(ndev, nullptr);
    for (int i = 0; i < ndev; i++) {
        // a failure here gives back the communicators and the streams made so far (polar_comm_destroy skips null entries)
        hipError_t e = hipSetDevice(c->devices[i]);
        if (e == hipSuccess) e = polar_fault_at("comm_init") ? hipErrorOutOfMemory : hipStreamCreateWithFlags(&c->streams[i], hipStreamNonBlocking);
        if (e != hipSuccess) {
            c->streams[i] = nullptr;
            polar_set_error("stream for device %d: %s", c->devices[i], hipGetErrorString(e));
            (void)polar_comm_destroy(c);
            return POLAR_EHIP;
        }
    }
    *out = c;
    return POLAR_OK;
}

extern "C" int polar_comm_destroy(polar_comm *c)
{
    if (!c) return POLAR_OK;
    for (size_t i = 0; i < c->comms.size(); i++) {
        (void)hipSetDevice(c->devices[i]);
        if (i < c->streams.size() && c->streams[i]) {
            (void)hipStreamSynchronize(c->streams[i]);
            (void)hipStreamDestroy(c->streams[i]);
        }
        if (c->comms[i]) (void)g_rccl.CommDestroy(c->comms[i]);
    }
    delete c;
    return POLAR_OK;
}

extern "C" int polar_comm_size(const polar_comm *c) { return c ? (int)c->devices.size() : 0; }

extern "C" int polar_allreduce_counters(polar_comm *c, uint64_t *const *dev_counters, int n)
{
    ARG_TRY(c && dev_counters, "NULL argument");
    ARG_TRY(n >= 1, "n must be positive");
    const int ndev = (int)c->devices.size();
    for (int i = 0; i < ndev; i++) ARG_TRY(dev_counters[i] != nullptr, "NULL counter buffer");
    // work queued on the devices' default streams (e.g. by polar_ctx_sync'ed decoder calls) has to be finished by the caller
    RCCL_TRY(g_rccl.GroupStart());
    // once the group is open every path reaches GroupEnd: an open group would swallow this thread's later collectives
    int rc = POLAR_OK;
    for (int i = 0; i < ndev && rc == POLAR_OK; i++) {
        const hipError_t e = hipSetDevice(c->devices[i]);
        if (e != hipSuccess) {
            polar_set_error("hipSetDevice(%d) failed: %s", c->devices[i], hipGetErrorString(e));
            rc = POLAR_EHIP;
            break;
        }
        const ncclResult_t r = polar_fault_at("allreduce") ? 1 : g_rccl.AllReduce(dev_counters[i], dev_counters[i], (size_t)n, NCCL_UINT64, NCCL_SUM, c->comms[i], c->streams[i]);
        if (r != 0) {
            polar_set_error("ncclAllReduce on device %d failed: %s", c->devices[i], g_rccl.GetErrorString(r));
            rc = POLAR_EHIP;
        }
    }
    const ncclResult_t rend = g_rccl.GroupEnd();
    if (rc != POLAR_OK) return rc;  // the message of the first failure stays
    if (rend != 0) {
        polar_set_error("ncclGroupEnd failed: %s", g_rccl.GetErrorString(rend));
        return POLAR_EHIP;
    }
    for (int i = 0; i < ndev; i++) {
        HIP_TRY(hipSetDevice(c->devices[i]));
        HIP_TRY(hipStreamSynchronize(c->streams[i]));
    }
    return POLAR_OK;
}
