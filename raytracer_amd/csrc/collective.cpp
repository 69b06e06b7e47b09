// collective.cpp - the exchange step of a multi-GPU render for single-process callers (part of
// libclray_hip.so): RCCL over xGMI, without torch.
//
// The reference has one device and no collective (/root/reference/accelerator.py:24-25); the path
// shards because the image is a sum over passes (/root/reference/renderer.py:24-28).  A host that
// drives several GPUs from one process (one clr_ctx per GPU, as the reference's single Python
// process would) sums the per-GPU accumulators, or gathers disjoint tiles, with these calls;
// bench.py and clray.py --gpus use one process per GPU and torch.distributed instead
// (raytracer_amd/multi_gpu.py), which is the same RCCL underneath.
//
// RCCL is loaded with dlopen on first use, not linked: a process that never calls these entry
// points never maps it, and a process that already has an RCCL mapped (PyTorch brings its own
// librccl.so) keeps using that one.
#include "clray_hip.h"

#include <hip/hip_runtime.h>
#include <dlfcn.h>

#include <mutex>
#include <string>
#include <vector>

extern "C" __attribute__((visibility("hidden"))) void clr_set_error(const char *text);
extern "C" __attribute__((visibility("hidden"))) int clr_ctx_device(const clr_ctx *ctx);

namespace {

typedef void *nccl_comm_t;
enum { kNcclSum = 0, kNcclFloat = 7 };     // ncclSum, ncclFloat32 (rccl.h)

struct Rccl {
    void *handle = nullptr;
    int (*CommInitAll)(nccl_comm_t *, int, const int *) = nullptr;
    int (*CommDestroy)(nccl_comm_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    int (*AllReduce)(const void *, void *, size_t, int, int, nccl_comm_t, hipStream_t) = nullptr;
    int (*AllGather)(const void *, void *, size_t, int, nccl_comm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
};

Rccl g_rccl;
std::mutex g_rccl_mutex;

int fail(const std::string &what) {
    clr_set_error(what.c_str());
    return 1;
}

int load_rccl() {
    std::lock_guard<std::mutex> lock(g_rccl_mutex);
    if (g_rccl.handle) return 0;
    void *h = nullptr;
    for (const char *name : {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"}) {
        h = dlopen(name, RTLD_NOW | RTLD_LOCAL);
        if (h) break;
    }
    if (!h) return fail(std::string("cannot load librccl.so: ") + dlerror());
    Rccl r;
    r.handle = h;
#define CLR_SYM(field, symbol) \
    *(void **)(&r.field) = dlsym(h, symbol); \
    if (!r.field) { dlclose(h); return fail(std::string("librccl.so lacks ") + symbol); }
    CLR_SYM(CommInitAll, "ncclCommInitAll")
    CLR_SYM(CommDestroy, "ncclCommDestroy")
    CLR_SYM(GroupStart, "ncclGroupStart")
    CLR_SYM(GroupEnd, "ncclGroupEnd")
    CLR_SYM(AllReduce, "ncclAllReduce")
    CLR_SYM(AllGather, "ncclAllGather")
    CLR_SYM(GetErrorString, "ncclGetErrorString")
#undef CLR_SYM
    g_rccl = r;
    return 0;
}

int check_nccl(int rc, const char *what) {
    if (rc == 0) return 0;
    return fail(std::string(what) + ": " + (g_rccl.GetErrorString ? g_rccl.GetErrorString(rc) : "RCCL error"));
}

// the collective calls make every rank's device current in turn; the caller's current device is
// put back when they return
struct DeviceRestorer {
    int device = -1;
    DeviceRestorer() { if (hipGetDevice(&device) != hipSuccess) device = -1; }
    ~DeviceRestorer() { if (device >= 0) (void)hipSetDevice(device); }
};

}  // namespace

struct clr_comm {
    std::vector<int> devices;
    std::vector<nccl_comm_t> comms;
};

extern "C" {

int clr_comm_create(clr_ctx *const *ctxs, int n, clr_comm **out) {
    if (!ctxs || !out || n < 1) return fail("clr_comm_create: bad arguments");
    DeviceRestorer keep_device;
    std::vector<int> devices;
    for (int i = 0; i < n; ++i) {
        if (!ctxs[i]) return fail("clr_comm_create: null context");
        const int d = clr_ctx_device(ctxs[i]);
        for (int seen : devices)
            if (seen == d) return fail("clr_comm_create: two contexts on one GPU (RCCL needs one rank per device)");
        devices.push_back(d);
    }
    if (int rc = load_rccl()) return rc;
    clr_comm *comm = nullptr;
    try {
        comm = new clr_comm;
        comm->devices = devices;
        comm->comms.assign(n, nullptr);
    } catch (...) {
        delete comm;
        return fail("clr_comm_create: out of memory");
    }
    if (int rc = check_nccl(g_rccl.CommInitAll(comm->comms.data(), n, comm->devices.data()), "ncclCommInitAll")) {
        delete comm;
        return rc;
    }
    *out = comm;
    return 0;
}

int clr_comm_destroy(clr_comm *comm) {
    if (!comm) return 0;
    for (nccl_comm_t c : comm->comms)
        if (c) (void)g_rccl.CommDestroy(c);
    delete comm;
    return 0;
}

int clr_comm_size(const clr_comm *comm, int *n) {
    if (!comm || !n) return fail("clr_comm_size: null argument");
    *n = (int)comm->comms.size();
    return 0;
}

int clr_allreduce_sum_f32(clr_comm *comm, void *const *buffers, size_t count, void *const *streams) {
    if (!comm || !buffers) return fail("clr_allreduce_sum_f32: null argument");
    const int n = (int)comm->comms.size();
    for (int i = 0; i < n; ++i)
        if (!buffers[i]) return fail("clr_allreduce_sum_f32: null buffer");
    DeviceRestorer keep_device;
    if (int rc = check_nccl(g_rccl.GroupStart(), "ncclGroupStart")) return rc;
    int first_error = 0;
    for (int i = 0; i < n && !first_error; ++i) {
        if (hipSetDevice(comm->devices[i]) != hipSuccess) { first_error = -1; break; }
        first_error = g_rccl.AllReduce(buffers[i], buffers[i], count, kNcclFloat, kNcclSum, comm->comms[i],
                                       streams ? (hipStream_t)streams[i] : nullptr);
    }
    const int end = g_rccl.GroupEnd();
    if (first_error == -1) return fail("clr_allreduce_sum_f32: hipSetDevice failed");
    if (first_error) return check_nccl(first_error, "ncclAllReduce");
    return check_nccl(end, "ncclGroupEnd");
}

int clr_allgather_f32(clr_comm *comm, const void *const *send, void *const *recv, size_t count_per_rank,
                      void *const *streams) {
    if (!comm || !send || !recv) return fail("clr_allgather_f32: null argument");
    const int n = (int)comm->comms.size();
    for (int i = 0; i < n; ++i)
        if (!send[i] || !recv[i]) return fail("clr_allgather_f32: null buffer");
    DeviceRestorer keep_device;
    if (int rc = check_nccl(g_rccl.GroupStart(), "ncclGroupStart")) return rc;
    int first_error = 0;
    for (int i = 0; i < n && !first_error; ++i) {
        if (hipSetDevice(comm->devices[i]) != hipSuccess) { first_error = -1; break; }
        first_error = g_rccl.AllGather(send[i], recv[i], count_per_rank, kNcclFloat, comm->comms[i],
                                       streams ? (hipStream_t)streams[i] : nullptr);
    }
    const int end = g_rccl.GroupEnd();
    if (first_error == -1) return fail("clr_allgather_f32: hipSetDevice failed");
    if (first_error) return check_nccl(first_error, "ncclAllGather");
    return check_nccl(end, "ncclGroupEnd");
}

}  // extern "C"
