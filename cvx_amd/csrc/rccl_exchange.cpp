// Built-in implementation of the exchange hook (cvx_allreduce_fn) on RCCL over xGMI:
// one communicator per process / GPU, one ncclAllReduce(sum, double) per call, enqueued on the
// stream the handle runs on.  The reference has no counterpart (it is single-process,
// single-threaded); SURVEY.md section 8(e) defines the exchange step.
//
// RCCL is resolved at run time (dlopen) so that libcvx_amd.so loads in processes that never
// shard -- and so that a process that already carries an RCCL (PyTorch's) shares that copy.
#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <cstring>
#include <mutex>
#include <string>

#include "common.h"

namespace {

// the slice of the RCCL API used here (nccl.h: ncclUniqueId is 128 bytes, ncclDouble = 8,
// ncclSum = 0, ncclSuccess = 0)
struct UniqueId { char internal[128]; };
typedef void *Comm;
typedef int (*fn_get_unique_id)(UniqueId *);
typedef int (*fn_comm_init_rank)(Comm *, int, UniqueId, int);
typedef int (*fn_all_reduce)(const void *, void *, size_t, int, int, Comm, hipStream_t);
typedef int (*fn_comm_destroy)(Comm);
typedef const char *(*fn_get_error_string)(int);

struct Api {
    void *lib = nullptr;
    fn_get_unique_id get_unique_id = nullptr;
    fn_comm_init_rank comm_init_rank = nullptr;
    fn_all_reduce all_reduce = nullptr;
    fn_comm_destroy comm_destroy = nullptr;
    fn_get_error_string get_error_string = nullptr;
    bool ok = false;
};

Api &api() {
    static Api a;
    static std::once_flag once;
    std::call_once(once, [] {
        const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
        for (const char *n : names) {
            a.lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
            if (a.lib) break;
        }
        if (!a.lib) return;
        a.get_unique_id = (fn_get_unique_id)dlsym(a.lib, "ncclGetUniqueId");
        a.comm_init_rank = (fn_comm_init_rank)dlsym(a.lib, "ncclCommInitRank");
        a.all_reduce = (fn_all_reduce)dlsym(a.lib, "ncclAllReduce");
        a.comm_destroy = (fn_comm_destroy)dlsym(a.lib, "ncclCommDestroy");
        a.get_error_string = (fn_get_error_string)dlsym(a.lib, "ncclGetErrorString");
        a.ok = a.get_unique_id && a.comm_init_rank && a.all_reduce && a.comm_destroy;
    });
    return a;
}

int fail(const char *what, int code) {
    Api &a = api();
    std::string s = std::string(what) + " failed";
    if (a.get_error_string && code) s += std::string(": ") + a.get_error_string(code);
    cvx::set_last_error(s);
    return CVX_ERR_EXCHANGE;
}

struct RcclComm {
    Comm comm = nullptr;
    int device = 0, nranks = 1;
    double *staging = nullptr;   // device staging for host-buffer calls
    size_t staging_count = 0;
    hipStream_t stream = nullptr;
};

}  // namespace

static_assert(CVX_RCCL_ID_BYTES == sizeof(UniqueId), "ncclUniqueId is 128 bytes");

extern "C" int cvx_rccl_unique_id(char id[CVX_RCCL_ID_BYTES]) {
    if (!id) return CVX_ERR_INVALID_ARG;
    Api &a = api();
    if (!a.ok) { cvx::set_last_error("librccl not found"); return CVX_ERR_EXCHANGE; }
    UniqueId u;
    int rc = a.get_unique_id(&u);
    if (rc) return fail("ncclGetUniqueId", rc);
    std::memcpy(id, u.internal, sizeof u.internal);
    return CVX_OK;
}

extern "C" int cvx_rccl_create(int rank, int nranks, const char id[CVX_RCCL_ID_BYTES], int device, void **comm) {
    if (!comm || !id || nranks < 1 || rank < 0 || rank >= nranks) return CVX_ERR_INVALID_ARG;
    *comm = nullptr;
    Api &a = api();
    if (!a.ok) { cvx::set_last_error("librccl not found"); return CVX_ERR_EXCHANGE; }
    CVX_HIP(hipSetDevice(device));
    RcclComm *c = new RcclComm;
    c->device = device; c->nranks = nranks;
    UniqueId u;
    std::memcpy(u.internal, id, sizeof u.internal);
    int rc = a.comm_init_rank(&c->comm, nranks, u, rank);
    if (rc) { delete c; return fail("ncclCommInitRank", rc); }
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) {
        a.comm_destroy(c->comm);
        delete c;
        return CVX_ERR_HIP;
    }
    *comm = c;
    return CVX_OK;
}

// matches cvx_allreduce_fn: ctx is the communicator returned by cvx_rccl_create
extern "C" int cvx_rccl_allreduce(void *ctx, double *buf, size_t count, int on_device, void *stream) {
    RcclComm *c = (RcclComm *)ctx;
    if (!c || !buf) return CVX_ERR_INVALID_ARG;
    if (count == 0) return CVX_OK;
    Api &a = api();
    CVX_HIP(hipSetDevice(c->device));
    if (on_device) {
        int rc = a.all_reduce(buf, buf, count, /*ncclDouble*/ 8, /*ncclSum*/ 0, c->comm, (hipStream_t)stream);
        if (rc) return fail("ncclAllReduce", rc);
        return CVX_OK;
    }
    // host buffer: stage through device memory on the communicator's own stream
    if (c->staging_count < count) {
        if (c->staging) (void)hipFree(c->staging);
        c->staging = nullptr; c->staging_count = 0;
        CVX_HIP(hipMalloc((void **)&c->staging, count * sizeof(double)));
        c->staging_count = count;
    }
    CVX_HIP(hipMemcpyAsync(c->staging, buf, count * sizeof(double), hipMemcpyHostToDevice, c->stream));
    int rc = a.all_reduce(c->staging, c->staging, count, 8, 0, c->comm, c->stream);
    if (rc) return fail("ncclAllReduce", rc);
    CVX_HIP(hipMemcpyAsync(buf, c->staging, count * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    CVX_HIP(hipStreamSynchronize(c->stream));
    return CVX_OK;
}

extern "C" void cvx_rccl_destroy(void *ctx) {
    RcclComm *c = (RcclComm *)ctx;
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->staging) (void)hipFree(c->staging);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    if (c->comm) api().comm_destroy(c->comm);
    delete c;
}
