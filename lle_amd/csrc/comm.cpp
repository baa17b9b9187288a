// comm.cpp -- the ONE collective of the path, at the C-ABI level: the end-of-batch all-reduce of the rollout counters
// (BASELINE north_star "RCCL over xGMI only for the end-of-batch reduction of throughput / reward
// statistics").  The reference has no collective at all (a single-process Rust library), so there is no file to match:
// this is what a host bound to include/lle_hip.h -- Rust, C, one process per GPU or one process owning a handle per GPU --
// calls where the Python host calls torch.distributed.all_reduce (lle_amd/distributed.py).
//
// librccl is loaded with dlopen on first use: a host that never reduces (one GPU) does not pay for it, and inside a
// Python process the already loaded RCCL of torch (same SONAME) is the one that answers.  Message: 64 bytes -- latency
// bound, the xGMI link bandwidth (7 x ~153 GB/s) is irrelevant; no ring / bucket tuning applies.
#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <string>

#include "../../include/lle_hip.h"
#include "capi_internal.hpp"

namespace {

struct NcclUniqueId { char internal[LLE_COMM_ID_BYTES]; };  // rccl.h: NCCL_UNIQUE_ID_BYTES = 128
typedef void* ncclComm_t;
typedef int ncclResult_t;
constexpr int NCCL_INT64 = 4, NCCL_SUM = 0, NCCL_MAX = 2;  // rccl.h ncclDataType_t / ncclRedOp_t

struct Rccl {
    void* handle = nullptr;
    std::string path, error;
    ncclResult_t (*GetUniqueId)(NcclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, NcclUniqueId, int) = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
};

Rccl* rccl() {
    static Rccl R;
    static std::once_flag once;
    std::call_once(once, [] {
        const char* env = std::getenv("LLE_RCCL_LIB");
        const char* names[] = {env, "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
        for (const char* n : names) {
            if (!n || !*n) continue;
            R.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
            if (R.handle) { R.path = n; break; }
            R.error = dlerror();
        }
        if (!R.handle) return;
#define LLE_SYM(field, name)                                                        \
        R.field = reinterpret_cast<decltype(R.field)>(dlsym(R.handle, name));       \
        if (!R.field) { R.error = std::string("librccl lacks ") + name; dlclose(R.handle); R.handle = nullptr; return; }
        LLE_SYM(GetUniqueId, "ncclGetUniqueId")
        LLE_SYM(CommInitRank, "ncclCommInitRank")
        LLE_SYM(CommInitAll, "ncclCommInitAll")
        LLE_SYM(CommDestroy, "ncclCommDestroy")
        LLE_SYM(AllReduce, "ncclAllReduce")
        LLE_SYM(GroupStart, "ncclGroupStart")
        LLE_SYM(GroupEnd, "ncclGroupEnd")
        LLE_SYM(GetErrorString, "ncclGetErrorString")
#undef LLE_SYM
    });
    return R.handle ? &R : nullptr;
}

int no_rccl() {
    return lle::capi_fail(LLE_ERR_UNSUPPORTED, "librccl could not be loaded (set LLE_RCCL_LIB to its path): multi-GPU reductions need RCCL");
}

#define RCCL_TRY(R, expr)                                                                                            \
    do {                                                                                                             \
        ncclResult_t r_ = (expr);                                                                                    \
        if (r_ != 0) return lle::capi_fail(LLE_ERR_HIP, std::string(#expr) + ": " + (R)->GetErrorString(r_));        \
    } while (0)
#define HIP_TRY_C(expr)                                                                                              \
    do {                                                                                                             \
        hipError_t e_ = (expr);                                                                                      \
        if (e_ != hipSuccess) return lle::capi_fail(LLE_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)

}  // namespace

struct lle_comm {
    ncclComm_t comm = nullptr;
    int n_ranks = 0, rank = 0, device = 0;
    int64_t* scratch = nullptr;  // 8 x i64 of device memory: the reduction runs in place on it
};

namespace {
struct DeviceScopeC {
    int prev = -1;
    bool switched = false;
    explicit DeviceScopeC(int dev) {
        if (hipGetDevice(&prev) == hipSuccess && prev != dev) switched = hipSetDevice(dev) == hipSuccess;
    }
    ~DeviceScopeC() { if (switched) (void)hipSetDevice(prev); }
};

int finish_comm(lle_comm* c) {
    DeviceScopeC scope(c->device);
    void* p = nullptr;
    HIP_TRY_C(hipMalloc(&p, 8 * sizeof(int64_t)));
    c->scratch = static_cast<int64_t*>(p);
    return LLE_OK;
}
}  // namespace

extern "C" {

int lle_comm_unique_id(uint8_t id[LLE_COMM_ID_BYTES]) {
    if (!id) return lle::capi_fail(LLE_ERR_NULL, "NULL id");
    Rccl* R = rccl();
    if (!R) return no_rccl();
    NcclUniqueId u;
    RCCL_TRY(R, R->GetUniqueId(&u));
    std::memcpy(id, u.internal, LLE_COMM_ID_BYTES);
    return lle::capi_ok();
}

lle_comm* lle_comm_create(const uint8_t id[LLE_COMM_ID_BYTES], int n_ranks, int rank, int device_id) {
    if (!id) { lle::capi_fail(LLE_ERR_NULL, "NULL id"); return nullptr; }
    if (n_ranks < 1 || rank < 0 || rank >= n_ranks) { lle::capi_fail(LLE_ERR_ARG, "rank / n_ranks out of range"); return nullptr; }
    Rccl* R = rccl();
    if (!R) { no_rccl(); return nullptr; }
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || device_id < 0 || device_id >= n_dev) {
        lle::capi_fail(n_dev > 0 ? LLE_ERR_ARG : LLE_ERR_NO_DEVICE, n_dev > 0 ? "device_id out of range" : "no HIP device");
        return nullptr;
    }
    lle_comm* c = new (std::nothrow) lle_comm();
    if (!c) { lle::capi_fail(LLE_ERR_HIP, "out of memory"); return nullptr; }
    c->n_ranks = n_ranks; c->rank = rank; c->device = device_id;
    {
        DeviceScopeC scope(device_id);  // ncclCommInitRank binds the communicator to the CURRENT device
        NcclUniqueId u;
        std::memcpy(u.internal, id, LLE_COMM_ID_BYTES);
        ncclResult_t r = R->CommInitRank(&c->comm, n_ranks, u, rank);
        if (r != 0) {
            lle::capi_fail(LLE_ERR_HIP, std::string("ncclCommInitRank: ") + R->GetErrorString(r));
            delete c;
            return nullptr;
        }
    }
    if (finish_comm(c) != LLE_OK) { lle_comm_free(c); return nullptr; }
    lle::capi_ok();
    return c;
}

int lle_comm_create_all(lle_comm** out, int n_devices, const int* device_ids) {
    if (!out) return lle::capi_fail(LLE_ERR_NULL, "NULL out");
    if (n_devices < 1 || n_devices > 64) return lle::capi_fail(LLE_ERR_ARG, "n_devices out of range");
    Rccl* R = rccl();
    if (!R) return no_rccl();
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev <= 0) return lle::capi_fail(LLE_ERR_NO_DEVICE, "no HIP device");
    int devs[64];
    for (int k = 0; k < n_devices; k++) {
        devs[k] = device_ids ? device_ids[k] : k;
        if (devs[k] < 0 || devs[k] >= n_dev) return lle::capi_fail(LLE_ERR_ARG, "device id out of range");
        for (int q = 0; q < k; q++)
            if (devs[q] == devs[k]) return lle::capi_fail(LLE_ERR_ARG, "a device may appear once in a communicator");
    }
    ncclComm_t comms[64];
    {
        DeviceScopeC keep(devs[0]);  // (ncclCommInitAll walks the devices itself; put the caller's back afterwards)
        RCCL_TRY(R, R->CommInitAll(comms, n_devices, devs));
    }
    for (int k = 0; k < n_devices; k++) out[k] = nullptr;
    // On any failure nothing is left behind: the handles built so far are freed (each destroys its communicator), the
    // communicators that never got a handle are destroyed here, and out[] is all NULL again.
    auto undo = [&](int built, int rc) {
        for (int q = 0; q < built; q++) { lle_comm_free(out[q]); out[q] = nullptr; }
        for (int q = built; q < n_devices; q++) {
            DeviceScopeC scope(devs[q]);
            (void)R->CommDestroy(comms[q]);
        }
        return rc;
    };
    for (int k = 0; k < n_devices; k++) {
        lle_comm* c = new (std::nothrow) lle_comm();
        if (!c) return undo(k, lle::capi_fail(LLE_ERR_HIP, "out of memory"));
        c->comm = comms[k]; c->n_ranks = n_devices; c->rank = k; c->device = devs[k];
        out[k] = c;
        int rc = finish_comm(c);
        if (rc != LLE_OK) return undo(k + 1, rc);
    }
    return lle::capi_ok();
}

void lle_comm_free(lle_comm* c) {
    if (!c) return;
    DeviceScopeC scope(c->device);
    if (c->scratch) (void)hipFree(c->scratch);
    Rccl* R = rccl();
    if (R && c->comm) (void)R->CommDestroy(c->comm);
    delete c;
}

int lle_comm_rank(const lle_comm* c, int* rank, int* n_ranks) {
    if (!c) return lle::capi_fail(LLE_ERR_NULL, "NULL comm");
    if (rank) *rank = c->rank;
    if (n_ranks) *n_ranks = c->n_ranks;
    return lle::capi_ok();
}

int lle_comm_allreduce_i64(lle_comm* c, int64_t* buf_dev, int count, int op, void* stream) {
    if (!c || !buf_dev) return lle::capi_fail(LLE_ERR_NULL, "NULL argument");
    if (count < 1 || (op != LLE_COMM_SUM && op != LLE_COMM_MAX)) return lle::capi_fail(LLE_ERR_ARG, "count / op out of range");
    Rccl* R = rccl();
    if (!R) return no_rccl();
    DeviceScopeC scope(c->device);
    RCCL_TRY(R, R->AllReduce(buf_dev, buf_dev, (size_t)count, NCCL_INT64, op == LLE_COMM_SUM ? NCCL_SUM : NCCL_MAX, c->comm, (hipStream_t)stream));
    return lle::capi_ok();
}

int lle_comm_allreduce_i64_group(lle_comm* const* comms, int64_t* const* bufs_dev, void* const* streams, int n, int count, int op) {
    if (!comms || !bufs_dev) return lle::capi_fail(LLE_ERR_NULL, "NULL argument");
    if (n < 1 || n > 64 || count < 1 || (op != LLE_COMM_SUM && op != LLE_COMM_MAX)) return lle::capi_fail(LLE_ERR_ARG, "n / count / op out of range");
    Rccl* R = rccl();
    if (!R) return no_rccl();
    for (int k = 0; k < n; k++) {
        if (!comms[k] || !bufs_dev[k]) return lle::capi_fail(LLE_ERR_NULL, "NULL handle");
        if (comms[k]->n_ranks != n) return lle::capi_fail(LLE_ERR_ARG, "the group must hold every rank of the communicator");
    }
    RCCL_TRY(R, R->GroupStart());  // one process drives every rank: all calls inside one group, or the first would block
    for (int k = 0; k < n; k++) {
        DeviceScopeC scope(comms[k]->device);
        ncclResult_t r = R->AllReduce(bufs_dev[k], bufs_dev[k], (size_t)count, NCCL_INT64, op == LLE_COMM_SUM ? NCCL_SUM : NCCL_MAX, comms[k]->comm,
                                      (hipStream_t)(streams ? streams[k] : nullptr));
        if (r != 0) { (void)R->GroupEnd(); return lle::capi_fail(LLE_ERR_HIP, std::string("ncclAllReduce: ") + R->GetErrorString(r)); }
    }
    RCCL_TRY(R, R->GroupEnd());
    return lle::capi_ok();
}

int lle_batch_stats_allreduce(lle_batch* b, lle_comm* c, int64_t out[8], int reset_counters, void* stream) {
    if (!b || !c || !out) return lle::capi_fail(LLE_ERR_NULL, "NULL argument");
    if (lle::capi_batch_device(b) != c->device) return lle::capi_fail(LLE_ERR_ARG, "batch and communicator live on different devices");
    Rccl* R = rccl();
    if (!R) return no_rccl();
    DeviceScopeC scope(c->device);
    hipStream_t st = (hipStream_t)stream;
    int rc = lle::capi_batch_stats_to_device(b, c->scratch, 0, stream);  // this rank's eight sums, on the device
    if (rc != LLE_OK) return rc;
    RCCL_TRY(R, R->AllReduce(c->scratch, c->scratch, 8, NCCL_INT64, NCCL_SUM, c->comm, st));
    HIP_TRY_C(hipMemcpyAsync(out, c->scratch, 8 * sizeof(int64_t), hipMemcpyDeviceToHost, st));
    HIP_TRY_C(hipStreamSynchronize(st));
    // the counters go back to zero only once the reduced values are in the caller's hands: a failed collective loses nothing
    if (reset_counters && (rc = lle::capi_batch_reset_counters(b, stream)) != LLE_OK) return rc;
    return lle::capi_ok();
}

int lle_batch_stats_allreduce_group(lle_batch* const* batches, lle_comm* const* comms, void* const* streams, int n, int64_t out[8],
                                    int reset_counters) {
    if (!batches || !comms || !out) return lle::capi_fail(LLE_ERR_NULL, "NULL argument");
    if (n < 1 || n > 64) return lle::capi_fail(LLE_ERR_ARG, "n out of range");
    Rccl* R = rccl();
    if (!R) return no_rccl();
    for (int k = 0; k < n; k++) {
        if (!batches[k] || !comms[k]) return lle::capi_fail(LLE_ERR_NULL, "NULL handle");
        if (lle::capi_batch_device(batches[k]) != comms[k]->device) return lle::capi_fail(LLE_ERR_ARG, "batch and communicator live on different devices");
        if (comms[k]->n_ranks != n) return lle::capi_fail(LLE_ERR_ARG, "the group must hold every rank of the communicator");
    }
    for (int k = 0; k < n; k++) {
        int rc = lle::capi_batch_stats_to_device(batches[k], comms[k]->scratch, 0, streams ? streams[k] : nullptr);
        if (rc != LLE_OK) return rc;
    }
    // one process drives every rank: the calls of all ranks must be posted inside one group or the first would block
    RCCL_TRY(R, R->GroupStart());
    for (int k = 0; k < n; k++) {
        DeviceScopeC scope(comms[k]->device);
        ncclResult_t r = R->AllReduce(comms[k]->scratch, comms[k]->scratch, 8, NCCL_INT64, NCCL_SUM, comms[k]->comm,
                                      (hipStream_t)(streams ? streams[k] : nullptr));
        if (r != 0) { (void)R->GroupEnd(); return lle::capi_fail(LLE_ERR_HIP, std::string("ncclAllReduce: ") + R->GetErrorString(r)); }
    }
    RCCL_TRY(R, R->GroupEnd());
    for (int k = 0; k < n; k++) {
        DeviceScopeC scope(comms[k]->device);
        hipStream_t st = (hipStream_t)(streams ? streams[k] : nullptr);
        int64_t got[8];
        HIP_TRY_C(hipMemcpyAsync(got, comms[k]->scratch, sizeof got, hipMemcpyDeviceToHost, st));
        HIP_TRY_C(hipStreamSynchronize(st));
        if (k == 0) std::memcpy(out, got, sizeof got);
        else if (std::memcmp(out, got, sizeof got) != 0) return lle::capi_fail(LLE_ERR_HIP, "ranks disagree on the reduced counters");
    }
    if (reset_counters)  // (after the reduction succeeded on every rank)
        for (int k = 0; k < n; k++) {
            int rc = lle::capi_batch_reset_counters(batches[k], streams ? streams[k] : nullptr);
            if (rc != LLE_OK) return rc;
        }
    return lle::capi_ok();
}

}  // extern "C"
