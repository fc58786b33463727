// comm.hip — the one exchange of the window-sharded sequence driver, on RCCL itself.
//
// Windows of a sequence are independent (test_kitti_pose.py:134-145), so ranks (one process per
// GPU) run contiguous window ranges with no data-path collective; before the sequential 4x4 chain
// (test_kitti_pose.py:147-149) every rank needs all [n,2,6] float32 poses: one ncclAllGather of
// equal padded counts (48 B per window: 218 KB for KITTI seq 00 — latency only, over xGMI).
// The same communicator carries the bench's barrier and max-over-ranks (one-element all-reduces).
//
// librccl.so (573 MB) is opened with dlopen at the first davo_comm_* call, so single-GPU users of
// libdavo_hip.so never map it; types and enums come from <rccl/rccl.h> at compile time.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <cstdlib>
#include <cstring>
#include <mutex>

#include "ctx.h"

namespace davo {

struct Rccl {
    void* handle = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    std::string err;
};

// process-wide, loaded once; nullptr + message on failure
static Rccl* rccl(std::string* why) {
    static std::mutex mu;
    static Rccl R;
    static bool tried = false;
    std::lock_guard<std::mutex> lock(mu);
    if (!tried) {
        tried = true;
        // the loader's own search order (the library's rpath is /opt/rocm/lib, then LD_LIBRARY_PATH / ld.so.conf), then the
        // image's fixed location; the library itself reads no environment variable
        std::string names[2] = {"librccl.so.1", "/opt/rocm/lib/librccl.so.1"};
        for (const auto& n : names) {
            if (n.empty()) continue;
            R.handle = dlopen(n.c_str(), RTLD_NOW | RTLD_LOCAL);
            if (R.handle) break;
            R.err = dlerror();
        }
        if (R.handle) {
            bool ok = true;
            auto sym = [&](const char* name) { void* p = dlsym(R.handle, name); if (!p) { ok = false; R.err = std::string("librccl lacks ") + name; } return p; };
            R.GetUniqueId = reinterpret_cast<decltype(R.GetUniqueId)>(sym("ncclGetUniqueId"));
            R.CommInitRank = reinterpret_cast<decltype(R.CommInitRank)>(sym("ncclCommInitRank"));
            R.CommDestroy = reinterpret_cast<decltype(R.CommDestroy)>(sym("ncclCommDestroy"));
            R.AllGather = reinterpret_cast<decltype(R.AllGather)>(sym("ncclAllGather"));
            R.AllReduce = reinterpret_cast<decltype(R.AllReduce)>(sym("ncclAllReduce"));
            R.GetErrorString = reinterpret_cast<decltype(R.GetErrorString)>(sym("ncclGetErrorString"));
            if (!ok) { dlclose(R.handle); R.handle = nullptr; }
        }
    }
    if (!R.handle) { if (why) *why = R.err; return nullptr; }
    return &R;
}

struct Comm {
    ncclComm_t comm = nullptr;
    int nranks = 0, rank = 0;
    hipStream_t stream = nullptr;
    void *d_send = nullptr, *d_recv = nullptr;     // staging for the host-buffer entry points
    size_t send_bytes = 0, recv_bytes = 0;
    hipEvent_t e0 = nullptr, e1 = nullptr;
};

#define NCCL_TRY(c, R, expr)                                                                          \
    do {                                                                                              \
        ncclResult_t r_ = (expr);                                                                     \
        if (r_ != ncclSuccess)                                                                        \
            return fail(c, DAVO_ERR_COMM, "%s failed: %s", #expr, (R)->GetErrorString(r_));           \
    } while (0)

static int need_comm(davo_ctx* c, Rccl** R) {
    if (!c) return DAVO_ERR_INVALID;
    if (!c->comm || !c->comm->comm) return fail(c, DAVO_ERR_NOT_READY, "no communicator: call davo_comm_init first");
    std::string why;
    *R = rccl(&why);
    if (!*R) return fail(c, DAVO_ERR_COMM, "librccl could not be loaded: %s", why.c_str());
    HIP_TRY(c, hipSetDevice(c->device));
    return DAVO_OK;
}

static int grow(davo_ctx* c, void** p, size_t* have, size_t want) {
    if (*have >= want) return DAVO_OK;
    if (*p) { HIP_TRY(c, hipFree(*p)); *p = nullptr; *have = 0; }
    HIP_TRY(c, hipMalloc(p, want));
    *have = want;
    return DAVO_OK;
}

void comm_release(davo_ctx* c) {
    if (!c || !c->comm) return;
    Comm* m = c->comm;
    std::string why;
    Rccl* R = rccl(&why);
    if (m->stream) (void)hipStreamSynchronize(m->stream);
    if (m->comm && R) (void)R->CommDestroy(m->comm);
    if (m->d_send) (void)hipFree(m->d_send);
    if (m->d_recv) (void)hipFree(m->d_recv);
    if (m->e0) (void)hipEventDestroy(m->e0);
    if (m->e1) (void)hipEventDestroy(m->e1);
    if (m->stream) (void)hipStreamDestroy(m->stream);
    delete m;
    c->comm = nullptr;
}

}  // namespace davo

using namespace davo;

extern "C" {

int davo_comm_preload(char* err, int err_len) {
    std::string why;
    if (rccl(&why)) return DAVO_OK;
    if (err && err_len > 0) { strncpy(err, ("librccl could not be loaded: " + why).c_str(), err_len - 1); err[err_len - 1] = 0; }
    return DAVO_ERR_COMM;
}

int davo_comm_unique_id(void* id_out, char* err, int err_len) {
    auto bad = [&](const std::string& m, int code) {
        if (err && err_len > 0) { strncpy(err, m.c_str(), err_len - 1); err[err_len - 1] = 0; }
        return code;
    };
    if (!id_out) return bad("null id buffer", DAVO_ERR_INVALID);
    static_assert(sizeof(ncclUniqueId) == DAVO_COMM_ID_BYTES, "ncclUniqueId size is part of the ABI");
    std::string why;
    Rccl* R = rccl(&why);
    if (!R) return bad("librccl could not be loaded: " + why, DAVO_ERR_COMM);
    ncclUniqueId id;
    const ncclResult_t r = R->GetUniqueId(&id);
    if (r != ncclSuccess) return bad(std::string("ncclGetUniqueId failed: ") + R->GetErrorString(r), DAVO_ERR_COMM);
    memcpy(id_out, &id, sizeof id);
    return DAVO_OK;
}

int davo_comm_init(davo_ctx* c, int nranks, int rank, const void* id) {
    if (!c) return DAVO_ERR_INVALID;
    if (!id || nranks < 1 || rank < 0 || rank >= nranks) return fail(c, DAVO_ERR_INVALID, "davo_comm_init: rank %d of %d", rank, nranks);
    if (c->comm) return fail(c, DAVO_ERR_INVALID, "the context already has a communicator (davo_comm_destroy first)");
    std::string why;
    Rccl* R = rccl(&why);
    if (!R) return fail(c, DAVO_ERR_COMM, "librccl could not be loaded: %s", why.c_str());
    // Built aside and published last: this call takes seconds (library load, bootstrap, topology) and a host may run it on a
    // second thread while the context's owner thread issues forwards (davo_hip.h) - until the last line it touches the context
    // only to report a failure.
    if (hipSetDevice(c->device) != hipSuccess) return fail(c, DAVO_ERR_HIP, "hipSetDevice(%d) failed", c->device);
    Comm* m = new Comm();
    m->nranks = nranks; m->rank = rank;
    ncclUniqueId uid;
    memcpy(&uid, id, sizeof uid);
    const ncclResult_t r = R->CommInitRank(&m->comm, nranks, uid, rank);
    if (r != ncclSuccess) {
        delete m;
        return fail(c, DAVO_ERR_COMM, "ncclCommInitRank(rank %d of %d, device %d) failed: %s", rank, nranks, c->device, R->GetErrorString(r));
    }
    if (hipStreamCreateWithFlags(&m->stream, hipStreamNonBlocking) != hipSuccess || hipEventCreate(&m->e0) != hipSuccess ||
        hipEventCreate(&m->e1) != hipSuccess) {
        (void)R->CommDestroy(m->comm);
        if (m->stream) (void)hipStreamDestroy(m->stream);
        if (m->e0) (void)hipEventDestroy(m->e0);
        delete m;
        return fail(c, DAVO_ERR_HIP, "creating the communicator's stream and events failed");
    }
    c->comm = m;
    return DAVO_OK;
}

int davo_comm_destroy(davo_ctx* c) {
    if (!c) return DAVO_ERR_INVALID;
    comm_release(c);
    return DAVO_OK;
}

int davo_comm_size(davo_ctx* c, int* nranks, int* rank) {
    if (!c) return DAVO_ERR_INVALID;
    if (!c->comm) return fail(c, DAVO_ERR_NOT_READY, "no communicator: call davo_comm_init first");
    if (nranks) *nranks = c->comm->nranks;
    if (rank) *rank = c->comm->rank;
    return DAVO_OK;
}

int davo_allgather_poses_device(davo_ctx* c, const void* d_local, int n_per_rank, void* d_all, float* elapsed_ms) {
    Rccl* R = nullptr;
    { int rc = need_comm(c, &R); if (rc) return rc; }
    if (!d_local || !d_all || n_per_rank < 1) return fail(c, DAVO_ERR_INVALID, "davo_allgather_poses_device: bad argument");
    Comm* m = c->comm;
    { int rc = sync_all_slots(c); if (rc) return rc; }                // the poses of the shard are complete
    if (elapsed_ms) HIP_TRY(c, hipEventRecord(m->e0, m->stream));
    NCCL_TRY(c, R, R->AllGather(d_local, d_all, (size_t)n_per_rank * 12, ncclFloat32, m->comm, m->stream));
    if (elapsed_ms) HIP_TRY(c, hipEventRecord(m->e1, m->stream));
    HIP_TRY(c, hipStreamSynchronize(m->stream));
    if (elapsed_ms) HIP_TRY(c, hipEventElapsedTime(elapsed_ms, m->e0, m->e1));
    return DAVO_OK;
}

int davo_allgather_poses(davo_ctx* c, const float* local, int n_local, int n_per_rank, float* all, float* elapsed_ms) {
    Rccl* R = nullptr;
    { int rc = need_comm(c, &R); if (rc) return rc; }
    if (!all || n_per_rank < 1 || n_local < 0 || n_local > n_per_rank || (n_local > 0 && !local))
        return fail(c, DAVO_ERR_INVALID, "davo_allgather_poses: %d local windows, %d per rank", n_local, n_per_rank);
    Comm* m = c->comm;
    const size_t per = (size_t)n_per_rank * 12 * sizeof(float);
    { int rc = grow(c, &m->d_send, &m->send_bytes, per); if (rc) return rc; }
    { int rc = grow(c, &m->d_recv, &m->recv_bytes, per * m->nranks); if (rc) return rc; }
    HIP_TRY(c, hipMemsetAsync(m->d_send, 0, per, m->stream));         // ranks with a short last shard pad with zeros
    if (n_local) HIP_TRY(c, hipMemcpyAsync(m->d_send, local, (size_t)n_local * 12 * sizeof(float), hipMemcpyHostToDevice, m->stream));
    if (elapsed_ms) HIP_TRY(c, hipEventRecord(m->e0, m->stream));
    NCCL_TRY(c, R, R->AllGather(m->d_send, m->d_recv, (size_t)n_per_rank * 12, ncclFloat32, m->comm, m->stream));
    if (elapsed_ms) HIP_TRY(c, hipEventRecord(m->e1, m->stream));
    HIP_TRY(c, hipMemcpyAsync(all, m->d_recv, per * m->nranks, hipMemcpyDeviceToHost, m->stream));
    HIP_TRY(c, hipStreamSynchronize(m->stream));
    if (elapsed_ms) HIP_TRY(c, hipEventElapsedTime(elapsed_ms, m->e0, m->e1));
    return DAVO_OK;
}

// one-element all-reduce: op 0 = sum, 1 = max, 2 = min.  `value` is in/out (host).
int davo_comm_allreduce(davo_ctx* c, double* value, int op) {
    Rccl* R = nullptr;
    { int rc = need_comm(c, &R); if (rc) return rc; }
    if (!value || op < 0 || op > 2) return fail(c, DAVO_ERR_INVALID, "davo_comm_allreduce: bad argument");
    Comm* m = c->comm;
    { int rc = grow(c, &m->d_send, &m->send_bytes, 64); if (rc) return rc; }
    { int rc = grow(c, &m->d_recv, &m->recv_bytes, 64); if (rc) return rc; }
    HIP_TRY(c, hipMemcpyAsync(m->d_send, value, sizeof(double), hipMemcpyHostToDevice, m->stream));
    const ncclRedOp_t ops[3] = {ncclSum, ncclMax, ncclMin};
    NCCL_TRY(c, R, R->AllReduce(m->d_send, m->d_recv, 1, ncclFloat64, ops[op], m->comm, m->stream));
    HIP_TRY(c, hipMemcpyAsync(value, m->d_recv, sizeof(double), hipMemcpyDeviceToHost, m->stream));
    HIP_TRY(c, hipStreamSynchronize(m->stream));
    return DAVO_OK;
}

// every stream of this context idle, then a rendezvous of all ranks
int davo_comm_barrier(davo_ctx* c) {
    if (!c) return DAVO_ERR_INVALID;
    { int rc = sync_all_slots(c); if (rc) return rc; }
    double one = 1.0;
    const int rc = davo_comm_allreduce(c, &one, 0);
    if (rc) return rc;
    if (c->comm && (int)(one + 0.5) != c->comm->nranks)
        return fail(c, DAVO_ERR_COMM, "barrier all-reduce returned %g for %d ranks", one, c->comm->nranks);
    return DAVO_OK;
}

}  // extern "C"
