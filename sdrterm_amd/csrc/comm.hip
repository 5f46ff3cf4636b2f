// Multi-GPU --simo: distribution of the raw IQ block between the per-GPU processes (include/sdrterm_hip.h, sdrt_comm_*).
// RCCL is the transport (xGMI between the GPUs of one node); librccl is loaded on first use, so a single-GPU host never
// needs it.  One process per GPU, one communicator per process; the host side exchanges the 128-byte unique id.
#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>

#include "../../include/sdrterm_hip.h"

int sdrt_set_error(int code, const char *fmt, ...);      // api.hip: fills the thread-local message

namespace {

// the handful of RCCL entry points used, with the types of rccl.h restated so that the header is not needed to build
typedef struct ncclComm *ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef int ncclResult_t;
enum { kNcclUint8 = 1 };

struct Rccl {
    void *handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*Broadcast)(const void *, void *, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, int, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    // optional (the scatter + all-gather form of the block broadcast)
    ncclResult_t (*Send)(const void *, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void *, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
};

Rccl g_rccl;

int load_rccl() {
    if (g_rccl.handle) return SDRT_OK;
    const char *names[] = {getenv("SDRTERM_RCCL_LIB"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void *h = nullptr;
    for (const char *n : names) {
        if (!n || !*n) continue;
        h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
        if (h) break;
    }
    if (!h) return sdrt_set_error(SDRT_EINVAL, "cannot load librccl (set SDRTERM_RCCL_LIB): %s", dlerror());
    Rccl r;
    r.handle = h;
    r.GetUniqueId = (decltype(r.GetUniqueId)) dlsym(h, "ncclGetUniqueId");
    r.CommInitRank = (decltype(r.CommInitRank)) dlsym(h, "ncclCommInitRank");
    r.CommDestroy = (decltype(r.CommDestroy)) dlsym(h, "ncclCommDestroy");
    r.Broadcast = (decltype(r.Broadcast)) dlsym(h, "ncclBroadcast");
    r.AllGather = (decltype(r.AllGather)) dlsym(h, "ncclAllGather");
    r.GetErrorString = (decltype(r.GetErrorString)) dlsym(h, "ncclGetErrorString");
    r.Send = (decltype(r.Send)) dlsym(h, "ncclSend");
    r.Recv = (decltype(r.Recv)) dlsym(h, "ncclRecv");
    r.GroupStart = (decltype(r.GroupStart)) dlsym(h, "ncclGroupStart");
    r.GroupEnd = (decltype(r.GroupEnd)) dlsym(h, "ncclGroupEnd");
    if (!r.GetUniqueId || !r.CommInitRank || !r.CommDestroy || !r.Broadcast || !r.AllGather || !r.GetErrorString) {
        dlclose(h);
        return sdrt_set_error(SDRT_EINVAL, "librccl lacks an expected entry point");
    }
    g_rccl = r;
    return SDRT_OK;
}

int rccl_fail(const char *what, ncclResult_t e) {
    return sdrt_set_error(SDRT_EHIP, "%s failed: %s", what, g_rccl.GetErrorString ? g_rccl.GetErrorString(e) : "?");
}

}  // namespace

struct sdrt_comm {
    ncclComm_t comm = nullptr;
    int world = 1, rank = 0, device = 0;
};

extern "C" {

int sdrt_comm_unique_id(void *id128) {
    if (!id128) return sdrt_set_error(SDRT_EINVAL, "null argument");
    int rc = load_rccl();
    if (rc != SDRT_OK) return rc;
    ncclUniqueId id;
    const ncclResult_t e = g_rccl.GetUniqueId(&id);
    if (e != 0) return rccl_fail("ncclGetUniqueId", e);
    memcpy(id128, id.internal, sizeof(id.internal));
    return SDRT_OK;
}

int sdrt_comm_init(const void *id128, int world, int rank, int device, sdrt_comm **out) {
    if (!id128 || !out || world < 1 || rank < 0 || rank >= world) return sdrt_set_error(SDRT_EINVAL, "bad argument");
    *out = nullptr;
    int rc = load_rccl();
    if (rc != SDRT_OK) return rc;
    hipError_t he = hipSetDevice(device);
    if (he != hipSuccess) return sdrt_set_error(SDRT_EHIP, "hipSetDevice(%d): %s", device, hipGetErrorString(he));
    sdrt_comm *c = new (std::nothrow) sdrt_comm();
    if (!c) return sdrt_set_error(SDRT_ENOMEM, "out of host memory");
    ncclUniqueId id;
    memcpy(id.internal, id128, sizeof(id.internal));
    const ncclResult_t e = g_rccl.CommInitRank(&c->comm, world, id, rank);
    if (e != 0) { delete c; return rccl_fail("ncclCommInitRank", e); }
    c->world = world;
    c->rank = rank;
    c->device = device;
    *out = c;
    return SDRT_OK;
}

int sdrt_comm_bcast(sdrt_comm *c, void *dev_buf, size_t bytes, int root, void *stream) {
    if (!c || (!dev_buf && bytes) || root < 0 || root >= c->world) return sdrt_set_error(SDRT_EINVAL, "bad argument");
    if (bytes == 0) return SDRT_OK;
    const ncclResult_t e = g_rccl.Broadcast(dev_buf, dev_buf, bytes, kNcclUint8, root, c->comm, (hipStream_t) stream);
    return e == 0 ? SDRT_OK : rccl_fail("ncclBroadcast", e);
}

// The same result as sdrt_comm_bcast, built for point-to-point links: the root sends every rank ITS 1/world of the block (all of
// the root's links at once, each carrying one shard), then the ranks all-gather the shards in place (every link of every rank
// carries one shard).  A ring broadcast moves the whole block over every hop instead.  Needs bytes % world == 0 (a chunked block
// always divides by the GPU counts of a node); otherwise, and with a single rank, this is sdrt_comm_bcast.
#define SDRT_SCATTER_MIN_BYTES ((size_t) 1 << 20)
int sdrt_comm_bcast_scatter(sdrt_comm *c, void *dev_buf, size_t bytes, int root, void *stream) {
    if (!c || (!dev_buf && bytes) || root < 0 || root >= c->world) return sdrt_set_error(SDRT_EINVAL, "bad argument");
    if (bytes == 0) return SDRT_OK;
    // small payloads (headers, a few chunks): two collectives' latency buys nothing over one; and a librccl without the
    // point-to-point entry points still has ncclBroadcast
    if (c->world == 1 || bytes % (size_t) c->world || bytes < SDRT_SCATTER_MIN_BYTES || !g_rccl.Send || !g_rccl.Recv || !g_rccl.GroupStart ||
        !g_rccl.GroupEnd)
        return sdrt_comm_bcast(c, dev_buf, bytes, root, stream);
    const size_t shard = bytes / (size_t) c->world;
    char *buf = (char *) dev_buf;
    ncclResult_t e = g_rccl.GroupStart();
    if (e != 0) return rccl_fail("ncclGroupStart", e);
    if (c->rank == root) {
        for (int r = 0; r < c->world && e == 0; ++r)
            if (r != root) e = g_rccl.Send(buf + (size_t) r * shard, shard, kNcclUint8, r, c->comm, (hipStream_t) stream);
    } else {
        e = g_rccl.Recv(buf + (size_t) c->rank * shard, shard, kNcclUint8, root, c->comm, (hipStream_t) stream);
    }
    const ncclResult_t e2 = g_rccl.GroupEnd();
    if (e != 0) return rccl_fail("ncclSend / ncclRecv", e);
    if (e2 != 0) return rccl_fail("ncclGroupEnd", e2);
    // in place: every rank's shard sits where the gathered block wants it
    e = g_rccl.AllGather(buf + (size_t) c->rank * shard, buf, shard, kNcclUint8, c->comm, (hipStream_t) stream);
    return e == 0 ? SDRT_OK : rccl_fail("ncclAllGather", e);
}

int sdrt_comm_allgather(sdrt_comm *c, const void *send_dev, void *recv_dev, size_t bytes_per_rank, void *stream) {
    if (!c || !send_dev || !recv_dev) return sdrt_set_error(SDRT_EINVAL, "bad argument");
    if (bytes_per_rank == 0) return SDRT_OK;
    const ncclResult_t e = g_rccl.AllGather(send_dev, recv_dev, bytes_per_rank, kNcclUint8, c->comm, (hipStream_t) stream);
    return e == 0 ? SDRT_OK : rccl_fail("ncclAllGather", e);
}

int sdrt_comm_world(const sdrt_comm *c) { return c ? c->world : -1; }
int sdrt_comm_rank(const sdrt_comm *c) { return c ? c->rank : -1; }

void sdrt_comm_destroy(sdrt_comm *c) {
    if (!c) return;
    if (c->comm && g_rccl.CommDestroy) (void) g_rccl.CommDestroy(c->comm);
    delete c;
}

}  // extern "C"
