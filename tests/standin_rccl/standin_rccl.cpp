// TEST INFRASTRUCTURE, not part of the product: a stand-in for the RCCL entry points sdrterm_amd/csrc/comm.hip binds
// (ncclGetUniqueId, ncclCommInitRank, ncclCommDestroy, ncclBroadcast, ncclAllGather, ncclGetErrorString; ncclSend, ncclRecv,
// ncclGroupStart, ncclGroupEnd for the scatter form of the block broadcast), for ranks that SHARE
// one GPU.  RCCL refuses two ranks on one device ("Duplicate GPU detected"), and the boxes this build is developed on have one:
// with SDRTERM_RCCL_LIB pointing here, tests/test_gpu_ranks.py runs bench.py and VfoProcessor with two and three rank
// processes on that one GPU -- every line of the rank plumbing (unique-id exchange, communicator set-up, block exchange on the
// communication stream, barrier / maximum over ranks, the order of collectives on every rank) with the real kernels.  What it
// does not test is RCCL itself: tests/test_gpu_processors.py does that with a world of one.
//
// Data moves through POSIX shared memory with blocking copies (the calling stream is drained first, so the stream order the
// callers rely on holds); collectives must be entered by every rank in the same order, as with the real library; a rank that
// waits longer than SDRT_STANDIN_TIMEOUT seconds (default 120) gives up with an error instead of hanging.
#include <fcntl.h>
#include <hip/hip_runtime.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>

namespace {
struct Header {
    std::atomic<unsigned> count, generation;
    std::atomic<int> failed;
    std::atomic<unsigned long long> p2p_bytes;      // size of the point-to-point message being moved (0: none for this pair)
};
constexpr size_t HEADER = 4096;
}  // namespace

struct ncclComm {
    Header *h = nullptr;
    unsigned char *stage = nullptr;
    size_t stage_bytes = 0, map_bytes = 0;
    int world = 1, rank = 0;
    double timeout = 120.0;
};
typedef ncclComm *ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef int ncclResult_t;

namespace {
const char *g_msg = "no error";
ncclResult_t fail(const char *m) { g_msg = m; return 2; }

ncclResult_t barrier(ncclComm *c) {
    if (c->world == 1) return 0;
    const unsigned g = c->h->generation.load();
    if (c->h->count.fetch_add(1) + 1 == (unsigned) c->world) {
        c->h->count.store(0);
        c->h->generation.store(g + 1);
        return 0;
    }
    const auto t0 = std::chrono::steady_clock::now();
    while (c->h->generation.load() == g) {
        if (c->h->failed.load()) return fail("stand-in transport: another rank gave up");
        if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > c->timeout) {
            c->h->failed.store(1);
            return fail("stand-in transport: timed out waiting for the other ranks");
        }
        usleep(20);
    }
    return 0;
}
}  // namespace

extern "C" {

ncclResult_t ncclGetUniqueId(ncclUniqueId *id) {
    std::random_device rd;
    memset(id->internal, 0, sizeof(id->internal));
    snprintf(id->internal, sizeof(id->internal), "/sdrt_standin_%d_%08x%08x", (int) getpid(), (unsigned) rd(), (unsigned) rd());
    return 0;
}

ncclResult_t ncclCommInitRank(ncclComm_t *out, int nranks, ncclUniqueId id, int rank) {
    if (id.internal[0] != '/') return fail("stand-in transport: not one of its unique ids");
    ncclComm *c = new ncclComm();
    c->world = nranks;
    c->rank = rank;
    if (const char *e = getenv("SDRT_STANDIN_TIMEOUT")) c->timeout = atof(e);
    const char *mb = getenv("SDRT_STANDIN_STAGE_MB");
    c->stage_bytes = (size_t) (mb ? atoi(mb) : 64) << 20;
    c->map_bytes = HEADER + c->stage_bytes;
    const int fd = shm_open(id.internal, O_CREAT | O_RDWR, 0600);
    if (fd < 0) { delete c; return fail("stand-in transport: shm_open failed"); }
    if (ftruncate(fd, (off_t) c->map_bytes) != 0) { close(fd); delete c; return fail("stand-in transport: ftruncate failed"); }
    void *p = mmap(nullptr, c->map_bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (p == MAP_FAILED) { delete c; return fail("stand-in transport: mmap failed"); }
    c->h = (Header *) p;                      // (a fresh segment is zero-filled: counters start at 0 on whichever rank comes first)
    c->stage = (unsigned char *) p + HEADER;
    const ncclResult_t e = barrier(c);        // everybody has mapped it ...
    if (rank == 0) shm_unlink(id.internal);   // ... so the name can go: nothing is left in /dev/shm whatever happens later
    if (e) { munmap(p, c->map_bytes); delete c; return e; }
    *out = c;
    return 0;
}

ncclResult_t ncclCommDestroy(ncclComm_t c) {
    if (!c) return 0;
    munmap((void *) c->h, c->map_bytes);
    delete c;
    return 0;
}

// datatype 1 = ncclUint8 is the only one comm.hip uses: counts are bytes
ncclResult_t ncclBroadcast(const void *send, void *recv, size_t count, int datatype, int root, ncclComm_t c, hipStream_t stream) {
    if (datatype != 1) return fail("stand-in transport: bytes only");
    if (hipStreamSynchronize(stream) != hipSuccess) return fail("stand-in transport: hipStreamSynchronize failed");
    for (size_t off = 0; off < count; off += c->stage_bytes) {
        const size_t len = count - off < c->stage_bytes ? count - off : c->stage_bytes;
        if (c->rank == root && hipMemcpy(c->stage, (const char *) send + off, len, hipMemcpyDeviceToHost) != hipSuccess) return fail("stand-in transport: copy failed");
        if (ncclResult_t e = barrier(c)) return e;
        if (c->rank != root) {
            if (hipMemcpy((char *) recv + off, c->stage, len, hipMemcpyHostToDevice) != hipSuccess) return fail("stand-in transport: copy failed");
        } else if (recv != send) {
            if (hipMemcpy((char *) recv + off, (const char *) send + off, len, hipMemcpyDeviceToDevice) != hipSuccess) return fail("stand-in transport: copy failed");
        }
        if (ncclResult_t e = barrier(c)) return e;
    }
    return 0;
}

ncclResult_t ncclAllGather(const void *send, void *recv, size_t count, int datatype, ncclComm_t c, hipStream_t stream) {
    if (datatype != 1) return fail("stand-in transport: bytes only");
    if (hipStreamSynchronize(stream) != hipSuccess) return fail("stand-in transport: hipStreamSynchronize failed");
    for (int r = 0; r < c->world; ++r)
        for (size_t off = 0; off < count; off += c->stage_bytes) {
            const size_t len = count - off < c->stage_bytes ? count - off : c->stage_bytes;
            if (c->rank == r && hipMemcpy(c->stage, (const char *) send + off, len, hipMemcpyDeviceToHost) != hipSuccess) return fail("stand-in transport: copy failed");
            if (ncclResult_t e = barrier(c)) return e;
            if (hipMemcpy((char *) recv + (size_t) r * count + off, c->stage, len, hipMemcpyHostToDevice) != hipSuccess) return fail("stand-in transport: copy failed");
            if (ncclResult_t e = barrier(c)) return e;
        }
    return 0;
}

// Point-to-point messages are collected between ncclGroupStart and ncclGroupEnd and moved at ncclGroupEnd: every rank walks over all
// ordered (sender, receiver) pairs in the same order, the sender announces the size (0: it has nothing for that pair), both sides
// copy through the staging area.  Every rank of the communicator must reach ncclGroupEnd (comm.hip's scatter does).
namespace {
struct P2p { bool send; void *buf; size_t bytes; int peer; ncclComm *comm; hipStream_t stream; };
thread_local P2p g_ops[64];
thread_local int g_nops = 0, g_depth = 0;

ncclResult_t run_group() {
    if (g_nops == 0) return 0;
    ncclComm *c = g_ops[0].comm;
    for (int i = 0; i < g_nops; ++i)
        if (hipStreamSynchronize(g_ops[i].stream) != hipSuccess) return fail("stand-in transport: hipStreamSynchronize failed");
    for (int src = 0; src < c->world; ++src)
        for (int dst = 0; dst < c->world; ++dst) {
            if (src == dst) continue;
            P2p *mine = nullptr;
            for (int i = 0; i < g_nops; ++i)
                if ((c->rank == src && g_ops[i].send && g_ops[i].peer == dst) || (c->rank == dst && !g_ops[i].send && g_ops[i].peer == src)) mine = &g_ops[i];
            if (c->rank == src) c->h->p2p_bytes.store(mine ? mine->bytes : 0);
            if (ncclResult_t e = barrier(c)) return e;
            const size_t total = (size_t) c->h->p2p_bytes.load();
            if (c->rank == dst && total && (!mine || mine->bytes != total)) { c->h->failed.store(1); return fail("stand-in transport: send without a matching receive"); }
            for (size_t off = 0; off < total; off += c->stage_bytes) {
                const size_t len = total - off < c->stage_bytes ? total - off : c->stage_bytes;
                if (c->rank == src && hipMemcpy(c->stage, (const char *) mine->buf + off, len, hipMemcpyDeviceToHost) != hipSuccess) return fail("stand-in transport: copy failed");
                if (ncclResult_t e = barrier(c)) return e;
                if (c->rank == dst && hipMemcpy((char *) mine->buf + off, c->stage, len, hipMemcpyHostToDevice) != hipSuccess) return fail("stand-in transport: copy failed");
                if (ncclResult_t e = barrier(c)) return e;
            }
            if (ncclResult_t e = barrier(c)) return e;      // (the size field is free for the next pair)
        }
    g_nops = 0;
    return 0;
}
}  // namespace

ncclResult_t ncclGroupStart() { ++g_depth; return 0; }
ncclResult_t ncclGroupEnd() {
    if (--g_depth > 0) return 0;
    g_depth = 0;
    const ncclResult_t e = run_group();
    g_nops = 0;
    return e;
}
ncclResult_t ncclSend(const void *buf, size_t count, int datatype, int peer, ncclComm_t c, hipStream_t stream) {
    if (datatype != 1 || g_depth == 0 || g_nops == 64) return fail("stand-in transport: bytes, inside a group, at most 64 messages");
    g_ops[g_nops++] = {true, (void *) buf, count, peer, c, stream};
    return 0;
}
ncclResult_t ncclRecv(void *buf, size_t count, int datatype, int peer, ncclComm_t c, hipStream_t stream) {
    if (datatype != 1 || g_depth == 0 || g_nops == 64) return fail("stand-in transport: bytes, inside a group, at most 64 messages");
    g_ops[g_nops++] = {false, buf, count, peer, c, stream};
    return 0;
}

const char *ncclGetErrorString(ncclResult_t) { return g_msg; }

}  // extern "C"
