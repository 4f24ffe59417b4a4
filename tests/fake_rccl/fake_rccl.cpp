// fake_rccl.cpp -- TESTS-ONLY stand-in for librccl, so that the MULTI-PROCESS path of the library (gphip_create_rank:
// one process per rank, collectives between processes) can run with world_size 2 on a box that has ONE GPU, where real
// RCCL refuses two ranks on the same device.  libgphip binds RCCL with dlopen (csrc/rccl_dyn.h) and honours
// $GPHIP_RCCL_PATH, which the test points here.  Same nine entry points, same call semantics as the library uses them
// (in-place broadcast of bytes on a stream, sum all-reduce of doubles, grouped calls); the transport is a POSIX
// shared-memory segment + a process-shared barrier, staged through the host:
//     root:   stream-ordered D2H copy of its buffer into the segment, wait, barrier
//     others: barrier, stream-ordered H2D copy out of the segment, wait;  barrier again before the segment is reused
// Blocking the host inside a collective is slower than RCCL's asynchronous kernels but orders the data the same way.
// Segment name: $FAKE_RCCL_SHM (every rank of a job gets the same value) + the first four bytes of the unique id.
// In-process mode (ncclCommInitAll: ONE process, several "devices" -- here the same GPU listed several times, which the
// library accepts for RCCL only under GPHIP_COMM=rccl): calls between ncclGroupStart and ncclGroupEnd are recorded and
// executed together at ncclGroupEnd -- the root's stream is drained, then every other rank's buffer is filled by a
// device-to-device copy on ITS stream.  That exercises the library's grouped single-process broadcast path.
#include <fcntl.h>
#include <hip/hip_runtime.h>
#include <pthread.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>

namespace {
constexpr size_t CAP = 96u << 20;
struct Segment {
    std::atomic<int> ready;
    pthread_barrier_t bar;
    double red[8 * 64];
    char data[CAP];
};
struct Comm { int rank, nranks; Segment* seg; char name[160]; };
// kind: 0 broadcast (in-process mode only), 1 send (peer = root field), 2 recv (peer = root field), 3 all-gather (in-process mode)
struct Op { const void* send; void* recv; size_t bytes; int root; Comm* comm; hipStream_t st; int kind; };
int g_depth = 0;
Op g_ops[64];
int g_nops = 0;
size_t type_size(int t) { return (t == 0 || t == 1) ? 1 : (t == 7 ? 4 : 8); }     // ncclChar/Uint8, Float32, Float64
}  // namespace

struct FakeId { char internal[128]; };

extern "C" {
int ncclGetUniqueId(FakeId* id) { memset(id, 0, sizeof *id); memcpy(id->internal, "fake-rccl", 9); return 0; }

int ncclCommInitRank(void** comm, int nranks, FakeId id, int rank) {
    const char* base = getenv("FAKE_RCCL_SHM");
    if (!base) return 5;
    // one segment per communicator: the job's name + the first bytes of the unique id (a job that makes a second communicator
    // -- tests/multiproc_create_worker.py -- must not meet the first one's segment)
    char name[160];
    snprintf(name, sizeof name, "%.120s_%02x%02x%02x%02x", base, (unsigned char)id.internal[0], (unsigned char)id.internal[1],
             (unsigned char)id.internal[2], (unsigned char)id.internal[3]);
    int fd = -1;
    if (rank == 0) {
        shm_unlink(name);
        fd = shm_open(name, O_CREAT | O_RDWR, 0600);
        if (fd < 0 || ftruncate(fd, sizeof(Segment)) != 0) return 2;
    } else {
        for (int i = 0; i < 20000 && fd < 0; ++i) { fd = shm_open(name, O_RDWR, 0600); if (fd < 0) usleep(1000); }
        if (fd < 0) return 2;
        // rank 0 creates the object and THEN sizes it: touching a page of a still-empty object is a SIGBUS (seen once as a
        // "hung" test on a cold box: this rank died, rank 0 waited at the barrier) -- wait until it has its full size
        struct stat sb;
        for (int i = 0; i < 20000; ++i) {
            if (fstat(fd, &sb) == 0 && (size_t)sb.st_size >= sizeof(Segment)) break;
            usleep(1000);
        }
        if (fstat(fd, &sb) != 0 || (size_t)sb.st_size < sizeof(Segment)) { close(fd); return 2; }
    }
    void* p = mmap(nullptr, sizeof(Segment), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (p == MAP_FAILED) return 2;
    Segment* seg = static_cast<Segment*>(p);
    if (rank == 0) {
        pthread_barrierattr_t a;
        pthread_barrierattr_init(&a);
        pthread_barrierattr_setpshared(&a, PTHREAD_PROCESS_SHARED);
        pthread_barrier_init(&seg->bar, &a, (unsigned)nranks);
        seg->ready.store(1);
    } else {
        for (int i = 0; i < 20000 && seg->ready.load() != 1; ++i) usleep(1000);
        if (seg->ready.load() != 1) return 2;
    }
    Comm* cm = new Comm{rank, nranks, seg, {0}};
    snprintf(cm->name, sizeof cm->name, "%s", name);
    *comm = cm;
    pthread_barrier_wait(&seg->bar);
    return 0;
}
int ncclCommInitAll(void** comms, int n, const int*) {
    if (n > 64) return 3;
    for (int i = 0; i < n; ++i) comms[i] = new Comm{i, n, nullptr, {0}};
    return 0;
}
int ncclCommDestroy(void* c) {
    Comm* comm = static_cast<Comm*>(c);
    if (!comm->seg) { delete comm; return 0; }
    munmap(comm->seg, sizeof(Segment));
    if (comm->rank == 0 && comm->name[0]) shm_unlink(comm->name);
    delete comm;
    return 0;
}
int ncclBroadcast(const void* send, void* recv, size_t count, int dtype, int root, void* c, hipStream_t st) {
    Comm* comm = static_cast<Comm*>(c);
    const size_t bytes = count * type_size(dtype);
    if (!comm->seg) {                               // in-process communicator: executed at ncclGroupEnd
        if (g_depth < 1 || g_nops >= 64) return 4;
        g_ops[g_nops++] = Op{send, recv, bytes, root, comm, st, 0};
        return 0;
    }
    if (bytes > CAP) return 3;
    if (comm->rank == root) {
        if (hipMemcpyAsync(comm->seg->data, send, bytes, hipMemcpyDeviceToHost, st) != hipSuccess) return 1;
        if (send != recv && hipMemcpyAsync(recv, send, bytes, hipMemcpyDeviceToDevice, st) != hipSuccess) return 1;
        if (hipStreamSynchronize(st) != hipSuccess) return 1;
        pthread_barrier_wait(&comm->seg->bar);
    } else {
        pthread_barrier_wait(&comm->seg->bar);
        if (hipMemcpyAsync(recv, comm->seg->data, bytes, hipMemcpyHostToDevice, st) != hipSuccess) return 1;
        if (hipStreamSynchronize(st) != hipSuccess) return 1;
    }
    pthread_barrier_wait(&comm->seg->bar);          // everyone has read: the segment may be overwritten
    return 0;
}
int ncclAllReduce(const void* send, void* recv, size_t count, int dtype, int op /* 0 sum, 2 max */, void* c, hipStream_t st) {
    Comm* comm = static_cast<Comm*>(c);
    if (dtype != 8 || count > 8 || comm->nranks > 64 || (op != 0 && op != 2)) return 3;
    double mine[8], tot[8] = {0};
    if (hipMemcpyAsync(mine, send, count * 8, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) return 1;
    if (!comm->seg) {                               // in-process communicator of one rank
        if (hipMemcpyAsync(recv, mine, count * 8, hipMemcpyHostToDevice, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) return 1;
        return 0;
    }
    memcpy(comm->seg->red + 8 * comm->rank, mine, count * 8);
    pthread_barrier_wait(&comm->seg->bar);
    for (int r = 0; r < comm->nranks; ++r)
        for (size_t i = 0; i < count; ++i) {
            const double v = comm->seg->red[8 * r + i];
            tot[i] = (op == 2 && r > 0) ? (v > tot[i] ? v : tot[i]) : (op == 2 ? v : tot[i] + v);
        }
    pthread_barrier_wait(&comm->seg->bar);
    if (hipMemcpyAsync(recv, tot, count * 8, hipMemcpyHostToDevice, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) return 1;
    return 0;
}
// ---- the three calls of the library's two-hop panel broadcast.  Point-to-point calls are only meaningful inside a group
// (as in NCCL); they are recorded and executed at ncclGroupEnd.  Multi-process mode implements exactly the pattern the
// library issues -- ONE rank sends equal pieces to all the others, every other rank receives one piece from it -- as a
// collective step over the shared segment (piece for rank r at r * bytes).
int ncclSend(const void* send, size_t count, int dtype, int peer, void* c, hipStream_t st) {
    if (g_depth < 1 || g_nops >= 64) return 4;
    g_ops[g_nops++] = Op{send, nullptr, count * type_size(dtype), peer, static_cast<Comm*>(c), st, 1};
    return 0;
}
int ncclRecv(void* recv, size_t count, int dtype, int peer, void* c, hipStream_t st) {
    if (g_depth < 1 || g_nops >= 64) return 4;
    g_ops[g_nops++] = Op{nullptr, recv, count * type_size(dtype), peer, static_cast<Comm*>(c), st, 2};
    return 0;
}
int ncclAllGather(const void* send, void* recv, size_t count, int dtype, void* c, hipStream_t st) {
    Comm* comm = static_cast<Comm*>(c);
    const size_t bytes = count * type_size(dtype);
    if (!comm->seg) {                               // in-process communicator: executed at ncclGroupEnd
        if (g_depth < 1 || g_nops >= 64) return 4;
        g_ops[g_nops++] = Op{send, recv, bytes, -1, comm, st, 3};
        return 0;
    }
    if (bytes * (size_t)comm->nranks > CAP) return 3;
    if (hipMemcpyAsync(comm->seg->data + (size_t)comm->rank * bytes, send, bytes, hipMemcpyDeviceToHost, st) != hipSuccess) return 1;
    if (hipStreamSynchronize(st) != hipSuccess) return 1;
    pthread_barrier_wait(&comm->seg->bar);
    for (int r = 0; r < comm->nranks; ++r) {
        char* dst = static_cast<char*>(recv) + (size_t)r * bytes;
        if (r == comm->rank && dst == send) continue;   // in place
        if (hipMemcpyAsync(dst, comm->seg->data + (size_t)r * bytes, bytes, hipMemcpyHostToDevice, st) != hipSuccess) return 1;
    }
    if (hipStreamSynchronize(st) != hipSuccess) return 1;
    pthread_barrier_wait(&comm->seg->bar);
    return 0;
}
int ncclGroupStart() { ++g_depth; return 0; }
int ncclGroupEnd() {
    if (--g_depth > 0) return 0;
    int rc = 0;
    // point-to-point operations recorded in this group
    bool p2p = false, gather = false;
    for (int i = 0; i < g_nops; ++i) { p2p = p2p || g_ops[i].kind == 1 || g_ops[i].kind == 2; gather = gather || g_ops[i].kind == 3; }
    if (p2p) {
        Comm* c0 = g_ops[0].comm;
        if (c0->seg) {
            // multi-process: this rank either sends (pieces to everybody else) or receives its one piece
            for (int i = 0; i < g_nops && rc == 0; ++i) {
                const Op& o = g_ops[i];
                if (o.kind != 1) continue;
                if (o.bytes * (size_t)o.comm->nranks > CAP) { rc = 3; break; }
                if (hipMemcpyAsync(o.comm->seg->data + (size_t)o.root * o.bytes, o.send, o.bytes, hipMemcpyDeviceToHost, o.st) != hipSuccess) rc = 1;
            }
            for (int i = 0; i < g_nops && rc == 0; ++i)
                if (g_ops[i].kind == 1 && hipStreamSynchronize(g_ops[i].st) != hipSuccess) rc = 1;
            pthread_barrier_wait(&c0->seg->bar);
            for (int i = 0; i < g_nops && rc == 0; ++i) {
                const Op& o = g_ops[i];
                if (o.kind != 2) continue;
                if (hipMemcpyAsync(o.recv, o.comm->seg->data + (size_t)o.comm->rank * o.bytes, o.bytes, hipMemcpyHostToDevice, o.st) != hipSuccess ||
                    hipStreamSynchronize(o.st) != hipSuccess) rc = 1;
            }
            pthread_barrier_wait(&c0->seg->bar);
        } else {
            // in-process: match every recv (on rank b, from peer a) with the send (on rank a, to peer b)
            for (int i = 0; i < g_nops && rc == 0; ++i) {
                const Op& r = g_ops[i];
                if (r.kind != 2) continue;
                const Op* sd = nullptr;
                for (int j = 0; j < g_nops; ++j)
                    if (g_ops[j].kind == 1 && g_ops[j].comm->rank == r.root && g_ops[j].root == r.comm->rank) sd = &g_ops[j];
                if (!sd || sd->bytes != r.bytes) { rc = 4; break; }
                if (hipStreamSynchronize(sd->st) != hipSuccess ||
                    hipMemcpyAsync(r.recv, sd->send, r.bytes, hipMemcpyDeviceToDevice, r.st) != hipSuccess) rc = 1;
            }
        }
        g_nops = 0;
        return rc;
    }
    if (gather) {                                   // in-process all-gather: every rank's piece to every other rank's buffer
        for (int i = 0; i < g_nops && rc == 0; ++i)
            if (hipStreamSynchronize(g_ops[i].st) != hipSuccess) rc = 1;
        for (int d = 0; d < g_nops && rc == 0; ++d)
            for (int sI = 0; sI < g_nops && rc == 0; ++sI) {
                const Op &dst = g_ops[d], &src = g_ops[sI];
                char* to = static_cast<char*>(dst.recv) + (size_t)src.comm->rank * src.bytes;
                if (to == src.send) continue;
                if (hipMemcpyAsync(to, src.send, src.bytes, hipMemcpyDeviceToDevice, dst.st) != hipSuccess) rc = 1;
            }
        g_nops = 0;
        return rc;
    }
    const Op* src = nullptr;
    for (int i = 0; i < g_nops; ++i)
        if (g_ops[i].comm->rank == g_ops[i].root) src = &g_ops[i];
    if (g_nops > 0 && !src) rc = 4;
    if (src) {
        if (hipStreamSynchronize(src->st) != hipSuccess) rc = 1;       // the root's data is final
        for (int i = 0; i < g_nops && rc == 0; ++i) {
            const Op& o = g_ops[i];
            const void* from = (&o == src) ? o.send : src->send;
            if (o.recv != from && hipMemcpyAsync(o.recv, from, o.bytes, hipMemcpyDeviceToDevice, o.st) != hipSuccess) rc = 1;
        }
    }
    g_nops = 0;
    return rc;
}
const char* ncclGetErrorString(int r) {
    static const char* names[] = {"success", "hip error", "shared-memory setup failed", "message too large for the fake",
                                  "not supported by the fake", "FAKE_RCCL_SHM not set"};
    return (r >= 0 && r < 6) ? names[r] : "unknown";
}
}
