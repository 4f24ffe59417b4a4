// fake_rccl.cpp -- TESTS-ONLY stand-in for librccl, so that the MULTI-PROCESS path of the library (gphip_create_rank:
// one process per rank, collectives between processes) can run with world_size 2 on a box that has ONE GPU, where real
// RCCL refuses two ranks on the same device.  libgphip binds RCCL with dlopen (csrc/rccl_dyn.h) and honours
// $GPHIP_RCCL_PATH, which the test points here.  Same nine entry points, same call semantics as the library uses them
// (in-place broadcast of bytes on a stream, sum all-reduce of doubles, grouped calls); the transport is a POSIX
// shared-memory segment + a process-shared barrier, staged through the host:
//     root:   stream-ordered D2H copy of its buffer into the segment, wait, barrier
//     others: barrier, stream-ordered H2D copy out of the segment, wait;  barrier again before the segment is reused
// Blocking the host inside a collective is slower than RCCL's asynchronous kernels but orders the data the same way.
// Segment name: $FAKE_RCCL_SHM (every rank of a job gets the same value); capacity FAKE_RCCL_CAP bytes.
#include <fcntl.h>
#include <hip/hip_runtime.h>
#include <pthread.h>
#include <sys/mman.h>
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
struct Comm { int rank, nranks; Segment* seg; };
size_t type_size(int t) { return (t == 0 || t == 1) ? 1 : (t == 7 ? 4 : 8); }     // ncclChar/Uint8, Float32, Float64
}  // namespace

struct FakeId { char internal[128]; };

extern "C" {
int ncclGetUniqueId(FakeId* id) { memset(id, 0, sizeof *id); memcpy(id->internal, "fake-rccl", 9); return 0; }

int ncclCommInitRank(void** comm, int nranks, FakeId, int rank) {
    const char* name = getenv("FAKE_RCCL_SHM");
    if (!name) return 5;
    int fd = -1;
    if (rank == 0) {
        shm_unlink(name);
        fd = shm_open(name, O_CREAT | O_RDWR, 0600);
        if (fd < 0 || ftruncate(fd, sizeof(Segment)) != 0) return 2;
    } else {
        for (int i = 0; i < 20000 && fd < 0; ++i) { fd = shm_open(name, O_RDWR, 0600); if (fd < 0) usleep(1000); }
        if (fd < 0) return 2;
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
    *comm = new Comm{rank, nranks, seg};
    pthread_barrier_wait(&seg->bar);
    return 0;
}
int ncclCommInitAll(void**, int, const int*) { return 4; }      // single-process multi-device: not what this fake is for
int ncclCommDestroy(void* c) {
    Comm* comm = static_cast<Comm*>(c);
    munmap(comm->seg, sizeof(Segment));
    if (comm->rank == 0 && getenv("FAKE_RCCL_SHM")) shm_unlink(getenv("FAKE_RCCL_SHM"));
    delete comm;
    return 0;
}
int ncclBroadcast(const void* send, void* recv, size_t count, int dtype, int root, void* c, hipStream_t st) {
    Comm* comm = static_cast<Comm*>(c);
    const size_t bytes = count * type_size(dtype);
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
int ncclAllReduce(const void* send, void* recv, size_t count, int dtype, int /*op: sum*/, void* c, hipStream_t st) {
    Comm* comm = static_cast<Comm*>(c);
    if (dtype != 8 || count > 8 || comm->nranks > 64) return 3;
    double mine[8], tot[8] = {0};
    if (hipMemcpyAsync(mine, send, count * 8, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) return 1;
    memcpy(comm->seg->red + 8 * comm->rank, mine, count * 8);
    pthread_barrier_wait(&comm->seg->bar);
    for (int r = 0; r < comm->nranks; ++r)
        for (size_t i = 0; i < count; ++i) tot[i] += comm->seg->red[8 * r + i];
    pthread_barrier_wait(&comm->seg->bar);
    if (hipMemcpyAsync(recv, tot, count * 8, hipMemcpyHostToDevice, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) return 1;
    return 0;
}
int ncclGroupStart() { return 0; }
int ncclGroupEnd() { return 0; }
const char* ncclGetErrorString(int r) {
    static const char* names[] = {"success", "hip error", "shared-memory setup failed", "message too large for the fake",
                                  "not supported by the fake", "FAKE_RCCL_SHM not set"};
    return (r >= 0 && r < 6) ? names[r] : "unknown";
}
}
