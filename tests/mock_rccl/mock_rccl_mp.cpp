// mock_rccl_mp.cpp -- TEST INFRASTRUCTURE.  Multi-PROCESS stand-in for the RCCL entry points
// liblam_hip.so uses, so that bench.py's torchrun path (one process per rank, gloo control plane,
// unique-id broadcast, both exchange modes, max-over-ranks timing) can be run end to end with 2-4
// ranks on a box that has ONE GPU (RCCL refuses two ranks on one device).  LD_PRELOADed in front of
// librccl.so.  Every collective is: hipStreamSynchronize -> copy the send buffer to a POSIX shared
// memory slot -> barrier -> read the peers' slots -> copy to the receive buffer -> barrier.  It checks
// that all ranks issue the same operation with the same count.  Nothing about it is fast.
// Build: hipcc -O2 -shared -fPIC tests/mock_rccl/mock_rccl_mp.cpp -o tests/mock_rccl/libmock_rccl_mp.so -lrt
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <atomic>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include <fcntl.h>
#include <sys/mman.h>
#include <unistd.h>

namespace
{
constexpr int kMaxRanks = 8;
constexpr size_t kSlot = 8u << 20;   // bytes one rank can contribute to one collective

struct Shm {
    std::atomic<int> count, gen, attached;
    int nranks;
    size_t counts[kMaxRanks];
    int ops[kMaxRanks];
    char data[kMaxRanks][kSlot];
};
struct Comm { Shm *shm; int rank, nranks; std::string name; };

void barrier(Shm *s, int n)
{
    const int g = s->gen.load();
    if (s->count.fetch_add(1) + 1 == n) { s->count.store(0); s->gen.fetch_add(1); }
    else while (s->gen.load() == g) usleep(20);
}

size_t dsize(ncclDataType_t t)
{
    return t == ncclDouble ? 8 : (t == ncclFloat ? 4 : ((t == ncclChar || t == ncclUint8) ? 1 : 0));
}

ncclResult_t publish(Comm *c, const void *send, size_t bytes, size_t count, int op, hipStream_t stream, bool contribute)
{
    if (bytes > kSlot) { fprintf(stderr, "[mock rccl mp] message of %zu bytes exceeds the slot\n", bytes); return ncclInvalidArgument; }
    if (hipStreamSynchronize(stream) != hipSuccess) return ncclUnhandledCudaError;
    if (contribute && hipMemcpy(c->shm->data[c->rank], send, bytes, hipMemcpyDeviceToHost) != hipSuccess) return ncclUnhandledCudaError;
    c->shm->counts[c->rank] = count;
    c->shm->ops[c->rank] = op;
    barrier(c->shm, c->nranks);
    for (int q = 0; q < c->nranks; q++)
        if (c->shm->ops[q] != op || (op != 3 && c->shm->counts[q] != count)) {
            fprintf(stderr, "[mock rccl mp] rank %d: mismatched collective (op %d vs %d, count %zu vs %zu)\n", c->rank, op,
                    c->shm->ops[q], count, c->shm->counts[q]);
            return ncclInvalidUsage;
        }
    return ncclSuccess;
}
}  // namespace

extern "C" {

const char *ncclGetErrorString(ncclResult_t r) { return r == ncclSuccess ? "no error" : "mock rccl (mp) error"; }

ncclResult_t ncclGetUniqueId(ncclUniqueId *id)
{
    static int counter = 0;
    memset(id, 0, sizeof *id);
    snprintf(id->internal, sizeof id->internal, "/lam_mock_rccl_%d_%d", (int)getpid(), counter++);
    return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t *comm, int nranks, ncclUniqueId id, int rank)
{
    if (nranks > kMaxRanks) return ncclInvalidArgument;
    const std::string name(id.internal);
    int fd = -1;
    if (rank == 0) {
        shm_unlink(name.c_str());
        fd = shm_open(name.c_str(), O_CREAT | O_RDWR, 0600);
        if (fd < 0 || ftruncate(fd, sizeof(Shm)) != 0) return ncclSystemError;
    } else {
        for (int tries = 0; tries < 3000 && fd < 0; tries++) {   // wait for rank 0 (<= 30 s)
            fd = shm_open(name.c_str(), O_RDWR, 0600);
            if (fd < 0) usleep(10000);
        }
        if (fd < 0) return ncclSystemError;
        off_t sz = 0;
        for (int tries = 0; tries < 3000 && sz < (off_t)sizeof(Shm); tries++) { sz = lseek(fd, 0, SEEK_END); if (sz < (off_t)sizeof(Shm)) usleep(10000); }
    }
    void *p = mmap(nullptr, sizeof(Shm), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (p == MAP_FAILED) return ncclSystemError;
    Shm *s = static_cast<Shm *>(p);
    if (rank == 0) s->nranks = nranks;               // a fresh segment is zero-filled: counters start at 0
    s->attached.fetch_add(1);
    while (s->attached.load() < nranks) usleep(100);  // everyone is mapped before the first barrier
    *comm = reinterpret_cast<ncclComm_t>(new Comm{s, rank, nranks, name});
    barrier(s, nranks);
    if (rank == 0) shm_unlink(name.c_str());          // the mapping stays valid
    return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t comm)
{
    Comm *c = reinterpret_cast<Comm *>(comm);
    munmap(c->shm, sizeof(Shm));
    delete c;
    return ncclSuccess;
}
ncclResult_t ncclCommCount(const ncclComm_t comm, int *count) { *count = reinterpret_cast<const Comm *>(comm)->nranks; return ncclSuccess; }
ncclResult_t ncclCommUserRank(const ncclComm_t comm, int *rank) { *rank = reinterpret_cast<const Comm *>(comm)->rank; return ncclSuccess; }

ncclResult_t ncclGroupStart() { return ncclSuccess; }
ncclResult_t ncclGroupEnd() { return ncclSuccess; }

ncclResult_t ncclAllReduce(const void *sendbuff, void *recvbuff, size_t count, ncclDataType_t dt, ncclRedOp_t op,
                           ncclComm_t comm, hipStream_t stream)
{
    Comm *c = reinterpret_cast<Comm *>(comm);
    if (dt != ncclDouble || op != ncclSum) return ncclInvalidArgument;
    ncclResult_t r = publish(c, sendbuff, count * 8, count, 1, stream, true);
    if (r != ncclSuccess) return r;
    std::vector<double> acc(count);
    for (int q = 0; q < c->nranks; q++) {
        const double *v = reinterpret_cast<const double *>(c->shm->data[q]);
        for (size_t i = 0; i < count; i++) acc[i] = q == 0 ? v[i] : acc[i] + v[i];
    }
    if (hipMemcpy(recvbuff, acc.data(), count * 8, hipMemcpyHostToDevice) != hipSuccess) return ncclUnhandledCudaError;
    barrier(c->shm, c->nranks);
    return ncclSuccess;
}

ncclResult_t ncclAllGather(const void *sendbuff, void *recvbuff, size_t sendcount, ncclDataType_t dt, ncclComm_t comm,
                           hipStream_t stream)
{
    Comm *c = reinterpret_cast<Comm *>(comm);
    const size_t bytes = sendcount * dsize(dt);
    if (bytes == 0) return ncclInvalidArgument;
    ncclResult_t r = publish(c, sendbuff, bytes, sendcount, 2, stream, true);
    if (r != ncclSuccess) return r;
    for (int q = 0; q < c->nranks; q++)
        if (hipMemcpy(static_cast<char *>(recvbuff) + (size_t)q * bytes, c->shm->data[q], bytes, hipMemcpyHostToDevice) != hipSuccess)
            return ncclUnhandledCudaError;
    barrier(c->shm, c->nranks);
    return ncclSuccess;
}

ncclResult_t ncclBroadcast(const void *sendbuff, void *recvbuff, size_t count, ncclDataType_t dt, int root, ncclComm_t comm,
                           hipStream_t stream)
{
    Comm *c = reinterpret_cast<Comm *>(comm);
    const size_t bytes = count * dsize(dt);
    ncclResult_t r = publish(c, sendbuff, bytes, count, 3, stream, c->rank == root);
    if (r != ncclSuccess) return r;
    if (c->shm->counts[root] != count) return ncclInvalidUsage;
    if (c->rank != root || recvbuff != sendbuff)
        if (hipMemcpy(recvbuff, c->shm->data[root], bytes, hipMemcpyHostToDevice) != hipSuccess) return ncclUnhandledCudaError;
    barrier(c->shm, c->nranks);
    return ncclSuccess;
}

}  // extern "C"
