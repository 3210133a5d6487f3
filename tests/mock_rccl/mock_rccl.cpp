// mock_rccl.cpp -- TEST INFRASTRUCTURE.  A host-synchronous, in-process stand-in for the handful of
// RCCL entry points liblam_hip.so uses, so that the one-process-per-GPU ("rank") mode can be driven
// with P > 1 ranks on a box that has ONE GPU (RCCL itself refuses two ranks on one device).
// The P ranks are P threads of one process, each with its own lam_hip context on device 0; this
// library is LD_PRELOADed in front of librccl.so and implements every collective as
//     hipStreamSynchronize(stream) -> publish pointers -> barrier -> copy/sum -> barrier.
// It checks what the real library would need from the caller: same call sequence and counts on every
// rank, in-place all-gather pointer arithmetic, root semantics of broadcast.  It says nothing about
// performance.  Build: hipcc -shared -fPIC tests/mock_rccl/mock_rccl.cpp -o tests/mock_rccl/libmock_rccl.so
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <atomic>
#include <condition_variable>
#include <cstdio>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>

namespace
{
struct Barrier {
    std::mutex mu;
    std::condition_variable cv;
    int count = 0, gen = 0, n = 0;
    void wait()
    {
        std::unique_lock<std::mutex> lk(mu);
        const int g = gen;
        if (++count == n) { count = 0; gen++; cv.notify_all(); }
        else cv.wait(lk, [&] { return gen != g; });
    }
};
struct World {
    int nranks = 0;
    Barrier bar;
    std::vector<const void *> send;
    std::vector<size_t> count;
    std::vector<int> opcode;
    int joined = 0;
};
struct Comm { World *w; int rank; };

std::mutex g_mu;
std::map<std::string, World *> g_worlds;
std::atomic<int> g_ids{1};

size_t dsize(ncclDataType_t t)
{
    return t == ncclDouble ? 8 : (t == ncclFloat ? 4 : ((t == ncclChar || t == ncclUint8) ? 1 : 0));
}

// every rank must be executing the same operation with the same count
ncclResult_t publish(Comm *c, const void *send, size_t count, int opcode, hipStream_t stream)
{
    if (hipStreamSynchronize(stream) != hipSuccess) return ncclUnhandledCudaError;
    World *w = c->w;
    w->send[c->rank] = send;
    w->count[c->rank] = count;
    w->opcode[c->rank] = opcode;
    w->bar.wait();
    for (int q = 0; q < w->nranks; q++)
        if (w->opcode[q] != opcode || (opcode != 3 && w->count[q] != count)) {
            fprintf(stderr, "[mock rccl] rank %d: mismatched collective (op %d/%d count %zu/%zu)\n", c->rank, opcode,
                    w->opcode[q], count, w->count[q]);
            return ncclInvalidUsage;
        }
    return ncclSuccess;
}
}  // namespace

extern "C" {

const char *ncclGetErrorString(ncclResult_t r) { return r == ncclSuccess ? "no error" : "mock rccl error"; }

ncclResult_t ncclGetUniqueId(ncclUniqueId *id)
{
    memset(id, 0, sizeof *id);
    snprintf(id->internal, sizeof id->internal, "mock-rccl-%d", g_ids.fetch_add(1));
    return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t *comm, int nranks, ncclUniqueId id, int rank)
{
    World *w;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        std::string key(id.internal, sizeof id.internal);
        auto it = g_worlds.find(key);
        if (it == g_worlds.end()) {
            w = new World;
            w->nranks = nranks;
            w->bar.n = nranks;
            w->send.resize(nranks); w->count.resize(nranks); w->opcode.resize(nranks);
            g_worlds[key] = w;
        } else w = it->second;
        if (w->nranks != nranks || rank < 0 || rank >= nranks) return ncclInvalidArgument;
        w->joined++;
    }
    *comm = reinterpret_cast<ncclComm_t>(new Comm{w, rank});
    w->bar.wait();
    return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t comm) { delete reinterpret_cast<Comm *>(comm); return ncclSuccess; }
ncclResult_t ncclGroupStart() { return ncclSuccess; }
ncclResult_t ncclGroupEnd() { return ncclSuccess; }

ncclResult_t ncclAllReduce(const void *sendbuff, void *recvbuff, size_t count, ncclDataType_t dt, ncclRedOp_t op,
                           ncclComm_t comm, hipStream_t stream)
{
    Comm *c = reinterpret_cast<Comm *>(comm);
    if (dt != ncclDouble || op != ncclSum) return ncclInvalidArgument;
    ncclResult_t r = publish(c, sendbuff, count, 1, stream);
    if (r != ncclSuccess) return r;
    std::vector<double> acc(count, 0.0), tmp(count);
    for (int q = 0; q < c->w->nranks; q++) {
        if (hipMemcpy(tmp.data(), c->w->send[q], count * 8, hipMemcpyDeviceToHost) != hipSuccess) return ncclUnhandledCudaError;
        for (size_t i = 0; i < count; i++) acc[i] = q == 0 ? tmp[i] : acc[i] + tmp[i];
    }
    c->w->bar.wait();                                   // everyone has read every send buffer
    if (hipMemcpy(recvbuff, acc.data(), count * 8, hipMemcpyHostToDevice) != hipSuccess) return ncclUnhandledCudaError;
    c->w->bar.wait();
    return ncclSuccess;
}

ncclResult_t ncclAllGather(const void *sendbuff, void *recvbuff, size_t sendcount, ncclDataType_t dt, ncclComm_t comm,
                           hipStream_t stream)
{
    Comm *c = reinterpret_cast<Comm *>(comm);
    const size_t bytes = sendcount * dsize(dt);
    if (bytes == 0) return ncclInvalidArgument;
    ncclResult_t r = publish(c, sendbuff, sendcount, 2, stream);
    if (r != ncclSuccess) return r;
    for (int q = 0; q < c->w->nranks; q++) {
        char *dst = static_cast<char *>(recvbuff) + (size_t)q * bytes;
        if (dst == c->w->send[q]) continue;             // in-place slot of this rank
        if (q == c->rank && dst != sendbuff && (const char *)sendbuff > (const char *)recvbuff &&
            (const char *)sendbuff < (const char *)recvbuff + (size_t)c->w->nranks * bytes) {
            fprintf(stderr, "[mock rccl] rank %d: in-place all-gather with sendbuff != recvbuff + rank*count\n", c->rank);
            return ncclInvalidUsage;
        }
        if (hipMemcpy(dst, c->w->send[q], bytes, hipMemcpyDeviceToDevice) != hipSuccess) return ncclUnhandledCudaError;
    }
    if (hipDeviceSynchronize() != hipSuccess) return ncclUnhandledCudaError;
    c->w->bar.wait();
    return ncclSuccess;
}

ncclResult_t ncclBroadcast(const void *sendbuff, void *recvbuff, size_t count, ncclDataType_t dt, int root, ncclComm_t comm,
                           hipStream_t stream)
{
    Comm *c = reinterpret_cast<Comm *>(comm);
    const size_t bytes = count * dsize(dt);
    ncclResult_t r = publish(c, sendbuff, count, 3, stream);
    if (r != ncclSuccess) return r;
    if (c->w->count[root] != count) return ncclInvalidUsage;
    const void *src = c->w->send[root];
    if (recvbuff != src && hipMemcpy(recvbuff, src, bytes, hipMemcpyDeviceToDevice) != hipSuccess) return ncclUnhandledCudaError;
    if (hipDeviceSynchronize() != hipSuccess) return ncclUnhandledCudaError;
    c->w->bar.wait();
    return ncclSuccess;
}

}  // extern "C"
