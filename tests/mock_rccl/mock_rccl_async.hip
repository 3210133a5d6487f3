// mock_rccl_async.hip -- TEST INFRASTRUCTURE.  A STREAM-ORDERED stand-in for the RCCL entry points
// liblam_hip.so uses, so that the one-process-per-GPU ("rank") mode can be driven with P > 1 ranks on
// a box that has ONE GPU (RCCL itself refuses two ranks on one device) under the semantics real RCCL
// has: a collective call only ENQUEUES work on the caller's stream and returns; nothing synchronises
// the hosts, nothing synchronises a stream.  LD_PRELOADed in front of librccl.so.
//
// Every collective is one 1-workgroup kernel on the caller's stream:
//     wait until ring entry (seq % kRing) is free          (all ranks finished collective seq - kRing)
//     copy the send buffer into this rank's slot of the entry, publish {seq, op, count}
//     spin until every rank has published seq               (generation check, op/count compared)
//     combine: all-reduce = sum in RANK ORDER, all-gather / broadcast = copy;  mark the entry read
// `seq` is the per-communicator call counter of the CALLING HOST: ranks whose hosts enqueue a
// different sequence of collectives (the failure a host-synchronous mock cannot show) either disagree
// on {op, count} for some seq -- flagged -- or leave a kernel waiting for a call that never comes:
// every spin is bounded (MOCK_RCCL_TIMEOUT_MS, default 20000) and ends with the abort flag up, after
// which every other spin returns at once, so the grid always drains.
// Ranks may be threads of one process (they share the device buffers by pointer) or separate
// processes on the same GPU (HIP IPC handles published through POSIX shared memory).
//
// Streams must not share a hardware queue, or a waiting kernel could sit in front of the kernel it
// waits for: run with GPU_MAX_HW_QUEUES >= 2 * ranks + 2 (tests/test_gpu_rank_mock.py sets it).
//
// Diagnostics: MOCK_RCCL_STATS_FILE=<path> gets one JSON line per communicator at ncclCommDestroy
// ({"rank", "calls", "abort", "err", ...}); MOCK_RCCL_HOST_DELAY_US="r:us,r:us" sleeps on rank r before
// every enqueue (makes that rank's host lag behind its GPU).
// Build: hipcc -O2 -std=c++17 --offload-arch=gfx950 -shared -fPIC mock_rccl_async.hip -o libmock_rccl_async.so -lrt
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <vector>
#include <mutex>
#include <string>
#include <thread>

#include <fcntl.h>
#include <sys/mman.h>
#include <unistd.h>

namespace
{
constexpr int kMaxRanks = 64;
constexpr int kRing = 4;
constexpr size_t kSlotBytes = 1u << 20;        // one rank's contribution to one collective
constexpr int kThreads = 256;
enum Op : unsigned { kAllReduce = 1, kAllGather = 2, kBroadcast = 3 };
enum Err : unsigned { kErrNone = 0, kErrTimeoutRing = 1, kErrTimeoutPeer = 2, kErrMismatch = 3, kErrOrder = 4 };

struct Ctrl {                                   // device memory, zero-initialised by rank 0
    unsigned long long seq[kRing][kMaxRanks];   // published call number + 1
    unsigned long long count[kRing][kMaxRanks];
    unsigned op[kRing][kMaxRanks];
    unsigned long long done[kRing];             // ranks that finished reading the entry, monotonic
    unsigned long long completed[kMaxRanks];    // calls of rank q that have finished (its next call number)
    unsigned abort;                             // a kernel timed out or saw a mismatch: everybody bail out
    unsigned err, err_rank, err_peer;
    unsigned long long err_seq;
};

__device__ __forceinline__ unsigned long long ld64(const unsigned long long *p)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ unsigned ld32(const unsigned *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

__device__ void raise_error(Ctrl *c, volatile int *host_abort, unsigned err, int rank, int peer, unsigned long long seq)
{
    if (atomicCAS(&c->abort, 0u, 1u) == 0u) {
        c->err = err; c->err_rank = (unsigned)rank; c->err_peer = (unsigned)peer; c->err_seq = seq;
    }
    __threadfence_system();
    *host_abort = (int)err;
}

__device__ void copy_bytes(char *dst, const char *src, size_t bytes)
{
    if ((((size_t)dst | (size_t)src | bytes) & 7) == 0) {
        const size_t n = bytes / 8;
        for (size_t i = threadIdx.x; i < n; i += kThreads) reinterpret_cast<unsigned long long *>(dst)[i] = reinterpret_cast<const unsigned long long *>(src)[i];
    } else {
        for (size_t i = threadIdx.x; i < bytes; i += kThreads) dst[i] = src[i];
    }
}

__global__ void __launch_bounds__(kThreads)
collective_kernel(Ctrl *ctrl, char *slots, volatile int *host_abort, int rank, int nranks, unsigned long long seq, unsigned op,
                  const void *send, void *recv, size_t bytes, unsigned long long count, int root, unsigned long long timeout_ticks)
{
    __shared__ int s_ok;
    const int tid = threadIdx.x;
    const int e = (int)(seq % kRing);
    const unsigned long long t0 = wall_clock64();
    // 1. the ring entry must have been read by every rank in its previous use
    if (tid == 0) {
        s_ok = 1;
        // operations on one communicator must be ordered by the caller (stream order or events): this
        // rank's previous call has to be complete when this kernel starts, whichever stream it ran on
        if (ld64(&ctrl->completed[rank]) != seq) { raise_error(ctrl, host_abort, kErrOrder, rank, -1, seq); s_ok = 0; }
        const unsigned long long need = (unsigned long long)nranks * (seq / kRing);
        while (s_ok && ld64(&ctrl->done[e]) < need) {
            if (ld32(&ctrl->abort)) { s_ok = 0; break; }
            if (wall_clock64() - t0 > timeout_ticks) { raise_error(ctrl, host_abort, kErrTimeoutRing, rank, -1, seq); s_ok = 0; break; }
            __builtin_amdgcn_s_sleep(20);
        }
    }
    __syncthreads();
    if (!s_ok) return;
    // 2. contribute and publish
    char *entry = slots + (size_t)e * kMaxRanks * kSlotBytes;
    if (op != kBroadcast || rank == root) copy_bytes(entry + (size_t)rank * kSlotBytes, static_cast<const char *>(send), bytes);
    __threadfence();
    __syncthreads();
    if (tid == 0) {
        ctrl->op[e][rank] = op;
        ctrl->count[e][rank] = count;
        __threadfence();
        __hip_atomic_store(&ctrl->seq[e][rank], seq + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    // 3. every rank has published this call number (thread q watches rank q)
    __syncthreads();
    if (tid < nranks) {
        while (ld64(&ctrl->seq[e][tid]) != seq + 1) {
            if (ld32(&ctrl->abort)) { s_ok = 0; break; }
            if (wall_clock64() - t0 > timeout_ticks) { raise_error(ctrl, host_abort, kErrTimeoutPeer, rank, tid, seq); s_ok = 0; break; }
            __builtin_amdgcn_s_sleep(20);
        }
    }
    __syncthreads();
    if (!s_ok) return;
    __threadfence();
    if (tid < nranks) {
        const unsigned pop = ld32(&ctrl->op[e][tid]);
        const unsigned long long pcount = ld64(&ctrl->count[e][tid]);
        if (pop != op || (op != kBroadcast && pcount != count) || (op == kBroadcast && tid == root && pcount != count)) {
            raise_error(ctrl, host_abort, kErrMismatch, rank, tid, seq);
            s_ok = 0;
        }
    }
    __syncthreads();
    if (!s_ok) return;
    // 4. combine
    if (op == kAllReduce) {
        double *out = static_cast<double *>(recv);
        for (size_t i = tid; i < count; i += kThreads) {
            double acc = reinterpret_cast<const double *>(entry)[i];
            for (int q = 1; q < nranks; q++) acc += reinterpret_cast<const double *>(entry + (size_t)q * kSlotBytes)[i];
            out[i] = acc;
        }
    } else if (op == kAllGather) {
        for (int q = 0; q < nranks; q++) copy_bytes(static_cast<char *>(recv) + (size_t)q * bytes, entry + (size_t)q * kSlotBytes, bytes);
    } else {
        if (!(rank == root && recv == send)) copy_bytes(static_cast<char *>(recv), entry + (size_t)root * kSlotBytes, bytes);
    }
    __threadfence();
    __syncthreads();
    if (tid == 0) {
        __hip_atomic_store(&ctrl->completed[rank], seq + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_fetch_add(&ctrl->done[e], 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// ---- host side ----------------------------------------------------------------------------------
struct ShmHeader {                       // POSIX shared memory, one per unique id (multi-process worlds)
    std::atomic<int> ready;
    int pid;
    int ipc_ok;
    hipIpcMemHandle_t ctrl_h, slots_h;
};

struct World {
    int nranks = 0;
    Ctrl *ctrl = nullptr;
    char *slots = nullptr;
    bool owner = false;                  // this process allocated the buffers
    std::string shm_name;
    int refs = 0;
};
struct Comm {
    World *w;
    int rank;
    unsigned long long seq = 0;
    int *host_abort = nullptr;           // pinned, written by this rank's kernels
    long delay_us = 0;
    struct Call { unsigned long long seq; unsigned op; size_t count, bytes; int root; void *stream; };
    std::vector<Call> trace;             // MOCK_RCCL_TRACE_DIR
};

std::mutex g_mu;
std::map<std::string, World *> g_worlds;
std::atomic<int> g_ids{1};

size_t dsize(ncclDataType_t t)
{
    return t == ncclDouble ? 8 : (t == ncclFloat ? 4 : ((t == ncclChar || t == ncclUint8) ? 1 : 0));
}

unsigned long long timeout_ticks()
{
    const char *e = getenv("MOCK_RCCL_TIMEOUT_MS");
    const double ms = e && *e ? atof(e) : 20000.0;
    return (unsigned long long)(ms * 1e-3 * 100e6);      // wall_clock64: 100 MHz
}

long host_delay_us(int rank)
{
    const char *e = getenv("MOCK_RCCL_HOST_DELAY_US");
    if (!e) return 0;
    std::string s(e);
    size_t pos = 0;
    while (pos < s.size()) {
        const size_t comma = s.find(',', pos);
        const std::string item = s.substr(pos, comma == std::string::npos ? std::string::npos : comma - pos);
        const size_t colon = item.find(':');
        if (colon != std::string::npos && atoi(item.substr(0, colon).c_str()) == rank) return atol(item.substr(colon + 1).c_str());
        if (comma == std::string::npos) break;
        pos = comma + 1;
    }
    return 0;
}

ncclResult_t enqueue(Comm *c, unsigned op, const void *send, void *recv, size_t bytes, size_t count, int root, hipStream_t stream)
{
    if (bytes > kSlotBytes) {
        fprintf(stderr, "[mock rccl async] message of %zu bytes exceeds the %zu-byte slot\n", bytes, kSlotBytes);
        return ncclInvalidArgument;
    }
    if (*(volatile int *)c->host_abort) {
        fprintf(stderr, "[mock rccl async] rank %d: an earlier collective failed (code %d: 1/2 = timeout waiting for the ring/a peer, "
                        "3 = op/count mismatch, 4 = two calls on one communicator not ordered): the ranks' call sequences differ\n", c->rank, *c->host_abort);
        return ncclInternalError;
    }
    if (c->delay_us > 0) std::this_thread::sleep_for(std::chrono::microseconds(c->delay_us));
    // MOCK_RCCL_TRACE_DIR=<dir>: every call is noted in memory and written to <dir>/rank<r>.txt when the communicator goes (to find
    // where the ranks' call sequences part; a file write per call would change the timing the question is about)
    static const bool tracing = getenv("MOCK_RCCL_TRACE_DIR") != nullptr;
    if (tracing) c->trace.push_back({c->seq, op, count, bytes, root, (void *)stream});
    hipLaunchKernelGGL(collective_kernel, dim3(1), dim3(kThreads), 0, stream, c->w->ctrl, c->w->slots, (volatile int *)c->host_abort,
                       c->rank, c->w->nranks, c->seq, op, send, recv, bytes, (unsigned long long)count, root, timeout_ticks());
    c->seq++;
    return hipGetLastError() == hipSuccess ? ncclSuccess : ncclUnhandledCudaError;
}
}  // namespace

extern "C" {

const char *ncclGetErrorString(ncclResult_t r) { return r == ncclSuccess ? "no error" : "mock rccl (async) error"; }

ncclResult_t ncclGetVersion(int *version) { if (version) *version = 0; return ncclSuccess; }   // 0 = the mock

ncclResult_t ncclGetUniqueId(ncclUniqueId *id)
{
    memset(id, 0, sizeof *id);
    const long long now = std::chrono::steady_clock::now().time_since_epoch().count();
    snprintf(id->internal, sizeof id->internal, "/lam_mock_async_%d_%d_%llx", (int)getpid(), g_ids.fetch_add(1), now & 0xffffffll);
    return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t *comm, int nranks, ncclUniqueId id, int rank)
{
    if (nranks < 1 || nranks > kMaxRanks || rank < 0 || rank >= nranks) return ncclInvalidArgument;
    const std::string key(id.internal);
    World *w = nullptr;
    if (rank == 0) {
        w = new World;
        w->nranks = nranks;
        w->owner = true;
        w->shm_name = key;
        if (hipMalloc((void **)&w->ctrl, sizeof(Ctrl)) != hipSuccess || hipMalloc((void **)&w->slots, (size_t)kRing * kMaxRanks * kSlotBytes) != hipSuccess ||
            hipMemset(w->ctrl, 0, sizeof(Ctrl)) != hipSuccess || hipDeviceSynchronize() != hipSuccess)
            return ncclUnhandledCudaError;
        {
            std::lock_guard<std::mutex> lk(g_mu);
            g_worlds[key] = w;
        }
        // publish for ranks that live in other processes
        shm_unlink(key.c_str());
        const int fd = shm_open(key.c_str(), O_CREAT | O_RDWR, 0600);
        if (fd >= 0 && ftruncate(fd, sizeof(ShmHeader)) == 0) {
            void *p = mmap(nullptr, sizeof(ShmHeader), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
            if (p != MAP_FAILED) {
                ShmHeader *h = static_cast<ShmHeader *>(p);
                h->pid = (int)getpid();
                h->ipc_ok = hipIpcGetMemHandle(&h->ctrl_h, w->ctrl) == hipSuccess && hipIpcGetMemHandle(&h->slots_h, w->slots) == hipSuccess;
                (void)hipGetLastError();
                h->ready.store(1);
                munmap(p, sizeof(ShmHeader));
            }
        }
        if (fd >= 0) close(fd);
    } else {
        for (int tries = 0; tries < 6000 && !w; tries++) {          // <= 60 s
            {
                std::lock_guard<std::mutex> lk(g_mu);
                auto it = g_worlds.find(key);
                if (it != g_worlds.end()) { w = it->second; break; }
            }
            const int fd = shm_open(key.c_str(), O_RDWR, 0600);
            if (fd >= 0) {
                void *p = mmap(nullptr, sizeof(ShmHeader), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
                close(fd);
                if (p != MAP_FAILED) {
                    ShmHeader *h = static_cast<ShmHeader *>(p);
                    if (h->ready.load() && h->pid != (int)getpid()) {
                        if (!h->ipc_ok) { fprintf(stderr, "[mock rccl async] HIP IPC is not available: ranks must be threads of one process\n"); munmap(p, sizeof(ShmHeader)); return ncclSystemError; }
                        std::lock_guard<std::mutex> lk(g_mu);
                        auto it = g_worlds.find(key);       // another thread of THIS process may have opened it already
                        if (it != g_worlds.end()) w = it->second;
                        else {
                            w = new World;
                            w->nranks = nranks;
                            if (hipIpcOpenMemHandle((void **)&w->ctrl, h->ctrl_h, hipIpcMemLazyEnablePeerAccess) != hipSuccess ||
                                hipIpcOpenMemHandle((void **)&w->slots, h->slots_h, hipIpcMemLazyEnablePeerAccess) != hipSuccess) {
                                fprintf(stderr, "[mock rccl async] hipIpcOpenMemHandle failed: %s\n", hipGetErrorString(hipGetLastError()));
                                munmap(p, sizeof(ShmHeader));
                                return ncclSystemError;
                            }
                            g_worlds[key] = w;
                        }
                    }
                    munmap(p, sizeof(ShmHeader));
                }
            }
            if (!w) std::this_thread::sleep_for(std::chrono::milliseconds(10));
        }
        if (!w) { fprintf(stderr, "[mock rccl async] rank %d: rank 0 never showed up\n", rank); return ncclSystemError; }
        if (w->nranks != nranks) return ncclInvalidArgument;
    }
    Comm *c = new Comm{w, rank};
    if (hipHostMalloc((void **)&c->host_abort, 64, hipHostMallocDefault) != hipSuccess) return ncclUnhandledCudaError;
    *c->host_abort = 0;
    c->delay_us = host_delay_us(rank);
    {
        std::lock_guard<std::mutex> lk(g_mu);
        w->refs++;
    }
    *comm = reinterpret_cast<ncclComm_t>(c);
    return ncclSuccess;
}

// The library synchronises its streams before it destroys the communicator.
ncclResult_t ncclCommDestroy(ncclComm_t comm)
{
    Comm *c = reinterpret_cast<Comm *>(comm);
    Ctrl h;
    memset(&h, 0, sizeof h);
    (void)hipMemcpy(&h, c->w->ctrl, sizeof h, hipMemcpyDeviceToHost);
    if (const char *dir = getenv("MOCK_RCCL_TRACE_DIR")) {
        char path[512];
        snprintf(path, sizeof path, "%s/rank%d.txt", dir, c->rank);
        if (FILE *f = fopen(path, "a")) {
            fprintf(f, "# communicator of %d ranks, %llu calls, err %u (rank %u, peer %u, call %llu)\n", c->w->nranks, c->seq, h.err, h.err_rank, h.err_peer, h.err_seq);
            for (const auto &t : c->trace) fprintf(f, "%llu op %u count %zu bytes %zu root %d stream %p\n", t.seq, t.op, t.count, t.bytes, t.root, t.stream);
            fclose(f);
        }
    }
    if (const char *path = getenv("MOCK_RCCL_STATS_FILE")) {
        char line[512];
        const int n = snprintf(line, sizeof line,
                               "{\"rank\": %d, \"nranks\": %d, \"calls\": %llu, \"abort\": %u, \"err\": %u, \"err_rank\": %u, \"err_peer\": %u, "
                               "\"err_seq\": %llu, \"host_abort\": %d, \"pid\": %d}\n",
                               c->rank, c->w->nranks, c->seq, h.abort, h.err, h.err_rank, h.err_peer, h.err_seq, *c->host_abort, (int)getpid());
        const int fd = open(path, O_WRONLY | O_CREAT | O_APPEND, 0600);
        if (fd >= 0) { (void)!write(fd, line, (size_t)n); close(fd); }     // one write: lines of concurrent ranks stay whole
    }
    (void)hipHostFree(c->host_abort);
    {
        std::lock_guard<std::mutex> lk(g_mu);
        if (--c->w->refs == 0 && c->w->owner) shm_unlink(c->w->shm_name.c_str());
        // device buffers are left to process exit: other ranks' kernels may still be reading them
    }
    delete c;
    return ncclSuccess;
}

ncclResult_t ncclCommCount(const ncclComm_t comm, int *count) { *count = reinterpret_cast<const Comm *>(comm)->w->nranks; return ncclSuccess; }
ncclResult_t ncclCommUserRank(const ncclComm_t comm, int *rank) { *rank = reinterpret_cast<const Comm *>(comm)->rank; return ncclSuccess; }

ncclResult_t ncclGroupStart() { return ncclSuccess; }
ncclResult_t ncclGroupEnd() { return ncclSuccess; }

ncclResult_t ncclAllReduce(const void *sendbuff, void *recvbuff, size_t count, ncclDataType_t dt, ncclRedOp_t op, ncclComm_t comm,
                           hipStream_t stream)
{
    if (dt != ncclDouble || op != ncclSum) return ncclInvalidArgument;
    return enqueue(reinterpret_cast<Comm *>(comm), kAllReduce, sendbuff, recvbuff, count * 8, count, 0, stream);
}

ncclResult_t ncclAllGather(const void *sendbuff, void *recvbuff, size_t sendcount, ncclDataType_t dt, ncclComm_t comm, hipStream_t stream)
{
    Comm *c = reinterpret_cast<Comm *>(comm);
    const size_t bytes = sendcount * dsize(dt);
    if (bytes == 0) return ncclInvalidArgument;
    // what real RCCL requires of an overlapping call: in place means sendbuff == recvbuff + rank * count
    const char *s = static_cast<const char *>(sendbuff), *r = static_cast<const char *>(recvbuff);
    if (s + bytes > r && s < r + (size_t)c->w->nranks * bytes && s != r + (size_t)c->rank * bytes) {
        fprintf(stderr, "[mock rccl async] rank %d: overlapping all-gather with sendbuff != recvbuff + rank*count\n", c->rank);
        return ncclInvalidUsage;
    }
    return enqueue(c, kAllGather, sendbuff, recvbuff, bytes, sendcount, 0, stream);
}

ncclResult_t ncclBroadcast(const void *sendbuff, void *recvbuff, size_t count, ncclDataType_t dt, int root, ncclComm_t comm,
                           hipStream_t stream)
{
    Comm *c = reinterpret_cast<Comm *>(comm);
    if (root < 0 || root >= c->w->nranks) return ncclInvalidArgument;
    return enqueue(c, kBroadcast, sendbuff, recvbuff, count * dsize(dt), count, root, stream);
}

}  // extern "C"
