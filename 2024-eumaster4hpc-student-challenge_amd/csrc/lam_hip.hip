// lam_hip.hip -- context, orchestration and the C ABI (include/lam_hip.h) of the MI355X-native
// dense Conjugate-Gradient hot path.  Kernels: lam_kernels.h.  gfx950 only, no CPU fallback.
//
// Orchestration of the loop body of
//   /root/reference/challenge/main/LAM/src/CPU/ConjugateGradient_CPU_MPI_OMP.hpp:98-116
// per shard and iteration k:
//   gemv_coop_kernel -> [exchange p.Ap partials] -> update_xr_kernel -> [exchange r.r partials]
//   -> update_p_kernel (stores the new p slice into every replica of p) -> [all-gather p]
// With several shards a reducer workgroup inside the producer launch leaves the shard's partial as one
// double (lam_kernels.h, Finalize), so an iteration is 3 launches for every shard count.  "Exchange"
// is (a) nothing for one shard, (b) direct peer stores + cross-stream events when one process drives
// several shards (xGMI point-to-point), (c) RCCL when there is one process per GPU: an 8-byte-per-
// rank ncclAllGather for each dot product (summed in rank order by the consumer: deterministic and
// bit-identical to (b)) + ncclAllGather(p) on a second stream under the own-slice GEMV panel
// (exchange 0), or one ncclAllGather of [Ap slice | p.Ap partial] with full-length r/p per rank
// (exchange 1).  Scalars stay on the device; the host only reads the stopping iteration from pinned
// memory with a lag of kLag iterations, so the queue never drains and every rank enqueues the same
// number of collectives.  Option "symmetric" (one shard) replaces the GEMV by the two-pass
// upper-triangle product.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <type_traits>
#include <vector>

#include <sched.h>
#include <time.h>
#include <unistd.h>

#include "../../include/lam_hip.h"
#include "lam_kernels.h"

using namespace lam;

namespace {

thread_local std::string g_create_error;

double now_s()
{
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

uint64_t thread_cpu_ns()
{
    struct timespec ts;
    if (clock_gettime(CLOCK_THREAD_CPUTIME_ID, &ts) != 0) return 0;
    return (uint64_t)ts.tv_sec * 1000000000ull + (uint64_t)ts.tv_nsec;
}

constexpr int kLag = 4;          // iterations the host may run ahead of the stop flag
constexpr int kVecBlocksMax = 256;

struct ShardBase {
    int dev = 0;
    int index = 0;               // global shard index
    uint64_t row0 = 0, nrows = 0;
    hipStream_t stream = nullptr;
    void *A = nullptr;           // nrows x n
    size_t A_capacity = 0;       // bytes behind A: the allocation is kept across lam_hip_set_problem calls (grow-only)
    void *p = nullptr;           // n (replica)
    void *Ap = nullptr, *x = nullptr, *r = nullptr, *b = nullptr;  // nrows each
    void *tmp = nullptr;         // n: scratch vector (gemv op input / residual)
    void *r_full = nullptr;      // n: replicated r (gather-Ap exchange only)
    void *ap_gather = nullptr;   // P records [Ap slice | double]: gather-Ap exchange only.  One process with several shards:
                                 // TWO such buffers back to back (iteration parity), because there the producers store into
                                 // their peers' buffers themselves and a shard may start the next GEMV while a slower peer is
                                 // still reading this iteration's records (ap_gather_bytes = one buffer)
    size_t ap_gather_bytes = 0;
    void *symv_rowpart = nullptr, *symv_colpart = nullptr;   // symmetric product (option "symmetric")
    SymvTask *symv_tasks = nullptr;
    int symv_ntasks = 0;
    double *part_gemv = nullptr; // [gemv_blocks]
    double *part_vec = nullptr;  // [vec_blocks]
    double *gather_a = nullptr;  // [kMaxShards] p.Ap partials of all shards (or the reduced scalar at [0])
    double *gather_b = nullptr;  // [kMaxShards] r.r partials
    double *part_aux = nullptr;  // [kVecBlocksMax] partials of the checks outside the iteration (true residual)
    int part_gemv_cap = 0;       // entries allocated behind part_gemv
    CgScalars *sc = nullptr;     // device scalars
    CgScalars *sc_host = nullptr;// pinned mirror (filled by an async copy at the end of a call)
    int *host_flags = nullptr;   // pinned, device-visible progress word (lam_kernels.h, post_progress): low half = last
                                 // finished iteration, high half = the stopping iteration (0 = none)
    int gemv_blocks = 0, vec_blocks = 0;
    hipStream_t comm_stream = nullptr;                          // rank mode: the all-gather of p runs here
    hipEvent_t ev_a = nullptr, ev_b = nullptr, ev_p = nullptr;  // cross-shard ordering
    hipEvent_t ev_gathered = nullptr;                           // all-gather on comm_stream finished
    hipEvent_t ev_g0[kLag] = {}, ev_g1[kLag] = {};              // gemv timing ring (whole GEMV, or its first panel)
    hipEvent_t ev_g2[kLag] = {}, ev_g3[kLag] = {};              // second panel of a split GEMV
    bool split_slot[kLag] = {};
    bool timed_slot[kLag] = {};                                 // the slot's iteration recorded its timing events
    // in-launch hand-over / direct exchange (lam_kernels.h, Mail): the shard's mailbox (fine-grained device memory where
    // the runtime offers it), the broadcast lines of its fused update launch, and the iteration whose fused launch has
    // already waited for the peers' p slices.  Kept for the life of the context.
    Mail *mail = nullptr;
    bool mail_coarse = false;
    BcastLine *bcast = nullptr;
    int waited_k = 0;
};

}  // namespace

struct lam_hip_ctx {
    int dtype = LAM_HIP_F64;
    int total_shards = 1;          // P
    int rank = 0, nranks = 1;      // rank mode (one local shard == shard `rank`)
    bool rank_mode = false;
    ncclComm_t comm = nullptr;
    double t_comm_init = 0.0;
    uint64_t n = 0;
    bool have_problem = false, have_matrix = false, have_rhs = false, cg_ready = false;
    int k_done = 0;                // CG iterations enqueued since cg_init
    std::vector<ShardBase> sh;     // local shards
    std::string err;
    // options
    int64_t opt_gemv_variant = -1; // -1 = production shape for the dtype (see Impl::variant)
    int64_t opt_nt = 1;
    int64_t opt_generic = 0;       // force the generic kernel
    int64_t opt_probe_rows = 0;    // gemv_only: use only the first probe_rows rows of each shard (0 = all)
    int64_t opt_overlap = 1;       // rank mode: all-gather on its own stream under the own-slice GEMV panel
    int64_t opt_panel_lo = 0, opt_panel_hi = 0;  // testing: split the CG GEMV into [lo,hi) + the rest
    bool gather_pending = false;   // an all-gather of p is in flight on comm_stream
    int64_t opt_symmetric = 0;     // single shard: read only the upper triangle (caller asserts A == A^T)
    int64_t opt_exchange = 0;      // 0 = sliced vectors, three exchanges per iteration (p.Ap, r.r, p slices); 1 = gather-Ap: ONE
                                   // exchange of [Ap slice | p.Ap partial] per iteration, r and p full-length on every shard;
                                   // 2 = direct (in-kernel flags)
    int64_t opt_join = 1;          // one process, gather-Ap: 1 = the iteration's single join goes through shard 0's stream (it
                                   // waits for the other shards' posts and records ONE join event they wait for: 2(P-1)+1
                                   // runtime calls); 0 = every stream waits for every other one (P(P-1) calls)
    int64_t opt_fuse = 1;          // one shard / direct exchange: x, r and p updates in ONE launch (update_fused_kernel)
    int64_t opt_reuse_matrix = 1;  // lam_hip_set_problem keeps (and re-uses) the matrix allocation when it is large enough
    int64_t opt_upload_staging = 0; // lam_hip_upload_rows: 1 = pipeline through two pinned staging buffers
    int64_t opt_finalize = 1;      // several shards: 1 = producer kernels reduce their partials themselves (Finalize);
                                   // 0 = separate 1-block finalize_sum_kernel launches (A/B measurements)
    uint64_t n_collectives = 0;    // RCCL calls enqueued by this context (diagnostics: must match across ranks)
    int64_t opt_gemv_timing = 8;   // HIP-event pair around the GEMV of every T-th iteration (t_gemv of the stats); 0 = never.
                                   // Every record is a marker packet between the kernels: timing every iteration costs 8 us
                                   // per iteration (profiles/r03_event_cost.txt)
    // One process, several shards: how the host orders the shards' streams (profiles/r03_host_enqueue_cost.txt).  Every
    // cross-stream event operation costs the host 3-5 us, so the event-based forms all stay above 0.2 ms per iteration
    // at P = 8; the form that does not (no events at all) is the in-kernel flag exchange, option "exchange" = 2.
    int64_t opt_host_threads = 0;  // 1 = every shard is enqueued by a host thread of its own (the reference's shape)
    int64_t opt_hub = 0;           // 1 = the shards' streams meet at ONE join event per exchange (P waits on a hub stream +
                                   // P waits on its event: 3(3P+1) calls per iteration) instead of every stream waiting
                                   // for every other one (3P^2 calls); fewer host calls, one more event hop on the device
    hipStream_t hub_stream = nullptr;          // on shard 0's device
    hipEvent_t ev_join[3] = {};                // p.Ap partials posted / r.r partials posted / p slices stored
    int64_t opt_assume_cus = 0;    // testing: pretend the device has this many CUs when checking that a launch whose
                                   // workgroups wait for each other is fully resident (0 = ask the device)
    bool fuse_active = false;      // the current CG state uses update_fused_kernel (decided in cg_init: option + residency)
    // whole-iteration persistent launch (lam_kernels.h, cg_persist_kernel): EXPERIMENT, option "persistent", off by default
    int64_t opt_persistent = 0;
    int64_t opt_persist_chunk = 32;             // iterations per launch
    bool persist_active = false;                // the current CG state runs on it (decided in cg_init)
    int persist_W = 0;                          // worker workgroups (+ 1 reducer)
    BcastLine *persist_bc = nullptr;            // [kVecBlocksMax + persist lines] broadcast lines (device memory)
    unsigned long long *persist_ticks = nullptr;        // device: [0] GEMV-phase ticks (100 MHz), [1] phases
    unsigned long long *persist_ticks_host = nullptr;   // pinned mirror
    // runtime calls issued by the iteration loop (diagnostics: host cost of an iteration, tools/host_enqueue_cost.py)
    std::atomic<uint64_t> n_launch{0}, n_record{0}, n_wait{0}, n_setdev{0};
    uint64_t enqueue_ns = 0;       // host time spent issuing iterations (the waits for the device's progress excluded)
    uint64_t host_cpu_ns = 0;      // CPU time (CLOCK_THREAD_CPUTIME_ID) the calling thread spent inside lam_hip_cg_iterate
    double iter_est_s = 0.0;       // observed seconds per iteration (await_progress sleeps a fraction of it between polls)
    double prog_t = 0.0;           // when / at which iteration the last awaited progress was seen
    int prog_iter = 0;
    std::mutex err_mu;             // `err` may be written by the per-shard enqueue threads
    // direct exchange (option exchange = 2): peer-mapped p replicas and mailboxes (lam_kernels.h, Mail)
    Mail *peer_mail[kMaxShards] = {};           // every shard's mailbox as seen from this process (own included)
    void *peer_p[kMaxShards] = {};              // every shard's p replica as seen from this process (own included)
    void *ipc_opened[2 * kMaxShards] = {};      // mappings to close again
    int n_ipc_opened = 0;
    uint64_t problem_gen = 0, direct_gen = ~0ull;   // direct mappings belong to one set_problem generation
    bool direct_ok = false;
    bool cg_direct = false;                     // the current CG state runs on the direct exchange
    int64_t opt_verify_direct = 1;              // lam_hip_solve on the direct exchange: compare the recomputed residual with
                                                // the recursive one afterwards; on a mismatch solve again on the RCCL exchange
    int64_t direct_fallbacks = 0;               // how often that happened
    uint32_t epoch = 0;                         // bumped by every cg_init
    uint64_t seq_base = 0, seq_span = 1;        // hand-over number of iteration k of the current solve = seq_base + k: grows by one
                                                // per iteration over the whole life of the context and never restarts (cg_init moves
                                                // the base past every iteration of the previous solve), so the 32-bit tags of the
                                                // in-kernel hand-overs (lam_kernels.h, MailSlot) cannot meet a stale equal; the same
                                                // on every rank (all ranks enqueue the same iterations)
    int *direct_err = nullptr;                  // pinned host: a bounded in-kernel wait expired ([0] = which, see cg_iterate)
    double *agree_buf = nullptr;                // 4 KiB device scratch of the small set-up collectives (kept: no hipFree in them)
    bool cg_exchange1 = false;     // the exchange the current CG state was initialised for

    // the symmetric product exists for one shard, fp64/fp32 storage, n a multiple of its column tile
    uint64_t symv_tile() const { return 8ull * kBlock * (16 / esz_a()); }
    bool symv_active() const
    {
        return opt_symmetric && !rank_mode && total_shards == 1 && dtype != LAM_HIP_BF16 && n > 0 && n % symv_tile() == 0;
    }

    bool exchange2_wanted() const { return (rank_mode || total_shards > 1) && opt_exchange == 2 && opt_finalize != 0; }
    // gather-Ap needs equal slices and an 8-byte aligned tail for the double (total_shards == nranks in rank mode)
    bool exchange1_ok() const
    {
        return (rank_mode || total_shards > 1) && opt_exchange == 1 && n % (uint64_t)total_shards == 0 &&
               ((n / (uint64_t)total_shards) * esz_v()) % 8 == 0;
    }
    uint64_t ex1_base() const { return n / (uint64_t)total_shards; }
    uint64_t ex1_stride_bytes() const { return ex1_base() * esz_v() + 8; }

    size_t esz_a() const { return dtype == LAM_HIP_F64 ? 8 : (dtype == LAM_HIP_F32 ? 4 : 2); }
    size_t esz_v() const { return dtype == LAM_HIP_F64 ? 8 : 4; }
};

namespace {

int fail(lam_hip_ctx *c, int code, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (c) { std::lock_guard<std::mutex> lk(c->err_mu); c->err = buf; }
    else g_create_error = buf;
    return code;
}

#define HIPCHK(c, call)                                                                       \
    do {                                                                                      \
        hipError_t e_ = (call);                                                               \
        if (e_ != hipSuccess)                                                                 \
            return fail((c), e_ == hipErrorOutOfMemory ? LAM_HIP_ENOMEM : LAM_HIP_EHIP,       \
                        "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

#define NCCLCHK(c, call)                                                                      \
    do {                                                                                      \
        ncclResult_t r_ = (call);                                                             \
        if (r_ != ncclSuccess)                                                                \
            return fail((c), LAM_HIP_ERCCL, "%s failed: %s (%s:%d)", #call,                   \
                        ncclGetErrorString(r_), __FILE__, __LINE__);                          \
    } while (0)

#define LAMCHK(expr)                 \
    do {                             \
        int rc_ = (expr);            \
        if (rc_ != 0) return rc_;    \
    } while (0)

// the iteration loop's runtime calls, counted (option "hip_calls_*"): what an iteration costs the host
#define LAUNCHED(c)                                  \
    do {                                             \
        (c)->n_launch++;                             \
        HIPCHK((c), hipGetLastError());              \
    } while (0)
#define RECORD(c, ev, st)                            \
    do {                                             \
        (c)->n_record++;                             \
        HIPCHK((c), hipEventRecord((ev), (st)));     \
    } while (0)
#define WAITEV(c, st, ev)                            \
    do {                                             \
        (c)->n_wait++;                               \
        HIPCHK((c), hipStreamWaitEvent((st), (ev), 0)); \
    } while (0)

void partition(uint64_t n, int P, int q, uint64_t *row0, uint64_t *nrows)
{
    // ConjugateGradient_CPU_MPI_OMP.hpp:176-184: n/P rows each, the remainder on the LAST rank
    const uint64_t base = n / (uint64_t)P;
    *row0 = base * (uint64_t)q;
    *nrows = base + ((q == P - 1) ? n % (uint64_t)P : 0);
}

// OPT-IN (environment LAM_HIP_QUIET_RCCL=1, set by this package's drivers, whose stdout is a one-line protocol): file
// descriptor 1 points at stderr while at least one of these exists, i.e. for the duration of ncclCommInitRank, which
// prints a version banner to stdout.  A library must not move a host application's stdout around by default, so
// without the variable nothing is touched.  Counted under a lock: contexts may be created from several threads at
// once (the ranks-as-threads test double), and the first one in must be the one that remembers the real stdout, the
// last one out the one that restores it.
struct StdoutToStderr {
    static bool wanted()
    {
        const char *q = getenv("LAM_HIP_QUIET_RCCL");
        return q && *q && strcmp(q, "0") != 0;
    }
    const bool on = wanted();
    static std::mutex &mu() { static std::mutex m; return m; }
    static int &depth() { static int d = 0; return d; }
    static int &saved() { static int fd = -1; return fd; }
    StdoutToStderr()
    {
        if (!on) return;
        std::lock_guard<std::mutex> lk(mu());
        if (depth()++ == 0) {
            fflush(stdout);
            saved() = dup(1);
            if (saved() >= 0) (void)dup2(2, 1);
        }
    }
    ~StdoutToStderr()
    {
        if (!on) return;
        std::lock_guard<std::mutex> lk(mu());
        if (--depth() == 0 && saved() >= 0) {
            fflush(stdout);
            (void)dup2(saved(), 1);
            (void)close(saved());
            saved() = -1;
        }
    }
};

// temporaries of one call: released on every exit path (HIPCHK returns from the middle of a function)
struct DevBuf {
    void *p = nullptr;
    ~DevBuf() { if (p) (void)hipFree(p); }
    template <typename T> T *as() const { return static_cast<T *>(p); }
};
struct PinnedBuf {
    void *p = nullptr;
    ~PinnedBuf() { if (p) (void)hipHostFree(p); }
};

int vec_grid(uint64_t n_loc)
{
    uint64_t b = (n_loc + kBlock - 1) / kBlock;
    return (int)std::max<uint64_t>(1, std::min<uint64_t>(b, kVecBlocksMax));
}

// ---- typed implementation ----------------------------------------------------------------------
template <typename TA, typename TV>
struct Impl {
    static constexpr int VEC = MatVec<TA>::N;

    // GEMV shapes.  Variants 0-8: gemv_tile_kernel {rows per wave, p-tile columns, p in LDS, rotated
    // tile order}; 9-18: gemv_coop_kernel {rows per workgroup, tile, waves}; 19-22: the MFMA experiment (bf16).
    // The PRODUCT library holds the shapes that are some dtype's default: 10 (cooperative rows, 2 rows per 4-wave
    // workgroup: fp64/fp32 production, fastest at N=65536 and N=32768, profiles/r01_gemv_variant_sweep.txt) and 0 (4 rows
    // per wave: bf16 production).  Everything else -- the other tile / cooperative shapes, the grouped probe and the
    // MFMA-fed bf16 GEMV of BASELINE configs[3]'s comparison (slower than the VALU kernel) -- exists only in the library
    // built with -DLAM_TUNING_VARIANTS (`make tuning` -> liblam_hip_tuning.so; tools/gemv_probe.py, bench.py's MFMA child).
    static constexpr int kNumVariants = 25;      // 23, 24: tuning probes gemv_coop_group_kernel (2 / 4 row pairs per workgroup)
    static bool variant_available(int v)
    {
#ifdef LAM_TUNING_VARIANTS
        return v >= 0 && v < kNumVariants;
#else
        return v == 0 || v == 10;
#endif
    }
    // rows per WORKGROUP of each variant (variants 9.. are the cooperative-row shape: R rows per workgroup)
    static int variant_rows_per_block(int v)
    {
        static const int rows[kNumVariants] = {16, 8, 32, 16, 16, 8, 16, 16, 4, 1, 2, 4, 8, 2, 2, 2, 2, 4, 3,
                                                  /* 19-22: MFMA bf16 experiment (bf16 storage only) */ 8, 8, 16, 4,
                                                  /* 23, 24: grouped cooperative rows (tuning probe) */ 4, 8};
        return rows[v];
    }

    static bool fast_ok(const lam_hip_ctx *c) { return !c->opt_generic && (c->n % VEC) == 0; }
    static int variant(const lam_hip_ctx *c)
    {
        if (c->opt_gemv_variant >= 0 && variant_available((int)c->opt_gemv_variant)) return (int)c->opt_gemv_variant;
        // production shapes: fp64/fp32 -> cooperative rows (variant 10); bf16 storage spends more VALU
        // per byte (widening) and measures best with 4 rows per wave (variant 0, 6.77 vs 6.40 TB/s)
        return sizeof(TA) == 2 ? 0 : 10;
    }

    // name of the kernel instantiation launch_gemv() picks for this context (roofline records)
    static std::string kernel_name(const lam_hip_ctx *c)
    {
        const char *ta = sizeof(TA) == 8 ? "double" : (sizeof(TA) == 4 ? "float" : "__hip_bfloat16");
        const char *tv = sizeof(TV) == 8 ? "double" : "float";
        char buf[192];
        if (c->symv_active()) { snprintf(buf, sizeof buf, "symv_task_kernel<%s> + symv_reduce_kernel<%s>", ta, ta); return buf; }
        if (!fast_ok(c)) { snprintf(buf, sizeof buf, "gemv_generic_kernel<%s,%s>", ta, tv); return buf; }
        const int v = variant(c);
        const char *nt = c->opt_nt ? "true" : "false";
        struct Tile { int r, tile; bool lds, rot; };
        static const Tile tiles[9] = {{4, 4096, true, true}, {2, 4096, true, true}, {8, 4096, true, true}, {4, 2048, true, true},
                                      {4, 8192, true, true}, {2, 8192, true, true}, {4, 4096, false, true}, {4, 4096, true, false},
                                      {1, 4096, true, true}};
        struct Coop { int r, tile, waves, unroll; };
        static const Coop coops[10] = {{1, 4096, 4, 4}, {2, 4096, 4, 4}, {4, 4096, 4, 4}, {8, 4096, 4, 4}, {2, 4096, 8, 4},
                                       {2, 2048, 4, 4}, {2, 8192, 8, 8}, {2, 8192, 4, 8}, {4, 4096, 8, 4}, {3, 4096, 4, 4}};
        if (v <= 8)
            snprintf(buf, sizeof buf, "gemv_tile_kernel<%s,%s,R=%d,TILE=%d,NT=%s,UNROLL=4,LDS=%s,ROT=%s>", ta, tv, tiles[v].r,
                     tiles[v].tile, nt, tiles[v].lds ? "true" : "false", tiles[v].rot ? "true" : "false");
        else if (v <= 18)
            snprintf(buf, sizeof buf, "gemv_coop_kernel<%s,%s,R=%d,TILE=%d,NT=%s,UNROLL=%d,WAVES=%d>", ta, tv, coops[v - 9].r,
                     coops[v - 9].tile, nt, coops[v - 9].unroll, coops[v - 9].waves);
        else if (v >= 23)
            snprintf(buf, sizeof buf, "gemv_coop_group_kernel<%s,%s,GROUP=%d>", ta, tv, v == 23 ? 2 : 4);
        else {
            static const int mf[4][2] = {{2, 3}, {2, 1}, {4, 3}, {1, 3}};      // {R, SPLIT} of variants 19..22
            snprintf(buf, sizeof buf, "gemv_mfma_bf16_kernel<R=%d,TILE=4096,NT=true,SPLIT=%d>", mf[v - 19][0], mf[v - 19][1]);
        }
        return buf;
    }

    // number of p.Ap partials the product step of a CG iteration leaves in part_gemv
    static int gemv_grid(const lam_hip_ctx *c, uint64_t nrows)
    {
        if (nrows == 0) return 0;
        if (c->symv_active()) return (int)(c->n / kSymvRows);     // symmetric product: one per 32-row block
        return kernel_grid(c, nrows);
    }

    // workgroups of the general GEMV kernel (also used on its own by the residual check)
    static int kernel_grid(const lam_hip_ctx *c, uint64_t nrows)
    {
        if (nrows == 0) return 0;
        const uint64_t rows_per_block = fast_ok(c) ? (uint64_t)variant_rows_per_block(variant(c)) : (uint64_t)kWaves;
        return (int)((nrows + rows_per_block - 1) / rows_per_block);
    }

    template <int R, int TILE, bool LDS, bool ROT>
    static void launch_tile(const lam_hip_ctx *c, int grid, hipStream_t st, const GemvArgs<TA, TV> &a)
    {
        if (c->opt_nt)
            hipLaunchKernelGGL((gemv_tile_kernel<TA, TV, R, TILE, true, 4, LDS, ROT>), dim3(grid), dim3(kBlock), 0, st, a);
        else
            hipLaunchKernelGGL((gemv_tile_kernel<TA, TV, R, TILE, false, 4, LDS, ROT>), dim3(grid), dim3(kBlock), 0, st, a);
    }

    // y = A p from the upper triangle only (lam_kernels.h, "Symmetric product")
    static int launch_symv(lam_hip_ctx *c, ShardBase &s, const TV *p, TV *y, double *partial, const CgScalars *sc)
    {
        if constexpr (std::is_same<TA, TV>::value) {
            const uint64_t n = c->n;
            const uint32_t ntiles = (uint32_t)(n / SymvShape<TA>::TILE), nblk = (uint32_t)(n / kSymvRows);
            if (s.symv_tasks == nullptr) {
                std::vector<SymvTask> tasks;
                for (uint32_t I = 0; I < nblk; I++)
                    for (uint32_t j = (uint32_t)(((uint64_t)I * kSymvRows) / SymvShape<TA>::TILE); j < ntiles; j++) tasks.push_back({I, j});
                // all three or none: a later failure must not leave the earlier buffers behind
                DevBuf t, rp, cp;
                HIPCHK(c, hipMalloc(&t.p, tasks.size() * sizeof(SymvTask)));
                HIPCHK(c, hipMalloc(&rp.p, (size_t)nblk * ntiles * kSymvRows * sizeof(TA)));
                HIPCHK(c, hipMalloc(&cp.p, (size_t)nblk * n * sizeof(TA)));
                HIPCHK(c, hipMemcpy(t.p, tasks.data(), tasks.size() * sizeof(SymvTask), hipMemcpyHostToDevice));
                HIPCHK(c, hipMemsetAsync(cp.p, 0, (size_t)nblk * n * sizeof(TA), s.stream));
                s.symv_tasks = t.as<SymvTask>(); s.symv_rowpart = rp.p; s.symv_colpart = cp.p;
                t.p = rp.p = cp.p = nullptr;
                s.symv_ntasks = (int)tasks.size();
            }
            hipLaunchKernelGGL((symv_task_kernel<TA>), dim3(s.symv_ntasks), dim3(kBlock), 0, s.stream, (const TA *)s.A, (const TA *)p,
                               (const SymvTask *)s.symv_tasks, (TA *)s.symv_rowpart, (TA *)s.symv_colpart, n, ntiles, sc);
            HIPCHK(c, hipGetLastError());
            hipLaunchKernelGGL((symv_reduce_kernel<TA>), dim3(nblk), dim3(kBlock), 0, s.stream, (const TA *)s.symv_rowpart,
                               (const TA *)s.symv_colpart, (const TA *)p, (TA *)y, partial, n, ntiles, sc);
            HIPCHK(c, hipGetLastError());
            c->n_launch += 2;
            return 0;
        } else {
            return fail(c, LAM_HIP_EINVAL, "the symmetric product needs matrix and vector of one type");
        }
    }

    // panel: 0 = whole GEMV; 1 = only columns [lo,hi); 2 = everything but [lo,hi), accumulated onto y
    template <int R, int TILE = 4096, int WAVES = 4, int UNROLL = 4>
    static void launch_coop(const lam_hip_ctx *c, int grid, hipStream_t st, const GemvArgs<TA, TV> &a)
    {
        if (c->opt_nt)
            hipLaunchKernelGGL((gemv_coop_kernel<TA, TV, R, TILE, true, UNROLL, WAVES>), dim3(grid), dim3(WAVES * 64), 0, st, a);
        else
            hipLaunchKernelGGL((gemv_coop_kernel<TA, TV, R, TILE, false, UNROLL, WAVES>), dim3(grid), dim3(WAVES * 64), 0, st, a);
    }

    static int launch_gemv(lam_hip_ctx *c, ShardBase &s, const TV *p, TV *y, double *partial, const CgScalars *sc,
                           int panel = 0, uint64_t lo = 0, uint64_t hi = 0, const Finalize *fin = nullptr, const PtrList *ypeers = nullptr)
    {
        if (s.nrows == 0) return 0;
        GemvArgs<TA, TV> a;
        a.A = (const TA *)s.A; a.p = p; a.y = y; a.partial = partial; a.sc = sc;
        a.n_ypeer = 0;
        for (auto &yp : a.ypeer) yp = nullptr;
        if (ypeers != nullptr)
            for (int j = 0; j < ypeers->n && a.n_ypeer < kMaxShards - 1; j++) a.ypeer[a.n_ypeer++] = (TV *)ypeers->p[j];
        if (fin != nullptr && partial != nullptr) a.fin = *fin;
        else { a.fin.active = 0; a.fin.mail = 0; a.fin.seq = 0; a.fin.dst.n = 0; a.fin.slot = 0; a.fin.host_err = c->direct_err; }
        a.nrows = s.nrows; a.n = c->n; a.row0 = s.row0;
        a.seg_begin[0] = 0; a.seg_end[0] = c->n; a.seg_begin[1] = a.seg_end[1] = 0; a.nseg = 1; a.accumulate = 0;
        if (panel == 1) { a.seg_begin[0] = lo; a.seg_end[0] = hi; }
        else if (panel == 2) {
            a.accumulate = 1;
            a.nseg = 0;
            if (lo > 0) { a.seg_begin[a.nseg] = 0; a.seg_end[a.nseg] = lo; a.nseg++; }
            if (hi < c->n) { a.seg_begin[a.nseg] = hi; a.seg_end[a.nseg] = c->n; a.nseg++; }
            if (a.nseg == 0) return 0;
            if (a.nseg == 1) { a.seg_begin[1] = a.seg_end[1] = 0; }
        }
        const int grid = kernel_grid(c, s.nrows) + (a.fin.active ? 1 : 0);     // + the reducer workgroup (Finalize)
        if (fast_ok(c)) {
            switch (variant(c)) {
            default:
            case 0: launch_tile<4, 4096, true, true>(c, grid, s.stream, a); break;
            case 10: launch_coop<2>(c, grid, s.stream, a); break;
#ifdef LAM_TUNING_VARIANTS
            case 1: launch_tile<2, 4096, true, true>(c, grid, s.stream, a); break;
            case 2: launch_tile<8, 4096, true, true>(c, grid, s.stream, a); break;
            case 3: launch_tile<4, 2048, true, true>(c, grid, s.stream, a); break;
            case 4: launch_tile<4, 8192, true, true>(c, grid, s.stream, a); break;
            case 5: launch_tile<2, 8192, true, true>(c, grid, s.stream, a); break;
            case 6: launch_tile<4, 4096, false, true>(c, grid, s.stream, a); break;
            case 7: launch_tile<4, 4096, true, false>(c, grid, s.stream, a); break;
            case 8: launch_tile<1, 4096, true, true>(c, grid, s.stream, a); break;
            case 9: launch_coop<1>(c, grid, s.stream, a); break;
            case 11: launch_coop<4>(c, grid, s.stream, a); break;
            case 12: launch_coop<8>(c, grid, s.stream, a); break;
            case 13: launch_coop<2, 4096, 8>(c, grid, s.stream, a); break;
            case 14: launch_coop<2, 2048, 4, 4>(c, grid, s.stream, a); break;
            case 15: launch_coop<2, 8192, 8, 8>(c, grid, s.stream, a); break;
            case 16: launch_coop<2, 8192, 4, 8>(c, grid, s.stream, a); break;
            case 17: launch_coop<4, 4096, 8>(c, grid, s.stream, a); break;
            case 18: launch_coop<3>(c, grid, s.stream, a); break;
            case 23: hipLaunchKernelGGL((gemv_coop_group_kernel<TA, TV, 2>), dim3(grid), dim3(kBlock), 0, s.stream, a); break;
            case 24: hipLaunchKernelGGL((gemv_coop_group_kernel<TA, TV, 4>), dim3(grid), dim3(kBlock), 0, s.stream, a); break;
            case 19: case 20: case 21: case 22:
                if constexpr (sizeof(TA) == 2) {
                    const int v = variant(c);
                    if (v == 20) hipLaunchKernelGGL((gemv_mfma_bf16_kernel<2, 4096, true, 1>), dim3(grid), dim3(kBlock), 0, s.stream, a);
                    else if (v == 21) hipLaunchKernelGGL((gemv_mfma_bf16_kernel<4, 4096, true, 3>), dim3(grid), dim3(kBlock), 0, s.stream, a);
                    else if (v == 19) hipLaunchKernelGGL((gemv_mfma_bf16_kernel<2, 4096, true, 3>), dim3(grid), dim3(kBlock), 0, s.stream, a);
                    else hipLaunchKernelGGL((gemv_mfma_bf16_kernel<1, 4096, true, 3>), dim3(grid), dim3(kBlock), 0, s.stream, a);
                } else {
                    return fail(c, LAM_HIP_EINVAL, "gemv_variant 19-22 (MFMA) exist for LAM_HIP_BF16 only");
                }
                break;
#endif
            }
        } else {
            hipLaunchKernelGGL((gemv_generic_kernel<TA, TV>), dim3(grid), dim3(kBlock), 0, s.stream, a);
        }
        LAUNCHED(c);
        return 0;
    }
};

// Own-slice panel [lo,hi) of the CG GEMV, or lo == hi when the GEMV stays one launch.  Panels need
// 16-byte aligned segment starts (lo, hi multiples of the vector width) unless the generic kernel runs.
template <typename I>
void cg_panel(const lam_hip_ctx *c, const ShardBase &s, uint64_t *lo, uint64_t *hi)
{
    *lo = *hi = 0;
    uint64_t a = 0, b = 0;
    if (c->opt_panel_hi > c->opt_panel_lo) { a = (uint64_t)c->opt_panel_lo; b = std::min<uint64_t>((uint64_t)c->opt_panel_hi, c->n); }
    else if (c->rank_mode && c->opt_overlap && c->nranks > 1) { a = s.row0; b = s.row0 + s.nrows; }
    if (b <= a || (a == 0 && b >= c->n)) return;
    if (I::fast_ok(c) && (a % I::VEC != 0 || b % I::VEC != 0)) return;
    *lo = a; *hi = b;
}

template <typename F>
int dispatch(lam_hip_ctx *c, F &&f)
{
    switch (c->dtype) {
    case LAM_HIP_F64: return f(Impl<double, double>());
    case LAM_HIP_F32: return f(Impl<float, float>());
    case LAM_HIP_BF16: return f(Impl<__hip_bfloat16, float>());
    }
    return fail(c, LAM_HIP_EINVAL, "bad dtype %d", c->dtype);
}

int set_dev(lam_hip_ctx *c, const ShardBase &s)
{
    c->n_setdev++;
    HIPCHK(c, hipSetDevice(s.dev));
    // hipGetLastError() is only used to pick up launch failures right after a launch; drop whatever an
    // earlier, already reported failure (possibly of another context) left in the thread's error slot
    (void)hipGetLastError();
    return 0;
}

PtrList plist_p(lam_hip_ctx *c)
{
    PtrList l;
    l.n = (int)c->sh.size();
    for (int j = 0; j < l.n; j++) l.p[j] = c->sh[j].p;
    return l;
}
PtrList plist_gather(lam_hip_ctx *c, bool second)
{
    PtrList l;
    l.n = (int)c->sh.size();
    for (int j = 0; j < l.n; j++) l.p[j] = second ? (void *)c->sh[j].gather_b : (void *)c->sh[j].gather_a;
    return l;
}

// keep_matrix: leave the matrix allocation alone (lam_hip_set_problem re-uses it when it is large enough)
void free_shard(ShardBase &s, bool keep_matrix = false)
{
    if (hipSetDevice(s.dev) != hipSuccess) { (void)hipGetLastError(); return; }   // never created on a real device
    void *const keepA = keep_matrix ? s.A : nullptr;
    const size_t keepCap = keep_matrix ? s.A_capacity : 0;
    if (keep_matrix) s.A = nullptr;
    void *ptrs[] = {s.A, s.p, s.Ap, s.x, s.r, s.b, s.tmp, s.part_gemv, s.part_vec, s.gather_a, s.gather_b, s.sc,
                    s.r_full, s.ap_gather, s.symv_rowpart, s.symv_colpart, s.symv_tasks, s.part_aux};
    for (void *q : ptrs) if (q) (void)hipFree(q);
    s.r_full = s.ap_gather = s.symv_rowpart = s.symv_colpart = nullptr;
    s.symv_tasks = nullptr;
    s.symv_ntasks = 0;
    if (s.sc_host) (void)hipHostFree(s.sc_host);
    if (s.host_flags) (void)hipHostFree(s.host_flags);
    s.host_flags = nullptr;
    s.A = s.p = s.Ap = s.x = s.r = s.b = s.tmp = nullptr;
    s.A = keepA;
    s.A_capacity = keepCap;
    s.part_gemv = s.part_vec = s.gather_a = s.gather_b = nullptr;
    s.part_aux = nullptr;
    s.sc = nullptr; s.sc_host = nullptr;
}

// streams and events of one shard (device memory is released by free_shard)
void release_handles(ShardBase &s)
{
    if (hipSetDevice(s.dev) != hipSuccess) { (void)hipGetLastError(); return; }
    hipEvent_t *evs[] = {&s.ev_a, &s.ev_b, &s.ev_p, &s.ev_gathered};
    for (auto e : evs) if (*e) { (void)hipEventDestroy(*e); *e = nullptr; }
    for (int i = 0; i < kLag; i++) {
        hipEvent_t *ring[] = {&s.ev_g0[i], &s.ev_g1[i], &s.ev_g2[i], &s.ev_g3[i]};
        for (auto e : ring) if (*e) { (void)hipEventDestroy(*e); *e = nullptr; }
    }
    if (s.comm_stream) { (void)hipStreamSynchronize(s.comm_stream); (void)hipStreamDestroy(s.comm_stream); s.comm_stream = nullptr; }
    if (s.stream) { (void)hipStreamSynchronize(s.stream); (void)hipStreamDestroy(s.stream); s.stream = nullptr; }
}

// a context whose creation failed half-way: give back what it already holds
void release_hub(lam_hip_ctx *c)
{
    if (c->sh.empty() || hipSetDevice(c->sh[0].dev) != hipSuccess) { (void)hipGetLastError(); return; }
    for (auto &ev : c->ev_join) if (ev) { (void)hipEventDestroy(ev); ev = nullptr; }
    if (c->hub_stream) { (void)hipStreamSynchronize(c->hub_stream); (void)hipStreamDestroy(c->hub_stream); c->hub_stream = nullptr; }
}

void abandon(lam_hip_ctx *c)
{
    release_hub(c);
    for (auto &s : c->sh) { free_shard(s); release_handles(s); }
    if (c->direct_err) { (void)hipHostFree(c->direct_err); c->direct_err = nullptr; }
}

// Environment LAM_HIP_EXCHANGE = default exchange of new contexts (drivers have no other way to choose one).  The direct
// exchange (2) is EXPERIMENTAL -- never yet run on separate GPUs -- and lam_hip_cg_init + lam_hip_cg_iterate do not check
// themselves the way lam_hip_solve does, so the environment alone must not make it anybody's default: it is honoured only
// together with LAM_HIP_EXPERIMENTAL_DIRECT=1; lam_hip_set_option("exchange", 2) stays the explicit opt-in.
int64_t exchange_from_env(int64_t dflt)
{
    const char *ex = getenv("LAM_HIP_EXCHANGE");
    if (ex == nullptr || *ex == '\0') return dflt;
    const int v = atoi(ex);
    if (v == 2) {
        const char *ok = getenv("LAM_HIP_EXPERIMENTAL_DIRECT");
        if (!(ok && *ok && strcmp(ok, "0") != 0)) {
            static std::atomic<bool> told{false};
            if (!told.exchange(true))
                fprintf(stderr, "lam_hip: LAM_HIP_EXCHANGE=2 (direct exchange, experimental) ignored: set LAM_HIP_EXPERIMENTAL_DIRECT=1 as well\n");
            return dflt;
        }
    }
    return (v >= 0 && v <= 2) ? v : dflt;
}

int create_common(lam_hip_ctx *c)
{
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0)
        return fail(nullptr, LAM_HIP_ENODEV, "no usable HIP device (%s); this library has no CPU path",
                    e == hipSuccess ? "device count is 0" : hipGetErrorString(e));
    for (auto &s : c->sh) {
        if (s.dev < 0 || s.dev >= ndev) return fail(nullptr, LAM_HIP_EINVAL, "device id %d out of range (have %d)", s.dev, ndev);
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, s.dev) != hipSuccess)
            return fail(nullptr, LAM_HIP_EHIP, "hipGetDeviceProperties(%d) failed", s.dev);
        if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
            return fail(nullptr, LAM_HIP_ENODEV, "device %d is %s; this library is built for gfx950 (MI355X) only", s.dev, prop.gcnArchName);
        if (hipSetDevice(s.dev) != hipSuccess) return fail(nullptr, LAM_HIP_EHIP, "hipSetDevice(%d) failed", s.dev);
        if (hipStreamCreateWithFlags(&s.stream, hipStreamNonBlocking) != hipSuccess ||
            hipStreamCreateWithFlags(&s.comm_stream, hipStreamNonBlocking) != hipSuccess)
            return fail(nullptr, LAM_HIP_EHIP, "hipStreamCreate failed on device %d", s.dev);
        // cross-shard hand-over events: SYSTEM-scope release, so that a shard's stores into a peer device's
        // p replica / gather array have left its L2 when the peer's stream passes the event
        hipEvent_t *evs[] = {&s.ev_a, &s.ev_b, &s.ev_p, &s.ev_gathered};
        for (auto ev : evs)
            if (hipEventCreateWithFlags(ev, hipEventDisableTiming | hipEventReleaseToSystem) != hipSuccess)
                return fail(nullptr, LAM_HIP_EHIP, "hipEventCreate failed");
        for (int i = 0; i < kLag; i++) {
            if (hipEventCreate(&s.ev_g0[i]) != hipSuccess || hipEventCreate(&s.ev_g1[i]) != hipSuccess ||
                hipEventCreate(&s.ev_g2[i]) != hipSuccess || hipEventCreate(&s.ev_g3[i]) != hipSuccess)
                return fail(nullptr, LAM_HIP_EHIP, "hipEventCreate failed");
        }
    }
    // hub of the one-process exchange (see hub_join): a stream on shard 0's device and one join event per exchange
    if (!c->rank_mode && c->sh.size() > 1) {
        if (hipSetDevice(c->sh[0].dev) != hipSuccess || hipStreamCreateWithFlags(&c->hub_stream, hipStreamNonBlocking) != hipSuccess)
            return fail(nullptr, LAM_HIP_EHIP, "hipStreamCreate (hub) failed");
        for (auto &ev : c->ev_join)
            if (hipEventCreateWithFlags(&ev, hipEventDisableTiming | hipEventReleaseToSystem) != hipSuccess)
                return fail(nullptr, LAM_HIP_EHIP, "hipEventCreate failed");
    }
    // the error word of the bounded in-kernel waits (reducer workgroups, fused update, direct exchange)
    if (hipSetDevice(c->sh[0].dev) != hipSuccess || hipHostMalloc((void **)&c->direct_err, 64, hipHostMallocPortable | hipHostMallocMapped) != hipSuccess)
        return fail(nullptr, LAM_HIP_EHIP, "hipHostMalloc (error word) failed");
    memset(c->direct_err, 0, 64);
    // peer access between distinct devices of one process (direct xGMI stores)
    for (auto &s : c->sh)
        for (auto &t : c->sh)
            if (s.dev != t.dev) {
                int can = 0;
                (void)hipSetDevice(s.dev);
                if (hipDeviceCanAccessPeer(&can, s.dev, t.dev) != hipSuccess || !can)
                    return fail(nullptr, LAM_HIP_EHIP, "device %d cannot access peer %d", s.dev, t.dev);
                hipError_t pe = hipDeviceEnablePeerAccess(t.dev, 0);
                if (pe != hipSuccess && pe != hipErrorPeerAccessAlreadyEnabled)
                    return fail(nullptr, LAM_HIP_EHIP, "hipDeviceEnablePeerAccess(%d->%d): %s", s.dev, t.dev, hipGetErrorString(pe));
                (void)hipGetLastError();
            }
    return 0;
}

// make the compute stream wait for an all-gather still in flight on the comm stream
int settle_gather(lam_hip_ctx *c)
{
    if (!c->gather_pending) return 0;
    for (auto &s : c->sh) {
        LAMCHK(set_dev(c, s));
        HIPCHK(c, hipStreamWaitEvent(s.stream, s.ev_gathered, 0));
    }
    c->gather_pending = false;
    return 0;
}

int sync_all(lam_hip_ctx *c)
{
    for (auto &s : c->sh) {
        LAMCHK(set_dev(c, s));
        HIPCHK(c, hipStreamSynchronize(s.stream));
    }
    return 0;
}

// Does the producer launch of this dot product carry a reducer workgroup (lam_kernels.h, Finalize)?  The
// symmetric product's second pass writes plain per-workgroup partials of p.Ap: its consumer sums them.
bool producer_reduces(const lam_hip_ctx *c, bool second) { return c->opt_finalize != 0 && (second || !c->symv_active()); }

Finalize no_finalize(const lam_hip_ctx *c)
{
    Finalize f;
    f.active = 0;
    f.mail = 0;
    f.seq = 0;
    f.dst.n = 0;
    f.slot = 0;
    f.host_err = c ? c->direct_err : nullptr;
    return f;
}

// Where the reduced partial of shard `s` goes (see lam_kernels.h, Finalize): slot `index` of the
// gather array of every local shard (one process: peer stores) or of this rank (rank mode).
Finalize make_finalize(lam_hip_ctx *c, ShardBase &s, bool second)
{
    Finalize f = no_finalize(c);
    f.slot = s.index;
    if (c->rank_mode) { f.dst.n = 1; f.dst.p[0] = second ? s.gather_b : s.gather_a; }
    else {
        // one process: slot q of every local shard's gather array (one shard: its own array, slot 0 -- the
        // consumer then reads ONE number instead of summing 32768 GEMV partials in each of its workgroups)
        f.dst.n = (int)c->sh.size();
        for (int j = 0; j < f.dst.n; j++) f.dst.p[j] = second ? (void *)c->sh[j].gather_b : (void *)c->sh[j].gather_a;
    }
    f.active = producer_reduces(c, second) ? 1 : 0;
    return f;
}

// Exchange the shards' partials of a dot product so that the next kernel can sum them in shard order.
//   1 shard            : nothing (the producer's reducer workgroup left the total in gather[0]; without a
//                        reducer -- option finalize = 0, the symmetric product -- the consumer sums the partials)
//   several, 1 process : the producer's reducer workgroup stored the shard's partial into slot q of every
//                        shard's gather array (peer stores); events order the consumers behind them
//   rank mode          : in-place ncclAllGather of the 8-byte partials (slot = rank)
// `finalized` = the producer kernel already reduced its partials (Finalize); otherwise a 1-block
// finalize_sum_kernel does it here (cg_init, and option "finalize" = 0).
// Two halves per shard, so that every shard can be driven by a host thread of its own: reduce_post is what the
// PRODUCING shard puts on its stream behind the producer kernel, reduce_wait makes a CONSUMING shard's stream wait
// for its peers' posts -- which must all have been issued by then (single thread: post for all shards, then wait
// for all; threads: a host barrier in between).  The shard's device is current in both.
int reduce_post(lam_hip_ctx *c, ShardBase &s, bool second, bool use_gemv_part, bool check_stop, bool finalized)
{
    if (!c->rank_mode && c->total_shards == 1) return 0;
    if (!finalized) {
        Finalize f = make_finalize(c, s, second);
        const double *src = use_gemv_part ? s.part_gemv : s.part_vec;
        const int nsrc = use_gemv_part ? s.gemv_blocks : s.vec_blocks;
        hipLaunchKernelGGL(finalize_sum_kernel, dim3(1), dim3(kBlock), 0, s.stream, src, nsrc, f.dst, f.slot,
                           check_stop ? (const CgScalars *)s.sc : (const CgScalars *)nullptr);
        LAUNCHED(c);
    }
    if (c->rank_mode) {
        double *buf = second ? s.gather_b : s.gather_a;
        NCCLCHK(c, ncclAllGather(buf + c->rank, buf, 1, ncclDouble, c->comm, s.stream));
        c->n_collectives++;
    } else {
        RECORD(c, second ? s.ev_b : s.ev_a, s.stream);
    }
    return 0;
}

// Is the all-to-all ordering between the shards' streams done through the hub?
#ifdef LAM_TUNING_VARIANTS
bool hub_active(const lam_hip_ctx *c) { return !c->rank_mode && c->total_shards > 2 && c->opt_hub != 0 && c->hub_stream != nullptr; }
#else
constexpr bool hub_active(const lam_hip_ctx *) { return false; }     // tuning build only (measured slower in wall time)
#endif

// One process, several shards: after every shard has posted exchange `which` (0 = p.Ap partials, 1 = r.r partials,
// 2 = p slices), the hub stream waits for the P posts and records ONE join event; every shard then waits for that
// event -- 2P + 1 runtime calls where the all-to-all form needs P(P-1) (P = 8: 17 instead of 56).  The join adds
// one event hop on the device; every stream still depends on every post (the hub's wait list is all of them), and the
// events carry the same system-scope release / acquire as before.
int hub_join(lam_hip_ctx *c, int which)
{
    if (!hub_active(c)) return 0;
    LAMCHK(set_dev(c, c->sh[0]));
    for (auto &t : c->sh) WAITEV(c, c->hub_stream, which == 0 ? t.ev_a : (which == 1 ? t.ev_b : t.ev_p));
    RECORD(c, c->ev_join[which], c->hub_stream);
    return 0;
}

int reduce_wait(lam_hip_ctx *c, ShardBase &s, bool second)
{
    if (c->rank_mode || c->total_shards == 1) return 0;
    if (hub_active(c)) { WAITEV(c, s.stream, c->ev_join[second ? 1 : 0]); return 0; }
    for (auto &t : c->sh)
        if (&t != &s) WAITEV(c, s.stream, second ? t.ev_b : t.ev_a);
    return 0;
}

// both halves for all shards from one thread (cg_init)
int reduce_step(lam_hip_ctx *c, bool second, bool use_gemv_part, bool check_stop, bool finalized)
{
    if (!c->rank_mode && c->total_shards == 1) return 0;
    for (auto &s : c->sh) {
        LAMCHK(set_dev(c, s));
        LAMCHK(reduce_post(c, s, second, use_gemv_part, check_stop, finalized));
    }
    LAMCHK(hub_join(c, second ? 1 : 0));
    for (auto &s : c->sh) {
        LAMCHK(set_dev(c, s));
        LAMCHK(reduce_wait(c, s, second));
    }
    return 0;
}

void red_source(lam_hip_ctx *c, ShardBase &s, bool second, bool use_gemv_part, bool finalized, const double **red, int *nred)
{
    if (!c->rank_mode && c->total_shards == 1 && !finalized) {
        *red = use_gemv_part ? s.part_gemv : s.part_vec;
        *nred = use_gemv_part ? s.gemv_blocks : s.vec_blocks;
    } else {
        *red = second ? s.gather_b : s.gather_a;
        *nred = c->total_shards;       // == nranks in rank mode
    }
}

// rank mode: make this rank's replica of p complete after the slices were stored (RCCL all-gather)
int gather_p_rank(lam_hip_ctx *c)
{
    ShardBase &s = c->sh[0];
    const uint64_t base = c->n / (uint64_t)c->nranks;
    const size_t ev = c->esz_v();
    const ncclDataType_t dt = c->dtype == LAM_HIP_F64 ? ncclDouble : ncclFloat;
    // The all-gather runs on its own stream so that the next GEMV's own-slice panel overlaps it;
    // the two streams are tied by events, so operations on the communicator stay totally ordered
    // (every other collective is enqueued on s.stream after a wait on ev_gathered).
    hipStream_t cs = c->opt_overlap ? s.comm_stream : s.stream;
    if (c->opt_overlap) {
        RECORD(c, s.ev_p, s.stream);
        WAITEV(c, cs, s.ev_p);
    }
    struct Done {   // record ev_gathered on every exit path below
        lam_hip_ctx *c; ShardBase &s; hipStream_t cs;
        int finish() {
            if (!c->opt_overlap) return 0;
            RECORD(c, s.ev_gathered, cs);
            c->gather_pending = true;
            return 0;
        }
    } done{c, s, cs};
    if (c->n % (uint64_t)c->nranks == 0) {
        NCCLCHK(c, ncclAllGather((const char *)s.p + s.row0 * ev, s.p, base, dt, c->comm, cs));
        c->n_collectives++;
        return done.finish();
    }
    // uneven last block (reference: MPI_Allgatherv): one broadcast per owner
    NCCLCHK(c, ncclGroupStart());
    for (int q = 0; q < c->nranks; q++) {
        uint64_t r0, nr;
        partition(c->n, c->nranks, q, &r0, &nr);
        char *ptr = (char *)s.p + r0 * ev;
        NCCLCHK(c, ncclBroadcast(ptr, ptr, nr, dt, q, c->comm, cs));
        c->n_collectives++;
    }
    NCCLCHK(c, ncclGroupEnd());
    return done.finish();
}

// one process, several shards: the slices were stored straight into every replica (peer stores); events order the
// next reader of a replica behind all of its writers.  Halves as for reduce_post / reduce_wait.
int gather_post(lam_hip_ctx *c, ShardBase &s)
{
    if (c->rank_mode || c->total_shards == 1) return 0;
    RECORD(c, s.ev_p, s.stream);
    return 0;
}
int gather_wait(lam_hip_ctx *c, ShardBase &s)
{
    if (c->rank_mode || c->total_shards == 1) return 0;
    if (hub_active(c)) { WAITEV(c, s.stream, c->ev_join[2]); return 0; }
    for (auto &t : c->sh)
        if (&t != &s) WAITEV(c, s.stream, t.ev_p);
    return 0;
}

// make every replica of p complete after the slices were stored (all shards, one thread: cg_init)
int gather_p_step(lam_hip_ctx *c)
{
    if (!c->rank_mode && c->total_shards == 1) return 0;
    if (c->rank_mode) return gather_p_rank(c);
    for (auto &s : c->sh) {
        LAMCHK(set_dev(c, s));
        LAMCHK(gather_post(c, s));
    }
    LAMCHK(hub_join(c, 2));
    for (auto &s : c->sh) {
        LAMCHK(set_dev(c, s));
        LAMCHK(gather_wait(c, s));
    }
    return 0;
}

}  // namespace

// The typed bodies below are written as generic lambdas over Impl<TA,TV>; TV is recovered with
// this small trait.
namespace {
template <typename T> struct ImplTraits;
template <typename TA_, typename TV_> struct ImplTraits<Impl<TA_, TV_>> { using TA = TA_; using TV = TV_; };

// Fill the partial arrays with the sentinel the reducer workgroups wait on (lam_kernels.h, Finalize).
// Enqueued at the end of cg_init: whatever wrote plain values into them before (cg_init's own partials,
// a roofline probe) is behind it in stream order.
int arm_partials(lam_hip_ctx *c)
{
    if (!c->opt_finalize) return 0;
    for (auto &s : c->sh) {
        LAMCHK(set_dev(c, s));
        hipLaunchKernelGGL(arm_partials_kernel, dim3(std::max(1, std::min(64, s.part_gemv_cap / kBlock))), dim3(kBlock), 0, s.stream,
                           s.part_gemv, s.part_gemv_cap);
        HIPCHK(c, hipGetLastError());
        hipLaunchKernelGGL(arm_partials_kernel, dim3(1), dim3(kBlock), 0, s.stream, s.part_vec, kVecBlocksMax);
        HIPCHK(c, hipGetLastError());
    }
    return 0;
}

MailPost no_post()
{
    MailPost p;
    p.n = 0; p.rank = 0; p.seq = 0;
    for (auto &m : p.mail) m = nullptr;
    return p;
}

// The context's mailbox (lam_kernels.h, Mail) and the pinned error word of the bounded waits.  Fine-grained
// (uncached) memory where the runtime offers it: in the direct exchange it is polled by this rank's kernels
// while peers write it over xGMI; with one shard only the launch's own reducer workgroup writes it.
int ensure_mail(lam_hip_ctx *c, ShardBase &s, bool *got_finegrained)
{
    if (got_finegrained) *got_finegrained = true;
    LAMCHK(set_dev(c, s));
    if (s.mail == nullptr) {
        if (hipExtMallocWithFlags((void **)&s.mail, sizeof(Mail), hipDeviceMallocUncached) != hipSuccess) {
            (void)hipGetLastError();
            if (hipExtMallocWithFlags((void **)&s.mail, sizeof(Mail), hipDeviceMallocFinegrained) != hipSuccess) {
                (void)hipGetLastError();
                s.mail = nullptr;
                s.mail_coarse = true;
                HIPCHK(c, hipMalloc((void **)&s.mail, sizeof(Mail)));
            }
        }
        HIPCHK(c, hipMemset(s.mail, 0, sizeof(Mail)));
    }
    if (got_finegrained) *got_finegrained = !s.mail_coarse;
    if (s.bcast == nullptr) {
        HIPCHK(c, hipMalloc((void **)&s.bcast, 2 * kBcastLines * sizeof(BcastLine)));
        HIPCHK(c, hipMemset(s.bcast, 0, 2 * kBcastLines * sizeof(BcastLine)));
    }
    return 0;
}

// ---- direct exchange (option exchange = 2) --------------------------------------------------------
void close_direct(lam_hip_ctx *c)
{
    for (int i = 0; i < c->n_ipc_opened; i++) (void)hipIpcCloseMemHandle(c->ipc_opened[i]);
    c->n_ipc_opened = 0;
    c->direct_ok = false;
    c->direct_gen = ~0ull;
}

// What a rank tells the others about its buffers.  Same process (ranks as threads): the pointers are
// used as they are; another process: the HIP IPC handles are opened.
struct DirectHello {
    int pid, dev;
    void *p, *mail;
    hipIpcMemHandle_t hp, hm;
    int have_handles;
};

// Collective: every rank must call it the same number of times (it is part of lam_hip_cg_init).  Ends
// with an agreement, so either all ranks use the direct exchange or none does.
// One process, several shards: the same exchange without any mapping step -- all shards live in this address space and
// peer access between their devices was enabled when the context was created.  The kernels of one shard wait (bounded)
// for stores made by the kernels of the other shards, so every shard's stream must be able to make progress on its own:
// guaranteed when every shard has a device of its own; shards that SHARE a device could sit behind each other in one
// hardware queue (a waiting kernel in front of the kernel it waits for), so that layout gets the direct exchange only on
// request (LAM_HIP_DIRECT_SAME_DEVICE=1: tests, with GPU_MAX_HW_QUEUES >= number of shards) and the event exchange otherwise.
int setup_direct_local(lam_hip_ctx *c)
{
    if (c->direct_gen == c->problem_gen) return 0;
    close_direct(c);
    bool ok = true;
    if (const char *off = getenv("LAM_HIP_DIRECT_DISABLE"))
        if (*off && strcmp(off, "0") != 0) ok = false;
    bool shared = false;
    for (auto &s : c->sh)
        for (auto &t : c->sh)
            if (&s != &t && s.dev == t.dev) shared = true;
    if (shared) {
        const char *same = getenv("LAM_HIP_DIRECT_SAME_DEVICE");
        if (!(same && *same && strcmp(same, "0") != 0)) ok = false;
    }
    for (auto &s : c->sh) {
        bool fine = false;
        LAMCHK(ensure_mail(c, s, &fine));
        if (!fine) ok = false;                  // peers must not poll-and-write ordinary (cached) memory
        c->peer_p[s.index] = s.p;
        c->peer_mail[s.index] = s.mail;
    }
    c->direct_ok = ok;
    c->direct_gen = c->problem_gen;
    return 0;
}

int setup_direct(lam_hip_ctx *c)
{
    if (!c->rank_mode) return setup_direct_local(c);
    if (c->direct_gen == c->problem_gen) return 0;
    close_direct(c);
    ShardBase &s = c->sh[0];
    LAMCHK(set_dev(c, s));
    bool ok = true;
    if (const char *off = getenv("LAM_HIP_DIRECT_DISABLE"))        // pretend this rank cannot map its peers: every rank
        if (*off && strcmp(off, "0") != 0) ok = false;            // must then fall back together (tests; a kill switch)
    {
        bool fine = false;
        LAMCHK(ensure_mail(c, s, &fine));
        if (!fine) ok = false;                  // peers must not poll-and-write ordinary (cached) memory
    }
    const int P = c->nranks;
    constexpr size_t kRec = 256;
    static_assert(sizeof(DirectHello) <= kRec, "hello record");
    static_assert(kRec * kMaxShards <= 4096, "hello records fit the set-up scratch");
    if (c->agree_buf == nullptr) HIPCHK(c, hipMalloc((void **)&c->agree_buf, 4096));
    struct { void *p; } dev{c->agree_buf};       // kept for the life of the context (no hipFree in a collective path)
    std::vector<char> host(kRec * (size_t)P, 0);
    DirectHello me;
    memset(&me, 0, sizeof me);
    me.pid = (int)getpid();
    me.dev = s.dev;
    me.p = s.p;
    me.mail = s.mail;
    me.have_handles = ok && hipIpcGetMemHandle(&me.hp, s.p) == hipSuccess && hipIpcGetMemHandle(&me.hm, s.mail) == hipSuccess;
    (void)hipGetLastError();
    memcpy(host.data() + kRec * (size_t)c->rank, &me, sizeof me);
    HIPCHK(c, hipMemcpyAsync((char *)dev.p + kRec * (size_t)c->rank, host.data() + kRec * (size_t)c->rank, kRec, hipMemcpyHostToDevice, s.stream));
    NCCLCHK(c, ncclAllGather((char *)dev.p + kRec * (size_t)c->rank, dev.p, kRec, ncclChar, c->comm, s.stream));
    c->n_collectives++;
    HIPCHK(c, hipMemcpyAsync(host.data(), dev.p, kRec * (size_t)P, hipMemcpyDeviceToHost, s.stream));
    HIPCHK(c, hipStreamSynchronize(s.stream));
    for (int q = 0; q < P && ok; q++) {
        DirectHello h;
        memcpy(&h, host.data() + kRec * (size_t)q, sizeof h);
        if (q == c->rank) { c->peer_p[q] = s.p; c->peer_mail[q] = s.mail; continue; }
        if (h.mail == nullptr) { ok = false; break; }
        if (h.pid == me.pid) {
            // a thread of this process: same address space; another device needs peer access
            if (h.dev != s.dev) {
                hipError_t pe = hipDeviceEnablePeerAccess(h.dev, 0);
                if (pe != hipSuccess && pe != hipErrorPeerAccessAlreadyEnabled) ok = false;
                (void)hipGetLastError();
            }
            c->peer_p[q] = h.p;
            c->peer_mail[q] = (Mail *)h.mail;
        } else {
            void *pp = nullptr, *pm = nullptr;
            if (!h.have_handles || hipIpcOpenMemHandle(&pp, h.hp, hipIpcMemLazyEnablePeerAccess) != hipSuccess) { (void)hipGetLastError(); ok = false; break; }
            c->ipc_opened[c->n_ipc_opened++] = pp;
            if (hipIpcOpenMemHandle(&pm, h.hm, hipIpcMemLazyEnablePeerAccess) != hipSuccess) { (void)hipGetLastError(); ok = false; break; }
            c->ipc_opened[c->n_ipc_opened++] = pm;
            c->peer_p[q] = pp;
            c->peer_mail[q] = (Mail *)pm;
        }
    }
    int all = 0;
    LAMCHK(lam_hip_all_ok(c, ok ? 1 : 0, &all));
    if (!all) close_direct(c);
    c->direct_ok = all != 0;
    c->direct_gen = c->problem_gen;
    return 0;
}

// Is the GEMV of iteration k timed (HIP-event pair on the launch stream, shard 0 only)?  Option "gemv_timing" = T
// times every T-th iteration; each record is a marker packet between the iteration's kernels, so T > 1 keeps most
// iterations free of them.
bool timed_iteration(const lam_hip_ctx *c, const ShardBase &s, int k)
{
    return &s == &c->sh[0] && c->opt_gemv_timing > 0 && (k - 1) % c->opt_gemv_timing == 0;
}

// Can a launch of `blocks` workgroups of update_fused_kernel be resident all at once?  Its workgroups wait for each
// other inside the launch (compute workgroups for the reducer's broadcast, the reducer for their partials), so a
// workgroup that cannot start until another one exits would hold everybody until the bounded waits expire.  256-thread
// workgroups are admitted per CU up to min(occupancy API, 8) (MI355X_MICROARCH.md, residency); a CU mask or a
// partitioned device that the runtime reports shows up in the CU count.  What the query cannot see (other kernels on
// the device) is still caught by the bounded waits, which end in an error, never in a hang or a silent NaN.
template <typename TV>
bool fused_launch_resident(lam_hip_ctx *c, const ShardBase &s, int blocks)
{
    int per_cu = 0, cus = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, update_fused_kernel<TV>, kBlock, 0) != hipSuccess ||
        hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, s.dev) != hipSuccess) {
        (void)hipGetLastError();
        return false;
    }
    if (c->opt_assume_cus > 0) cus = (int)c->opt_assume_cus;
    return (int64_t)std::min(per_cu, 8) * (int64_t)cus >= (int64_t)blocks;
}

// One shard's iteration on the direct exchange: rank mode has one local shard (index = rank); one process with several
// shards enqueues them one after the other -- no event, no stream wait, no collective: 2-4 launches per shard.
template <typename I>
int enqueue_shard_direct(lam_hip_ctx *c, ShardBase &s, int k, double rel_error, int slot)
{
        using TV = typename ImplTraits<I>::TV;
        LAMCHK(set_dev(c, s));
        const int P = c->total_shards;
        const int me = s.index;
        const unsigned long long seq = c->seq_base + (unsigned)k;
        c->seq_span = std::max<uint64_t>(c->seq_span, (uint64_t)k + 1);
        // 1. GEMV.  p for this iteration: the own slice is local; the others were stored into this rank's
        //    replica by the peers' update_p of iteration k-1 (k == 1: by cg_init) -- wait for their flags
        //    behind the own-slice panel.
        uint64_t lo = 0, hi = 0;
        uint64_t a = s.row0, b = s.row0 + s.nrows;
        // option "overlap" = 0: no own-slice panel -- wait for the flags first, then one GEMV launch (the split
        // costs ~8 us of launch and ramp; it pays when the slices arrive later than that)
        if (P > 1 && c->opt_overlap && (!I::fast_ok(c) || (a % I::VEC == 0 && b % I::VEC == 0))) { lo = a; hi = b; }
        Finalize fa = no_finalize(c);
        fa.active = 1; fa.mail = 1; fa.seq = seq; fa.slot = 0; fa.dst.n = P;
        for (int q = 0; q < P; q++) fa.dst.p[q] = &c->peer_mail[q]->pap[me];
        BlockCounts nb;
        for (int q = 0; q < kMaxShards; q++) {
            uint64_t r0 = 0, nr = 0;
            if (q < P) partition(c->n, P, q, &r0, &nr);
            nb.n[q] = q < P ? vec_grid(nr) : 0;
        }
        // LAM_HIP_DEBUG_DIRECT_STALE=<rank>: test hook -- that rank's p replica is perturbed in front of the GEMV of
        // iteration 3, which is what a stale read of a peer's slice would amount to: the ranks stay in step, the result
        // is wrong, and lam_hip_solve's residual check must notice and solve again on the RCCL exchange.  Never set it otherwise.
        static const char *stale = getenv("LAM_HIP_DEBUG_DIRECT_STALE");
        if (stale && *stale && atoi(stale) == me && k == 3) {
            hipLaunchKernelGGL((axpby_kernel<TV>), dim3(vec_grid(c->n)), dim3(kBlock), 0, s.stream, (TV)0, (const TV *)s.p, (TV)1.001, (TV *)s.p, c->n);
            LAUNCHED(c);
        }
        // the fused update launch of iteration k-1 may have waited for the slices already (its waiter workgroup)
        const bool need_wait = P > 1 && k > 1 && s.waited_k != k - 1;
        const bool timed = timed_iteration(c, s, k);
        s.split_slot[slot] = hi > lo;
        s.timed_slot[slot] = timed;
        if (hi > lo) {
            if (timed) RECORD(c, s.ev_g0[slot], s.stream);
            LAMCHK(I::launch_gemv(c, s, (const TV *)s.p, (TV *)s.Ap, nullptr, s.sc, 1, lo, hi));
            if (timed) RECORD(c, s.ev_g1[slot], s.stream);
        }
        if (need_wait) {
            hipLaunchKernelGGL(wait_p_kernel, dim3(1), dim3(kBlock), 0, s.stream, (const Mail *)s.mail, P, me, nb, seq - 1,
                               (const CgScalars *)s.sc, c->direct_err);
            LAUNCHED(c);
        }
        if (hi > lo) {
            if (timed) RECORD(c, s.ev_g2[slot], s.stream);
            LAMCHK(I::launch_gemv(c, s, (const TV *)s.p, (TV *)s.Ap, s.part_gemv, s.sc, 2, lo, hi, &fa));
            if (timed) RECORD(c, s.ev_g3[slot], s.stream);
        } else {
            if (timed) RECORD(c, s.ev_g0[slot], s.stream);
            LAMCHK(I::launch_gemv(c, s, (const TV *)s.p, (TV *)s.Ap, s.part_gemv, s.sc, 0, 0, 0, &fa));
            if (timed) RECORD(c, s.ev_g1[slot], s.stream);
        }
        // 2. x, r: waits in the kernel for the P partials of p.Ap; its reducer posts the r.r partial
        Finalize fb = fa;
        for (int q = 0; q < P; q++) fb.dst.p[q] = &c->peer_mail[q]->rr[me];
        if (c->fuse_active) {
            // steps 2 and 3 in ONE launch; without an own-slice panel (overlap 0) a waiter workgroup also holds the
            // launch open until the peers' slices for the next GEMV are in: 2 launches per iteration
            PtrList plf;
            plf.n = P;
            for (int q = 0; q < P; q++) plf.p[q] = c->peer_p[q];
            MailPost postf = no_post();
            postf.n = P; postf.rank = me; postf.seq = seq;
            for (int q = 0; q < P; q++) postf.mail[q] = c->peer_mail[q];
            static const char *dropf = getenv("LAM_HIP_DEBUG_DIRECT_DROP");
            if (dropf && *dropf && atoi(dropf) == me && k == 3) postf.seq = ~0ull;      // test hook, see below
            const bool waiter = P > 1 && !(hi > lo);
            hipLaunchKernelGGL((update_fused_kernel<TV>), dim3(s.vec_blocks + 1 + (waiter ? 1 : 0)), dim3(kBlock), 0, s.stream,
                               (const double *)nullptr, 0, s.sc, k, rel_error, (const TV *)s.p + s.row0, (const TV *)s.Ap, (TV *)s.x,
                               (TV *)s.r, s.nrows, s.part_vec, s.vec_blocks, fb, MailWait{s.mail->pap, P, seq, c->direct_err},
                               MailWait{s.mail->rr, P, seq, c->direct_err}, s.bcast, plf, s.row0, (volatile int *)s.host_flags, postf,
                               (const Mail *)s.mail, nb);
            LAUNCHED(c);
            if (waiter) s.waited_k = k;
            return 0;
        }
        hipLaunchKernelGGL((update_xr_kernel<TV>), dim3(s.vec_blocks + 1), dim3(kBlock), 0, s.stream, (const double *)nullptr, 0, s.sc, k,
                           (const TV *)s.p + s.row0, (const TV *)s.Ap, (TV *)s.x, (TV *)s.r, s.nrows, s.part_vec, fb,
                           MailWait{s.mail->pap, P, seq, c->direct_err});
        LAUNCHED(c);
        // 3. stop test + p slice into every replica + flags
        PtrList pl;
        pl.n = P;
        for (int q = 0; q < P; q++) pl.p[q] = c->peer_p[q];
        MailPost post = no_post();
        post.n = P; post.rank = me; post.seq = seq;
        for (int q = 0; q < P; q++) post.mail[q] = c->peer_mail[q];
        // LAM_HIP_DEBUG_DIRECT_DROP=<rank>: test hook -- that rank "forgets" to raise its p-slice flags in iteration
        // 3, so every bounded wait downstream of it expires: shows that the grid drains, the error surfaces on all
        // ranks and the caller survives (tests/test_gpu_rank_mock.py).  Never set it otherwise.
        static const char *drop = getenv("LAM_HIP_DEBUG_DIRECT_DROP");
        if (drop && *drop && atoi(drop) == me && k == 3) post.seq = ~0ull;
        hipLaunchKernelGGL((update_p_kernel<TV>), dim3(s.vec_blocks), dim3(kBlock), 0, s.stream, (const double *)nullptr, 0, s.sc, k,
                           rel_error, (const TV *)s.r, (const TV *)s.p + s.row0, pl, s.row0, s.nrows, (volatile int *)s.host_flags,
                           MailWait{s.mail->rr, P, seq, c->direct_err}, post);
        LAUNCHED(c);
        return 0;
}

int enqueue_iteration_direct(lam_hip_ctx *c, int k, double rel_error, int slot)
{
    return dispatch(c, [&](auto impl) -> int {
        using I = decltype(impl);
        for (auto &s : c->sh) LAMCHK(enqueue_shard_direct<I>(c, s, k, rel_error, slot));
        return 0;
    });
}

// gather-Ap exchange, one process with several shards: CG state = x slice, FULL r and p on every shard.  The rhs
// slices are replicated once with peer copies (the rank mode's one-off all-gather), after that no vector is exchanged
// but Ap.
int do_cg_init_exchange1_local(lam_hip_ctx *c)
{
    return dispatch(c, [&](auto impl) -> int {
        using TV = typename ImplTraits<decltype(impl)>::TV;
        const size_t ev = c->esz_v();
        LAMCHK(sync_all(c));
        for (auto &dst : c->sh) {
            LAMCHK(set_dev(c, dst));
            for (auto &src : c->sh)
                HIPCHK(c, hipMemcpyAsync((char *)dst.r_full + src.row0 * ev, src.b, src.nrows * ev, hipMemcpyDefault, dst.stream));
        }
        const int grid = vec_grid(c->n);
        for (auto &s : c->sh) {
            LAMCHK(set_dev(c, s));
            hipLaunchKernelGGL((cg_init_full_kernel<TV>), dim3(grid), dim3(kBlock), 0, s.stream, (TV *)s.r_full, (TV *)s.p,
                               (TV *)s.x, c->n, s.nrows, s.part_vec);
            HIPCHK(c, hipGetLastError());
            hipLaunchKernelGGL(cg_init_scalars_kernel, dim3(1), dim3(kBlock), 0, s.stream, (const double *)s.part_vec, grid, s.sc);
            HIPCHK(c, hipGetLastError());
        }
        LAMCHK(arm_partials(c));
        c->k_done = 0;
        c->cg_ready = true;
        c->cg_exchange1 = true;
        return 0;
    });
}

// One iteration on the gather-Ap exchange with several shards in one process (the reference's CPU path gathers Ap too,
// ConjugateGradient_CPU_MPI_OMP.hpp:505; the single-process CUDA class gathers it on device 0,
// ConjugateGradient_MultiGPUS_CUDA.cu:362-376).  Per shard: the GEMV stores every row of its Ap slice into its record in
// EVERY shard's gather buffer (peer stores over xGMI) and its reducer workgroup does the same with the shard's p.Ap
// partial; one event record.  Then the iteration's ONLY join -- through shard 0's stream (2(P-1)+1 runtime calls) or
// all-to-all (P(P-1)) -- and the two full-length vector kernels, which need nothing from the peers any more: r.r is the
// same sum on every shard.  Same kernels, same arithmetic as the rank mode's exchange 1: bit-identical to it.
// The gather buffer is double (iteration parity): shard q may start GEMV k+1 -- which stores into its peers' buffers --
// as soon as ITS update of iteration k is done, while a slower peer still reads the records of iteration k; GEMV k+2
// cannot start before every peer has finished GEMV k+1, i.e. its update k.
int enqueue_iteration_exchange1_local(lam_hip_ctx *c, int k, double rel_error, int slot)
{
    return dispatch(c, [&](auto impl) -> int {
        using I = decltype(impl);
        using TV = typename ImplTraits<I>::TV;
        const int P = c->total_shards;
        const uint64_t stride = c->ex1_stride_bytes(), base = c->ex1_base();
        auto buf = [&](ShardBase &t) { return (char *)t.ap_gather + (size_t)(k & 1) * t.ap_gather_bytes; };
        for (auto &s : c->sh) {
            LAMCHK(set_dev(c, s));
            const uint64_t off = (uint64_t)s.index * stride;
            Finalize f = no_finalize(c);
            f.active = c->opt_finalize ? 1 : 0;
            f.slot = 0;
            f.dst.n = P;
            PtrList yp;
            yp.n = 0;
            for (auto &t : c->sh) {
                f.dst.p[t.index] = buf(t) + off + base * sizeof(TV);
                if (&t != &s) yp.p[yp.n++] = buf(t) + off;
            }
            const bool timed = timed_iteration(c, s, k);
            s.split_slot[slot] = false;
            s.timed_slot[slot] = timed;
            if (timed) RECORD(c, s.ev_g0[slot], s.stream);
            LAMCHK(I::launch_gemv(c, s, (const TV *)s.p, (TV *)(buf(s) + off), s.part_gemv, s.sc, 0, 0, 0, &f, &yp));
            if (timed) RECORD(c, s.ev_g1[slot], s.stream);
            if (!c->opt_finalize) {
                hipLaunchKernelGGL(finalize_sum_kernel, dim3(1), dim3(kBlock), 0, s.stream, (const double *)s.part_gemv, s.gemv_blocks,
                                   f.dst, 0, (const CgScalars *)s.sc);
                LAUNCHED(c);
            }
            if (!(c->opt_join && &s == &c->sh[0])) RECORD(c, s.ev_a, s.stream);
        }
        // the join
        if (c->opt_join) {
            ShardBase &s0 = c->sh[0];
            LAMCHK(set_dev(c, s0));
            for (auto &t : c->sh)
                if (&t != &s0) WAITEV(c, s0.stream, t.ev_a);
            RECORD(c, c->ev_join[0], s0.stream);
        }
        const int grid = vec_grid(c->n);
        for (auto &s : c->sh) {
            LAMCHK(set_dev(c, s));
            if (c->opt_join) {
                if (&s != &c->sh[0]) WAITEV(c, s.stream, c->ev_join[0]);
            } else {
                for (auto &t : c->sh)
                    if (&t != &s) WAITEV(c, s.stream, t.ev_a);
            }
            hipLaunchKernelGGL((update_xr_full_kernel<TV>), dim3(grid), dim3(kBlock), 0, s.stream, (const char *)buf(s), stride,
                               base, P, s.sc, k, (const TV *)s.p, (TV *)s.x, (TV *)s.r_full, c->n, s.row0, s.nrows, s.part_vec);
            LAUNCHED(c);
            hipLaunchKernelGGL((update_p_full_kernel<TV>), dim3(grid), dim3(kBlock), 0, s.stream, (const double *)s.part_vec, grid, s.sc,
                               k, rel_error, (const TV *)s.r_full, (TV *)s.p, c->n, (volatile int *)s.host_flags);
            LAUNCHED(c);
        }
        return 0;
    });
}

// gather-Ap exchange: CG state = x slice, FULL r and p on every rank
int do_cg_init_exchange1(lam_hip_ctx *c)
{
    if (!c->rank_mode) return do_cg_init_exchange1_local(c);
    return dispatch(c, [&](auto impl) -> int {
        using TV = typename ImplTraits<decltype(impl)>::TV;
        ShardBase &s = c->sh[0];
        LAMCHK(set_dev(c, s));
        const ncclDataType_t dt = c->dtype == LAM_HIP_F64 ? ncclDouble : ncclFloat;
        NCCLCHK(c, ncclAllGather(s.b, s.r_full, c->ex1_base(), dt, c->comm, s.stream));   // r_full = b
        c->n_collectives++;
        const int grid = vec_grid(c->n);
        hipLaunchKernelGGL((cg_init_full_kernel<TV>), dim3(grid), dim3(kBlock), 0, s.stream, (TV *)s.r_full, (TV *)s.p,
                           (TV *)s.x, c->n, s.nrows, s.part_vec);
        HIPCHK(c, hipGetLastError());
        hipLaunchKernelGGL(cg_init_scalars_kernel, dim3(1), dim3(kBlock), 0, s.stream, (const double *)s.part_vec, grid, s.sc);
        HIPCHK(c, hipGetLastError());
        LAMCHK(arm_partials(c));
        c->k_done = 0;
        c->cg_ready = true;
        c->cg_exchange1 = true;
        return 0;
    });
}

int enqueue_iteration_exchange1(lam_hip_ctx *c, int k, double rel_error, int slot)
{
    if (!c->rank_mode) return enqueue_iteration_exchange1_local(c, k, rel_error, slot);
    return dispatch(c, [&](auto impl) -> int {
        using I = decltype(impl);
        using TV = typename ImplTraits<I>::TV;
        ShardBase &s = c->sh[0];
        LAMCHK(set_dev(c, s));
        const uint64_t stride = c->ex1_stride_bytes(), base = c->ex1_base();
        char *rec = (char *)s.ap_gather + (uint64_t)c->rank * stride;
        // 1. GEMV straight into this rank's record; its last workgroup leaves the rank's p.Ap partial
        //    behind the slice (with option "finalize" = 0: a 1-block launch does)
        Finalize f = no_finalize(c);
        f.active = c->opt_finalize ? 1 : 0;
        f.dst.n = 1; f.dst.p[0] = rec + base * sizeof(TV); f.slot = 0;
        const bool timed = timed_iteration(c, s, k);
        s.split_slot[slot] = false;
        s.timed_slot[slot] = timed;
        if (timed) RECORD(c, s.ev_g0[slot], s.stream);
        LAMCHK(I::launch_gemv(c, s, (const TV *)s.p, (TV *)rec, s.part_gemv, s.sc, 0, 0, 0, &f));
        if (timed) RECORD(c, s.ev_g1[slot], s.stream);
        if (!c->opt_finalize) {
            hipLaunchKernelGGL(finalize_sum_kernel, dim3(1), dim3(kBlock), 0, s.stream, (const double *)s.part_gemv, s.gemv_blocks,
                               f.dst, 0, (const CgScalars *)s.sc);
            LAUNCHED(c);
        }
        // 2. the iteration's only collective
        NCCLCHK(c, ncclAllGather(rec, s.ap_gather, stride, ncclChar, c->comm, s.stream));
        c->n_collectives++;
        // 3. alpha, x slice, FULL r (+ partials of r.r over the full vector: no collective needed)
        const int grid = vec_grid(c->n);
        hipLaunchKernelGGL((update_xr_full_kernel<TV>), dim3(grid), dim3(kBlock), 0, s.stream, (const char *)s.ap_gather, stride,
                           base, c->nranks, s.sc, k, (const TV *)s.p, (TV *)s.x, (TV *)s.r_full, c->n, s.row0, s.nrows, s.part_vec);
        LAUNCHED(c);
        // 4. beta, stop test, FULL p
        hipLaunchKernelGGL((update_p_full_kernel<TV>), dim3(grid), dim3(kBlock), 0, s.stream, (const double *)s.part_vec, grid, s.sc,
                           k, rel_error, (const TV *)s.r_full, (TV *)s.p, c->n, (volatile int *)s.host_flags);
        LAUNCHED(c);
        return 0;
    });
}

#ifdef LAM_TUNING_VARIANTS
// ---- TUNING BUILD ONLY: the whole-iteration persistent launch (experiment; measured 0.7-2 % slower than the two-launch chain)
constexpr int kPersistLinesMax = 2048;      // >= the most worker workgroups a device can hold (8 x 256 CUs)

// Can the current CG state run on the whole-iteration persistent launch, and with how many workers?  One shard, fp64 /
// fp32, the fast GEMV path, not the symmetric product; the grid (W workers + the reducer) must be RESIDENT at once --
// its workgroups wait for each other for the whole launch -- so W comes from the occupancy query (capped at 8 workgroups
// of 256 threads per CU), rounded down to a multiple of the number of p tiles (every worker's pairs then share one
// rotated tile order, which is what lets a group of pairs share a staged tile).
int decide_persistent(lam_hip_ctx *c)
{
    c->persist_active = false;
    if (c->persist_ticks_host) c->persist_ticks_host[0] = c->persist_ticks_host[1] = 0;
    if (!c->opt_persistent || c->rank_mode || c->total_shards != 1 || c->dtype == LAM_HIP_BF16 || c->symv_active() || !c->opt_finalize) return 0;
    ShardBase &s = c->sh[0];
    LAMCHK(set_dev(c, s));
    return dispatch(c, [&](auto impl) -> int {
        using I = decltype(impl);
        using TA = typename ImplTraits<I>::TA;
        using TV = typename ImplTraits<I>::TV;
        if constexpr (!std::is_same<TA, TV>::value) {
            return 0;
        } else {
            if (!I::fast_ok(c) || s.nrows != c->n || (c->n % 2) != 0) return 0;
            int per_cu = 0, cus = 0;
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, cg_persist_kernel<TA, TV>, kBlock, 0) != hipSuccess ||
                hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, s.dev) != hipSuccess) {
                (void)hipGetLastError();
                return 0;
            }
            if (c->opt_assume_cus > 0) cus = (int)c->opt_assume_cus;
            const int64_t npairs = (int64_t)(c->n / 2);
            const int64_t ntiles = (int64_t)((c->n + 4095) / 4096);
            int64_t W = std::min<int64_t>({(int64_t)std::min(per_cu, 8) * cus - 1, npairs, (int64_t)kPersistLinesMax});
            W = W / ntiles * ntiles;
            if (W < (int64_t)s.vec_blocks || W < 64) return 0;      // too few resident workgroups: two-launch form
            if (c->persist_bc == nullptr) {
                HIPCHK(c, hipMalloc((void **)&c->persist_bc, (size_t)(kVecBlocksMax + kPersistLinesMax) * sizeof(BcastLine)));
                HIPCHK(c, hipMemset(c->persist_bc, 0, (size_t)(kVecBlocksMax + kPersistLinesMax) * sizeof(BcastLine)));
                HIPCHK(c, hipMalloc((void **)&c->persist_ticks, 2 * sizeof(unsigned long long)));
                HIPCHK(c, hipHostMalloc((void **)&c->persist_ticks_host, 2 * sizeof(unsigned long long), hipHostMallocDefault));
            }
            HIPCHK(c, hipMemsetAsync(c->persist_ticks, 0, 2 * sizeof(unsigned long long), s.stream));
            c->persist_W = (int)W;
            c->persist_active = true;
            return 0;
        }
    });
}

// `count` iterations starting at k_first in ONE launch
int enqueue_persist_chunk(lam_hip_ctx *c, int k_first, int count, double rel_error)
{
    return dispatch(c, [&](auto impl) -> int {
        using I = decltype(impl);
        using TA = typename ImplTraits<I>::TA;
        using TV = typename ImplTraits<I>::TV;
        if constexpr (!std::is_same<TA, TV>::value) {
            return fail(c, LAM_HIP_EINVAL, "persistent launch: not for this dtype");
        } else {
            ShardBase &s = c->sh[0];
            LAMCHK(set_dev(c, s));
            PersistArgs<TA, TV> a;
            a.A = (const TA *)s.A; a.n = c->n;
            a.pbuf[0] = (TV *)s.p; a.pbuf[1] = (TV *)s.tmp;
            a.r = (TV *)s.r; a.x = (TV *)s.x; a.Ap = (TV *)s.Ap;
            a.part_gemv = s.part_gemv; a.part_vec = s.part_vec;
            a.sc = s.sc; a.k_first = k_first; a.k_count = count; a.rel_error = rel_error;
            a.host_flags = (volatile int *)s.host_flags; a.host_err = c->direct_err;
            a.bc_pap = c->persist_bc; a.bc_rr = c->persist_bc + kVecBlocksMax;
            a.W = c->persist_W; a.vec_blocks = s.vec_blocks;
            a.npairs = (uint32_t)(c->n / 2); a.ntiles = (uint32_t)((c->n + 4095) / 4096);
            a.seq_base = c->seq_base;
            c->seq_span = std::max<uint64_t>(c->seq_span, (uint64_t)(k_first + count));
            a.ticks = c->persist_ticks;
            hipLaunchKernelGGL((cg_persist_kernel<TA, TV>), dim3(c->persist_W + 1), dim3(kBlock), 0, s.stream, a);
            LAUNCHED(c);
            return 0;
        }
    });
}
#else
int decide_persistent(lam_hip_ctx *c) { c->persist_active = false; return 0; }
#endif  // LAM_TUNING_VARIANTS

int do_cg_init(lam_hip_ctx *c)
{
    c->cg_direct = false;
    c->epoch++;
    c->seq_base += c->seq_span;    // hand-over numbers never restart (see seq_base)
    c->seq_span = 1;
    for (auto &sh_ : c->sh) sh_.waited_k = 0;
    if (c->direct_err) memset(c->direct_err, 0, 64);
    if (c->exchange2_wanted()) {
        // the state is initialised through RCCL (one-off); the iterations then run on the mailboxes
        LAMCHK(setup_direct(c));
        c->cg_direct = c->direct_ok;
    }
    // The fused vector step (one shard; the direct exchange) is a launch whose workgroups wait for each other: used
    // only when the whole grid (compute workgroups + reducer + waiter) can be resident at once, else the two-kernel form.
    c->fuse_active = false;
    if (c->opt_fuse && c->opt_finalize && (c->cg_direct || (!c->rank_mode && c->total_shards == 1))) {
        ShardBase &s0 = c->sh[0];
        LAMCHK(set_dev(c, s0));
        c->fuse_active = dispatch(c, [&](auto impl) -> int {
            using TV = typename ImplTraits<decltype(impl)>::TV;
            return fused_launch_resident<TV>(c, s0, s0.vec_blocks + 2) ? 1 : 0;
        }) == 1;
    }
    if (c->fuse_active)
        for (auto &sh_ : c->sh) LAMCHK(ensure_mail(c, sh_, nullptr));
    LAMCHK(decide_persistent(c));
    if (c->exchange1_ok()) return do_cg_init_exchange1(c);
    c->cg_exchange1 = false;
    return dispatch(c, [&](auto impl) -> int {
        using TV = typename ImplTraits<decltype(impl)>::TV;
        PtrList pl = plist_p(c);
        for (auto &s : c->sh) {
            LAMCHK(set_dev(c, s));
            hipLaunchKernelGGL((cg_init_kernel<TV>), dim3(s.vec_blocks), dim3(kBlock), 0, s.stream, (const TV *)s.b,
                               (TV *)s.x, (TV *)s.r, pl, s.row0, s.nrows, s.part_vec);
            HIPCHK(c, hipGetLastError());
        }
        LAMCHK(reduce_step(c, /*second=*/true, /*gemv_part=*/false, /*check_stop=*/false, /*finalized=*/false));
        for (auto &s : c->sh) {
            LAMCHK(set_dev(c, s));
            const double *red; int nred;
            red_source(c, s, true, false, /*finalized=*/false, &red, &nred);
            hipLaunchKernelGGL(cg_init_scalars_kernel, dim3(1), dim3(kBlock), 0, s.stream, red, nred, s.sc);
            HIPCHK(c, hipGetLastError());
        }
        LAMCHK(gather_p_step(c));
        LAMCHK(arm_partials(c));
        c->k_done = 0;
        c->cg_ready = true;
        return 0;
    });
}

// ---- the general iteration, one shard at a time -------------------------------------------------------------------
// Four phases per shard; between two phases every shard must have ISSUED the previous one (its event records are
// what the next phase's stream waits refer to).  One host thread: phase by phase over all shards.  One host thread
// per shard (option "host_threads", the shape of the reference's OpenMP-thread-per-device loop,
// ConjugateGradient_MultiGPUS_CUDA.cu:337-378): a host barrier between the phases (iterate_threaded).
//   A  GEMV (+ partial p.Ap) and its post             B  wait for the peers' p.Ap; x, r update (+ partial r.r); post
//   C  wait for the peers' r.r; stop test + p update into every replica; post        D  wait for the peers' p slices
template <typename I>
int phase_gemv(lam_hip_ctx *c, ShardBase &s, int k, int slot)
{
    using TV = typename ImplTraits<I>::TV;
    // With an own-slice panel: that panel first (it only needs the p slice this shard wrote itself), then wait for
    // the all-gather, then the remaining columns.
    uint64_t lo, hi;
    cg_panel<I>(c, s, &lo, &hi);
    const bool timed = timed_iteration(c, s, k);
    s.timed_slot[slot] = timed;
    const bool fin_a = producer_reduces(c, false);
    if (c->symv_active()) {
        s.split_slot[slot] = false;
        if (timed) RECORD(c, s.ev_g0[slot], s.stream);
        LAMCHK(I::launch_symv(c, s, (const TV *)s.p, (TV *)s.Ap, s.part_gemv, s.sc));
        if (timed) RECORD(c, s.ev_g1[slot], s.stream);
        return reduce_post(c, s, false, true, true, fin_a);
    }
    s.split_slot[slot] = hi > lo;
    const Finalize fa = make_finalize(c, s, false);
    if (hi > lo) {
        // the two panels are timed separately so that t_gemv is kernel time, not the wait in between
        if (timed) RECORD(c, s.ev_g0[slot], s.stream);
        LAMCHK(I::launch_gemv(c, s, (const TV *)s.p, (TV *)s.Ap, nullptr, s.sc, 1, lo, hi));
        if (timed) RECORD(c, s.ev_g1[slot], s.stream);
        if (c->gather_pending) WAITEV(c, s.stream, s.ev_gathered);
        if (timed) RECORD(c, s.ev_g2[slot], s.stream);
        LAMCHK(I::launch_gemv(c, s, (const TV *)s.p, (TV *)s.Ap, s.part_gemv, s.sc, 2, lo, hi, &fa));
        if (timed) RECORD(c, s.ev_g3[slot], s.stream);
    } else {
        if (c->gather_pending) WAITEV(c, s.stream, s.ev_gathered);
        if (timed) RECORD(c, s.ev_g0[slot], s.stream);
        LAMCHK(I::launch_gemv(c, s, (const TV *)s.p, (TV *)s.Ap, s.part_gemv, s.sc, 0, 0, 0, &fa));
        if (timed) RECORD(c, s.ev_g1[slot], s.stream);
    }
    return reduce_post(c, s, false, true, true, fin_a);
}

template <typename I>
int phase_xr(lam_hip_ctx *c, ShardBase &s, int k, double rel_error)
{
    using TV = typename ImplTraits<I>::TV;
    const bool fin_a = producer_reduces(c, false), fin_b = producer_reduces(c, true);
    LAMCHK(reduce_wait(c, s, false));
    const double *red; int nred;
    red_source(c, s, false, true, fin_a, &red, &nred);
    if (c->fuse_active) {
        // one shard: phases B and C in ONE launch; the r.r total travels through the context's own mailbox
        const unsigned long long seq = c->seq_base + (unsigned)k;
        c->seq_span = std::max<uint64_t>(c->seq_span, (uint64_t)k + 1);
        Finalize fr = no_finalize(c);
        fr.active = 1; fr.seq = seq;                 // one shard: the total goes straight to the broadcast slot
        BlockCounts nb;
        for (auto &v : nb.n) v = 0;
        hipLaunchKernelGGL((update_fused_kernel<TV>), dim3(s.vec_blocks + 1), dim3(kBlock), 0, s.stream, red, nred, s.sc, k, rel_error,
                           (const TV *)s.p + s.row0, (const TV *)s.Ap, (TV *)s.x, (TV *)s.r, s.nrows, s.part_vec, s.vec_blocks, fr,
                           MailWait{nullptr, 0, 0, nullptr}, MailWait{s.mail->rr, 1, seq, c->direct_err}, s.bcast, plist_p(c), s.row0,
                           (volatile int *)s.host_flags, no_post(), (const Mail *)s.mail, nb);
        LAUNCHED(c);
        return 0;
    }
    const Finalize fb = make_finalize(c, s, true);
    hipLaunchKernelGGL((update_xr_kernel<TV>), dim3(s.vec_blocks + (fb.active ? 1 : 0)), dim3(kBlock), 0, s.stream, red, nred,
                       s.sc, k, (const TV *)s.p + s.row0, (const TV *)s.Ap, (TV *)s.x, (TV *)s.r, s.nrows, s.part_vec, fb,
                       MailWait{nullptr, 0, 0, nullptr});
    LAUNCHED(c);
    return reduce_post(c, s, true, false, true, fin_b);
}

template <typename I>
int phase_p(lam_hip_ctx *c, ShardBase &s, int k, double rel_error)
{
    using TV = typename ImplTraits<I>::TV;
    if (c->fuse_active) return 0;
    const bool fin_b = producer_reduces(c, true);
    LAMCHK(reduce_wait(c, s, true));
    const double *red; int nred;
    red_source(c, s, true, false, fin_b, &red, &nred);
    hipLaunchKernelGGL((update_p_kernel<TV>), dim3(s.vec_blocks), dim3(kBlock), 0, s.stream, red, nred, s.sc, k,
                       rel_error, (const TV *)s.r, (const TV *)s.p + s.row0, plist_p(c), s.row0, s.nrows,
                       (volatile int *)s.host_flags, MailWait{nullptr, 0, 0, nullptr}, no_post());
    LAUNCHED(c);
    return gather_post(c, s);
}

int enqueue_iteration(lam_hip_ctx *c, int k, double rel_error, int slot)
{
    if (c->cg_direct) return enqueue_iteration_direct(c, k, rel_error, slot);
    if (c->cg_exchange1) return enqueue_iteration_exchange1(c, k, rel_error, slot);
    return dispatch(c, [&](auto impl) -> int {
        using I = decltype(impl);
        for (auto &s : c->sh) {
            LAMCHK(set_dev(c, s));
            LAMCHK(phase_gemv<I>(c, s, k, slot));
        }
        c->gather_pending = false;
        LAMCHK(hub_join(c, 0));
        for (auto &s : c->sh) {
            LAMCHK(set_dev(c, s));
            LAMCHK(phase_xr<I>(c, s, k, rel_error));
        }
        if (c->fuse_active) return 0;
        LAMCHK(hub_join(c, 1));
        for (auto &s : c->sh) {
            LAMCHK(set_dev(c, s));
            LAMCHK(phase_p<I>(c, s, k, rel_error));
        }
        if (c->rank_mode) return gather_p_rank(c);
        LAMCHK(hub_join(c, 2));
        for (auto &s : c->sh) {
            LAMCHK(set_dev(c, s));
            LAMCHK(gather_wait(c, s));
        }
        return 0;
    });
}

// ---- the host's view of the iteration's progress -----------------------------------------------------------------
struct Progress { int iters, stop_at; };
Progress read_progress(const ShardBase &s)
{
    const unsigned long long v = *reinterpret_cast<volatile unsigned long long *>(s.host_flags);
    return {(int)(unsigned)(v & 0xffffffffull), (int)(unsigned)(v >> 32)};
}

// Wait until iteration `target` has made its stop decision (update_p_kernel's progress word in pinned memory), or
// some iteration has stopped, or a bounded in-kernel wait has expired.  No event per iteration is involved: an event
// record is a marker packet between the iteration's kernels.  A stream error (a fault, a lost device) ends the wait.
// The host does not burn a core while it waits (round 4; in rank mode that was one spinning core per GPU next to RCCL's
// proxy threads): after a short spin -- an iteration that is about to report costs nothing -- it SLEEPS between polls,
// for a quarter of the iteration time observed so far (clamped to 20 us .. 1 ms).  The host enqueues kLag iterations
// ahead of the one it awaits, so a wake-up that comes a whole iteration late is still free: the queue never drains.
int await_progress(lam_hip_ctx *c, ShardBase &s0, int target, Progress *out)
{
    unsigned polls = 0;
    double t_query = 0.0;
    for (;;) {
        const Progress pr = read_progress(s0);
        if (pr.iters >= target || pr.stop_at != 0 || *(volatile int *)c->direct_err != 0) {
            // iteration-time estimate: progress made since the previous successful wait / time since then
            const double t = now_s();
            if (c->prog_t > 0.0 && pr.iters > c->prog_iter && pr.stop_at == 0) {
                const double per = (t - c->prog_t) / (double)(pr.iters - c->prog_iter);
                c->iter_est_s = c->iter_est_s > 0.0 ? 0.75 * c->iter_est_s + 0.25 * per : per;
            }
            c->prog_t = t;
            c->prog_iter = pr.iters;
            *out = pr;
            return 0;
        }
        if (++polls <= 64u) { __builtin_ia32_pause(); continue; }
        const double t = now_s();
        if (t_query == 0.0) t_query = t;
        if (t - t_query > 2e-3) {                       // liveness: look at the stream every 2 ms of waiting
            t_query = t;
            const hipError_t e = hipStreamQuery(s0.stream);
            if (e == hipSuccess) {
                // everything enqueued has run: the word is final (it may have been written since the read above)
                const Progress again = read_progress(s0);
                if (again.iters >= target || again.stop_at != 0 || *(volatile int *)c->direct_err != 0) { *out = again; return 0; }
                return fail(c, LAM_HIP_EHIP, "iteration %d was enqueued but never reported (progress word at %d)", target, again.iters);
            }
            if (e != hipErrorNotReady) return fail(c, LAM_HIP_EHIP, "stream error while iterating: %s", hipGetErrorString(e));
        }
        const double nap = std::min(1e-3, std::max(20e-6, 0.25 * c->iter_est_s));
        struct timespec ts = {0, (long)(nap * 1e9)};
        (void)nanosleep(&ts, nullptr);
    }
}

#ifdef LAM_TUNING_VARIANTS
// TUNING BUILD ONLY (option "host_threads").
// Host barrier of the per-shard enqueue threads.  wait(flags) returns the OR of the flags every thread brought to
// THIS barrier, so all threads leave the loop at the same barrier (a flag raised between two barriers is seen by
// everybody at the next one, by nobody before).
struct HostBarrier {
    explicit HostBarrier(int n_) : n(n_) {}
    const int n;
    std::atomic<int> count{0}, gen{0}, acc{0};
    int result[2] = {0, 0};
    int wait(int flags)
    {
        if (flags) acc.fetch_or(flags, std::memory_order_acq_rel);
        const int g = gen.load(std::memory_order_acquire);
        if (count.fetch_add(1, std::memory_order_acq_rel) + 1 == n) {
            result[(g + 1) & 1] = acc.exchange(0, std::memory_order_acq_rel);
            count.store(0, std::memory_order_relaxed);
            gen.store(g + 1, std::memory_order_release);
        } else {
            unsigned spins = 0;
            while (gen.load(std::memory_order_acquire) == g) {
                if (++spins > 20000u) sched_yield(); else __builtin_ia32_pause();
            }
        }
        return result[(g + 1) & 1];
    }
};
#endif

}  // namespace

// =================================================================================================
// C ABI
// =================================================================================================
extern "C" {

int lam_hip_abi_version(void) { return LAM_HIP_ABI_VERSION; }

#ifndef LAM_SOURCE_ID
#define LAM_SOURCE_ID "unknown"
#endif
const char *lam_hip_build_id(void) { return LAM_SOURCE_ID; }

int lam_hip_device_count(int *count)
{
    if (!count) return LAM_HIP_EINVAL;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) { *count = 0; return fail(nullptr, LAM_HIP_ENODEV, "hipGetDeviceCount: %s", hipGetErrorString(e)); }
    *count = n;
    return 0;
}

const char *lam_hip_last_error(const lam_hip_ctx *ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

int lam_hip_create(lam_hip_ctx **out, int dtype, int n_shards, const int *device_ids)
{
    if (!out) return LAM_HIP_EINVAL;
    *out = nullptr;
    if (dtype < LAM_HIP_F64 || dtype > LAM_HIP_BF16) return fail(nullptr, LAM_HIP_EINVAL, "bad dtype %d", dtype);
    if (n_shards < 1 || n_shards > kMaxShards) return fail(nullptr, LAM_HIP_EINVAL, "n_shards must be 1..%d", kMaxShards);
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(nullptr, LAM_HIP_ENODEV, "no usable HIP device; this library has no CPU path");
    std::unique_ptr<lam_hip_ctx> c(new lam_hip_ctx);
    c->dtype = dtype;
    c->total_shards = n_shards;
    c->sh.resize(n_shards);
    for (int q = 0; q < n_shards; q++) {
        c->sh[q].index = q;
        c->sh[q].dev = device_ids ? device_ids[q] : q % ndev;
    }
    // Default exchange of a one-process context with several shards: gather-Ap -- ONE event join per iteration instead of
    // three (host time to enqueue an iteration at 8 shards 0.15 ms against 0.59 ms, profiles/r04_host_enqueue_cost.txt),
    // same HIP-guaranteed ordering; sizes it cannot take (N % shards != 0) run on exchange 0 ("exchange_effective" tells).
    // The join goes through shard 0's stream when there are more than two shards (2(P-1)+1 runtime calls instead of P(P-1)).
    if (n_shards > 1) c->opt_exchange = 1;
    c->opt_join = n_shards > 2 ? 1 : 0;
    c->opt_exchange = exchange_from_env(c->opt_exchange);
    int rc = create_common(c.get());
    if (rc != 0) { abandon(c.get()); return rc; }
    *out = c.release();
    return 0;
}

int lam_hip_get_unique_id(void *unique_id_out)
{
    if (!unique_id_out) return LAM_HIP_EINVAL;
    static_assert(sizeof(ncclUniqueId) == LAM_HIP_UNIQUE_ID_BYTES, "unique id size");
    ncclUniqueId id;
    ncclResult_t r = ncclGetUniqueId(&id);
    if (r != ncclSuccess) return fail(nullptr, LAM_HIP_ERCCL, "ncclGetUniqueId: %s", ncclGetErrorString(r));
    memcpy(unique_id_out, &id, sizeof id);
    return 0;
}

int lam_hip_create_rank(lam_hip_ctx **out, int dtype, int device_id, int rank, int nranks, const void *unique_id)
{
    if (!out) return LAM_HIP_EINVAL;
    *out = nullptr;
    if (dtype < LAM_HIP_F64 || dtype > LAM_HIP_BF16) return fail(nullptr, LAM_HIP_EINVAL, "bad dtype %d", dtype);
    if (nranks < 1 || nranks > kMaxShards || rank < 0 || rank >= nranks)
        return fail(nullptr, LAM_HIP_EINVAL, "bad rank %d / nranks %d", rank, nranks);
    if (nranks > 1 && !unique_id) return fail(nullptr, LAM_HIP_EINVAL, "unique_id required when nranks > 1");
    std::unique_ptr<lam_hip_ctx> c(new lam_hip_ctx);
    c->dtype = dtype;
    c->total_shards = nranks;
    c->rank = rank;
    c->nranks = nranks;
    // LAM_HIP_FORCE_RCCL=1 keeps the RCCL exchange even for a single rank (a 1-rank communicator):
    // lets a one-GPU box exercise every collective call of the multi-rank path
    const char *force = getenv("LAM_HIP_FORCE_RCCL");
    const bool forced = force && *force && strcmp(force, "0") != 0;
    c->rank_mode = nranks > 1 || forced;
    c->opt_exchange = exchange_from_env(c->opt_exchange);   // default exchange for this context
    c->sh.resize(1);
    c->sh[0].index = rank;
    c->sh[0].dev = device_id;
    int rc = create_common(c.get());
    if (rc != 0) { abandon(c.get()); return rc; }
    if (c->rank_mode) {
        const double t0 = now_s();
        ncclUniqueId id;
        // a 1-rank communicator (LAM_HIP_FORCE_RCCL) has nobody to share an id with: make one here, whatever the
        // caller passed (the C++ class hands over an all-zero buffer when the launch has a single rank)
        if (unique_id && nranks > 1) memcpy(&id, unique_id, sizeof id);
        else if (ncclGetUniqueId(&id) != ncclSuccess) {
            abandon(c.get());
            return fail(nullptr, LAM_HIP_ERCCL, "ncclGetUniqueId failed");
        }
        (void)hipSetDevice(device_id);
        // RCCL writes a five-line version banner to STDOUT when a communicator is created.  Callers whose stdout
        // is a protocol (this package's drivers: the CSV line) set LAM_HIP_QUIET_RCCL=1 and get it on stderr instead
        // for the duration of the call; by default the library leaves the process's descriptors alone
        ncclResult_t r;
        {
            StdoutToStderr quiet;
            r = ncclCommInitRank(&c->comm, nranks, id, rank);
        }
        if (r != ncclSuccess) {
            abandon(c.get());
            return fail(nullptr, LAM_HIP_ERCCL, "ncclCommInitRank(rank %d of %d): %s", rank, nranks, ncclGetErrorString(r));
        }
        c->t_comm_init = now_s() - t0;
    }
    *out = c.release();
    return 0;
}

void lam_hip_destroy(lam_hip_ctx *c)
{
    if (!c) return;
    for (auto &s : c->sh) {
        (void)hipSetDevice(s.dev);
        if (s.stream) (void)hipStreamSynchronize(s.stream);
    }
    close_direct(c);
    if (c->agree_buf) (void)hipFree(c->agree_buf);
    for (auto &s : c->sh) {
        if (hipSetDevice(s.dev) != hipSuccess) { (void)hipGetLastError(); continue; }
        if (s.mail) { (void)hipFree(s.mail); s.mail = nullptr; }
        if (s.bcast) { (void)hipFree(s.bcast); s.bcast = nullptr; }
    }
    if (c->direct_err) (void)hipHostFree(c->direct_err);
    if (c->comm) (void)ncclCommDestroy(c->comm);
    if (!c->sh.empty() && hipSetDevice(c->sh[0].dev) == hipSuccess) {
        if (c->persist_bc) (void)hipFree(c->persist_bc);
        if (c->persist_ticks) (void)hipFree(c->persist_ticks);
        if (c->persist_ticks_host) (void)hipHostFree(c->persist_ticks_host);
    }
    release_hub(c);
    for (auto &s : c->sh) {
        free_shard(s);
        release_handles(s);
    }
    delete c;
}

int lam_hip_set_problem(lam_hip_ctx *c, uint64_t n)
{
    if (!c) return LAM_HIP_EINVAL;
    if (n == 0) return fail(c, LAM_HIP_EINVAL, "n must be > 0");
    if (n < (uint64_t)c->total_shards) return fail(c, LAM_HIP_EINVAL, "n (%llu) smaller than the number of shards", (unsigned long long)n);
    c->n = n;
    c->problem_gen++;              // peers' mappings of the old p replica are stale from here on
    c->have_problem = c->have_matrix = c->have_rhs = c->cg_ready = false;
    const size_t ea = c->esz_a(), ev = c->esz_v();
    for (auto &s : c->sh) {
        // The matrix allocation is grow-only: a context that loads a smaller system after a large one keeps
        // streaming from the pages it already owns instead of handing 34 GB back to the runtime and carving
        // a new block out of whatever that leaves behind ("allocation history").
        // Option "reuse_matrix" = 0 restores free + hipMalloc.
        free_shard(s, c->opt_reuse_matrix != 0);
        partition(n, c->total_shards, s.index, &s.row0, &s.nrows);
        LAMCHK(set_dev(c, s));
        const size_t needA = std::max<size_t>(16, s.nrows * n * ea);
        if (s.A != nullptr && s.A_capacity < needA) { (void)hipFree(s.A); s.A = nullptr; s.A_capacity = 0; }
        if (s.A == nullptr) {
            HIPCHK(c, hipMalloc(&s.A, needA));
            s.A_capacity = needA;
        }
        HIPCHK(c, hipMalloc(&s.p, n * ev + 16));
        HIPCHK(c, hipMalloc(&s.tmp, n * ev + 16));
        void **vecs[] = {&s.Ap, &s.x, &s.r, &s.b};
        for (auto v : vecs) HIPCHK(c, hipMalloc(v, s.nrows * ev + 16));
        s.gemv_blocks = dispatch(c, [&](auto impl) -> int { return decltype(impl)::gemv_grid(c, s.nrows); });
        // worst case over kernel variants (generic kernel: 4 rows per workgroup)
        const int gemv_blocks_max = (int)s.nrows + 1;
        s.vec_blocks = vec_grid(s.nrows);
        HIPCHK(c, hipMalloc((void **)&s.part_gemv, sizeof(double) * (size_t)gemv_blocks_max));
        s.part_gemv_cap = gemv_blocks_max;
        HIPCHK(c, hipMalloc((void **)&s.part_vec, sizeof(double) * kVecBlocksMax));
        HIPCHK(c, hipMalloc((void **)&s.part_aux, sizeof(double) * kVecBlocksMax));
        HIPCHK(c, hipMalloc((void **)&s.gather_a, sizeof(double) * kMaxShards));
        HIPCHK(c, hipMalloc((void **)&s.gather_b, sizeof(double) * kMaxShards));
        if (c->rank_mode || c->total_shards > 1) {
            // gather-Ap exchange (option exchange = 1): full-length r, and the records [Ap slice | p.Ap partial] of all shards
            HIPCHK(c, hipMalloc(&s.r_full, n * ev + 16));
            s.ap_gather_bytes = ((size_t)c->total_shards * ((n / (uint64_t)c->total_shards + 1) * ev + 8) + 16 + 255) / 256 * 256;
            HIPCHK(c, hipMalloc(&s.ap_gather, s.ap_gather_bytes * (c->rank_mode ? 1 : 2)));
        }
        HIPCHK(c, hipMalloc((void **)&s.sc, sizeof(CgScalars)));
        HIPCHK(c, hipHostMalloc((void **)&s.sc_host, sizeof(CgScalars), hipHostMallocDefault));
        HIPCHK(c, hipHostMalloc((void **)&s.host_flags, 64, hipHostMallocDefault));
        s.host_flags[0] = s.host_flags[1] = 0;
        HIPCHK(c, hipMemsetAsync(s.sc, 0, sizeof(CgScalars), s.stream));
        HIPCHK(c, hipMemsetAsync(s.gather_a, 0, sizeof(double) * kMaxShards, s.stream));
        HIPCHK(c, hipMemsetAsync(s.gather_b, 0, sizeof(double) * kMaxShards, s.stream));
        HIPCHK(c, hipMemsetAsync(s.p, 0, n * ev, s.stream));
        memset(s.sc_host, 0, sizeof(CgScalars));
    }
    LAMCHK(sync_all(c));
    c->have_problem = true;
    return 0;
}

int lam_hip_partition(uint64_t n, int num_shards, int shard, uint64_t *row0, uint64_t *nrows)
{
    if (!row0 || !nrows || num_shards < 1 || shard < 0 || shard >= num_shards) return LAM_HIP_EINVAL;
    partition(n, num_shards, shard, row0, nrows);
    return 0;
}

int lam_hip_n(const lam_hip_ctx *c, uint64_t *n)
{
    if (!c || !n) return LAM_HIP_EINVAL;
    *n = c->n;
    return 0;
}

int lam_hip_num_shards(const lam_hip_ctx *c, int *total, int *local)
{
    if (!c) return LAM_HIP_EINVAL;
    if (total) *total = c->total_shards;
    if (local) *local = (int)c->sh.size();
    return 0;
}

int lam_hip_get_partition(const lam_hip_ctx *c, int shard, uint64_t *row0, uint64_t *nrows)
{
    if (!c || !row0 || !nrows || shard < 0 || shard >= c->total_shards || !c->have_problem) return LAM_HIP_EINVAL;
    partition(c->n, c->total_shards, shard, row0, nrows);
    return 0;
}

static int rows_xfer(lam_hip_ctx *c, uint64_t row0, uint64_t nrows, void *host, bool upload)
{
    if (!c || (!host && nrows)) return LAM_HIP_EINVAL;
    if (!c->have_problem) return fail(c, LAM_HIP_ESTATE, "call lam_hip_set_problem first");
    if (row0 + nrows > c->n) return fail(c, LAM_HIP_EINVAL, "rows [%llu,+%llu) outside the matrix", (unsigned long long)row0, (unsigned long long)nrows);
    const size_t ea = c->esz_a();
    const size_t eh = c->dtype == LAM_HIP_BF16 ? 4 : ea;   // host element size (bf16 travels as float)
    uint64_t covered = 0;
    for (auto &s : c->sh) {
        const uint64_t lo = std::max(row0, s.row0), hi = std::min(row0 + nrows, s.row0 + s.nrows);
        if (lo >= hi) continue;
        LAMCHK(set_dev(c, s));
        const uint64_t cnt = (hi - lo) * c->n;
        char *hptr = (char *)host + (lo - row0) * c->n * eh;
        char *dptr = (char *)s.A + (lo - s.row0) * c->n * ea;
        if (c->dtype == LAM_HIP_BF16) {
            // stage through a device float buffer in chunks of rows
            const uint64_t chunk_rows = std::max<uint64_t>(1, (64ull << 20) / (c->n * 4));
            DevBuf stage_buf;
            HIPCHK(c, hipMalloc(&stage_buf.p, chunk_rows * c->n * 4));
            float *stage = stage_buf.as<float>();
            for (uint64_t r = lo; r < hi; r += chunk_rows) {
                const uint64_t nr = std::min(chunk_rows, hi - r), ne = nr * c->n;
                char *hp = (char *)host + (r - row0) * c->n * 4;
                __hip_bfloat16 *dp = (__hip_bfloat16 *)s.A + (r - s.row0) * c->n;
                if (upload) {
                    HIPCHK(c, hipMemcpyAsync(stage, hp, ne * 4, hipMemcpyHostToDevice, s.stream));
                    hipLaunchKernelGGL((f32_to_bf16_kernel<float>), dim3(1024), dim3(kBlock), 0, s.stream, stage, dp, ne);
                    HIPCHK(c, hipGetLastError());
                    HIPCHK(c, hipStreamSynchronize(s.stream));
                } else {
                    std::vector<unsigned short> tmp(ne);
                    HIPCHK(c, hipMemcpyAsync(tmp.data(), dp, ne * 2, hipMemcpyDeviceToHost, s.stream));
                    HIPCHK(c, hipStreamSynchronize(s.stream));
                    float *fp = (float *)hp;
                    for (uint64_t i = 0; i < ne; i++) { unsigned u = ((unsigned)tmp[i]) << 16; memcpy(&fp[i], &u, 4); }
                }
            }
        } else if (upload && c->opt_upload_staging) {
            // Option "upload_staging": pipeline through two pinned buffers -- a host memcpy into one
            // while the DMA engine drains the other.  Measured against the default (the runtime pins the
            // caller's pages and DMAs from them directly): see DESIGN.md section 6, f1.
            const uint64_t stage_elems = (64ull << 20) / ea;
            PinnedBuf pin[2];
            hipEvent_t evs[2] = {nullptr, nullptr};
            struct EvGuard { hipEvent_t *e; ~EvGuard() { for (int i = 0; i < 2; i++) if (e[i]) (void)hipEventDestroy(e[i]); } } evg{evs};
            for (int i = 0; i < 2; i++) {
                HIPCHK(c, hipHostMalloc(&pin[i].p, stage_elems * ea, hipHostMallocDefault));
                HIPCHK(c, hipEventCreateWithFlags(&evs[i], hipEventDisableTiming));
            }
            int b = 0;
            for (uint64_t off = 0; off < cnt; off += stage_elems, b ^= 1) {
                const uint64_t ne = std::min(stage_elems, cnt - off);
                HIPCHK(c, hipEventSynchronize(evs[b]));                    // the DMA that last used this buffer is done
                memcpy(pin[b].p, hptr + off * ea, ne * ea);
                HIPCHK(c, hipMemcpyAsync(dptr + off * ea, pin[b].p, ne * ea, hipMemcpyHostToDevice, s.stream));
                HIPCHK(c, hipEventRecord(evs[b], s.stream));
            }
            HIPCHK(c, hipStreamSynchronize(s.stream));
        } else {
            // chunked so that a single call never exceeds 2^31 elements (the reference's int-count trap)
            const uint64_t chunk = 1ull << 28;
            for (uint64_t off = 0; off < cnt; off += chunk) {
                const uint64_t ne = std::min(chunk, cnt - off);
                if (upload) HIPCHK(c, hipMemcpyAsync(dptr + off * ea, hptr + off * ea, ne * ea, hipMemcpyHostToDevice, s.stream));
                else HIPCHK(c, hipMemcpyAsync(hptr + off * ea, dptr + off * ea, ne * ea, hipMemcpyDeviceToHost, s.stream));
            }
            HIPCHK(c, hipStreamSynchronize(s.stream));
        }
        covered += hi - lo;
    }
    if (covered != nrows)
        return fail(c, LAM_HIP_EINVAL, "rows [%llu,+%llu) are not all owned by this process", (unsigned long long)row0, (unsigned long long)nrows);
    if (upload) { c->have_matrix = true; c->cg_ready = false; }
    return 0;
}

int lam_hip_upload_rows(lam_hip_ctx *c, uint64_t row0, uint64_t nrows, const void *host_rows)
{
    return rows_xfer(c, row0, nrows, const_cast<void *>(host_rows), true);
}
int lam_hip_download_rows(lam_hip_ctx *c, uint64_t row0, uint64_t nrows, void *host_rows)
{
    return rows_xfer(c, row0, nrows, host_rows, false);
}

int lam_hip_generate_tridiag(lam_hip_ctx *c)
{
    if (!c) return LAM_HIP_EINVAL;
    if (!c->have_problem) return fail(c, LAM_HIP_ESTATE, "call lam_hip_set_problem first");
    LAMCHK(dispatch(c, [&](auto impl) -> int {
        using TA = typename ImplTraits<decltype(impl)>::TA;
        for (auto &s : c->sh) {
            if (s.nrows == 0) continue;
            LAMCHK(set_dev(c, s));
            hipLaunchKernelGGL((gen_tridiag_kernel<TA>), dim3(4096), dim3(kBlock), 0, s.stream, (TA *)s.A, s.row0, s.nrows, c->n);
            HIPCHK(c, hipGetLastError());
        }
        return 0;
    }));
    LAMCHK(sync_all(c));
    c->have_matrix = true; c->cg_ready = false;
    return 0;
}

int lam_hip_generate_random_spd(lam_hip_ctx *c, uint64_t seed, double cond)
{
    if (!c) return LAM_HIP_EINVAL;
    if (!c->have_problem) return fail(c, LAM_HIP_ESTATE, "call lam_hip_set_problem first");
    if (!(cond >= 1.0)) return fail(c, LAM_HIP_EINVAL, "cond must be >= 1");
    LAMCHK(dispatch(c, [&](auto impl) -> int {
        using TA = typename ImplTraits<decltype(impl)>::TA;
        for (auto &s : c->sh) {
            if (s.nrows == 0) continue;
            LAMCHK(set_dev(c, s));
            hipLaunchKernelGGL((gen_random_spd_kernel<TA>), dim3(4096), dim3(kBlock), 0, s.stream, (TA *)s.A, s.row0, s.nrows, c->n, seed, cond);
            HIPCHK(c, hipGetLastError());
        }
        return 0;
    }));
    LAMCHK(sync_all(c));
    c->have_matrix = true; c->cg_ready = false;
    return 0;
}

int lam_hip_generate_spectrum_spd(lam_hip_ctx *c, const double *eig, const double *v, int k)
{
    if (!c || !eig || (k > 0 && !v) || k < 0) return LAM_HIP_EINVAL;
    if (!c->have_problem) return fail(c, LAM_HIP_ESTATE, "call lam_hip_set_problem first");
    if (c->dtype == LAM_HIP_BF16) return fail(c, LAM_HIP_EINVAL, "the spectrum generator works in the storage type: fp64 / fp32 only");
    const uint64_t n = c->n;
    for (uint64_t i = 0; i < n; i++)
        if (!(eig[i] > 0.0)) return fail(c, LAM_HIP_EINVAL, "eigenvalue %llu is not positive", (unsigned long long)i);
    const bool f64 = c->dtype == LAM_HIP_F64;
    const size_t ev = c->esz_v();
    // host vectors in the context's vector type (what lam_hip_gemv and the uploads take)
    std::vector<double> vd(n), ud(n), wd(n);
    std::vector<float> vf(f64 ? 0 : n), uf(f64 ? 0 : n), wf(f64 ? 0 : n);
    auto to_dev = [&](const std::vector<double> &src, std::vector<float> &tmp) -> const void * {
        if (f64) return src.data();
        for (uint64_t i = 0; i < n; i++) tmp[i] = (float)src[i];
        return tmp.data();
    };
    auto upload = [&](void *ShardBase::*dst, const void *host) -> int {
        for (auto &s : c->sh) {
            LAMCHK(set_dev(c, s));
            HIPCHK(c, hipMemcpyAsync(s.*dst, host, n * ev, hipMemcpyHostToDevice, s.stream));
            HIPCHK(c, hipStreamSynchronize(s.stream));
        }
        return 0;
    };
    // 0. A = diag(eig)
    for (uint64_t i = 0; i < n; i++) vd[i] = eig[i];
    LAMCHK(upload(&ShardBase::tmp, to_dev(vd, vf)));
    LAMCHK(dispatch(c, [&](auto impl) -> int {
        using TA = typename ImplTraits<decltype(impl)>::TA;
        using TV = typename ImplTraits<decltype(impl)>::TV;
        for (auto &s : c->sh) {
            if (s.nrows == 0) continue;
            LAMCHK(set_dev(c, s));
            hipLaunchKernelGGL((gen_diag_kernel<TA, TV>), dim3(4096), dim3(kBlock), 0, s.stream, (TA *)s.A, s.row0, s.nrows, n, (const TV *)s.tmp);
            HIPCHK(c, hipGetLastError());
        }
        return 0;
    }));
    LAMCHK(sync_all(c));
    c->have_matrix = true; c->cg_ready = false;
    // 1. one two-sided reflection per vector
    for (int j = 0; j < k; j++) {
        const double *vj = v + (size_t)j * n;
        double vv = 0.0;
        for (uint64_t i = 0; i < n; i++) { vd[i] = vj[i]; vv += vj[i] * vj[i]; }
        if (!(vv > 0.0)) return fail(c, LAM_HIP_EINVAL, "reflector %d is the zero vector", j);
        const double tau = 2.0 / vv;
        // w = A v through the product GEMV (a collective in rank mode: every rank receives the full vector)
        const void *vdev = to_dev(vd, vf);
        LAMCHK(lam_hip_gemv(c, vdev, f64 ? (void *)wd.data() : (void *)wf.data()));
        if (!f64) for (uint64_t i = 0; i < n; i++) wd[i] = wf[i];
        double alpha = 0.0;
        for (uint64_t i = 0; i < n; i++) alpha += vj[i] * wd[i];
        for (uint64_t i = 0; i < n; i++) { ud[i] = wd[i] - 0.5 * tau * alpha * vj[i]; vd[i] = tau * vj[i]; }
        LAMCHK(upload(&ShardBase::tmp, to_dev(vd, vf)));        // tau v
        LAMCHK(upload(&ShardBase::p, to_dev(ud, uf)));          // u (p is rewritten by cg_init anyway)
        LAMCHK(dispatch(c, [&](auto impl) -> int {
            using TA = typename ImplTraits<decltype(impl)>::TA;
            using TV = typename ImplTraits<decltype(impl)>::TV;
            for (auto &s : c->sh) {
                if (s.nrows == 0) continue;
                LAMCHK(set_dev(c, s));
                hipLaunchKernelGGL((rank2_update_kernel<TA, TV>), dim3(4096), dim3(kBlock), 0, s.stream, (TA *)s.A, s.row0, s.nrows, n,
                                   (const TV *)s.tmp, (const TV *)s.p);
                HIPCHK(c, hipGetLastError());
            }
            return 0;
        }));
        LAMCHK(sync_all(c));
    }
    c->have_matrix = true; c->cg_ready = false;
    return 0;
}

int lam_hip_set_rhs(lam_hip_ctx *c, const void *b_host)
{
    if (!c || !b_host) return LAM_HIP_EINVAL;
    if (!c->have_problem) return fail(c, LAM_HIP_ESTATE, "call lam_hip_set_problem first");
    const size_t ev = c->esz_v();
    for (auto &s : c->sh) {
        LAMCHK(set_dev(c, s));
        HIPCHK(c, hipMemcpyAsync(s.b, (const char *)b_host + s.row0 * ev, s.nrows * ev, hipMemcpyHostToDevice, s.stream));
        HIPCHK(c, hipStreamSynchronize(s.stream));
    }
    c->have_rhs = true; c->cg_ready = false;
    return 0;
}

int lam_hip_get_rhs(lam_hip_ctx *c, void *b_host)
{
    if (!c || !b_host) return LAM_HIP_EINVAL;
    if (!c->have_rhs) return fail(c, LAM_HIP_ESTATE, "rhs not set");
    if (c->rank_mode) return fail(c, LAM_HIP_EINVAL, "lam_hip_get_rhs is for single-process contexts");
    const size_t ev = c->esz_v();
    for (auto &s : c->sh) {
        LAMCHK(set_dev(c, s));
        HIPCHK(c, hipMemcpyAsync((char *)b_host + s.row0 * ev, s.b, s.nrows * ev, hipMemcpyDeviceToHost, s.stream));
        HIPCHK(c, hipStreamSynchronize(s.stream));
    }
    return 0;
}

static int gen_rhs(lam_hip_ctx *c, int random, uint64_t seed, double value)
{
    if (!c) return LAM_HIP_EINVAL;
    if (!c->have_problem) return fail(c, LAM_HIP_ESTATE, "call lam_hip_set_problem first");
    LAMCHK(dispatch(c, [&](auto impl) -> int {
        using TV = typename ImplTraits<decltype(impl)>::TV;
        for (auto &s : c->sh) {
            LAMCHK(set_dev(c, s));
            hipLaunchKernelGGL((gen_rhs_kernel<TV>), dim3(s.vec_blocks), dim3(kBlock), 0, s.stream, (TV *)s.b, s.row0, s.nrows, random, seed, value);
            HIPCHK(c, hipGetLastError());
        }
        return 0;
    }));
    LAMCHK(sync_all(c));
    c->have_rhs = true; c->cg_ready = false;
    return 0;
}
int lam_hip_generate_rhs(lam_hip_ctx *c, double value) { return gen_rhs(c, 0, 0, value); }
int lam_hip_generate_random_rhs(lam_hip_ctx *c, uint64_t seed) { return gen_rhs(c, 1, seed, 0.0); }

int lam_hip_cg_init(lam_hip_ctx *c)
{
    if (!c) return LAM_HIP_EINVAL;
    if (!c->have_matrix || !c->have_rhs) return fail(c, LAM_HIP_ESTATE, "matrix and rhs must be set before cg_init");
    LAMCHK(do_cg_init(c));
    LAMCHK(settle_gather(c));
    LAMCHK(sync_all(c));
    for (auto &s : c->sh) s.host_flags[0] = s.host_flags[1] = 0;
    return 0;
}

// GEMV device time of the iteration that used ring slot `slot` (shard 0), if that iteration was timed
static void harvest_gemv_time(ShardBase &s0, int slot, double *ms_sum, int *samples)
{
    if (!s0.timed_slot[slot]) return;
    s0.timed_slot[slot] = false;
    float ms = 0.f, ms2 = 0.f;
    if (hipEventElapsedTime(&ms, s0.ev_g0[slot], s0.ev_g1[slot]) != hipSuccess) { (void)hipGetLastError(); return; }
    if (s0.split_slot[slot] && hipEventElapsedTime(&ms2, s0.ev_g2[slot], s0.ev_g3[slot]) != hipSuccess) { (void)hipGetLastError(); return; }
    *ms_sum += ms + ms2;
    (*samples)++;
}

// The lag rule (every enqueue loop uses it).  Before enqueueing iteration k the host makes sure iteration k - kLag
// has reported, then looks at the stopping iteration the update kernel left in pinned memory.  Later iterations may
// or may not have finished by now -- that depends on how far this rank's GPU is ahead of its host -- so the value
// only counts if it names an iteration whose report has been AWAITED: stop_at <= k - kLag.  A stop at iteration j
// is therefore acted on at k = j + kLag on every rank, whatever the timing: all ranks enqueue the same number of
// (no-op) iterations and their collectives stay matched.  (The reference broadcasts the decision instead:
// MPI_Bcast(&stop), ConjugateGradient_MultiGPUS_CUDA_NCCL.cu:404-407.)
// Returns 0 go on, 1 stop here, 2 a bounded in-kernel wait expired (reported after the final sync), < 0 error.
static int lag_check(lam_hip_ctx *c, ShardBase &s0, int k)
{
    Progress pr;
    LAMCHK(await_progress(c, s0, k - kLag, &pr));
    if (*(volatile int *)c->direct_err != 0) return 2;
    // LAM_HIP_DEBUG_LEVEL_STOP: test hook that restores the timing-dependent decision ("any stop seen so far") so
    // that the stream-ordered RCCL test double can be shown to catch the rank desynchronisation it causes
    // (tests/test_gpu_rank_mock.py).  Never set it otherwise.
    static const bool level_stop = getenv("LAM_HIP_DEBUG_LEVEL_STOP") != nullptr;
    return (pr.stop_at != 0 && (pr.stop_at <= k - kLag || level_stop)) ? 1 : 0;
}

#ifdef LAM_TUNING_VARIANTS
// TUNING BUILD ONLY (option "host_threads": 0.30 ms of host time per iteration at 8 shards where the plain loop takes 0.59 and the
// gather-Ap exchange 0.15 -- the runtime serialises much of it).
// One process, several shards: every shard is enqueued by a host thread of its own (the reference drives each device
// from its own OpenMP thread, ConjugateGradient_MultiGPUS_CUDA.cu:264-283,337-378).  With one thread for P shards an
// iteration costs the host 3P launches + ~3P event records + 3P(P-1) stream waits one after the other; here they are
// issued P-wide, with a host barrier between the phases (a stream wait must follow the record it refers to).
static int iterate_threaded(lam_hip_ctx *c, int iters, int k_first, double rel_error, int *enq_out, double *gemv_ms, int *gemv_samples)
{
    const int L = (int)c->sh.size();
    HostBarrier bar(L);
    std::vector<int> rcs(L, 0);
    int enq = 0;
    std::atomic<int> go{0};                         // 0: wait, 1: run, 2: cancelled (a thread could not be created)
    auto worker = [&](int q) {
        while (go.load(std::memory_order_acquire) == 0) sched_yield();
        if (go.load(std::memory_order_acquire) == 2) return;
        ShardBase &s = c->sh[q];
        int rc = set_dev(c, s);
        auto phase = [&](auto &&fn) {               // run one phase unless this thread has already failed
            if (rc == 0) rc = fn();
            return bar.wait(rc != 0 ? 1 : 0);
        };
        for (int i = 0; i < iters; i++) {
            const int k = k_first + i, slot = i % kLag;
            if (i >= kLag) {
                int flags = rc != 0 ? 1 : 0;
                if (q == 0 && rc == 0) {
                    const int d = lag_check(c, s, k);
                    if (d < 0) { rc = d; flags |= 1; }
                    else if (d != 0) flags |= 2;
                    else harvest_gemv_time(s, slot, gemv_ms, gemv_samples);
                }
                if (bar.wait(flags) != 0) break;
            }
            const double te = q == 0 ? now_s() : 0.0;
            // with the hub, thread 0 issues the join between two barriers (everybody's post before it, everybody's wait after it)
            auto join = [&](int which) {
                if (!hub_active(c)) return 0;
                if (q == 0 && rc == 0) rc = hub_join(c, which);
                return bar.wait(rc != 0 ? 1 : 0);
            };
            if (phase([&] { return dispatch(c, [&](auto impl) -> int { return phase_gemv<decltype(impl)>(c, s, k, slot); }); })) break;
            if (join(0)) break;
            if (phase([&] { return dispatch(c, [&](auto impl) -> int { return phase_xr<decltype(impl)>(c, s, k, rel_error); }); })) break;
            if (join(1)) break;
            if (phase([&] { return dispatch(c, [&](auto impl) -> int { return phase_p<decltype(impl)>(c, s, k, rel_error); }); })) break;
            if (join(2)) break;
            if (rc == 0) rc = gather_wait(c, s);       // refers to records issued before the last barrier: no barrier needed
            if (q == 0) { enq++; c->enqueue_ns += (uint64_t)((now_s() - te) * 1e9); }
        }
        rcs[q] = rc;
    };
    std::vector<std::thread> th;
    th.reserve(L);
    try {
        for (int q = 1; q < L; q++) th.emplace_back(worker, q);
    } catch (...) {
        // no exception may cross the C ABI, and the threads that did start must not wait at a barrier for ever
        go.store(2, std::memory_order_release);
        for (auto &t : th) t.join();
        return fail(c, LAM_HIP_ENOMEM, "could not start the per-shard enqueue threads (option host_threads)");
    }
    go.store(1, std::memory_order_release);
    worker(0);
    for (auto &t : th) t.join();
    *enq_out = enq;
    for (int q = 0; q < L; q++)
        if (rcs[q] != 0) return rcs[q];
    return 0;
}
#endif  // LAM_TUNING_VARIANTS

int lam_hip_cg_iterate(lam_hip_ctx *c, int iters, double rel_error, lam_hip_stats *st)
{
    if (!c) return LAM_HIP_EINVAL;
    if (!c->cg_ready) return fail(c, LAM_HIP_ESTATE, "call lam_hip_cg_init first");
    if (iters < 0) return fail(c, LAM_HIP_EINVAL, "iters must be >= 0");
    const double t0 = now_s();
    const uint64_t cpu0 = thread_cpu_ns();
    struct CpuAccount {            // every exit path
        lam_hip_ctx *c; uint64_t t0;
        ~CpuAccount() { c->host_cpu_ns += thread_cpu_ns() - t0; }
    } cpu_account{c, cpu0};
    ShardBase &s0 = c->sh[0];
    c->prog_t = 0.0;               // the estimate of the iteration time carries over, the reference point does not
    double gemv_ms = 0.0;
    int gemv_samples = 0;
    int enq = 0;
    // already converged in an earlier call?
    LAMCHK(set_dev(c, s0));
    HIPCHK(c, hipMemcpyAsync(s0.sc_host, s0.sc, sizeof(CgScalars), hipMemcpyDeviceToHost, s0.stream));
    HIPCHK(c, hipStreamSynchronize(s0.stream));
    const bool stopped = s0.sc_host->stop != 0;
    const int k_first = c->k_done + 1;
    for (int i = 0; i < kLag; i++) s0.timed_slot[i] = false;
    if (stopped) {
        // nothing to enqueue
    }
#ifdef LAM_TUNING_VARIANTS
    else if (c->persist_active) {
        // whole-iteration launches of `chunk` iterations; at most two of them are in the queue (the host waits for the
        // last iteration of the launch before the previous one to report).  After a stop the queued launch returns at once.
        const int chunk = (int)std::max<int64_t>(1, c->opt_persist_chunk);
        int prev_last = 0, prev_prev_last = 0;
        for (int done = 0; done < iters; ) {
            const int cnt = std::min(chunk, iters - done);
            if (prev_prev_last != 0) {
                Progress pr;
                LAMCHK(await_progress(c, s0, prev_prev_last, &pr));
                if (pr.stop_at != 0 || *(volatile int *)c->direct_err != 0) break;
            }
            const double te = now_s();
            LAMCHK(enqueue_persist_chunk(c, k_first + done, cnt, rel_error));
            c->enqueue_ns += (uint64_t)((now_s() - te) * 1e9);
            done += cnt;
            prev_prev_last = prev_last;
            prev_last = k_first + done - 1;
            enq += cnt;
        }
    } else if (!c->rank_mode && c->total_shards > 1 && c->opt_host_threads != 0 && !c->cg_direct && !c->cg_exchange1) {
        LAMCHK(iterate_threaded(c, iters, k_first, rel_error, &enq, &gemv_ms, &gemv_samples));
        c->gather_pending = false;
    }
#endif
    else {
        for (int i = 0; i < iters; i++) {
            const int k = k_first + i;
            const int slot = i % kLag;
            if (i >= kLag) {
                LAMCHK(set_dev(c, s0));
                const int d = lag_check(c, s0, k);
                if (d < 0) return d;
                if (d != 0) break;
                harvest_gemv_time(s0, slot, &gemv_ms, &gemv_samples);
            }
            const double te = now_s();
            LAMCHK(enqueue_iteration(c, k, rel_error, slot));
            c->enqueue_ns += (uint64_t)((now_s() - te) * 1e9);
            enq++;
        }
    }
    if (enq > 0) {
        // the last iterations are still in the queue: wait for the last one's report with the sleeping poll (a
        // hipStreamSynchronize would spin a core for up to kLag iterations), then synchronise for real
        Progress pr;
        LAMCHK(set_dev(c, s0));
        LAMCHK(await_progress(c, s0, k_first + enq - 1, &pr));
    }
    LAMCHK(settle_gather(c));
    LAMCHK(sync_all(c));
    if (*c->direct_err != 0) {
        static const char *what[] = {"", "", "a peer's partial dot product (direct exchange)", "a peer's p slice (direct exchange)",
                                     "the reducer workgroup's broadcast inside the fused update launch (its workgroups were not all resident?)",
                                     "a workgroup's partial inside a launch (reducer workgroup)"};
        const int code = c->direct_err[0];
        return fail(c, LAM_HIP_EHIP, "rank %d: a bounded in-kernel wait expired waiting for %s (code %d; slot %d, expected tag %d:%d, saw %d:%d); "
                                     "the iteration state is no longer valid", c->rank, code >= 2 && code <= 5 ? what[code] : "?", code,
                    c->direct_err[1], c->direct_err[4], c->direct_err[2], c->direct_err[5], c->direct_err[3]);
    }
    if (c->cg_direct && c->rank_mode && c->nranks > 1 && enq > 0) {
        // Direct exchange, one process per GPU: this rank's last iteration is complete once it has the peers' r.r, while
        // those peers may still be storing their p slices into THIS rank's replica (nobody waits for them any more).  A
        // caller that destroys the context or sets a new problem right after this call would free or unmap memory a peer
        // kernel is still writing.  One stream-ordered agreement closes the call: every rank's contribution is
        // enqueued behind its last iteration, so when it completes here, every peer's stores have been issued and drained.
        // (Only on the success path: after an expired wait every rank is already on its way out with an error.)
        int all = 0;
        LAMCHK(lam_hip_all_ok(c, 1, &all));
    }
    // harvest the GEMV timings still in the ring
    for (int j = 0; j < kLag; j++) harvest_gemv_time(s0, j, &gemv_ms, &gemv_samples);
    LAMCHK(set_dev(c, s0));
    HIPCHK(c, hipMemcpyAsync(s0.sc_host, s0.sc, sizeof(CgScalars), hipMemcpyDeviceToHost, s0.stream));
    HIPCHK(c, hipStreamSynchronize(s0.stream));
    const CgScalars sc = *s0.sc_host;
    c->k_done = sc.iters;  // iterations after a stop were no-ops
    const double t1 = now_s();
    if (st) {
        memset(st, 0, sizeof *st);
        const int ran = sc.iters - (k_first - 1);
        st->converged = sc.stop != 0;
        // reference loop counter on exit: the converging iteration, else (last iteration)+1
        st->num_iters = sc.stop ? sc.iters : sc.iters + 1;
        st->rel_err = std::sqrt(sc.rr[sc.iters & 1] / sc.bb);
        st->t_total = t1 - t0;
        st->t_iter = ran > 0 ? (t1 - t0) / ran : 0.0;
        st->t_gemv = gemv_samples > 0 ? gemv_ms * 1e-3 / gemv_samples : 0.0;
#ifdef LAM_TUNING_VARIANTS
        if (c->persist_active && c->persist_ticks != nullptr) {
            // the persistent launch times its GEMV phases itself (constant-rate 100 MHz counter, reducer workgroup):
            // from the previous hand-over to the moment the last partial of p.Ap has been summed
            unsigned long long before[2] = {c->persist_ticks_host[0], c->persist_ticks_host[1]};
            HIPCHK(c, hipMemcpyAsync(c->persist_ticks_host, c->persist_ticks, 2 * sizeof(unsigned long long), hipMemcpyDeviceToHost, s0.stream));
            HIPCHK(c, hipStreamSynchronize(s0.stream));
            const unsigned long long dt = c->persist_ticks_host[0] - before[0], dn = c->persist_ticks_host[1] - before[1];
            st->t_gemv = dn > 0 ? (double)dt * 1e-8 / (double)dn : 0.0;
        }
#endif
        st->t_comm_init = c->t_comm_init;
        st->gemv_bytes = (double)c->esz_a() * (double)s0.nrows * (double)c->n + (double)c->esz_v() * (double)(c->n + s0.nrows);
    }
    (void)enq;
    return 0;
}

int lam_hip_solve(lam_hip_ctx *c, int max_iters, double rel_error, lam_hip_stats *st)
{
    if (!c) return LAM_HIP_EINVAL;
    if (max_iters < 0) return fail(c, LAM_HIP_EINVAL, "max_iters must be >= 0");
    const double t0 = now_s();
    lam_hip_stats local;
    LAMCHK(lam_hip_cg_init(c));
    LAMCHK(lam_hip_cg_iterate(c, max_iters, rel_error, &local));
    if (c->cg_direct && c->opt_verify_direct) {
        // The direct exchange (EXPERIMENTAL until it has run on real peers, see lam_hip.h) hands p slices over
        // through peer-mapped memory and flags; if a rank ever read a slice that was not there yet, the recursion
        // would drift away from b - A x.  So the solve checks itself: the recomputed residual (a collective: every
        // rank gets the same number and takes the same branch) against the recursive one, with room for the
        // attainable accuracy of the dtype.  On a mismatch the system is solved again on the RCCL exchange.
        double tr = 0.0;
        LAMCHK(lam_hip_true_residual(c, &tr));
        // room for what the recursion legitimately drifts by (~ eps x cond): 1e6 eps in fp64, 1e-4 with fp32 vectors;
        // a false alarm costs one more solve, a miss costs a wrong answer
        const double slack = c->dtype == LAM_HIP_F64 ? 2.2e-10 : 1e-4;
        if (!(tr <= 10.0 * local.rel_err + slack)) {
            if (c->rank == 0)
                fprintf(stderr, "lam_hip: direct exchange: recomputed residual %.3e does not match the recursive residual %.3e -- "
                                "solving again on the RCCL exchange\n", tr, local.rel_err);
            c->direct_fallbacks++;
            c->opt_exchange = 0;
            c->cg_ready = false;
            LAMCHK(lam_hip_cg_init(c));
            LAMCHK(lam_hip_cg_iterate(c, max_iters, rel_error, &local));
        }
    }
    local.t_total = now_s() - t0;
    if (st) *st = local;
    return 0;
}

// all-gather a per-shard slice vector (x) into a full host vector
int lam_hip_get_solution(lam_hip_ctx *c, void *x_host)
{
    if (!c || !x_host) return LAM_HIP_EINVAL;
    if (!c->cg_ready) return fail(c, LAM_HIP_ESTATE, "no solution yet");
    const size_t ev = c->esz_v();
    if (c->rank_mode) {
        ShardBase &s = c->sh[0];
        LAMCHK(set_dev(c, s));
        const ncclDataType_t dt = c->dtype == LAM_HIP_F64 ? ncclDouble : ncclFloat;
        HIPCHK(c, hipMemcpyAsync((char *)s.tmp + s.row0 * ev, s.x, s.nrows * ev, hipMemcpyDeviceToDevice, s.stream));
        NCCLCHK(c, ncclGroupStart());
        for (int q = 0; q < c->nranks; q++) {
            uint64_t r0, nr;
            partition(c->n, c->nranks, q, &r0, &nr);
            char *ptr = (char *)s.tmp + r0 * ev;
            NCCLCHK(c, ncclBroadcast(ptr, ptr, nr, dt, q, c->comm, s.stream));
            c->n_collectives++;
        }
        NCCLCHK(c, ncclGroupEnd());
        HIPCHK(c, hipMemcpyAsync(x_host, s.tmp, c->n * ev, hipMemcpyDeviceToHost, s.stream));
        HIPCHK(c, hipStreamSynchronize(s.stream));
        return 0;
    }
    for (auto &s : c->sh) {
        LAMCHK(set_dev(c, s));
        HIPCHK(c, hipMemcpyAsync((char *)x_host + s.row0 * ev, s.x, s.nrows * ev, hipMemcpyDeviceToHost, s.stream));
        HIPCHK(c, hipStreamSynchronize(s.stream));
    }
    return 0;
}

// y = A v for a full-length DEVICE-replicated vector held in every shard's tmp; result slices in Ap
static int gemv_tmp(lam_hip_ctx *c)
{
    return dispatch(c, [&](auto impl) -> int {
        using I = decltype(impl);
        using TV = typename ImplTraits<I>::TV;
        for (auto &s : c->sh) {
            LAMCHK(set_dev(c, s));
            LAMCHK(I::launch_gemv(c, s, (const TV *)s.tmp, (TV *)s.Ap, nullptr, nullptr));
        }
        return 0;
    });
}

int lam_hip_gemv(lam_hip_ctx *c, const void *x_host, void *y_host)
{
    if (!c || !x_host || !y_host) return LAM_HIP_EINVAL;
    if (!c->have_matrix) return fail(c, LAM_HIP_ESTATE, "matrix not set");
    const size_t ev = c->esz_v();
    for (auto &s : c->sh) {
        LAMCHK(set_dev(c, s));
        HIPCHK(c, hipMemcpyAsync(s.tmp, x_host, c->n * ev, hipMemcpyHostToDevice, s.stream));
    }
    if (c->symv_active()) {
        LAMCHK(dispatch(c, [&](auto impl) -> int {
            using I = decltype(impl);
            using TV = typename ImplTraits<I>::TV;
            ShardBase &s = c->sh[0];
            LAMCHK(set_dev(c, s));
            return I::launch_symv(c, s, (const TV *)s.tmp, (TV *)s.Ap, nullptr, nullptr);
        }));
    } else {
        LAMCHK(gemv_tmp(c));
    }
    c->cg_ready = false;  // Ap was overwritten
    if (c->rank_mode) {
        ShardBase &s = c->sh[0];
        const ncclDataType_t dt = c->dtype == LAM_HIP_F64 ? ncclDouble : ncclFloat;
        HIPCHK(c, hipMemcpyAsync((char *)s.tmp + s.row0 * ev, s.Ap, s.nrows * ev, hipMemcpyDeviceToDevice, s.stream));
        NCCLCHK(c, ncclGroupStart());
        for (int q = 0; q < c->nranks; q++) {
            uint64_t r0, nr;
            partition(c->n, c->nranks, q, &r0, &nr);
            char *ptr = (char *)s.tmp + r0 * ev;
            NCCLCHK(c, ncclBroadcast(ptr, ptr, nr, dt, q, c->comm, s.stream));
            c->n_collectives++;
        }
        NCCLCHK(c, ncclGroupEnd());
        HIPCHK(c, hipMemcpyAsync(y_host, s.tmp, c->n * ev, hipMemcpyDeviceToHost, s.stream));
        HIPCHK(c, hipStreamSynchronize(s.stream));
        return 0;
    }
    for (auto &s : c->sh) {
        LAMCHK(set_dev(c, s));
        HIPCHK(c, hipMemcpyAsync((char *)y_host + s.row0 * ev, s.Ap, s.nrows * ev, hipMemcpyDeviceToHost, s.stream));
        HIPCHK(c, hipStreamSynchronize(s.stream));
    }
    return 0;
}

int lam_hip_gemv_only(lam_hip_ctx *c, int reps, double *sec)
{
    if (!c || !sec || reps < 1) return LAM_HIP_EINVAL;
    if (!c->have_matrix) return fail(c, LAM_HIP_ESTATE, "matrix not set");
    c->cg_ready = false;
    double worst = 0.0;
    LAMCHK(dispatch(c, [&](auto impl) -> int {
        using I = decltype(impl);
        using TV = typename ImplTraits<I>::TV;
        for (auto &s : c->sh) {
            LAMCHK(set_dev(c, s));
            const uint64_t nrows_saved = s.nrows;
            if (c->opt_probe_rows > 0) s.nrows = std::min<uint64_t>(s.nrows, (uint64_t)c->opt_probe_rows);
            // with panel_lo/panel_hi set the probe times the split form the rank mode uses (own-slice
            // panel, then the remaining columns accumulated on top)
            const bool split = c->opt_panel_hi > c->opt_panel_lo;
            const uint64_t lo = (uint64_t)c->opt_panel_lo, hi = (uint64_t)c->opt_panel_hi;
            auto one = [&]() -> int {
                if (c->symv_active()) return I::launch_symv(c, s, (const TV *)s.p, (TV *)s.Ap, s.part_gemv, nullptr);
                if (!split) return I::launch_gemv(c, s, (const TV *)s.p, (TV *)s.Ap, s.part_gemv, nullptr);
                int r1 = I::launch_gemv(c, s, (const TV *)s.p, (TV *)s.Ap, nullptr, nullptr, 1, lo, hi);
                return r1 != 0 ? r1 : I::launch_gemv(c, s, (const TV *)s.p, (TV *)s.Ap, s.part_gemv, nullptr, 2, lo, hi);
            };
            int rc = one();  // warm-up
            if (rc == 0 && hipEventRecord(s.ev_g0[0], s.stream) != hipSuccess) rc = LAM_HIP_EHIP;
            for (int i = 0; i < reps && rc == 0; i++) rc = one();
            if (rc == 0 && hipEventRecord(s.ev_g1[0], s.stream) != hipSuccess) rc = LAM_HIP_EHIP;
            s.nrows = nrows_saved;
            if (rc != 0) return rc == LAM_HIP_EHIP ? fail(c, rc, "hipEventRecord failed in gemv_only") : rc;
        }
        for (auto &s : c->sh) {
            LAMCHK(set_dev(c, s));
            HIPCHK(c, hipEventSynchronize(s.ev_g1[0]));
            float ms = 0.f;
            HIPCHK(c, hipEventElapsedTime(&ms, s.ev_g0[0], s.ev_g1[0]));
            worst = std::max(worst, (double)ms * 1e-3 / reps);
        }
        return 0;
    }));
    *sec = worst;
    return 0;
}

int lam_hip_true_residual(lam_hip_ctx *c, double *rel_res)
{
    if (!c || !rel_res) return LAM_HIP_EINVAL;
    if (!c->cg_ready) return fail(c, LAM_HIP_ESTATE, "no solution yet");
    const size_t ev = c->esz_v();
    // replicate x into every shard's tmp
    if (c->rank_mode) {
        ShardBase &s = c->sh[0];
        LAMCHK(set_dev(c, s));
        const ncclDataType_t dt = c->dtype == LAM_HIP_F64 ? ncclDouble : ncclFloat;
        HIPCHK(c, hipMemcpyAsync((char *)s.tmp + s.row0 * ev, s.x, s.nrows * ev, hipMemcpyDeviceToDevice, s.stream));
        NCCLCHK(c, ncclGroupStart());
        for (int q = 0; q < c->nranks; q++) {
            uint64_t r0, nr;
            partition(c->n, c->nranks, q, &r0, &nr);
            char *ptr = (char *)s.tmp + r0 * ev;
            NCCLCHK(c, ncclBroadcast(ptr, ptr, nr, dt, q, c->comm, s.stream));
            c->n_collectives++;
        }
        NCCLCHK(c, ncclGroupEnd());
    } else {
        LAMCHK(sync_all(c));
        for (auto &dst : c->sh)
            for (auto &src : c->sh) {
                LAMCHK(set_dev(c, dst));
                HIPCHK(c, hipMemcpyAsync((char *)dst.tmp + src.row0 * ev, src.x, src.nrows * ev, hipMemcpyDefault, dst.stream));
            }
        LAMCHK(sync_all(c));
    }
    // Ap is reused as scratch for A x: CG state stays valid because every iteration rewrites Ap first
    LAMCHK(gemv_tmp(c));
    double num = 0.0, den = 0.0;
    LAMCHK(dispatch(c, [&](auto impl) -> int {
        using TV = typename ImplTraits<decltype(impl)>::TV;
        for (auto &s : c->sh) {
            LAMCHK(set_dev(c, s));
            std::vector<double> h(2 * kVecBlocksMax);
            // part_aux, not part_vec: the iteration's partial arrays hold the reducer's sentinels between iterations
            hipLaunchKernelGGL((resid_partial_kernel<TV>), dim3(s.vec_blocks), dim3(kBlock), 0, s.stream, (const TV *)s.b, (const TV *)s.Ap, s.nrows, s.part_aux);
            HIPCHK(c, hipGetLastError());
            HIPCHK(c, hipMemcpyAsync(h.data(), s.part_aux, sizeof(double) * s.vec_blocks, hipMemcpyDeviceToHost, s.stream));
            HIPCHK(c, hipStreamSynchronize(s.stream));
            for (int i = 0; i < s.vec_blocks; i++) num += h[i];
            hipLaunchKernelGGL((dot_partial_kernel<TV>), dim3(s.vec_blocks), dim3(kBlock), 0, s.stream, (const TV *)s.b, (const TV *)s.b, s.nrows, s.part_aux);
            HIPCHK(c, hipGetLastError());
            HIPCHK(c, hipMemcpyAsync(h.data(), s.part_aux, sizeof(double) * s.vec_blocks, hipMemcpyDeviceToHost, s.stream));
            HIPCHK(c, hipStreamSynchronize(s.stream));
            for (int i = 0; i < s.vec_blocks; i++) den += h[i];
        }
        return 0;
    }));
    if (c->rank_mode) {
        ShardBase &s = c->sh[0];
        double hv[2] = {num, den};
        HIPCHK(c, hipMemcpyAsync(s.gather_a, hv, sizeof hv, hipMemcpyHostToDevice, s.stream));
        NCCLCHK(c, ncclAllReduce(s.gather_a, s.gather_a, 2, ncclDouble, ncclSum, c->comm, s.stream));
        c->n_collectives++;
        HIPCHK(c, hipMemcpyAsync(hv, s.gather_a, sizeof hv, hipMemcpyDeviceToHost, s.stream));
        HIPCHK(c, hipStreamSynchronize(s.stream));
        num = hv[0]; den = hv[1];
    }
    *rel_res = std::sqrt(num / den);
    return 0;
}

int lam_hip_check_symmetry(lam_hip_ctx *c, double *max_abs_asymmetry)
{
    if (!c || !max_abs_asymmetry) return LAM_HIP_EINVAL;
    if (!c->have_matrix) return fail(c, LAM_HIP_ESTATE, "matrix not set");
    if (c->rank_mode || c->total_shards != 1) return fail(c, LAM_HIP_EINVAL, "symmetry check needs the whole matrix on one shard");
    return dispatch(c, [&](auto impl) -> int {
        using TA = typename ImplTraits<decltype(impl)>::TA;
        ShardBase &s = c->sh[0];
        LAMCHK(set_dev(c, s));
        const int grid = 2048;
        DevBuf outb;
        HIPCHK(c, hipMalloc(&outb.p, sizeof(double) * grid));
        double *out = outb.as<double>();
        std::vector<double> h(grid);
        if constexpr (sizeof(TA) == 2) {
            return fail(c, LAM_HIP_EINVAL, "symmetry check is implemented for fp64/fp32 storage");
        } else {
            hipLaunchKernelGGL((asymmetry_kernel<TA>), dim3(grid), dim3(kBlock), 0, s.stream, (const TA *)s.A, c->n, out);
            hipError_t e = hipGetLastError();
            if (e == hipSuccess) e = hipMemcpyAsync(h.data(), out, sizeof(double) * grid, hipMemcpyDeviceToHost, s.stream);
            if (e == hipSuccess) e = hipStreamSynchronize(s.stream);
            if (e != hipSuccess) return fail(c, LAM_HIP_EHIP, "symmetry check: %s", hipGetErrorString(e));
            double m = 0.0;
            for (double v : h) m = std::max(m, v);
            *max_abs_asymmetry = m;
            return 0;
        }
    });
}

int lam_hip_dot(lam_hip_ctx *c, const void *x_host, const void *y_host, uint64_t n, double *result)
{
    if (!c || !x_host || !y_host || !result) return LAM_HIP_EINVAL;
    ShardBase &s = c->sh[0];
    LAMCHK(set_dev(c, s));
    const size_t ev = c->esz_v();
    DevBuf bx, by, bpart;
    HIPCHK(c, hipMalloc(&bx.p, n * ev + 16));
    HIPCHK(c, hipMalloc(&by.p, n * ev + 16));
    HIPCHK(c, hipMalloc(&bpart.p, sizeof(double) * kVecBlocksMax));
    void *dx = bx.p, *dy = by.p;
    double *part = bpart.as<double>();
    HIPCHK(c, hipMemcpyAsync(dx, x_host, n * ev, hipMemcpyHostToDevice, s.stream));
    HIPCHK(c, hipMemcpyAsync(dy, y_host, n * ev, hipMemcpyHostToDevice, s.stream));
    const int grid = vec_grid(n);
    int rc = dispatch(c, [&](auto impl) -> int {
        using TV = typename ImplTraits<decltype(impl)>::TV;
        hipLaunchKernelGGL((dot_partial_kernel<TV>), dim3(grid), dim3(kBlock), 0, s.stream, (const TV *)dx, (const TV *)dy, n, part);
        HIPCHK(c, hipGetLastError());
        return 0;
    });
    if (rc == 0) {
        std::vector<double> h(grid);
        hipError_t e = hipMemcpyAsync(h.data(), part, sizeof(double) * grid, hipMemcpyDeviceToHost, s.stream);
        if (e == hipSuccess) e = hipStreamSynchronize(s.stream);
        if (e != hipSuccess) rc = fail(c, LAM_HIP_EHIP, "dot readback: %s", hipGetErrorString(e));
        double t = 0.0;
        for (int i = 0; i < grid; i++) t += h[i];   // fixed order: reproducible
        *result = t;
    }
    return rc;
}

int lam_hip_axpby(lam_hip_ctx *c, double alpha, const void *x_host, double beta, void *y_host, uint64_t n)
{
    if (!c || !x_host || !y_host) return LAM_HIP_EINVAL;
    ShardBase &s = c->sh[0];
    LAMCHK(set_dev(c, s));
    const size_t ev = c->esz_v();
    DevBuf bx, by;
    HIPCHK(c, hipMalloc(&bx.p, n * ev + 16));
    HIPCHK(c, hipMalloc(&by.p, n * ev + 16));
    void *dx = bx.p, *dy = by.p;
    HIPCHK(c, hipMemcpyAsync(dx, x_host, n * ev, hipMemcpyHostToDevice, s.stream));
    HIPCHK(c, hipMemcpyAsync(dy, y_host, n * ev, hipMemcpyHostToDevice, s.stream));
    int rc = dispatch(c, [&](auto impl) -> int {
        using TV = typename ImplTraits<decltype(impl)>::TV;
        hipLaunchKernelGGL((axpby_kernel<TV>), dim3(vec_grid(n)), dim3(kBlock), 0, s.stream, (TV)alpha, (const TV *)dx, (TV)beta, (TV *)dy, n);
        HIPCHK(c, hipGetLastError());
        return 0;
    });
    if (rc == 0) {
        hipError_t e = hipMemcpyAsync(y_host, dy, n * ev, hipMemcpyDeviceToHost, s.stream);
        if (e == hipSuccess) e = hipStreamSynchronize(s.stream);
        if (e != hipSuccess) rc = fail(c, LAM_HIP_EHIP, "axpby readback: %s", hipGetErrorString(e));
    }
    return rc;
}

int lam_hip_all_ok(lam_hip_ctx *c, int local_ok, int *global_ok)
{
    if (!c || !global_ok) return LAM_HIP_EINVAL;
    *global_ok = local_ok ? 1 : 0;
    if (!c->rank_mode) return 0;
    ShardBase &s = c->sh[0];
    LAMCHK(set_dev(c, s));
    // No hipMalloc/hipFree here: hipFree waits for the whole device, and when the ranks are threads of one
    // process (the test double) that includes peers' kernels that are waiting for THIS rank's next call.
    if (c->agree_buf == nullptr) HIPCHK(c, hipMalloc((void **)&c->agree_buf, 4096));
    double v = local_ok ? 0.0 : 1.0;       // number of ranks that failed
    HIPCHK(c, hipMemcpyAsync(c->agree_buf, &v, sizeof v, hipMemcpyHostToDevice, s.stream));
    NCCLCHK(c, ncclAllReduce(c->agree_buf, c->agree_buf, 1, ncclDouble, ncclSum, c->comm, s.stream));
    c->n_collectives++;
    HIPCHK(c, hipMemcpyAsync(&v, c->agree_buf, sizeof v, hipMemcpyDeviceToHost, s.stream));
    HIPCHK(c, hipStreamSynchronize(s.stream));
    *global_ok = v == 0.0 ? 1 : 0;
    return 0;
}

int lam_hip_rccl_version(int *version)
{
    if (!version) return LAM_HIP_EINVAL;
    ncclResult_t r = ncclGetVersion(version);
    if (r != ncclSuccess) return fail(nullptr, LAM_HIP_ERCCL, "ncclGetVersion: %s", ncclGetErrorString(r));
    return 0;
}

int lam_hip_gemv_kernel_name(const lam_hip_ctx *c, char *buf, size_t len)
{
    if (!c || !buf || len == 0) return LAM_HIP_EINVAL;
    std::string name;
    switch (c->dtype) {
    case LAM_HIP_F64: name = Impl<double, double>::kernel_name(c); break;
    case LAM_HIP_F32: name = Impl<float, float>::kernel_name(c); break;
    default: name = Impl<__hip_bfloat16, float>::kernel_name(c); break;
    }
    snprintf(buf, len, "%s", name.c_str());
    return 0;
}

#ifdef LAM_TUNING_VARIANTS
static constexpr bool kTuningBuild = true;
#else
static constexpr bool kTuningBuild = false;
#endif
static const char *const kTuningOnly = "%s is an experiment that did not win: it exists in the tuning build only (`make tuning`, load "
                                       "liblam_hip_tuning.so through LAM_HIP_LIB), not in the product library";

int lam_hip_set_option(lam_hip_ctx *c, const char *name, int64_t value)
{
    if (!c || !name) return LAM_HIP_EINVAL;
    if (!strcmp(name, "gemv_variant")) {
        if (value >= 0 && !Impl<double, double>::variant_available((int)value))
            return fail(c, LAM_HIP_EINVAL, "gemv_variant %lld is a tuning shape: not in the product library (build `make tuning` and "
                                           "load liblam_hip_tuning.so, see tools/gemv_probe.py)", (long long)value);
        c->opt_gemv_variant = value;
    }
    else if (!strcmp(name, "nt_loads")) c->opt_nt = value;
    else if (!strcmp(name, "force_generic")) c->opt_generic = value;
    else if (!strcmp(name, "probe_rows")) c->opt_probe_rows = value;
    else if (!strcmp(name, "overlap")) c->opt_overlap = value;
    else if (!strcmp(name, "exchange")) { c->opt_exchange = value; c->cg_ready = false; }
    else if (!strcmp(name, "exchange_join")) c->opt_join = value;
    else if (!strcmp(name, "finalize")) {
        if (value == 0 && !kTuningBuild) return fail(c, LAM_HIP_EINVAL, kTuningOnly, "finalize = 0 (separate reduction launches, the round-1 chain)");
        c->opt_finalize = value; c->cg_ready = false;
    }
    else if (!strcmp(name, "upload_staging")) c->opt_upload_staging = value;
    else if (!strcmp(name, "reuse_matrix")) c->opt_reuse_matrix = value;
    else if (!strcmp(name, "fuse_update")) { c->opt_fuse = value; c->cg_ready = false; }
    else if (!strcmp(name, "persistent") || !strcmp(name, "persist_chunk")) {
        if (!kTuningBuild && !(value == 0 && !strcmp(name, "persistent"))) return fail(c, LAM_HIP_EINVAL, kTuningOnly, "the whole-iteration persistent launch");
        if (!strcmp(name, "persistent")) { c->opt_persistent = value; c->cg_ready = false; }
        else c->opt_persist_chunk = value;
    }
    else if (!strcmp(name, "symmetric")) { c->opt_symmetric = value; c->cg_ready = false; }   // other kernels, other partial arrays
    else if (!strcmp(name, "gemv_timing")) c->opt_gemv_timing = value < 0 ? 0 : value;
    else if (!strcmp(name, "verify_direct")) c->opt_verify_direct = value;
    else if (!strcmp(name, "host_threads") || !strcmp(name, "exchange_hub")) {
        if (value != 0 && !kTuningBuild) return fail(c, LAM_HIP_EINVAL, kTuningOnly, name);
        if (!strcmp(name, "host_threads")) c->opt_host_threads = value;
        else c->opt_hub = value;
    }
    else if (!strcmp(name, "assume_cus")) { c->opt_assume_cus = value; c->cg_ready = false; }
    else if (!strcmp(name, "panel_lo")) c->opt_panel_lo = value;
    else if (!strcmp(name, "panel_hi")) c->opt_panel_hi = value;
    else return fail(c, LAM_HIP_EINVAL, "unknown option '%s'", name);
    // the GEMV grid (= number of p.Ap partials the next kernel sums) depends on the kernel shape
    for (auto &s : c->sh)
        if (c->have_problem) s.gemv_blocks = dispatch(c, [&](auto impl) -> int { return decltype(impl)::gemv_grid(c, s.nrows); });
    return 0;
}

int lam_hip_get_option(const lam_hip_ctx *c, const char *name, int64_t *value)
{
    if (!c || !name || !value) return LAM_HIP_EINVAL;
    if (!strcmp(name, "gemv_variant")) *value = c->opt_gemv_variant;
    else if (!strcmp(name, "nt_loads")) *value = c->opt_nt;
    else if (!strcmp(name, "force_generic")) *value = c->opt_generic;
    else if (!strcmp(name, "probe_rows")) *value = c->opt_probe_rows;
    else if (!strcmp(name, "overlap")) *value = c->opt_overlap;
    else if (!strcmp(name, "exchange")) *value = c->opt_exchange;
    else if (!strcmp(name, "exchange_join")) *value = c->opt_join;
    else if (!strcmp(name, "finalize")) *value = c->opt_finalize;
    else if (!strcmp(name, "upload_staging")) *value = c->opt_upload_staging;
    else if (!strcmp(name, "reuse_matrix")) *value = c->opt_reuse_matrix;
    else if (!strcmp(name, "fuse_update")) *value = c->opt_fuse;
    else if (!strcmp(name, "collectives_enqueued")) *value = (int64_t)c->n_collectives;
    else if (!strcmp(name, "tuning_variants")) *value = Impl<double, double>::variant_available(1) ? 1 : 0;
    else if (!strcmp(name, "fuse_effective")) *value = c->fuse_active ? 1 : 0;
    else if (!strcmp(name, "persistent")) *value = c->opt_persistent;
    else if (!strcmp(name, "persistent_effective")) *value = c->persist_active ? 1 : 0;
    else if (!strcmp(name, "persistent_workers")) *value = c->persist_active ? c->persist_W : 0;
    else if (!strcmp(name, "persist_chunk")) *value = c->opt_persist_chunk;
    else if (!strcmp(name, "gemv_timing")) *value = c->opt_gemv_timing;
    else if (!strcmp(name, "verify_direct")) *value = c->opt_verify_direct;
    else if (!strcmp(name, "direct_fallbacks")) *value = c->direct_fallbacks;
    else if (!strcmp(name, "host_threads")) *value = c->opt_host_threads;
    else if (!strcmp(name, "exchange_hub")) *value = c->opt_hub;
    else if (!strcmp(name, "assume_cus")) *value = c->opt_assume_cus;
    else if (!strcmp(name, "host_enqueue_ns")) *value = (int64_t)c->enqueue_ns;
    else if (!strcmp(name, "host_cpu_ns")) *value = (int64_t)c->host_cpu_ns;
    else if (!strcmp(name, "hip_calls_launch")) *value = (int64_t)c->n_launch.load();
    else if (!strcmp(name, "hip_calls_record")) *value = (int64_t)c->n_record.load();
    else if (!strcmp(name, "hip_calls_wait")) *value = (int64_t)c->n_wait.load();
    else if (!strcmp(name, "hip_calls_setdevice")) *value = (int64_t)c->n_setdev.load();
    else if (!strcmp(name, "symmetric")) *value = c->opt_symmetric;
    else if (!strcmp(name, "symmetric_effective")) *value = c->symv_active() ? 1 : 0;
    else if (!strcmp(name, "exchange_effective")) *value = c->cg_direct ? 2 : ((c->exchange1_ok() && !c->exchange2_wanted()) ? 1 : 0);
    else if (!strcmp(name, "panel_lo")) *value = c->opt_panel_lo;
    else if (!strcmp(name, "panel_hi")) *value = c->opt_panel_hi;
    else return LAM_HIP_EINVAL;
    return 0;
}

}  // extern "C"
