// lam_ctx.h -- shard and context state, error plumbing and small host helpers of the MI355X-native dense CG hot path.
// Part of the one translation unit csrc/lam_hip.hip (included from there, in order; not a stand-alone header).
#pragma once

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
// Default exchange of new contexts, ONE LINE EACH to flip once a run on separate GPUs has priced the collectives (bench.py records
// every exchange of both topologies under exchange_modes).  Gather-Ap in both since round 5: one collective / one event join per
// iteration instead of three, two launches instead of three, everything on one stream -- 620 + L' against 644 + 2 L us per iteration
// at P = 8 by the one-GPU model, 3 x the iteration rate on the stream-ordered RCCL double.
constexpr int64_t kRankModeDefaultExchange = 1;      // lam_hip_create_rank (round 4: 0)
constexpr int64_t kOneProcessDefaultExchange = 1;    // lam_hip_create with several shards
constexpr int kVecBlocksMax = 256;

struct ShardBase {
    int dev = 0;
    int index = 0;               // global shard index
    uint64_t row0 = 0, nrows = 0;
    hipStream_t stream = nullptr;
    void *A = nullptr;           // nrows x n
    size_t A_capacity = 0;       // bytes behind A: the allocation is kept across lam_hip_set_problem calls (grow-only)
    void *xfer_stage = nullptr;  // dense staging buffer of padded / bf16 row transfers (lam_hip_upload_rows / download_rows): kept
    size_t xfer_stage_bytes = 0; // for the life of the context, grow-only -- a hipMalloc + hipFree of up to 1 GiB per chunk of a file
                                 // load was a device-wide synchronisation each and left VRAM for the driver to wipe under the next kernels
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
    uint32_t *symv_index = nullptr;       // what the second pass needs to find the partials (lam_kernels.h, SymvIndex)
    SymvIndex symv_ix = {};
    int symv_ntasks = 0;
    void *symv_gather = nullptr;          // several shards: P records [full-length contribution to A p | double], double-buffered
                                          // like ap_gather in one process (symv_gather_bytes = one buffer)
    size_t symv_gather_bytes = 0;
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
    hipEvent_t ev_x[kLag][6] = {};                              // exchange timing ring (shard 0): up to three begin / end pairs per timed
    int nx[kLag] = {};                                          // iteration -- one per exchange step (lam_exchange.h, xt_begin / xt_end)
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
    uint64_t lda = 0;              // row pitch of A in ELEMENTS: n rounded up so that a row is a whole number of 4-KiB pages (of
                                   // 16-byte vectors when a row is shorter than a page); the padding is zero.  Every row then
                                   // starts page-aligned -- round decimal sizes (N = 10000 ... 70000, the reference's grid) stream
                                   // 0.5-1.5 % faster than with rows packed back to back (profiles/r04_size_sweep_align.txt) --
                                   // and the 16-byte-vector GEMV kernels serve ANY N, odd ones included (the scalar-peel kernel ran
                                   // those at 5.8-5.9 TB/s)
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
    double *agree_buf = nullptr;                // kAgreeBytes of device scratch of the small set-up collectives (kept: no hipFree in them)
    bool cg_exchange1 = false;     // the exchange the current CG state was initialised for
    double t_gemv_min = 0.0, t_gemv_max = 0.0;      // fastest / slowest local shard's average GEMV time of the last cg_iterate call (seconds)
    int xt_slot = -1;              // timing-ring slot of the iteration being enqueued if that iteration is timed (option gemv_timing),
                                   // else -1: its exchange steps are bracketed with HIP events too (lam_hip_stats.t_exchange)
    int ranks_on_device = 1;       // rank mode: ranks of the communicator that share THIS rank's GPU (emulations; counted once at
                                   // creation over the communicator): their fused vector-step launches must all be resident together
    bool symmetric_from_env = false;   // option "symmetric" came from LAM_HIP_SYMMETRIC (a driver that cannot call set_option): the
                                       // library then verifies A = A^T itself where it can and says when the option is not effective
    uint64_t matrix_gen = 0, sym_checked_gen = ~0ull;   // the matrix contents changed / were last checked for symmetry
    bool sym_refused = false;      // the environment asked for the symmetric product and the matrix is not symmetric: general GEMV
    bool told_sym_ineffective = false;

    // the symmetric product exists for every storage type and any n: one shard, or several row shards on the gather-Ap exchange
    // (inside CG only: every shard contributes a full-length vector per iteration, lam_exchange.h)
    // Option value 1: where it pays -- for a matrix below ~200 MB the two passes' fixed costs outweigh the halved stream (fp64: N = 3072
    // 1.0 x the general GEMV, 4096 1.13-1.18 x, 5120 1.3 x; fp32: 4096 0.8-1.0 x, 8192 1.1 x; bf16: 8192 0.93 x, 16384 1.24 x;
    // profiles/r04_symmetric_probe.txt): from 192 MiB of matrix on (fp64 N >= 5017, fp32 7095, bf16 10033); value 2: always.
    bool symv_wanted() const
    {
        return n > 0 && !sym_refused && (opt_symmetric >= 2 || (opt_symmetric == 1 && n * n * (uint64_t)esz_a() >= (192ull << 20)));
    }
    bool symv_active() const { return symv_wanted() && !rank_mode && total_shards == 1; }
    bool symv_multi_active() const { return symv_wanted() && exchange1_ok(); }
    // a shard's record in the symmetric product's gather: its full-length contribution, then (8-byte aligned) its part of p.Ap
    uint64_t symv_stride_bytes() const { return (n * esz_v() + 7) / 8 * 8 + 8; }

    bool exchange2_wanted() const { return (rank_mode || total_shards > 1) && opt_exchange == 2 && opt_finalize != 0; }
    // gather-Ap serves the reference's partition as it is (ConjugateGradient_CPU_MPI_OMP.hpp:176-196: n / P rows each, the
    // remainder on the LAST shard, gathered there with MPI_Allgatherv + sendcounts / displs, :505): every record has room for the
    // LONGEST slice (n / P + n % P rows; the shorter ones leave the rest unused -- one equal-count ncclAllGather in rank mode),
    // followed by the shard's part of p.Ap as a double on an 8-byte boundary.  Any n >= P.
    bool exchange1_ok() const { return (rank_mode || total_shards > 1) && opt_exchange == 1 && n >= (uint64_t)total_shards; }
    uint64_t ex1_base() const { return n / (uint64_t)total_shards; }
    uint64_t ex1_maxrows() const { return n / (uint64_t)total_shards + n % (uint64_t)total_shards; }
    uint64_t ex1_stride_bytes() const { return (ex1_maxrows() * esz_v() + 7) / 8 * 8 + 8; }

    size_t esz_a() const { return dtype == LAM_HIP_F64 ? 8 : (dtype == LAM_HIP_F32 ? 4 : 2); }
    static uint64_t pitch_for(uint64_t n, size_t ea) { return lam::row_pitch(n, ea); }      // lam_host_plan.h
    // columns the 16-byte-vector kernels cover: n rounded up to a whole vector (the extra columns are zeros of the padding,
    // met by zeros behind the end of p)
    uint64_t ncols_vec() const { const uint64_t v = 16 / esz_a(); return (n + v - 1) / v * v; }
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

// the matrix contents changed (upload, generators): the CG state and what was known about the old contents are void
void matrix_changed(lam_hip_ctx *c)
{
    c->have_matrix = true;
    c->cg_ready = false;
    c->matrix_gen++;
    c->sym_refused = false;
}

// ConjugateGradient_CPU_MPI_OMP.hpp:176-184: n/P rows each, the remainder on the LAST rank (lam_host_plan.h)
void partition(uint64_t n, int P, int q, uint64_t *row0, uint64_t *nrows) { lam::partition_rows(n, P, q, row0, nrows); }

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

}  // namespace
