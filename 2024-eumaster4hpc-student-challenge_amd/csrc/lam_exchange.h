// lam_exchange.h -- context creation and the exchanges between shards: event-ordered peer stores, RCCL, the direct (in-kernel flag) exchange, gather-Ap.
// Part of the one translation unit csrc/lam_hip.hip (included from there, in order; not a stand-alone header).
#pragma once

namespace {

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

// Environment LAM_HIP_SYMMETRIC = 1 | 2: option "symmetric" of new contexts (the reference's drivers compile unchanged against
// these headers and have no other way to ask for it).  The caller vouches for A = A^T, as CG itself does.
int64_t symmetric_from_env(int64_t dflt)
{
    const char *v = getenv("LAM_HIP_SYMMETRIC");
    if (v == nullptr || *v == '\0') return dflt;
    const int k = atoi(v);
    return (k >= 0 && k <= 2) ? k : dflt;
}

// max |A - A^T| and max |A| of the matrix a single-PROCESS context holds (one tiled pass over the upper triangle and its mirror
// image; with several shards shard 0's device reads the other shards' rows through peer access)
int measure_asymmetry(lam_hip_ctx *c, double *max_asym, double *max_abs)
{
    return dispatch(c, [&](auto impl) -> int {
        using TA = typename ImplTraits<decltype(impl)>::TA;
        for (auto &t : c->sh) {                      // the generators / uploads of every shard are done
            LAMCHK(set_dev(c, t));
            HIPCHK(c, hipStreamSynchronize(t.stream));
        }
        ShardBase &s = c->sh[0];
        LAMCHK(set_dev(c, s));
        PtrList shards;
        shards.n = (int)c->sh.size();
        for (int q = 0; q < shards.n; q++) shards.p[q] = c->sh[q].A;
        const int grid = 2048;
        DevBuf outb;
        HIPCHK(c, hipMalloc(&outb.p, sizeof(double) * 2 * grid));
        std::vector<double> h(2 * grid);
        hipLaunchKernelGGL((asymmetry_kernel<TA>), dim3(grid), dim3(kBlock), 0, s.stream, shards, c->n / (uint64_t)c->total_shards, c->lda, c->n, outb.as<double>());
        HIPCHK(c, hipGetLastError());
        HIPCHK(c, hipMemcpyAsync(h.data(), outb.p, sizeof(double) * 2 * grid, hipMemcpyDeviceToHost, s.stream));
        HIPCHK(c, hipStreamSynchronize(s.stream));
        double m = 0.0, a = 0.0;
        for (int i = 0; i < grid; i++) { m = std::max(m, h[i]); a = std::max(a, h[grid + i]); }
        *max_asym = m;
        *max_abs = a;
        return 0;
    });
}

// Option "symmetric" asked for through the ENVIRONMENT (a driver that cannot call lam_hip_set_option or lam_hip_check_symmetry,
// e.g. the reference's own driver sources compiled against these headers): the library vouches for the precondition itself.
//   * not effective (rank mode / several shards on an exchange other than gather-Ap): said once on stderr, the general GEMV runs;
//   * one process (one shard or several): A is compared with its transpose once per matrix (one pass over A; the rows of other
//     shards through peer access).  Equal bit for bit: nothing to say.
//     Unequal at rounding level (<= 64 ulp of the largest element: a file written by a generator that rounds A_ij and A_ji
//     separately, like the reference's MKL-based one): a warning -- the upper triangle then DEFINES the system that is solved.
//     More: refused, the general GEMV runs (and says so);
//   * rank mode: the transpose lives in other processes -- not checked, the caller vouches as with the option.
int env_symmetric_check(lam_hip_ctx *c)
{
    if (!c->symmetric_from_env || c->opt_symmetric == 0) return 0;
    if (c->symv_wanted() && !c->symv_active() && !c->symv_multi_active()) {
        if (!c->told_sym_ineffective && c->rank == 0)
            fprintf(stderr, "lam_hip: LAM_HIP_SYMMETRIC=%lld is not effective here: with several shards / ranks the symmetric product runs on "
                            "the gather-Ap exchange only (exchange is %lld; set LAM_HIP_EXCHANGE=1) -- using the general GEMV\n",
                    (long long)c->opt_symmetric, (long long)c->opt_exchange);
        c->told_sym_ineffective = true;       // once per context
        return 0;
    }
    if (!c->symv_active() && !c->symv_multi_active()) return 0;           // not asked for at this size
    if (c->rank_mode || c->sym_checked_gen == c->matrix_gen) return 0;     // rank mode: the transpose lives in other processes
    double asym = 0.0, amax = 0.0;
    LAMCHK(measure_asymmetry(c, &asym, &amax));
    c->sym_checked_gen = c->matrix_gen;
    const double eps = c->dtype == LAM_HIP_F64 ? 2.220446049250313e-16 : (c->dtype == LAM_HIP_F32 ? 1.1920929e-07 : 7.8125e-03);
    if (asym == 0.0) return 0;
    if (asym <= 64.0 * eps * amax) {
        fprintf(stderr, "lam_hip: LAM_HIP_SYMMETRIC: max|A - A^T| = %.3e (max|A| = %.3e): equal to rounding only -- the upper triangle "
                        "defines the system that is solved\n", asym, amax);
        return 0;
    }
    fprintf(stderr, "lam_hip: LAM_HIP_SYMMETRIC refused: the matrix is not symmetric (max|A - A^T| = %.3e, max|A| = %.3e) -- using the "
                    "general GEMV\n", asym, amax);
    c->sym_refused = true;        // until the matrix changes; do_cg_init recomputes the partial counts right after this
    return 0;
}

int create_common(lam_hip_ctx *c)
{
    {
        const int64_t before = c->opt_symmetric;
        c->opt_symmetric = symmetric_from_env(c->opt_symmetric);
        c->symmetric_from_env = c->opt_symmetric != before || (getenv("LAM_HIP_SYMMETRIC") != nullptr && c->opt_symmetric != 0);
        // (several shards / ranks run the symmetric product on the gather-Ap exchange only -- the default of both topologies; with
        // LAM_HIP_EXCHANGE=0 the option is not effective and env_symmetric_check says so)
    }
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
        if (&s == &c->sh[0])
            for (int i = 0; i < kLag; i++)
                for (auto &ev : s.ev_x[i])
                    if (hipEventCreate(&ev) != hipSuccess) return fail(nullptr, LAM_HIP_EHIP, "hipEventCreate failed");
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

// Rank mode, once per context (collective: part of lam_hip_create_rank): how many ranks of the communicator sit on THIS rank's
// GPU?  One per device in production; emulations and the MPI driver's `local_rank % device count` mapping put several there, and
// launches whose workgroups wait for each other (the fused vector steps) must then be resident for all of them together.  The ranks
// all-gather {host id, PCI bus id of the device} through the set-up scratch.
int count_ranks_on_device(lam_hip_ctx *c)
{
    c->ranks_on_device = 1;
    if (!c->rank_mode || c->nranks <= 1) return 0;
    ShardBase &s = c->sh[0];
    LAMCHK(set_dev(c, s));
    constexpr size_t kRec = 128;
    static_assert(kRec * kMaxShards <= kAgreeBytes, "device records fit the set-up scratch");
    if (c->agree_buf == nullptr) HIPCHK(c, hipMalloc((void **)&c->agree_buf, kAgreeBytes));
    std::vector<char> host(kRec * (size_t)c->nranks, 0);
    char *mine = host.data() + kRec * (size_t)c->rank;
    (void)gethostname(mine, 63);
    mine[63] = '\0';
    if (hipDeviceGetPCIBusId(mine + 64, 63, s.dev) != hipSuccess) { (void)hipGetLastError(); snprintf(mine + 64, 63, "device %d", s.dev); }
    char *dev = (char *)c->agree_buf;
    HIPCHK(c, hipMemcpyAsync(dev + kRec * (size_t)c->rank, mine, kRec, hipMemcpyHostToDevice, s.stream));
    NCCLCHK(c, ncclAllGather(dev + kRec * (size_t)c->rank, dev, kRec, ncclChar, c->comm, s.stream));
    c->n_collectives++;
    HIPCHK(c, hipMemcpyAsync(host.data(), dev, kRec * (size_t)c->nranks, hipMemcpyDeviceToHost, s.stream));
    HIPCHK(c, hipStreamSynchronize(s.stream));
    int same = 0;
    for (int q = 0; q < c->nranks; q++) same += memcmp(host.data() + kRec * (size_t)q, mine, kRec) == 0 ? 1 : 0;
    c->ranks_on_device = std::max(1, same);
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

// Exchange timing (lam_hip_stats.t_exchange; the reference's t_gemv column INCLUDES its broadcast + gather,
// ConjugateGradient_MultiGPUS_CUDA_NCCL.cu:352-377 -- here the two are separate numbers).  In an iteration whose GEMV is timed
// (option gemv_timing: every T-th) shard 0 / this rank also brackets every EXCHANGE step with a HIP-event pair on the stream the
// step runs on: the collective(s) of the rank mode, the event join(s) of one process driving several shards (from the post behind
// the producer kernel to the point where the consumer's stream has passed its waits -- what the peers' skew costs is part of it).
// Up to three steps per iteration (exchange 0: p.Ap, r.r, p slices -- the last one on the comm stream, where it overlaps the next
// GEMV's own-slice panel; gather-Ap: one).  One shard and the direct exchange have none (the latter waits inside its kernels).
bool xt_on(const lam_hip_ctx *c, const ShardBase &s) { return c->xt_slot >= 0 && &s == &c->sh[0] && s.nx[c->xt_slot] < 3; }
int xt_begin(lam_hip_ctx *c, ShardBase &s, hipStream_t st)
{
    if (!xt_on(c, s)) return 0;
    RECORD(c, s.ev_x[c->xt_slot][2 * s.nx[c->xt_slot]], st);
    return 0;
}
int xt_end(lam_hip_ctx *c, ShardBase &s, hipStream_t st)
{
    if (!xt_on(c, s)) return 0;
    RECORD(c, s.ev_x[c->xt_slot][2 * s.nx[c->xt_slot] + 1], st);
    s.nx[c->xt_slot]++;
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
    LAMCHK(xt_begin(c, s, s.stream));
    if (c->rank_mode) {
        double *buf = second ? s.gather_b : s.gather_a;
        NCCLCHK(c, ncclAllGather(buf + c->rank, buf, 1, ncclDouble, c->comm, s.stream));
        c->n_collectives++;
        LAMCHK(xt_end(c, s, s.stream));
    } else {
        RECORD(c, second ? s.ev_b : s.ev_a, s.stream);       // the step ends in reduce_wait, behind this stream's waits
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
    if (hub_active(c)) { WAITEV(c, s.stream, c->ev_join[second ? 1 : 0]); return xt_end(c, s, s.stream); }
    for (auto &t : c->sh)
        if (&t != &s) WAITEV(c, s.stream, second ? t.ev_b : t.ev_a);
    return xt_end(c, s, s.stream);
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
    LAMCHK(xt_begin(c, s, cs));
    struct Done {   // record ev_gathered on every exit path below
        lam_hip_ctx *c; ShardBase &s; hipStream_t cs;
        int finish() {
            LAMCHK(xt_end(c, s, cs));
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
    LAMCHK(xt_begin(c, s, s.stream));
    RECORD(c, s.ev_p, s.stream);
    return 0;
}
int gather_wait(lam_hip_ctx *c, ShardBase &s)
{
    if (c->rank_mode || c->total_shards == 1) return 0;
    if (hub_active(c)) { WAITEV(c, s.stream, c->ev_join[2]); return xt_end(c, s, s.stream); }
    for (auto &t : c->sh)
        if (&t != &s) WAITEV(c, s.stream, t.ev_p);
    return xt_end(c, s, s.stream);
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

namespace {

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
    static_assert(kRec * kMaxShards <= kAgreeBytes, "hello records fit the set-up scratch");
    if (c->agree_buf == nullptr) HIPCHK(c, hipMalloc((void **)&c->agree_buf, kAgreeBytes));
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

// Is the GEMV of iteration k timed (HIP-event pair on the launch stream of EVERY local shard: lam_hip_stats.t_gemv is the slowest
// shard's average -- on a real multi-GPU node that is the device that bounds the iteration)?  Option "gemv_timing" = T times every
// T-th iteration; each record is a marker packet between the iteration's kernels, so T > 1 keeps most iterations free of them.
bool timed_iteration(const lam_hip_ctx *c, const ShardBase &s, int k)
{
    (void)s;
    return c->opt_gemv_timing > 0 && (k - 1) % c->opt_gemv_timing == 0;
}

// Can a launch of `blocks` workgroups of update_fused_kernel be resident all at once?  Its workgroups wait for each
// other inside the launch (compute workgroups for the reducer's broadcast, the reducer for their partials), so a
// workgroup that cannot start until another one exits would hold everybody until the bounded waits expire.  256-thread
// workgroups are admitted per CU up to min(occupancy API, 8) (MI355X_MICROARCH.md, residency); a CU mask or a
// partitioned device that the runtime reports shows up in the CU count.  What the query cannot see (other kernels on
// the device) is still caught by the bounded waits, which end in an error, never in a hang or a silent NaN.
template <typename K>
bool launch_resident(lam_hip_ctx *c, const ShardBase &s, K kernel, int blocks)
{
    int per_cu = 0, cus = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, kBlock, 0) != hipSuccess ||
        hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, s.dev) != hipSuccess) {
        (void)hipGetLastError();
        return false;
    }
    if (c->opt_assume_cus > 0) cus = (int)c->opt_assume_cus;
    return (int64_t)std::min(per_cu, 8) * (int64_t)cus >= (int64_t)blocks;
}
template <typename TV>
bool fused_launch_resident(lam_hip_ctx *c, const ShardBase &s, int blocks)
{
    return launch_resident(c, s, update_fused_kernel<TV>, blocks);
}
// The gather-Ap exchange's fused vector step (update_full_fused_kernel): every shard launches vec_grid(n) + 1 workgroups that
// wait for each other inside the launch.  Shards that share a device (an emulation; the product gives every shard a device of its
// own) run those launches side by side on their streams, so ALL of them must fit on that device together.
template <typename TV>
bool full_fused_launch_resident(lam_hip_ctx *c)
{
    for (const auto &s : c->sh) {
        // one process: the shards of this context on the same device; rank mode: the ranks of the communicator on this rank's
        // device (count_ranks_on_device, at creation) -- 8 ranks x 257 workgroups would not fit the 2048 slots of one MI355X
        int sharing = c->rank_mode ? c->ranks_on_device : 0;
        if (!c->rank_mode) for (const auto &t : c->sh) sharing += t.dev == s.dev ? 1 : 0;
        if (!launch_resident(c, s, update_full_fused_kernel<TV>, sharing * (vec_grid(c->n) + 1))) return false;
    }
    return true;
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
        if (P > 1 && c->opt_overlap && (!I::fast_ok(c) || (a % I::VEC == 0 && (b % I::VEC == 0 || b == c->n)))) { lo = a; hi = b; }
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
// Symmetric product on several shards (option "symmetric" on the gather-Ap exchange): the record a shard gathers is not its Ap
// slice but its full-length CONTRIBUTION to A p (its rows' products over their cyclic half windows plus the mirrored products
// for the columns it touched), followed by its part of p.Ap; the full-length vector step adds the P records in shard order.
// One buffer per iteration parity like ap_gather (one process) or one (rank mode: the collective's destination).
int ensure_symv_gather(lam_hip_ctx *c, ShardBase &s)
{
    const size_t one = (size_t)c->total_shards * c->symv_stride_bytes();
    if (s.symv_gather != nullptr && s.symv_gather_bytes == one) return 0;
    LAMCHK(set_dev(c, s));
    if (s.symv_gather) { (void)hipFree(s.symv_gather); s.symv_gather = nullptr; }
    HIPCHK(c, hipMalloc(&s.symv_gather, one * (c->rank_mode ? 1 : 2)));
    s.symv_gather_bytes = one;
    return 0;
}

int enqueue_iteration_exchange1_local(lam_hip_ctx *c, int k, double rel_error, int slot)
{
    return dispatch(c, [&](auto impl) -> int {
        using I = decltype(impl);
        using TV = typename ImplTraits<I>::TV;
        const int P = c->total_shards;
        const bool sym = c->symv_multi_active();
        if (sym) for (auto &s : c->sh) LAMCHK(ensure_symv_gather(c, s));
        const uint64_t stride = sym ? c->symv_stride_bytes() : c->ex1_stride_bytes(), base = c->ex1_base();
        auto buf = [&](ShardBase &t) {
            return sym ? (char *)t.symv_gather + (size_t)(k & 1) * t.symv_gather_bytes : (char *)t.ap_gather + (size_t)(k & 1) * t.ap_gather_bytes;
        };
        for (auto &s : c->sh) {
            LAMCHK(set_dev(c, s));
            const uint64_t off = (uint64_t)s.index * stride;
            if (sym) {
                PtrList recs, tails;
                recs.n = tails.n = P;
                for (auto &t : c->sh) {
                    recs.p[t.index] = buf(t) + off;
                    tails.p[t.index] = buf(t) + off + stride - 8;
                }
                const bool timed = timed_iteration(c, s, k);
                s.split_slot[slot] = false;
                s.timed_slot[slot] = timed;
                if (timed) RECORD(c, s.ev_g0[slot], s.stream);
                Finalize fs = no_finalize(c);
                fs.active = c->opt_finalize ? 1 : 0;
                fs.slot = 0;
                fs.dst = tails;
                LAMCHK(I::launch_symv(c, s, (const TV *)s.p, nullptr, s.part_gemv, s.sc, &recs, &fs));
                if (!c->opt_finalize) {
                    hipLaunchKernelGGL(finalize_sum_kernel, dim3(1), dim3(kBlock), 0, s.stream, (const double *)s.part_gemv,
                                       I::symv_reduce_grid(c->n), tails, 0, (const CgScalars *)s.sc);
                    LAUNCHED(c);
                }
                if (timed) RECORD(c, s.ev_g1[slot], s.stream);
                LAMCHK(xt_begin(c, s, s.stream));
                if (!(c->opt_join && &s == &c->sh[0])) RECORD(c, s.ev_a, s.stream);
                continue;
            }
            Finalize f = no_finalize(c);
            f.active = c->opt_finalize ? 1 : 0;
            f.slot = 0;
            f.dst.n = P;
            PtrList yp;
            yp.n = 0;
            for (auto &t : c->sh) {
                f.dst.p[t.index] = buf(t) + off + stride - 8;
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
            LAMCHK(xt_begin(c, s, s.stream));
            if (!(c->opt_join && &s == &c->sh[0])) RECORD(c, s.ev_a, s.stream);
        }
        // the join
        if (c->opt_join) {
            ShardBase &s0 = c->sh[0];
            LAMCHK(set_dev(c, s0));
            for (auto &t : c->sh)
                if (&t != &s0) WAITEV(c, s0.stream, t.ev_a);
            LAMCHK(xt_end(c, s0, s0.stream));
            RECORD(c, c->ev_join[0], s0.stream);
        }
        const int grid = vec_grid(c->n);
        const unsigned long long seq = c->seq_base + (unsigned)k;
        c->seq_span = std::max<uint64_t>(c->seq_span, (uint64_t)k + 1);
        for (auto &s : c->sh) {
            LAMCHK(set_dev(c, s));
            if (c->opt_join) {
                if (&s != &c->sh[0]) WAITEV(c, s.stream, c->ev_join[0]);
            } else {
                for (auto &t : c->sh)
                    if (&t != &s) WAITEV(c, s.stream, t.ev_a);
                LAMCHK(xt_end(c, s, s.stream));
            }
            if (c->fuse_active) {
                // the two vector kernels in ONE launch (r.r resolved by its reducer workgroup): 2 launches per shard and iteration
                hipLaunchKernelGGL((update_full_fused_kernel<TV>), dim3(grid + 1), dim3(kBlock), 0, s.stream, (const char *)buf(s), stride,
                                   base, P, s.sc, k, rel_error, (TV *)s.p, (TV *)s.x, (TV *)s.r_full, c->n, s.row0, s.nrows, s.part_vec,
                                   grid, s.bcast, seq, c->direct_err, (volatile int *)s.host_flags, sym ? 1 : 0);
                LAUNCHED(c);
                continue;
            }
            hipLaunchKernelGGL((update_xr_full_kernel<TV>), dim3(grid), dim3(kBlock), 0, s.stream, (const char *)buf(s), stride,
                               base, P, s.sc, k, (const TV *)s.p, (TV *)s.x, (TV *)s.r_full, c->n, s.row0, s.nrows, s.part_vec, sym ? 1 : 0);
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
        // r_full = b.  Equal slices: one all-gather; the reference's uneven partition (remainder on the last rank, gathered
        // there with MPI_Allgatherv): one grouped broadcast per owner, as for p on exchange 0
        if (c->n % (uint64_t)c->nranks == 0) {
            NCCLCHK(c, ncclAllGather(s.b, s.r_full, c->ex1_base(), dt, c->comm, s.stream));
            c->n_collectives++;
        } else {
            const size_t ev = c->esz_v();
            HIPCHK(c, hipMemcpyAsync((char *)s.r_full + s.row0 * ev, s.b, s.nrows * ev, hipMemcpyDeviceToDevice, s.stream));
            NCCLCHK(c, ncclGroupStart());
            for (int q = 0; q < c->nranks; q++) {
                uint64_t r0, nr;
                partition(c->n, c->nranks, q, &r0, &nr);
                char *ptr = (char *)s.r_full + r0 * ev;
                NCCLCHK(c, ncclBroadcast(ptr, ptr, nr, dt, q, c->comm, s.stream));
                c->n_collectives++;
            }
            NCCLCHK(c, ncclGroupEnd());
        }
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
        const bool sym = c->symv_multi_active();
        if (sym) LAMCHK(ensure_symv_gather(c, s));
        const uint64_t stride = sym ? c->symv_stride_bytes() : c->ex1_stride_bytes(), base = c->ex1_base();
        char *const gathered = sym ? (char *)s.symv_gather : (char *)s.ap_gather;
        char *rec = gathered + (uint64_t)c->rank * stride;
        // 1. GEMV straight into this rank's record; its last workgroup leaves the rank's p.Ap partial
        //    behind the slice (with option "finalize" = 0: a 1-block launch does)
        Finalize f = no_finalize(c);
        f.active = c->opt_finalize ? 1 : 0;
        f.dst.n = 1; f.dst.p[0] = rec + stride - 8; f.slot = 0;
        const bool timed = timed_iteration(c, s, k);
        s.split_slot[slot] = false;
        s.timed_slot[slot] = timed;
        if (timed) RECORD(c, s.ev_g0[slot], s.stream);
        if (sym) {
            // the symmetric product's two passes leave this rank's full-length contribution and its p.Ap part in its record
            PtrList recs, tails;
            recs.n = tails.n = 1;
            recs.p[0] = rec;
            tails.p[0] = rec + stride - 8;
            Finalize fs = no_finalize(c);
            fs.active = c->opt_finalize ? 1 : 0;
            fs.slot = 0;
            fs.dst = tails;
            LAMCHK(I::launch_symv(c, s, (const TV *)s.p, nullptr, s.part_gemv, s.sc, &recs, &fs));
            if (!c->opt_finalize) {
                hipLaunchKernelGGL(finalize_sum_kernel, dim3(1), dim3(kBlock), 0, s.stream, (const double *)s.part_gemv,
                                   I::symv_reduce_grid(c->n), tails, 0, (const CgScalars *)s.sc);
                LAUNCHED(c);
            }
        } else {
        LAMCHK(I::launch_gemv(c, s, (const TV *)s.p, (TV *)rec, s.part_gemv, s.sc, 0, 0, 0, &f));
        }
        if (timed) RECORD(c, s.ev_g1[slot], s.stream);
        if (!sym && !c->opt_finalize) {
            hipLaunchKernelGGL(finalize_sum_kernel, dim3(1), dim3(kBlock), 0, s.stream, (const double *)s.part_gemv, s.gemv_blocks,
                               f.dst, 0, (const CgScalars *)s.sc);
            LAUNCHED(c);
        }
        // 2. the iteration's only collective
        LAMCHK(xt_begin(c, s, s.stream));
        NCCLCHK(c, ncclAllGather(rec, gathered, stride, ncclChar, c->comm, s.stream));
        c->n_collectives++;
        LAMCHK(xt_end(c, s, s.stream));
        // 3. alpha, x slice, FULL r (+ partials of r.r over the full vector: no collective needed)
        const int grid = vec_grid(c->n);
        if (c->fuse_active) {
            // 3 + 4 in ONE launch: GEMV, collective, vector step
            const unsigned long long seq = c->seq_base + (unsigned)k;
            c->seq_span = std::max<uint64_t>(c->seq_span, (uint64_t)k + 1);
            hipLaunchKernelGGL((update_full_fused_kernel<TV>), dim3(grid + 1), dim3(kBlock), 0, s.stream, (const char *)gathered, stride,
                               base, c->nranks, s.sc, k, rel_error, (TV *)s.p, (TV *)s.x, (TV *)s.r_full, c->n, s.row0, s.nrows, s.part_vec,
                               grid, s.bcast, seq, c->direct_err, (volatile int *)s.host_flags, sym ? 1 : 0);
            LAUNCHED(c);
            return 0;
        }
        hipLaunchKernelGGL((update_xr_full_kernel<TV>), dim3(grid), dim3(kBlock), 0, s.stream, (const char *)gathered, stride,
                           base, c->nranks, s.sc, k, (const TV *)s.p, (TV *)s.x, (TV *)s.r_full, c->n, s.row0, s.nrows, s.part_vec, sym ? 1 : 0);
        LAUNCHED(c);
        // 4. beta, stop test, FULL p
        hipLaunchKernelGGL((update_p_full_kernel<TV>), dim3(grid), dim3(kBlock), 0, s.stream, (const double *)s.part_vec, grid, s.sc,
                           k, rel_error, (const TV *)s.r_full, (TV *)s.p, c->n, (volatile int *)s.host_flags);
        LAUNCHED(c);
        return 0;
    });
}

}  // namespace
