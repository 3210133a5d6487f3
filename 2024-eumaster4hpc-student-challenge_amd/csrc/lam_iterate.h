// lam_iterate.h -- one CG iteration per shard (phases, enqueue), cg_init, the host's view of the progress word, the lag rule; tuning build: persistent launch, enqueue threads.
// Part of the one translation unit csrc/lam_hip.hip (included from there, in order; not a stand-alone header).
#pragma once

namespace {

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
            // the launch's GEMV phase is the tile body of the 2-rows-per-4-wave-workgroup shape (variant 10): it is bit-identical to
            // the two-launch chain only when that chain runs the same shape
            if (!I::fast_ok(c) || I::variant(c) != 10 || s.nrows != c->n || (c->n % 2) != 0 || (c->n % I::VEC) != 0) return 0;
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
            a.A = (const TA *)s.A; a.n = c->n; a.lda = c->lda;
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
    LAMCHK(env_symmetric_check(c));
    // how many p.Ap partials the product step leaves (= what the consumer sums) depends on which product runs: settled here, at the
    // one place where that is decided (options, a refused / re-admitted symmetric product)
    for (auto &sh_ : c->sh)
        sh_.gemv_blocks = dispatch(c, [&](auto impl) -> int { return decltype(impl)::gemv_grid(c, sh_.nrows); });
    if (c->exchange2_wanted()) {
        // the state is initialised through RCCL (one-off); the iterations then run on the mailboxes
        LAMCHK(setup_direct(c));
        c->cg_direct = c->direct_ok;
    }
    // The fused vector step (one shard; the direct exchange) is a launch whose workgroups wait for each other: used
    // only when the whole grid (compute workgroups + reducer + waiter) can be resident at once, else the two-kernel form.
    // The gather-Ap exchange has a fused vector step of its own (update_full_fused_kernel, full-length r and p).
    c->fuse_active = false;
    const bool ex1 = !c->cg_direct && c->exchange1_ok();
    if (c->opt_fuse && c->opt_finalize && (c->cg_direct || ex1 || (!c->rank_mode && c->total_shards == 1))) {
        ShardBase &s0 = c->sh[0];
        LAMCHK(set_dev(c, s0));
        c->fuse_active = dispatch(c, [&](auto impl) -> int {
            using TV = typename ImplTraits<decltype(impl)>::TV;
            if (ex1) return full_fused_launch_resident<TV>(c) ? 1 : 0;
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

int enqueue_iteration_body(lam_hip_ctx *c, int k, double rel_error, int slot);

// one iteration, all local shards; if its GEMV is timed, so are its exchange steps (xt_begin / xt_end, shard 0)
int enqueue_iteration(lam_hip_ctx *c, int k, double rel_error, int slot)
{
    ShardBase &s0 = c->sh[0];
    s0.nx[slot] = 0;
    c->xt_slot = timed_iteration(c, s0, k) ? slot : -1;
    const int rc = enqueue_iteration_body(c, k, rel_error, slot);
    c->xt_slot = -1;
    return rc;
}

int enqueue_iteration_body(lam_hip_ctx *c, int k, double rel_error, int slot)
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
// `precise` = the wait that ENDS a call (for the last enqueued iteration): there a late wake-up is latency of the call -- a
// 1 ms nap too many is 1 % of a 20-iteration run at N=65536 -- so the naps shrink with the time the awaited iteration is
// expected to need (half of it at a time, from the observed iteration time and the moment the progress word last moved) and
// the last ~300 us are polled.
int await_progress(lam_hip_ctx *c, ShardBase &s0, int target, Progress *out, bool precise = false)
{
    unsigned polls = 0;
    double t_query = 0.0, t_change = 0.0;
    int last_iters = -1;
    for (;;) {
        const Progress pr = read_progress(s0);
        if (precise && pr.iters != last_iters) { last_iters = pr.iters; t_change = now_s(); }
        if (pr.iters >= target || pr.stop_at != 0 || *(volatile int *)c->direct_err != 0) {
            // iteration-time estimate: progress made since the previous successful wait / time since then
            const double t = now_s();
            if (c->prog_t > 0.0 && pr.iters > c->prog_iter && pr.stop_at == 0) {
                const double per = (t - c->prog_t) / (double)(pr.iters - c->prog_iter);
                // smoothed, but a jump by more than 2x (another problem size, another exchange) is taken at once: a stale
                // estimate would oversleep and drain the queue
                const double e = c->iter_est_s;
                c->iter_est_s = (e > 0.0 && per < 2.0 * e && per > 0.5 * e) ? 0.75 * e + 0.25 * per : per;
            }
            c->prog_t = t;
            c->prog_iter = pr.iters;
            *out = pr;
            return 0;
        }
        if (++polls <= 64u) { __builtin_ia32_pause(); continue; }
        const double t = now_s();
        if (t_query == 0.0) t_query = t;
        // Liveness: look at the stream only after 0.25 s of waiting for ONE iteration (a fault, a lost device).  Not more often:
        // a hipStreamQuery on a busy stream makes a helper thread of the runtime wait ACTIVELY for the stream's outstanding
        // signal -- called every 2 ms it kept a second core busy for the whole solve at N=65536 (tools/thread_cpu.py).
        if (t - t_query > 0.25) {
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
        // Iterations shorter than ~60 us (N below ~7000 on one GPU): a sleep cannot be shorter than the kernel's timer slack
        // (~50 us), i.e. several iterations, and the queue would drain -- keep polling, yielding the core between polls; such
        // solves are short.  Otherwise sleep a quarter of an iteration (at most 1 ms).
        if (c->iter_est_s > 0.0 && c->iter_est_s < 60e-6) {
            if ((polls & 63u) == 0) sched_yield(); else __builtin_ia32_pause();
            continue;
        }
        double nap = std::min(1e-3, std::max(15e-6, 0.25 * c->iter_est_s));
        if (precise) {
            const double est = c->iter_est_s;
            const double left = est > 0.0 ? (double)(target - pr.iters - 1) * est + std::max(0.0, est - (t - t_change)) : 0.0;
            if (left < 300e-6) {                         // nearly there (or nothing known yet): poll
                if ((polls & 63u) == 0) sched_yield(); else __builtin_ia32_pause();
                continue;
            }
            nap = std::min(1e-3, 0.5 * left);
        }
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


// GEMV device time of the iteration that used ring slot `slot`, per local shard, if that iteration was timed; for shard 0 also the
// time of its exchange steps (xt_begin / xt_end pairs)
struct IterTimes {
    double gemv_ms[kMaxShards] = {};
    int samples[kMaxShards] = {};
    double xch_ms = 0.0;
};
static void harvest_shard(ShardBase &s, int j, int slot, IterTimes *t)
{
    if (!s.timed_slot[slot]) return;
    s.timed_slot[slot] = false;
    const int nx = j == 0 ? s.nx[slot] : 0;
    if (j == 0) s.nx[slot] = 0;
    float ms = 0.f, ms2 = 0.f;
    if (hipEventElapsedTime(&ms, s.ev_g0[slot], s.ev_g1[slot]) != hipSuccess) { (void)hipGetLastError(); return; }
    if (s.split_slot[slot] && hipEventElapsedTime(&ms2, s.ev_g2[slot], s.ev_g3[slot]) != hipSuccess) { (void)hipGetLastError(); return; }
    double xms = 0.0;
    for (int q = 0; q < nx; q++) {
        float x = 0.f;
        hipError_t e = hipEventElapsedTime(&x, s.ev_x[slot][2 * q], s.ev_x[slot][2 * q + 1]);
        if (e == hipErrorNotReady) {
            // the all-gather of p runs on the comm stream and may still be in flight when its iteration has reported (a few
            // microseconds: the next GEMV's second panel waits for it)
            (void)hipGetLastError();
            if (hipEventSynchronize(s.ev_x[slot][2 * q + 1]) == hipSuccess) e = hipEventElapsedTime(&x, s.ev_x[slot][2 * q], s.ev_x[slot][2 * q + 1]);
        }
        if (e != hipSuccess) { (void)hipGetLastError(); return; }
        xms += x;
    }
    t->gemv_ms[j] += ms + ms2;
    t->samples[j]++;
    if (j == 0) t->xch_ms += xms;
}
// all local shards (the caller's current device is restored to shard 0's)
static void harvest_times(lam_hip_ctx *c, int slot, IterTimes *t)
{
    for (size_t j = 0; j < c->sh.size(); j++) {
        if (!c->sh[j].timed_slot[slot]) continue;
        if (c->sh.size() > 1 && set_dev(c, c->sh[j]) != 0) continue;
        harvest_shard(c->sh[j], (int)j, slot, t);
    }
    if (c->sh.size() > 1) (void)set_dev(c, c->sh[0]);
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
static int iterate_threaded(lam_hip_ctx *c, int iters, int k_first, double rel_error, int *enq_out, IterTimes *times)
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
                    else harvest_shard(s, 0, slot, times);
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
}  // namespace
