// lam_kernels.h -- hand-written gfx950 (CDNA4, wave64) kernels of the dense CG hot path.
//
// Reference counterparts (semantics only; nothing here is derived from that code):
//   gemv / partialDot / reduce / dot / divide / axpy / minusaxpy / xpby CUDA kernels,
//   /root/reference/challenge/main/LAM/src/GPU/distributed/ConjugateGradient_MultiGPUS_CUDA_NCCL.cu:34-238
//   and the CPU members dot/axpby/gemv, LAM/src/CPU/ConjugateGradient_CPU_MPI_OMP.hpp:446-508.
//
// Design:
//   * one CG iteration = 3 launches on a shard, 2 where the vector step is fused into one launch -- update_fused_kernel (one
//     shard, direct exchange), update_full_fused_kernel (gather-Ap exchange: the multi-shard default of one process) --:
//       gemv_coop_kernel   Ap_loc = A_loc p          (+ per-workgroup partials of p.Ap)
//       update_xr_kernel   alpha = rr/(p.Ap); x += alpha p; r -= alpha Ap   (+ partials of r.r)
//       update_p_kernel    rr' = r.r; beta = rr'/rr; stop test; p_slice = r + beta p  (stored into
//                          every shard's replicated p)
//     alpha, beta, rr, the stop flag and the iteration counter live in device memory
//     (CgScalars); the host never has to read a scalar to enqueue the next iteration.
//   * all reductions are two-stage and fixed-order (no floating-point atomics): results are
//     bit-reproducible run to run and identical on every shard.
//   * GEMV is HBM-bound (0.25 flop/B in fp64).  Production shape (gemv_coop_kernel): the 4 waves of a
//     workgroup stream the same 2 matrix rows with 16-byte non-temporal loads, 4 KiB contiguous per row
//     per super-step; the matching 4096-column tile of p is staged once per workgroup in LDS; lane
//     partials are combined with wave64 shuffles and across waves through LDS.  gemv_tile_kernel (one
//     group of R rows per wave) is the production shape for bf16 storage and carries the tuning
//     variants; gemv_generic_kernel handles any N / alignment; gemv_mfma_bf16_kernel is the MFMA
//     experiment (slower, kept for the record).  A launch can cover one or two column panels and
//     accumulate, which is how the rank mode overlaps the all-gather of p with its own-slice panel.
//   * the *_full_kernel family implements the single-exchange iteration (gather of Ap, full-length r and p on every shard);
//     symv_*_kernel implement the opt-in symmetric product (every pair {i, j} read once: the upper triangle on one shard,
//     cyclic half windows on several row shards).
#pragma once

#include <hip/hip_runtime.h>
#include <hip/hip_bf16.h>
#include <stdint.h>

#include <type_traits>

#include "lam_host_plan.h"

namespace lam {

constexpr int kBlock = 256;          // 4 waves of 64
constexpr int kWaves = kBlock / 64;
constexpr int kMaxShards = 64;         // shards of one process / ranks of one communicator (the reference's largest published run: 64 GPUs)
constexpr size_t kAgreeBytes = 256 * (size_t)kMaxShards;   // device scratch of the small set-up collectives: one 256-byte record per rank

struct CgScalars {
    double bb;        // b.b
    double rr[2];     // ping-pong: iteration k reads rr[(k+1)&1], writes rr[k&1]
    double pAp;       // last p.Ap (diagnostics)
    double alpha;
    double beta;
    int iters;        // last completed iteration (the converging one once stop is set)
    int stop;         // set by update_p_kernel when sqrt(rr/bb) < rel_error
};

// ---------------------------------------------------------------------------------------------
// small helpers
// ---------------------------------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ T wave_sum(T v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// Sum over the workgroup; every thread returns the same value.  Fixed order -> deterministic.
__device__ __forceinline__ double block_sum(double v, double *s_red /*[kWaves]*/)
{
    v = wave_sum(v);
    __syncthreads();  // s_red may still be read from a previous call
    if ((threadIdx.x & 63) == 0 && (threadIdx.x >> 6) < kWaves) s_red[threadIdx.x >> 6] = v;
    __syncthreads();
    double t = s_red[0];
#pragma unroll
    for (int w = 1; w < kWaves; w++) t += s_red[w];
    return t;
}

// Sum of src[0..n) by one workgroup, identical on every workgroup that calls it.
__device__ __forceinline__ double block_sum_array(const double *__restrict__ src, int n, double *s_red)
{
    // eight independent loads per step (unrolled twice -> 16 in flight): with up to 32768 GEMV
    // partials a one-load-per-step loop is a chain of L2 latencies (measured 19 us per launch, 9.5 us
    // with four).  The order stays fixed, so the sum is reproducible and the same in every workgroup.
    double v = 0.0;
    int i = threadIdx.x;
#pragma unroll 2
    for (; i + 7 * kBlock < n; i += 8 * kBlock) {
        const double a0 = src[i], a1 = src[i + kBlock], a2 = src[i + 2 * kBlock], a3 = src[i + 3 * kBlock];
        const double a4 = src[i + 4 * kBlock], a5 = src[i + 5 * kBlock], a6 = src[i + 6 * kBlock], a7 = src[i + 7 * kBlock];
        v += ((a0 + a1) + (a2 + a3)) + ((a4 + a5) + (a6 + a7));
    }
    for (; i < n; i += kBlock) v += src[i];
    return block_sum(v, s_red);
}

// The iteration's progress word in PINNED HOST memory, written by ONE thread per iteration with ONE 8-byte store:
// low half = the iteration just finished, high half = that same iteration if it met the stop test (else 0).  The host
// polls it instead of waiting on an event per iteration (no marker packets between the iteration's kernels); one
// word, so it can never see a new iteration count next to an old stop decision.  After a stop the later (no-op)
// iterations return before writing, so the stopping iteration stays readable.
__device__ __forceinline__ void post_progress(volatile int *host_flags, int k, bool stop)
{
    if (host_flags == nullptr) return;
    *reinterpret_cast<volatile unsigned long long *>(host_flags) =
        ((unsigned long long)(unsigned)(stop ? k : 0) << 32) | (unsigned long long)(unsigned)k;
}

struct PtrList {                      // destinations of a replicated store (one per shard)
    void *p[kMaxShards];
    int n;
};

// In-kernel reduction of the per-workgroup partials (more than one shard only).  With one shard the
// consumer kernel sums the partials itself; with several, the shard's partial has to exist as ONE number
// before the exchange, and a separate 1-workgroup launch for that costs a launch + a kernel boundary on
// the iteration's critical path (measured 6.5 us each, profiles/r02_rank_mode_chain.csv).  Instead the
// producer launch carries ONE EXTRA workgroup, the reducer (block gridDim.x-1):
//   * every compute workgroup stores its partial with a single 8-byte agent-scope (sc1, write-through)
//     store and is done -- no fence, no counter, no wait: the value is its own "ready" flag, because the
//     slot held a sentinel (a NaN bit pattern no computation produces) until then;
//   * the reducer polls the slots with sc1 loads in the fixed order of block_sum_array -- so the sum is
//     bit-identical to a separate reduction launch --, writes the total to slot `slot` of every
//     destination and re-arms the slots with the sentinel for the next launch.
// The reducer is dispatched after the compute workgroups and only ever waits for workgroups that are
// already running; its spin is bounded (SpinGuard) and ends in a NaN total, never in a hang.
// (An arrival counter drawn by every workgroup -- store, drain, fetch_add -- was measured first: the
// two memory round trips at the end of each of 11584 GEMV workgroups cost 45 us per launch, four times
// what the removed launch had cost.)
constexpr unsigned long long kPartialSentinel = 0x7ff8dead5eedbeefull;   // quiet NaN with a payload nothing computes
// Bounded spinning with nothing slow on the fast path: the CU-local shader clock (clock64(), s_memtime -- not the
// shared constant-rate counter behind wall_clock64()) and the abort word are looked at once every 256 failed polls;
// limit ~5 s at 2.4 GHz (longer if the clock is lower -- it only bounds a wait that should never expire).
constexpr unsigned long long kSpinTimeoutCycles = 12000000000ull;
struct SpinGuard {
    unsigned spins = 0;
    unsigned long long t0 = 0;
    // once per failed poll; true once every 256 polls: the moment to look at slow things (the clock, and the
    // abort word in PINNED HOST memory -- a PCIe read: 256 workgroups doing one per poll cost 45-130 us per launch)
    __device__ __forceinline__ bool slow_path() { return (++spins & 255u) == 0; }
    __device__ __forceinline__ bool expired()          // only on the slow path
    {
        const unsigned long long now = (unsigned long long)clock64();
        if (t0 == 0) { t0 = now | 1ull; return false; }
        return now - t0 > kSpinTimeoutCycles;
    }
};

struct Finalize {                     // (the pointer list comes LAST here and in GemvArgs: the scalars keep their kernarg offsets
    int active;                       //  whatever kMaxShards is)
                                      // 0: no in-kernel reduction (the launch has no reducer workgroup)
    int mail;                         // 1: dst.p[j] is a MailSlot of rank j's mailbox (direct exchange, see Mail)
    int slot;
    unsigned long long seq;           // mail: the iteration tag
    int *host_err;                    // pinned host error word of the bounded waits (may be null): an expired wait is an
                                      // error the caller sees (lam_hip_cg_iterate), never a silent NaN
    PtrList dst;
};

// ---------------------------------------------------------------------------------------------
// Direct exchange (rank mode, option exchange = 2): no collective call inside the
// iteration.  Every rank maps every other rank's p replica and MAILBOX (HIP IPC, or plain pointers when
// the ranks are threads of one process) and the producer kernels store straight into them over xGMI:
//   * the reducer workgroup of the GEMV / update_xr launch writes the rank's partial dot product into
//     slot [rank] of EVERY rank's mailbox: value (system-scope write-through store), drain, then the tag
//     `seq` = the context's hand-over number of that iteration.  The consumer kernel (update_xr / update_p) polls the P tags
//     of its own mailbox and sums the P values in rank order: the fused, latency-optimal form of the two
//     scalar all-reduces, deterministic and bit-identical to the RCCL exchange;
//   * update_p stores its p slice into every rank's p replica with system-scope write-through stores,
//     drains them, and raises pflag[rank][workgroup] = seq in every mailbox; a 1-workgroup wait_p_kernel
//     in front of the next GEMV (behind its own-slice panel) polls those flags: the direct all-gather.
// Mailboxes are fine-grained (uncached) device memory; hand-over numbers grow by one per iteration and never restart, so nothing is ever
// reset and a slot can be read by any number of workgroups.  Every poll is bounded (SpinGuard):
// a peer that never shows up ends in an error flag in pinned host memory, not in a hang.
// ---------------------------------------------------------------------------------------------
// A {value, tag} hand-over as TWO SELF-VALIDATING 8-byte words (round 4): word h carries half h of the double's
// bits in its low 32 bits and the 32-bit tag in its high 32 bits.  Each word is written by ONE relaxed atomic store and read
// by ONE relaxed atomic load, and an 8-byte atomic object is its own ready flag: the reader takes the value only from words
// whose tag matches, so nothing is assumed about the order of accesses to DIFFERENT locations -- no fence, no drain, and no
// data race in the HIP memory model's terms.  (Rounds 2-3 wrote {value; s_waitcnt vmcnt(0); tag} and read the tag and then
// the value with two relaxed loads: correct on this hardware, which returns loads in order, but formally a race.)
// Tags are the low 32 bits of a hand-over number that grows by one per iteration over the whole life of a context
// (lam_hip.hip, seq_base): a slot is rewritten every iteration, so a stale tag is the previous iteration's, never equal.
struct MailSlot {
    unsigned long long w[2];
};
template <int SCOPE>
__device__ __forceinline__ void handover_store(MailSlot *slot, double value, unsigned long long seq)
{
    const unsigned long long bits = (unsigned long long)__double_as_longlong(value), tag = (seq & 0xffffffffull) << 32;
    __hip_atomic_store(&slot->w[0], tag | (bits & 0xffffffffull), __ATOMIC_RELAXED, SCOPE);
    __hip_atomic_store(&slot->w[1], tag | (bits >> 32), __ATOMIC_RELAXED, SCOPE);
}
// true (and the value) once BOTH words carry the tag; *seen = what the first word's tag was (diagnostics)
template <int SCOPE>
__device__ __forceinline__ bool handover_try_load(const MailSlot *slot, unsigned long long seq, double *value, unsigned *seen)
{
    const unsigned long long w0 = __hip_atomic_load(&slot->w[0], __ATOMIC_RELAXED, SCOPE);
    const unsigned long long w1 = __hip_atomic_load(&slot->w[1], __ATOMIC_RELAXED, SCOPE);
    const unsigned tag = (unsigned)(seq & 0xffffffffull);
    *seen = (unsigned)(w0 >> 32);
    if ((unsigned)(w0 >> 32) != tag || (unsigned)(w1 >> 32) != tag) return false;
    *value = __longlong_as_double((long long)((w1 << 32) | (w0 & 0xffffffffull)));
    return true;
}
struct Mail {
    MailSlot pap[kMaxShards];
    MailSlot rr[kMaxShards];
    unsigned long long pflag[kMaxShards][256];   // [source rank][its update_p workgroup] (256 == kVecBlocksMax)
};
struct MailWait {                     // consumer side of a scalar exchange; n == 0: not used
    const MailSlot *slots;            // this rank's pap[] or rr[]
    int n;
    unsigned long long seq;
    int *host_err;                    // pinned host memory
};
struct MailPost {                     // update_p: flags to raise after the p slice has been stored; n == 0: not used
    Mail *mail[kMaxShards];
    int n, rank;
    unsigned long long seq;
};

struct BlockCounts { int n[kMaxShards]; };   // update_p workgroups of every rank (p-slice flags to wait for)

__device__ __forceinline__ unsigned long long ld_sys(const unsigned long long *p)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ void st_sys(unsigned long long *p, unsigned long long v)
{
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// Sum of the P partials other ranks posted for iteration `w.seq`; every thread of every workgroup gets the
// same value.  Same reduction tree as block_sum_array(src, P), so exchange 2 and exchange 0 agree bit for bit.
__device__ __forceinline__ double mail_sum(const MailWait &w, double *s_red)
{
    double v = 0.0;
    if ((int)threadIdx.x < w.n) {
        const MailSlot *m = w.slots + threadIdx.x;
        SpinGuard guard;
        unsigned seen;
        while (!handover_try_load<__HIP_MEMORY_SCOPE_SYSTEM>(m, w.seq, &v, &seen)) {
            if (guard.slow_path()) {
            // an expired wait anywhere (this kernel or an earlier one) ends all waiting: the solve has failed
            if (*(volatile int *)w.host_err != 0) break;
            if (guard.expired()) {
                w.host_err[1] = (int)threadIdx.x; w.host_err[2] = (int)(unsigned)w.seq; w.host_err[3] = (int)seen;
                w.host_err[4] = (int)(w.seq >> 32); w.host_err[5] = 0;
                *(volatile int *)w.host_err = 2;
                break;
            }
            }
            __builtin_amdgcn_s_sleep(4);
        }
    }
    return block_sum(v, s_red);
}
__device__ __forceinline__ bool is_reducer_block(const Finalize &f) { return f.active && blockIdx.x == gridDim.x - 1; }
__device__ __forceinline__ unsigned compute_blocks(const Finalize &f) { return gridDim.x - (f.active ? 1u : 0u); }

// compute workgroups: thread 0 stores the workgroup's partial (see Finalize)
__device__ __forceinline__ void publish_partial(double t, double *partial, const Finalize &f)
{
    if (threadIdx.x != 0) return;
    if (f.active) __hip_atomic_store(partial + blockIdx.x, t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else partial[blockIdx.x] = t;
}

// the reducer workgroup (every thread of it; workgroups wider than kBlock: the extra waves idle)
__device__ __forceinline__ double reduce_partials_sum(double *partial, int n /*compute workgroups*/, double *s_red /*[kWaves]*/,
                                                      int *host_err)
{
    SpinGuard guard;
    bool timed_out = false;
    unsigned long long *slots = reinterpret_cast<unsigned long long *>(partial);
    auto ld = [&](int j) { return __hip_atomic_load(slots + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
    // CNT slots (stride kBlock): all loads of a polling round are in flight together, so the reducer is at
    // most a round trip or two behind the last producer; then the slots are re-armed
    auto take = [&](int base, auto cnt_tag, double *out) {
        constexpr int CNT = decltype(cnt_tag)::value;
        unsigned long long b[CNT];
#pragma unroll
        for (int k = 0; k < CNT; k++) b[k] = ld(base + k * kBlock);
        bool pending = true;
        while (pending && !timed_out) {
            pending = false;
#pragma unroll
            for (int k = 0; k < CNT; k++)
                if (b[k] == kPartialSentinel) { b[k] = ld(base + k * kBlock); pending |= b[k] == kPartialSentinel; }
            if (pending) {
                __builtin_amdgcn_s_sleep(8);
                if (guard.slow_path()) {
                    // another bounded wait of this solve has already expired: stop waiting (the solve has failed)
                    if (host_err != nullptr && *(volatile int *)host_err != 0) timed_out = true;
                    else if (guard.expired()) {
                        timed_out = true;
                        if (host_err != nullptr) { host_err[1] = base; host_err[2] = n; *(volatile int *)host_err = 5; }
                    }
                }
            }
        }
#pragma unroll
        for (int k = 0; k < CNT; k++) {
            __hip_atomic_store(slots + base + k * kBlock, kPartialSentinel, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            out[k] = __longlong_as_double((long long)b[k]);
        }
    };
    double v = 0.0;
    int i = threadIdx.x < kBlock ? (int)threadIdx.x : n;
    for (; i + 7 * kBlock < n; i += 8 * kBlock) {      // same grouping and order as block_sum_array
        double a[8];
        take(i, std::integral_constant<int, 8>(), a);
        v += ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
    }
    for (; i < n; i += kBlock) {
        double a[1];
        take(i, std::integral_constant<int, 1>(), a);
        v += a[0];
    }
    return block_sum(v, s_red);
}

// thread 0 of the reducer workgroup: hand the shard's total to whoever consumes it (see Finalize)
__device__ __forceinline__ void post_total(const Finalize &f, double total)
{
    if (threadIdx.x != 0) return;
    if (!f.mail) {
        for (int j = 0; j < f.dst.n; j++) reinterpret_cast<double *>(f.dst.p[j])[f.slot] = total;
    } else {
        // direct exchange: the tagged words into every rank's mailbox (system-scope write-through stores)
        for (int j = 0; j < f.dst.n; j++) handover_store<__HIP_MEMORY_SCOPE_SYSTEM>(reinterpret_cast<MailSlot *>(f.dst.p[j]), total, f.seq);
    }
}

__device__ __forceinline__ void reduce_partials(double *partial, int n, const Finalize &f, double *s_red)
{
    post_total(f, reduce_partials_sum(partial, n, s_red, f.host_err));
}

// Hand-over INSIDE a launch (update_fused_kernel): the reducer workgroup publishes a scalar for the launch's
// other workgroups in ordinary device memory -- two tagged words (MailSlot), agent-scope write-through stores -- and
// they poll them with agent-scope loads.  One thread polls; the value reaches the workgroup through block_sum.
// Every listener has a LINE OF ITS OWN (BcastLine, 128 B): hundreds of workgroups polling one word serialise at the
// memory side and the writer's store queues behind them -- measured 0.2-0.35 us per polling workgroup, 45-90 us per
// iteration with 256 of them.  The reducer's threads write the lines in parallel.
constexpr int kBcastLines = 264;      // >= 256 compute workgroups + the waiter
struct BcastLine {
    MailSlot s;
    char pad[128 - sizeof(MailSlot)];
};
__device__ __forceinline__ void bcast_post(BcastLine *lines, int nlisteners, double value, unsigned long long seq)
{
    for (int t = threadIdx.x; t < nlisteners; t += kBlock) handover_store<__HIP_MEMORY_SCOPE_AGENT>(&lines[t].s, value, seq);
}
__device__ __forceinline__ double bcast_wait(const BcastLine *line, unsigned long long seq, int *host_err, double *s_red)
{
    const MailSlot *slot = &line->s;
    double v = 0.0;
    if (threadIdx.x == 0) {
        SpinGuard guard;
        unsigned seen;
        while (!handover_try_load<__HIP_MEMORY_SCOPE_AGENT>(slot, seq, &v, &seen)) {
            if (guard.slow_path()) {
            if (*(volatile int *)host_err != 0) break;
            if (guard.expired()) {
                host_err[1] = -1; host_err[2] = (int)(unsigned)seq; host_err[3] = (int)seen;
                host_err[4] = (int)(seq >> 32); host_err[5] = 0;
                *(volatile int *)host_err = 4;
                break;
            }
            }
            __builtin_amdgcn_s_sleep(4);
        }
    }
    return block_sum(v, s_red);
}

// arm the slots (cg_init, and whenever a launch without a reducer may have written plain values)
__global__ void __launch_bounds__(kBlock)
arm_partials_kernel(double *partial, int n)
{
    for (int i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock)
        reinterpret_cast<unsigned long long *>(partial)[i] = kPartialSentinel;
}

template <typename TA> struct MatVec;  // 16-byte vector of matrix elements
template <> struct MatVec<double> {
    typedef double vec_t __attribute__((ext_vector_type(2)));
    static constexpr int N = 2;
    static __device__ __forceinline__ double get(const vec_t &v, int i) { return v[i]; }
};
template <> struct MatVec<float> {
    typedef float vec_t __attribute__((ext_vector_type(4)));
    static constexpr int N = 4;
    static __device__ __forceinline__ float get(const vec_t &v, int i) { return v[i]; }
};
// bf16 storage: 8 elements per 16 bytes, widened to fp32 by a 16-bit shift (exact)
template <> struct MatVec<__hip_bfloat16> {
    typedef unsigned short vec_t __attribute__((ext_vector_type(8)));
    static constexpr int N = 8;
    static __device__ __forceinline__ float get(const vec_t &v, int i)
    {
        return __uint_as_float(((unsigned)v[i]) << 16);
    }
};

// non-temporal scalar load of one matrix element (bf16 goes through its 16-bit pattern)
template <typename T> __device__ __forceinline__ T nt_load(const T *p) { return __builtin_nontemporal_load(p); }
template <> __device__ __forceinline__ __hip_bfloat16 nt_load<__hip_bfloat16>(const __hip_bfloat16 *p)
{
    const unsigned short bits = __builtin_nontemporal_load(reinterpret_cast<const unsigned short *>(p));
    return __builtin_bit_cast(__hip_bfloat16, bits);
}

// scalar widening of one matrix element to the vector type
template <typename TV> __device__ __forceinline__ TV widen(double v) { return (TV)v; }
template <typename TV> __device__ __forceinline__ TV widen(float v) { return (TV)v; }
template <typename TV> __device__ __forceinline__ TV widen(__hip_bfloat16 v)
{
    return (TV)__uint_as_float(((unsigned)*reinterpret_cast<const unsigned short *>(&v)) << 16);
}

// The GEMV's multiply-add, fused EXPLICITLY: left to -ffp-contract the fp32 kernels came out as packed multiplies
// followed by separate adds in some instantiations (v_pk_mul_f32 + v_pk_add_f32: twice the VALU work, and a rounding that
// depended on how the loop happened to be vectorised) and as v_pk_fma_f32 in others.  fp64 always compiled to v_fma_f64.
__device__ __forceinline__ double fma_tv(double a, double b, double c) { return __builtin_fma(a, b, c); }
__device__ __forceinline__ float fma_tv(float a, float b, float c) { return __builtin_fmaf(a, b, c); }

// ---------------------------------------------------------------------------------------------
// GEMV  y_loc = A_loc * p        (A_loc: nrows x n row-major, p: n, y_loc: nrows)
// ---------------------------------------------------------------------------------------------
template <typename TA, typename TV>
struct GemvArgs {
    const TA *A;
    const TV *p;            // full-length vector
    TV *y;                  // local rows
    double *partial;        // [gridDim.x] per-workgroup partial of sum_r y[r]*p[row0+r]; may be null
    const CgScalars *sc;    // may be null; if sc->stop the kernel does nothing
    uint64_t nrows;         // local rows
    uint64_t n;             // columns (= global N)
    uint64_t lda;           // row pitch of A in elements (>= n): rows are padded to a 4-KiB multiple (16 B for tiny rows) and the
                            // padding is ZERO, so every row starts aligned and the 16-byte-vector kernels serve any N
    uint64_t row0;          // global index of local row 0 (for the fused dot)
    // column panels: the launch covers columns [seg_begin[i], seg_end[i]) for i < nseg (nseg <= 2).
    // A whole GEMV is one segment [0,n).  In rank mode the iteration's GEMV is two launches: the
    // panel of the rank's OWN p slice first (runs while the all-gather of the other slices is in
    // flight), then the rest with accumulate = 1.
    uint64_t seg_begin[2];
    uint64_t seg_end[2];
    int nseg;
    int accumulate;         // y[row] += ... instead of y[row] = ...
    int n_ypeer;
    Finalize fin;           // in-kernel reduction of `partial` by the last workgroup (several shards)
    // gather-Ap exchange with several shards in one process: y[row] also goes, as a peer store over xGMI, to the same
    // row of this shard's record in every OTHER shard's gather buffer (n_ypeer = 0 everywhere else)
    TV *ypeer[kMaxShards - 1];
};

// the row's result: the shard's own copy and the peers' (see GemvArgs::ypeer)
template <typename TA, typename TV>
__device__ __forceinline__ void store_y(const GemvArgs<TA, TV> &a, uint64_t row, TV v)
{
    a.y[row] = v;
    for (int j = 0; j < a.n_ypeer; j++) a.ypeer[j][row] = v;
}

// Fast path: n % (16/sizeof(TA)) == 0, A and p 16-byte aligned.
//   workgroup = 4 waves, wave w owns rows (4*blockIdx+w)*R .. +R-1, all n columns.
//   TILE elements of p live in LDS at a time.
//   USE_LDS = false reads p straight from global memory (L2-resident) instead -- kept as a tuning
//   variant; ROT = false visits the tiles in natural order.
template <typename TA, typename TV, int R, int TILE, bool NT, int UNROLL, bool USE_LDS = true, bool ROT = true>
__global__ void __launch_bounds__(kBlock)
gemv_tile_kernel(GemvArgs<TA, TV> a)
{
    using MV = MatVec<TA>;
    using avec_t = typename MV::vec_t;
    constexpr int VEC = MV::N;
    constexpr int STEP = 64 * VEC;          // columns one wave instruction covers
    static_assert(TILE % STEP == 0, "tile must be a whole number of wave steps");
    constexpr int STEPS = TILE / STEP;

    __shared__ __attribute__((aligned(16))) TV s_p[USE_LDS ? TILE : 4];
    __shared__ double s_red[kWaves];

    if (a.sc != nullptr && a.sc->stop) return;
    if (is_reducer_block(a.fin)) { reduce_partials(a.partial, (int)gridDim.x - 1, a.fin, s_red); return; }

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const uint64_t row_first = ((uint64_t)blockIdx.x * kWaves + wave) * R;

    const TA *rowp[R];
#pragma unroll
    for (int r = 0; r < R; r++) {
        uint64_t row = row_first + r;
        if (row >= a.nrows) row = a.nrows - 1;   // keep loads in bounds; result discarded below
        rowp[r] = a.A + row * a.lda + (uint64_t)lane * VEC;
    }
    TV acc[R];
#pragma unroll
    for (int r = 0; r < R; r++) acc[r] = (TV)0;

    const uint32_t ntiles0 = (uint32_t)((a.seg_end[0] - a.seg_begin[0] + TILE - 1) / TILE);
    const uint32_t ntiles = ntiles0 + (a.nseg > 1 ? (uint32_t)((a.seg_end[1] - a.seg_begin[1] + TILE - 1) / TILE) : 0u);
    uint32_t tt = ROT ? blockIdx.x % ntiles : 0;     // rotated start
    for (uint32_t t = 0; t < ntiles; t++) {
        const bool second = tt >= ntiles0;
        const uint64_t c0 = second ? a.seg_begin[1] + (uint64_t)(tt - ntiles0) * TILE : a.seg_begin[0] + (uint64_t)tt * TILE;
        const uint64_t cend = second ? a.seg_end[1] : a.seg_end[0];
        const uint32_t cols = (uint32_t)((cend - c0 < (uint64_t)TILE) ? (cend - c0) : (uint64_t)TILE);
        if constexpr (USE_LDS) {
        __syncthreads();                              // previous tile fully consumed
        // stage p[c0 .. c0+cols) -- 16-byte loads, cols is a multiple of VEC (>= 16 B of TV too)
        {
            constexpr int PV = 16 / sizeof(TV);
            typedef TV pvec_t __attribute__((ext_vector_type(PV)));
            const pvec_t *src = reinterpret_cast<const pvec_t *>(a.p + c0);
            pvec_t *dst = reinterpret_cast<pvec_t *>(s_p);
            const uint32_t nv = cols / PV;
            for (uint32_t i = tid; i < nv; i += kBlock) dst[i] = src[i];
            for (uint32_t i = nv * PV + tid; i < cols; i += kBlock) s_p[i] = a.p[c0 + i];
            // ragged tile: zeros up to the next whole wave step (what the lanes past the end multiply with)
            for (uint32_t i = cols + tid; i < (uint32_t)TILE && i < (cols + STEP - 1) / STEP * STEP; i += kBlock) s_p[i] = (TV)0;
        }
        __syncthreads();
        }
        const TV *pt = USE_LDS ? (const TV *)s_p : a.p + c0;

        if (cols == TILE) {
#pragma unroll UNROLL
            for (int s = 0; s < STEPS; s++) {
                avec_t av[R];
#pragma unroll
                for (int r = 0; r < R; r++) {
                    const avec_t *src = reinterpret_cast<const avec_t *>(rowp[r] + c0 + (uint64_t)s * STEP);
                    av[r] = NT ? __builtin_nontemporal_load(src) : *src;
                }
                TV pv[VEC];
#pragma unroll
                for (int i = 0; i < VEC; i++) pv[i] = pt[s * STEP + lane * VEC + i];
#pragma unroll
                for (int r = 0; r < R; r++)
#pragma unroll
                    for (int i = 0; i < VEC; i++) acc[r] = fma_tv((TV)MV::get(av[r], i), pv[i], acc[r]);
            }
        } else {
            // Ragged last tile of a segment (N not a multiple of TILE: 18 % of the columns at N=10000): the SAME
            // streaming body over ceil(cols / STEP) steps.  Lanes past the end of the segment read the row's last
            // vector again (address clamped: in bounds) against a ZERO of p -- the p tile is zero-filled behind `cols`
            // (LDS) or the product is masked (p from L2) -- so they add +-0 and the sums keep their bits.  No second
            // code path with its own registers: round 2's separately unrolled ragged body doubled the kernels' VGPRs
            // (fp32 54 -> 98, bf16 104 -> 256: occupancy 8 -> 4 and 4 -> 1) and the full-tile path paid for it.
            const int nsteps = (int)((cols + STEP - 1) / STEP);
            const uint32_t last = cols - VEC;                 // cols is a multiple of VEC (>= VEC)
#pragma unroll UNROLL
            for (int s = 0; s < nsteps; s++) {
                const uint32_t col = (uint32_t)s * STEP + (uint32_t)lane * VEC;
                const uint32_t colc = col < last ? col : last;
                avec_t av[R];
#pragma unroll
                for (int r = 0; r < R; r++) {
                    const avec_t *src = reinterpret_cast<const avec_t *>(rowp[r] - (uint64_t)lane * VEC + c0 + colc);
                    av[r] = NT ? __builtin_nontemporal_load(src) : *src;
                }
                TV pv[VEC];
#pragma unroll
                for (int i = 0; i < VEC; i++) pv[i] = USE_LDS ? pt[col + i] : (col < cols ? pt[col + i] : (TV)0);
#pragma unroll
                for (int r = 0; r < R; r++)
#pragma unroll
                    for (int i = 0; i < VEC; i++) acc[r] = fma_tv((TV)MV::get(av[r], i), pv[i], acc[r]);
            }
        }
        tt = (tt + 1 == ntiles) ? 0 : tt + 1;
    }

    // lane partials -> row sums (wave64 butterfly), fused p.Ap partial
    double dotp = 0.0;
#pragma unroll
    for (int r = 0; r < R; r++) {
        TV s = wave_sum(acc[r]);
        const uint64_t row = row_first + r;
        if (lane == 0 && row < a.nrows) {
            if (a.accumulate) s += a.y[row];
            store_y(a, row, s);
            dotp += (double)s * (double)a.p[a.row0 + row];
        }
    }
    if (a.partial != nullptr) {
        __syncthreads();
        if (lane == 0) s_red[wave] = dotp;
        __syncthreads();
        double t = 0.0;
        if (tid == 0) {
            t = s_red[0];
#pragma unroll
            for (int w = 1; w < kWaves; w++) t += s_red[w];
        }
        publish_partial(t, a.partial, a.fin);
    }
}

// Cooperative-row variant: the 4 waves of a workgroup stream the SAME R rows, wave w taking every
// 4th 1-KiB step, so a workgroup reads 4 KiB contiguous per row per super-step and the chip has
// 4x fewer concurrent row streams than the wave-per-row shape (DRAM page locality experiment).
// The row sums are combined across waves through LDS in a fixed order.
template <typename TA, typename TV, int R, int TILE, bool NT, int UNROLL, int WAVES = 4>
__global__ void __launch_bounds__(WAVES * 64)
gemv_coop_kernel(GemvArgs<TA, TV> a)
{
    using MV = MatVec<TA>;
    using avec_t = typename MV::vec_t;
    constexpr int VEC = MV::N;
    constexpr int STEP = 64 * VEC;
    static_assert(TILE % (STEP * WAVES) == 0, "tile must be a whole number of workgroup super-steps");
    constexpr int WSTEPS = TILE / (STEP * WAVES);
    constexpr int NTHREADS = WAVES * 64;    // steps per wave per tile

    __shared__ __attribute__((aligned(16))) TV s_p[TILE];
    __shared__ TV s_part[R][WAVES];
    __shared__ double s_dot[R];
    __shared__ double s_red[kWaves];

    if (a.sc != nullptr && a.sc->stop) return;
    if (is_reducer_block(a.fin)) { reduce_partials(a.partial, (int)gridDim.x - 1, a.fin, s_red); return; }

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const uint64_t row_first = (uint64_t)blockIdx.x * R;
    const uint32_t woff = (uint32_t)wave * STEP + (uint32_t)lane * VEC;   // this lane's column inside a super-step

    const TA *rowp[R];
#pragma unroll
    for (int r = 0; r < R; r++) {
        uint64_t row = row_first + r;
        if (row >= a.nrows) row = a.nrows - 1;
        rowp[r] = a.A + row * a.lda + woff;
    }
    TV acc[R];
#pragma unroll
    for (int r = 0; r < R; r++) acc[r] = (TV)0;

    const uint32_t ntiles0 = (uint32_t)((a.seg_end[0] - a.seg_begin[0] + TILE - 1) / TILE);
    const uint32_t ntiles = ntiles0 + (a.nseg > 1 ? (uint32_t)((a.seg_end[1] - a.seg_begin[1] + TILE - 1) / TILE) : 0u);
    uint32_t tt = blockIdx.x % ntiles;
    for (uint32_t t = 0; t < ntiles; t++) {
        const bool second = tt >= ntiles0;
        const uint64_t c0 = second ? a.seg_begin[1] + (uint64_t)(tt - ntiles0) * TILE : a.seg_begin[0] + (uint64_t)tt * TILE;
        const uint64_t cend = second ? a.seg_end[1] : a.seg_end[0];
        const uint32_t cols = (uint32_t)((cend - c0 < (uint64_t)TILE) ? (cend - c0) : (uint64_t)TILE);
        __syncthreads();
        {
            constexpr int PV = 16 / sizeof(TV);
            typedef TV pvec_t __attribute__((ext_vector_type(PV)));
            const pvec_t *src = reinterpret_cast<const pvec_t *>(a.p + c0);
            pvec_t *dst = reinterpret_cast<pvec_t *>(s_p);
            const uint32_t nv = cols / PV;
            for (uint32_t i = tid; i < nv; i += NTHREADS) dst[i] = src[i];
            for (uint32_t i = nv * PV + tid; i < cols; i += NTHREADS) s_p[i] = a.p[c0 + i];
            // ragged tile: zeros up to the next whole super-step (what the lanes past the end multiply with)
            for (uint32_t i = cols + tid; i < (uint32_t)TILE && i < (cols + STEP * WAVES - 1) / (STEP * WAVES) * (STEP * WAVES); i += NTHREADS)
                s_p[i] = (TV)0;
        }
        __syncthreads();
        if (cols == TILE) {
#pragma unroll UNROLL
            for (int s = 0; s < WSTEPS; s++) {
                avec_t av[R];
#pragma unroll
                for (int r = 0; r < R; r++) {
                    const avec_t *src = reinterpret_cast<const avec_t *>(rowp[r] + c0 + (uint64_t)s * (STEP * WAVES));
                    av[r] = NT ? __builtin_nontemporal_load(src) : *src;
                }
                TV pv[VEC];
#pragma unroll
                for (int i = 0; i < VEC; i++) pv[i] = s_p[s * (STEP * WAVES) + woff + i];
#pragma unroll
                for (int r = 0; r < R; r++)
#pragma unroll
                    for (int i = 0; i < VEC; i++) acc[r] = fma_tv((TV)MV::get(av[r], i), pv[i], acc[r]);
            }
        } else {
            // ragged last tile of a segment: the same streaming body over ceil(cols / super-step) super-steps; lanes past
            // the end re-read the row's last vector (clamped address) against the zeros the staging put behind `cols`,
            // so they add +-0 (see gemv_tile_kernel: no second unrolled body, no second set of registers)
            const int nsteps = (int)((cols + STEP * WAVES - 1) / (STEP * WAVES));
            const uint32_t last = cols - VEC;
#pragma unroll UNROLL
            for (int s = 0; s < nsteps; s++) {
                const uint32_t col = (uint32_t)s * (STEP * WAVES) + woff;
                const uint32_t colc = col < last ? col : last;
                avec_t av[R];
#pragma unroll
                for (int r = 0; r < R; r++) {
                    const avec_t *src = reinterpret_cast<const avec_t *>(rowp[r] - woff + c0 + colc);
                    av[r] = NT ? __builtin_nontemporal_load(src) : *src;
                }
                TV pv[VEC];
#pragma unroll
                for (int i = 0; i < VEC; i++) pv[i] = s_p[col + i];
#pragma unroll
                for (int r = 0; r < R; r++)
#pragma unroll
                    for (int i = 0; i < VEC; i++) acc[r] = fma_tv((TV)MV::get(av[r], i), pv[i], acc[r]);
            }
        }
        tt = (tt + 1 == ntiles) ? 0 : tt + 1;
    }

#pragma unroll
    for (int r = 0; r < R; r++) {
        TV s = wave_sum(acc[r]);
        if (lane == 0) s_part[r][wave] = s;
    }
    __syncthreads();
    if (tid < R) {
        const uint64_t row = row_first + tid;
        double d = 0.0;
        if (row < a.nrows) {
            TV s = s_part[tid][0];
#pragma unroll
            for (int w = 1; w < WAVES; w++) s += s_part[tid][w];
            if (a.accumulate) s += a.y[row];
            store_y(a, row, s);
            d = (double)s * (double)a.p[a.row0 + row];
        }
        s_dot[tid] = d;
    }
    if (a.partial != nullptr) {
        __syncthreads();
        double t = 0.0;
        if (tid == 0) {
            t = s_dot[0];
#pragma unroll
            for (int r = 1; r < R; r++) t += s_dot[r];
        }
        publish_partial(t, a.partial, a.fin);
    }
}

// ==== From here to the matching #endif: TUNING BUILD ONLY (-DLAM_TUNING_VARIANTS, `make tuning` -> liblam_hip_tuning.so).
// Experiments that were built, tested, measured and did not win: the grouped cooperative-row
// probe, the MFMA-fed bf16 GEMV.  The product library does not carry them.
#ifdef LAM_TUNING_VARIANTS
// One p tile's worth of the cooperative-row stream as a function: R rows, columns [c0, c0 + cols) of each, accumulated
// into acc[R] -- statement for statement the tile body of gemv_coop_kernel (which keeps its own inline copy: moving it
// behind this call changed the production kernels' register allocation), so that cg_persist_kernel adds a row's
// products in exactly the same order.  rowp[r] points at the row's element `woff` (this lane's column inside a
// super-step); s_p holds p[c0 ...] with zeros behind `cols` up to the next whole super-step.
template <typename TA, typename TV, int R, int TILE, bool NT, int UNROLL, int WAVES>
__device__ __forceinline__ void coop_stream_tile(const TA *const (&rowp)[R], uint64_t c0, uint32_t cols, const TV *s_p, uint32_t woff, TV (&acc)[R])
{
    using MV = MatVec<TA>;
    using avec_t = typename MV::vec_t;
    constexpr int VEC = MV::N;
    constexpr int STEP = 64 * VEC;
    constexpr int WSTEPS = TILE / (STEP * WAVES);
    if (cols == TILE) {
#pragma unroll UNROLL
        for (int s = 0; s < WSTEPS; s++) {
            avec_t av[R];
#pragma unroll
            for (int r = 0; r < R; r++) {
                const avec_t *src = reinterpret_cast<const avec_t *>(rowp[r] + c0 + (uint64_t)s * (STEP * WAVES));
                av[r] = NT ? __builtin_nontemporal_load(src) : *src;
            }
            TV pv[VEC];
#pragma unroll
            for (int i = 0; i < VEC; i++) pv[i] = s_p[s * (STEP * WAVES) + woff + i];
#pragma unroll
            for (int r = 0; r < R; r++)
#pragma unroll
                for (int i = 0; i < VEC; i++) acc[r] = fma_tv((TV)MV::get(av[r], i), pv[i], acc[r]);
        }
    } else {
        const int nsteps = (int)((cols + STEP * WAVES - 1) / (STEP * WAVES));
        const uint32_t last = cols - VEC;
#pragma unroll UNROLL
        for (int s = 0; s < nsteps; s++) {
            const uint32_t col = (uint32_t)s * (STEP * WAVES) + woff;
            const uint32_t colc = col < last ? col : last;
            avec_t av[R];
#pragma unroll
            for (int r = 0; r < R; r++) {
                const avec_t *src = reinterpret_cast<const avec_t *>(rowp[r] - woff + c0 + colc);
                av[r] = NT ? __builtin_nontemporal_load(src) : *src;
            }
            TV pv[VEC];
#pragma unroll
            for (int i = 0; i < VEC; i++) pv[i] = s_p[col + i];
#pragma unroll
            for (int r = 0; r < R; r++)
#pragma unroll
                for (int i = 0; i < VEC; i++) acc[r] = fma_tv((TV)MV::get(av[r], i), pv[i], acc[r]);
        }
    }
}

// Tuning probe: the cooperative-row GEMV with GROUP consecutive row pairs per workgroup that
// share every staged p tile -- does halving / quartering the number of workgroups (launches, epilogues, staged tiles) lift
// the short-row sizes?  Same per-row arithmetic; the rotated tile order starts at blockIdx (not pair index) % tiles, so the
// bits differ from the production kernel's.
template <typename TA, typename TV, int GROUP>
__global__ void __launch_bounds__(kBlock)
gemv_coop_group_kernel(GemvArgs<TA, TV> a)
{
    using MV = MatVec<TA>;
    constexpr int VEC = MV::N, R = 2, TILE = 4096, WAVES = 4, UNROLL = 4;
    constexpr int STEP = 64 * VEC;
    __shared__ __attribute__((aligned(16))) TV s_p[TILE];
    __shared__ TV s_part[GROUP][R][WAVES];
    __shared__ double s_dot[GROUP * R];
    __shared__ double s_red[kWaves];
    if (a.sc != nullptr && a.sc->stop) return;
    if (is_reducer_block(a.fin)) { reduce_partials(a.partial, (int)gridDim.x - 1, a.fin, s_red); return; }
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint64_t n = a.n;
    const uint64_t row_first = (uint64_t)blockIdx.x * R * GROUP;
    const uint32_t woff = (uint32_t)wave * STEP + (uint32_t)lane * VEC;
    TV acc[GROUP][R];
#pragma unroll
    for (int g = 0; g < GROUP; g++)
#pragma unroll
        for (int r = 0; r < R; r++) acc[g][r] = (TV)0;
    const uint32_t ntiles = (uint32_t)((n + TILE - 1) / TILE);
    uint32_t tt = blockIdx.x % ntiles;
    for (uint32_t t = 0; t < ntiles; t++) {
        const uint64_t c0 = (uint64_t)tt * TILE;
        const uint32_t cols = (uint32_t)((n - c0 < (uint64_t)TILE) ? (n - c0) : (uint64_t)TILE);
        __syncthreads();
        {
            constexpr int PV = 16 / sizeof(TV);
            typedef TV pvec_t __attribute__((ext_vector_type(PV)));
            const pvec_t *src = reinterpret_cast<const pvec_t *>(a.p + c0);
            pvec_t *dst = reinterpret_cast<pvec_t *>(s_p);
            const uint32_t nv = cols / PV;
            for (uint32_t i = tid; i < nv; i += kBlock) dst[i] = src[i];
            for (uint32_t i = nv * PV + tid; i < cols; i += kBlock) s_p[i] = a.p[c0 + i];
            for (uint32_t i = cols + tid; i < (uint32_t)TILE && i < (cols + STEP * WAVES - 1) / (STEP * WAVES) * (STEP * WAVES); i += kBlock) s_p[i] = (TV)0;
        }
        __syncthreads();
#pragma unroll
        for (int g = 0; g < GROUP; g++) {
            const TA *rowp[R];
#pragma unroll
            for (int r = 0; r < R; r++) {
                uint64_t row = row_first + (uint64_t)g * R + r;
                if (row >= a.nrows) row = a.nrows - 1;
                rowp[r] = a.A + row * a.lda + woff;
            }
            coop_stream_tile<TA, TV, R, TILE, true, UNROLL, WAVES>(rowp, c0, cols, s_p, woff, acc[g]);
        }
        tt = (tt + 1 == ntiles) ? 0 : tt + 1;
    }
#pragma unroll
    for (int g = 0; g < GROUP; g++)
#pragma unroll
        for (int r = 0; r < R; r++) {
            const TV sacc = wave_sum(acc[g][r]);
            if (lane == 0) s_part[g][r][wave] = sacc;
        }
    __syncthreads();
    if (tid < GROUP * R) {
        const int g = tid / R, r = tid % R;
        const uint64_t row = row_first + (uint64_t)g * R + r;
        double d = 0.0;
        if (row < a.nrows) {
            TV sum = s_part[g][r][0];
#pragma unroll
            for (int wv = 1; wv < WAVES; wv++) sum += s_part[g][r][wv];
            store_y(a, row, sum);
            d = (double)sum * (double)a.p[a.row0 + row];
        }
        s_dot[tid] = d;
    }
    if (a.partial != nullptr) {
        __syncthreads();
        double t = 0.0;
        if (tid == 0)
            for (int i = 0; i < GROUP * R; i++) t += s_dot[i];
        publish_partial(t, a.partial, a.fin);
    }
}

// ---------------------------------------------------------------------------------------------
// MFMA experiment for the bf16-storage GEMV (BASELINE configs[3]): can the matrix cores take the
// widening + FMA work off the VALU?  A wave loads 1 KiB = 512 contiguous bf16 of ONE matrix row
// (perfectly coalesced, exactly like the VALU kernels) and feeds the raw bytes as the A fragment of
// v_mfma_f32_32x32x16_bf16: lane l (r = l&31, h = l>>5) holds A[r][8h+j] = row[c + 8l + j].  The B
// fragment is the bf16 image of p at the SAME columns, B[8h+j][r] = p[c + 8l + j], so
//     D[m][n] = sum_{h,j} row[c + 8(32h+m) + j] * p[c + 8(32h+n) + j]
// and the DIAGONAL D[m][m] is the partial dot product of lanes m and m+32: trace(D) accumulated over
// the row is the row's dot product.  31/32 of the MFMA flops are discarded -- they are free, the
// kernel stays HBM-bound.  p is fp32: SPLIT = 3 feeds it as three bf16 terms (p = hi + mid + lo
// exactly, 8 mantissa bits each), i.e. three MFMAs per KiB and fp32-exact products; SPLIT = 1 rounds
// p to bf16 (classical bf16 x bf16 -> fp32 GEMV).
// ---------------------------------------------------------------------------------------------
typedef __bf16 mfma_bf16x8 __attribute__((ext_vector_type(8)));
typedef float mfma_f32x16 __attribute__((ext_vector_type(16)));

template <int R, int TILE, bool NT, int SPLIT>
__global__ void __launch_bounds__(kBlock)
gemv_mfma_bf16_kernel(GemvArgs<__hip_bfloat16, float> a)
{
    typedef unsigned short u16x8 __attribute__((ext_vector_type(8)));
    constexpr int STEP = 512;                       // bf16 elements per wave instruction (1 KiB)
    static_assert(TILE % STEP == 0, "tile must be a whole number of wave steps");
    constexpr int STEPS = TILE / STEP;

    __shared__ __attribute__((aligned(16))) unsigned short s_pb[SPLIT][TILE];
    __shared__ double s_red[kWaves];

    if (a.sc != nullptr && a.sc->stop) return;
    if (is_reducer_block(a.fin)) { reduce_partials(a.partial, (int)gridDim.x - 1, a.fin, s_red); return; }

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const uint64_t row_first = ((uint64_t)blockIdx.x * kWaves + wave) * R;

    const __hip_bfloat16 *rowp[R];
#pragma unroll
    for (int r = 0; r < R; r++) {
        uint64_t row = row_first + r;
        if (row >= a.nrows) row = a.nrows - 1;
        rowp[r] = a.A + row * a.lda + (uint64_t)lane * 8;
    }
    mfma_f32x16 acc[R];
#pragma unroll
    for (int r = 0; r < R; r++)
#pragma unroll
        for (int i = 0; i < 16; i++) acc[r][i] = 0.f;
    float tail[R];                                   // columns of a ragged last tile (VALU)
#pragma unroll
    for (int r = 0; r < R; r++) tail[r] = 0.f;

    const uint32_t ntiles0 = (uint32_t)((a.seg_end[0] - a.seg_begin[0] + TILE - 1) / TILE);
    const uint32_t ntiles = ntiles0 + (a.nseg > 1 ? (uint32_t)((a.seg_end[1] - a.seg_begin[1] + TILE - 1) / TILE) : 0u);
    uint32_t tt = blockIdx.x % ntiles;
    for (uint32_t t = 0; t < ntiles; t++) {
        const bool second = tt >= ntiles0;
        const uint64_t c0 = second ? a.seg_begin[1] + (uint64_t)(tt - ntiles0) * TILE : a.seg_begin[0] + (uint64_t)tt * TILE;
        const uint64_t cend = second ? a.seg_end[1] : a.seg_end[0];
        const uint32_t cols = (uint32_t)((cend - c0 < (uint64_t)TILE) ? (cend - c0) : (uint64_t)TILE);
        __syncthreads();
        for (uint32_t i = tid; i < (uint32_t)TILE; i += kBlock) {
            float v = i < cols ? a.p[c0 + i] : 0.f;            // zero padding: padded columns add 0
            if (SPLIT == 1) {
                s_pb[0][i] = __builtin_bit_cast(unsigned short, __float2bfloat16(v));
            } else {
#pragma unroll
                for (int sp = 0; sp < SPLIT; sp++) {
                    const unsigned bits = __float_as_uint(v) & 0xFFFF0000u;   // truncate to bf16: exact remainder
                    s_pb[sp][i] = (unsigned short)(bits >> 16);
                    v -= __uint_as_float(bits);
                }
            }
        }
        __syncthreads();
        const int full_steps = (int)(cols / STEP);
        auto step = [&](int s) {
            u16x8 av[R];
#pragma unroll
            for (int r = 0; r < R; r++) {
                const u16x8 *src = reinterpret_cast<const u16x8 *>(rowp[r] + c0 + (uint64_t)s * STEP);
                av[r] = NT ? __builtin_nontemporal_load(src) : *src;
            }
#pragma unroll
            for (int sp = 0; sp < SPLIT; sp++) {
                const u16x8 bv = *reinterpret_cast<const u16x8 *>(&s_pb[sp][s * STEP + lane * 8]);
#pragma unroll
                for (int r = 0; r < R; r++)
                    acc[r] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(mfma_bf16x8, av[r]),
                                                                     __builtin_bit_cast(mfma_bf16x8, bv), acc[r], 0, 0, 0);
            }
        };
        if (cols == TILE) {
#pragma unroll 4
            for (int s = 0; s < STEPS; s++) step(s);
        } else {
            for (int s = 0; s < full_steps; s++) step(s);
        }
        // ragged remainder of the last tile (cols % 512 != 0): plain VALU on the fp32 p
        for (uint32_t c = (uint32_t)full_steps * STEP + lane * 8; c < cols; c += STEP) {
#pragma unroll
            for (int r = 0; r < R; r++) {
                const u16x8 av = *reinterpret_cast<const u16x8 *>(rowp[r] + c0 + c - (uint64_t)lane * 8);
#pragma unroll
                for (int i = 0; i < 8; i++) tail[r] += __uint_as_float(((unsigned)av[i]) << 16) * a.p[c0 + c + i];
            }
        }
        tt = (tt + 1 == ntiles) ? 0 : tt + 1;
    }

    // trace of the accumulator: lane l (n = l&31, h = l>>5) holds D[n][n] in register
    // reg = ((n-4h)&3) + 4*((n-4h)>>3) when ((n-4h)&7) < 4  (C/D map: row = (reg&3) + 8*(reg>>2) + 4h)
    const int nn = lane & 31, hh = lane >> 5;
    const int d = nn - 4 * hh;
    const bool has_diag = d >= 0 && (d & 7) < 4;
    const int myreg = (d & 3) + 4 * (d >> 3);
    double dotp = 0.0;
#pragma unroll
    for (int r = 0; r < R; r++) {
        float v = 0.f;
#pragma unroll
        for (int i = 0; i < 16; i++) v = (has_diag && i == myreg) ? acc[r][i] : v;
        float s = wave_sum(v + tail[r]);
        const uint64_t row = row_first + r;
        if (lane == 0 && row < a.nrows) {
            if (a.accumulate) s += a.y[row];
            store_y(a, row, s);
            dotp += (double)s * (double)a.p[a.row0 + row];
        }
    }
    if (a.partial != nullptr) {
        __syncthreads();
        if (lane == 0) s_red[wave] = dotp;
        __syncthreads();
        double t = 0.0;
        if (tid == 0) {
            t = s_red[0];
#pragma unroll
            for (int w = 1; w < kWaves; w++) t += s_red[w];
        }
        publish_partial(t, a.partial, a.fin);
    }
}

#endif  // LAM_TUNING_VARIANTS (grouped cooperative rows, MFMA-fed bf16 GEMV)

// ---------------------------------------------------------------------------------------------
// Symmetric product (option "symmetric", single shard): y = A p reading only the UPPER triangle.
// CG needs A symmetric positive definite, so A[i][j] (j > i) can serve both y_i += A_ij p_j and
// y_j += A_ij p_i: 17.2 GB per product instead of 34.4 GB at N=65536 fp64 -- the only way past the
// HBM roofline of the plain GEMV.  Two passes (round 4: second shape; the round-1 shape flushed a column
// partial per 32 rows -- 0.55 GB written and read again per product -- and kept 32 row partials per lane):
//   symv_task_kernel    one workgroup per TASK = a column strip (NV x 4 KiB per row, read contiguously by the
//                       4 waves; the product uses NV = 1: 512 fp64 / 1024 fp32 columns) x a run of rows (2048 for the
//                       bulk of the triangle, 512 from 60 % of the work on, 64 for the last 8 %, so that the launch
//                       ends on short tasks; shorter runs for small N).  A lane keeps the column partials of its
//                       columns in registers over ALL rows of the task and 8 row partials at a time: every 8 rows they
//                       are summed across the wave by a transposed butterfly (wave_sum8: 7 exchanged values
//                       for 8 rows instead of 48) and parked in LDS, which is flushed every 256 rows.  Tasks are
//                       dispatched row run by row run, all strips of a run side by side: whole rows stream, as
//                       in the GEMV.  Elements left of the diagonal are masked (not even read where a whole
//                       16-byte vector lies left of it); the diagonal counts once.
//   symv_reduce_kernel  y[i] = the row partials of i's row run + the column partials of the tasks of i's strip,
//                       fixed order (deterministic), plus the workgroup's partial of p.y.
// Partials: ~70 MB written and read per product at N=65536, indexed by task in dispatch order (see symv_reduce_kernel:
// their stores are what separates the first pass from the rate of its loads alone).  Any N (rows are padded to whole
// vectors with zeros, p likewise).  The caller asserts symmetry (lam_hip_check_symmetry measures it).
// ---------------------------------------------------------------------------------------------
constexpr int kSymvRowsLds = 256;      // row partials parked in LDS before they are stored (a task flushes them every 256 rows)
constexpr int kSymvRowsMax = 2048;     // rows of the tallest task
constexpr int kSymvListChunk = 1024;   // entries of a strip's task list the second pass stages in LDS at a time
constexpr int kSymvReduceRows = 32;    // rows per workgroup of the second pass = p.Ap partials per product: ceil(n / 32)
// SymvTask, SymvIndex, the task flags and the use rule symv_use<CYC> live in lam_host_plan.h: the host's planner and its exhaustive
// check are built from the same definitions, also by plain g++ under AddressSanitizer (tests/host_asan).

// lane exchanges of the transposed butterfly.  gfx950: v_permlane32_swap / v_permlane16_swap move both directions of a
// halving step in one instruction (no select, no LDS crossbar); inside a 16-lane row DPP: rotation by 8 (= lane ^ 8),
// half-row mirror (lane -> 7 - lane within 8 lanes), quad permutes (lane ^ 2, lane ^ 1).
template <int CTRL> __device__ __forceinline__ unsigned dpp_u32(unsigned v) { return __builtin_amdgcn_update_dpp(0u, v, CTRL, 0xf, 0xf, false); }
template <int CTRL> __device__ __forceinline__ float dpp_t(float v) { return __uint_as_float(dpp_u32<CTRL>(__float_as_uint(v))); }
template <int CTRL> __device__ __forceinline__ double dpp_t(double v)
{
    const unsigned long long u = (unsigned long long)__double_as_longlong(v);
    const unsigned lo = dpp_u32<CTRL>((unsigned)u), hi = dpp_u32<CTRL>((unsigned)(u >> 32));
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
template <int W> __device__ __forceinline__ void permlane_swap(unsigned &x, unsigned &y)
{
    if (W == 32) { auto r = __builtin_amdgcn_permlane32_swap(x, y, false, false); x = r[0]; y = r[1]; }
    else { auto r = __builtin_amdgcn_permlane16_swap(x, y, false, false); x = r[0]; y = r[1]; }
}
// x belongs to the lower half (of 2W lanes), y to the upper: lower lanes return x(l) + x(l + W), upper lanes y(l - W) + y(l)
template <int W> __device__ __forceinline__ float fold_halves(float x, float y)
{
    unsigned a = __float_as_uint(x), b = __float_as_uint(y);
    permlane_swap<W>(a, b);
    return __uint_as_float(a) + __uint_as_float(b);
}
template <int W> __device__ __forceinline__ double fold_halves(double x, double y)
{
    const unsigned long long ux = (unsigned long long)__double_as_longlong(x), uy = (unsigned long long)__double_as_longlong(y);
    unsigned xl = (unsigned)ux, xh = (unsigned)(ux >> 32), yl = (unsigned)uy, yh = (unsigned)(uy >> 32);
    permlane_swap<W>(xl, yl);
    permlane_swap<W>(xh, yh);
    return __longlong_as_double((long long)(((unsigned long long)xh << 32) | xl)) +
           __longlong_as_double((long long)(((unsigned long long)yh << 32) | yl));
}
// 8 per-lane values -> lane L returns the wave total of value (L >> 3) & 7.  Fixed order.
template <typename T>
__device__ __forceinline__ T wave_sum8(const T (&v)[8])
{
    const bool b3 = threadIdx.x & 8;
    T w[4], u[2];
#pragma unroll
    for (int j = 0; j < 4; j++) w[j] = fold_halves<32>(v[j], v[4 + j]);
#pragma unroll
    for (int j = 0; j < 2; j++) u[j] = fold_halves<16>(w[j], w[2 + j]);
    const T keep = b3 ? u[1] : u[0], send = b3 ? u[0] : u[1];
    T t = keep + dpp_t<0x128>(send);      // row_ror:8
    t += dpp_t<0x141>(t);                 // row_half_mirror
    t += dpp_t<0x4e>(t);                  // quad_perm [2,3,0,1]
    t += dpp_t<0xb1>(t);                  // quad_perm [1,0,3,2]
    return t;
}

// Launched with kSymvLdsPerWorkgroup bytes of LDS in all (its row-partial buffer + idle dynamic LDS), i.e. THREE workgroups per
// CU: the tasks' loads alone run at 7.22 TB/s with three workgroups per CU against 7.0 with the five its registers allow, the
// whole pass 1-2 % faster (fewer row streams in flight at a time; profiles/r04_symv2_probe.txt).
constexpr unsigned kSymvLdsPerWorkgroup = 53 * 1024;
template <typename T> constexpr unsigned symv_lds_pad() { return kSymvLdsPerWorkgroup - (unsigned)sizeof(T) * kWaves * kSymvRowsLds; }   // T: vector type
template <typename TA, typename T, int NV, bool CYC>     // TA: storage of the matrix; T: vectors, products, partials (bf16 storage: float)
__global__ void __launch_bounds__(kBlock)
symv_task_kernel(const TA *__restrict__ A, const T *__restrict__ p, const SymvTask *__restrict__ tasks, T *__restrict__ rowpart,
                 T *__restrict__ colpart, uint64_t lda, uint64_t ncols_vec, uint64_t n, uint64_t row_off, const CgScalars *sc)
{
    using MV = MatVec<TA>;
    using vec_t = typename MV::vec_t;                                // 16 bytes of matrix elements
    constexpr int VEC = MV::N, CW = kBlock * VEC, SS = NV * CW;     // CW: columns the workgroup covers with one vector per lane
    typedef T tvec_t __attribute__((ext_vector_type(VEC)));          // the same columns of a vector / a column partial
    __shared__ T s_rows[kWaves][kSymvRowsLds];
    if (sc != nullptr && sc->stop) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    SymvTask t = tasks[blockIdx.x];                                                  // row0: LOCAL row (A, rowpart); global = row_off + row0
    const bool interior = (t.nrows & kSymvInterior) != 0, full = (t.nrows & kSymvFull) != 0;
    t.nrows &= ~kSymvFlags;
    const uint64_t c0 = (uint64_t)t.strip * SS, c = c0 + (uint64_t)tid * VEC;       // this lane's columns: c + v * CW + i
    const uint64_t grow0 = row_off + t.row0;
    bool live[NV];                                                                   // columns behind the row's end: nothing to do
    tvec_t pc[NV];
    T cacc[NV][VEC];
#pragma unroll
    for (int v = 0; v < NV; v++) {
        live[v] = c + (uint64_t)v * CW < ncols_vec;
#pragma unroll
        for (int i = 0; i < VEC; i++) { pc[v][i] = (T)0; cacc[v][i] = (T)0; }
        if (live[v]) pc[v] = *reinterpret_cast<const tvec_t *>(p + c + (uint64_t)v * CW);
    }
    // Interior tasks -- nearly all of them -- take the lean loop: unconditional loads from UNIFORM row bases plus the lane's
    // constant offset (no exec-mask juggling between the loads: the predicated form below issued its loads one branch at a time
    // and spent half of its wave cycles on issue stalls), the rows' values of p by scalar loads; software-pipelined in two halves
    // of 4 rows -- the loads of the next step's half are issued as soon as this step's half has been consumed, so a wave always
    // has 4 * NV ... 8 * NV loads in flight (a loop that drains its loads, computes its ~125 vector instructions and only then
    // issues the next ones measured 1-4 % slower, profiles/r04_symv2_probe.txt).
    // the row partials of the last (up to) 256 rows, ending at row `end` of the task: out of LDS into rowpart (uniform call sites)
    auto flush_rows = [&](uint32_t end) {
        const uint32_t beg = (end - 1) & ~(uint32_t)(kSymvRowsLds - 1);
        __syncthreads();
        for (uint32_t r = beg + tid; r < end && r < t.nrows; r += kBlock) {
            const uint32_t q = r - beg;
#ifdef LAM_SYMV_PROBE_NO_PARTIAL_STORES      /* tools/symv2_probe only: what the partial stores cost (results are wrong without them) */
            if (s_rows[0][q] == (T)123.456)
#endif
            __builtin_nontemporal_store((s_rows[0][q] + s_rows[1][q]) + (s_rows[2][q] + s_rows[3][q]), rowpart + (uint64_t)t.rp + r);
        }
        __syncthreads();
    };
    auto lean_loop = [&](auto masked_tag) {
        constexpr bool MASKED = decltype(masked_tag)::value;
        const TA *rows = A + (uint64_t)t.row0 * lda + c0;          // uniform
        const T *prow = p + grow0;
        vec_t a0[4][NV], a1[4][NV];
#pragma unroll
        for (int k = 0; k < 4; k++)
#pragma unroll
            for (int v = 0; v < NV; v++) {
                a0[k][v] = __builtin_nontemporal_load(reinterpret_cast<const vec_t *>(rows + (uint64_t)k * lda + v * CW) + tid);
                a1[k][v] = __builtin_nontemporal_load(reinterpret_cast<const vec_t *>(rows + (uint64_t)(k + 4) * lda + v * CW) + tid);
            }
        for (uint32_t b = 0; b < t.nrows; b += 8, prow += 8) {
            rows += 8 * lda;
            const bool more = b + 8 < t.nrows;
            T racc[8];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const T pr = prow[k];
                T r = (T)0;
#pragma unroll
                for (int v = 0; v < NV; v++)
#pragma unroll
                    for (int i = 0; i < VEC; i++) {
                        bool rs = true, cs = true;
                        if (MASKED) symv_use<CYC>(c + (uint64_t)v * CW + i, grow0 + b + k, n, &rs, &cs);
                        if (rs) r = fma_tv((T)MV::get(a0[k][v], i), (T)pc[v][i], r);
                        if (cs) cacc[v][i] = fma_tv((T)MV::get(a0[k][v], i), pr, cacc[v][i]);
                    }
                racc[k] = r;
            }
            if (more) {
#pragma unroll
                for (int k = 0; k < 4; k++)
#pragma unroll
                    for (int v = 0; v < NV; v++)
                        a0[k][v] = __builtin_nontemporal_load(reinterpret_cast<const vec_t *>(rows + (uint64_t)k * lda + v * CW) + tid);
            }
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const T pr = prow[4 + k];
                T r = (T)0;
#pragma unroll
                for (int v = 0; v < NV; v++)
#pragma unroll
                    for (int i = 0; i < VEC; i++) {
                        bool rs = true, cs = true;
                        if (MASKED) symv_use<CYC>(c + (uint64_t)v * CW + i, grow0 + b + 4 + k, n, &rs, &cs);
                        if (rs) r = fma_tv((T)MV::get(a1[k][v], i), (T)pc[v][i], r);
                        if (cs) cacc[v][i] = fma_tv((T)MV::get(a1[k][v], i), pr, cacc[v][i]);
                    }
                racc[4 + k] = r;
            }
            if (more) {
#pragma unroll
                for (int k = 0; k < 4; k++)
#pragma unroll
                    for (int v = 0; v < NV; v++)
                        a1[k][v] = __builtin_nontemporal_load(reinterpret_cast<const vec_t *>(rows + (uint64_t)(k + 4) * lda + v * CW) + tid);
            }
            const T tot = wave_sum8(racc);
            if ((lane & 7) == 0) s_rows[wave][(b & (kSymvRowsLds - 1)) + (lane >> 3)] = tot;
            if (((b + 8) & (kSymvRowsLds - 1)) == 0 || b + 8 >= t.nrows) flush_rows(b + 8);
        }
    };
    if (interior) lean_loop(std::false_type());
    else if (full) lean_loop(std::true_type());      // the rim of the rows' windows in full strips: the same loads, every product tested
    else {
        // ragged strips and row runs: every load and every product tested
        const TA *Arow = A + (uint64_t)t.row0 * lda + c;
        for (uint32_t b = 0; b < t.nrows; b += 8) {
            vec_t a[8][NV];
            T pr[8];
#pragma unroll
            for (int k = 0; k < 8; k++) {
                const uint64_t grow = grow0 + b + k;
#pragma unroll
                for (int v = 0; v < NV; v++) {
#pragma unroll
                    for (int i = 0; i < VEC; i++) a[k][v][i] = 0;
                    // one shard: a vector whose columns all lie left of the diagonal is not read at all
                    if (live[v] && b + k < t.nrows && (CYC || c + (uint64_t)v * CW + VEC > grow))
                        a[k][v] = __builtin_nontemporal_load(reinterpret_cast<const vec_t *>(Arow + (uint64_t)(b + k) * lda + (uint64_t)v * CW));
                }
                pr[k] = b + k < t.nrows ? p[grow] : (T)0;
            }
            T racc[8];
#pragma unroll
            for (int k = 0; k < 8; k++) {
                const uint64_t grow = grow0 + b + k;
                T r = (T)0;
#pragma unroll
                for (int v = 0; v < NV; v++)
#pragma unroll
                    for (int i = 0; i < VEC; i++) {
                        bool rs, cs;
                        symv_use<CYC>(c + (uint64_t)v * CW + i, grow, n, &rs, &cs);
                        if (rs) r = fma_tv((T)MV::get(a[k][v], i), (T)pc[v][i], r);
                        if (cs) cacc[v][i] = fma_tv((T)MV::get(a[k][v], i), pr[k], cacc[v][i]);
                    }
                racc[k] = r;
            }
            const T tot = wave_sum8(racc);
            if ((lane & 7) == 0) s_rows[wave][(b & (kSymvRowsLds - 1)) + (lane >> 3)] = tot;
            if (((b + 8) & (kSymvRowsLds - 1)) == 0 || b + 8 >= t.nrows) flush_rows(b + 8);
        }
    }
#pragma unroll
    for (int v = 0; v < NV; v++)
        if (live[v]) {
            tvec_t out;
#pragma unroll
            for (int i = 0; i < VEC; i++) out[i] = cacc[v][i];
#ifdef LAM_SYMV_PROBE_NO_PARTIAL_STORES      /* tools/symv2_probe only: what the partial stores cost (results are wrong without them) */
            if (out[0] == (T)123.456)
#endif
            __builtin_nontemporal_store(out, reinterpret_cast<tvec_t *>(colpart + (uint64_t)blockIdx.x * SS + (uint64_t)v * CW + (uint64_t)tid * VEC));
        }
}

// Second pass.  32 entries of y per workgroup; the terms of an entry are dealt round-robin to 8 groups of 32 lanes and combined
// in a fixed order.  Both kinds of partial are indexed BY TASK, in dispatch order -- rowpart[task][row of the run],
// colpart[task][column of the strip] --, so that the tasks in flight together write next to each other: with the partials
// laid out by strip (a 4-KiB column partial every ~1 MiB, a 2-KiB row partial every 512 KiB) the first pass ran up to 10 %
// slower depending on where the buffers happened to be placed (2.57 ... 2.85 ms for the same launch at N=65536; with every task
// writing ONE slot 2.52 ... 2.61: profiles/r04_symv2_probe.txt) -- 1 % of the traffic, but scattered over as many pages as
// tasks.  Row side: the tasks of the row's run (consecutive) for the rows this shard owns, [row_off, row_off + nloc); column
// side: the tasks listed for the column's strip.  One shard: y is the product, `partial` its p.y per workgroup.  Several shards
// (dst.n > 0): the entry is this shard's CONTRIBUTION to y[i], stored into its record in every shard's gather buffer; the
// consumer adds the shards' records in shard order; the launch carries one extra workgroup (fin.active) that sums the
// workgroups' parts of p.Ap in the fixed order of block_sum_array and writes the total behind the record (see Finalize).
template <typename T, int SS>         // T: vectors and partials; SS: columns of a strip
__global__ void __launch_bounds__(kBlock)
symv_reduce_kernel(const T *__restrict__ rowpart, const T *__restrict__ colpart, const uint32_t *__restrict__ index, SymvIndex ix,
                   const T *__restrict__ p, T *__restrict__ y, double *__restrict__ partial, uint64_t n, uint64_t row_off,
                   uint64_t nloc, PtrList dst, Finalize fin, const CgScalars *sc)
{
    constexpr int RB = kSymvReduceRows, G = kBlock / RB;
    __shared__ T s[G][RB];
    __shared__ double s_dot[RB];
    __shared__ uint32_t s_list[kSymvListChunk];
    if (sc != nullptr && sc->stop) return;
    if (is_reducer_block(fin)) {                        // several shards: the shard's part of p.Ap as ONE number, inside this launch
        reduce_partials(partial, (int)compute_blocks(fin), fin, s_dot);
        return;
    }
    const int l = threadIdx.x % RB, g = threadIdx.x / RB;
    const uint64_t i = (uint64_t)blockIdx.x * RB + l;
    T acc = (T)0;
    if (i < n && i >= row_off && i < row_off + nloc) {
        const uint64_t il = i - row_off;
        const uint32_t *run = index + ix.runs + 5 * index[ix.row8 + (il >> 3)];
        const uint32_t cnt = run[1], h = run[3];
        const T *src = rowpart + (uint64_t)run[4] + (il - run[2]);
        uint32_t q = g;
        for (; q + 3 * G < cnt; q += 4 * G) {
            const T a0 = src[(uint64_t)q * h], a1 = src[(uint64_t)(q + G) * h];
            const T a2 = src[(uint64_t)(q + 2 * G) * h], a3 = src[(uint64_t)(q + 3 * G) * h];
            acc += (a0 + a1) + (a2 + a3);
        }
        for (; q < cnt; q += G) acc += src[(uint64_t)q * h];
    }
    // column side: the workgroup's 32 columns lie in ONE strip (32 divides the strip width), so its task list is staged in LDS
    // once per 1024 entries -- a load of the list in front of every load of a partial doubled the chain of memory latencies that
    // this pass is made of at small N
    {
        const uint32_t s0 = (uint32_t)(((uint64_t)blockIdx.x * RB) / SS);
        const uint32_t lb = index[ix.strip_base + s0], le = index[ix.strip_base + s0 + 1];
        const uint32_t *list = index + ix.strip_tasks;
        const T *src = colpart + (i - (uint64_t)s0 * SS);
        for (uint32_t base = lb; base < le; base += kSymvListChunk) {
            const uint32_t cnt = le - base < (uint32_t)kSymvListChunk ? le - base : (uint32_t)kSymvListChunk;
            __syncthreads();                                  // the previous chunk has been consumed
            for (uint32_t q = threadIdx.x; q < cnt; q += kBlock) s_list[q] = list[base + q];
            __syncthreads();
            if (i < n) {
                uint32_t k = g;
                for (; k + 7 * G < cnt; k += 8 * G) {
                    T a[8];
#pragma unroll
                    for (int u = 0; u < 8; u++) a[u] = src[(uint64_t)s_list[k + u * G] * SS];
                    acc += ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
                }
                for (; k < cnt; k += G) acc += src[(uint64_t)s_list[k] * SS];
            }
        }
    }
    s[g][l] = acc;
    __syncthreads();
    if (g == 0) {
        T t = s[0][l];
#pragma unroll
        for (int k = 1; k < G; k++) t += s[k][l];
        if (i < n) {
            if (dst.n == 0) y[i] = t;
            else for (int j = 0; j < dst.n; j++) reinterpret_cast<T *>(dst.p[j])[i] = t;
        }
        s_dot[l] = i < n ? (double)t * (double)p[i] : 0.0;
    }
    __syncthreads();
    if (threadIdx.x == 0 && partial != nullptr) {
        double d = 0.0;
#pragma unroll
        for (int k = 0; k < RB; k++) d += s_dot[k];
        publish_partial(d, partial, fin);               // plain store, or the self-flagging slot the reducer workgroup waits for
    }
}

// max |A[i][j] - A[j][i]| over the local matrix (single shard) and max |A[i][j]| over the same elements: per-workgroup maxima in
// out[blockIdx.x] and out[gridDim.x + blockIdx.x].  32 x 32 tiles of the upper triangle through LDS, so that both the tile and its
// mirror image are read along rows (one pass over the matrix at a useful fraction of the stream rate: this runs once per matrix when
// a driver asks for the symmetric product through the environment, lam_exchange.h env_symmetric_check).
__device__ __forceinline__ double elem_as_double(double v) { return v; }
__device__ __forceinline__ double elem_as_double(float v) { return (double)v; }
__device__ __forceinline__ double elem_as_double(__hip_bfloat16 v) { return (double)__uint_as_float(((unsigned)*reinterpret_cast<const unsigned short *>(&v)) << 16); }
// Several row shards of ONE process (`shards.n` > 1): row i lives in shard min(i / base_rows, shards.n - 1) (the reference's
// partition), whose matrix the launching device reads through peer access -- one-off, at the speed of the links.
template <typename T>
__global__ void __launch_bounds__(kBlock)
asymmetry_kernel(PtrList shards, uint64_t base_rows, uint64_t lda, uint64_t n, double *__restrict__ out)
{
    auto elem = [&](uint64_t i, uint64_t j) -> double {
        uint64_t q = shards.n > 1 ? i / base_rows : 0;
        if (q >= (uint64_t)shards.n) q = (uint64_t)shards.n - 1;
        return elem_as_double(static_cast<const T *>(shards.p[q])[(i - q * base_rows) * lda + j]);
    };
    constexpr int TS = 32;
    __shared__ double s_up[TS][TS + 1], s_lo[TS][TS + 1];
    __shared__ double s_max[2][kWaves];
    double m = 0.0, a = 0.0;
    const uint64_t nt = (n + TS - 1) / TS, ntiles = nt * (nt + 1) / 2;
    const int tx = threadIdx.x % TS, ty = threadIdx.x / TS;          // 32 x 8 threads: four row passes per tile
    for (uint64_t t = blockIdx.x; t < ntiles; t += gridDim.x) {
        // tile (bi, bj) with bj >= bi, numbered row by row of the upper block triangle
        uint64_t bi = (uint64_t)((2.0 * (double)nt + 1.0 - sqrt((2.0 * (double)nt + 1.0) * (2.0 * (double)nt + 1.0) - 8.0 * (double)t)) * 0.5);
        while (bi > 0 && bi * nt - bi * (bi - 1) / 2 > t) bi--;
        while ((bi + 1) * nt - (bi + 1) * bi / 2 <= t) bi++;
        const uint64_t bj = bi + (t - (bi * nt - bi * (bi - 1) / 2));
        __syncthreads();
        for (int r = ty; r < TS; r += kBlock / TS) {
            const uint64_t iu = bi * TS + r, ju = bj * TS + tx;       // element (iu, ju) of the upper tile
            const uint64_t il = bj * TS + r, jl = bi * TS + tx;       // element (il, jl) of its mirror image
            s_up[r][tx] = (iu < n && ju < n) ? elem(iu, ju) : 0.0;
            s_lo[r][tx] = (il < n && jl < n) ? elem(il, jl) : 0.0;
        }
        __syncthreads();
        for (int r = ty; r < TS; r += kBlock / TS) {
            const double u = s_up[r][tx], l = s_lo[tx][r];             // A[bi*TS + r][bj*TS + tx] and A[bj*TS + tx][bi*TS + r]
            const double d = fabs(u - l), v = fmax(fabs(u), fabs(l));
            m = d > m ? d : m;
            a = v > a ? v : a;
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const double o = __shfl_xor(m, off, 64), q = __shfl_xor(a, off, 64);
        m = o > m ? o : m;
        a = q > a ? q : a;
    }
    if ((threadIdx.x & 63) == 0) { s_max[0][threadIdx.x >> 6] = m; s_max[1][threadIdx.x >> 6] = a; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double t0 = s_max[0][0], t1 = s_max[1][0];
        for (int w = 1; w < kWaves; w++) { t0 = s_max[0][w] > t0 ? s_max[0][w] : t0; t1 = s_max[1][w] > t1 ? s_max[1][w] : t1; }
        out[blockIdx.x] = t0;
        out[gridDim.x + blockIdx.x] = t1;
    }
}

// General path (any n, any alignment): one wave per row, per-row alignment peel + 16-B loads, p from L2.
template <typename TA, typename TV>
__global__ void __launch_bounds__(kBlock)
gemv_generic_kernel(GemvArgs<TA, TV> a)
{
    __shared__ double s_red[kWaves];
    if (a.sc != nullptr && a.sc->stop) return;
    if (is_reducer_block(a.fin)) { reduce_partials(a.partial, (int)gridDim.x - 1, a.fin, s_red); return; }
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const uint64_t row = (uint64_t)blockIdx.x * kWaves + wave;
    double dotp = 0.0;
    if (row < a.nrows) {
        const TA *ar = a.A + row * a.lda;
        TV acc = (TV)0;
        using MV = MatVec<TA>;
        using avec_t = typename MV::vec_t;
        constexpr int VEC = MV::N;
        for (int sg = 0; sg < a.nseg; sg++) {
            const uint64_t beg = a.seg_begin[sg], end = a.seg_end[sg];
            // Rows of an odd-N matrix start at any element offset: peel scalar columns up to the next
            // 16-byte boundary of THIS row, stream the aligned middle with 16-B non-temporal loads
            // (p is read element-wise from L2: its alignment differs from the row's), finish scalar.
            const uint64_t addr = reinterpret_cast<uint64_t>(ar + beg);
            uint64_t peel = ((16 - (addr & 15)) & 15) / sizeof(TA);
            if (peel > end - beg) peel = end - beg;
            if ((uint64_t)lane < peel) acc += widen<TV>(ar[beg + lane]) * a.p[beg + lane];
            const uint64_t mid = beg + peel;
            const uint64_t nvec = (end - mid) / VEC;
            uint64_t v = lane;
#pragma unroll 1
            for (; v + 3 * 64 < nvec; v += 4 * 64) {
                avec_t av[4];
#pragma unroll
                for (int u = 0; u < 4; u++)
                    av[u] = __builtin_nontemporal_load(reinterpret_cast<const avec_t *>(ar + mid) + v + u * 64);
#pragma unroll
                for (int u = 0; u < 4; u++)
#pragma unroll
                    for (int i = 0; i < VEC; i++) acc += (TV)MV::get(av[u], i) * a.p[mid + (v + u * 64) * VEC + i];
            }
            for (; v < nvec; v += 64) {
                const avec_t av = *(reinterpret_cast<const avec_t *>(ar + mid) + v);
#pragma unroll
                for (int i = 0; i < VEC; i++) acc += (TV)MV::get(av, i) * a.p[mid + v * VEC + i];
            }
            for (uint64_t c = mid + nvec * VEC + lane; c < end; c += 64) acc += widen<TV>(ar[c]) * a.p[c];
        }
        acc = wave_sum(acc);
        if (lane == 0) {
            if (a.accumulate) acc += a.y[row];
            store_y(a, row, acc);
            dotp = (double)acc * (double)a.p[a.row0 + row];
        }
    }
    if (a.partial != nullptr) {
        if (lane == 0) s_red[wave] = dotp;
        __syncthreads();
        double t = 0.0;
        if (threadIdx.x == 0) {
            t = s_red[0];
#pragma unroll
            for (int w = 1; w < kWaves; w++) t += s_red[w];
        }
        publish_partial(t, a.partial, a.fin);
    }
}

// ---------------------------------------------------------------------------------------------
// vector kernels
// ---------------------------------------------------------------------------------------------
// sum src[0..n) with one workgroup and store it at index `slot` of every destination array
// (cg_init only: inside the iteration the producer kernels do this themselves, see Finalize)
__global__ void __launch_bounds__(kBlock)
finalize_sum_kernel(const double *__restrict__ src, int n, PtrList dst, int slot, const CgScalars *sc)
{
    __shared__ double s_red[kWaves];
    if (sc != nullptr && sc->stop) return;
    double t = block_sum_array(src, n, s_red);
    if (threadIdx.x == 0)
        for (int j = 0; j < dst.n; j++) reinterpret_cast<double *>(dst.p[j])[slot] = t;
}

// x = 0, r = b, p_slice = b (into every shard's p), partials of b.b
template <typename TV>
__global__ void __launch_bounds__(kBlock)
cg_init_kernel(const TV *__restrict__ b, TV *__restrict__ x, TV *__restrict__ r, PtrList pdst,
               uint64_t row0, uint64_t n_loc, double *__restrict__ partial)
{
    __shared__ double s_red[kWaves];
    double acc = 0.0;
    for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < n_loc; i += (uint64_t)gridDim.x * kBlock) {
        const TV bi = b[i];
        x[i] = (TV)0;
        r[i] = bi;
        for (int j = 0; j < pdst.n; j++) reinterpret_cast<TV *>(pdst.p[j])[row0 + i] = bi;
        acc += (double)bi * (double)bi;
    }
    double t = block_sum(acc, s_red);
    if (threadIdx.x == 0) partial[blockIdx.x] = t;
}

// bb = rr[0] = sum(red); iters = 0; stop = 0
__global__ void __launch_bounds__(kBlock)
cg_init_scalars_kernel(const double *__restrict__ red, int nred, CgScalars *sc)
{
    __shared__ double s_red[kWaves];
    double t = block_sum_array(red, nred, s_red);
    if (threadIdx.x == 0) {
        sc->bb = t;
        sc->rr[0] = t;
        sc->rr[1] = 0.0;
        sc->pAp = 0.0;
        sc->alpha = 0.0;
        sc->beta = 0.0;
        sc->iters = 0;
        sc->stop = 0;
    }
}

// alpha = rr / p.Ap ; x += alpha p ; r -= alpha Ap ; partials of r.r
// (axpby(alpha,p,1,x); axpby(-alpha,Ap,1,r); dot(r,r): ConjugateGradient_CPU_MPI_OMP.hpp:107-110)
template <typename TV>
__global__ void __launch_bounds__(kBlock)
update_xr_kernel(const double *__restrict__ red, int nred, CgScalars *sc, int k,
                 const TV *__restrict__ p_loc, const TV *__restrict__ Ap, TV *__restrict__ x,
                 TV *__restrict__ r, uint64_t n_loc, double *__restrict__ partial, Finalize fin, MailWait mw)
{
    __shared__ double s_red[kWaves];
    if (sc->stop) return;
    if (is_reducer_block(fin)) { reduce_partials(partial, (int)gridDim.x - 1, fin, s_red); return; }
    const double pAp = mw.n > 0 ? mail_sum(mw, s_red) : block_sum_array(red, nred, s_red);
    const double rr = sc->rr[(k + 1) & 1];
    const double alpha_d = rr / pAp;
    const TV alpha = (TV)alpha_d;
    double acc = 0.0;
    const uint64_t stride = (uint64_t)compute_blocks(fin) * kBlock;
    for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < n_loc; i += stride) {
        x[i] = alpha * p_loc[i] + x[i];
        const TV ri = -alpha * Ap[i] + r[i];
        r[i] = ri;
        acc += (double)ri * (double)ri;
    }
    double t = block_sum(acc, s_red);
    if (threadIdx.x == 0 && blockIdx.x == 0) { sc->pAp = pAp; sc->alpha = alpha_d; }
    publish_partial(t, partial, fin);
}

// rr' = r.r ; beta = rr'/rr ; if sqrt(rr'/bb) < tol: stop (p untouched) else p_slice = r + beta p
// (ConjugateGradient_CPU_MPI_OMP.hpp:110-114: the test comes BEFORE the p update)
template <typename TV>
__global__ void __launch_bounds__(kBlock)
update_p_kernel(const double *__restrict__ red, int nred, CgScalars *sc, int k, double rel_error,
                const TV *__restrict__ r, const TV *__restrict__ p_loc, PtrList pdst, uint64_t row0,
                uint64_t n_loc, volatile int *host_flags /* pinned host progress word, see post_progress */,
                MailWait mw, MailPost post)
{
    __shared__ double s_red[kWaves];
    if (sc->stop) return;
    const double rr_new = mw.n > 0 ? mail_sum(mw, s_red) : block_sum_array(red, nred, s_red);
    const double rr = sc->rr[(k + 1) & 1];
    const double bb = sc->bb;
    const double beta_d = rr_new / rr;
    const bool stop = sqrt(rr_new / bb) < rel_error;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        sc->rr[k & 1] = rr_new;
        sc->beta = beta_d;
        sc->iters = k;
        post_progress(host_flags, k, stop);   // WHICH iteration stopped: the host compares it with the iteration whose
                                              // completion it has awaited (lam_hip_cg_iterate)
    }
    if (stop) {
        // every workgroup reaches the same decision from the same bits; a workgroup that starts
        // after the flag is up returns at the top, which is the same outcome (p is not updated)
        if (blockIdx.x == 0 && threadIdx.x == 0) sc->stop = 1;
        return;
    }
    const TV beta = (TV)beta_d;
    if (post.n == 0) {
        for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < n_loc; i += (uint64_t)gridDim.x * kBlock) {
            const TV pi = r[i] + beta * p_loc[i];
            for (int j = 0; j < pdst.n; j++) reinterpret_cast<TV *>(pdst.p[j])[row0 + i] = pi;
        }
        return;
    }
    // direct exchange: the slice goes into every rank's replica with system-scope (write-through) stores;
    // once this workgroup's stores have drained it raises its flag in every mailbox
    for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < n_loc; i += (uint64_t)gridDim.x * kBlock) {
        const TV pi = r[i] + beta * p_loc[i];
        for (int j = 0; j < pdst.n; j++)
            __hip_atomic_store(reinterpret_cast<TV *>(pdst.p[j]) + row0 + i, pi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    // every thread: its slice stores happen-before the workgroup's flag stores (system-scope release fence, then the
    // workgroup barrier, then relaxed flag stores; the reader pairs it with an acquire fence behind its poll)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if ((int)threadIdx.x < post.n && (int)threadIdx.x != post.rank)
        st_sys(&post.mail[threadIdx.x]->pflag[post.rank][blockIdx.x], post.seq);
}

// direct exchange: wait until every other rank's update_p workgroups have flagged their p slice for `seq`
__device__ __forceinline__ void wait_p_flags(const Mail *mine, int nranks, int rank, const BlockCounts &nb, unsigned long long seq, int *host_err)
{
    SpinGuard guard;
    for (int q = 0; q < nranks; q++) {
        if (q == rank) continue;
        for (int b = threadIdx.x; b < nb.n[q]; b += kBlock) {
            unsigned long long seen;
            while ((seen = ld_sys(&mine->pflag[q][b])) != seq) {
                if (guard.slow_path()) {
                if (*(volatile int *)host_err != 0) return;
                if (guard.expired()) {
                    host_err[1] = q * 1000 + b; host_err[2] = (int)(unsigned)seq; host_err[3] = (int)(unsigned)seen;
                    host_err[4] = (int)(seq >> 32); host_err[5] = (int)(seen >> 32);
                    *(volatile int *)host_err = 3;
                    return;
                }
                }
                __builtin_amdgcn_s_sleep(4);
            }
        }
    }
    // the peers' slices (stored before their release fences) happen-before everything this thread -- and, through the
    // end of the kernel, the next kernel on this stream -- does from here on
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
}

// ---------------------------------------------------------------------------------------------
// Fused vector step: update_xr_kernel + update_p_kernel in ONE launch (one shard, and the direct exchange).
// What separates the two kernels is a grid-wide dependency -- r.r needs every workgroup's partial --, and that
// is exactly what the reducer workgroup + mailbox machinery already provides inside a launch: the compute
// workgroups publish their partials of r.r, the reducer workgroup turns them into the total -- with the direct
// exchange by way of every rank's mailbox, which it alone polls -- and hands {value, tag} to the launch's other
// workgroups through a broadcast slot in ordinary device memory, which they poll (bounded) at agent scope.  All workgroups of the launch are resident
// together (at most 256 + 2 of them), so nobody waits for a workgroup that cannot start.  Arithmetic, element
// -> thread mapping and reduction order are those of the two kernels: results are bit-identical.
// Roles by workgroup index: 0 reducer, [1, ncompute] compute, ncompute + 1 (direct exchange without
// the own-slice GEMV panel only) the WAITER that holds the launch open until the peers' p slices for the next
// GEMV have arrived -- which makes wait_p_kernel unnecessary: 2 launches per iteration.
// ---------------------------------------------------------------------------------------------
template <typename TV>
__global__ void __launch_bounds__(kBlock)
update_fused_kernel(const double *__restrict__ red, int nred, CgScalars *sc, int k, double rel_error,
                    const TV *p_loc, const TV *__restrict__ Ap, TV *__restrict__ x, TV *__restrict__ r, uint64_t n_loc,
                    double *partial, int ncompute, Finalize fin, MailWait mw_pap, MailWait mw_rr, BcastLine *bc /*[2][kBcastLines], local*/,
                    PtrList pdst, uint64_t row0, volatile int *host_flags, MailPost post, const Mail *mine, BlockCounts nb)
{
    __shared__ double s_red[kWaves];
    if (sc->stop) return;
    const unsigned long long seq = mw_rr.seq;
    int *host_err = mw_rr.host_err;
    // workgroup 0 is the reducer (dispatched first: everybody else ends up waiting for it), 1..ncompute compute,
    // ncompute + 1 the waiter
    const int cb = (int)blockIdx.x - 1;
    if (blockIdx.x == 0) {
        // reducer: the ONLY workgroup that polls the (uncached) mailbox; everything it learns goes to the launch's
        // other workgroups through the two local broadcast slots
        const int nl = (int)gridDim.x - 1;                                             // listeners: compute workgroups (+ waiter)
        if (mw_pap.n > 0) bcast_post(bc, nl, mail_sum(mw_pap, s_red), seq);            // direct exchange: p.Ap of all ranks
        const double local = reduce_partials_sum(partial, ncompute, s_red, host_err);  // this shard's r.r
        if (fin.mail) {
            post_total(fin, local);                                                    // ... to every rank's mailbox
            bcast_post(bc + kBcastLines, nl, mail_sum(mw_rr, s_red), seq);             // r.r of all ranks
        } else {
            bcast_post(bc + kBcastLines, nl, local, seq);                              // one shard
        }
        return;
    }
    const double bb = sc->bb;
    if (cb >= ncompute) {                               // waiter
        const double rr_w = bcast_wait(bc + kBcastLines + cb, seq, host_err, s_red);
        if (sqrt(rr_w / bb) < rel_error) return;        // the solve stops here: nobody posts a p slice
        wait_p_flags(mine, post.n, post.rank, nb, post.seq, host_err);
        return;
    }
    // ---- update_xr_kernel
    const double pAp = mw_pap.n > 0 ? bcast_wait(bc + cb, seq, host_err, s_red) : block_sum_array(red, nred, s_red);
    const double rr = sc->rr[(k + 1) & 1];
    const double alpha_d = rr / pAp;
    const TV alpha = (TV)alpha_d;
    double acc = 0.0;
    const uint64_t stride = (uint64_t)ncompute * kBlock;
    for (uint64_t i = (uint64_t)cb * kBlock + threadIdx.x; i < n_loc; i += stride) {
        x[i] = alpha * p_loc[i] + x[i];
        const TV ri = -alpha * Ap[i] + r[i];
        r[i] = ri;
        acc += (double)ri * (double)ri;
    }
    const double t = block_sum(acc, s_red);
    if (threadIdx.x == 0 && cb == 0) { sc->pAp = pAp; sc->alpha = alpha_d; }
    if (threadIdx.x == 0) __hip_atomic_store(partial + cb, t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // ---- update_p_kernel
    const double rr_new = bcast_wait(bc + kBcastLines + cb, seq, host_err, s_red);
    const double beta_d = rr_new / rr;
    const bool stop = sqrt(rr_new / bb) < rel_error;
    if (cb == 0 && threadIdx.x == 0) {
        sc->rr[k & 1] = rr_new;
        sc->beta = beta_d;
        sc->iters = k;
        post_progress(host_flags, k, stop);
        if (stop) sc->stop = 1;
    }
    if (stop) return;
    const TV beta = (TV)beta_d;
    if (post.n == 0) {
        for (uint64_t i = (uint64_t)cb * kBlock + threadIdx.x; i < n_loc; i += stride) {
            const TV pi = r[i] + beta * p_loc[i];
            for (int j = 0; j < pdst.n; j++) reinterpret_cast<TV *>(pdst.p[j])[row0 + i] = pi;
        }
        return;
    }
    for (uint64_t i = (uint64_t)cb * kBlock + threadIdx.x; i < n_loc; i += stride) {
        const TV pi = r[i] + beta * p_loc[i];
        for (int j = 0; j < pdst.n; j++)
            __hip_atomic_store(reinterpret_cast<TV *>(pdst.p[j]) + row0 + i, pi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");        // as in update_p_kernel
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if ((int)threadIdx.x < post.n && (int)threadIdx.x != post.rank)
        st_sys(&post.mail[threadIdx.x]->pflag[post.rank][cb], post.seq);
}

__global__ void __launch_bounds__(kBlock)
wait_p_kernel(const Mail *mine, int nranks, int rank, BlockCounts nb, unsigned long long seq, const CgScalars *sc, int *host_err)
{
    if (sc->stop) return;
    wait_p_flags(mine, nranks, rank, nb, seq, host_err);
}

// ---------------------------------------------------------------------------------------------
// "gather-Ap" exchange (rank mode, option exchange = 1): ONE collective per iteration.
// After the GEMV every rank all-gathers [Ap_slice | its partial of p.Ap] (the reference CPU path also
// gathers Ap, ConjugateGradient_CPU_MPI_OMP.hpp:505) and then updates FULL-length r and p
// redundantly -- O(N) work per rank, like the reference's full-length axpby (:476) -- so r.r needs
// no collective at all.  Layout of the gathered buffer: rank q's record starts at q*stride_bytes: room for the LONGEST
// slice (base + n % P values of TV; rank q < P-1 fills the first `base`), then -- at stride_bytes - 8 -- one double.
// ---------------------------------------------------------------------------------------------
template <typename TV>
__global__ void __launch_bounds__(kBlock)
cg_init_full_kernel(TV *__restrict__ r_full /* holds b on entry */, TV *__restrict__ p_full, TV *__restrict__ x,
                    uint64_t n, uint64_t n_loc, double *__restrict__ partial)
{
    __shared__ double s_red[kWaves];
    double acc = 0.0;
    for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (uint64_t)gridDim.x * kBlock) {
        const TV bi = r_full[i];
        p_full[i] = bi;
        if (i < n_loc) x[i] = (TV)0;
        acc += (double)bi * (double)bi;
    }
    double t = block_sum(acc, s_red);
    if (threadIdx.x == 0) partial[blockIdx.x] = t;
}

// write this rank's p.Ap partial (sum of the GEMV workgroup partials) behind its Ap slice
__global__ void __launch_bounds__(kBlock)
finalize_tail_kernel(const double *__restrict__ src, int n, char *record, uint64_t tail_offset_bytes, const CgScalars *sc)
{
    __shared__ double s_red[kWaves];
    if (sc != nullptr && sc->stop) return;
    double t = block_sum_array(src, n, s_red);
    if (threadIdx.x == 0) *reinterpret_cast<double *>(record + tail_offset_bytes) = t;
}

// One element of the gathered Ap.  Gather-Ap exchange: shard i / base holds it in its record.  Symmetric product on several
// shards (sum_records): every shard's record is a full-length CONTRIBUTION to A p; they are added in shard order (the same
// order on every shard, so all shards keep identical r and p).
template <typename TV>
__device__ __forceinline__ TV gathered_ap(const char *__restrict__ gathered, uint64_t stride_bytes, uint64_t base, int nranks,
                                          int sum_records, uint64_t i)
{
    if (!sum_records) {
        // the reference's partition: `base` rows per shard, the remainder on the LAST one (its record is the long one)
        uint64_t q = i / base;
        if (q >= (uint64_t)nranks) q = (uint64_t)nranks - 1;
        return reinterpret_cast<const TV *>(gathered + q * stride_bytes)[i - q * base];
    }
    TV v = reinterpret_cast<const TV *>(gathered)[i];
    for (int q = 1; q < nranks; q++) v += reinterpret_cast<const TV *>(gathered + (uint64_t)q * stride_bytes)[i];
    return v;
}

template <typename TV>
__global__ void __launch_bounds__(kBlock)
update_xr_full_kernel(const char *__restrict__ gathered, uint64_t stride_bytes, uint64_t base, int nranks,
                      CgScalars *sc, int k, const TV *__restrict__ p_full, TV *__restrict__ x, TV *__restrict__ r_full,
                      uint64_t n, uint64_t row0, uint64_t n_loc, double *__restrict__ partial, int sum_records)
{
    __shared__ double s_red[kWaves];
    if (sc->stop) return;
    double pAp = 0.0;                                   // rank order, same on every rank
    for (int q = 0; q < nranks; q++)
        pAp += *reinterpret_cast<const double *>(gathered + (uint64_t)q * stride_bytes + stride_bytes - 8);
    const double rr = sc->rr[(k + 1) & 1];
    const double alpha_d = rr / pAp;
    const TV alpha = (TV)alpha_d;
    double acc = 0.0;
    for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (uint64_t)gridDim.x * kBlock) {
        const TV api = gathered_ap<TV>(gathered, stride_bytes, base, nranks, sum_records, i);
        const TV ri = -alpha * api + r_full[i];
        r_full[i] = ri;
        acc += (double)ri * (double)ri;
        if (i >= row0 && i < row0 + n_loc) x[i - row0] = alpha * p_full[i] + x[i - row0];
    }
    double t = block_sum(acc, s_red);
    if (threadIdx.x == 0) {
        partial[blockIdx.x] = t;
        if (blockIdx.x == 0) { sc->pAp = pAp; sc->alpha = alpha_d; }
    }
}

template <typename TV>
__global__ void __launch_bounds__(kBlock)
update_p_full_kernel(const double *__restrict__ red, int nred, CgScalars *sc, int k, double rel_error,
                     const TV *__restrict__ r_full, TV *__restrict__ p_full, uint64_t n, volatile int *host_flags)
{
    __shared__ double s_red[kWaves];
    if (sc->stop) return;
    const double rr_new = block_sum_array(red, nred, s_red);
    const double rr = sc->rr[(k + 1) & 1];
    const double bb = sc->bb;
    const double beta_d = rr_new / rr;
    const bool stop = sqrt(rr_new / bb) < rel_error;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        sc->rr[k & 1] = rr_new;
        sc->beta = beta_d;
        sc->iters = k;
        post_progress(host_flags, k, stop);   // WHICH iteration stopped: the host compares it with the iteration whose
                                              // completion it has awaited (lam_hip_cg_iterate)
    }
    if (stop) {
        if (blockIdx.x == 0 && threadIdx.x == 0) sc->stop = 1;
        return;
    }
    const TV beta = (TV)beta_d;
    for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (uint64_t)gridDim.x * kBlock)
        p_full[i] = r_full[i] + beta * p_full[i];
}

// update_xr_full_kernel + update_p_full_kernel in ONE launch (round 4): what separates them is the grid-wide r.r, which the
// reducer workgroup resolves inside the launch exactly as in update_fused_kernel -- compute workgroups publish their partials in
// self-flagging slots, workgroup 0 sums them in the order of block_sum_array and hands {value, tag} to every compute workgroup on
// a line of its own.  Same element -> thread mapping, same partial grouping and reduction order as the two kernels: bit-identical
// (the number of compute workgroups equals the two-kernel grid).  The gather-Ap iteration is then GEMV + one collective or join +
// this launch: two launches for every shard count, like the single-shard iteration.  All workgroups must be resident at once.
template <typename TV>
__global__ void __launch_bounds__(kBlock)
update_full_fused_kernel(const char *__restrict__ gathered, uint64_t stride_bytes, uint64_t base, int nranks, CgScalars *sc, int k,
                         double rel_error, TV *p_full, TV *__restrict__ x, TV *r_full, uint64_t n, uint64_t row0, uint64_t n_loc,
                         double *partial, int ncompute, BcastLine *bc, unsigned long long seq, int *host_err, volatile int *host_flags,
                         int sum_records)
{
    __shared__ double s_red[kWaves];
    if (sc->stop) return;
    if (blockIdx.x == 0) {                              // reducer: r.r of the full vector, to every compute workgroup
        const double total = reduce_partials_sum(partial, ncompute, s_red, host_err);
        bcast_post(bc, ncompute, total, seq);
        return;
    }
    const int cb = (int)blockIdx.x - 1;
    double pAp = 0.0;                                   // rank order, same on every shard
    for (int q = 0; q < nranks; q++)
        pAp += *reinterpret_cast<const double *>(gathered + (uint64_t)q * stride_bytes + stride_bytes - 8);
    const double rr = sc->rr[(k + 1) & 1];
    const double bb = sc->bb;
    const double alpha_d = rr / pAp;
    const TV alpha = (TV)alpha_d;
    double acc = 0.0;
    const uint64_t stride = (uint64_t)ncompute * kBlock;
    for (uint64_t i = (uint64_t)cb * kBlock + threadIdx.x; i < n; i += stride) {
        const TV api = gathered_ap<TV>(gathered, stride_bytes, base, nranks, sum_records, i);
        const TV ri = -alpha * api + r_full[i];
        r_full[i] = ri;
        acc += (double)ri * (double)ri;
        if (i >= row0 && i < row0 + n_loc) x[i - row0] = alpha * p_full[i] + x[i - row0];
    }
    const double t = block_sum(acc, s_red);
    if (threadIdx.x == 0) {
        __hip_atomic_store(partial + cb, t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (cb == 0) { sc->pAp = pAp; sc->alpha = alpha_d; }
    }
    const double rr_new = bcast_wait(bc + cb, seq, host_err, s_red);
    const double beta_d = rr_new / rr;
    const bool stop = sqrt(rr_new / bb) < rel_error;
    if (cb == 0 && threadIdx.x == 0) {
        sc->rr[k & 1] = rr_new;
        sc->beta = beta_d;
        sc->iters = k;
        post_progress(host_flags, k, stop);
        if (stop) sc->stop = 1;
    }
    if (stop) return;
    const TV beta = (TV)beta_d;
    for (uint64_t i = (uint64_t)cb * kBlock + threadIdx.x; i < n; i += stride)
        p_full[i] = r_full[i] + beta * p_full[i];
}

#ifdef LAM_TUNING_VARIANTS
// ---------------------------------------------------------------------------------------------
// TUNING BUILD ONLY.
// Whole-iteration persistent launch (option "persistent"; one shard, fp64 / fp32, N a multiple of the vector width).
// EXPERIMENT, off by default; DESIGN.md section 6 has the measurement.
//
// One launch runs `k_count` CG iterations.  The grid is W worker workgroups + ONE reducer workgroup, all resident at
// once (the host checks the occupancy before it uses this kernel).  Per iteration:
//   * GEMV.  Worker w owns the row pairs q = w, w + W, w + 2W, ... (W is a multiple of the number of p tiles, so all
//     of them start their rotated tile order at the same tile) and handles them in groups of kPersistGroup: one p tile
//     is staged in LDS ONCE per group and every pair of the group streams its two rows against it with the tile body
//     of gemv_coop_kernel (coop_stream_tile) -- a row's products are added in exactly the order of the two-launch form.
//     Ap goes to memory with agent-scope (write-through) stores; the pair's partial of p.Ap goes to part_gemv[q], the
//     same slot and value the two-launch form writes.
//   * p is never "updated" in a phase of its own: from the second iteration of a launch on, a tile of
//     p_k = r_k + beta_k p_{k-1} is formed WHILE IT IS STAGED, from r and the last explicitly stored p (two 16-byte
//     sc1 loads and one FMA per element instead of one load; a quarter of the staging the two-launch GEMV does, because
//     of the groups).  That removes the third grid-wide dependency of an iteration: two hand-overs remain.
//   * hand-over 1: the reducer workgroup sums the partials in the order of block_sum_array (bit-identical p.Ap) while
//     the GEMV is still running, and broadcasts the total to the first `vec_blocks` workers, each on a line of its own.
//   * vector step on those workers, with the element -> (workgroup, thread) mapping of update_xr_kernel: p_k[i] is
//     formed once more (and stored: it is the "last explicit p" of the next iteration, in the other of two buffers),
//     x += alpha p, r -= alpha Ap, partials of r.r to part_vec[cb].
//   * hand-over 2: the reducer sums them (bit-identical r.r) and broadcasts to ALL workers: beta, the stop test (every
//     workgroup takes the same decision from the same bits), and the next GEMV starts -- no kernel boundary, no launch
//     ramp.  Worker 0 keeps CgScalars and the host's progress word up to date.
//   * on exit the explicit p of the two-launch form is materialised in pbuf[0], so either form can continue the solve.
// Everything handed over inside the launch is written with sc1 stores that the writing wave drains (s_waitcnt vmcnt(0))
// before the workgroup's flag goes out, and read with sc1 loads (MI355X_MICROARCH.md, "Valid forms"); every wait is
// bounded (SpinGuard) and reports through host_err.
// ---------------------------------------------------------------------------------------------
constexpr int kPersistGroup = 4;

template <typename TA, typename TV>
struct PersistArgs {
    const TA *A;
    uint64_t n;
    uint64_t lda;                 // row pitch of A in elements
    TV *pbuf[2];                  // [0] the shard's p (explicit on entry and on exit), [1] a scratch vector
    TV *r, *x, *Ap;
    double *part_gemv;            // [npairs], armed with the sentinel
    double *part_vec;             // [vec_blocks], armed
    CgScalars *sc;
    int k_first, k_count;
    double rel_error;
    volatile int *host_flags;
    int *host_err;
    BcastLine *bc_pap;            // [vec_blocks]
    BcastLine *bc_rr;             // [W]
    int W, vec_blocks;
    uint32_t npairs, ntiles;
    unsigned long long seq_base;  // hand-over number of iteration k = seq_base + k (grows over the life of the context)
    unsigned long long *ticks;    // [0] += constant-rate ticks (100 MHz) spent in GEMV phases, [1] += phases
};

template <typename T> __device__ __forceinline__ T ld_agent(const T *p);
template <> __device__ __forceinline__ double ld_agent<double>(const double *p)
{
    return __longlong_as_double((long long)__hip_atomic_load(reinterpret_cast<const unsigned long long *>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}
template <> __device__ __forceinline__ float ld_agent<float>(const float *p)
{
    return __uint_as_float(__hip_atomic_load(reinterpret_cast<const unsigned *>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}
template <typename T> __device__ __forceinline__ void st_agent(T *p, T v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// 16-byte sc1 load (L1-bypassing, what a hand-over inside a launch needs) as issued / waited-for pair: the compiler does
// not know the load is asynchronous, so every register it fills is tied to the wait below before anything reads it
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ u32x4_t ld16_sc1_issue(const void *p)
{
    u32x4_t v;
    asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v) : "v"(p) : "memory");
    return v;
}
__device__ __forceinline__ void ld16_wait(u32x4_t &a, u32x4_t &b, u32x4_t &c, u32x4_t &d, u32x4_t &e, u32x4_t &f, u32x4_t &g, u32x4_t &h)
{
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h) : : "memory");
}

// bcast_wait for the persistent launch.  There a worker waits for a WHOLE GEMV phase of the slowest worker (tens of
// microseconds to milliseconds), and a thousand workgroups wait at once: the abort word in pinned host memory (a PCIe read)
// and the clock are looked at once every 8192 failed polls (the first version read it every 256 polls and once per iteration
// in every thread: ~650 us per iteration at every N, all of it PCIe reads).  Polls back off from s_sleep 1 to s_sleep 8 after 512 of them.
__device__ __forceinline__ double persist_wait(const BcastLine *line, unsigned long long seq, int *host_err, double *s_red)
{
    const MailSlot *slot = &line->s;
    double v = 0.0;
    if (threadIdx.x == 0) {
        unsigned polls = 0, seen;
        unsigned long long t0 = 0;
        while (!handover_try_load<__HIP_MEMORY_SCOPE_AGENT>(slot, seq, &v, &seen)) {
            if ((++polls & 8191u) == 0) {
                const unsigned long long now = (unsigned long long)clock64();
                if (t0 == 0) t0 = now | 1ull;
                if (*(volatile int *)host_err != 0) break;
                if (now - t0 > kSpinTimeoutCycles) {
                    host_err[1] = -1; host_err[2] = (int)(unsigned)seq; host_err[3] = (int)seen;
                    host_err[4] = (int)(seq >> 32); host_err[5] = 0;
                    *(volatile int *)host_err = 4;
                    break;
                }
            }
            if (polls < 512u) __builtin_amdgcn_s_sleep(1); else __builtin_amdgcn_s_sleep(8);
        }
    }
    return block_sum(v, s_red);
}

template <typename TA, typename TV>
__global__ void __launch_bounds__(kBlock, 4)      // 4 waves per SIMD = 4 workgroups per CU (what the LDS tile allows in fp64): <= 128 VGPRs
cg_persist_kernel(PersistArgs<TA, TV> a)
{
    static_assert(std::is_same<TA, TV>::value, "persistent launch: matrix and vectors of one type");
    using MV = MatVec<TA>;
    constexpr int VEC = MV::N, R = 2, TILE = 4096, WAVES = 4, UNROLL = 4, G = kPersistGroup;
    constexpr int STEP = 64 * VEC;
    constexpr int PV = 16 / sizeof(TV);
    typedef TV pvec_t __attribute__((ext_vector_type(PV)));

    __shared__ __attribute__((aligned(16))) TV s_p[TILE];
    __shared__ TV s_part[G][R][WAVES];
    __shared__ double s_dot[G * R];
    __shared__ double s_red[kWaves];

    if (a.sc->stop) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int w = (int)blockIdx.x;
    const uint64_t n = a.n;
    const double bb = a.sc->bb;

    if (w == a.W) {
        // ---- the reducer workgroup: both hand-overs of every iteration
        unsigned long long t_prev = wall_clock64();
        for (int it = 0; it < a.k_count; it++) {
            const unsigned long long seq = a.seq_base + (unsigned)(a.k_first + it);
            const double pAp = reduce_partials_sum(a.part_gemv, (int)a.npairs, s_red, a.host_err);
            const unsigned long long t_gemv_end = wall_clock64();
            bcast_post(a.bc_pap, a.vec_blocks, pAp, seq);
            const double rr_new = reduce_partials_sum(a.part_vec, a.vec_blocks, s_red, a.host_err);
            bcast_post(a.bc_rr, a.W, rr_new, seq);
            if (tid == 0) { a.ticks[0] += t_gemv_end - t_prev; a.ticks[1] += 1; }
            t_prev = wall_clock64();
            if (sqrt(rr_new / bb) < a.rel_error) return;
            if (!(rr_new == rr_new)) return;              // a bounded wait expired somewhere (NaN total): the launch drains
        }
        return;
    }

    const uint32_t woff = (uint32_t)wave * STEP + (uint32_t)lane * VEC;
    double rr_old = a.sc->rr[(a.k_first + 1) & 1];
    TV beta = (TV)0;
    int cur = 0;                                  // pbuf[cur]: the last explicitly stored p
    for (int it = 0; it < a.k_count; it++) {
        const int k = a.k_first + it;
        const unsigned long long seq = a.seq_base + (unsigned)k;
        const bool direct = it == 0;              // p_{k-1} is explicit in pbuf[cur]; else p_{k-1} = r + beta pbuf[cur]
        const TV *pold = a.pbuf[cur];
        TV *pdst = a.pbuf[cur ^ 1];

        // ---- GEMV over the pairs this worker owns, kPersistGroup at a time
        for (uint32_t q0 = (uint32_t)w; q0 < a.npairs; q0 += (uint32_t)(G * a.W)) {
            TV acc[G][R];
#pragma unroll
            for (int g = 0; g < G; g++)
#pragma unroll
                for (int r = 0; r < R; r++) acc[g][r] = (TV)0;
            const TA *const lane0 = a.A + woff;            // row pointers are rebuilt per (tile, pair): 16 registers less
            uint32_t tt = (uint32_t)w % a.ntiles;          // == q % ntiles for every pair of this worker (W % ntiles == 0)
            for (uint32_t t = 0; t < a.ntiles; t++) {
                const uint64_t c0 = (uint64_t)tt * TILE;
                const uint32_t cols = (uint32_t)((n - c0 < (uint64_t)TILE) ? (n - c0) : (uint64_t)TILE);
                const uint32_t nv = cols / PV;
                __syncthreads();                           // the previous tile is fully consumed
                if (direct) {
                    const pvec_t *src = reinterpret_cast<const pvec_t *>(pold + c0);
                    pvec_t *dst = reinterpret_cast<pvec_t *>(s_p);
                    for (uint32_t i = tid; i < nv; i += kBlock) dst[i] = src[i];
                } else {
                    // p_k = r_k + beta p_{k-1}, formed while it is staged (same expression as update_p_kernel)
                    const char *rsrc = reinterpret_cast<const char *>(a.r + c0), *psrc = reinterpret_cast<const char *>(pold + c0);
                    for (uint32_t base = 0; base < nv; base += 4 * kBlock) {
                        u32x4_t rv[4], pv[4];
#pragma unroll
                        for (int u = 0; u < 4; u++) {
                            uint32_t i = base + u * kBlock + tid;
                            if (i >= nv) i = nv - 1;        // clamped: in bounds, result not stored
                            rv[u] = ld16_sc1_issue(rsrc + (size_t)i * 16);
                            pv[u] = ld16_sc1_issue(psrc + (size_t)i * 16);
                        }
                        ld16_wait(rv[0], rv[1], rv[2], rv[3], pv[0], pv[1], pv[2], pv[3]);
#pragma unroll
                        for (int u = 0; u < 4; u++) {
                            const uint32_t i = base + u * kBlock + tid;
                            if (i < nv) {
                                const pvec_t rr_ = __builtin_bit_cast(pvec_t, rv[u]), pp_ = __builtin_bit_cast(pvec_t, pv[u]);
                                pvec_t o;
#pragma unroll
                                for (int e = 0; e < PV; e++) o[e] = rr_[e] + beta * pp_[e];
                                reinterpret_cast<pvec_t *>(s_p)[i] = o;
                            }
                        }
                    }
                }
                for (uint32_t i = cols + tid; i < (uint32_t)TILE && i < (cols + STEP * WAVES - 1) / (STEP * WAVES) * (STEP * WAVES); i += kBlock)
                    s_p[i] = (TV)0;
                __syncthreads();
#pragma unroll
                for (int g = 0; g < G; g++) {
                    const uint64_t q = (uint64_t)q0 + (uint64_t)g * (uint64_t)a.W;
                    if (q < a.npairs) {
                        const TA *rowp[R];
#pragma unroll
                        for (int r = 0; r < R; r++) rowp[r] = lane0 + (q * R + r) * a.lda;
                        coop_stream_tile<TA, TV, R, TILE, true, UNROLL, WAVES>(rowp, c0, cols, s_p, woff, acc[g]);
                    }
                }
                tt = (tt + 1 == a.ntiles) ? 0 : tt + 1;
            }
            // epilogue of the group: row sums across the waves in the fixed order of gemv_coop_kernel
#pragma unroll
            for (int g = 0; g < G; g++)
#pragma unroll
                for (int r = 0; r < R; r++) {
                    const TV sacc = wave_sum(acc[g][r]);
                    if (lane == 0) s_part[g][r][wave] = sacc;
                }
            __syncthreads();
            if (tid < G * R) {
                const int g = tid / R, r = tid % R;
                const uint64_t q = (uint64_t)q0 + (uint64_t)g * (uint64_t)a.W;
                double d = 0.0;
                if (q < a.npairs) {
                    const uint64_t row = q * R + r;
                    TV sum = s_part[g][r][0];
#pragma unroll
                    for (int wv = 1; wv < WAVES; wv++) sum += s_part[g][r][wv];
                    st_agent(a.Ap + row, sum);
                    const TV pk = direct ? pold[row] : (TV)(ld_agent(a.r + row) + beta * ld_agent(pold + row));
                    d = (double)sum * (double)pk;
                }
                s_dot[tid] = d;
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // the Ap stores have left before the partial says so
            __syncthreads();
            if (tid < G) {
                const uint64_t q = (uint64_t)q0 + (uint64_t)tid * (uint64_t)a.W;
                if (q < a.npairs) {
                    double t = s_dot[tid * R];
#pragma unroll
                    for (int r = 1; r < R; r++) t += s_dot[tid * R + r];
                    __hip_atomic_store(a.part_gemv + q, t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
        }

        // ---- vector step on the first vec_blocks workers (update_xr_kernel's mapping and arithmetic)
        if (w < a.vec_blocks) {
            // what does not depend on p.Ap is fetched BEFORE waiting for it (r, p and x of the thread's first element --
            // its only one up to N = 65536): the loads' latency then hides under the end of the GEMV phase
            const uint64_t stride = (uint64_t)a.vec_blocks * kBlock;
            const uint64_t i_first = (uint64_t)w * kBlock + tid;
            TV r_first = (TV)0, p_first = (TV)0, x_first = (TV)0;
            if (i_first < n) {
                r_first = ld_agent(a.r + i_first);
                p_first = direct ? pold[i_first] : ld_agent(pold + i_first);
                x_first = a.x[i_first];
            }
            const double pAp = persist_wait(a.bc_pap + w, seq, a.host_err, s_red);
            const double alpha_d = rr_old / pAp;
            const TV alpha = (TV)alpha_d;
            double accv = 0.0;
            for (uint64_t i = i_first; i < n; i += stride) {
                const bool first = i == i_first;
                const TV ri0 = first ? r_first : ld_agent(a.r + i);
                TV pi;
                if (direct) pi = first ? p_first : pold[i];
                else {
                    pi = ri0 + beta * (first ? p_first : ld_agent(pold + i));
                    st_agent(pdst + i, pi);
                }
                a.x[i] = alpha * pi + (first ? x_first : a.x[i]);
                const TV ri = -alpha * ld_agent(a.Ap + i) + ri0;
                st_agent(a.r + i, ri);
                accv += (double)ri * (double)ri;
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // r and p stores have left before the partial says so
            const double t = block_sum(accv, s_red);
            if (tid == 0) {
                if (w == 0) { a.sc->pAp = pAp; a.sc->alpha = alpha_d; }
                __hip_atomic_store(a.part_vec + w, t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        if (!direct) cur ^= 1;                         // pdst now holds the explicit p_{k-1}

        // ---- hand-over 2: r.r of this iteration, for everybody
        const double rr_new = persist_wait(a.bc_rr + w, seq, a.host_err, s_red);
        const double beta_d = rr_new / rr_old;
        const bool stop = sqrt(rr_new / bb) < a.rel_error;
        if (w == 0 && tid == 0) {
            a.sc->rr[k & 1] = rr_new;
            a.sc->beta = beta_d;
            a.sc->iters = k;
            post_progress(a.host_flags, k, stop);
            if (stop) a.sc->stop = 1;
        }
        if (stop) return;                              // like the two-launch form: p is not updated by the stopping iteration
        if (!(rr_new == rr_new)) return;               // a bounded wait expired somewhere (NaN total): the launch drains
        beta = (TV)beta_d;
        rr_old = rr_new;
    }
    // ---- leave the explicit p of the two-launch form behind: p_k = r_k + beta_k p_{k-1}
    if (w < a.vec_blocks) {
        const TV *pold = a.pbuf[cur];
        const uint64_t stride = (uint64_t)a.vec_blocks * kBlock;
        for (uint64_t i = (uint64_t)w * kBlock + tid; i < n; i += stride)
            a.pbuf[0][i] = ld_agent(a.r + i) + beta * ld_agent(pold + i);
    }
}
#endif  // LAM_TUNING_VARIANTS (persistent launch)

// standalone BLAS-1 pieces (lam_hip_dot / lam_hip_axpby and the residual check)
template <typename TV>
__global__ void __launch_bounds__(kBlock)
dot_partial_kernel(const TV *__restrict__ x, const TV *__restrict__ y, uint64_t n, double *__restrict__ partial)
{
    __shared__ double s_red[kWaves];
    double acc = 0.0;
    for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (uint64_t)gridDim.x * kBlock)
        acc += (double)x[i] * (double)y[i];
    double t = block_sum(acc, s_red);
    if (threadIdx.x == 0) partial[blockIdx.x] = t;
}

template <typename TV>
__global__ void __launch_bounds__(kBlock)
axpby_kernel(TV alpha, const TV *__restrict__ x, TV beta, TV *__restrict__ y, uint64_t n)
{
    for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (uint64_t)gridDim.x * kBlock)
        y[i] = alpha * x[i] + beta * y[i];
}

// partial of sum (b - y)^2 over a slice (true-residual check)
template <typename TV>
__global__ void __launch_bounds__(kBlock)
resid_partial_kernel(const TV *__restrict__ b, const TV *__restrict__ y, uint64_t n, double *__restrict__ partial)
{
    __shared__ double s_red[kWaves];
    double acc = 0.0;
    for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (uint64_t)gridDim.x * kBlock) {
        const double d = (double)b[i] - (double)y[i];
        acc += d * d;
    }
    double t = block_sum(acc, s_red);
    if (threadIdx.x == 0) partial[blockIdx.x] = t;
}

// ---------------------------------------------------------------------------------------------
// generators (one-off, not on the hot path)
// ---------------------------------------------------------------------------------------------
__host__ __device__ __forceinline__ uint64_t splitmix64(uint64_t z)
{
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
__host__ __device__ __forceinline__ double u01(uint64_t h) { return (double)(h >> 11) * (1.0 / 9007199254740992.0); }

template <typename TA>
__device__ __forceinline__ TA to_storage(double v);
template <> __device__ __forceinline__ double to_storage<double>(double v) { return v; }
template <> __device__ __forceinline__ float to_storage<float>(double v) { return (float)v; }
template <> __device__ __forceinline__ __hip_bfloat16 to_storage<__hip_bfloat16>(double v) { return __float2bfloat16((float)v); }

// rows [row0,row0+nrows) of dense tridiag(1,2,1) by GLOBAL row index
template <typename TA>
__global__ void __launch_bounds__(kBlock)
gen_tridiag_kernel(TA *__restrict__ A, uint64_t lda, uint64_t row0, uint64_t nrows, uint64_t n)
{
    const uint64_t total = nrows * n;
    for (uint64_t idx = (uint64_t)blockIdx.x * kBlock + threadIdx.x; idx < total; idx += (uint64_t)gridDim.x * kBlock) {
        const uint64_t i = idx / n + row0, j = idx % n;
        double v = 0.0;
        if (i == j) v = 2.0;
        else if (i + 1 == j || i == j + 1) v = 1.0;
        A[(i - row0) * lda + j] = to_storage<TA>(v);
    }
}

template <typename TA>
__global__ void __launch_bounds__(kBlock)
gen_random_spd_kernel(TA *__restrict__ A, uint64_t lda, uint64_t row0, uint64_t nrows, uint64_t n, uint64_t seed, double cond)
{
    const uint64_t total = nrows * n;
    const double inv_n = 1.0 / (double)n;
    for (uint64_t idx = (uint64_t)blockIdx.x * kBlock + threadIdx.x; idx < total; idx += (uint64_t)gridDim.x * kBlock) {
        const uint64_t i = idx / n + row0, j = idx % n;
        double v;
        if (i == j) {
            v = 1.0 + (cond - 1.0) * u01(splitmix64(seed ^ splitmix64(i * 2 + 1)));
        } else {
            const uint64_t lo = i < j ? i : j, hi = i < j ? j : i;
            v = (2.0 * u01(splitmix64(seed + splitmix64(lo * n + hi))) - 1.0) * inv_n;
        }
        A[(i - row0) * lda + j] = to_storage<TA>(v);
    }
}

// ---- dense SPD matrix with a PRESCRIBED SPECTRUM: A = H_k ... H_1 diag(eig) H_1 ... H_k, H_j = I - tau_j v_j v_j^T
// (the reference generator's law A = Q diag(exp(3.5 u)) Q^T, challenge/main/random_spd_system.cpp:66-97, with Q a product of
// Householder reflectors instead of an O(N^3) Gram-Schmidt: exact spectrum up to rounding, exactly symmetric, O(k N^2)).
// Step 0: the diagonal matrix; then per reflector  w = A v (the GEMV kernel),  u = w - (tau v.w / 2) v (host),
// A <- A - (tau v) u^T - u (tau v)^T  by the kernel below.
template <typename TA, typename TV>
__global__ void __launch_bounds__(kBlock)
gen_diag_kernel(TA *__restrict__ A, uint64_t lda, uint64_t row0, uint64_t nrows, uint64_t n, const TV *__restrict__ eig /* full length */)
{
    const uint64_t total = nrows * n;
    for (uint64_t idx = (uint64_t)blockIdx.x * kBlock + threadIdx.x; idx < total; idx += (uint64_t)gridDim.x * kBlock) {
        const uint64_t i = idx / n + row0, j = idx % n;
        A[(i - row0) * lda + j] = i == j ? to_storage<TA>((double)eig[i]) : to_storage<TA>(0.0);
    }
}

// A[i][j] -= tv[i] u[j] + u[i] tv[j]  with the two products rounded SEPARATELY and then added (no FMA contraction): element
// (j, i) adds the same two products in the other order, and a + b == b + a, so a symmetric A stays symmetric bit for bit.
template <typename TA, typename TV>
__global__ void __launch_bounds__(kBlock)
rank2_update_kernel(TA *__restrict__ A, uint64_t lda, uint64_t row0, uint64_t nrows, uint64_t n, const TV *__restrict__ tv, const TV *__restrict__ u)
{
#pragma clang fp contract(off)      // plain operators under contract(off): nothing here may be fused into an fma -- fma(tv_i, u_j,
                                    // u_i tv_j) rounds one product and not the other, and (i, j) != (j, i) in the last bit.  (The
                                    // __dmul_rn / __dadd_rn intrinsics do NOT help: they are inline functions of the HIP headers
                                    // with the default contract(fast), and came out as v_fmac_f64.)
    const uint64_t total = nrows * n;
    for (uint64_t idx = (uint64_t)blockIdx.x * kBlock + threadIdx.x; idx < total; idx += (uint64_t)gridDim.x * kBlock) {
        const uint64_t i = idx / n + row0, j = idx % n;
        const TV p1 = tv[i] * u[j];
        const TV p2 = u[i] * tv[j];
        const TV s12 = p1 + p2;
        TA *e = A + (i - row0) * lda + j;
        *e = to_storage<TA>((double)((TV)*e - s12));
    }
}

template <typename TV>
__global__ void __launch_bounds__(kBlock)
gen_rhs_kernel(TV *__restrict__ b, uint64_t row0, uint64_t n_loc, int random, uint64_t seed, double value)
{
    for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < n_loc; i += (uint64_t)gridDim.x * kBlock)
        b[i] = random ? (TV)(2.0 * u01(splitmix64(seed ^ splitmix64(0xB5ull + row0 + i))) - 1.0) : (TV)value;
}

// rows of `cols` elements between a dense staging buffer (pitch = cols) and the matrix (pitch = lda), either direction;
// TD <- TS conversion by to_storage (float -> bf16 rounds, bf16 -> float widens, same type copies)
template <typename TS, typename TD>
__global__ void __launch_bounds__(kBlock)
pitch_copy_kernel(const TS *__restrict__ src, uint64_t src_pitch, TD *__restrict__ dst, uint64_t dst_pitch, uint64_t nrows, uint64_t cols)
{
    const uint64_t total = nrows * cols;
    for (uint64_t idx = (uint64_t)blockIdx.x * kBlock + threadIdx.x; idx < total; idx += (uint64_t)gridDim.x * kBlock) {
        const uint64_t r = idx / cols, j = idx % cols;
        if constexpr (std::is_same<TS, TD>::value) dst[r * dst_pitch + j] = src[r * src_pitch + j];
        else if constexpr (std::is_same<TD, __hip_bfloat16>::value) dst[r * dst_pitch + j] = __float2bfloat16((float)src[r * src_pitch + j]);
        else dst[r * dst_pitch + j] = (TD)__bfloat162float(src[r * src_pitch + j]);
    }
}

}  // namespace lam
