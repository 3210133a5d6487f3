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
#include <new>
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

#include "lam_ctx.h"
#include "lam_launch.h"
#include "lam_exchange.h"
#include "lam_iterate.h"


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

static_assert(lam::kMaxShards == LAM_HIP_MAX_SHARDS, "include/lam_hip.h states the limit");
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
    if (n_shards > 1) c->opt_exchange = kOneProcessDefaultExchange;
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
    c->opt_exchange = exchange_from_env(kRankModeDefaultExchange);   // default exchange for this context
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
        rc = count_ranks_on_device(c.get());
        if (rc != 0) {
            g_create_error = c->err;
            (void)ncclCommDestroy(c->comm);
            c->comm = nullptr;
            abandon(c.get());
            return rc;
        }
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
    c->iter_est_s = 0.0;           // the observed iteration time belongs to the previous problem
    c->lda = lam_hip_ctx::pitch_for(n, c->esz_a());
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
        const size_t needA = std::max<size_t>(16, s.nrows * c->lda * ea);
        if (s.A != nullptr && s.A_capacity < needA) { (void)hipFree(s.A); s.A = nullptr; s.A_capacity = 0; }
        if (s.A == nullptr) {
            HIPCHK(c, hipMalloc(&s.A, needA));
            s.A_capacity = needA;
        }
        // p and tmp are what the GEMV reads as its vector: the vector kernels read whole 16-byte vectors of the matrix row, so up
        // to 7 elements behind the end of the vector are touched (against zeros of the row padding): allocated and kept ZERO
        HIPCHK(c, hipMalloc(&s.p, n * ev + 64));
        HIPCHK(c, hipMalloc(&s.tmp, n * ev + 64));
        void **vecs[] = {&s.Ap, &s.x, &s.r, &s.b};
        for (auto v : vecs) HIPCHK(c, hipMalloc(v, s.nrows * ev + 16));
        s.gemv_blocks = dispatch(c, [&](auto impl) -> int { return decltype(impl)::gemv_grid(c, s.nrows); });
        // worst case over kernel variants (generic kernel: 4 rows per workgroup) and the symmetric product, whose second pass
        // leaves one partial per kSymvReduceRows entries of the FULL-length vector whatever the shard's share of the rows (more
        // than the shard has rows from 33 shards on)
        const int gemv_blocks_max = (int)std::max<uint64_t>(s.nrows, (n + kSymvReduceRows - 1) / kSymvReduceRows) + 1;
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
            s.ap_gather_bytes = ((size_t)c->total_shards * c->ex1_stride_bytes() + 16 + 255) / 256 * 256;
            HIPCHK(c, hipMalloc(&s.ap_gather, s.ap_gather_bytes * (c->rank_mode ? 1 : 2)));
        }
        HIPCHK(c, hipMalloc((void **)&s.sc, sizeof(CgScalars)));
        HIPCHK(c, hipHostMalloc((void **)&s.sc_host, sizeof(CgScalars), hipHostMallocDefault));
        HIPCHK(c, hipHostMalloc((void **)&s.host_flags, 64, hipHostMallocDefault));
        s.host_flags[0] = s.host_flags[1] = 0;
        HIPCHK(c, hipMemsetAsync(s.sc, 0, sizeof(CgScalars), s.stream));
        HIPCHK(c, hipMemsetAsync(s.gather_a, 0, sizeof(double) * kMaxShards, s.stream));
        HIPCHK(c, hipMemsetAsync(s.gather_b, 0, sizeof(double) * kMaxShards, s.stream));
        HIPCHK(c, hipMemsetAsync(s.p, 0, n * ev + 64, s.stream));
        HIPCHK(c, hipMemsetAsync(s.tmp, 0, n * ev + 64, s.stream));
        // the padding columns of the matrix must read as zero (they meet the zeros behind p, and 0 x garbage could be a NaN):
        // one memset of the shard when rows are padded at all (a round N = 65536 has no padding and costs nothing here)
        if (c->lda != n) HIPCHK(c, hipMemsetAsync(s.A, 0, s.nrows * c->lda * ea, s.stream));
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
        char *dptr = (char *)s.A + (lo - s.row0) * c->lda * ea;
        if (c->dtype == LAM_HIP_BF16 || c->lda != c->n) {
            // The device rows are padded (pitch lda > N) and / or of another type than the host's (bf16 storage travels as float):
            // whole rows go through a DENSE device staging buffer with ONE contiguous copy per chunk (the rate of the plain path:
            // the runtime pins the caller's pages), and a kernel moves them between the two layouts at HBM speed.
            const uint64_t chunk_rows = std::max<uint64_t>(1, ((upload ? 256ull : 1024ull) << 20) / (c->n * eh));   // D2H to pageable memory likes big pieces
            const size_t need = std::min<uint64_t>(chunk_rows, hi - lo) * c->n * eh;
            if (s.xfer_stage_bytes < need) {
                if (s.xfer_stage) { (void)hipFree(s.xfer_stage); s.xfer_stage = nullptr; s.xfer_stage_bytes = 0; }
                HIPCHK(c, hipMalloc(&s.xfer_stage, need));
                s.xfer_stage_bytes = need;
            }
            struct { void *p; } stage_buf{s.xfer_stage};
            for (uint64_t r = lo; r < hi; r += chunk_rows) {
                const uint64_t nr = std::min(chunk_rows, hi - r);
                char *hp = (char *)host + (r - row0) * c->n * eh;
                char *dp = (char *)s.A + (r - s.row0) * c->lda * ea;
                if (upload) HIPCHK(c, hipMemcpyAsync(stage_buf.p, hp, nr * c->n * eh, hipMemcpyHostToDevice, s.stream));
                const dim3 grid((unsigned)std::max<uint64_t>(1, std::min<uint64_t>(4096, (nr * c->n + kBlock - 1) / kBlock)));
                if (c->dtype == LAM_HIP_BF16) {
                    if (upload) hipLaunchKernelGGL((pitch_copy_kernel<float, __hip_bfloat16>), grid, dim3(kBlock), 0, s.stream, (const float *)stage_buf.p, c->n, (__hip_bfloat16 *)dp, c->lda, nr, c->n);
                    else hipLaunchKernelGGL((pitch_copy_kernel<__hip_bfloat16, float>), grid, dim3(kBlock), 0, s.stream, (const __hip_bfloat16 *)dp, c->lda, (float *)stage_buf.p, c->n, nr, c->n);
                } else if (c->dtype == LAM_HIP_F64) {
                    if (upload) hipLaunchKernelGGL((pitch_copy_kernel<double, double>), grid, dim3(kBlock), 0, s.stream, (const double *)stage_buf.p, c->n, (double *)dp, c->lda, nr, c->n);
                    else hipLaunchKernelGGL((pitch_copy_kernel<double, double>), grid, dim3(kBlock), 0, s.stream, (const double *)dp, c->lda, (double *)stage_buf.p, c->n, nr, c->n);
                } else {
                    if (upload) hipLaunchKernelGGL((pitch_copy_kernel<float, float>), grid, dim3(kBlock), 0, s.stream, (const float *)stage_buf.p, c->n, (float *)dp, c->lda, nr, c->n);
                    else hipLaunchKernelGGL((pitch_copy_kernel<float, float>), grid, dim3(kBlock), 0, s.stream, (const float *)dp, c->lda, (float *)stage_buf.p, c->n, nr, c->n);
                }
                HIPCHK(c, hipGetLastError());
                if (!upload) HIPCHK(c, hipMemcpyAsync(hp, stage_buf.p, nr * c->n * eh, hipMemcpyDeviceToHost, s.stream));
                HIPCHK(c, hipStreamSynchronize(s.stream));
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
    if (upload) matrix_changed(c);
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
            hipLaunchKernelGGL((gen_tridiag_kernel<TA>), dim3(4096), dim3(kBlock), 0, s.stream, (TA *)s.A, c->lda, s.row0, s.nrows, c->n);
            HIPCHK(c, hipGetLastError());
        }
        return 0;
    }));
    LAMCHK(sync_all(c));
    matrix_changed(c);
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
            hipLaunchKernelGGL((gen_random_spd_kernel<TA>), dim3(4096), dim3(kBlock), 0, s.stream, (TA *)s.A, c->lda, s.row0, s.nrows, c->n, seed, cond);
            HIPCHK(c, hipGetLastError());
        }
        return 0;
    }));
    LAMCHK(sync_all(c));
    matrix_changed(c);
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
            hipLaunchKernelGGL((gen_diag_kernel<TA, TV>), dim3(4096), dim3(kBlock), 0, s.stream, (TA *)s.A, c->lda, s.row0, s.nrows, n, (const TV *)s.tmp);
            HIPCHK(c, hipGetLastError());
        }
        return 0;
    }));
    LAMCHK(sync_all(c));
    matrix_changed(c);
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
                hipLaunchKernelGGL((rank2_update_kernel<TA, TV>), dim3(4096), dim3(kBlock), 0, s.stream, (TA *)s.A, c->lda, s.row0, s.nrows, n,
                                   (const TV *)s.tmp, (const TV *)s.p);
                HIPCHK(c, hipGetLastError());
            }
            return 0;
        }));
        LAMCHK(sync_all(c));
    }
    matrix_changed(c);
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
    IterTimes times;
    int enq = 0;
    // already converged in an earlier call?
    LAMCHK(set_dev(c, s0));
    HIPCHK(c, hipMemcpyAsync(s0.sc_host, s0.sc, sizeof(CgScalars), hipMemcpyDeviceToHost, s0.stream));
    HIPCHK(c, hipStreamSynchronize(s0.stream));
    const bool stopped = s0.sc_host->stop != 0;
    const int k_first = c->k_done + 1;
    for (auto &sh_ : c->sh)
        for (int i = 0; i < kLag; i++) { sh_.timed_slot[i] = false; sh_.nx[i] = 0; }
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
        LAMCHK(iterate_threaded(c, iters, k_first, rel_error, &enq, &times));
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
                harvest_times(c, slot, &times);
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
        LAMCHK(await_progress(c, s0, k_first + enq - 1, &pr, /*precise=*/true));
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
    for (int j = 0; j < kLag; j++) harvest_times(c, j, &times);
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
        // the slowest local shard's average (one process driving several GPUs: the device that bounds the iteration); the fastest
        // one's is kept next to it (options "gemv_ns_min_shard" / "gemv_ns_max_shard": their difference is the shards' skew)
        double tmin = 0.0, tmax = 0.0;
        for (size_t j = 0; j < c->sh.size(); j++) {
            if (times.samples[j] == 0) continue;
            const double avg = times.gemv_ms[j] * 1e-3 / times.samples[j];
            tmax = std::max(tmax, avg);
            tmin = tmin == 0.0 ? avg : std::min(tmin, avg);
        }
        c->t_gemv_min = tmin;
        c->t_gemv_max = tmax;
        st->t_gemv = tmax;
        st->t_exchange = times.samples[0] > 0 ? times.xch_ms * 1e-3 / times.samples[0] : 0.0;
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
    if (c->rank_mode) return fail(c, LAM_HIP_EINVAL, "symmetry check needs the whole matrix in one process");
    double max_abs = 0.0;
    return measure_asymmetry(c, max_abs_asymmetry, &max_abs);
}

int lam_hip_debug_symv_plan(uint64_t n, int shards, int dtype, uint64_t *bad_pairs, uint64_t *bad_interior, uint64_t *ntasks)
{
    if (dtype != LAM_HIP_F64 && dtype != LAM_HIP_F32 && dtype != LAM_HIP_BF16) return LAM_HIP_EINVAL;
    // host-only arithmetic (lam_host_plan.h): the same code runs under AddressSanitizer in tests/host_asan
    const uint64_t vec = dtype == LAM_HIP_F64 ? 2 : (dtype == LAM_HIP_F32 ? 4 : 8);
    try {
        return lam::symv_plan_check(n, shards, vec, kMaxShards, bad_pairs, bad_interior, ntasks) == 0 ? 0 : LAM_HIP_EINVAL;
    } catch (const std::bad_alloc &) {
        return LAM_HIP_ENOMEM;       // two bitmaps of n^2 / 8 bytes each
    }
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
    if (c->agree_buf == nullptr) HIPCHK(c, hipMalloc((void **)&c->agree_buf, kAgreeBytes));
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
    else if (!strcmp(name, "rccl_ranks")) {
        // the size of the communicator AS RCCL REPORTS IT (ncclCommCount), 0 without one: "did RCCL see N ranks" (bench.py)
        int count = 0;
        if (c->comm != nullptr && ncclCommCount(c->comm, &count) != ncclSuccess) count = -1;
        *value = count;
    }
    else if (!strcmp(name, "ranks_on_device")) *value = c->ranks_on_device;
    else if (!strcmp(name, "gemv_ns_min_shard")) *value = (int64_t)(c->t_gemv_min * 1e9);
    else if (!strcmp(name, "gemv_ns_max_shard")) *value = (int64_t)(c->t_gemv_max * 1e9);
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
    else if (!strcmp(name, "row_pitch")) *value = (int64_t)c->lda;
    else if (!strcmp(name, "hip_calls_launch")) *value = (int64_t)c->n_launch.load();
    else if (!strcmp(name, "hip_calls_record")) *value = (int64_t)c->n_record.load();
    else if (!strcmp(name, "hip_calls_wait")) *value = (int64_t)c->n_wait.load();
    else if (!strcmp(name, "hip_calls_setdevice")) *value = (int64_t)c->n_setdev.load();
    else if (!strcmp(name, "symmetric")) *value = c->opt_symmetric;
    else if (!strcmp(name, "symmetric_effective")) *value = (c->symv_active() || c->symv_multi_active()) ? 1 : 0;
    else if (!strcmp(name, "exchange_effective")) *value = c->cg_direct ? 2 : ((c->exchange1_ok() && !c->exchange2_wanted()) ? 1 : 0);
    else if (!strcmp(name, "panel_lo")) *value = c->opt_panel_lo;
    else if (!strcmp(name, "panel_hi")) *value = c->opt_panel_hi;
    else return LAM_HIP_EINVAL;
    return 0;
}

}  // extern "C"
