/*
 * lam_hip.h -- C ABI of the MI355X-native dense Conjugate-Gradient hot path.
 *
 * This is the drop-in boundary: plain pointers, sizes and scalars only (no C++, no torch
 * types).  The reference has no FFI layer; its operator API is the abstract class
 * LAM::ConjugateGradient<T> (challenge/main/LAM/src/ConjugateGradient.hpp:9-28) plus the
 * generate/getter methods of the distributed classes
 * (LAM/src/CPU/ConjugateGradient_CPU_MPI_OMP.hpp:31-35,
 *  LAM/src/GPU/distributed/ConjugateGradient_MultiGPUS_CUDA_NCCL.cuh:37-41).
 * The C++ classes in 2024-eumaster4hpc-student-challenge_amd/LAM/ implement that class
 * interface on top of the functions below; every entry point says which reference member
 * (file:line) it stands in for.  Paths are relative to /root/reference/challenge/main/.
 *
 * Conventions
 *   - return 0 on success, a negative LAM_HIP_E* code on failure; lam_hip_last_error()
 *     gives the message (HIP / RCCL status text included).  Every HIP and RCCL call is
 *     checked.  No exceptions cross the ABI.
 *   - a context owns all device memory, streams, events and the RCCL communicator; host
 *     buffers passed in are borrowed for the duration of the call only.
 *   - a context is not thread-safe; one solve at a time (same as the reference classes).
 *   - "shard" = one block of consecutive matrix rows living on one device, partitioned
 *     exactly like the reference: shard q of P owns rows [q*(N/P), (q+1)*(N/P)), the last
 *     shard also takes N%P (LAM/src/CPU/ConjugateGradient_CPU_MPI_OMP.hpp:176-184).
 *   - the library never falls back to the CPU: without a usable GPU every compute entry
 *     point fails with LAM_HIP_ENODEV.
 */
#ifndef LAM_HIP_H
#define LAM_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ABI history (a caller compares lam_hip_abi_version() with the LAM_HIP_ABI_VERSION it was compiled against):
 *   1  rounds 1-3.
 *   2  + lam_hip_build_id, lam_hip_generate_spectrum_spd.
 *   3  + lam_hip_debug_symv_plan; BEHAVIOUR: lam_hip_create with more than one shard defaults to the gather-Ap exchange
 *      (option "exchange" = 1; was 0), and option "symmetric" = 1 means "from 192 MiB of matrix on, any N, one or several
 *      shards" (was: one shard, N a multiple of 4096).
 *   4  (round 5) lam_hip_stats grows by t_exchange (appended: the older fields keep their offsets, but a caller must pass
 *      the larger struct); BEHAVIOUR: the gather-Ap exchange takes any N >= shards (the reference's uneven partition), so
 *      "exchange_effective" no longer drops to 0 for N % shards != 0; lam_hip_create_rank defaults to the gather-Ap
 *      exchange too (option "exchange" = 1; was 0). */
#define LAM_HIP_ABI_VERSION 4

/* most row shards of one process (lam_hip_create) / ranks of one communicator (lam_hip_create_rank); more -> LAM_HIP_EINVAL.
 * 16 until round 5; the reference's largest published GPU run has 64 ranks (TESTS/results/STRESS_TEST_GPU_MPI.txt:18). */
#define LAM_HIP_MAX_SHARDS 64

/* storage / arithmetic type of the matrix and vectors */
#define LAM_HIP_F64 0  /* double everywhere (the reference drivers hard-code <double>) */
#define LAM_HIP_F32 1  /* float storage, float FMA, double only for the reduced scalars */
#define LAM_HIP_BF16 2 /* bf16 matrix storage, fp32 vectors and accumulation (config 4) */

#define LAM_HIP_EINVAL (-1)  /* bad argument / call order */
#define LAM_HIP_ENODEV (-2)  /* no usable GPU */
#define LAM_HIP_EHIP (-3)    /* a HIP call failed */
#define LAM_HIP_ERCCL (-4)   /* an RCCL call failed */
#define LAM_HIP_ENOMEM (-5)  /* device or host allocation failed */
#define LAM_HIP_ESTATE (-6)  /* problem / matrix / rhs not set yet */

#define LAM_HIP_UNIQUE_ID_BYTES 128 /* == NCCL_UNIQUE_ID_BYTES */

typedef struct lam_hip_ctx lam_hip_ctx;

/* What the reference prints per run (test/test_CG_CPU_MPI_OMP.cpp:201-203 and
 * ConjugateGradient_MultiGPUS_CUDA_NCCL.cu:332-334,424-427), as numbers. */
typedef struct lam_hip_stats {
    int32_t num_iters;    /* loop counter on exit: converging iteration, or max_iters+1 at the cap */
    int32_t converged;    /* solve()'s bool */
    double rel_err;       /* sqrt(rr/bb), the recursive relative residual */
    double t_gemv;        /* average seconds per iteration in the GEMV kernel (device time; several local shards: the slowest one's) */
    double t_iter;        /* average seconds per iteration (wall, whole loop / iterations run) */
    double t_total;       /* wall seconds of the call */
    double t_comm_init;   /* seconds spent creating the RCCL communicator (0 if none) */
    double gemv_bytes;    /* algorithmic bytes one GEMV launch on this rank reads+writes */
    double t_exchange;    /* average seconds per iteration in the iteration's exchange step(s) on this rank / shard 0: the
                           * RCCL collective(s), or the event join(s) of one process driving several shards (from the post
                           * behind the producer kernel until the consumer's stream has passed its waits); sampled on the
                           * iterations whose GEMV is timed (option "gemv_timing").  0 for one shard and for the direct
                           * exchange (which waits inside its kernels).  t_gemv + t_exchange is what the reference prints
                           * as its t_gemv column, which includes broadcast + gather
                           * (ConjugateGradient_MultiGPUS_CUDA_NCCL.cu:352-377) */
} lam_hip_stats;

/* ---- lifecycle ---------------------------------------------------------------------------- */

int lam_hip_abi_version(void);
/* 16 hex digits identifying the sources (csrc/lam_hip.hip, csrc/lam_kernels.h, this header) the library was built
 * from (sha256 prefix, set by the Makefile).  The Python binding compares it with the sources next to it and refuses
 * a stale library.  No reference counterpart. */
const char *lam_hip_build_id(void);
int lam_hip_device_count(int *count);

/* One process driving `n_shards` row shards, shard q on device_ids[q] (device ids may repeat,
 * which puts several shards on one GPU; device_ids == NULL means 0,1,..,n_shards-1 modulo the
 * device count).  Shards exchange p slices and partial dot products by direct peer stores
 * over xGMI.  Stands in for the constructor + device discovery of the single-process class
 * LAM/src/GPU/local/ConjugateGradient_MultiGPUS_CUDA.cuh:20-22 and, with n_shards == 1, of
 * LAM/src/GPU/local/ConjugateGradient_GPU_CUDA.cuh. */
int lam_hip_create(lam_hip_ctx **out, int dtype, int n_shards, const int *device_ids);

/* One process per GPU: this process owns shard `rank` of `nranks` on `device_id`; the per
 * iteration exchange is RCCL (all-gather of p, and of the ranks' partial dot products).
 * `unique_id` = LAM_HIP_UNIQUE_ID_BYTES bytes obtained from lam_hip_get_unique_id() on one rank
 * and distributed by the caller (MPI_Bcast, torch.distributed, a file ...), exactly the
 * bootstrap of ConjugateGradient_MultiGPUS_CUDA_NCCL.cu:306-334 (ncclGetUniqueId + MPI_Bcast +
 * ncclCommInitRank, timed into the extra CSV column -> lam_hip_stats.t_comm_init). */
/* RCCL prints a version banner to STDOUT when a communicator is created.  The library leaves the process's file
 * descriptors alone by default; a caller whose stdout is a protocol (this package's drivers: one CSV line) sets the
 * environment variable LAM_HIP_QUIET_RCCL=1, and file descriptor 1 then points at stderr for the duration of
 * ncclCommInitRank (process-wide: other threads' stdout goes there too for that window). */
int lam_hip_get_unique_id(void *unique_id_out);
int lam_hip_create_rank(lam_hip_ctx **out, int dtype, int device_id, int rank, int nranks,
                        const void *unique_id);

void lam_hip_destroy(lam_hip_ctx *ctx);
const char *lam_hip_last_error(const lam_hip_ctx *ctx); /* ctx may be NULL: last create error */

/* ---- problem definition ------------------------------------------------------------------- */

/* Fix N, compute the reference row partition and allocate A (rows_loc x N, row-major, per
 * shard; on the device every row is padded to a whole number of 4-KiB pages -- an internal layout, upload / download take and
 * give dense rows -- so that any N, odd ones included, streams through the aligned 16-byte-vector kernels) and the work vectors.  Stands in for the allocation half of load_matrix_from_file /
 * generate_matrix (ConjugateGradient_CPU_MPI_OMP.hpp:176-196,214,250-253;
 * ConjugateGradient_MultiGPUS_CUDA_NCCL.cu:544-568).  Frees the previous problem's vectors first: hipFree waits for the
 * whole device AS THIS PROCESS sees it, so a process that hosts several rank contexts as threads (the test harness's shape;
 * a deployment has one rank per process) must not call it while another of its ranks has a collective in flight that waits
 * for this one.  A new matrix of the SAME size needs no new lam_hip_set_problem. */
int lam_hip_set_problem(lam_hip_ctx *ctx, uint64_t n);

/* The row partition itself, usable without a context (and without a GPU): rows of shard q of P for
 * an n x n matrix, ConjugateGradient_CPU_MPI_OMP.hpp:176-184. */
int lam_hip_partition(uint64_t n, int num_shards, int shard, uint64_t *row0, uint64_t *nrows);

int lam_hip_n(const lam_hip_ctx *ctx, uint64_t *n);
int lam_hip_num_shards(const lam_hip_ctx *ctx, int *total_shards, int *local_shards);
/* rows of global shard q (any q in [0,total_shards), local or not) */
int lam_hip_get_partition(const lam_hip_ctx *ctx, int shard, uint64_t *row0, uint64_t *nrows);

/* Copy rows [row0,row0+nrows) of the global matrix (host, row-major, nrows*N elements of the
 * context dtype; for LAM_HIP_BF16 the host rows are float and are rounded on upload) into the
 * local shard(s) that own them.  Rows owned by other processes are an error.  Stands in for
 * the read + H2D copy of load_matrix_from_file (ConjugateGradient_MultiGPUS_CUDA_NCCL.cu:
 * 568-583), with 64-bit counts. */
int lam_hip_upload_rows(lam_hip_ctx *ctx, uint64_t row0, uint64_t nrows, const void *host_rows);
/* inverse of upload (tests, and save of generated systems) */
int lam_hip_download_rows(lam_hip_ctx *ctx, uint64_t row0, uint64_t nrows, void *host_rows);

/* Dense tridiag(1,2,1) filled on device by GLOBAL row index: generate_matrix,
 * ConjugateGradient_CPU_MPI_OMP.hpp:237-247 / ConjugateGradient_MultiGPUS_CUDA_NCCL.cu:628-714. */
int lam_hip_generate_tridiag(lam_hip_ctx *ctx);
/* Dense symmetric strictly diagonally dominant SPD test matrix, generated on device from a
 * counter-based hash (no reference counterpart; replaces the MKL-based
 * challenge/main/random_spd_system.cpp for sizes that do not fit a file):
 *   A[i][j] = A[j][i] = u(seed,min,max)/N, u in [-1,1);  A[i][i] = 1 + (cond-1)*v(seed,i), v in [0,1)
 * eigenvalues lie in (0, cond+1): `cond` spreads the spectrum so CG does not converge at once. */
int lam_hip_generate_random_spd(lam_hip_ctx *ctx, uint64_t seed, double cond);

/* Dense SPD matrix with a PRESCRIBED SPECTRUM, the law of the reference's fixture generator
 * (challenge/main/random_spd_system.cpp:66-97: A = Q diag(d) Q^T, d_i = exp(3.5 U[-1,1]), cond ~ 1.1e3), built on the device:
 *   A = H_k ... H_1 diag(eig) H_1 ... H_k,   H_j = I - 2 v_j v_j^T / (v_j . v_j),   v_j = v[j*N .. j*N+N)
 * i.e. Q is a product of k Householder reflectors instead of the reference's O(N^3) Gram-Schmidt with MKL: the spectrum is
 * `eig` up to rounding, A is symmetric bit for bit, cost O(k N^2) (one GEMV + one rank-2 update pass per reflector).  eig:
 * N positive values, v: k x N (host, double).  fp64 / fp32 storage.  Collective in rank mode (every rank passes the same
 * arrays).  apps/random_spd_system.cpp draws eig, v and the rhs from srand/rand exactly as the reference does. */
int lam_hip_generate_spectrum_spd(lam_hip_ctx *ctx, const double *eig, const double *v, int k);

/* b: load_rhs_from_file (ConjugateGradient_CPU_MPI_OMP.hpp:258-305) / generate_rhs (:144-165).
 * b_host has N elements of the vector dtype (double for F64, float otherwise). */
int lam_hip_set_rhs(lam_hip_ctx *ctx, const void *b_host);
/* b back to the host (N elements of the vector dtype; single-process contexts): lets a generated system be
 * written to files in the reference's format (apps/random_spd_system.cpp, the counterpart of
 * challenge/main/random_spd_system.cpp:160-185). */
int lam_hip_get_rhs(lam_hip_ctx *ctx, void *b_host);
int lam_hip_generate_rhs(lam_hip_ctx *ctx, double value);          /* b == value (reference: 1.0) */
int lam_hip_generate_random_rhs(lam_hip_ctx *ctx, uint64_t seed);  /* b ~ U[-1,1) */

/* ---- the hot path ------------------------------------------------------------------------- */

/* solve(): ConjugateGradient_CPU_MPI_OMP.hpp:71-142 (recurrence, stop test BEFORE the p update,
 * num_iters = max_iters+1 at the cap).  Returns 0 whether or not it converged; see
 * stats->converged.  Equivalent to lam_hip_cg_init + lam_hip_cg_iterate(max_iters). */
int lam_hip_solve(lam_hip_ctx *ctx, int max_iters, double rel_error, lam_hip_stats *stats);

/* x=0, r=p=b, bb=b.b (ConjugateGradient_CPU_MPI_OMP.hpp:82-93). */
int lam_hip_cg_init(lam_hip_ctx *ctx);
/* Run up to `iters` further iterations of the loop (:98-116), continuing the iteration count
 * of earlier calls.  rel_error <= 0 never stops early (the stop test is still evaluated). */
int lam_hip_cg_iterate(lam_hip_ctx *ctx, int iters, double rel_error, lam_hip_stats *stats);

/* x (N elements of the vector dtype) to the host: the payload of save_result_to_file
 * (ConjugateGradient_CPU_OMP.hpp:199-217).  In rank mode this is a collective: every rank
 * must call it and every rank receives the full vector. */
int lam_hip_get_solution(lam_hip_ctx *ctx, void *x_host);
/* true relative residual ||b - A x||_2 / ||b||_2 recomputed on device with the GEMV kernel
 * (collective in rank mode).  Not in the reference; used for parity checks at full size. */
int lam_hip_true_residual(lam_hip_ctx *ctx, double *rel_res);

/* ---- single operators (reference private members / CUDA kernels, for parity tests, roofline
 *      probes and callers that want the BLAS pieces) ------------------------------------------ */

/* y = A x with the production GEMV kernel on the context's matrix.  x_host, y_host: N elements
 * (vector dtype).  Collective in rank mode.  gemv: ConjugateGradient_CPU_MPI_OMP.hpp:482-508,
 * CUDA gemv kernel ConjugateGradient_MultiGPUS_CUDA_NCCL.cu:185-238. */
int lam_hip_gemv(lam_hip_ctx *ctx, const void *x_host, void *y_host);
/* `reps` back-to-back launches of the production GEMV kernel(s) over the local shard(s),
 * timed with HIP events on the launch stream; *sec_per_gemv = average seconds per launch
 * (max over local shards).  No reference counterpart (roofline probe). */
int lam_hip_gemv_only(lam_hip_ctx *ctx, int reps, double *sec_per_gemv);
/* dot (ConjugateGradient_CPU_MPI_OMP.hpp:446-467; CUDA partialDot+reduce :50-131) and axpby
 * (:469-480; CUDA axpy/minusaxpy/xpby :133-183) on host vectors of n elements, run on shard 0's
 * device with the same reduction code the CG kernels use. */
int lam_hip_dot(lam_hip_ctx *ctx, const void *x_host, const void *y_host, uint64_t n, double *result);
int lam_hip_axpby(lam_hip_ctx *ctx, double alpha, const void *x_host, double beta, void *y_host,
                  uint64_t n);

/* max |A[i][j] - A[j][i]| of the matrix held by a single-PROCESS context (one shard or several; every storage type): lets a
 * caller verify the precondition of option "symmetric".  No reference counterpart. */
int lam_hip_check_symmetry(lam_hip_ctx *ctx, double *max_abs_asymmetry);

/* Host-only check of the symmetric product's PLAN (no device needed, no context): builds the task lists of all `shards` row shards
 * of an n x n problem of `dtype` exactly as the launcher does, walks every element of every task through the kernel's own use rule,
 * and reports how many directed products (y_i += A_ij p_j, i and j in [0, n)) are produced not exactly once (*bad_pairs, must be
 * 0) and how many elements of tasks flagged "interior" -- which the kernel processes without any test -- are not used by both
 * sides or lie outside the matrix (*bad_interior, must be 0); *tasks = number of tasks.  shards == 1: the upper triangle; more:
 * cyclic half windows.  O(n^2) time, n^2 / 4 bytes of memory (two bitmaps: 1 GiB at n = 65536, 4 GiB at n = 131072).  The
 * arithmetic is csrc/lam_host_plan.h, which tests/host_asan also builds with g++ -fsanitize=address,undefined.  No reference
 * counterpart. */
int lam_hip_debug_symv_plan(uint64_t n, int shards, int dtype, uint64_t *bad_pairs, uint64_t *bad_interior, uint64_t *tasks);

/* Agreement across ranks (collective in rank mode, identity otherwise): *global_ok = 1 iff every rank passed
 * local_ok != 0.  Lets a step that can fail on ONE rank (reading its row block from a file) fail on ALL of
 * them instead of leaving the others waiting in the next collective.  The reference has MPI_Abort on file
 * errors for this (ConjugateGradient_CPU_MPI_OMP.hpp:325-329). */
int lam_hip_all_ok(lam_hip_ctx *ctx, int local_ok, int *global_ok);

/* Version of the RCCL library this process is bound to (ncclGetVersion; the reference links NCCL 2.18.3,
 * LAM/CMakeLists.txt:11) -- bench.py records it next to its multi-GPU numbers. */
int lam_hip_rccl_version(int *version);
/* Name of the GEMV kernel instantiation the context launches for its current dtype / N / options, e.g.
 * "gemv_coop_kernel<double,double,R=2,TILE=4096,NT=true,UNROLL=4,WAVES=4>" (what a profiler shows).
 * No reference counterpart (the reference has one gemv kernel, NCCL.cu:185-226). */
int lam_hip_gemv_kernel_name(const lam_hip_ctx *ctx, char *buf, size_t len);

/* ---- options --------------------------------------------------------------------------------- */
/* name/value pairs; unknown names -> LAM_HIP_EINVAL.  Options that change which kernels an iteration uses need a new
 * lam_hip_cg_init.  Everything here is the PRODUCT; the experiments that were built, measured and lost (persistent launch,
 * enqueue threads, hub join, separate reduction launches, MFMA-fed GEMV, 19 GEMV tuning shapes) exist only in
 * liblam_hip_tuning.so and are described in include/lam_hip_tuning.md -- this library refuses to switch them on.
 *
 *   "exchange"      how row shards exchange per iteration.  Default 1 in both multi-GPU topologies (rank mode: 0 until round
 *                   4); environment LAM_HIP_EXCHANGE sets the default of new contexts.
 *                     1  gather-Ap: ONE exchange of [Ap slice | p.Ap part] per iteration (one ncclAllGather / one event
 *                        join), r and p full-length on every shard (the reference CPU path's layout, CPU_MPI_OMP.hpp:476,505).
 *                        Any N >= shards: records hold the longest slice of the reference's uneven partition (:176-196).
 *                     0  sliced vectors: 8-byte-per-rank all-gathers for p.Ap and r.r (summed in rank order by the consumer)
 *                        + all-gather of the p slices / three event joins.
 *                     2  DIRECT, EXPERIMENTAL (never yet run on separate GPUs; cross-GPU parity unpinned): peer-mapped
 *                        mailboxes and p replicas, tagged in-kernel hand-overs, no collective and no event in the iteration;
 *                        bit-identical to 0.  lam_hip_solve verifies itself on it ("verify_direct", "direct_fallbacks") and
 *                        falls back to 0; from the environment only together with LAM_HIP_EXPERIMENTAL_DIRECT=1; shards
 *                        sharing a device get it only with LAM_HIP_DIRECT_SAME_DEVICE=1 (tests).
 *                   "exchange_effective" (get) tells what the current CG state runs on.
 *   "exchange_join" one process, exchange 1: 1 (default for > 2 shards) = the join goes through shard 0's stream
 *                   (2(P-1)+1 runtime calls), 0 = every stream waits for every other one (P(P-1)).  Same bits.
 *   "overlap"       rank mode, exchange 0: 1 (default) = all-gather of p on a second stream under the own-slice GEMV panel.
 *                   Exchange 2: 1 = own-slice panel in front of the wait for the peers' slices, 0 = wait first.
 *   "symmetric"     the product reads every pair {A[i][j], A[j][i]} ONCE (A must equal its transpose): half the HBM traffic.
 *                   One shard: the upper triangle; several (exchange 1): cyclic half windows per row, each shard contributes
 *                   a full-length vector to the exchange.  1 = where it pays (from 192 MiB of matrix on), 2 = always, 0
 *                   (default) = the reference's general GEMV.  Same results to rounding.  "symmetric_effective" (get).
 *                   Environment LAM_HIP_SYMMETRIC = 1 | 2 (drivers): the library then checks A = A^T itself once per matrix
 *                   (one process; warns at rounding level, refuses beyond; rank mode: unchecked) and says on stderr when the
 *                   option is not effective (several shards / ranks on an exchange other than 1).
 *   "fuse_update"   1 (default) = the x, r, p updates of an iteration are ONE launch (r.r handed over inside the launch):
 *                   2 launches per shard and iteration.  Used only when the whole grid of all shards / ranks on the device
 *                   is resident ("fuse_effective" tells; "assume_cus" overrides the CU count for tests).  Same bits.
 *   "gemv_timing"   T (default 8): HIP-event pairs bracket the GEMV -- and the exchange step(s) -- of every T-th iteration
 *                   (lam_hip_stats.t_gemv / t_exchange); 0 = never.
 *   "gemv_variant"  -1 (default) = production shape of the dtype (13 fp64, 10 fp32, 0 bf16); 17 = 4 rows per 8-wave
 *                   workgroup (faster at exactly 16 column tiles only).  Others: tuning build.
 *   "nt_loads" 1, "force_generic" 0, "probe_rows", "panel_lo"/"panel_hi"   kernel-level switches for tests and probes.
 *   "reuse_matrix"  1 (default) = lam_hip_set_problem keeps the matrix allocation when it is large enough (grow-only).
 *   "upload_staging" 1 = lam_hip_upload_rows copies through two pinned staging buffers (default 0: measured slower).
 *   get only: "row_pitch" (elements between rows on the device), "collectives_enqueued", "rccl_ranks" (ncclCommCount of
 *                   the context's communicator, 0 without one), "ranks_on_device", "gemv_ns_min_shard" / "gemv_ns_max_shard" (fastest /
 *                   slowest local shard's average GEMV of the last cg_iterate call: their difference is the skew), "host_cpu_ns", "host_enqueue_ns",
 *                   "hip_calls_launch" / "_record" / "_wait" / "_setdevice", "tuning_variants" (1 in the tuning build). */
int lam_hip_set_option(lam_hip_ctx *ctx, const char *name, int64_t value);
int lam_hip_get_option(const lam_hip_ctx *ctx, const char *name, int64_t *value);

#ifdef __cplusplus
}
#endif
#endif /* LAM_HIP_H */
