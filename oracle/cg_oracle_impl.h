/*
 * TEST INFRASTRUCTURE ONLY -- see cg_oracle.c for the header that applies to this file.
 *
 * Type-generic body of the CPU oracle.  Included twice by cg_oracle.c with
 *   ORACLE_T   = double / float
 *   ORACLE_(n) = n##_f64 / n##_f32
 *
 * "ref:" comments cite the reference file:line each function restates.  Paths are
 * relative to /root/reference/challenge/main/LAM/src/CPU/.
 */

/* ref: ConjugateGradient_CPU_OMP.hpp:246-263 and ConjugateGradient_CPU_MPI_OMP.hpp:482-503.
 * y[r] = beta*y[r] + sum_c (alpha*A[r,c])*x[c], c ascending, ONE accumulator per row,
 * term order (alpha*A)*x exactly as the reference writes it.  `y_off` lets the sharded
 * caller address y[offset+r] like the MPI variant does (:502). */
void ORACLE_(oracle_gemv)(ORACLE_T alpha, const ORACLE_T *A, const ORACLE_T *x, ORACLE_T beta,
                          ORACLE_T *y, size_t rows, size_t cols, int threads)
{
    if (threads < 1) threads = 1;
#pragma omp parallel for num_threads(threads) schedule(static)
    for (size_t r = 0; r < rows; r++) {
        ORACLE_T y_val = 0.0;
        const ORACLE_T *Ar = A + r * cols;
        for (size_t c = 0; c < cols; c++) {
            y_val += alpha * Ar[c] * x[c];
        }
        y[r] = beta * y[r] + y_val;
    }
}

/* ref: ConjugateGradient_CPU_OMP.hpp:219-231 / ..._MPI_OMP.hpp:446-461 (local part).
 * With threads == 1 the summation order is i ascending into one accumulator, which is what
 * the reference does at OMP_NUM_THREADS=1. */
ORACLE_T ORACLE_(oracle_dot)(const ORACLE_T *x, const ORACLE_T *y, size_t n, int threads)
{
    ORACLE_T result = 0.0;
    if (threads <= 1) {
        for (size_t i = 0; i < n; i++) result += x[i] * y[i];
        return result;
    }
#pragma omp parallel for num_threads(threads) reduction(+ : result) schedule(static)
    for (size_t i = 0; i < n; i++) result += x[i] * y[i];
    return result;
}

/* ref: ConjugateGradient_CPU_OMP.hpp:233-244 / ..._MPI_OMP.hpp:469-480.  y = alpha*x + beta*y */
void ORACLE_(oracle_axpby)(ORACLE_T alpha, const ORACLE_T *x, ORACLE_T beta, ORACLE_T *y, size_t n,
                           int threads)
{
    if (threads < 1) threads = 1;
#pragma omp parallel for num_threads(threads) schedule(static)
    for (size_t i = 0; i < n; i++) y[i] = alpha * x[i] + beta * y[i];
}

/* ref: ConjugateGradient_CPU_MPI_OMP.hpp:237-247 (matrix) -- rows [row0,row0+nrows) of the dense
 * N x N tridiag(1,2,1), selected by GLOBAL row index like `i+_offset`. */
void ORACLE_(oracle_generate_tridiag)(ORACLE_T *A_loc, size_t row0, size_t nrows, size_t n)
{
    for (size_t i = 0; i < nrows; i++) {
        size_t g = i + row0;
        ORACLE_T *Ar = A_loc + i * n;
        for (size_t j = 0; j < n; j++) {
            if (g + 1 == j || g == j + 1) Ar[j] = 1;
            else if (g == j) Ar[j] = 2;
            else Ar[j] = 0;
        }
    }
}

/* ref: ConjugateGradient_CPU_MPI_OMP.hpp:159-162.  b == 1 */
void ORACLE_(oracle_generate_rhs)(ORACLE_T *b, size_t n)
{
    for (size_t i = 0; i < n; i++) b[i] = 1.0;
}

/* ref: ConjugateGradient_CPU_OMP.hpp:49-91 (single process).  Returns 1 if converged within
 * max_iters (the reference's bool), 0 otherwise.  st->num_iters is the loop counter on exit:
 * the converging iteration, or max_iters+1 when the cap is hit (that is what the MPI variant
 * prints, ..._MPI_OMP.hpp:125).  The stop test is BEFORE the p update (:83-84). */
int ORACLE_(oracle_cg_solve)(const ORACLE_T *A, const ORACLE_T *b, ORACLE_T *x, size_t n,
                             int max_iters, ORACLE_T rel_error, int threads, oracle_stats *st)
{
    ORACLE_T alpha, beta, rhs_module, rr, rr_new;
    int num_iters;
    if (threads < 1) threads = 1;
    ORACLE_T *r = (ORACLE_T *)malloc(n * sizeof(ORACLE_T));
    ORACLE_T *p = (ORACLE_T *)malloc(n * sizeof(ORACLE_T));
    ORACLE_T *Ap = (ORACLE_T *)malloc(n * sizeof(ORACLE_T));
    if (!r || !p || !Ap) { free(r); free(p); free(Ap); return -1; }
    double t_gemv = 0.0, t0 = oracle_now();

#pragma omp parallel for num_threads(threads) schedule(static)
    for (size_t i = 0; i < n; i++) { /* :55-62 */
        Ap[i] = 0.0;
        x[i] = 0.0;
        r[i] = b[i];
        p[i] = b[i];
    }
    rhs_module = ORACLE_(oracle_dot)(b, b, n, threads); /* :64 */
    rr = rhs_module;
    for (num_iters = 1; num_iters <= max_iters; num_iters++) { /* :67 */
        double g0 = oracle_now();
        ORACLE_(oracle_gemv)(1.0, A, p, 0.0, Ap, n, n, threads);   /* :69 */
        t_gemv += oracle_now() - g0;
        alpha = rr / ORACLE_(oracle_dot)(p, Ap, n, threads);       /* :70 */
        ORACLE_(oracle_axpby)(alpha, p, 1.0, x, n, threads);       /* :71 */
        ORACLE_(oracle_axpby)(-alpha, Ap, 1.0, r, n, threads);     /* :72 */
        rr_new = ORACLE_(oracle_dot)(r, r, n, threads);            /* :73 */
        beta = rr_new / rr;                                        /* :74 */
        rr = rr_new;                                               /* :75 */
        if (ORACLE_SQRT(rr / rhs_module) < rel_error) break;        /* :76 */
        ORACLE_(oracle_axpby)(1.0, r, beta, p, n, threads);        /* :77 */
    }
    if (st) {
        st->num_iters = num_iters;
        st->rel_err = (double)ORACLE_SQRT(rr / rhs_module);
        st->converged = num_iters <= max_iters;
        st->t_total = oracle_now() - t0;
        st->t_gemv = t_gemv;
    }
    free(r); free(p); free(Ap);
    return num_iters <= max_iters;
}

/* ref: ConjugateGradient_CPU_MPI_OMP.hpp:71-142 with P emulated ranks in ONE process.
 * Rank q owns rows [q*(n/P), ...) and the last rank takes the n%P remainder (:176-184).
 * gemv: each rank computes its block into y_temp, Allgatherv -> full Ap (:482-508).
 * dot : each rank sums its own row range (:457-461); the Allreduce(SUM) (:464) is emulated by
 *       adding the P partials in rank order 0..P-1 (an MPI library may use another tree; the
 *       difference is O(eps) and parity against the real mpiexec runs is tolerance-based).
 * axpby: full length on every rank (:476) -- identical on all ranks, so done once.
 * A_full is the full n x n row-major matrix (the emulated ranks index their block in place). */
int ORACLE_(oracle_cg_solve_sharded)(const ORACLE_T *A_full, const ORACLE_T *b, ORACLE_T *x,
                                     size_t n, int P, int max_iters, ORACLE_T rel_error,
                                     oracle_stats *st)
{
    ORACLE_T alpha, beta, rhs_module, rr, rr_new;
    int num_iters;
    if (P < 1) return -1;
    size_t base = n / (size_t)P;
    ORACLE_T *r = (ORACLE_T *)malloc(n * sizeof(ORACLE_T));
    ORACLE_T *p = (ORACLE_T *)malloc(n * sizeof(ORACLE_T));
    ORACLE_T *Ap = (ORACLE_T *)malloc(n * sizeof(ORACLE_T));
    if (!r || !p || !Ap) { free(r); free(p); free(Ap); return -1; }
    double t0 = oracle_now(), t_gemv = 0.0;
    for (size_t i = 0; i < n; i++) { Ap[i] = 0.0; x[i] = 0.0; r[i] = b[i]; p[i] = b[i]; }

#define SHARDED_DOT(u, v, out)                                                         \
    do {                                                                               \
        ORACLE_T acc_ = 0.0;                                                           \
        for (int q_ = 0; q_ < P; q_++) {                                               \
            size_t off_ = base * (size_t)q_;                                           \
            size_t nl_ = base + ((q_ == P - 1) ? n % (size_t)P : 0);                   \
            ORACLE_T loc_ = ORACLE_(oracle_dot)((u) + off_, (v) + off_, nl_, 1);       \
            acc_ = (q_ == 0) ? loc_ : acc_ + loc_;                                     \
        }                                                                              \
        (out) = acc_;                                                                  \
    } while (0)

    SHARDED_DOT(b, b, rhs_module);
    rr = rhs_module;
    for (num_iters = 1; num_iters <= max_iters; num_iters++) {
        double g0 = oracle_now();
        for (int q = 0; q < P; q++) {
            size_t off = base * (size_t)q;
            size_t nl = base + ((q == P - 1) ? n % (size_t)P : 0);
            ORACLE_(oracle_gemv)(1.0, A_full + off * n, p, 0.0, Ap + off, nl, n, 1);
        }
        t_gemv += oracle_now() - g0;
        ORACLE_T pAp;
        SHARDED_DOT(p, Ap, pAp);
        alpha = rr / pAp;
        ORACLE_(oracle_axpby)(alpha, p, 1.0, x, n, 1);
        ORACLE_(oracle_axpby)(-alpha, Ap, 1.0, r, n, 1);
        SHARDED_DOT(r, r, rr_new);
        beta = rr_new / rr;
        rr = rr_new;
        if (ORACLE_SQRT(rr / rhs_module) < rel_error) break;
        ORACLE_(oracle_axpby)(1.0, r, beta, p, n, 1);
    }
#undef SHARDED_DOT
    if (st) {
        st->num_iters = num_iters;
        st->rel_err = (double)ORACLE_SQRT(rr / rhs_module);
        st->converged = num_iters <= max_iters;
        st->t_total = oracle_now() - t0;
        st->t_gemv = t_gemv;
    }
    free(r); free(p); free(Ap);
    return num_iters <= max_iters;
}

/* Timed fixed-iteration run for bench.py's cpu_baseline leg ("port" kind): allocates the
 * generate-mode system (tridiag(1,2,1), b = 1) with a parallel first touch like
 * ConjugateGradient_CPU_OMP.hpp:179-184, then runs exactly `iters` CG iterations
 * (rel_error = 0 never triggers the stop test).  Returns seconds per iteration in st. */
int ORACLE_(oracle_cpu_baseline)(size_t n, int iters, int threads, oracle_stats *st)
{
    if (threads < 1) threads = 1;
    ORACLE_T *A = (ORACLE_T *)malloc(n * n * sizeof(ORACLE_T));
    ORACLE_T *b = (ORACLE_T *)malloc(n * sizeof(ORACLE_T));
    ORACLE_T *x = (ORACLE_T *)malloc(n * sizeof(ORACLE_T));
    if (!A || !b || !x) { free(A); free(b); free(x); return -1; }
#pragma omp parallel for num_threads(threads) schedule(static)
    for (size_t i = 0; i < n; i++) {
        ORACLE_T *Ar = A + i * n;
        for (size_t j = 0; j < n; j++)
            Ar[j] = (i + 1 == j || i == j + 1) ? 1 : (i == j ? 2 : 0);
    }
    ORACLE_(oracle_generate_rhs)(b, n);
    int rc = ORACLE_(oracle_cg_solve)(A, b, x, n, iters, (ORACLE_T)0.0, threads, st);
    free(A); free(b); free(x);
    return rc < 0 ? rc : 0;
}
