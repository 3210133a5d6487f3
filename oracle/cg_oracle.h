/* cg_oracle.h -- declarations of the CPU parity oracle.  TEST INFRASTRUCTURE ONLY: see the
 * header of cg_oracle.c for who may use it and which reference lines each function follows. */
#ifndef CG_ORACLE_H
#define CG_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct oracle_stats {
    int num_iters;   /* loop counter on exit (max_iters+1 if the cap was hit) */
    int converged;   /* the reference's bool: num_iters <= max_iters */
    double rel_err;  /* sqrt(rr / bb), the recursive residual the reference prints */
    double t_total;  /* seconds in solve */
    double t_gemv;   /* seconds in gemv, summed over iterations */
} oracle_stats;

#define ORACLE_DECL(T, S)                                                                          \
    void oracle_gemv_##S(T alpha, const T *A, const T *x, T beta, T *y, size_t rows, size_t cols,   \
                         int threads);                                                             \
    T oracle_dot_##S(const T *x, const T *y, size_t n, int threads);                                \
    void oracle_axpby_##S(T alpha, const T *x, T beta, T *y, size_t n, int threads);                \
    void oracle_generate_tridiag_##S(T *A_loc, size_t row0, size_t nrows, size_t n);                \
    void oracle_generate_rhs_##S(T *b, size_t n);                                                   \
    int oracle_cg_solve_##S(const T *A, const T *b, T *x, size_t n, int max_iters, T rel_error,     \
                            int threads, oracle_stats *st);                                        \
    int oracle_cg_solve_sharded_##S(const T *A_full, const T *b, T *x, size_t n, int P,             \
                                    int max_iters, T rel_error, oracle_stats *st);                 \
    int oracle_cpu_baseline_##S(size_t n, int iters, int threads, oracle_stats *st);

ORACLE_DECL(double, f64)
ORACLE_DECL(float, f32)
#undef ORACLE_DECL

void oracle_partition(size_t n, int P, int q, size_t *row0, size_t *nrows);
int oracle_read_header(const char *path, uint64_t *rows, uint64_t *cols);
int oracle_read_f64(const char *path, double *dst, uint64_t count);
int oracle_write_f64(const char *path, const double *src, uint64_t rows, uint64_t cols);

#ifdef __cplusplus
}
#endif
#endif
