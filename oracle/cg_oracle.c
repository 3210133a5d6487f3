/*
 * cg_oracle.c -- CPU restatement of the reference's dense Conjugate-Gradient hot path.
 *
 * ============================ TEST INFRASTRUCTURE ONLY =============================
 * This file is the parity ORACLE.  Only tests/, __graft_entry__.smoke() and bench.py's
 * `cpu_baseline` leg may load it, and only as the checker / the reported CPU baseline.
 * Nothing in the product path (include/lam_hip.h, csrc/, the C++ host classes, the
 * drivers) links, loads or calls it; the product fails loudly without its HIP library.
 * ====================================================================================
 *
 * What it restates (plain C, same arithmetic order as the reference run single-threaded):
 *   /root/reference/challenge/main/LAM/src/CPU/ConjugateGradient_CPU_OMP.hpp
 *       solve :49-91   dot :219-231   axpby :233-244   gemv :246-263
 *   /root/reference/challenge/main/LAM/src/CPU/ConjugateGradient_CPU_MPI_OMP.hpp
 *       solve :71-142  generate_rhs :144-165  generate_matrix :167-256 (partition :176-196,
 *       tridiag fill :237-247)  dot+Allreduce :446-467  axpby :469-480  gemv+Allgatherv :482-508
 *
 * Pinning (see tests/golden/make_golden.py and tests/test_oracle_golden.py): the oracle is
 * checked against outputs of the reference itself, compiled here by oracle/Makefile into
 * oracle/_ref/ from the sources where they lie under /root/reference, run with
 * OMP_NUM_THREADS=1, plus the closed-form generate-mode answers that the reference's own
 * result CSVs contain (finding 1).
 *
 * Build flags must not allow FMA contraction or reassociation (-O3 -ffp-contract=off, no
 * -ffast-math), so that the single-thread result is bit-identical to the reference's
 * `g++ -O3` build on x86-64.
 */
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "cg_oracle.h"

static double oracle_now(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

#define ORACLE_T double
#define ORACLE_(n) n##_f64
#define ORACLE_SQRT(v) sqrt(v)
#include "cg_oracle_impl.h"
#undef ORACLE_T
#undef ORACLE_
#undef ORACLE_SQRT

#define ORACLE_T float
#define ORACLE_(n) n##_f32
#define ORACLE_SQRT(v) sqrtf(v)
#include "cg_oracle_impl.h"
#undef ORACLE_T
#undef ORACLE_
#undef ORACLE_SQRT

/* ref: ConjugateGradient_CPU_MPI_OMP.hpp:176-196 -- the 1-D block-row partition.
 * rank q: rows_loc = n/P (+ n%P on the last rank), offset = (n/P)*q. */
void oracle_partition(size_t n, int P, int q, size_t *row0, size_t *nrows)
{
    size_t base = n / (size_t)P;
    *row0 = base * (size_t)q;
    *nrows = base + ((q == P - 1) ? n % (size_t)P : 0);
}

/* On-disk format, ref: challenge/main/random_spd_system.cpp:105-121 (writer) and
 * ConjugateGradient_CPU_OMP.hpp:148-149,192 (reader): native-endian
 *   uint64 rows, uint64 cols, rows*cols values row-major.
 * The reference's save_result_to_file writes an `int` through sizeof(size_t)
 * (ConjugateGradient_CPU_OMP.hpp:208-210), so the upper 4 bytes of the cols word of a
 * solution file are garbage: readers mask cols with 0xffffffff. */
int oracle_read_header(const char *path, uint64_t *rows, uint64_t *cols)
{
    FILE *f = fopen(path, "rb");
    if (!f) return -1;
    uint64_t h[2];
    if (fread(h, sizeof(uint64_t), 2, f) != 2) { fclose(f); return -2; }
    fclose(f);
    *rows = h[0];
    *cols = h[1] & 0xffffffffull;
    return 0;
}

int oracle_read_f64(const char *path, double *dst, uint64_t count)
{
    FILE *f = fopen(path, "rb");
    if (!f) return -1;
    if (fseek(f, 16, SEEK_SET) != 0) { fclose(f); return -2; }
    size_t got = fread(dst, sizeof(double), count, f);
    fclose(f);
    return got == count ? 0 : -3;
}

int oracle_write_f64(const char *path, const double *src, uint64_t rows, uint64_t cols)
{
    FILE *f = fopen(path, "wb");
    if (!f) return -1;
    uint64_t h[2] = {rows, cols};
    int ok = fwrite(h, sizeof(uint64_t), 2, f) == 2 &&
             fwrite(src, sizeof(double), rows * cols, f) == rows * cols;
    fclose(f);
    return ok ? 0 : -2;
}
