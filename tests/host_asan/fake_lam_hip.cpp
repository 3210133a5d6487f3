// fake_lam_hip.cpp -- TEST INFRASTRUCTURE.  A host-memory stand-in for the C ABI of include/lam_hip.h, just enough of it for the
// HOST-side classes (LAM/src/HIP/*.hpp: file loaders, partition handling, launcher glue) to run without a GPU under
// g++ -fsanitize=address,undefined (tests/test_host_asan_cpu.py).  "Device" memory is a std::vector sized for exactly the rows this
// process owns, so a loader that uploads a row it does not own, reads past the end of its mapping, or miscounts a chunk trips either
// an explicit check here or the sanitizer.  Nothing of the product is replaced by this: the product's library is liblam_hip.so.
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/lam_hip.h"
#include "../../2024-eumaster4hpc-student-challenge_amd/csrc/lam_host_plan.h"

#ifndef LAM_LOADER_CHUNK_BYTES_MAX
#define LAM_LOADER_CHUNK_BYTES_MAX (1ull << 30)      /* what one lam_hip_upload_rows call may carry (the loaders' chunk) */
#endif

struct lam_hip_ctx {
    int dtype = 0, total = 1, local = 1, rank = 0;
    uint64_t n = 0, row0 = 0, nrows = 0;       // rows this process owns (all shards of a one-process context)
    std::vector<char> A, b;
    std::vector<unsigned char> row_uploaded;
    std::string err;
    size_t es() const { return dtype == LAM_HIP_F64 ? 8 : 4; }     // host element size (bf16 travels as float)
};
static std::string g_err;
static int fail(lam_hip_ctx *c, int code, const char *msg) { (c ? c->err : g_err) = msg; return code; }

extern "C" {
int lam_hip_abi_version(void) { return LAM_HIP_ABI_VERSION; }
const char *lam_hip_build_id(void) { return "fake"; }
int lam_hip_device_count(int *count) { *count = 1; return 0; }
const char *lam_hip_last_error(const lam_hip_ctx *c) { return c ? c->err.c_str() : g_err.c_str(); }
int lam_hip_create(lam_hip_ctx **out, int dtype, int n_shards, const int *)
{
    if (n_shards < 1 || n_shards > LAM_HIP_MAX_SHARDS) return fail(nullptr, LAM_HIP_EINVAL, "n_shards");
    auto *c = new lam_hip_ctx;
    c->dtype = dtype; c->total = c->local = n_shards;
    *out = c;
    return 0;
}
int lam_hip_get_unique_id(void *id)
{
    for (int i = 0; i < LAM_HIP_UNIQUE_ID_BYTES; i++) static_cast<unsigned char *>(id)[i] = (unsigned char)(0xA5 ^ (i * 7));
    return 0;
}
int lam_hip_create_rank(lam_hip_ctx **out, int dtype, int, int rank, int nranks, const void *)
{
    if (nranks < 1 || nranks > LAM_HIP_MAX_SHARDS || rank < 0 || rank >= nranks) return fail(nullptr, LAM_HIP_EINVAL, "rank");
    auto *c = new lam_hip_ctx;
    c->dtype = dtype; c->total = nranks; c->local = 1; c->rank = rank;
    *out = c;
    return 0;
}
void lam_hip_destroy(lam_hip_ctx *c) { delete c; }
int lam_hip_set_problem(lam_hip_ctx *c, uint64_t n)
{
    if (n == 0 || n < (uint64_t)c->total) return fail(c, LAM_HIP_EINVAL, "n");
    if (n > (1u << 16)) return fail(c, LAM_HIP_ENOMEM, "the fake refuses matrices beyond 65536 rows");
    c->n = n;
    if (c->local == c->total) { c->row0 = 0; c->nrows = n; }
    else lam::partition_rows(n, c->total, c->rank, &c->row0, &c->nrows);
    c->A.assign(c->nrows * n * c->es(), 0);            // EXACTLY the owned rows
    c->b.assign(n * c->es(), 0);
    c->row_uploaded.assign(c->nrows, 0);
    return 0;
}
int lam_hip_partition(uint64_t n, int P, int q, uint64_t *row0, uint64_t *nrows) { lam::partition_rows(n, P, q, row0, nrows); return 0; }
int lam_hip_num_shards(const lam_hip_ctx *c, int *total, int *local) { if (total) *total = c->total; if (local) *local = c->local; return 0; }
int lam_hip_get_partition(const lam_hip_ctx *c, int shard, uint64_t *row0, uint64_t *nrows)
{
    if (shard < 0 || shard >= c->total || c->n == 0) return LAM_HIP_EINVAL;
    lam::partition_rows(c->n, c->total, shard, row0, nrows);
    return 0;
}
int lam_hip_upload_rows(lam_hip_ctx *c, uint64_t row0, uint64_t nrows, const void *host)
{
    if (row0 < c->row0 || row0 + nrows > c->row0 + c->nrows) return fail(c, LAM_HIP_EINVAL, "rows not owned by this process");
    if (nrows * c->n * c->es() > (uint64_t)(LAM_LOADER_CHUNK_BYTES_MAX)) return fail(c, LAM_HIP_EINVAL, "one upload exceeds the chunk limit");
    memcpy(c->A.data() + (row0 - c->row0) * c->n * c->es(), host, nrows * c->n * c->es());      // reads every byte the caller promised
    for (uint64_t r = row0; r < row0 + nrows; r++) {
        if (c->row_uploaded[r - c->row0]) return fail(c, LAM_HIP_EINVAL, "a row was uploaded twice");
        c->row_uploaded[r - c->row0] = 1;
    }
    return 0;
}
int lam_hip_set_rhs(lam_hip_ctx *c, const void *b) { memcpy(c->b.data(), b, c->n * c->es()); return 0; }
int lam_hip_get_solution(lam_hip_ctx *c, void *x) { memcpy(x, c->b.data(), c->n * c->es()); return 0; }      // "x = b": something to save
int lam_hip_all_ok(lam_hip_ctx *, int local_ok, int *global_ok) { *global_ok = local_ok ? 1 : 0; return 0; }
int lam_hip_generate_tridiag(lam_hip_ctx *) { return 0; }
int lam_hip_generate_random_spd(lam_hip_ctx *, uint64_t, double) { return 0; }
int lam_hip_generate_random_rhs(lam_hip_ctx *, uint64_t) { return 0; }
int lam_hip_generate_rhs(lam_hip_ctx *, double) { return 0; }
int lam_hip_generate_spectrum_spd(lam_hip_ctx *c, const double *eig, const double *v, int k)
{
    double acc = 0.0;                                   // touch everything the caller promised: n eigenvalues, k x n reflector entries
    for (uint64_t i = 0; i < c->n; i++) acc += eig[i];
    for (uint64_t i = 0; i < (uint64_t)k * c->n; i++) acc += v[i];
    return acc == acc ? 0 : LAM_HIP_EINVAL;
}
int lam_hip_solve(lam_hip_ctx *, int, double, lam_hip_stats *st) { memset(st, 0, sizeof *st); st->converged = 1; st->num_iters = 1; return 0; }

// test accessors (not part of the ABI)
const void *fake_matrix(const lam_hip_ctx *c) { return c->A.data(); }
const void *fake_rhs(const lam_hip_ctx *c) { return c->b.data(); }
uint64_t fake_row0(const lam_hip_ctx *c) { return c->row0; }
uint64_t fake_nrows(const lam_hip_ctx *c) { return c->nrows; }
int fake_all_rows_uploaded(const lam_hip_ctx *c)
{
    for (unsigned char u : c->row_uploaded) if (!u) return 0;
    return 1;
}
}
