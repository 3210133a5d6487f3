// symv2_probe -- research probe, NOT part of the product: drives the PRODUCT's symmetric-product kernels (csrc/lam_kernels.h, included below)
// with its own task schedules, several plans timed INTERLEAVED in one process (the device's state and the placement of a plan's buffers move
// this kernel by several percent; only interleaved numbers compare), against a plain full-matrix product on the same data.
// The round-1 shape (its probe: git history, tools/symv_probe.hip; numbers: profiles/r01_symmetric_option.txt) flushed one column partial per
// 32 rows (0.55 GB written and read again per product at N=65536) and kept 32 row partials per lane; what this probe found on the way to the
// current shape is on file in profiles/r04_symv2_probe.txt.
//
// usage: symv2_probe.out N f64|f32 rounds SPEC ...        SPEC = [KiB*100+]NV:h1@f1,h2@f2,...,hlast
//        NV = 16-byte vectors per lane and row (1 | 2), KiB = idle dynamic LDS per workgroup of the first pass (caps the workgroups per CU),
//        schedule: tasks of h1 rows up to row f1 * N, h2 up to f2 * N, ..., hlast for the rest (heights: multiples of 8, at most 2048)
// environment: SYMV2_SHARD=q/P  one row shard of a P-way split (cyclic half windows)      SYMV2_READONLY=1  also time the tasks' loads alone
//              SYMV2_REVERSE=1  time the plans in reverse order      SYMV2_PAD_MB=n  an idle allocation between the matrix and the plans' buffers
//              SYMV2_GEN=1  a matrix with a large diagonal           SYMV2_STREAM=1  a non-blocking stream        SYMV2_NO_FULL=1  rim tasks on the per-element path
// build: hipcc --offload-arch=gfx950 -O3 tools/symv2_probe.hip -o tools/symv2_probe.out   (-DLAM_SYMV_PROBE_NO_PARTIAL_STORES: the partial stores compiled out)
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../2024-eumaster4hpc-student-challenge_amd/csrc/lam_kernels.h"

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

using lam::kBlock;
static hipStream_t g_stream = nullptr;
static uint64_t g_shard_q = 0, g_shard_P = 0;   // SYMV2_SHARD=q/P: time ONE row shard of a P-way split (cyclic half windows, the library's multi-shard form)     // SYMV2_STREAM=1: a non-blocking stream of its own (the library's way)

template <typename T> struct Vec;
template <> struct Vec<double> { typedef double t __attribute__((ext_vector_type(2))); static constexpr int N = 2; };
template <> struct Vec<float> { typedef float t __attribute__((ext_vector_type(4))); static constexpr int N = 4; };

__host__ __device__ inline uint64_t mix(uint64_t z)
{
    z += 0x9E3779B97F4A7C15ull; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
template <typename T>
__global__ void gen_sym(T *A, uint64_t n, uint64_t lda, int big_diag)
{
    const uint64_t total = n * lda;
    for (uint64_t idx = (uint64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (uint64_t)gridDim.x * 256) {
        const uint64_t i = idx / lda, j = idx % lda, lo = i < j ? i : j, hi = i < j ? j : i;
        const double u = (double)(mix(lo * n + hi) >> 11) * (1.0 / 9007199254740992.0);
        A[idx] = j >= n ? (T)0 : (T)(i == j ? (big_diag ? 1.0 + 999999.0 * u : 2.0 + u) : (2.0 * u - 1.0) / (double)n);
    }
}
template <typename T>
__global__ void gen_vec(T *p, uint64_t n, uint64_t npad)
{
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < npad; i += (uint64_t)gridDim.x * 256)
        p[i] = i < n ? (T)(2.0 * ((double)(mix(i ^ 0xABCDEFull) >> 11) * (1.0 / 9007199254740992.0)) - 1.0) : (T)0;
}

// plain full-matrix product for the check and the comparison
template <typename T>
__global__ void __launch_bounds__(256) gemv_full(const T *__restrict__ A, const T *__restrict__ p, T *__restrict__ y, uint64_t n, uint64_t lda)
{
    using V = typename Vec<T>::t;
    constexpr int VEC = Vec<T>::N;
    __shared__ T s_part[2][4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint64_t row0 = (uint64_t)blockIdx.x * 2;
    T acc[2] = {0, 0};
    for (uint64_t c = ((uint64_t)wave * 64 + lane) * VEC; c < n; c += 256 * VEC) {
        const V pv = *reinterpret_cast<const V *>(p + c);
#pragma unroll
        for (int r = 0; r < 2; r++) {
            if (row0 + r >= n) continue;
            const V a = __builtin_nontemporal_load(reinterpret_cast<const V *>(A + (row0 + r) * lda + c));
#pragma unroll
            for (int i = 0; i < VEC; i++) acc[r] += a[i] * pv[i];
        }
    }
#pragma unroll
    for (int r = 0; r < 2; r++) {
        T s = acc[r];
        for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
        if (lane == 0) s_part[r][wave] = s;
    }
    __syncthreads();
    if (threadIdx.x < 2 && row0 + threadIdx.x < n)
        y[row0 + threadIdx.x] = (s_part[threadIdx.x][0] + s_part[threadIdx.x][1]) + (s_part[threadIdx.x][2] + s_part[threadIdx.x][3]);
}

// ---- the product's kernels (lam::symv_task_kernel / lam::symv_reduce_kernel), driven with this probe's own task schedules ------------
using lam::SymvTask;
typedef SymvTask Task;
constexpr int kRowsMax = lam::kSymvRowsMax;

// the same tasks, the same loads, no arithmetic to speak of: what the ACCESS PATTERN alone reaches (SYMV2_READONLY=1 adds it to the table)
template <typename T, int NV>
__global__ void __launch_bounds__(kBlock)
pattern_read_kernel(const T *__restrict__ A, const Task *__restrict__ tasks, T *__restrict__ sink, uint64_t lda)
{
    using V = typename Vec<T>::t;
    constexpr int VEC = Vec<T>::N, CW = kBlock * VEC, SS = NV * CW;
    Task t = tasks[blockIdx.x];
    t.nrows &= ~lam::kSymvFlags;
    const T *rows = A + (uint64_t)t.row0 * lda + (uint64_t)t.strip * SS;
    T acc = (T)0;
    for (uint32_t b = 0; b + 8 <= t.nrows; b += 8, rows += 8 * lda) {
        V a[8][NV];
#pragma unroll
        for (int k = 0; k < 8; k++)
#pragma unroll
            for (int v = 0; v < NV; v++)
                a[k][v] = __builtin_nontemporal_load(reinterpret_cast<const V *>(rows + (uint64_t)k * lda + v * CW) + threadIdx.x);
#pragma unroll
        for (int k = 0; k < 8; k++)
#pragma unroll
            for (int v = 0; v < NV; v++) acc += a[k][v][0];
    }
    if (acc == (T)123.456) sink[0] = acc;
}

template <typename F>
double time_once_ms(F f, int reps)
{
    static hipEvent_t e0 = nullptr, e1 = nullptr;
    if (!e0) { (void)hipEventCreate(&e0); (void)hipEventCreate(&e1); }
    (void)hipEventRecord(e0, g_stream);
    for (int i = 0; i < reps; i++) f();
    (void)hipEventRecord(e1, g_stream);
    (void)hipEventSynchronize(e1);
    float ms = 0.f; (void)hipEventElapsedTime(&ms, e0, e1);
    return ms / reps;
}

// a plan = kernel shape + task schedule; all plans share the matrix and are timed in turn, `rounds` times (the device's clock and
// memory states drift by several percent within seconds: only interleaved measurements compare)
template <typename T>
struct Plan {
    std::string name;
    int NV;
    unsigned lds_extra = 0;      // bytes of idle dynamic LDS per workgroup of the first pass: caps the workgroups per CU (spec NV = KiB * 100 + NV)
    std::vector<std::pair<int, double>> sched;     // task height, up to which row fraction
    uint32_t nstrips = 0, ntasks = 0;
    T *rowpart = nullptr, *colpart = nullptr;
    Task *dt = nullptr;
    uint32_t *dsb = nullptr;      // the second pass's index (lam::SymvIndex)
    lam::SymvIndex ix = {};
    std::vector<double> t1, t12;
    double err = 0;
};

template <typename T>
bool pass1(const Plan<T> &pl, const T *A, const T *p, uint64_t n, uint64_t lda, uint64_t ncols_vec, uint64_t row_pitch)
{
    if (g_shard_P > 0) {
        const uint64_t R0 = g_shard_q * (n / g_shard_P);
        if (pl.NV == 1) hipLaunchKernelGGL((lam::symv_task_kernel<T, T, 1, true>), dim3(pl.ntasks), dim3(kBlock), 0, g_stream, A + R0 * lda, p, pl.dt, pl.rowpart, pl.colpart, lda, ncols_vec, n, R0, (const lam::CgScalars *)nullptr);
        else hipLaunchKernelGGL((lam::symv_task_kernel<T, T, 2, true>), dim3(pl.ntasks), dim3(kBlock), 0, g_stream, A + R0 * lda, p, pl.dt, pl.rowpart, pl.colpart, lda, ncols_vec, n, R0, (const lam::CgScalars *)nullptr);
        return true;
    }
    if (pl.NV == 1) hipLaunchKernelGGL((lam::symv_task_kernel<T, T, 1, false>), dim3(pl.ntasks), dim3(kBlock), pl.lds_extra, g_stream, A, p, pl.dt, pl.rowpart, pl.colpart, lda, ncols_vec, n, (uint64_t)0, (const lam::CgScalars *)nullptr);
    else if (pl.NV == 2) hipLaunchKernelGGL((lam::symv_task_kernel<T, T, 2, false>), dim3(pl.ntasks), dim3(kBlock), pl.lds_extra, g_stream, A, p, pl.dt, pl.rowpart, pl.colpart, lda, ncols_vec, n, (uint64_t)0, (const lam::CgScalars *)nullptr);
    else return false;
    return true;
}
template <typename T>
bool pass2(const Plan<T> &pl, const T *p, T *y, double *partial, uint64_t n, uint64_t row_pitch)
{
    const unsigned grid = (unsigned)((n + lam::kSymvReduceRows - 1) / lam::kSymvReduceRows);
    lam::PtrList none;
    none.n = 0;
    lam::Finalize nofin;
    nofin.active = 0; nofin.mail = 0; nofin.seq = 0; nofin.dst.n = 0; nofin.slot = 0; nofin.host_err = nullptr;
    if (g_shard_P > 0) {
        const uint64_t nloc = n / g_shard_P, R0 = g_shard_q * nloc;
        if (pl.NV == 1) hipLaunchKernelGGL((lam::symv_reduce_kernel<T, 1 * lam::kBlock * Vec<T>::N>), dim3(grid), dim3(kBlock), 0, g_stream, pl.rowpart, pl.colpart, pl.dsb, pl.ix, p, y, partial, n, R0, nloc, none, nofin, (const lam::CgScalars *)nullptr);
        else hipLaunchKernelGGL((lam::symv_reduce_kernel<T, 2 * lam::kBlock * Vec<T>::N>), dim3(grid), dim3(kBlock), 0, g_stream, pl.rowpart, pl.colpart, pl.dsb, pl.ix, p, y, partial, n, R0, nloc, none, nofin, (const lam::CgScalars *)nullptr);
        return true;
    }
    if (pl.NV == 1) hipLaunchKernelGGL((lam::symv_reduce_kernel<T, 1 * lam::kBlock * Vec<T>::N>), dim3(grid), dim3(kBlock), 0, g_stream, pl.rowpart, pl.colpart, pl.dsb, pl.ix, p, y, partial, n, (uint64_t)0, n, none, nofin, (const lam::CgScalars *)nullptr);
    else if (pl.NV == 2) hipLaunchKernelGGL((lam::symv_reduce_kernel<T, 2 * lam::kBlock * Vec<T>::N>), dim3(grid), dim3(kBlock), 0, g_stream, pl.rowpart, pl.colpart, pl.dsb, pl.ix, p, y, partial, n, (uint64_t)0, n, none, nofin, (const lam::CgScalars *)nullptr);
    else return false;
    return true;
}

template <typename T>
int run(uint64_t n, const std::vector<std::string> &specs, int rounds)
{
    constexpr int VEC = Vec<T>::N;
    const uint64_t align = n * sizeof(T) >= 4096 ? 4096 / sizeof(T) : 16 / sizeof(T);
    const uint64_t lda = (n + align - 1) / align * align, ncols_vec = (n + VEC - 1) / VEC * VEC;
    const uint64_t row_pitch = (n + 63) / 64 * 64, npad = (ncols_vec + 16383) / 16384 * 16384;
    T *A, *p, *y, *yref; double *partial;
    CHK(hipMalloc(&A, n * lda * sizeof(T))); CHK(hipMalloc(&p, npad * sizeof(T))); CHK(hipMalloc(&y, n * sizeof(T)));
    CHK(hipMalloc(&yref, n * sizeof(T))); CHK(hipMalloc(&partial, ((n + 31) / 32) * 8));
    hipLaunchKernelGGL(gen_sym<T>, dim3(8192), dim3(256), 0, 0, A, n, lda, getenv("SYMV2_GEN") ? 1 : 0);
    hipLaunchKernelGGL(gen_vec<T>, dim3(256), dim3(256), 0, 0, p, n, npad);
    auto full = [&] { hipLaunchKernelGGL(gemv_full<T>, dim3((unsigned)((n + 1) / 2)), dim3(256), 0, g_stream, A, p, yref, n, lda); };
    full();
    CHK(hipDeviceSynchronize());
    std::vector<T> h(n), hr(n);
    CHK(hipMemcpy(hr.data(), yref, n * sizeof(T), hipMemcpyDeviceToHost));
    double maxref = 0;
    for (uint64_t i = 0; i < n; i++) maxref = std::max(maxref, std::fabs((double)hr[i]));

    if (const char *pad = getenv("SYMV2_PAD_MB")) {        // an idle allocation between the matrix and the plans' buffers (placement experiments)
        void *dummy = nullptr;
        CHK(hipMalloc(&dummy, (size_t)atol(pad) << 20));
    }
    std::vector<Plan<T>> plans;
    for (const auto &spec : specs) {       // NV:h1@f1,h2@f2,...,hlast
        Plan<T> pl;
        pl.name = spec;
        char sched[256] = "";
        if (sscanf(spec.c_str(), "%d:%255s", &pl.NV, sched) != 2) { printf("bad spec %s\n", spec.c_str()); return 1; }
        pl.lds_extra = (unsigned)(pl.NV / 100) * 1024u;
        pl.NV %= 100;
        for (char *tok = strtok(sched, ","); tok; tok = strtok(nullptr, ",")) {
            int hgt = 0; double f = 2.0;
            if (sscanf(tok, "%d@%lf", &hgt, &f) < 1 || hgt % 8 || hgt > kRowsMax || hgt <= 0) { printf("bad schedule %s\n", tok); return 1; }
            pl.sched.push_back({hgt, f});
        }
        const uint64_t SS = (uint64_t)pl.NV * kBlock * VEC;
        pl.nstrips = (uint32_t)((ncols_vec + SS - 1) / SS);
        // tasks in dispatch order -- row run by row run, the strips of a run side by side -- and the second pass's index, as the
        // library's planner builds them (lam_launch.h, symv_plan), but with this probe's schedules of task heights
        std::vector<Task> tasks;
        std::vector<uint32_t> runs, index;
        uint64_t rp_elems = 0;
        const bool cyc = g_shard_P > 0;
        const uint64_t nloc = cyc ? n / g_shard_P : n, R0 = cyc ? g_shard_q * nloc : 0, H = (n - 1) / 2;
        std::vector<uint32_t> row8((nloc + 7) / 8, 0);
        std::vector<std::vector<uint32_t>> per_strip(pl.nstrips);
        std::vector<uint64_t> upto;
        for (auto &c : pl.sched) upto.push_back(c.second >= 1.0 ? nloc : (uint64_t)(c.second * (double)nloc) / pl.sched[0].first * pl.sched[0].first);
        auto meets = [](uint64_t a0, uint64_t a1, uint64_t b0, uint64_t b1) { return a0 <= b1 && b0 <= a1; };
        for (uint64_t r = 0; r < nloc;) {
            size_t cls = 0;
            while (cls + 1 < pl.sched.size() && r >= upto[cls]) cls++;
            const uint64_t hgt = std::min<uint64_t>(nloc - r, (uint64_t)pl.sched[cls].first), ga = R0 + r, gb = ga + hgt;
            const uint32_t run = (uint32_t)(runs.size() / 5), first = (uint32_t)tasks.size();
            for (uint32_t st = 0; st < pl.nstrips; st++) {
                const uint64_t c0 = (uint64_t)st * SS, c1 = std::min<uint64_t>(c0 + SS, n) - 1;
                const bool full = c0 + SS <= n && hgt % 8 == 0;
                bool needed, interior = full;
                if (!cyc) { needed = c1 >= ga; interior = interior && c0 >= gb; }
                else {
                    needed = meets(c0, c1, ga, gb - 1 + n / 2) || meets(c0 + n, c1 + n, ga, gb - 1 + n / 2);
                    bool in = false;
                    for (uint64_t k2 = 0; k2 < 2; k2++) {
                        const uint64_t u0 = c0 + k2 * n, u1 = c0 + SS - 1 + k2 * n;
                        in = in || (u0 >= gb && u1 - ga <= H);
                    }
                    interior = interior && in;
                }
                if (!needed) continue;
                per_strip[st].push_back((uint32_t)tasks.size());
                tasks.push_back({(uint32_t)r, (uint32_t)hgt | (interior ? lam::kSymvInterior : 0u) | (full && !getenv("SYMV2_NO_FULL") ? lam::kSymvFull : 0u), st, (uint32_t)rp_elems});
                rp_elems += hgt;
            }
            runs.insert(runs.end(), {first, (uint32_t)tasks.size() - first, (uint32_t)r, (uint32_t)hgt, tasks.size() > first ? tasks[first].rp : 0u});
            for (uint64_t q = r / 8; q < (r + hgt + 7) / 8; q++) row8[q] = run;
            r += hgt;
        }
        pl.ix.runs = 0;
        index = runs;
        pl.ix.row8 = (uint32_t)index.size();
        index.insert(index.end(), row8.begin(), row8.end());
        pl.ix.strip_base = (uint32_t)index.size();
        uint32_t accn = 0;
        for (uint32_t st = 0; st < pl.nstrips; st++) { index.push_back(accn); accn += (uint32_t)per_strip[st].size(); }
        index.push_back(accn);
        pl.ix.strip_tasks = (uint32_t)index.size();
        for (uint32_t st = 0; st < pl.nstrips; st++) index.insert(index.end(), per_strip[st].begin(), per_strip[st].end());
        pl.ntasks = (uint32_t)tasks.size();
        CHK(hipMalloc(&pl.rowpart, (size_t)rp_elems * sizeof(T))); CHK(hipMalloc(&pl.colpart, (size_t)pl.ntasks * SS * sizeof(T)));
        CHK(hipMalloc(&pl.dt, tasks.size() * sizeof(Task))); CHK(hipMalloc(&pl.dsb, index.size() * 4));
        CHK(hipMemcpy(pl.dt, tasks.data(), tasks.size() * sizeof(Task), hipMemcpyHostToDevice));
        CHK(hipMemcpy(pl.dsb, index.data(), index.size() * 4, hipMemcpyHostToDevice));
        CHK(hipMemset(pl.colpart, 0xff, (size_t)pl.ntasks * SS * sizeof(T)));     // NaNs: every partial the second pass reads must have been written
        CHK(hipMemset(pl.rowpart, 0xff, (size_t)rp_elems * sizeof(T)));
        if (!pass1(pl, A, p, n, lda, ncols_vec, row_pitch) || !pass2(pl, p, y, partial, n, row_pitch)) { printf("shape of %s not compiled in\n", spec.c_str()); return 1; }
        CHK(hipDeviceSynchronize());
        CHK(hipMemcpy(h.data(), y, n * sizeof(T), hipMemcpyDeviceToHost));
        for (uint64_t i = 0; i < n; i++) {
            if (!(h[i] == h[i])) pl.err = 1e300;
            if (g_shard_P == 0) pl.err = std::max(pl.err, std::fabs((double)h[i] - (double)hr[i]) / maxref);   // a shard's contribution is not the product
        }
        plans.push_back(pl);
    }
    std::vector<double> tf;
    for (int r = 0; r < rounds; r++) {
        tf.push_back(time_once_ms(full, 5));
        if (getenv("SYMV2_REVERSE")) std::reverse(plans.begin(), plans.end());     // time the plans in the opposite order (allocation order stays)
        for (auto &pl : plans) {
            pl.t1.push_back(time_once_ms([&] { pass1(pl, A, p, n, lda, ncols_vec, row_pitch); }, 5));
            pl.t12.push_back(time_once_ms([&] { pass1(pl, A, p, n, lda, ncols_vec, row_pitch); pass2(pl, p, y, partial, n, row_pitch); }, 5));
        }
    }
    if (getenv("SYMV2_READONLY")) {
        // ... at several occupancies: dynamic LDS the kernel does not use caps the workgroups per CU (160 KiB of LDS: 0 -> the register
        // limit, 32 KiB -> 5 workgroups = 5 waves per SIMD, 40 -> 4, 53 -> 3, 80 -> 2)
        for (auto &pl : plans) {
            for (unsigned lds_kib : {0u, 32u, 40u, 53u, 80u}) {
                std::vector<double> t;
                auto f = [&] {
                    if (pl.NV == 1) hipLaunchKernelGGL((pattern_read_kernel<T, 1>), dim3(pl.ntasks), dim3(kBlock), lds_kib * 1024, g_stream, A, pl.dt, y, lda);
                    else hipLaunchKernelGGL((pattern_read_kernel<T, 2>), dim3(pl.ntasks), dim3(kBlock), lds_kib * 1024, g_stream, A, pl.dt, y, lda);
                };
                for (int r = 0; r < rounds; r++) t.push_back(time_once_ms(f, 5));
                std::sort(t.begin(), t.end());
                printf("%-44s  the tasks' loads alone (no arithmetic), %2u KiB of idle LDS per workgroup: min %6.3f med %6.3f ms\n", pl.name.c_str(), lds_kib, t[0], t[t.size() / 2]);
            }
        }
    }
    if (getenv("SYMV2_REVERSE") && rounds % 2) std::reverse(plans.begin(), plans.end());
    auto stat = [](std::vector<double> v, double *mn, double *med) { std::sort(v.begin(), v.end()); *mn = v[0]; *med = v[v.size() / 2]; };
    const double gb_full = (double)sizeof(T) * n * n / 1e9, gb_tri = (gb_full / 2 + (double)sizeof(T) * n / 2 / 1e9) / (g_shard_P > 0 ? (double)g_shard_P : 1.0);
    if (g_shard_P > 0) printf("shard %llu of %llu: rows [%llu, %llu), cyclic half windows; its bytes = %.3f GB\n", (unsigned long long)g_shard_q, (unsigned long long)g_shard_P,
                              (unsigned long long)(g_shard_q * (n / g_shard_P)), (unsigned long long)((g_shard_q + 1) * (n / g_shard_P)), gb_tri);
    double mn, med;
    stat(tf, &mn, &med);
    printf("N=%llu %s lda=%llu, %d interleaved rounds of 5 launches; triangle = %.2f GB\n", (unsigned long long)n, sizeof(T) == 8 ? "fp64" : "fp32",
           (unsigned long long)lda, rounds, gb_tri);
    printf("%-44s  min %7.3f ms  median %7.3f ms  %7.1f GB/s (median, whole matrix)\n", "full product (simple 2-row kernel)", mn, med, gb_full / med * 1e3);
    const double med_full = med;
    int rc = 0;
    for (auto &pl : plans) {
        double mn1, med1, mn12, med12;
        stat(pl.t1, &mn1, &med1); stat(pl.t12, &mn12, &med12);
        printf("%-44s  tasks %6u  pass1 min %6.3f med %6.3f | 1+2 min %6.3f med %6.3f ms = %6.1f GB/s on the triangle, %.3fx full | err %.1e\n", pl.name.c_str(),
               pl.ntasks, mn1, med1, mn12, med12, gb_tri / med12 * 1e3, med_full / med12, pl.err);
        if (!(pl.err < (sizeof(T) == 8 ? 1e-12 : 1e-4))) rc = 2;
    }
    fflush(stdout);
    return rc;
}

int main(int argc, char **argv)
{
    if (argc < 4) { printf("usage: symv2_probe.out N f64|f32 rounds NV:h1@f1,h2@f2,...,hlast ...\n"); return 1; }
    const uint64_t n = strtoull(argv[1], nullptr, 10);
    const bool f32 = !strcmp(argv[2], "f32");
    if (getenv("SYMV2_STREAM")) CHK(hipStreamCreateWithFlags(&g_stream, hipStreamNonBlocking));
    (void)hipFuncSetAttribute((const void *)lam::symv_task_kernel<double, double, 1, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
    (void)hipFuncSetAttribute((const void *)lam::symv_task_kernel<double, double, 2, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
    (void)hipFuncSetAttribute((const void *)lam::symv_task_kernel<float, float, 1, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
    (void)hipFuncSetAttribute((const void *)lam::symv_task_kernel<float, float, 2, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
    (void)hipFuncSetAttribute((const void *)pattern_read_kernel<double, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
    (void)hipFuncSetAttribute((const void *)pattern_read_kernel<double, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
    (void)hipFuncSetAttribute((const void *)pattern_read_kernel<float, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
    (void)hipFuncSetAttribute((const void *)pattern_read_kernel<float, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
    if (const char *sh = getenv("SYMV2_SHARD")) { unsigned long long q = 0, P = 0; if (sscanf(sh, "%llu/%llu", &q, &P) == 2 && P > 0 && q < P && n % P == 0) { g_shard_q = q; g_shard_P = P; } }
    const int rounds = atoi(argv[3]);
    std::vector<std::string> specs;
    for (int i = 4; i < argc; i++) specs.push_back(argv[i]);
    return f32 ? run<float>(n, specs, rounds) : run<double>(n, specs, rounds);
}
