// ksplit_probe -- what would a K-split GEMV buy at small N?  (DESIGN.md "What comes next" (2); tools only, nothing of the product.)
// The production GEMV gives every workgroup whole rows: at N = 10000 that is 5000 workgroups of 160 KB in ~5 rounds over the chip, each
// re-reading the p tiles from L2, and the launch's fixed ramp + drain (~10 us of 123 us) is what keeps it at 0.82 of peak.  The K-split
// form cuts the columns into `ntiles` tiles: workgroup (row block, tile) keeps ONE p tile in LDS and streams `rows_per_wg` row segments
// through it, writing one partial per (tile, row); the consumer (the fused vector step) would add the ntiles partials of a row in a fixed
// order -- no extra launch, but other bits than today's row sums.  This probe times that first pass (with its p.y partial per workgroup,
// as the production kernel has) against the library's own GEMV (lam_hip_gemv_only through the C ABI) on the same device, interleaved.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude tools/ksplit_probe.hip -o tools/ksplit_probe.out \
//              -L2024-eumaster4hpc-student-challenge_amd -llam_hip -Wl,-rpath,$PWD/2024-eumaster4hpc-student-challenge_amd -Wl,-rpath,/opt/rocm/lib
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "lam_hip.h"

typedef double d2 __attribute__((ext_vector_type(2)));
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void fill(double *A, size_t lda, size_t n, double *p, size_t plen)
{
    const size_t total = lda * n;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t c = i % lda;
        unsigned long long h = i * 0x9E3779B97F4A7C15ull;
        h ^= h >> 29;
        A[i] = c < n ? (double)(h & 0xffff) / 65536.0 - 0.5 : 0.0;
    }
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < plen; i += (size_t)gridDim.x * blockDim.x)
        p[i] = i < n ? 1.0 + (double)(i % 7) * 0.125 : 0.0;
}

__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// grid (row blocks, tiles); part[tile * npad + row]; dotpart[blockIdx.y * gridDim.x + blockIdx.x]
template <int R, int WAVES, int TILE>
__global__ void __launch_bounds__(WAVES * 64)
ksplit(const double *__restrict__ A, size_t lda, const double *__restrict__ p, double *__restrict__ part, double *__restrict__ dotpart,
       size_t n, size_t npad, int rows_per_wg, int tile_w)
{
    __shared__ __attribute__((aligned(16))) double s_p[TILE];
    __shared__ double s_dot[WAVES];
    const int tid = threadIdx.x, wave = tid / 64, lane = tid % 64;
    const size_t c0 = (size_t)blockIdx.y * tile_w;
    for (int i = tid * 2; i < tile_w; i += WAVES * 128) *reinterpret_cast<d2 *>(&s_p[i]) = *reinterpret_cast<const d2 *>(&p[c0 + i]);
    __syncthreads();
    const size_t row_base = (size_t)blockIdx.x * rows_per_wg;
    const int steps = tile_w / 128;
    double dot = 0.0;
    for (int rg = wave * R; rg < rows_per_wg; rg += WAVES * R) {
        const d2 *rp[R];
        double acc[R];
#pragma unroll
        for (int j = 0; j < R; j++) {
            const size_t r = std::min(row_base + rg + j, n - 1);
            rp[j] = reinterpret_cast<const d2 *>(A + r * lda + c0) + lane;
            acc[j] = 0.0;
        }
#pragma unroll 4
        for (int s = 0; s < steps; s++) {
            d2 a[R];
#pragma unroll
            for (int j = 0; j < R; j++) a[j] = __builtin_nontemporal_load(rp[j] + (size_t)s * 64);
            const d2 pv = *reinterpret_cast<const d2 *>(&s_p[s * 128 + lane * 2]);
#pragma unroll
            for (int j = 0; j < R; j++) acc[j] = __builtin_fma(a[j][0], pv[0], __builtin_fma(a[j][1], pv[1], acc[j]));
        }
#pragma unroll
        for (int j = 0; j < R; j++) {
            const double t = wave_sum(acc[j]);
            const size_t r = row_base + rg + j;
            if (lane == 0 && r < n) {
                part[(size_t)blockIdx.y * npad + r] = t;
                dot += t * p[r];
            }
        }
    }
    if (lane == 0) s_dot[wave] = dot;
    __syncthreads();
    if (tid == 0) {
        double t = 0.0;
        for (int w = 0; w < WAVES; w++) t += s_dot[w];
        dotpart[(size_t)blockIdx.y * gridDim.x + blockIdx.x] = t;
    }
}

__global__ void sum_tiles(const double *part, size_t npad, int ntiles, size_t n, double *y)
{
    const size_t r = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    double t = part[r];
    for (int k = 1; k < ntiles; k++) t += part[(size_t)k * npad + r];
    y[r] = t;
}

template <typename F>
double time_us(F launch, int reps)
{
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    launch();
    std::vector<float> ts;
    for (int r = 0; r < 7; r++) {
        CHK(hipEventRecord(e0));
        for (int i = 0; i < reps; i++) launch();
        CHK(hipEventRecord(e1));
        CHK(hipEventSynchronize(e1));
        float ms = 0.f;
        CHK(hipEventElapsedTime(&ms, e0, e1));
        ts.push_back(ms / reps);
    }
    std::sort(ts.begin(), ts.end());
    CHK(hipEventDestroy(e0)); CHK(hipEventDestroy(e1));
    return ts[ts.size() / 2] * 1e3;
}

template <int R, int WAVES>
double run_shape(const double *A, size_t lda, const double *p, double *part, double *dotpart, size_t n, size_t npad, int ntiles, int tile_w, int rows_per_wg,
                 int reps)
{
    const dim3 grid((unsigned)((n + rows_per_wg - 1) / rows_per_wg), (unsigned)ntiles);
    return time_us([&] { hipLaunchKernelGGL((ksplit<R, WAVES, 4096>), grid, dim3(WAVES * 64), 0, 0, A, lda, p, part, dotpart, n, npad, rows_per_wg, tile_w); }, reps);
}

int main(int argc, char **argv)
{
    std::vector<size_t> sizes;
    for (int i = 1; i < argc; i++) sizes.push_back((size_t)atoll(argv[i]));
    if (sizes.empty()) sizes = {8192, 10000, 12000, 16384, 20000, 32768};
    printf("# K-split GEMV probe (fp64): first pass of the K-split form (per-(tile,row) partials + p.y partial per workgroup) against the library's GEMV; "
           "us per launch, median of 7 x reps; frac = algorithmic bytes (8 N^2 + 16 N) / time / 8 TB/s\n");
    for (size_t n : sizes) {
        // the library's GEMV on a system of the same size (its own matrix: only the time matters)
        lam_hip_ctx *ctx = nullptr;
        if (lam_hip_create(&ctx, LAM_HIP_F64, 1, nullptr) != 0 || lam_hip_set_problem(ctx, n) != 0 || lam_hip_generate_random_spd(ctx, 5, 100.0) != 0 ||
            lam_hip_generate_random_rhs(ctx, 6) != 0 || lam_hip_cg_init(ctx) != 0) { printf("library set-up failed: %s\n", lam_hip_last_error(ctx)); return 1; }
        const int ntiles = (int)((n + 4095) / 4096);
        const int tile_w = (int)(((n + ntiles - 1) / ntiles + 127) / 128 * 128);
        const size_t lda = ((size_t)ntiles * tile_w + 511) / 512 * 512, npad = (n + 63) / 64 * 64;
        double *A, *p, *part, *dotpart, *y;
        CHK(hipMalloc(&A, lda * n * 8)); CHK(hipMalloc(&p, lda * 8)); CHK(hipMalloc(&part, (size_t)ntiles * npad * 8));
        CHK(hipMalloc(&dotpart, (size_t)ntiles * (n / 8 + 8) * 8)); CHK(hipMalloc(&y, npad * 8));
        hipLaunchKernelGGL(fill, dim3(4096), dim3(256), 0, 0, A, lda, n, p, lda);
        CHK(hipDeviceSynchronize());
        const double bytes = 8.0 * n * n + 16.0 * n;
        const int reps = n <= 12000 ? 200 : (n <= 20000 ? 100 : 40);
        // correctness of one shape against the host (small sizes only)
        if (n <= 10000) {
            hipLaunchKernelGGL((ksplit<4, 8, 4096>), dim3((unsigned)((n + 63) / 64), (unsigned)ntiles), dim3(512), 0, 0, A, lda, p, part, dotpart, n, npad, 64, tile_w);
            hipLaunchKernelGGL(sum_tiles, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, part, npad, ntiles, n, y);
            std::vector<double> hy(n), hA(lda), hp(lda);
            CHK(hipMemcpy(hy.data(), y, n * 8, hipMemcpyDeviceToHost));
            CHK(hipMemcpy(hp.data(), p, lda * 8, hipMemcpyDeviceToHost));
            double worst = 0.0;
            for (size_t r : {(size_t)0, n / 2, n - 1}) {
                CHK(hipMemcpy(hA.data(), A + r * lda, lda * 8, hipMemcpyDeviceToHost));
                double t = 0.0, s = 0.0;
                for (size_t c = 0; c < n; c++) { t += hA[c] * hp[c]; s += std::fabs(hA[c] * hp[c]); }
                worst = std::max(worst, std::fabs(t - hy[r]) / s);
            }
            printf("N=%zu: K-split result vs host on 3 rows: max |diff| / sum|terms| = %.2e\n", n, worst);
        }
        double lib_s = 0.0;
        auto lib = [&] { (void)lam_hip_gemv_only(ctx, reps, &lib_s); return lib_s * 1e6; };
        lib();
        struct Row { const char *name; double us; };
        std::vector<Row> rows;
        for (int round = 0; round < 2; round++) {          // interleaved: library, shapes, library, shapes
            rows.push_back({"library GEMV (production shape)", lib()});
            rows.push_back({"K-split R=4 W=8 rows/wg=32", run_shape<4, 8>(A, lda, p, part, dotpart, n, npad, ntiles, tile_w, 32, reps)});
            rows.push_back({"K-split R=4 W=8 rows/wg=64", run_shape<4, 8>(A, lda, p, part, dotpart, n, npad, ntiles, tile_w, 64, reps)});
            rows.push_back({"K-split R=4 W=8 rows/wg=128", run_shape<4, 8>(A, lda, p, part, dotpart, n, npad, ntiles, tile_w, 128, reps)});
            rows.push_back({"K-split R=2 W=8 rows/wg=32", run_shape<2, 8>(A, lda, p, part, dotpart, n, npad, ntiles, tile_w, 32, reps)});
            rows.push_back({"K-split R=2 W=8 rows/wg=64", run_shape<2, 8>(A, lda, p, part, dotpart, n, npad, ntiles, tile_w, 64, reps)});
            rows.push_back({"K-split R=4 W=4 rows/wg=32", run_shape<4, 4>(A, lda, p, part, dotpart, n, npad, ntiles, tile_w, 32, reps)});
            rows.push_back({"K-split R=4 W=4 rows/wg=64", run_shape<4, 4>(A, lda, p, part, dotpart, n, npad, ntiles, tile_w, 64, reps)});
            rows.push_back({"K-split R=8 W=4 rows/wg=64", run_shape<8, 4>(A, lda, p, part, dotpart, n, npad, ntiles, tile_w, 64, reps)});
        }
        printf("N=%zu (%d tiles of %d columns):\n", n, ntiles, tile_w);
        const size_t half = rows.size() / 2;
        for (size_t i = 0; i < half; i++) {
            const double us = std::min(rows[i].us, rows[i + half].us);
            printf("   %-34s %8.1f / %8.1f us   best %8.1f us = %.3f of peak\n", rows[i].name, rows[i].us, rows[i + half].us, us, bytes / (us * 1e-6) / 8e12);
        }
        fflush(stdout);
        CHK(hipFree(A)); CHK(hipFree(p)); CHK(hipFree(part)); CHK(hipFree(dotpart)); CHK(hipFree(y));
        lam_hip_destroy(ctx);
    }
    return 0;
}
