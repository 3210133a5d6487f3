// symv_probe -- research probe, NOT part of the product: how much faster could one CG iteration get on
// one MI355X if the matrix-vector product read only the upper triangle of the symmetric matrix?
// ("The lever that is left".)  Implements the two-pass scheme sketched there for
// fp64, checks it against a plain full-matrix product on the same data, and times both.
//
//   pass 1  symv_task_kernel   one workgroup per task (row block I of 32 rows, column tile j of 4096
//                              columns, j >= the tile holding the diagonal).  The 4 waves read 4 KiB
//                              contiguous per row per super-step (512 columns); each lane keeps
//                              32 row partials (its 2 columns x 32 rows) in registers across the whole
//                              tile and 2 column partials (A^T contribution) across the 32 rows, which it
//                              flushes once per super-step to colpart[I][c].  Row partials are reduced
//                              over the workgroup once per task -> rowpart[I][j][32].
//   pass 2  symv_reduce_kernel y[i] = sum_j rowpart[I(i)][j][i%32] + sum_{I' <= i/32} colpart[I'][i]
//                              in a fixed order (deterministic).
//
// Build: hipcc --offload-arch=gfx950 -O3 tools/symv_probe.hip -o tools/symv_probe.out ; run: symv_probe.out [N]
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef double d2 __attribute__((ext_vector_type(2)));

constexpr int RB = 32;        // rows per task
constexpr int TILE = 4096;    // columns per task
constexpr int SS = 512;       // columns per super-step (4 waves x 64 lanes x 2)

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__host__ __device__ inline uint64_t mix(uint64_t z)
{
    z += 0x9E3779B97F4A7C15ull; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

__global__ void gen_sym(double *A, uint64_t n)
{
    const uint64_t total = n * n;
    for (uint64_t idx = (uint64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (uint64_t)gridDim.x * 256) {
        const uint64_t i = idx / n, j = idx % n, lo = i < j ? i : j, hi = i < j ? j : i;
        const double u = (double)(mix(lo * n + hi) >> 11) * (1.0 / 9007199254740992.0);
        A[idx] = i == j ? 2.0 + u : (2.0 * u - 1.0) / (double)n;
    }
}
__global__ void gen_vec(double *p, uint64_t n)
{
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256)
        p[i] = 2.0 * ((double)(mix(i ^ 0xABCDEFull) >> 11) * (1.0 / 9007199254740992.0)) - 1.0;
}

// plain full-matrix product for the check (cooperative rows, same shape as the product kernel)
__global__ void __launch_bounds__(256) gemv_full(const double *__restrict__ A, const double *__restrict__ p, double *__restrict__ y, uint64_t n)
{
    __shared__ double s_part[2][4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint64_t row0 = (uint64_t)blockIdx.x * 2;
    double acc[2] = {0, 0};
    for (uint64_t c = (uint64_t)wave * 128 + lane * 2; c < n; c += SS) {
        const d2 pv = *reinterpret_cast<const d2 *>(p + c);
#pragma unroll
        for (int r = 0; r < 2; r++) {
            const d2 a = __builtin_nontemporal_load(reinterpret_cast<const d2 *>(A + (row0 + r) * n + c));
            acc[r] += a[0] * pv[0] + a[1] * pv[1];
        }
    }
#pragma unroll
    for (int r = 0; r < 2; r++) {
        double s = acc[r];
        for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
        if (lane == 0) s_part[r][wave] = s;
    }
    __syncthreads();
    if (threadIdx.x < 2) y[row0 + threadIdx.x] = (s_part[threadIdx.x][0] + s_part[threadIdx.x][1]) + (s_part[threadIdx.x][2] + s_part[threadIdx.x][3]);
}

struct Task { uint32_t I, j; };

__global__ void __launch_bounds__(256)
symv_task_kernel(const double *__restrict__ A, const double *__restrict__ p, const Task *__restrict__ tasks,
                 double *__restrict__ rowpart, double *__restrict__ colpart, uint64_t n, uint32_t ntiles)
{
    __shared__ double s_pr[RB];
    __shared__ double s_red[RB][4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const Task t = tasks[blockIdx.x];
    const uint64_t r0 = (uint64_t)t.I * RB;
    if (tid < RB) s_pr[tid] = p[r0 + tid];
    __syncthreads();
    const uint32_t cw = (uint32_t)wave * 128 + (uint32_t)lane * 2;
    double racc[RB];
#pragma unroll
    for (int r = 0; r < RB; r++) racc[r] = 0.0;

    const uint64_t tile0 = (uint64_t)t.j * TILE;
    const double *Arow = A + r0 * n;
    for (int ss = 0; ss < TILE / SS; ss++) {
        const uint64_t c_ss = tile0 + (uint64_t)ss * SS;
        if (c_ss + SS <= r0) continue;                         // entirely left of the row block (lower triangle)
        const uint64_t c = c_ss + cw;                          // this lane's two columns
        const d2 pc = *reinterpret_cast<const d2 *>(p + c);
        double cacc0 = 0.0, cacc1 = 0.0;
        const bool masked = c_ss < r0 + RB;                    // the super-step touches the diagonal block
#pragma unroll
        for (int sub = 0; sub < RB / 8; sub++) {
            d2 a[8];
#pragma unroll
            for (int k = 0; k < 8; k++)
                a[k] = __builtin_nontemporal_load(reinterpret_cast<const d2 *>(Arow + (uint64_t)(sub * 8 + k) * n + c));
#pragma unroll
            for (int k = 0; k < 8; k++) {
                const int r = sub * 8 + k;
                const double pr = s_pr[r];
                if (!masked) {
                    racc[r] += a[k][0] * pc[0] + a[k][1] * pc[1];
                    cacc0 += a[k][0] * pr;
                    cacc1 += a[k][1] * pr;
                } else {
                    const uint64_t row = r0 + r;                // element (row, col): col > row both, col == row once
                    if (c >= row) racc[r] += a[k][0] * pc[0];
                    if (c + 1 >= row) racc[r] += a[k][1] * pc[1];
                    if (c > row) cacc0 += a[k][0] * pr;
                    if (c + 1 > row) cacc1 += a[k][1] * pr;
                }
            }
        }
        d2 out; out[0] = cacc0; out[1] = cacc1;
        *reinterpret_cast<d2 *>(colpart + (uint64_t)t.I * n + c) = out;
    }
    // row partials: wave butterfly, then across the 4 waves in a fixed order
#pragma unroll
    for (int r = 0; r < RB; r++) {
        double s = racc[r];
        for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
        if (lane == 0) s_red[r][wave] = s;
    }
    __syncthreads();
    if (tid < RB)
        rowpart[((uint64_t)t.I * ntiles + t.j) * RB + tid] = (s_red[tid][0] + s_red[tid][1]) + (s_red[tid][2] + s_red[tid][3]);
}

// y[i] for 32 consecutive i per workgroup; thread (cx, iy): columns cx, partial over I' = iy, iy+8, ...
__global__ void __launch_bounds__(256)
symv_reduce_kernel(const double *__restrict__ rowpart, const double *__restrict__ colpart, double *__restrict__ y,
                   uint64_t n, uint32_t ntiles)
{
    __shared__ double s[8][32];
    const int cx = threadIdx.x & 31, iy = threadIdx.x >> 5;
    const uint64_t I = blockIdx.x, i = I * RB + cx;
    double acc = 0.0;
    for (uint64_t Ip = iy; Ip <= I; Ip += 8) acc += colpart[Ip * n + i];
    s[iy][cx] = acc;
    __syncthreads();
    if (iy == 0) {
        double t = 0.0;
        for (int k = 0; k < 8; k++) t += s[k][cx];
        const uint32_t jd = (uint32_t)((I * RB) / TILE);
        for (uint32_t j = jd; j < ntiles; j++) t += rowpart[(I * ntiles + j) * RB + cx];
        y[i] = t;
    }
}

template <typename F>
double time_ms(F f, int reps)
{
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    f();
    std::vector<float> ts;
    for (int r = 0; r < 5; r++) {
        (void)hipEventRecord(e0);
        for (int i = 0; i < reps; i++) f();
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms = 0.f; (void)hipEventElapsedTime(&ms, e0, e1);
        ts.push_back(ms / reps);
    }
    std::sort(ts.begin(), ts.end());
    return ts[2];
}

int main(int argc, char **argv)
{
    const uint64_t n = argc > 1 ? strtoull(argv[1], nullptr, 10) : 32768;
    if (n % TILE != 0) { printf("N must be a multiple of %d\n", TILE); return 1; }
    const uint32_t ntiles = (uint32_t)(n / TILE), nblk = (uint32_t)(n / RB);
    std::vector<Task> tasks;
    for (uint32_t I = 0; I < nblk; I++)
        for (uint32_t j = (uint32_t)(((uint64_t)I * RB) / TILE); j < ntiles; j++) tasks.push_back({I, j});
    double *A, *p, *y, *yref, *rowpart, *colpart;
    Task *dt;
    CHK(hipMalloc(&A, n * n * 8)); CHK(hipMalloc(&p, n * 8)); CHK(hipMalloc(&y, n * 8)); CHK(hipMalloc(&yref, n * 8));
    CHK(hipMalloc(&rowpart, (size_t)nblk * ntiles * RB * 8)); CHK(hipMalloc(&colpart, (size_t)nblk * n * 8));
    CHK(hipMalloc(&dt, tasks.size() * sizeof(Task)));
    CHK(hipMemcpy(dt, tasks.data(), tasks.size() * sizeof(Task), hipMemcpyHostToDevice));
    CHK(hipMemset(colpart, 0, (size_t)nblk * n * 8));
    hipLaunchKernelGGL(gen_sym, dim3(8192), dim3(256), 0, 0, A, n);
    hipLaunchKernelGGL(gen_vec, dim3(256), dim3(256), 0, 0, p, n);
    CHK(hipDeviceSynchronize());

    auto full = [&] { hipLaunchKernelGGL(gemv_full, dim3((unsigned)(n / 2)), dim3(256), 0, 0, A, p, yref, n); };
    auto symv = [&] {
        hipLaunchKernelGGL(symv_task_kernel, dim3((unsigned)tasks.size()), dim3(256), 0, 0, A, p, dt, rowpart, colpart, n, ntiles);
        hipLaunchKernelGGL(symv_reduce_kernel, dim3(nblk), dim3(256), 0, 0, rowpart, colpart, y, n, ntiles);
    };
    full(); symv();
    CHK(hipDeviceSynchronize());
    std::vector<double> h(n), hr(n);
    CHK(hipMemcpy(h.data(), y, n * 8, hipMemcpyDeviceToHost));
    CHK(hipMemcpy(hr.data(), yref, n * 8, hipMemcpyDeviceToHost));
    double maxerr = 0, maxref = 0;
    for (uint64_t i = 0; i < n; i++) { maxerr = std::max(maxerr, std::fabs(h[i] - hr[i])); maxref = std::max(maxref, std::fabs(hr[i])); }
    printf("N=%llu tasks=%zu  max|y_symv - y_full| / max|y| = %.3e\n", (unsigned long long)n, tasks.size(), maxerr / maxref);

    const double t_full = time_ms(full, 10);
    const double t_task = time_ms([&] { hipLaunchKernelGGL(symv_task_kernel, dim3((unsigned)tasks.size()), dim3(256), 0, 0, A, p, dt, rowpart, colpart, n, ntiles); }, 10);
    const double t_symv = time_ms(symv, 10);
    const double gb_full = 8.0 * n * n / 1e9;
    const double gb_symv = gb_full / 2 + 8.0 * n * TILE / 2 / 1e9 /*diag tiles, upper bound*/ + 2 * 8.0 * nblk * n / 2 / 1e9 /*colpart w+r*/;
    printf("full product      : %8.3f ms  %7.1f GB/s\n", t_full, gb_full / t_full * 1e3);
    printf("symv pass 1       : %8.3f ms\n", t_task);
    printf("symv pass 1+2     : %8.3f ms  (~%.1f GB moved, %7.1f GB/s)  speed-up vs full product %.2fx\n", t_symv, gb_symv,
           gb_symv / t_symv * 1e3, t_full / t_symv);
    return maxerr / maxref < 1e-12 ? 0 : 2;
}
