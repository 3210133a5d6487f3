// lam_host_plan.h -- the HOST-ONLY arithmetic of the hot path: the reference's row partition, the device row pitch, the
// symmetric product's task plan and its exhaustive check.  No HIP include and no HIP call: this header is part of the product
// build (csrc/lam_hip.hip includes it through lam_kernels.h; the kernels use SymvTask / SymvIndex / symv_use from here) AND is
// compiled by plain g++ with -fsanitize=address,undefined into tests/host_asan/ (`-m "not gpu"`): whatever this code gets
// wrong becomes an out-of-bounds access on the DEVICE, where no sanitizer is available on this pool.
#pragma once

#include <stdint.h>

#include <algorithm>
#include <cmath>
#include <cstddef>
#include <vector>

#if defined(__HIPCC__)
#define LAM_HOST_DEVICE __host__ __device__ __forceinline__
#else
#define LAM_HOST_DEVICE inline
#endif

namespace lam {

// ConjugateGradient_CPU_MPI_OMP.hpp:176-184: n / P rows each, the remainder on the LAST rank
inline void partition_rows(uint64_t n, int P, int q, uint64_t *row0, uint64_t *nrows)
{
    const uint64_t base = n / (uint64_t)P;
    *row0 = base * (uint64_t)q;
    *nrows = base + ((q == P - 1) ? n % (uint64_t)P : 0);
}

// Row pitch of the matrix on the device, in ELEMENTS of `ea` bytes: n rounded up so that a row is a whole number of 4-KiB pages
// (of 16-byte vectors when a row is shorter than a page); the padding is zero.
inline uint64_t row_pitch(uint64_t n, size_t ea)
{
    const uint64_t align = n * ea >= 4096 ? 4096 / ea : 16 / ea;     // elements
    return (n + align - 1) / align * align;
}

// ---- symmetric product: tasks, index, use rule (kernels: lam_kernels.h, "Symmetric product") ------------------------------
struct SymvTask { uint32_t row0, nrows, strip, rp; };     // rp: where the task's row partials start in rowpart (elements)
// What the second pass needs to find the partials, one uint32 array on the device (offsets in uint32 units):
//   runs      5 words per row run: first task, number of tasks (the run's strips are consecutive tasks), first row, rows, and
//             where the first task's row partials start in rowpart (the run's tasks follow at a pitch of `rows`)
//   row8      for every 8 local rows the run they belong to (task heights are multiples of 8)
//   strip_base / strip_tasks   per strip the list of the tasks that cover it, in task order
struct SymvIndex { uint32_t runs, row8, strip_base, strip_tasks; };

constexpr uint32_t kSymvFull = 0x40000000u;       // flag in SymvTask::nrows (host: the whole strip lies inside the matrix and the run is a
                                                  // whole number of 8-row steps: no load needs a test, only the products may)
constexpr uint32_t kSymvFlags = 0xc0000000u;
constexpr uint32_t kSymvInterior = 0x80000000u;   // flag in SymvTask::nrows (host: every element of the task is used by both sides,
                                                  // the whole strip lies inside the row, a whole number of 8-row steps)

// Which elements a row uses.  One shard: the upper triangle (col >= row; the diagonal for the row side only).  Several row
// shards (CYC): an upper-triangle split would leave the first shard with (2P-1)/P^2 of the work, so every row takes the CYCLIC
// window of the (N-1)/2 columns behind its diagonal instead -- d = (col - row) mod N in [1, (N-1)/2], and for even N the antipode
// d = N/2 for the rows of the upper half only: every pair {i, j} is covered once, every row does the same work, contiguous row
// shards stay balanced.
template <bool CYC>
LAM_HOST_DEVICE void symv_use(uint64_t col, uint64_t grow, uint64_t n, bool *row_side, bool *col_side)
{
    if (!CYC) { *row_side = col >= grow; *col_side = col > grow; return; }
    const uint64_t d = col >= grow ? col - grow : col + n - grow;
    const bool in_window = (d >= 1 && d <= (n - 1) / 2) || ((n & 1) == 0 && d == n / 2 && grow < n / 2);
    *row_side = d == 0 || in_window;
    *col_side = in_window;
}

// The symmetric product's plan for one shard (rows [R0, R0 + nloc) of an n x n matrix, strips of SS columns, ncv = columns a row
// holds in whole vectors): pure host arithmetic, shared by the launcher (lam_launch.h) and by symv_plan_check below.
// tasks: in dispatch order -- row run by row run, the strips of a run side by side --, which is also the order of the partials.
struct SymvPlan {
    std::vector<SymvTask> tasks;
    std::vector<uint32_t> index;      // the device-side index (SymvIndex) ...
    SymvIndex ix;                     // ... and where its parts start
    uint32_t nruns = 0;
    uint64_t rowpart_elems = 0;       // row partials of all tasks (one per row of every task)
};
inline void symv_plan(uint64_t n, uint64_t ncv, uint64_t SS, uint64_t R0, uint64_t nloc, bool cyc, SymvPlan *out)
{
    const uint32_t nstrips = (uint32_t)((ncv + SS - 1) / SS);
    const uint64_t H = (n - 1) / 2;
    // Task heights.  The partial stores are what separates the first pass from the rate of its loads alone (190 MB of them cost
    // 4-10 % at N=65536, profiles/r04_symv2_probe.txt), and a task stores SS column partials whatever its height: tall tasks for
    // the bulk, shorter ones only for what is dispatched last (the launch hands out tasks in list order and should end on short
    // ones: the last ~8 % of the work).  One shard (the triangle: row r holds n - r elements): `tall` rows up to the row below
    // which 60 % of the work lies, tall / 4 up to 92 %, at most 64 after; tall = the power of two that leaves ~1500 or more tall
    // tasks, at most 2048 (fp64: 2048 from N = 65536 on, 512 at 32768); N < 16384: two classes, tall up to row 0.65 n and
    // tall / 8 after (a launch wants some thousands of tasks).  Several shards (every row holds n / 2 elements): tall so that a
    // shard has >= ~4000 tasks, tall / 4 for its last 8 % of rows.
    uint64_t tall = 32, mid_from, small_from;
    bool two_classes = cyc;
    if (!cyc) {
        while (tall < 2048 && 2 * tall * 3000 <= n * nstrips) tall *= 2;     // ~1500 or more tall tasks: n / tall runs x nstrips / 2 strips
        mid_from = (uint64_t)((1.0 - std::sqrt(0.40)) * (double)n);
        small_from = (uint64_t)((1.0 - std::sqrt(0.08)) * (double)n);
        if (n < 16384) {                                           // small systems: the second pass's fixed cost counts, fewer tasks win
            two_classes = true;
            mid_from = small_from = (uint64_t)(0.65 * (double)n);
        }
    } else {
        tall = 8;
        const uint64_t strips_per_run = n / 2 / SS + 2;
        while (tall < 1024 && nloc * strips_per_run / (2 * tall) >= 4000) tall *= 2;
        mid_from = small_from = (uint64_t)(0.92 * (double)nloc);
    }
    const uint64_t mid = std::max<uint64_t>(8, cyc ? tall / 4 : (two_classes ? tall / 8 : tall / 4));
    const uint64_t small = two_classes ? mid : std::min<uint64_t>(64, std::max<uint64_t>(8, tall / 16));
    mid_from = mid_from / tall * tall;                            // classes start on multiples of the height before them
    small_from = std::max(mid_from, small_from / mid * mid);
    auto meets = [](uint64_t a0, uint64_t a1, uint64_t b0, uint64_t b1) { return a0 <= b1 && b0 <= a1; };   // closed intervals
    std::vector<SymvTask> &tasks = out->tasks;
    std::vector<uint32_t> runs, row8((nloc + 7) / 8, 0);
    std::vector<std::vector<uint32_t>> per_strip(nstrips);
    for (uint64_t r = 0; r < nloc;) {
        const uint64_t h = std::min<uint64_t>(nloc - r, r < mid_from ? tall : (r < small_from ? mid : small));
        const uint64_t ga = R0 + r, gb = ga + h;                   // global rows [ga, gb)
        const uint32_t run = (uint32_t)(runs.size() / 5), first = (uint32_t)tasks.size();
        for (uint32_t st = 0; st < nstrips; st++) {
            const uint64_t c0 = (uint64_t)st * SS, c1 = std::min<uint64_t>(c0 + SS, n) - 1;     // real columns [c0, c1]
            const bool full = c0 + SS <= n && h % 8 == 0;             // the whole strip inside the matrix (no padding column), whole 8-row steps
            bool needed, interior = full;
            if (!cyc) {
                needed = c1 >= ga;                                 // some column at or right of the first row's diagonal
                interior = interior && c0 >= gb;                   // every column right of every row
            } else {
                // the union of the rows' windows (diagonal and antipode included) is the cyclic interval [ga, gb - 1 + n / 2]
                needed = meets(c0, c1, ga, gb - 1 + n / 2) || meets(c0 + n, c1 + n, ga, gb - 1 + n / 2);
                bool in = false;
                for (uint64_t k = 0; k < 2; k++) {                 // the strip as it lies behind the rows, unwrapped
                    const uint64_t u0 = c0 + k * n, u1 = c0 + SS - 1 + k * n;
                    in = in || (u0 >= gb && u1 - ga <= H);         // 1 <= d <= (n - 1) / 2 for every row and column
                }
                interior = interior && in;
            }
            if (!needed) continue;
            per_strip[st].push_back((uint32_t)tasks.size());
            tasks.push_back({(uint32_t)r, (uint32_t)h | (interior ? kSymvInterior : 0u) | (full ? kSymvFull : 0u), st, (uint32_t)out->rowpart_elems});
            out->rowpart_elems += h;
        }
        runs.insert(runs.end(), {first, (uint32_t)tasks.size() - first, (uint32_t)r, (uint32_t)h, tasks.size() > first ? tasks[first].rp : 0u});
        for (uint64_t q = r / 8; q < (r + h + 7) / 8; q++) row8[q] = run;
        r += h;
    }
    out->nruns = (uint32_t)(runs.size() / 5);
    std::vector<uint32_t> &index = out->index;
    out->ix.runs = 0;
    index = runs;
    out->ix.row8 = (uint32_t)index.size();
    index.insert(index.end(), row8.begin(), row8.end());
    out->ix.strip_base = (uint32_t)index.size();
    uint32_t acc = 0;
    for (uint32_t st = 0; st < nstrips; st++) { index.push_back(acc); acc += (uint32_t)per_strip[st].size(); }
    index.push_back(acc);
    out->ix.strip_tasks = (uint32_t)index.size();
    for (uint32_t st = 0; st < nstrips; st++) index.insert(index.end(), per_strip[st].begin(), per_strip[st].end());
}

// Exhaustive check of the plan of all `shards` row shards of an n x n problem whose storage holds `vec` elements per 16-byte
// vector (fp64 2, fp32 4, bf16 8; NV = 1 as launched: strips of 256 * vec columns): every task is walked element by element
// through the kernel's own use rule, and the directed products y_i += A_ij p_j (i, j in [0, n)) are counted in TWO BITMAPS
// (seen once / seen again: n^2 / 4 bytes in all -- N = 65536 needs 1 GiB, N = 131072 4 GiB; a byte per pair would need 17 GB).
//   *bad_pairs    products not made exactly once (must be 0)
//   *bad_interior elements of tasks flagged "interior" -- which the kernel processes without any test -- that are not used by
//                 both sides or lie outside the matrix, plus tasks flagged "full" that are not (must be 0)
//   *ntasks       tasks of all shards
// Also verifies the device-side index the second pass walks (every task listed once under its strip, inside its run, row
// partials contiguous).  Returns 0, or -1 when the index is inconsistent or an argument is bad.  O(n^2) time.
inline int symv_plan_check(uint64_t n, int shards, uint64_t vec, int max_shards, uint64_t *bad_pairs, uint64_t *bad_interior, uint64_t *ntasks)
{
    if (n == 0 || shards < 1 || shards > max_shards || (uint64_t)shards > n || !bad_pairs || !bad_interior || !ntasks || (vec != 2 && vec != 4 && vec != 8))
        return -1;
    const uint64_t SS = 256 * vec, ncv = (n + vec - 1) / vec * vec;
    const bool cyc = shards > 1;
    const uint64_t words = (n * n + 63) / 64;
    std::vector<uint64_t> once(words, 0), again(words, 0);
    auto bump = [&](uint64_t i, uint64_t j) {
        const uint64_t b = i * n + j, w = b >> 6, m = 1ull << (b & 63);
        if (once[w] & m) again[w] |= m; else once[w] |= m;
    };
    *bad_interior = 0;
    *ntasks = 0;
    for (int q = 0; q < shards; q++) {
        uint64_t R0 = 0, nloc = 0;
        partition_rows(n, shards, q, &R0, &nloc);
        SymvPlan plan;
        symv_plan(n, ncv, SS, R0, nloc, cyc, &plan);
        const std::vector<SymvTask> &tasks = plan.tasks;
        *ntasks += tasks.size();
        // the index the second pass walks: every task is listed once for its strip, and its run holds it
        {
            const uint32_t *ix = plan.index.data();
            const uint32_t nstrips = (uint32_t)((ncv + SS - 1) / SS);
            std::vector<uint8_t> seen(tasks.size(), 0);
            for (uint32_t st = 0; st < nstrips; st++)
                for (uint32_t k = ix[plan.ix.strip_base + st]; k < ix[plan.ix.strip_base + st + 1]; k++) {
                    const uint32_t t = ix[plan.ix.strip_tasks + k];
                    if (t >= tasks.size() || tasks[t].strip != st || seen[t]++) return -1;
                }
            uint64_t rp_expect = 0;
            for (size_t t = 0; t < tasks.size(); t++) {
                const uint32_t j = ix[plan.ix.row8 + tasks[t].row0 / 8], h = tasks[t].nrows & ~kSymvFlags;
                if (j >= plan.nruns) return -1;
                const uint32_t *run = ix + plan.ix.runs + 5 * j;
                if (!seen[t] || t < run[0] || t >= run[0] + run[1] || tasks[t].row0 != run[2] || h != run[3] ||
                    (h % 8 != 0 && tasks[t].row0 + h != nloc) || tasks[t].rp != rp_expect || tasks[t].rp != run[4] + (t - run[0]) * h)
                    return -1;
                rp_expect += h;
            }
            if (rp_expect != plan.rowpart_elems || (plan.rowpart_elems >> 32) != 0) return -1;
        }
        for (const SymvTask &t : tasks) {
            const bool interior = (t.nrows & kSymvInterior) != 0, full = (t.nrows & kSymvFull) != 0;
            const uint64_t h = t.nrows & ~kSymvFlags, c0 = (uint64_t)t.strip * SS;
            if ((interior && !full) || (full && (c0 + SS > n || h % 8 != 0))) ++*bad_interior;   // "full": every load is made without a test
            if (t.row0 + h > nloc) return -1;
            for (uint64_t r = 0; r < h; r++) {
                const uint64_t grow = R0 + t.row0 + r;
                for (uint64_t col = c0; col < c0 + SS; col++) {
                    if (col >= ncv) { if (interior) ++*bad_interior; continue; }      // the kernel's `live` test (interior: no test)
                    bool rs, cs;
                    if (cyc) symv_use<true>(col, grow, n, &rs, &cs); else symv_use<false>(col, grow, n, &rs, &cs);
                    if (interior) {                      // processed without any test: must be what the tests would have said
                        if (!(rs && cs) || col >= n) ++*bad_interior;
                        rs = cs = true;
                    }
                    if (col >= n) continue;              // padding column: the matrix holds zeros there, p likewise
                    if (rs) bump(grow, col);             // y_grow += A[grow][col] p[col]
                    if (cs) bump(col, grow);             // y_col  += A[grow][col] p[grow]  (A[col][grow] by symmetry)
                }
            }
        }
    }
    uint64_t bad = 0;
    const uint64_t total = n * n;
    for (uint64_t w = 0; w < words; w++) {
        const uint64_t valid = (w + 1) * 64 <= total ? ~0ull : ((1ull << (total & 63)) - 1);
        bad += (uint64_t)__builtin_popcountll(again[w] & valid) + (uint64_t)__builtin_popcountll(~once[w] & valid);
    }
    *bad_pairs = bad;
    return 0;
}

}  // namespace lam
