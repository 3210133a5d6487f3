// lam_launch.h -- typed kernel launchers (Impl<TA,TV>: GEMV shapes, symmetric product), dtype dispatch, shard resources.
// Part of the one translation unit csrc/lam_hip.hip (included from there, in order; not a stand-alone header).
#pragma once

namespace {

// ---- typed implementation ----------------------------------------------------------------------
// (the symmetric product's plan -- SymvPlan, symv_plan -- is host-only arithmetic: lam_host_plan.h)
template <typename TA, typename TV>
struct Impl {
    static constexpr int VEC = MatVec<TA>::N;

    // GEMV shapes.  Variants 0-8: gemv_tile_kernel {rows per wave, p-tile columns, p in LDS, rotated
    // tile order}; 9-18: gemv_coop_kernel {rows per workgroup, tile, waves}; 19-22: the MFMA experiment (bf16).
    // The PRODUCT library holds the shapes that are some dtype's default: 13 (cooperative rows, 2 rows per EIGHT-wave
    // workgroup: fp64 production since round 4 -- half as many concurrent row streams as the 4-wave shape, each read 8 KiB at a
    // time: equal at N=65536, 0.4-2.3 % faster at every other size from 8192 to 131072, profiles/r04_variant_vs_size.txt),
    // 10 (the same with 4 waves: fp32 production, where the two are level), 0 (4 rows per wave: bf16 production) and 17 (4 rows
    // per 8-wave workgroup: an OPTION, 0.5-0.8 % faster than 13 at exactly 16 column tiles -- N=65536, and its row shards -- and
    // slower almost everywhere else; not selected by size).  Everything else -- the other tile / cooperative shapes, the grouped probe and the
    // MFMA-fed bf16 GEMV of BASELINE configs[3]'s comparison (slower than the VALU kernel) -- exists only in the library
    // built with -DLAM_TUNING_VARIANTS (`make tuning` -> liblam_hip_tuning.so; tools/gemv_probe.py, bench.py's MFMA child).
    static constexpr int kNumVariants = 28;      // 23, 24: tuning probes gemv_coop_group_kernel (2 / 4 row pairs per workgroup);
                                                 // 25-27: cooperative rows with 8192-column tiles and 4 / 8 rows (bf16 probes, round 4)
    static bool variant_available(int v)
    {
#ifdef LAM_TUNING_VARIANTS
        return v >= 0 && v < kNumVariants;
#else
        return v == 0 || v == 10 || v == 13 || v == 17;
#endif
    }
    // rows per WORKGROUP of each variant (variants 9.. are the cooperative-row shape: R rows per workgroup)
    static int variant_rows_per_block(int v)
    {
        static const int rows[kNumVariants] = {16, 8, 32, 16, 16, 8, 16, 16, 4, 1, 2, 4, 8, 2, 2, 2, 2, 4, 3,
                                                  /* 19-22: MFMA bf16 experiment (bf16 storage only) */ 8, 8, 16, 4,
                                                  /* 23, 24: grouped cooperative rows (tuning probe) */ 4, 8,
                                                  /* 25-27: coop R4 T8192 W4, R4 T8192 W8, R8 T8192 W8 */ 4, 4, 8};
        return rows[v];
    }

    // rows are padded to a multiple of 16 bytes at least (lam_hip_ctx::lda), so the vector kernels serve any N; the any-alignment
    // kernel runs on request only (option force_generic: tests, comparisons)
    static bool fast_ok(const lam_hip_ctx *c) { return !c->opt_generic; }
    static int variant(const lam_hip_ctx *c)
    {
        if (c->opt_gemv_variant >= 0 && variant_available((int)c->opt_gemv_variant)) return (int)c->opt_gemv_variant;
        // production shapes: fp64 -> cooperative rows, 8 waves (variant 13); fp32 -> cooperative rows, 4 waves (variant 10);
        // bf16 storage spends more VALU per byte (widening) and measures best with 4 rows per wave (variant 0)
        return sizeof(TA) == 2 ? 0 : (sizeof(TA) == 8 ? 13 : 10);
    }

    // name of the kernel instantiation launch_gemv() picks for this context (roofline records)
    static std::string kernel_name(const lam_hip_ctx *c)
    {
        const char *ta = sizeof(TA) == 8 ? "double" : (sizeof(TA) == 4 ? "float" : "__hip_bfloat16");
        const char *tv = sizeof(TV) == 8 ? "double" : "float";
        char buf[192];
        if (c->symv_active()) { snprintf(buf, sizeof buf, "symv_task_kernel<%s,NV=%d> + symv_reduce_kernel<%s,NV=%d>", ta, symv_nv(c), ta, symv_nv(c)); return buf; }
        if (!fast_ok(c)) { snprintf(buf, sizeof buf, "gemv_generic_kernel<%s,%s>", ta, tv); return buf; }
        const int v = variant(c);
        const char *nt = c->opt_nt ? "true" : "false";
        struct Tile { int r, tile; bool lds, rot; };
        static const Tile tiles[9] = {{4, 4096, true, true}, {2, 4096, true, true}, {8, 4096, true, true}, {4, 2048, true, true},
                                      {4, 8192, true, true}, {2, 8192, true, true}, {4, 4096, false, true}, {4, 4096, true, false},
                                      {1, 4096, true, true}};
        struct Coop { int r, tile, waves, unroll; };
        static const Coop coops[10] = {{1, 4096, 4, 4}, {2, 4096, 4, 4}, {4, 4096, 4, 4}, {8, 4096, 4, 4}, {2, 4096, 8, 4},
                                       {2, 2048, 4, 4}, {2, 8192, 8, 8}, {2, 8192, 4, 8}, {4, 4096, 8, 4}, {3, 4096, 4, 4}};
        if (v <= 8)
            snprintf(buf, sizeof buf, "gemv_tile_kernel<%s,%s,R=%d,TILE=%d,NT=%s,UNROLL=4,LDS=%s,ROT=%s>", ta, tv, tiles[v].r,
                     tiles[v].tile, nt, tiles[v].lds ? "true" : "false", tiles[v].rot ? "true" : "false");
        else if (v <= 18)
            snprintf(buf, sizeof buf, "gemv_coop_kernel<%s,%s,R=%d,TILE=%d,NT=%s,UNROLL=%d,WAVES=%d>", ta, tv, coops[v - 9].r,
                     coops[v - 9].tile, nt, coops[v - 9].unroll, coops[v - 9].waves);
        else if (v >= 25)
            snprintf(buf, sizeof buf, "gemv_coop_kernel<%s,%s,R=%d,TILE=8192,NT=%s,UNROLL=4,WAVES=%d>", ta, tv, v == 27 ? 8 : 4, nt, v == 25 ? 4 : 8);
        else if (v >= 23)
            snprintf(buf, sizeof buf, "gemv_coop_group_kernel<%s,%s,GROUP=%d>", ta, tv, v == 23 ? 2 : 4);
        else {
            static const int mf[4][2] = {{2, 3}, {2, 1}, {4, 3}, {1, 3}};      // {R, SPLIT} of variants 19..22
            snprintf(buf, sizeof buf, "gemv_mfma_bf16_kernel<R=%d,TILE=4096,NT=true,SPLIT=%d>", mf[v - 19][0], mf[v - 19][1]);
        }
        return buf;
    }

    // number of p.Ap partials the product step of a CG iteration leaves in part_gemv
    static int gemv_grid(const lam_hip_ctx *c, uint64_t nrows)
    {
        if (nrows == 0) return 0;
        if (c->symv_active()) return symv_reduce_grid(c->n);      // symmetric product: one per 32 rows of the second pass
        return kernel_grid(c, nrows);
    }

    // workgroups of the general GEMV kernel (also used on its own by the residual check)
    static int kernel_grid(const lam_hip_ctx *c, uint64_t nrows)
    {
        if (nrows == 0) return 0;
        const uint64_t rows_per_block = fast_ok(c) ? (uint64_t)variant_rows_per_block(variant(c)) : (uint64_t)kWaves;
        return (int)((nrows + rows_per_block - 1) / rows_per_block);
    }

    template <int R, int TILE, bool LDS, bool ROT>
    static void launch_tile(const lam_hip_ctx *c, int grid, hipStream_t st, const GemvArgs<TA, TV> &a)
    {
        if (c->opt_nt)
            hipLaunchKernelGGL((gemv_tile_kernel<TA, TV, R, TILE, true, 4, LDS, ROT>), dim3(grid), dim3(kBlock), 0, st, a);
        else
            hipLaunchKernelGGL((gemv_tile_kernel<TA, TV, R, TILE, false, 4, LDS, ROT>), dim3(grid), dim3(kBlock), 0, st, a);
    }

    // y = A p reading every pair {i, j} once (lam_kernels.h, "Symmetric product").  The task list is built on first use.
    // One shard: the upper triangle -- strip st (SS columns) holds rows [0, min(n, c0 + SS)), cut into runs of `tall` rows up to
    // row 0.65 n and of `tall / 8` rows below (the launch dispatches tasks in list order and so ends on short ones).  Several
    // row shards: every row of the shard takes the cyclic window of (N-1)/2 columns behind its diagonal (symv_use), i.e. the
    // strips that window touches; runs of `tall` rows, `tall / 8` for the last 15 % of the shard's rows.  The list is ordered
    // by first row, strips of one row run side by side (whole rows stream, as in the GEMV).  Shapes from tools/symv2_probe
    // (profiles/r04_symv2_probe.txt): one 16-byte vector per lane and row (a 4-KiB strip per workgroup); one shard: tall = 256
    // rows from N = 16384 on; several: tall so that a shard has >= ~6000 tasks.
    template <int NV>
    static int build_symv_tasks(lam_hip_ctx *c, ShardBase &s, bool cyc)
    {
        const uint64_t n = c->n, SS = (uint64_t)NV * kBlock * VEC, ncv = c->ncols_vec();
        SymvPlan plan;
        symv_plan(n, ncv, SS, s.row0, s.nrows, cyc, &plan);
        const std::vector<SymvTask> &tasks = plan.tasks;
        if (tasks.empty()) return fail(c, LAM_HIP_EINVAL, "symmetric product: no tasks");
        // all four or none: a later failure must not leave the earlier buffers behind
        DevBuf t, rp, cp, sb;
        HIPCHK(c, hipMalloc(&t.p, tasks.size() * sizeof(SymvTask)));
        if (plan.rowpart_elems >> 32) return fail(c, LAM_HIP_EINVAL, "symmetric product: too many row partials for 32-bit offsets");
        HIPCHK(c, hipMalloc(&rp.p, (size_t)plan.rowpart_elems * sizeof(TV)));
        HIPCHK(c, hipMalloc(&cp.p, tasks.size() * SS * sizeof(TV)));
        HIPCHK(c, hipMalloc(&sb.p, plan.index.size() * sizeof(uint32_t)));
        HIPCHK(c, hipMemcpy(t.p, tasks.data(), tasks.size() * sizeof(SymvTask), hipMemcpyHostToDevice));
        HIPCHK(c, hipMemcpy(sb.p, plan.index.data(), plan.index.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
        // columns behind the end of a ragged strip are never written by a task: the second pass does not read them either
        HIPCHK(c, hipMemsetAsync(cp.p, 0, tasks.size() * SS * sizeof(TV), s.stream));
        s.symv_tasks = t.as<SymvTask>(); s.symv_rowpart = rp.p; s.symv_colpart = cp.p; s.symv_index = sb.as<uint32_t>();
        s.symv_ix = plan.ix;
        t.p = rp.p = cp.p = sb.p = nullptr;
        s.symv_ntasks = (int)tasks.size();
        return 0;
    }
    // the two passes.  dst.n == 0: one shard, y = A p and `partial` = the p.y partials (symv_reduce_grid(n) of them).  dst.n > 0:
    // several shards, the shard's full-length contribution goes into dst.p[] (its record in every shard's gather buffer).
    template <int NV>
    static int launch_symv_nv(lam_hip_ctx *c, ShardBase &s, const TV *p, TV *y, double *partial, const CgScalars *sc, const PtrList &dst,
                              const Finalize &fin)
    {
        const bool cyc = dst.n > 0;
        const uint64_t n = c->n, ncv = c->ncols_vec();
        if (s.symv_tasks == nullptr) LAMCHK(build_symv_tasks<NV>(c, s, cyc));
        if (cyc)
            hipLaunchKernelGGL((symv_task_kernel<TA, TV, NV, true>), dim3(s.symv_ntasks), dim3(kBlock), symv_lds_pad<TV>(), s.stream, (const TA *)s.A, p,
                               (const SymvTask *)s.symv_tasks, (TV *)s.symv_rowpart, (TV *)s.symv_colpart, c->lda, ncv, n, s.row0, sc);
        else
            hipLaunchKernelGGL((symv_task_kernel<TA, TV, NV, false>), dim3(s.symv_ntasks), dim3(kBlock), symv_lds_pad<TV>(), s.stream, (const TA *)s.A, p,
                               (const SymvTask *)s.symv_tasks, (TV *)s.symv_rowpart, (TV *)s.symv_colpart, c->lda, ncv, n, s.row0, sc);
        HIPCHK(c, hipGetLastError());
        hipLaunchKernelGGL((symv_reduce_kernel<TV, NV * kBlock * VEC>), dim3(symv_reduce_grid(n) + (fin.active ? 1 : 0)), dim3(kBlock), 0, s.stream,
                           (const TV *)s.symv_rowpart, (const TV *)s.symv_colpart, (const uint32_t *)s.symv_index, s.symv_ix, p, y,
                           partial, n, s.row0, s.nrows, dst, fin, sc);
        HIPCHK(c, hipGetLastError());
        c->n_launch += 2;
        return 0;
    }
    static int symv_reduce_grid(uint64_t n) { return (int)((n + kSymvReduceRows - 1) / kSymvReduceRows); }
    // vectors per lane and row: one everywhere -- with the pipelined interior loop one 4-KiB strip per workgroup runs level with
    // two at N = 65536 (fp64) / 131072 (fp32) and 2-5 % ahead at N <= 40000 (profiles/r04_symv2_probe.txt); two stays compiled
    static int symv_nv(const lam_hip_ctx *) { return 1; }
    // `fin` (several shards): where the reducer workgroup of the second pass writes the shard's part of p.Ap
    static int launch_symv(lam_hip_ctx *c, ShardBase &s, const TV *p, TV *y, double *partial, const CgScalars *sc, const PtrList *dst = nullptr,
                           const Finalize *fin = nullptr)
    {
        PtrList none;
        none.n = 0;
        const PtrList &d = dst ? *dst : none;
        Finalize off;
        off.active = 0; off.mail = 0; off.seq = 0; off.dst.n = 0; off.slot = 0; off.host_err = c->direct_err;
        const Finalize &f = fin ? *fin : off;
        return symv_nv(c) == 2 ? launch_symv_nv<2>(c, s, p, y, partial, sc, d, f) : launch_symv_nv<1>(c, s, p, y, partial, sc, d, f);
    }

    // panel: 0 = whole GEMV; 1 = only columns [lo,hi); 2 = everything but [lo,hi), accumulated onto y
    template <int R, int TILE = 4096, int WAVES = 4, int UNROLL = 4>
    static void launch_coop(const lam_hip_ctx *c, int grid, hipStream_t st, const GemvArgs<TA, TV> &a)
    {
        if (c->opt_nt)
            hipLaunchKernelGGL((gemv_coop_kernel<TA, TV, R, TILE, true, UNROLL, WAVES>), dim3(grid), dim3(WAVES * 64), 0, st, a);
        else
            hipLaunchKernelGGL((gemv_coop_kernel<TA, TV, R, TILE, false, UNROLL, WAVES>), dim3(grid), dim3(WAVES * 64), 0, st, a);
    }

    static int launch_gemv(lam_hip_ctx *c, ShardBase &s, const TV *p, TV *y, double *partial, const CgScalars *sc,
                           int panel = 0, uint64_t lo = 0, uint64_t hi = 0, const Finalize *fin = nullptr, const PtrList *ypeers = nullptr)
    {
        if (s.nrows == 0) return 0;
        GemvArgs<TA, TV> a;
        a.A = (const TA *)s.A; a.p = p; a.y = y; a.partial = partial; a.sc = sc;
        a.n_ypeer = 0;
        for (auto &yp : a.ypeer) yp = nullptr;
        if (ypeers != nullptr)
            for (int j = 0; j < ypeers->n && a.n_ypeer < kMaxShards - 1; j++) a.ypeer[a.n_ypeer++] = (TV *)ypeers->p[j];
        if (fin != nullptr && partial != nullptr) a.fin = *fin;
        else { a.fin.active = 0; a.fin.mail = 0; a.fin.seq = 0; a.fin.dst.n = 0; a.fin.slot = 0; a.fin.host_err = c->direct_err; }
        a.nrows = s.nrows; a.n = c->n; a.row0 = s.row0; a.lda = c->lda;
        // the vector kernels cover whole 16-byte vectors: for an N that is not a multiple of the vector width the last one
        // reaches into the (zero) padding of the row and behind the end of p (zero too: set_problem)
        const uint64_t ncols = fast_ok(c) ? c->ncols_vec() : c->n;
        if (fast_ok(c) && hi == c->n) hi = ncols;
        a.seg_begin[0] = 0; a.seg_end[0] = ncols; a.seg_begin[1] = a.seg_end[1] = 0; a.nseg = 1; a.accumulate = 0;
        if (panel == 1) { a.seg_begin[0] = lo; a.seg_end[0] = hi; }
        else if (panel == 2) {
            a.accumulate = 1;
            a.nseg = 0;
            if (lo > 0) { a.seg_begin[a.nseg] = 0; a.seg_end[a.nseg] = lo; a.nseg++; }
            if (hi < ncols) { a.seg_begin[a.nseg] = hi; a.seg_end[a.nseg] = ncols; a.nseg++; }
            if (a.nseg == 0) return 0;
            if (a.nseg == 1) { a.seg_begin[1] = a.seg_end[1] = 0; }
        }
        const int grid = kernel_grid(c, s.nrows) + (a.fin.active ? 1 : 0);     // + the reducer workgroup (Finalize)
        if (fast_ok(c)) {
            switch (variant(c)) {
            default:
            case 0: launch_tile<4, 4096, true, true>(c, grid, s.stream, a); break;
            case 10: launch_coop<2>(c, grid, s.stream, a); break;
            case 13: launch_coop<2, 4096, 8>(c, grid, s.stream, a); break;
            case 17: launch_coop<4, 4096, 8>(c, grid, s.stream, a); break;
#ifdef LAM_TUNING_VARIANTS
            case 1: launch_tile<2, 4096, true, true>(c, grid, s.stream, a); break;
            case 2: launch_tile<8, 4096, true, true>(c, grid, s.stream, a); break;
            case 3: launch_tile<4, 2048, true, true>(c, grid, s.stream, a); break;
            case 4: launch_tile<4, 8192, true, true>(c, grid, s.stream, a); break;
            case 5: launch_tile<2, 8192, true, true>(c, grid, s.stream, a); break;
            case 6: launch_tile<4, 4096, false, true>(c, grid, s.stream, a); break;
            case 7: launch_tile<4, 4096, true, false>(c, grid, s.stream, a); break;
            case 8: launch_tile<1, 4096, true, true>(c, grid, s.stream, a); break;
            case 9: launch_coop<1>(c, grid, s.stream, a); break;
            case 11: launch_coop<4>(c, grid, s.stream, a); break;
            case 12: launch_coop<8>(c, grid, s.stream, a); break;
            case 14: launch_coop<2, 2048, 4, 4>(c, grid, s.stream, a); break;
            case 15: launch_coop<2, 8192, 8, 8>(c, grid, s.stream, a); break;
            case 16: launch_coop<2, 8192, 4, 8>(c, grid, s.stream, a); break;
            case 18: launch_coop<3>(c, grid, s.stream, a); break;
            case 23: hipLaunchKernelGGL((gemv_coop_group_kernel<TA, TV, 2>), dim3(grid), dim3(kBlock), 0, s.stream, a); break;
            case 24: hipLaunchKernelGGL((gemv_coop_group_kernel<TA, TV, 4>), dim3(grid), dim3(kBlock), 0, s.stream, a); break;
            case 25: launch_coop<4, 8192, 4, 4>(c, grid, s.stream, a); break;
            case 26: launch_coop<4, 8192, 8, 4>(c, grid, s.stream, a); break;
            case 27: launch_coop<8, 8192, 8, 4>(c, grid, s.stream, a); break;
            case 19: case 20: case 21: case 22:
                if constexpr (sizeof(TA) == 2) {
                    const int v = variant(c);
                    if (v == 20) hipLaunchKernelGGL((gemv_mfma_bf16_kernel<2, 4096, true, 1>), dim3(grid), dim3(kBlock), 0, s.stream, a);
                    else if (v == 21) hipLaunchKernelGGL((gemv_mfma_bf16_kernel<4, 4096, true, 3>), dim3(grid), dim3(kBlock), 0, s.stream, a);
                    else if (v == 19) hipLaunchKernelGGL((gemv_mfma_bf16_kernel<2, 4096, true, 3>), dim3(grid), dim3(kBlock), 0, s.stream, a);
                    else hipLaunchKernelGGL((gemv_mfma_bf16_kernel<1, 4096, true, 3>), dim3(grid), dim3(kBlock), 0, s.stream, a);
                } else {
                    return fail(c, LAM_HIP_EINVAL, "gemv_variant 19-22 (MFMA) exist for LAM_HIP_BF16 only");
                }
                break;
#endif
            }
        } else {
            hipLaunchKernelGGL((gemv_generic_kernel<TA, TV>), dim3(grid), dim3(kBlock), 0, s.stream, a);
        }
        LAUNCHED(c);
        return 0;
    }
};

// The typed bodies of the host code are written as generic lambdas over Impl<TA,TV>; TA / TV are recovered with this small trait.
template <typename T> struct ImplTraits;
template <typename TA_, typename TV_> struct ImplTraits<Impl<TA_, TV_>> { using TA = TA_; using TV = TV_; };

// Own-slice panel [lo,hi) of the CG GEMV, or lo == hi when the GEMV stays one launch.  Panels need
// 16-byte aligned segment starts (lo, hi multiples of the vector width) unless the generic kernel runs.
template <typename I>
void cg_panel(const lam_hip_ctx *c, const ShardBase &s, uint64_t *lo, uint64_t *hi)
{
    *lo = *hi = 0;
    uint64_t a = 0, b = 0;
    if (c->opt_panel_hi > c->opt_panel_lo) { a = (uint64_t)c->opt_panel_lo; b = std::min<uint64_t>((uint64_t)c->opt_panel_hi, c->n); }
    else if (c->rank_mode && c->opt_overlap && c->nranks > 1) { a = s.row0; b = s.row0 + s.nrows; }
    if (b <= a || (a == 0 && b >= c->n)) return;
    if (I::fast_ok(c) && (a % I::VEC != 0 || (b % I::VEC != 0 && b != c->n))) return;
    *lo = a; *hi = b;
}

template <typename F>
int dispatch(lam_hip_ctx *c, F &&f)
{
    switch (c->dtype) {
    case LAM_HIP_F64: return f(Impl<double, double>());
    case LAM_HIP_F32: return f(Impl<float, float>());
    case LAM_HIP_BF16: return f(Impl<__hip_bfloat16, float>());
    }
    return fail(c, LAM_HIP_EINVAL, "bad dtype %d", c->dtype);
}

int set_dev(lam_hip_ctx *c, const ShardBase &s)
{
    c->n_setdev++;
    HIPCHK(c, hipSetDevice(s.dev));
    // hipGetLastError() is only used to pick up launch failures right after a launch; drop whatever an
    // earlier, already reported failure (possibly of another context) left in the thread's error slot
    (void)hipGetLastError();
    return 0;
}

PtrList plist_p(lam_hip_ctx *c)
{
    PtrList l;
    l.n = (int)c->sh.size();
    for (int j = 0; j < l.n; j++) l.p[j] = c->sh[j].p;
    return l;
}
PtrList plist_gather(lam_hip_ctx *c, bool second)
{
    PtrList l;
    l.n = (int)c->sh.size();
    for (int j = 0; j < l.n; j++) l.p[j] = second ? (void *)c->sh[j].gather_b : (void *)c->sh[j].gather_a;
    return l;
}

// keep_matrix: leave the matrix allocation alone (lam_hip_set_problem re-uses it when it is large enough)
void free_shard(ShardBase &s, bool keep_matrix = false)
{
    if (hipSetDevice(s.dev) != hipSuccess) { (void)hipGetLastError(); return; }   // never created on a real device
    void *const keepA = keep_matrix ? s.A : nullptr;
    const size_t keepCap = keep_matrix ? s.A_capacity : 0;
    if (keep_matrix) s.A = nullptr;
    // the row-transfer staging buffer belongs to the context, not to a problem: released only with the matrix
    if (!keep_matrix && s.xfer_stage) { (void)hipFree(s.xfer_stage); s.xfer_stage = nullptr; s.xfer_stage_bytes = 0; }
    void *ptrs[] = {s.A, s.p, s.Ap, s.x, s.r, s.b, s.tmp, s.part_gemv, s.part_vec, s.gather_a, s.gather_b, s.sc,
                    s.r_full, s.ap_gather, s.symv_rowpart, s.symv_colpart, s.symv_tasks, s.symv_index, s.symv_gather, s.part_aux};
    for (void *q : ptrs) if (q) (void)hipFree(q);
    s.r_full = s.ap_gather = s.symv_rowpart = s.symv_colpart = s.symv_gather = nullptr;
    s.symv_gather_bytes = 0;
    s.symv_tasks = nullptr;
    s.symv_index = nullptr;
    s.symv_ntasks = 0;
    if (s.sc_host) (void)hipHostFree(s.sc_host);
    if (s.host_flags) (void)hipHostFree(s.host_flags);
    s.host_flags = nullptr;
    s.A = s.p = s.Ap = s.x = s.r = s.b = s.tmp = nullptr;
    s.A = keepA;
    s.A_capacity = keepCap;
    s.part_gemv = s.part_vec = s.gather_a = s.gather_b = nullptr;
    s.part_aux = nullptr;
    s.sc = nullptr; s.sc_host = nullptr;
}

// streams and events of one shard (device memory is released by free_shard)
void release_handles(ShardBase &s)
{
    if (hipSetDevice(s.dev) != hipSuccess) { (void)hipGetLastError(); return; }
    hipEvent_t *evs[] = {&s.ev_a, &s.ev_b, &s.ev_p, &s.ev_gathered};
    for (auto e : evs) if (*e) { (void)hipEventDestroy(*e); *e = nullptr; }
    for (int i = 0; i < kLag; i++) {
        hipEvent_t *ring[] = {&s.ev_g0[i], &s.ev_g1[i], &s.ev_g2[i], &s.ev_g3[i]};
        for (auto e : ring) if (*e) { (void)hipEventDestroy(*e); *e = nullptr; }
    }
    for (int i = 0; i < kLag; i++)
        for (auto &e : s.ev_x[i]) if (e) { (void)hipEventDestroy(e); e = nullptr; }
    if (s.comm_stream) { (void)hipStreamSynchronize(s.comm_stream); (void)hipStreamDestroy(s.comm_stream); s.comm_stream = nullptr; }
    if (s.stream) { (void)hipStreamSynchronize(s.stream); (void)hipStreamDestroy(s.stream); s.stream = nullptr; }
}

// a context whose creation failed half-way: give back what it already holds
void release_hub(lam_hip_ctx *c)
{
    if (c->sh.empty() || hipSetDevice(c->sh[0].dev) != hipSuccess) { (void)hipGetLastError(); return; }
    for (auto &ev : c->ev_join) if (ev) { (void)hipEventDestroy(ev); ev = nullptr; }
    if (c->hub_stream) { (void)hipStreamSynchronize(c->hub_stream); (void)hipStreamDestroy(c->hub_stream); c->hub_stream = nullptr; }
}

void abandon(lam_hip_ctx *c)
{
    release_hub(c);
    for (auto &s : c->sh) { free_shard(s); release_handles(s); }
    if (c->direct_err) { (void)hipHostFree(c->direct_err); c->direct_err = nullptr; }
}

}  // namespace
