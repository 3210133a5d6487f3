// test_CG_MultiGPUS_HIP_RCCL.out (also installed under the reference's target names
// test_CG_MultiGPUS_CUDA_NCCL.out / test_CG_MultiGPUS_CUDA_MPI.out, test/CMakeLists.txt:19-29) -- one
// process per GPU, RCCL over xGMI.  Drop-in for the
// reference's getopt-style distributed drivers test_CG_MultiGPUS_CUDA_NCCL.out / _MPI.out /
// test_CPU_MPI_OMP.out (/root/reference/challenge/main/test/test_CG_CPU_MPI_OMP.cpp:205-291 -- the
// three differ only in the class name).  Same flags (-A -b | -s, -o -i -e -v -h), same defaults
// (io/matrix.bin io/rhs.bin io/sol.bin 10000 1e-9), same one-line CSV on stdout:
//   N, procs, threads, load_or_gen_s, comm_init_s, avg_gemv_s, avg_iter_s, iters, rel_err, cg_total_s
// (the comm_init column is the NCCL variant's, ConjugateGradient_MultiGPUS_CUDA_NCCL.cu:332-334).
// Differences, all deliberate: the per-iteration averages are true averages (the CPU variant's
// are divided by num_iters twice, CPU_MPI_OMP.hpp:119-124); cg_total_s is not truncated to whole
// seconds in generate mode (test_CG_CPU_MPI_OMP.cpp:173-178); running with neither -s nor -A/-b
// prints the usage and returns 1 instead of returning an uninitialised value (:281-291).
// Extensions: -r <seed> -c <cond> generate the seeded dense random SPD system of size -s.
// Launch: 1 rank = just run it; N ranks = any launcher that exports RANK/WORLD_SIZE/LOCAL_RANK
// (or PMI_*/OMPI_*/SLURM_*), e.g. `mpiexec -n 8` or `torchrun --no-python`; see lam_bootstrap.hpp.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <vector>

#include <unistd.h>

#include "LAM.hpp"

namespace
{
void usage(const char *exe)
{
    printf("Usage: %s [ (-A -b | -s) -o -e -i -h -v]\n", exe);
    printf("Options:\n");
    printf("  -A <file>       Read matrix from file\n");
    printf("  -b <file>       Read right hand side from file\n");
    printf("  -o <file>       Write solution to file\n");
    printf("  -i <int>        Maximum number of iterations\n");
    printf("  -e <float>      Relative error\n");
    printf("  -s <int>        Generate matrix of size n x n\n");
    printf("  -r <seed>       with -s: seeded dense random SPD system instead of tridiag(1,2,1)\n");
    printf("  -c <float>      with -r: spread of the spectrum (default 1e4)\n");
    printf("  -R <seed>       with -s: the system the reference GENERATOR makes for this seed (random_spd_system.cpp: spectrum\n");
    printf("                  exp(3.5 U[-1,1]), rhs U[-1,1], its srand/rand streams), built on the device without the files\n");
    printf("  -t <type>       f64 (default, what the reference drivers hard-code), f32, or bf16 (bf16 matrix\n");
    printf("                  storage, fp32 vectors); files hold doubles for f64 and floats otherwise\n");
    printf("  -P <shards>     ONE process drives <shards> row shards, dealt round-robin over the visible GPUs (the reference's\n");
    printf("                  test_CG_MultiGPUS_CUDA topology with this driver's flags and CSV; not under a multi-rank launcher)\n");
    printf("  -g              CSV: the GEMV column is GEMV + exchange (collectives / joins), the reference's convention\n");
    printf("  -v              Verbose mode\n");
    printf("  -h              Show this help message\n");
}
}  // namespace

struct Options {
    const char *matrix_file = "io/matrix.bin", *rhs_file = "io/rhs.bin", *sol_file = "io/sol.bin";
    int max_iters = 10000;
    double rel_error = 1e-9, cond = 1e4;
    size_t rows = 0;
    long seed = -1;
    int ref_seed = 0, shards = 0;
    bool have_ref_seed = false;
    bool verbose = false, mode_generate = false, mode_load = false, bf16_storage = false, gemv_plus_comm = false;
};

template <typename T>
int run(const lam_bootstrap::Launch &L, const Options &o, int ndev);

static int real_main(int argc, char **argv, const lam_bootstrap::Launch &L);

int main(int argc, char **argv)
{
    // this program's stdout is a one-line CSV protocol: ask the library to keep RCCL's version banner (printed to
    // stdout when the communicator is created) on stderr.  Opt-in: the library never moves a host's stdout by default.
    setenv("LAM_HIP_QUIET_RCCL", "1", 0);
    lam_bootstrap::Launch L;
    if (!lam_bootstrap::init(&argc, &argv, L)) {
        fprintf(stderr, "bootstrap failed: %s\n", lam_hip_last_error(nullptr));
        lam_bootstrap::finalize(L);
        return 1;
    }
    const int rc = real_main(argc, argv, L);     // every exit path passes through finalize (MPI_Finalize)
    lam_bootstrap::finalize(L);
    return rc;
}

static int real_main(int argc, char **argv, const lam_bootstrap::Launch &L)
{
    Options o;
    const char *&matrix_file = o.matrix_file, *&rhs_file = o.rhs_file, *&sol_file = o.sol_file;
    int &max_iters = o.max_iters;
    double &rel_error = o.rel_error, &cond = o.cond;
    size_t &rows = o.rows;
    long &seed = o.seed;
    bool &verbose = o.verbose, &mode_generate = o.mode_generate, &mode_load = o.mode_load;
    const bool root = L.rank == 0;
    const char *precision = "f64";

    int opt;
    while ((opt = getopt(argc, argv, "hvgA:b:o:i:e:s:r:R:c:t:P:")) != -1) {
        switch (opt) {
        case 'A':
        case 'b':
            if (mode_generate) {
                fprintf(stderr, "Option -s cannot be used with -%c.\n", opt);
                return 1;
            }
            mode_load = true;
            (opt == 'A' ? matrix_file : rhs_file) = optarg;
            break;
        case 'o': sol_file = optarg; break;
        case 'i': max_iters = atoi(optarg); break;
        case 'e': rel_error = atof(optarg); break;
        case 's':
            if (mode_load) {
                fprintf(stderr, "Option -A and -b cannot be used with -s.\n");
                return 1;
            }
            mode_generate = true;
            rows = (size_t)atoll(optarg);
            break;
        case 'r': seed = atol(optarg); break;
        case 'R': o.ref_seed = atoi(optarg); o.have_ref_seed = true; break;
        case 'c': cond = atof(optarg); break;
        case 't': precision = optarg; break;
        case 'g': o.gemv_plus_comm = true; break;
        case 'P': o.shards = atoi(optarg); break;
        case 'v': verbose = true; break;
        case 'h':
            if (root) usage(argv[0]);
            return 0;
        default:
            if (root) usage(argv[0]);
            return 1;
        }
    }
    if (!mode_generate && !mode_load) {
        if (root) usage(argv[0]);
        return 1;
    }

    int ndev = 0;
    if (lam_hip_device_count(&ndev) != 0 || ndev <= 0) {
        fprintf(stderr, "No GPU: %s\n", lam_hip_last_error(nullptr));
        return 1;
    }
    if (o.shards != 0 && (o.shards < 1 || o.shards > LAM_HIP_MAX_SHARDS || L.size > 1 || !strcmp(precision, "bf16"))) {
        if (root) fprintf(stderr, "Option -P takes 1 ... %d shards, runs as ONE process (no multi-rank launcher) and in f64 / f32\n", LAM_HIP_MAX_SHARDS);
        return 1;
    }
    int rc;
    if (!strcmp(precision, "f64")) rc = run<double>(L, o, ndev);
    else if (!strcmp(precision, "f32")) rc = run<float>(L, o, ndev);
    else if (!strcmp(precision, "bf16")) { o.bf16_storage = true; rc = run<float>(L, o, ndev); }
    else {
        if (root) fprintf(stderr, "Unknown precision '%s' (f64, f32, bf16)\n", precision);
        return 1;
    }
    return rc;
}

// the reference drivers hard-code <double> (test_CG_CPU_MPI_OMP.cpp:46,132); the class template is
// instantiated for float too, so the precision is a run-time choice here
template <typename Solver>
int run_solver(Solver &cg, const lam_bootstrap::Launch &L, const Options &o, int procs);

template <typename T>
int run(const lam_bootstrap::Launch &L, const Options &o, int ndev)
{
    if (o.shards > 0) {
        // one process, o.shards row shards dealt round-robin over the devices (several per GPU when there are fewer GPUs)
        std::vector<int> devs;
        for (int q = 0; q < o.shards; q++) devs.push_back(q % ndev);
        LAM::ConjugateGradient_MultiGPUS_HIP<T> cg(devs);
        cg.set_text_output(false);
        cg.set_comm_init_column(true);       // same ten CSV columns as the rank mode (the column is 0: no communicator)
        return run_solver(cg, L, o, o.shards);
    }
    LAM::ConjugateGradient_MultiGPUS_HIP_RCCL<T> cg(L.rank, L.size, L.local_rank % ndev, L.unique_id, o.bf16_storage);
    return run_solver(cg, L, o, L.size);
}

template <typename Solver>
int run_solver(Solver &cg, const lam_bootstrap::Launch &L, const Options &o, int procs)
{
    const char *matrix_file = o.matrix_file, *rhs_file = o.rhs_file, *sol_file = o.sol_file;
    const int max_iters = o.max_iters;
    const double rel_error = o.rel_error, cond = o.cond;
    const size_t rows = o.rows;
    const long seed = o.seed;
    const int ref_seed = o.ref_seed;
    const bool have_ref_seed = o.have_ref_seed;
    const bool verbose = o.verbose, mode_generate = o.mode_generate, root = L.rank == 0;
    cg.set_csv_output(!verbose);
    if (o.gemv_plus_comm) cg.set_gemv_plus_comm(true);
    if (cg.context() == nullptr) return 1;      // creates the RCCL communicator (collective)
    lam_bootstrap::communicator_ready(L);

    using clk = std::chrono::high_resolution_clock;
    if (verbose && root) {
        printf("Command line arguments:\n");
        if (mode_generate) printf("  rows: %zu  (%.3f GB)\n", rows, rows * (double)rows * 8 / 1024.0 / 1024.0 / 1024.0);
        else printf("  input_file_matrix: %s\n  input_file_rhs:    %s\n", matrix_file, rhs_file);
        printf("  output_file_sol:   %s\n  max_iters:         %d\n  rel_error:         %e\n", sol_file, max_iters, rel_error);
        if (o.shards > 0) printf("  One process, %d row shards\n\n", o.shards);
        else printf("  Number of processes: %d (1 GPU each)\n\n", L.size);
    }
    const auto t0 = clk::now();
    bool ok;
    if (mode_generate)
        ok = have_ref_seed ? cg.generate_reference_system(rows, ref_seed)
                           : (seed >= 0 ? cg.generate_random_system(rows, (uint64_t)seed, cond) : cg.generate_matrix(rows, rows));
    else ok = cg.load_matrix_from_file(matrix_file);
    const double t_load = std::chrono::duration<double>(clk::now() - t0).count();
    if (!ok) {      // the loaders agree across ranks: a block that failed on one rank fails here on all
        if (root) fprintf(stderr, "Failed to read matrix\n");
        return 1;
    }
    if (root && !verbose) std::cout << procs << "," << 1 << "," << t_load << ",";
    if (verbose && root) printf("Matrix ready in %f s\n", t_load);
    if (mode_generate) ok = (seed >= 0 || have_ref_seed) ? true : cg.generate_rhs();
    else ok = cg.load_rhs_from_file(rhs_file);
    if (!ok) {
        if (root) fprintf(stderr, "Failed to read right hand side\n");
        return 2;
    }

    const auto t1 = clk::now();
    cg.solve(max_iters, rel_error);
    const double t_cg = std::chrono::duration<double>(clk::now() - t1).count();
    if (root && !verbose) std::cout << t_cg;
    if (verbose && root) {
        const auto &st = cg.stats();
        printf("%s after %d iterations, relative error %e, %f s (GEMV %.4f ms = %.1f GB/s per GPU, exchange %.4f ms)\n",
               st.converged ? "Converged" : "Did not converge", st.num_iters, st.rel_err, t_cg, st.t_gemv * 1e3,
               st.t_gemv > 0 ? st.gemv_bytes / st.t_gemv / 1e9 : 0.0, st.t_exchange * 1e3);
    }
    if (const char *sym = getenv("LAM_HIP_SYMMETRIC")) {
        // no flag for the symmetric product here either: when the environment asks for it, say (on stderr: stdout is the CSV line)
        // whether the solve ran on it
        int64_t eff = 0;
        if (root && *sym && *sym != '0' && lam_hip_get_option(cg.context(), "symmetric_effective", &eff) == 0)
            fprintf(stderr, "Option symmetric (LAM_HIP_SYMMETRIC=%s): %s\n", sym, eff ? "effective" : "NOT effective (general GEMV)");
    }
    if (!cg.save_result_to_file(sol_file)) {
        if (root) fprintf(stderr, "Failed to save solution\n");
        return 6;
    }
    if (verbose && root) printf("Finished successfully\n");
    if (root) std::cout << std::endl;
    return 0;
}
