// Shared body of the positional-argument drivers (test_CG_single_GPU, test_CG_MultiGPUS_HIP).
// CLI contract of the reference's positional drivers
// (/root/reference/challenge/main/test/test_CG_single_GPU.cpp:17-27, test_CG_CPU_OMP.cpp:17-27):
//     <exe> [matrix.bin [rhs.bin [sol.bin [max_iters [rel_error]]]]]
// defaults io/matrix.bin io/rhs.bin io/sol.bin 1000 1e-9; exit codes 1 (matrix), 2 (rhs), 6 (save).
#ifndef LAM_POSITIONAL_DRIVER_HPP
#define LAM_POSITIONAL_DRIVER_HPP

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <iostream>

template <typename Solver>
int run_positional_driver(int argc, char **argv, Solver &cg, const char *label)
{
    const char *matrix_file = argc > 1 ? argv[1] : "io/matrix.bin";
    const char *rhs_file = argc > 2 ? argv[2] : "io/rhs.bin";
    const char *sol_file = argc > 3 ? argv[3] : "io/sol.bin";
    const int max_iters = argc > 4 ? atoi(argv[4]) : 1000;
    const double rel_error = argc > 5 ? atof(argv[5]) : 1e-9;

    printf("Usage: %s input_file_matrix.bin input_file_rhs.bin output_file_sol.bin max_iters rel_error\n", argv[0]);
    printf("All parameters are optional and have default values\n\n");
    printf("Command line arguments:\n");
    printf("  input_file_matrix: %s\n", matrix_file);
    printf("  input_file_rhs:    %s\n", rhs_file);
    printf("  output_file_sol:   %s\n", sol_file);
    printf("  max_iters:         %d\n", max_iters);
    printf("  rel_error:         %e\n\n", rel_error);

    printf("Reading matrix from file ...\n");
    if (!cg.load_matrix_from_file(matrix_file)) {
        fprintf(stderr, "Failed to read matrix\n");
        return 1;
    }
    printf("Done\n\n");
    printf("Reading right hand side from file ...\n");
    if (!cg.load_rhs_from_file(rhs_file)) {
        fprintf(stderr, "Failed to read right hand side\n");
        return 2;
    }
    printf("Done\n\n");

    printf("Solving the system ...\n");
    const auto t0 = std::chrono::high_resolution_clock::now();
    cg.solve(max_iters, rel_error);
    const auto t1 = std::chrono::high_resolution_clock::now();
    const double sec = std::chrono::duration<double>(t1 - t0).count();
    std::cout << "Time elapsed using " << label << ":" << sec << " s" << std::endl;
    const auto &st = cg.stats();
    printf("GEMV %.6f ms/iter (%.1f GB/s), iteration %.6f ms\n", st.t_gemv * 1e3,
           st.t_gemv > 0 ? st.gemv_bytes / st.t_gemv / 1e9 : 0.0, st.t_iter * 1e3);
    if (const char *sym = getenv("LAM_HIP_SYMMETRIC")) {
        // the drivers have no flag for the symmetric product: when the environment asks for it, say whether the solve ran on it
        int64_t eff = 0;
        if (*sym && *sym != '0' && lam_hip_get_option(cg.context(), "symmetric_effective", &eff) == 0)
            printf("Option symmetric (LAM_HIP_SYMMETRIC=%s): %s\n", sym, eff ? "effective" : "NOT effective (general GEMV)");
    }
    printf("Done\n\n");

    printf("Writing solution to file ...\n");
    if (!cg.save_result_to_file(sol_file)) {
        fprintf(stderr, "Failed to save solution\n");
        return 6;
    }
    printf("Done\n\n");
    printf("Finished successfully\n");
    return 0;
}

#endif
