// random_spd_system.out -- writes a seeded dense random SPD system (matrix + right hand side) in the LAM
// file format.  Same command line, defaults, prints and exit codes as the reference's generator
// (/root/reference/challenge/main/random_spd_system.cpp:127-196):
//     random_spd_system.out [matrix_size [output_file_matrix.bin [output_file_rhs.bin [random_seed]]]]
// defaults 10, io/matrix.bin, io/rhs.bin, time(); exit 1 = bad size, 2 / 3 = matrix / rhs not written.
// THE MATRIX LAW IS THE REFERENCE'S (round 4): A = Q diag(d) Q^T with d_i = exp(3.5 u_i), u_i in [-1, 1] (:66-97, cond ~ 1.1e3)
// and rhs_i in [-1, 1] (:166), and the random numbers are the reference's too -- srand(seed - 10) / rand() for d, srand(seed +
// 10) / rand() for the rhs, srand(seed) / rand() for the vectors behind Q, in the reference's order (:27-37,76,83,166).  The
// one difference is Q: the reference orthonormalises an N x N random matrix with a recursive block Gram-Schmidt on MKL dgemm
// (O(N^3) flops, two N x N host arrays, needs <mkl.h>); here Q = H_k ... H_1 is the product of k = 4 Householder reflectors
// whose vectors are the first k columns of that same random matrix, applied on the GPU (lam_hip_generate_spectrum_spd: one
// GEMV + one rank-2 update per reflector).  Same spectrum (exact up to rounding), same rhs, exactly symmetric, dense; CG
// converges like on the reference's matrices (359-360 iterations to 1e-9, TESTS/BEST_RESULTS:93-118 -- N-independent, the
// spectrum's law is).  The rows are streamed to the file in 1 GiB pieces: N = 65536 (34 GB) takes seconds and no host copy
// of the matrix.  Same format (:105-121); not bit-identical to the reference generator's files (its Q depends on MKL).
// Optional 5th argument `dominant`: the round-1..3 generator instead (symmetric, strictly diagonally dominant, spectrum
// spread over the same three decades; what bench.py generates in place for sizes that need no file).
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <ctime>
#include <string>
#include <vector>

#include <sys/types.h>

#include "../../include/lam_hip.h"
#include "../LAM/src/HIP/reference_system.hpp"

namespace
{
bool write_header(FILE *f, uint64_t rows, uint64_t cols)
{
    const uint64_t h[2] = {rows, cols};
    return fwrite(h, sizeof(uint64_t), 2, f) == 2;
}
}  // namespace

int main(int argc, char **argv)
{
    printf("Usage: ./random_spd_system matrix_size output_file_matrix.bin output_file_rhs.bin random_seed\n");
    printf("All parameters are optional and have default values\n\n");
    const char *output_file_matrix = "io/matrix.bin";
    const char *output_file_rhs = "io/rhs.bin";
    size_t size = 10;
    int seed = (int)time(nullptr);
    if (argc > 1) size = static_cast<size_t>(atoll(argv[1]));
    if (argc > 2) output_file_matrix = argv[2];
    if (argc > 3) output_file_rhs = argv[3];
    if (argc > 4) seed = atoi(argv[4]);
    const bool dominant = argc > 5 && std::string(argv[5]) == "dominant";
    const int reflectors = argc > 6 ? std::max(0, atoi(argv[6])) : 4;
    printf("Command line arguments:\n");
    printf("  matrix_size:        %zu\n", size);
    printf("  output_file_matrix: %s\n", output_file_matrix);
    printf("  output_file_rhs:    %s\n", output_file_rhs);
    printf("  seed:               %d\n\n", seed);
    if ((ssize_t)size <= 0) {
        fprintf(stderr, "Wrong argument value\n");
        return 1;
    }

    lam_hip_ctx *ctx = nullptr;
    const int dev = 0;
    if (lam_hip_create(&ctx, LAM_HIP_F64, 1, &dev) != 0) {
        fprintf(stderr, "No GPU: %s\n", lam_hip_last_error(nullptr));
        return 1;
    }
    auto die = [&](int code, const char *what) {
        fprintf(stderr, "%s: %s\n", what, lam_hip_last_error(ctx));
        lam_hip_destroy(ctx);
        return code;
    };
    printf("Generating the matrix ...\n");
    if (lam_hip_set_problem(ctx, size) != 0) return die(2, "Failed to allocate the matrix");
    LAM::ReferenceSystemStreams streams;
    if (dominant) {
        if (lam_hip_generate_random_spd(ctx, (uint64_t)(uint32_t)seed, 1.0e3) != 0) return die(2, "Failed to generate the matrix");
    } else {
        streams = LAM::reference_system_streams(size, seed, reflectors);    // the reference's srand/rand streams, its order
        if (lam_hip_generate_spectrum_spd(ctx, streams.eig.data(), streams.reflectors.data(), reflectors) != 0)
            return die(2, "Failed to generate the matrix");
    }
    printf("Done\n\n");
    printf("Generating the right hand side ...\n");
    if (dominant) {
        if (lam_hip_generate_random_rhs(ctx, (uint64_t)(uint32_t)seed + 10) != 0) return die(3, "Failed to generate the right hand side");
    } else {
        if (lam_hip_set_rhs(ctx, streams.rhs.data()) != 0) return die(3, "Failed to generate the right hand side");   // :166
    }
    printf("Done\n\n");

    printf("Writing matrix to file ...\n");
    {
        FILE *f = fopen(output_file_matrix, "wb");
        bool ok = f != nullptr && write_header(f, size, size);
        const uint64_t chunk_rows = std::max<uint64_t>(1, (1ull << 30) / (size * sizeof(double)));
        std::vector<double> rows((size_t)std::min<uint64_t>(chunk_rows, size) * size);
        for (uint64_t r = 0; ok && r < size; r += chunk_rows) {
            const uint64_t nr = std::min<uint64_t>(chunk_rows, size - r);
            ok = lam_hip_download_rows(ctx, r, nr, rows.data()) == 0 && fwrite(rows.data(), sizeof(double), nr * size, f) == nr * size;
        }
        if (f) ok = (fclose(f) == 0) && ok;
        if (!ok) {
            fprintf(stderr, "Failed to save matrix\n");
            lam_hip_destroy(ctx);
            return 2;
        }
    }
    printf("Done\n\n");
    printf("Writing right hand side to file ...\n");
    {
        std::vector<double> b(size);
        FILE *f = fopen(output_file_rhs, "wb");
        bool ok = f != nullptr && lam_hip_get_rhs(ctx, b.data()) == 0 && write_header(f, size, 1) &&
                  fwrite(b.data(), sizeof(double), size, f) == size;
        if (f) ok = (fclose(f) == 0) && ok;
        if (!ok) {
            fprintf(stderr, "Failed to save right hand side\n");
            lam_hip_destroy(ctx);
            return 3;
        }
    }
    printf("Done\n\n");
    lam_hip_destroy(ctx);
    printf("Finished successfully\n");
    return 0;
}
