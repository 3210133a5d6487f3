// random_spd_system.out -- writes a seeded dense random SPD system (matrix + right hand side) in the LAM
// file format.  Same command line, defaults, prints and exit codes as the reference's generator
// (/root/reference/challenge/main/random_spd_system.cpp:127-196):
//     random_spd_system.out [matrix_size [output_file_matrix.bin [output_file_rhs.bin [random_seed]]]]
// defaults 10, io/matrix.bin, io/rhs.bin, time(); exit 1 = bad size, 2 / 3 = matrix / rhs not written.
// The reference builds A = (Q sqrt(D))(Q sqrt(D))^T on the host with MKL dgemm (:66-103, spectrum
// exp(3.5 U[-1,1]), cond ~ 1.1e3) -- O(N^3) flops and two N x N host arrays; this one asks the GPU
// (lam_hip_generate_random_spd: symmetric, strictly diagonally dominant, spectrum spread over the same three
// decades) and streams the rows to the file in 1 GiB pieces, so N = 65536 (34 GB) takes seconds and needs no
// host copy of the matrix.  Same format (:105-121), different numbers: files of the reference generator and of
// this one are interchangeable as inputs, not bit-identical.
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <ctime>
#include <vector>

#include <sys/types.h>

#include "../../include/lam_hip.h"

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
    if (lam_hip_generate_random_spd(ctx, (uint64_t)(uint32_t)seed, 1.0e3) != 0) return die(2, "Failed to generate the matrix");
    printf("Done\n\n");
    printf("Generating the right hand side ...\n");
    if (lam_hip_generate_random_rhs(ctx, (uint64_t)(uint32_t)seed + 10) != 0) return die(3, "Failed to generate the right hand side");
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
