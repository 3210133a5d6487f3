// host_asan_main.cpp -- TEST INFRASTRUCTURE.  The host-only logic of the product under AddressSanitizer + UBSan (plain g++, no GPU):
//   plan        csrc/lam_host_plan.h -- the reference's row partition, the device row pitch, the symmetric product's task plan
//               checked exhaustively (every directed pair exactly once, nothing unused in an "interior" task, index consistent)
//   loaders     LAM/src/HIP/*.hpp -- the file loaders of the solver classes against a host-memory fake of the C ABI: good files in
//               every topology and precision, truncated / oversized / garbage-header / non-square / empty files, rhs mismatches,
//               the chunk loop (built with a 4-KiB chunk), save_result_to_file, the reference generator's random streams
//   bootstrap   LAM/src/HIP/lam_bootstrap.hpp -- the launcher glue: rank / size from the environment, the unique id through
//               the rendezvous file (run as two processes by the test)
// A bug in any of these becomes an out-of-bounds access on the device in the product, where no sanitizer is available on this pool.
#include <cinttypes>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <vector>

#include "LAM.hpp"
#include "../../2024-eumaster4hpc-student-challenge_amd/csrc/lam_host_plan.h"

extern "C" {
const void *fake_matrix(const lam_hip_ctx *c);
const void *fake_rhs(const lam_hip_ctx *c);
uint64_t fake_row0(const lam_hip_ctx *c);
uint64_t fake_nrows(const lam_hip_ctx *c);
int fake_all_rows_uploaded(const lam_hip_ctx *c);
}

static int g_fail = 0;
#define CHECK(cond, ...)                                                   \
    do {                                                                   \
        if (!(cond)) {                                                     \
            fprintf(stderr, "FAIL %s:%d: %s -- ", __FILE__, __LINE__, #cond); \
            fprintf(stderr, __VA_ARGS__);                                  \
            fprintf(stderr, "\n");                                         \
            g_fail++;                                                      \
        }                                                                  \
    } while (0)

// ---- plan ------------------------------------------------------------------------------------------------------------------
static int run_plan()
{
    // the partition rule (ConjugateGradient_CPU_MPI_OMP.hpp:176-184): contiguous, complete, remainder on the last shard
    for (uint64_t n : {1ull, 2ull, 7ull, 100ull, 1001ull, 65536ull, 50000ull, 4294967311ull})
        for (int P = 1; P <= LAM_HIP_MAX_SHARDS; P++) {
            if ((uint64_t)P > n) continue;
            uint64_t next = 0;
            for (int q = 0; q < P; q++) {
                uint64_t r0 = 0, nr = 0;
                lam::partition_rows(n, P, q, &r0, &nr);
                CHECK(r0 == next && nr == n / P + (q == P - 1 ? n % P : 0), "partition n=%" PRIu64 " P=%d q=%d", n, P, q);
                next = r0 + nr;
            }
            CHECK(next == n, "partition covers n=%" PRIu64 " P=%d", n, P);
        }
    // the row pitch: >= n, whole 4-KiB pages (16-byte vectors for rows shorter than a page), never more than a page of padding
    for (size_t ea : {(size_t)8, (size_t)4, (size_t)2})
        for (uint64_t n : {1ull, 2ull, 3ull, 255ull, 511ull, 512ull, 513ull, 2047ull, 2048ull, 2049ull, 10000ull, 10001ull, 65536ull, 131072ull, 180000ull}) {
            const uint64_t p = lam::row_pitch(n, ea);
            const bool paged = n * ea >= 4096;
            CHECK(p >= n && (p * ea) % (paged ? 4096 : 16) == 0 && (p - n) * ea < (paged ? 4096u : 16u), "pitch n=%" PRIu64 " ea=%zu -> %" PRIu64, n, ea, p);
        }
    // the symmetric product's plan, exhaustively
    const uint64_t sizes[] = {1, 2, 3, 7, 8, 9, 63, 64, 77, 255, 256, 257, 511, 512, 513, 1000, 1023, 1024, 1025, 1536, 2047, 2048, 2050, 3000, 4097};
    for (uint64_t vec : {2ull, 4ull, 8ull})
        for (uint64_t n : sizes)
            for (int shards : {1, 2, 3, 4, 5, 8, 16, 33, 64}) {
                if ((uint64_t)shards > n) continue;
                uint64_t bp = 1, bi = 1, nt = 0;
                const int rc = lam::symv_plan_check(n, shards, vec, LAM_HIP_MAX_SHARDS, &bp, &bi, &nt);
                CHECK(rc == 0 && bp == 0 && bi == 0 && nt > 0, "plan n=%" PRIu64 " shards=%d vec=%" PRIu64 ": rc %d, %" PRIu64 " bad pairs, %" PRIu64 " bad interior, %" PRIu64 " tasks",
                      n, shards, vec, rc, bp, bi, nt);
            }
    for (auto ns : {std::pair<uint64_t, int>{6144, 1}, {6144, 8}, {5000, 6}, {8190, 7}}) {
        uint64_t bp = 1, bi = 1, nt = 0;
        CHECK(lam::symv_plan_check(ns.first, ns.second, 2, LAM_HIP_MAX_SHARDS, &bp, &bi, &nt) == 0 && bp == 0 && bi == 0, "plan n=%" PRIu64 " shards=%d", ns.first, ns.second);
    }
    uint64_t a, b, c;
    CHECK(lam::symv_plan_check(0, 1, 2, LAM_HIP_MAX_SHARDS, &a, &b, &c) == -1 && lam::symv_plan_check(10, 11, 2, LAM_HIP_MAX_SHARDS, &a, &b, &c) == -1 &&
          lam::symv_plan_check(10, 1, 3, LAM_HIP_MAX_SHARDS, &a, &b, &c) == -1 && lam::symv_plan_check(100, LAM_HIP_MAX_SHARDS + 1, 2, LAM_HIP_MAX_SHARDS, &a, &b, &c) == -1, "bad arguments are refused");
    return g_fail;
}

// ---- loaders ---------------------------------------------------------------------------------------------------------------
template <typename T>
static void write_bin(const std::string &path, uint64_t rows, uint64_t cols, const std::vector<T> &data, uint64_t hdr_rows, uint64_t hdr_cols, size_t cut = 0)
{
    FILE *f = fopen(path.c_str(), "wb");
    const uint64_t hdr[2] = {hdr_rows, hdr_cols};
    fwrite(hdr, 8, 2, f);
    const size_t bytes = (size_t)(rows * cols) * sizeof(T);
    fwrite(data.data(), 1, bytes > cut ? bytes - cut : 0, f);
    fclose(f);
}

template <typename T, typename Solver>
static void expect_loaded(Solver &cg, const std::vector<T> &A, const std::vector<T> &b, uint64_t n, const char *what)
{
    lam_hip_ctx *ctx = cg.context();
    CHECK(ctx != nullptr, "%s: context", what);
    if (!ctx) return;
    const uint64_t r0 = fake_row0(ctx), nr = fake_nrows(ctx);
    CHECK(fake_all_rows_uploaded(ctx), "%s: every owned row uploaded exactly once", what);
    CHECK(memcmp(fake_matrix(ctx), A.data() + r0 * n, (size_t)(nr * n) * sizeof(T)) == 0, "%s: matrix rows [%" PRIu64 ", +%" PRIu64 ")", what, r0, nr);
    CHECK(memcmp(fake_rhs(ctx), b.data(), (size_t)n * sizeof(T)) == 0, "%s: rhs", what);
}

template <typename T>
static void loaders_for_type(const std::string &tmp, const char *tname)
{
    std::mt19937_64 rng(7);
    std::uniform_real_distribution<double> U(-1.0, 1.0);
    for (uint64_t n : {1ull, 5ull, 37ull, 64ull, 130ull}) {
        std::vector<T> A(n * n), b(n);
        for (auto &v : A) v = (T)U(rng);
        for (auto &v : b) v = (T)U(rng);
        const std::string m = tmp + "/m_" + tname + std::to_string(n) + ".bin", r = tmp + "/r_" + tname + std::to_string(n) + ".bin";
        write_bin(m, n, n, A, n, n);
        write_bin(r, n, 1, b, n, 1);
        {
            LAM::ConjugateGradient_HIP<T> cg;
            cg.set_text_output(false);
            CHECK(cg.load_matrix_from_file(m.c_str()) && cg.load_rhs_from_file(r.c_str()), "one shard n=%" PRIu64, n);
            expect_loaded(cg, A, b, n, "one shard");
            CHECK(cg.get_num_rows() == n && cg.get_num_cols() == n, "getters");
            const std::string sol = tmp + "/sol.bin";
            CHECK(cg.save_result_to_file(sol.c_str()), "save");
            FILE *f = fopen(sol.c_str(), "rb");
            uint64_t hdr[2] = {0, 0};
            std::vector<T> x(n);
            CHECK(f && fread(hdr, 8, 2, f) == 2 && hdr[0] == n && hdr[1] == 1 && fread(x.data(), sizeof(T), n, f) == n && fgetc(f) == EOF, "solution file n=%" PRIu64, n);
            if (f) fclose(f);
            CHECK(memcmp(x.data(), b.data(), n * sizeof(T)) == 0, "solution payload");
        }
        for (int P : {2, 3, 7, 16, 64}) {
            if ((uint64_t)P > n) continue;
            {
                LAM::ConjugateGradient_MultiGPUS_HIP<T> cg(std::vector<int>(P, 0));
                cg.set_text_output(false);
                CHECK(cg.load_matrix_from_file(m.c_str()) && cg.load_rhs_from_file(r.c_str()), "%d shards n=%" PRIu64, P, n);
                expect_loaded(cg, A, b, n, "one process, several shards");
            }
            for (int rank = 0; rank < P; rank++) {
                char id[LAM_HIP_UNIQUE_ID_BYTES] = {0};
                LAM::ConjugateGradient_MultiGPUS_HIP_RCCL<T> cg(rank, P, 0, id);
                cg.set_csv_output(false);
                CHECK(cg.load_matrix_from_file(m.c_str()) && cg.load_rhs_from_file(r.c_str()), "rank %d of %d n=%" PRIu64, rank, P, n);
                expect_loaded(cg, A, b, n, "rank mode");
                uint64_t r0 = 0, nr = 0;
                lam::partition_rows(n, P, rank, &r0, &nr);
                CHECK(cg.get_num_rows() == nr, "rank %d of %d holds %" PRIu64 " rows", rank, P, nr);
            }
        }
    }
    // ---- files that must be refused (and must not be read past their end) ----
    const uint64_t n = 37;
    std::vector<T> A(n * n, (T)1), b(n, (T)2);
    auto refuse_matrix = [&](const char *what, uint64_t hr, uint64_t hc, size_t cut, uint64_t rows = 37, uint64_t cols = 37) {
        const std::string m = tmp + "/bad.bin";
        std::vector<T> data(rows * cols, (T)1);
        write_bin(m, rows, cols, data, hr, hc, cut);
        LAM::ConjugateGradient_HIP<T> cg;
        cg.set_text_output(false);
        CHECK(!cg.load_matrix_from_file(m.c_str()), "refused: %s", what);
        LAM::ConjugateGradient_MultiGPUS_HIP_RCCL<T> cr(1, 3, 0, nullptr);
        cr.set_csv_output(false);
        CHECK(!cr.load_matrix_from_file(m.c_str()), "refused in rank mode: %s", what);
    };
    refuse_matrix("truncated by 100 bytes", n, n, 100);
    refuse_matrix("truncated to the header", n, n, (size_t)(n * n) * sizeof(T));
    refuse_matrix("header larger than the file", n + 1, n + 1, 0);
    refuse_matrix("rows * cols overflows 64 bits", 1ull << 40, 1ull << 40, 0);
    refuse_matrix("rows * cols * size overflows 64 bits", 1ull << 31, 1ull << 31, 0);
    refuse_matrix("not square", 37, 36, 0, 37, 36);
    refuse_matrix("zero rows", 0, 0, 0);
    {
        const std::string m = tmp + "/short.bin";
        FILE *f = fopen(m.c_str(), "wb");
        fwrite("12345678", 1, 8, f);
        fclose(f);
        LAM::ConjugateGradient_HIP<T> cg;
        cg.set_text_output(false);
        CHECK(!cg.load_matrix_from_file(m.c_str()), "refused: 8-byte file");
        CHECK(!cg.load_matrix_from_file((tmp + "/does_not_exist.bin").c_str()), "refused: missing file");
    }
    {   // the reference's writers leave stack garbage in the upper halves of the header words: accepted by their low halves
        const std::string m = tmp + "/garbage.bin", r = tmp + "/garbage_rhs.bin";
        write_bin(m, n, n, A, (0xdeadbeefull << 32) | n, (0x7fffull << 32) | n);
        write_bin(r, n, 1, b, n, (0xabcdull << 32) | 1);
        LAM::ConjugateGradient_HIP<T> cg;
        cg.set_text_output(false);
        CHECK(cg.load_matrix_from_file(m.c_str()) && cg.load_rhs_from_file(r.c_str()), "garbage in the upper header halves is masked");
        expect_loaded(cg, A, b, n, "garbage header");
    }
    {   // right hand sides that do not fit
        const std::string m = tmp + "/ok.bin";
        write_bin(m, n, n, A, n, n);
        auto refuse_rhs = [&](const char *what, uint64_t rows, uint64_t cols, uint64_t hr, uint64_t hc, size_t cut) {
            const std::string r = tmp + "/badrhs.bin";
            std::vector<T> data(rows * cols, (T)3);
            write_bin(r, rows, cols, data, hr, hc, cut);
            LAM::ConjugateGradient_HIP<T> cg;
            cg.set_text_output(false);
            CHECK(cg.load_matrix_from_file(m.c_str()) && !cg.load_rhs_from_file(r.c_str()), "rhs refused: %s", what);
        };
        refuse_rhs("wrong length", n + 1, 1, n + 1, 1, 0);
        refuse_rhs("two columns", n, 2, n, 2, 0);
        refuse_rhs("truncated", n, 1, n, 1, 8);
        refuse_rhs("header larger than the file", n, 1, 1ull << 33, 1, 0);
        LAM::ConjugateGradient_HIP<T> cg;
        cg.set_text_output(false);
        CHECK(cg.load_matrix_from_file(m.c_str()) && !cg.load_rhs_from_file((tmp + "/nope.bin").c_str()), "missing rhs");
    }
    {   // generate mode and the reference generator's streams (srand / rand in its order): array sizes as promised to the C ABI
        LAM::ConjugateGradient_MultiGPUS_HIP<T> cg(std::vector<int>(3, 0));
        cg.set_text_output(false);
        CHECK(cg.generate_matrix(100, 100) && cg.generate_rhs() && !cg.generate_matrix(100, 99), "generate mode");
        for (int k : {0, 1, 4}) CHECK(cg.generate_reference_system(211, 42, k), "reference streams k=%d", k);
        const LAM::ReferenceSystemStreams s1 = LAM::reference_system_streams(50, 9, 3), s2 = LAM::reference_system_streams(50, 9, 3);
        CHECK(s1.eig == s2.eig && s1.rhs == s2.rhs && s1.reflectors == s2.reflectors && s1.reflectors.size() == 150, "seeded streams are reproducible");
        for (double d : s1.eig) CHECK(d >= std::exp(-3.5) && d <= std::exp(3.5), "spectrum exp(3.5 U[-1,1])");
    }
}

static int run_loaders(const char *tmpdir)
{
    loaders_for_type<double>(tmpdir, "f64");
    loaders_for_type<float>(tmpdir, "f32");
    // the header rule on its own (ConjugateGradient_HIP_base.hpp parse_bin_header)
    uint64_t rows = 0, cols = 0;
    const uint64_t h1[2] = {3, 3}, h2[2] = {(5ull << 32) | 3, (9ull << 32) | 3}, h3[2] = {~0ull, ~0ull}, h4[2] = {0, 5};
    CHECK(LAM::parse_bin_header(h1, 16 + 72, 8, &rows, &cols) && rows == 3 && cols == 3, "plain header");
    CHECK(LAM::parse_bin_header(h2, 16 + 72, 8, &rows, &cols) && rows == 3 && cols == 3, "masked header");
    CHECK(!LAM::parse_bin_header(h1, 16 + 71, 8, &rows, &cols) && !LAM::parse_bin_header(h3, 1u << 20, 8, &rows, &cols) &&
          !LAM::parse_bin_header(h4, 1u << 20, 8, &rows, &cols) && !LAM::parse_bin_header(h1, 8, 8, &rows, &cols), "bad headers");
    return g_fail;
}

// ---- bootstrap -------------------------------------------------------------------------------------------------------------
static int run_bootstrap()
{
    lam_bootstrap::Launch L;
    int argc = 0;
    char **argv = nullptr;
    if (!lam_bootstrap::init(&argc, &argv, L)) { fprintf(stderr, "bootstrap failed\n"); return 1; }
    printf("%d %d %d ", L.rank, L.size, L.local_rank);
    for (int i = 0; i < LAM_HIP_UNIQUE_ID_BYTES; i++) printf("%02x", (unsigned char)L.unique_id[i]);
    printf("\n");
    // communicator_ready() may only be called once every rank has read the id -- in the product ncclCommInitRank, a collective, sits
    // in between.  Its stand-in here: every other rank leaves an acknowledgement file, rank 0 waits for all of them.
    if (L.size > 1) {
        if (L.rank != 0) {
            FILE *f = fopen((L.id_file + ".ack" + std::to_string(L.rank)).c_str(), "w");
            if (f) fclose(f);
        } else {
            for (int q = 1; q < L.size; q++) {
                const std::string ack = L.id_file + ".ack" + std::to_string(q);
                int tries = 0;
                while (access(ack.c_str(), F_OK) != 0 && tries++ < 3000) usleep(10000);
                unlink(ack.c_str());
            }
        }
    }
    lam_bootstrap::communicator_ready(L);
    lam_bootstrap::finalize(L);
    return 0;
}

int main(int argc, char **argv)
{
    if (argc >= 2 && !strcmp(argv[1], "plan")) return run_plan() ? 1 : (printf("ok plan\n"), 0);
    if (argc >= 3 && !strcmp(argv[1], "loaders")) return run_loaders(argv[2]) ? 1 : (printf("ok loaders\n"), 0);
    if (argc >= 2 && !strcmp(argv[1], "bootstrap")) return run_bootstrap();
    fprintf(stderr, "usage: %s plan | loaders <tmpdir> | bootstrap\n", argv[0]);
    return 2;
}
