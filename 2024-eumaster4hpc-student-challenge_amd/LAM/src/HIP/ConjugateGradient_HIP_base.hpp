// Shared implementation of the HIP-backed LAM solver classes: everything above the C ABI
// (include/lam_hip.h) that the reference keeps inside each of its GPU classes -- file I/O in the
// reference's binary format, the row partition, diagnostics on stderr, the CSV fragments printed by
// generate_matrix/solve -- lives here once.  Plain C++17, compiled by g++; no HIP headers needed.
//
// Reference behaviour mirrored (paths under /root/reference/challenge/main/LAM/src/):
//   load_matrix_from_file  CPU/ConjugateGradient_CPU_OMP.hpp:137-197 (header, square check, messages),
//                          GPU/distributed/ConjugateGradient_MultiGPUS_CUDA_NCCL.cu:489-585 (each
//                          owner reads its own row block; here from a read-only mmap, 64-bit counts)
//   load_rhs_from_file     CPU/ConjugateGradient_CPU_MPI_OMP.hpp:258-305
//   save_result_to_file    CPU/ConjugateGradient_CPU_OMP.hpp:199-217 -- writes x (the MPI variant
//                          writes _rhs by mistake, :439; the GPU classes are meant to write x); the
//                          cols word is written as a clean 64-bit 1
//   generate_matrix/rhs    CPU/ConjugateGradient_CPU_MPI_OMP.hpp:144-256 (prints "N," on rank 0 :203-205)
//   solve                  CPU/ConjugateGradient_CPU_MPI_OMP.hpp:71-142; CSV chunk
//                          "avg_gemv,avg_iter,num_iters,err," as the GPU variants print it
//                          (per-iteration averages divided by num_iters ONCE,
//                          ConjugateGradient_MultiGPUS_CUDA_NCCL.cu:424-427)
#ifndef LAM_CONJUGATEGRADIENT_HIP_BASE_HPP
#define LAM_CONJUGATEGRADIENT_HIP_BASE_HPP

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <iostream>
#include <type_traits>
#include <vector>

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include "../../../../include/lam_hip.h"
#include "../ConjugateGradient.hpp"
#include "reference_system.hpp"

// bytes one lam_hip_upload_rows call carries at most (tests/host_asan builds the loaders with a few KiB to walk the chunk loop)
#ifndef LAM_LOADER_CHUNK_BYTES
#define LAM_LOADER_CHUNK_BYTES (1ull << 30)
#endif

namespace LAM
{

// Header of a matrix / vector file (random_spd_system.cpp:105-121: two 64-bit words, rows and cols).
// The reference's own writers store an `int` with sizeof(size_t) -- cols in ConjugateGradient_CPU_OMP.hpp:
// 206-210, BOTH words in ConjugateGradient_MultiGPUS_CUDA_NCCL.cu:754-757 -- which leaves stack garbage
// in the upper 32 bits.  One rule for every loader: take the two words as they are if the file is long
// enough for them, otherwise their low 32 bits if THOSE fit the file, otherwise the file is bad.
inline bool parse_bin_header(const uint64_t hdr[2], uint64_t file_bytes, uint64_t elem_bytes, uint64_t *rows, uint64_t *cols)
{
    auto fits = [&](uint64_t r, uint64_t c) {
        if (r == 0 || c == 0 || file_bytes < 16) return false;
        const unsigned __int128 need = (unsigned __int128)r * c * elem_bytes;
        return need <= (unsigned __int128)(file_bytes - 16);
    };
    if (fits(hdr[0], hdr[1])) { *rows = hdr[0]; *cols = hdr[1]; return true; }
    const uint64_t r = hdr[0] & 0xffffffffull, c = hdr[1] & 0xffffffffull;
    if (fits(r, c)) { *rows = r; *cols = c; return true; }
    return false;
}

template <typename FloatingType>
class ConjugateGradient_HIP_base : public ConjugateGradient<FloatingType>
{
    static_assert(std::is_same<FloatingType, double>::value || std::is_same<FloatingType, float>::value,
                  "the HIP classes are instantiated for double and float");

  public:
    ~ConjugateGradient_HIP_base() override
    {
        if (_ctx) lam_hip_destroy(_ctx);
    }

    bool solve(int max_iters, FloatingType rel_error) override
    {
        if (!ensure_ctx()) return false;
        lam_hip_stats st;
        if (lam_hip_solve(_ctx, max_iters, (double)rel_error, &st) != 0) return report("solve");
        _stats = st;
        if (_print_csv && is_root()) {
            if (_comm_init_column) std::cout << st.t_comm_init << ",";
            // the reference's t_gemv column includes its broadcast + gather (ConjugateGradient_MultiGPUS_CUDA_NCCL.cu:352-377);
            // here the default column is the GEMV kernel alone and gemv_plus_comm adds the exchange step(s) to it
            std::cout << (_gemv_plus_comm ? st.t_gemv + st.t_exchange : st.t_gemv) << "," << st.t_iter << "," << st.num_iters << "," << st.rel_err << ",";
        }
        if (_print_text && is_root()) {
            if (st.converged)
                printf("Converged in %d iterations, relative error is %e\n", st.num_iters, st.rel_err);
            else
                printf("Did not converge in %d iterations, relative error is %e\n", max_iters, st.rel_err);
        }
        return st.converged != 0;
    }

    bool load_matrix_from_file(const char *filename) override
    {
        if (!ensure_ctx()) return false;
        // Whatever happens to THIS rank's part of the file, every rank reaches the agreement below: a rank
        // whose block failed must not leave the others waiting in the solve's first collective.
        const bool ok = agree(load_matrix_local(filename));
        if (!ok && is_root()) fprintf(stderr, "Failed to read matrix rows\n");
        return ok;
    }

    bool load_rhs_from_file(const char *filename) override
    {
        if (!ensure_ctx()) return false;
        return agree(load_rhs_local(filename));
    }

  private:
    bool load_matrix_local(const char *filename)
    {
        int fd = open(filename, O_RDONLY);
        if (fd < 0) {
            file_error("Cannot open output file\n");   // the reference's wording
            return false;
        }
        uint64_t hdr[2], rows = 0, cols = 0;
        struct stat sb;
        if (pread(fd, hdr, sizeof hdr, 0) != (ssize_t)sizeof hdr || fstat(fd, &sb) != 0) {
            file_error("Cannot read matrix header\n");
            close(fd);
            return false;
        }
        if (!parse_bin_header(hdr, (uint64_t)sb.st_size, sizeof(FloatingType), &rows, &cols)) {
            file_error("Matrix file is shorter than its header says\n");
            close(fd);
            return false;
        }
        if (rows != cols) {
            file_error("Matrix has to be square\n");
            close(fd);
            return false;
        }
        if (lam_hip_set_problem(_ctx, rows) != 0) { close(fd); return report("set_problem"); }
        _num_rows = rows;
        _num_cols = cols;
        if (_print_csv && is_root()) std::cout << rows << ",";
        // Every locally owned shard takes its own row block straight from a read-only mapping of the
        // file (no intermediate read buffer), in chunks of <= 1 GiB so that a single upload never
        // exceeds 2^31 elements -- the reference's MPI_File_read count is an int (CPU_MPI_OMP.hpp:408).
        int total = 0, local = 0;
        lam_hip_num_shards(_ctx, &total, &local);
        const uint64_t file_bytes = 16 + rows * cols * sizeof(FloatingType);
        void *map = mmap(nullptr, file_bytes, PROT_READ, MAP_PRIVATE, fd, 0);
        if (map == MAP_FAILED) {
            file_error("Cannot map matrix file\n");
            close(fd);
            return false;
        }
        (void)madvise(map, file_bytes, MADV_SEQUENTIAL);
        const FloatingType *data = reinterpret_cast<const FloatingType *>(static_cast<const char *>(map) + 16);
        const uint64_t chunk_rows = std::max<uint64_t>(1, (uint64_t)(LAM_LOADER_CHUNK_BYTES) / (cols * sizeof(FloatingType)));
        bool ok = true;
        for (int q = 0; q < total && ok; q++) {
            if (local != total && q != _rank) continue;
            uint64_t r0 = 0, nr = 0;
            lam_hip_get_partition(_ctx, q, &r0, &nr);
            for (uint64_t r = r0; r < r0 + nr && ok; r += chunk_rows) {
                const uint64_t n = std::min(chunk_rows, r0 + nr - r);
                if (lam_hip_upload_rows(_ctx, r, n, data + r * cols) != 0) { report("upload_rows"); ok = false; }
            }
        }
        munmap(map, file_bytes);
        close(fd);
        return ok;
    }

    bool load_rhs_local(const char *filename)
    {
        FILE *file = fopen(filename, "rb");
        if (file == nullptr) {
            file_error("Cannot open output file\n");
            return false;
        }
        uint64_t hdr[2] = {0, 0}, rows = 0, cols = 0;
        struct stat sb;
        if (fread(hdr, sizeof(uint64_t), 2, file) != 2 || fstat(fileno(file), &sb) != 0) { fclose(file); return false; }
        if (!parse_bin_header(hdr, (uint64_t)sb.st_size, sizeof(FloatingType), &rows, &cols)) {
            file_error("Right hand side file is shorter than its header says\n");
            fclose(file);
            return false;
        }
        if (cols != 1) {
            file_error("Right hand side has to have just a single column\n");
            fclose(file);
            return false;
        }
        if (rows != _num_cols) {
            file_error("Size of right hand side does not match the matrix\n");
            fclose(file);
            return false;
        }
        std::vector<FloatingType> b(rows);
        bool ok = fread(b.data(), sizeof(FloatingType), rows, file) == rows;
        fclose(file);
        if (ok && lam_hip_set_rhs(_ctx, b.data()) != 0) { report("set_rhs"); ok = false; }
        return ok;
    }

  public:

    bool save_result_to_file(const char *filename) const override
    {
        if (!_ctx) return false;
        std::vector<FloatingType> x(_num_cols);
        // collective in rank mode: every rank calls, only the root writes
        if (lam_hip_get_solution(_ctx, x.data()) != 0) return report("get_solution");
        if (!is_root()) return true;
        FILE *file = fopen(filename, "wb");
        if (file == nullptr) {
            fprintf(stderr, "Cannot open output file\n");
            return false;
        }
        const uint64_t hdr[2] = {(uint64_t)_num_cols, 1};
        const bool ok = fwrite(hdr, sizeof(uint64_t), 2, file) == 2 &&
                        fwrite(x.data(), sizeof(FloatingType), x.size(), file) == x.size();
        fclose(file);
        return ok;
    }

    // generate mode of the distributed classes: dense tridiag(1,2,1), b = 1
    virtual bool generate_matrix(const size_t rows, const size_t cols)
    {
        if (!ensure_ctx()) return false;
        if (rows != cols) {
            if (is_root()) fprintf(stderr, "Matrix has to be square\n");
            return false;
        }
        if (lam_hip_set_problem(_ctx, rows) != 0) return report("set_problem");
        _num_rows = rows;
        _num_cols = cols;
        if (_print_csv && is_root()) std::cout << rows << ",";
        if (lam_hip_generate_tridiag(_ctx) != 0) return report("generate_tridiag");
        return true;
    }
    // extension: seeded dense random SPD system generated on the device (no reference counterpart)
    virtual bool generate_random_system(const size_t rows, uint64_t seed, double cond)
    {
        if (!ensure_ctx()) return false;
        if (lam_hip_set_problem(_ctx, rows) != 0) return report("set_problem");
        _num_rows = _num_cols = rows;
        if (_print_csv && is_root()) std::cout << rows << ",";
        if (lam_hip_generate_random_spd(_ctx, seed, cond) != 0) return report("generate_random_spd");
        if (lam_hip_generate_random_rhs(_ctx, seed + 1) != 0) return report("generate_random_rhs");
        return true;
    }
    // extension: the reference GENERATOR's system (challenge/main/random_spd_system.cpp: spectrum exp(3.5 U[-1,1]), rhs
    // U[-1,1], its srand/rand streams) built in place, without the files: A = H_k..H_1 diag(d) H_1..H_k (reference_system.hpp)
    virtual bool generate_reference_system(const size_t rows, int seed, int reflectors = 4)
    {
        if (!ensure_ctx()) return false;
        if (lam_hip_set_problem(_ctx, rows) != 0) return report("set_problem");
        _num_rows = _num_cols = rows;
        if (_print_csv && is_root()) std::cout << rows << ",";
        const ReferenceSystemStreams st = reference_system_streams(rows, seed, reflectors);
        if (lam_hip_generate_spectrum_spd(_ctx, st.eig.data(), st.reflectors.data(), reflectors) != 0) return report("generate_spectrum_spd");
        if constexpr (std::is_same<FloatingType, double>::value) {
            if (lam_hip_set_rhs(_ctx, st.rhs.data()) != 0) return report("set_rhs");
        } else {
            const std::vector<float> bf(st.rhs.begin(), st.rhs.end());
            if (lam_hip_set_rhs(_ctx, bf.data()) != 0) return report("set_rhs");
        }
        return true;
    }
    virtual bool generate_rhs()
    {
        if (!ensure_ctx()) return false;
        if (lam_hip_generate_rhs(_ctx, 1.0) != 0) return report("generate_rhs");
        return true;
    }

    // rows held by this process (all of them in the single-process classes), like the reference getters
    size_t get_num_rows() const
    {
        if (!_ctx || _num_rows == 0) return 0;
        int total = 0, local = 0;
        lam_hip_num_shards(_ctx, &total, &local);
        if (local == total) return _num_rows;
        uint64_t r0 = 0, nr = 0;
        lam_hip_get_partition(_ctx, _rank, &r0, &nr);
        return nr;
    }
    size_t get_num_cols() const { return _num_cols; }

    const lam_hip_stats &stats() const { return _stats; }
    lam_hip_ctx *context() { return ensure_ctx() ? _ctx : nullptr; }
    void set_csv_output(bool on) { _print_csv = on; }
    void set_text_output(bool on) { _print_text = on; }
    void set_comm_init_column(bool on) { _comm_init_column = on; }     // the NCCL variant's extra CSV column (0 without a communicator)
    // CSV: print t_gemv + t_exchange in the GEMV column (the reference's convention); also switched on by the environment
    // variable LAM_CSV_GEMV_PLUS_COMM=1 (for drivers compiled from the reference's own sources)
    void set_gemv_plus_comm(bool on) { _gemv_plus_comm = on; }

  protected:
    // derived classes create the context (which devices, which exchange)
    virtual bool create_context(lam_hip_ctx **out) = 0;

    bool ensure_ctx() const
    {
        if (_ctx) return true;
        // the stats struct and the option semantics belong to one ABI version: a library of another one is refused, not guessed at
        if (lam_hip_abi_version() != LAM_HIP_ABI_VERSION) {
            fprintf(stderr, "LAM HIP: liblam_hip.so has ABI %d, these headers were written for ABI %d\n", lam_hip_abi_version(), LAM_HIP_ABI_VERSION);
            return false;
        }
        auto *self = const_cast<ConjugateGradient_HIP_base *>(this);
        if (!self->create_context(&self->_ctx) || !_ctx) {
            fprintf(stderr, "LAM HIP: cannot create the GPU context: %s\n", lam_hip_last_error(nullptr));
            self->_ctx = nullptr;
            return false;
        }
        return true;
    }
    bool is_root() const { return _rank == 0; }
    // a file problem may exist on one rank only (its own row block): say it wherever it happens
    void file_error(const char *msg) const
    {
        if (is_root()) fputs(msg, stderr);
        else fprintf(stderr, "rank %d: %s", _rank, msg);
    }
    // true iff the step succeeded on EVERY rank (collective in the one-process-per-GPU class)
    bool agree(bool ok) const
    {
        int all = ok ? 1 : 0;
        if (_ctx && lam_hip_all_ok(_ctx, ok ? 1 : 0, &all) != 0) return false;
        return all != 0;
    }
    bool report(const char *what) const
    {
        fprintf(stderr, "LAM HIP: %s failed: %s\n", what, lam_hip_last_error(_ctx));
        return false;
    }
    static constexpr int dtype() { return std::is_same<FloatingType, double>::value ? LAM_HIP_F64 : LAM_HIP_F32; }

    mutable lam_hip_ctx *_ctx = nullptr;
    size_t _num_rows = 0, _num_cols = 0;
    int _rank = 0;
    bool _print_csv = false;          // the getopt-style drivers' CSV fragments
    bool _print_text = false;         // the positional drivers' "Converged in ..." line
    bool _comm_init_column = false;   // extra column of the NCCL variant
    bool _gemv_plus_comm = [] { const char *v = getenv("LAM_CSV_GEMV_PLUS_COMM"); return v && *v && *v != '0'; }();
    lam_hip_stats _stats{};
};

}  // namespace LAM
#endif
