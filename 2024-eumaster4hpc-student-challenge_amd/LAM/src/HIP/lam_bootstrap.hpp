// Launcher glue for the one-process-per-GPU driver: who am I, and how do the ranks agree on the
// 128-byte RCCL unique id.  The reference does this with MPI_Comm_rank/size + MPI_Bcast
// (/root/reference/challenge/main/LAM/src/GPU/distributed/ConjugateGradient_MultiGPUS_CUDA_NCCL.cu:
// 320-327) and picks the device by hashing host names (:502-514).  MPI is not guaranteed next to an
// MI355X node, so the default is launcher-agnostic: rank/size/local rank from the environment
// (torchrun: RANK/WORLD_SIZE/LOCAL_RANK; MPICH/hydra: PMI_RANK/PMI_SIZE/MPI_LOCALRANKID; Open MPI:
// OMPI_COMM_WORLD_*; Slurm: SLURM_PROCID/SLURM_NTASKS/SLURM_LOCALID) and the id through a file that
// rank 0 publishes atomically (single node; the path can be put on a shared filesystem).  Compile
// with -DLAM_USE_MPI to use MPI_Bcast instead (taken whenever MPI has been initialised).
#ifndef LAM_BOOTSTRAP_HPP
#define LAM_BOOTSTRAP_HPP

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>

#include <fcntl.h>
#include <unistd.h>

#include "../../../../include/lam_hip.h"

#ifdef LAM_USE_MPI
#include <mpi.h>
#endif

namespace lam_bootstrap
{

struct Launch {
    int rank = 0, size = 1, local_rank = 0;
    char unique_id[LAM_HIP_UNIQUE_ID_BYTES] = {0};
    std::string id_file;   // rendezvous file (rank 0 removes it once the communicator exists)
};

inline int env_int(std::initializer_list<const char *> names, int dflt)
{
    for (const char *n : names) {
        const char *v = getenv(n);
        if (v && *v) return atoi(v);
    }
    return dflt;
}

// Rendezvous file of this launch.  The key must be the same on all ranks of ONE launch and differ
// between launches: LAM_RCCL_ID_FILE / LAM_JOB_ID if the caller sets them, else the launcher's
// job id or port plus the pid of the launcher process, which all local ranks share as their parent.
inline std::string id_file_path()
{
    if (const char *p = getenv("LAM_RCCL_ID_FILE")) return p;
    const char *job = getenv("LAM_JOB_ID");
    if (!job) job = getenv("MASTER_PORT");
    if (!job) job = getenv("SLURM_JOB_ID");
    if (!job) job = getenv("PMI_ID");
    return std::string("/tmp/lam_rccl_id.") + (job ? job : "default") + "." + std::to_string((long)getppid());
}

// rank / size / local rank and the unique id, WITHOUT initialising MPI: for a class that is
// default-constructed inside somebody else's driver (the reference's drivers call MPI_Init themselves,
// test_CG_MultiGPUS_CUDA_NCCL.cpp:207-209).  With -DLAM_USE_MPI and MPI already initialised the id
// travels by MPI_Bcast (ConjugateGradient_MultiGPUS_CUDA_NCCL.cu:320-327); otherwise through the
// environment + the rendezvous file.
inline bool attach(Launch &L);

inline bool init(int *argc, char ***argv, Launch &L)
{
#ifdef LAM_USE_MPI
    MPI_Init(argc, argv);
#else
    (void)argc; (void)argv;
#endif
    return attach(L);
}

inline bool attach(Launch &L)
{
#ifdef LAM_USE_MPI
    int mpi_up = 0;
    MPI_Initialized(&mpi_up);
    if (mpi_up) {
    MPI_Comm_rank(MPI_COMM_WORLD, &L.rank);
    MPI_Comm_size(MPI_COMM_WORLD, &L.size);
    MPI_Comm local;
    MPI_Comm_split_type(MPI_COMM_WORLD, MPI_COMM_TYPE_SHARED, L.rank, MPI_INFO_NULL, &local);
    MPI_Comm_rank(local, &L.local_rank);
    MPI_Comm_free(&local);
    if (L.size > 1) {
        if (L.rank == 0 && lam_hip_get_unique_id(L.unique_id) != 0) return false;
        MPI_Bcast(L.unique_id, LAM_HIP_UNIQUE_ID_BYTES, MPI_BYTE, 0, MPI_COMM_WORLD);
    }
    return true;
    }
#endif
    L.rank = env_int({"RANK", "PMI_RANK", "OMPI_COMM_WORLD_RANK", "SLURM_PROCID"}, 0);
    L.size = env_int({"WORLD_SIZE", "PMI_SIZE", "OMPI_COMM_WORLD_SIZE", "SLURM_NTASKS"}, 1);
    L.local_rank = env_int({"LOCAL_RANK", "MPI_LOCALRANKID", "OMPI_COMM_WORLD_LOCAL_RANK", "SLURM_LOCALID"}, L.rank);
    if (L.size <= 1) return true;
    const std::string path = id_file_path();
    L.id_file = path;
    if (L.rank == 0) {
        if (lam_hip_get_unique_id(L.unique_id) != 0) return false;
        unlink(path.c_str());              // whatever an earlier launch left under this name is not ours
        // created exclusively and never through a symlink planted at the predictable name, readable by this user only
        const std::string tmp = path + "." + std::to_string((long)getpid()) + ".tmp";
        unlink(tmp.c_str());
        const int fd = open(tmp.c_str(), O_WRONLY | O_CREAT | O_EXCL | O_NOFOLLOW, 0600);
        if (fd < 0) return false;
        const bool written = write(fd, L.unique_id, LAM_HIP_UNIQUE_ID_BYTES) == (ssize_t)LAM_HIP_UNIQUE_ID_BYTES;
        if (close(fd) != 0 || !written) { unlink(tmp.c_str()); return false; }
        return rename(tmp.c_str(), path.c_str()) == 0;
    }
    for (int tries = 0; tries < 6000; tries++) {   // up to 60 s
        FILE *f = fopen(path.c_str(), "rb");
        if (f) {
            const size_t got = fread(L.unique_id, 1, LAM_HIP_UNIQUE_ID_BYTES, f);
            fclose(f);
            if (got == LAM_HIP_UNIQUE_ID_BYTES) return true;
        }
        std::this_thread::sleep_for(std::chrono::milliseconds(10));
    }
    fprintf(stderr, "rank %d: timed out waiting for %s\n", L.rank, path.c_str());
    return false;
}

// Call after the communicator has been created on this rank (ncclCommInitRank is collective, so by
// then every rank has read the id): rank 0 removes the rendezvous file so that a later job with the
// same id cannot pick up a stale one.
inline void communicator_ready(const Launch &L)
{
    if (L.rank == 0 && !L.id_file.empty()) unlink(L.id_file.c_str());
}

inline void finalize(const Launch &L)
{
    (void)L;
#ifdef LAM_USE_MPI
    int up = 0, down = 0;
    MPI_Initialized(&up);
    MPI_Finalized(&down);
    if (up && !down) MPI_Finalize();
#endif
}

}  // namespace lam_bootstrap
#endif
