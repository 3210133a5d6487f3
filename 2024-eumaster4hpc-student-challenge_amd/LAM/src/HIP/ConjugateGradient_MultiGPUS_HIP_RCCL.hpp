// One process per GPU with RCCL over xGMI: drop-in for LAM::ConjugateGradient_MultiGPUS_CUDA_NCCL
// and ..._CUDA_MPI (/root/reference/challenge/main/LAM/src/GPU/distributed/
// ConjugateGradient_MultiGPUS_CUDA_NCCL.cuh:24-88).  The reference bootstraps NCCL with
// ncclGetUniqueId + MPI_Bcast + ncclCommInitRank inside solve() (NCCL.cu:306-334); here the class asks
// the launcher at first use (lam_bootstrap.hpp: MPI, or the launcher's environment + a rendezvous file)
// or takes rank / world size / device / id from the explicit constructor, and the communicator is
// created once, timed into the CSV's extra column like the reference does.
#ifndef LAM_CONJUGATEGRADIENT_MULTIGPUS_HIP_RCCL_HPP
#define LAM_CONJUGATEGRADIENT_MULTIGPUS_HIP_RCCL_HPP

#include <cstring>

#include "ConjugateGradient_HIP_base.hpp"
#include "lam_bootstrap.hpp"

namespace LAM
{

template <typename FloatingType>
class ConjugateGradient_MultiGPUS_HIP_RCCL : public ConjugateGradient_HIP_base<FloatingType>
{
  public:
    // Default-constructible like the reference class (`ConjugateGradient_MultiGPUS_CUDA_NCCL<double> CG_P;`,
    // test_CG_MultiGPUS_CUDA_NCCL.cpp:46,132): rank, world size, local device and the unique id are taken
    // from the launcher at first use (lam_bootstrap::attach -- MPI if the caller initialised it and this
    // was compiled with -DLAM_USE_MPI, else the launcher's environment + a rendezvous file), where the
    // reference asks MPI inside load_matrix_from_file / solve (NCCL.cu:320-327,502-514).
    ConjugateGradient_MultiGPUS_HIP_RCCL() : _from_launcher(true)
    {
        this->_print_csv = true;
        this->_comm_init_column = true;
        memset(_id, 0, sizeof _id);
    }
    // Explicit form for callers that have their own launcher glue.
    // bf16_storage (float instantiation only): keep the matrix in bf16, vectors and accumulation in fp32
    ConjugateGradient_MultiGPUS_HIP_RCCL(int rank, int num_ranks, int device, const void *unique_id, bool bf16_storage = false)
        : _num_ranks(num_ranks), _device(device), _bf16(bf16_storage && std::is_same<FloatingType, float>::value)
    {
        this->_rank = rank;
        this->_print_csv = true;
        this->_comm_init_column = true;
        if (unique_id) memcpy(_id, unique_id, LAM_HIP_UNIQUE_ID_BYTES);
        else memset(_id, 0, sizeof _id);
    }

    int get_rank() const { return this->_rank; }
    int get_num_ranks() const { return _num_ranks; }

  protected:
    bool create_context(lam_hip_ctx **out) override
    {
        lam_bootstrap::Launch L;
        if (_from_launcher) {
            int ndev = 0;
            if (lam_hip_device_count(&ndev) != 0 || ndev <= 0) return false;
            if (!lam_bootstrap::attach(L)) return false;
            this->_rank = L.rank;
            _num_ranks = L.size;
            _device = L.local_rank % ndev;
            memcpy(_id, L.unique_id, LAM_HIP_UNIQUE_ID_BYTES);
        }
        const bool ok = lam_hip_create_rank(out, _bf16 ? LAM_HIP_BF16 : this->dtype(), _device, this->_rank, _num_ranks, _id) == 0;
        if (_from_launcher) lam_bootstrap::communicator_ready(L);   // every rank has read the id by now
        return ok;
    }

  private:
    int _num_ranks = 1, _device = 0;
    bool _bf16 = false, _from_launcher = false;
    char _id[LAM_HIP_UNIQUE_ID_BYTES];
};

}  // namespace LAM
#endif
