// One process per GPU with RCCL over xGMI: drop-in for LAM::ConjugateGradient_MultiGPUS_CUDA_NCCL
// and ..._CUDA_MPI (/root/reference/challenge/main/LAM/src/GPU/distributed/
// ConjugateGradient_MultiGPUS_CUDA_NCCL.cuh:24-88).  The reference bootstraps NCCL with
// ncclGetUniqueId + MPI_Bcast + ncclCommInitRank inside solve() (NCCL.cu:306-334); here the launcher
// hands rank / world size / local device to the constructor together with the 128-byte unique id
// (see test/lam_bootstrap.hpp for the MPI-free exchange the drivers use), and the communicator is
// created once, timed into the CSV's extra column like the reference does.
#ifndef LAM_CONJUGATEGRADIENT_MULTIGPUS_HIP_RCCL_HPP
#define LAM_CONJUGATEGRADIENT_MULTIGPUS_HIP_RCCL_HPP

#include <cstring>

#include "ConjugateGradient_HIP_base.hpp"

namespace LAM
{

template <typename FloatingType>
class ConjugateGradient_MultiGPUS_HIP_RCCL : public ConjugateGradient_HIP_base<FloatingType>
{
  public:
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

  protected:
    bool create_context(lam_hip_ctx **out) override
    {
        return lam_hip_create_rank(out, _bf16 ? LAM_HIP_BF16 : this->dtype(), _device, this->_rank, _num_ranks, _id) == 0;
    }

  private:
    int _num_ranks, _device;
    bool _bf16;
    char _id[LAM_HIP_UNIQUE_ID_BYTES];
};

}  // namespace LAM
#endif
