// Single-GPU solver: drop-in for LAM::ConjugateGradient_GPU_CUDA
// (/root/reference/challenge/main/LAM/src/GPU/local/ConjugateGradient_GPU_CUDA.cuh:14-31).
#ifndef LAM_CONJUGATEGRADIENT_HIP_HPP
#define LAM_CONJUGATEGRADIENT_HIP_HPP

#include "ConjugateGradient_HIP_base.hpp"

namespace LAM
{

template <typename FloatingType>
class ConjugateGradient_HIP : public ConjugateGradient_HIP_base<FloatingType>
{
  public:
    explicit ConjugateGradient_HIP(int device = 0) : _device(device) { this->_print_text = true; }

  protected:
    bool create_context(lam_hip_ctx **out) override
    {
        return lam_hip_create(out, this->dtype(), 1, &_device) == 0;
    }

  private:
    int _device;
};

// the reference's name for this variant keeps working
template <typename FloatingType>
using ConjugateGradient_GPU_HIP = ConjugateGradient_HIP<FloatingType>;

}  // namespace LAM
#endif
