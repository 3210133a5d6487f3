// One process driving every GPU of the node: drop-in for LAM::ConjugateGradient_MultiGPUS_CUDA
// (/root/reference/challenge/main/LAM/src/GPU/local/ConjugateGradient_MultiGPUS_CUDA.cuh:14-36,
// which discovers the devices in its constructor, :20-22).  The matrix is row-sharded over the
// devices; p slices and partial dot products travel by direct peer stores over xGMI.
#ifndef LAM_CONJUGATEGRADIENT_MULTIGPUS_HIP_HPP
#define LAM_CONJUGATEGRADIENT_MULTIGPUS_HIP_HPP

#include <vector>

#include "ConjugateGradient_HIP_base.hpp"

namespace LAM
{

template <typename FloatingType>
class ConjugateGradient_MultiGPUS_HIP : public ConjugateGradient_HIP_base<FloatingType>
{
  public:
    // num_devices <= 0: use every visible GPU.  `devices` may repeat an id (several shards per GPU).
    explicit ConjugateGradient_MultiGPUS_HIP(int num_devices = 0) : _num_devices(num_devices) { this->_print_text = true; }
    explicit ConjugateGradient_MultiGPUS_HIP(const std::vector<int> &devices)
        : _num_devices((int)devices.size()), _devices(devices) { this->_print_text = true; }

    int get_num_devices() const { return _num_devices; }

  protected:
    bool create_context(lam_hip_ctx **out) override
    {
        if (_num_devices <= 0) {
            if (lam_hip_device_count(&_num_devices) != 0 || _num_devices <= 0) return false;
        }
        return lam_hip_create(out, this->dtype(), _num_devices, _devices.empty() ? nullptr : _devices.data()) == 0;
    }

  private:
    int _num_devices;
    std::vector<int> _devices;
};

}  // namespace LAM
#endif
