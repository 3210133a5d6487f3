// The reference's class names and helper macros, so that ITS drivers compile against these headers
// unchanged (/root/reference/challenge/main/test/test_CG_MultiGPUS_CUDA_NCCL.cpp, _CUDA_MPI.cpp,
// test_CG_single_GPU.cpp, test_CG_MultiGPUS_CUDA.cpp -- tests/test_dropin_reference_drivers.py builds them
// from where they lie, with -DUSE_HIP, and runs them).
//   ConjugateGradient_MultiGPUS_CUDA_NCCL / _CUDA_MPI  -> one process per GPU, RCCL over xGMI
//       (LAM/src/GPU/distributed/ConjugateGradient_MultiGPUS_CUDA_NCCL.cuh:24-29, ..._CUDA_MPI.cuh)
//   ConjugateGradient_MultiGPUS_CUDA                    -> one process, all GPUs (LAM/src/GPU/local/
//       ConjugateGradient_MultiGPUS_CUDA.cuh:14-22)
//   ConjugateGradient_GPU_CUDA                          -> one GPU (LAM/src/GPU/local/ConjugateGradient_GPU_CUDA.cuh)
//   PRINT_RANK0 / PRINT_ERR_RANK0: the drivers use these macros, which the reference defines in its class
//       headers (ConjugateGradient_MultiGPUS_CUDA_NCCL.cuh:17-18, ConjugateGradient_CPU_MPI_OMP.hpp:11-12).
#ifndef LAM_REFERENCE_NAMES_HPP
#define LAM_REFERENCE_NAMES_HPP

#include <cstdio>

#include "ConjugateGradient_HIP.hpp"
#include "ConjugateGradient_MultiGPUS_HIP.hpp"
#include "ConjugateGradient_MultiGPUS_HIP_RCCL.hpp"

#ifndef PRINT_RANK0
#define PRINT_RANK0(...) if(rank==0) printf(__VA_ARGS__)
#endif
#ifndef PRINT_ERR_RANK0
#define PRINT_ERR_RANK0(...) if(rank==0) fprintf(stderr, __VA_ARGS__)
#endif

namespace LAM
{
template <typename FloatingType> using ConjugateGradient_MultiGPUS_CUDA_NCCL = ConjugateGradient_MultiGPUS_HIP_RCCL<FloatingType>;
template <typename FloatingType> using ConjugateGradient_MultiGPUS_CUDA_MPI = ConjugateGradient_MultiGPUS_HIP_RCCL<FloatingType>;
template <typename FloatingType> using ConjugateGradient_MultiGPUS_CUDA = ConjugateGradient_MultiGPUS_HIP<FloatingType>;
template <typename FloatingType> using ConjugateGradient_GPU_CUDA = ConjugateGradient_HIP<FloatingType>;
}  // namespace LAM
#endif
