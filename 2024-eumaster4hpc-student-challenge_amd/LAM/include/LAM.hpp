// Umbrella header, like /root/reference/challenge/main/LAM/include/LAM.hpp.  The HIP classes are
// the whole library here: there is no CPU variant in the product (the CPU restatement used by the
// tests is kept outside the package and never linked).
#ifndef LinearAlgebraMI355X_HPP
#define LinearAlgebraMI355X_HPP

#include "../src/ConjugateGradient.hpp"
#include "../src/HIP/ConjugateGradient_HIP.hpp"
#include "../src/HIP/ConjugateGradient_MultiGPUS_HIP.hpp"
#include "../src/HIP/ConjugateGradient_MultiGPUS_HIP_RCCL.hpp"

#endif
