// Umbrella header, like /root/reference/challenge/main/LAM/include/LAM.hpp: the interface always, the
// GPU classes behind a build flag -- USE_HIP here where the reference has USE_CUDA (:8-13).  There is
// no CPU variant in the product (the CPU restatement used by the tests lives outside the package and is
// never linked).  With USE_HIP the reference's own class names are available too (reference_names.hpp).
#ifndef LinearAlgebraMI355X_HPP
#define LinearAlgebraMI355X_HPP

#include "../src/ConjugateGradient.hpp"

#ifdef USE_HIP
#include "../src/HIP/ConjugateGradient_HIP.hpp"
#include "../src/HIP/ConjugateGradient_MultiGPUS_HIP.hpp"
#include "../src/HIP/ConjugateGradient_MultiGPUS_HIP_RCCL.hpp"
#include "../src/HIP/reference_names.hpp"
#endif

#endif
