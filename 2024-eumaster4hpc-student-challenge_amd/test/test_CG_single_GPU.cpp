// test_CG_single_GPU.out -- one MI355X.  Drop-in for the reference executable of the same name
// (/root/reference/challenge/main/test/test_CG_single_GPU.cpp, target test/CMakeLists.txt:25).
#include "LAM.hpp"
#include "positional_driver.hpp"

int main(int argc, char **argv)
{
    LAM::ConjugateGradient_HIP<double> cg(0);
    return run_positional_driver(argc, argv, cg, "LAM HIP (1 GPU)");
}
