// ref_float_harness.cpp -- TEST INFRASTRUCTURE (oracle/): runs the REFERENCE's own CPU solver class instantiated with float.
//
// The reference's drivers hard-code <double> (challenge/main/test/test_CG_CPU_OMP.cpp:41), but its CPU classes are generic in
// FloatingType (LAM/src/CPU/ConjugateGradient_CPU_OMP.hpp:49-91,219-263) and its GPU classes are instantiated with float
// (LAM/src/GPU/distributed/ConjugateGradient_MultiGPUS_CUDA_NCCL.cu:767, ..._CUDA_MPI.cu:707,
// LAM/src/GPU/local/ConjugateGradient_MultiGPUS_CUDA.cu:539).  This file is OUR code; the class comes from the reference's
// header, included from where it lies (-I /root/reference/challenge/main/LAM/include, see oracle/Makefile `ref`); the output
// binary goes to oracle/_ref/.  tests/golden/make_golden.py runs it at OMP_NUM_THREADS=1 to capture float fixtures, which pin
// oracle_cg_solve_f32 (tests/test_oracle_golden.py) -- the oracle the fp32 / bf16 GPU parity tests compare with.
//
//   usage: ref_float_harness.out matrix.bin rhs.bin sol.bin max_iters rel_error     (files hold float elements)
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <iostream>

#include "LAM.hpp"     // the reference's umbrella header (relies on the includes above, like its own drivers)

int main(int argc, char **argv)
{
    if (argc != 6) { fprintf(stderr, "usage: %s matrix.bin rhs.bin sol.bin max_iters rel_error\n", argv[0]); return 64; }
    LAM::ConjugateGradient_CPU_OMP<float> cg;
    if (!cg.load_matrix_from_file(argv[1])) return 1;
    if (!cg.load_rhs_from_file(argv[2])) return 2;
    cg.solve(atoi(argv[4]), (float)atof(argv[5]));       // prints "Converged in K iterations, relative error is E"
    return cg.save_result_to_file(argv[3]) ? 0 : 6;
}
