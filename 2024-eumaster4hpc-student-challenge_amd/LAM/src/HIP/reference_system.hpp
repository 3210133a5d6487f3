// reference_system.hpp -- the random numbers of the reference's fixture generator, in its order.
//
// /root/reference/challenge/main/random_spd_system.cpp draws everything from srand(seed_x) / rand():
//   random_matrix(Q, N, N, seed)      :27-37,76   column-major fill, value = 2 rand() / RAND_MAX - 1
//   random_matrix(D, N, 1, seed - 10) :83-87      then D[i] = exp(3.5 D[i])          (the spectrum, cond ~ 1.1e3)
//   random_matrix(rhs, N, 1, seed+10) :166
// and builds A = Q diag(D) Q^T after orthonormalising Q with MKL.  Here the same three streams give: the eigenvalues, the
// right hand side, and -- from the first k columns of that random matrix -- the vectors of the k Householder reflectors whose
// product stands in for Q (lam_hip_generate_spectrum_spd).  Same libc, same numbers as the reference would draw on this host.
#pragma once

#include <cmath>
#include <cstddef>
#include <cstdlib>
#include <vector>

namespace LAM
{
struct ReferenceSystemStreams {
    std::vector<double> eig;         // N
    std::vector<double> reflectors;  // k x N
    std::vector<double> rhs;         // N
};

inline ReferenceSystemStreams reference_system_streams(size_t n, int seed, int k = 4)
{
    ReferenceSystemStreams s;
    auto draw = [] { return ((2.0 * rand()) / RAND_MAX) - 1.0; };
    s.reflectors.resize((size_t)k * n);
    srand(seed);
    for (auto &x : s.reflectors) x = draw();
    s.eig.resize(n);
    srand(seed - 10);
    for (auto &d : s.eig) d = std::exp(3.5 * draw());
    s.rhs.resize(n);
    srand(seed + 10);
    for (auto &x : s.rhs) x = draw();
    return s;
}
}  // namespace LAM
