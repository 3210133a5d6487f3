// The solver interface every LAM Conjugate-Gradient variant implements.  Same surface as the
// reference's abstract class (/root/reference/challenge/main/LAM/src/ConjugateGradient.hpp:9-28):
// four virtuals, all returning bool (true = ok; solve: true = converged within max_iters), so the
// reference's drivers compile against these classes unchanged.
#ifndef LAM_CONJUGATEGRADIENT_HPP
#define LAM_CONJUGATEGRADIENT_HPP

#include <type_traits>

namespace LAM
{

template <typename FloatingType>
class ConjugateGradient
{
    static_assert(std::is_floating_point<FloatingType>::value, "DataType must be floating point");

  public:
    ConjugateGradient() = default;
    virtual ~ConjugateGradient() = default;

    virtual bool solve(int max_iters, FloatingType rel_error) = 0;

    virtual bool load_matrix_from_file(const char *filename) = 0;
    virtual bool load_rhs_from_file(const char *filename) = 0;
    virtual bool save_result_to_file(const char *filename) const = 0;
};

}  // namespace LAM
#endif
