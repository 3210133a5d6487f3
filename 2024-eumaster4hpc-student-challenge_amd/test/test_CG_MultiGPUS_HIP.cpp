// test_CG_MultiGPUS_HIP.out -- one process, every GPU of the node (xGMI peer stores).  Counterpart
// of the reference's test_CG_MultiGPUS_CUDA.out (/root/reference/challenge/main/test/
// test_CG_MultiGPUS_CUDA.cpp, target test/CMakeLists.txt:31).  The number of row shards can be
// forced with LAM_NUM_SHARDS (shards are dealt round-robin over the visible devices).
#include <cstdlib>
#include <vector>

#include "LAM.hpp"
#include "positional_driver.hpp"

int main(int argc, char **argv)
{
    const char *env = getenv("LAM_NUM_SHARDS");
    if (env && atoi(env) > 0) {
        int ndev = 0;
        if (lam_hip_device_count(&ndev) != 0 || ndev <= 0) {
            fprintf(stderr, "No GPU: %s\n", lam_hip_last_error(nullptr));
            return 1;
        }
        std::vector<int> devs;
        for (int q = 0; q < atoi(env); q++) devs.push_back(q % ndev);
        LAM::ConjugateGradient_MultiGPUS_HIP<double> cg(devs);
        return run_positional_driver(argc, argv, cg, "LAM HIP (multi-GPU, 1 process)");
    }
    LAM::ConjugateGradient_MultiGPUS_HIP<double> cg;
    return run_positional_driver(argc, argv, cg, "LAM HIP (multi-GPU, 1 process)");
}
