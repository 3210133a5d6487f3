// permlane_semantics -- what v_permlane32_swap / v_permlane16_swap and the DPP controls used by wave_sum8 (lam_kernels.h) do on gfx950, printed
// lane by lane: the transposed butterfly was written against this output.  hipcc --offload-arch=gfx950 -O2 tools/permlane_semantics.hip
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned *o) {
    unsigned a = threadIdx.x, b = 100 + threadIdx.x;
    auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
    auto q = __builtin_amdgcn_permlane16_swap(a, b, false, false);
    o[threadIdx.x] = r[0]; o[64 + threadIdx.x] = r[1]; o[128 + threadIdx.x] = q[0]; o[192 + threadIdx.x] = q[1];
    o[256 + threadIdx.x] = __builtin_amdgcn_update_dpp(0u, a, 0x128, 0xf, 0xf, false);
    o[320 + threadIdx.x] = __builtin_amdgcn_update_dpp(0u, a, 0x121, 0xf, 0xf, false);
}
int main() {
    unsigned *d, h[384];
    hipMalloc(&d, sizeof h);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    const char *names[] = {"p32 r0", "p32 r1", "p16 r0", "p16 r1", "ror8", "ror1"};
    for (int j = 0; j < 6; j++) { printf("%s:", names[j]); for (int i = 0; i < 64; i++) printf(" %u", h[j * 64 + i]); printf("\n"); }
}
