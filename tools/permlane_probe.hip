// Probe of v_permlane16_swap_b32 on gfx950 (GPU box): prints, per lane, which lane's value each of the two results holds when
// both operands are the lane id (+ 100 for the second).  hipcc --offload-arch=gfx950 -O2 tools/permlane_probe.hip -o tools/permlane_probe
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned* o) {
    unsigned a = threadIdx.x, b = threadIdx.x + 100;
    auto r = __builtin_amdgcn_permlane16_swap(a, b, false, false);
    o[threadIdx.x] = r[0];
    o[64 + threadIdx.x] = r[1];
}
int main() {
    unsigned* d; unsigned h[128];
    if (hipMalloc(&d, sizeof(h)) != hipSuccess) return 1;
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    if (hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost) != hipSuccess) return 2;
    for (int r = 0; r < 2; ++r) {
        printf("result %d:", r);
        for (int i = 0; i < 64; i += 8) printf(" [%d]=%u", i, h[r * 64 + i]);
        printf("\n");
    }
    return 0;
}
