// Microbenchmark: issue rate of v_fma_f32 vs v_pk_fma_f32 and ds_read_b128 on gfx950.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
__global__ __launch_bounds__(256) void k_fma(float* out, int iters) {
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0+4, a5=a0+5, a6=a0+6, a7=a0+7;
    float b = 1.0001f, c = 0.5f;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            a0 = fmaf(a0, b, c); a1 = fmaf(a1, b, c); a2 = fmaf(a2, b, c); a3 = fmaf(a3, b, c);
            a4 = fmaf(a4, b, c); a5 = fmaf(a5, b, c); a6 = fmaf(a6, b, c); a7 = fmaf(a7, b, c);
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}
__global__ __launch_bounds__(256) void k_pkfma(float* out, int iters) {
    f2 a0 = {(float)threadIdx.x, 1.f}, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, a4=a0+4.f,a5=a0+5.f,a6=a0+6.f,a7=a0+7.f;
    f2 b = {1.0001f, 1.0002f}, c = {0.5f, 0.25f};
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            a0 = __builtin_elementwise_fma(a0, b, c); a1 = __builtin_elementwise_fma(a1, b, c);
            a2 = __builtin_elementwise_fma(a2, b, c); a3 = __builtin_elementwise_fma(a3, b, c);
            a4 = __builtin_elementwise_fma(a4, b, c); a5 = __builtin_elementwise_fma(a5, b, c);
            a6 = __builtin_elementwise_fma(a6, b, c); a7 = __builtin_elementwise_fma(a7, b, c);
        }
    }
    f2 s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s.x + s.y;
}
int main() {
    float* out; hipMalloc(&out, 256 * 2048 * 4 * sizeof(float));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int wpc = 1; wpc <= 4; wpc *= 2) {  // blocks per CU
        int blocks = 256 * wpc, iters = 4096;
        for (int which = 0; which < 2; ++which) {
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(e0);
                if (which == 0) hipLaunchKernelGGL(k_fma, dim3(blocks), dim3(256), 0, 0, out, iters);
                else hipLaunchKernelGGL(k_pkfma, dim3(blocks), dim3(256), 0, 0, out, iters);
                hipEventRecord(e1); hipEventSynchronize(e1);
            }
            float ms; hipEventElapsedTime(&ms, e0, e1);
            double instr = (double)blocks * 4 /*waves*/ * iters * 64.0;  // wave-instructions
            double flop = instr * 64 * 2 * (which ? 2 : 1);
            printf("%s blocks/CU=%d: %.3f ms  %.1f TFLOP/s  %.2f Gwave-instr/s\n", which ? "pk_fma" : "fma   ", wpc, ms,
                   flop / ms / 1e9, instr / ms / 1e6);
        }
    }
    return 0;
}
