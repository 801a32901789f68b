// Microbenchmark (round 2): issue cost of the instructions an fp16-tap blend can be built from (cycles per
// wave-instruction per SIMD at 1 / 2 / 4 waves per SIMD): v_fma_f32, v_fma_mix_f32 (fp16 operand widened in the
// instruction), v_cvt_f32_f16 (+ SDWA high half), v_pk_fma_f32, v_pk_fma_f16.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE, int NT>
__global__ __launch_bounds__(NT) void k(float* out, int iters) {
    const int lane = threadIdx.x;
    float a[8], t[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { a[i] = lane * 0.001f + i; t[i] = __builtin_bit_cast(float, 0x3c003c00u + (unsigned)(lane + i)); }
    const float w = 1.0001f + lane * 1e-6f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (MODE == 0) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(t[i]), "v"(w));
                else if (MODE == 1) asm volatile("v_fma_mix_f32 %0, %1, %2, %0 op_sel_hi:[1,0,0]" : "+v"(a[i]) : "v"(t[i]), "v"(w));
                else if (MODE == 2) asm volatile("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(a[i]) : "v"(t[i]), "v"(w));
                else if (MODE == 3) asm volatile("v_cvt_f32_f16_e32 %0, %1" : "=v"(a[i]) : "v"(t[i]));
                else if (MODE == 4) asm volatile("v_cvt_f32_f16_sdwa %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1" : "=v"(a[i]) : "v"(t[i]));
                else if (MODE == 5) asm volatile("v_pk_fma_f16 %0, %1, %2, %0" : "+v"(a[i]) : "v"(t[i]), "v"(w));
                else if (MODE == 6) asm volatile("v_mad_mix_f32 %0, %1, %2, %0 op_sel_hi:[1,0,0]" : "+v"(a[i]) : "v"(t[i]), "v"(w));
            }
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += a[i];
    out[blockIdx.x * NT + threadIdx.x] = s;
}
template <int MODE, int NT>
static double cyc(float* out, int iters) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL((k<MODE, NT>), dim3(256), dim3(NT), 0, 0, out, iters);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        (void)hipEventElapsedTime(&ms, e0, e1);
    }
    return ms * 1e-3 * 2.4e9 / ((double)iters * 32.0) / (NT / 256);
}
#define ROW(name, M) printf("  %-28s: %5.2f %5.2f %5.2f\n", name, cyc<M, 256>(out, iters), cyc<M, 512>(out, iters), cyc<M, 1024>(out, iters));
int main() {
    float* out; (void)hipMalloc(&out, 256 * 1024 * sizeof(float));
    const int iters = 40000;
    printf("cycles (2.4 GHz) per wave-instruction per SIMD at 1 / 2 / 4 waves per SIMD\n");
    ROW("v_fma_f32", 0) ROW("v_fma_mix_f32 (lo half)", 1) ROW("v_fma_mix_f32 (hi half)", 2) ROW("v_cvt_f32_f16", 3) ROW("v_cvt_f32_f16 sdwa WORD_1", 4) ROW("v_pk_fma_f16", 5)
    return 0;
}
