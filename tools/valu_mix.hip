// Microbenchmark: issue rate of the non-FMA VALU instructions the sweep's geometry uses, relative to v_fma_f32.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int OP>
__global__ __launch_bounds__(256) void k(float* out, int iters, int seed) {
    float f[8];
    int n[8];
    for (int j = 0; j < 8; ++j) { f[j] = threadIdx.x * 0.001f + j; n[j] = threadIdx.x + j * 7 + seed; }
    const float b = 1.0001f, c = 0.5f;
    const int m = seed | 3;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int r = 0; r < 8; ++r)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                if (OP == 0) f[j] = fmaf(f[j], b, c);                                   // v_fma_f32 / v_fmac
                if (OP == 1) n[j] = n[j] + m;                                          // v_add_u32
                if (OP == 2) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(n[j]) : "v"(m));
                if (OP == 3) asm volatile("v_cvt_i32_f32 %0, %1" : "=v"(n[j]) : "v"(f[j]));
                if (OP == 4) asm volatile("v_min_u32 %0, %0, %1" : "+v"(n[j]) : "v"(m));
                if (OP == 5) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(n[j]) : "v"(m));
                if (OP == 6) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(n[j]) : "v"(m));
                if (OP == 7) asm volatile("v_floor_f32 %0, %0" : "+v"(f[j]));
                if (OP == 8) asm volatile("v_add3_u32 %0, %0, %1, %1" : "+v"(n[j]) : "v"(m));
                if (OP == 9) asm volatile("v_mov_b32 %0, %1" : "=v"(n[j]) : "v"(n[(j + 1) & 7]));
                if (OP == 10) asm volatile("v_rcp_f32 %0, %0" : "+v"(f[j]));
                if (OP == 11) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(f[j]) : "v"(c));
                if (OP == 12) asm volatile("v_cmp_gt_u32 vcc, %0, %1" : : "v"(n[j]), "v"(m) : "vcc");
            }
    }
    float s = 0; int t = 0;
    for (int j = 0; j < 8; ++j) { s += f[j]; t += n[j]; }
    out[blockIdx.x * 256 + threadIdx.x] = s + t;
}
template <int OP>
static void run(const char* name, float* out) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int blocks = 256 * 2, iters = 2048;  // 2 waves per SIMD, as in the sweep kernel
    float ms = 0;
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, out, iters, 5);
        hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
    }
    const double instr = (double)blocks * 4 * iters * 64.0;
    printf("%-16s %7.1f Gwave-instr/s\n", name, instr / ms / 1e6);
}
int main() {
    float* out; hipMalloc(&out, 512 * 256 * sizeof(float));
    run<0>("v_fma_f32", out); run<11>("v_sub_f32", out); run<1>("v_add_u32", out); run<2>("v_cndmask_b32", out);
    run<3>("v_cvt_i32_f32", out); run<7>("v_floor_f32", out); run<4>("v_min_u32", out); run<8>("v_add3_u32", out);
    run<9>("v_mov_b32", out); run<12>("v_cmp_gt_u32", out); run<6>("v_mul_u32_u24", out); run<5>("v_mul_lo_u32", out);
    run<10>("v_rcp_f32", out);
    return 0;
}
