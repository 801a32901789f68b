// Microbenchmark (round 2): raw rates behind the sweep's hot loop on gfx950.
//   part A: ds_read_b128 / ds_read_b64 only (taps at the ring's 80-byte position stride, or 16-byte dense),
//           bytes per clock per CU at 4 / 8 / 16 waves per CU, with a minimal consumer (one v_add per read).
//   part B: VALU only: v_mov_b32_dpp / v_fmac_f32_dpp forms (row_shl:1, wave_shl:1, quad_perm) per wave-instruction.
//   part C: ds_read_b128 + N independent plain FMAs that do NOT consume the loaded data (does LDS traffic alone
//           slow the VALU stream of the same waves?)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));

template <int WIDE, int NT>
__global__ __launch_bounds__(NT) void lds_only(float* out, int iters, int stride_b) {
    extern __shared__ float lds[];
    for (int i = threadIdx.x; i < 32 * 1024; i += NT) lds[i] = (float)(i & 255) * 0.001f;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const char* base = reinterpret_cast<const char*>(lds);
    f4 acc = {0, 0, 0, 0};
    int a = ((lane + wave * 7) & 255) * stride_b;
    for (int it = 0; it < iters; ++it) {
        if (WIDE == 16) {
            f4 x0 = *reinterpret_cast<const f4*>(base + a);
            f4 x1 = *reinterpret_cast<const f4*>(base + a + 16);
            f4 x2 = *reinterpret_cast<const f4*>(base + a + 40 * 80);
            f4 x3 = *reinterpret_cast<const f4*>(base + a + 40 * 80 + 16);
            acc += x0; acc += x1; acc += x2; acc += x3;
        } else {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                f2 x = *reinterpret_cast<const f2*>(base + a + 8 * k + (k >= 4 ? 40 * 80 - 32 : 0));
                acc[k & 3] += x[0] + x[1];
            }
        }
        a = (a + 5 * stride_b) & 0x7fff;
        a = a - (a % 16);
    }
    out[blockIdx.x * NT + threadIdx.x] = acc[0] + acc[1] + acc[2] + acc[3];
}

// MODE 0 plain v_fmac; 1 v_mov_dpp row_shl:1 + fmac; 2 v_mov_dpp wave_shl:1 + fmac; 3 v_fmac_dpp row_shl:1 (asm);
// 4 v_fmac_dpp wave_shl:1 (asm); 5 v_fmac_dpp quad_perm:[1,2,3,3] (asm)
template <int MODE, int NT>
__global__ __launch_bounds__(NT) void dpp_only(float* out, int iters) {
    const int lane = threadIdx.x;
    float a[8], t[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { a[i] = lane * 0.001f + i; t[i] = lane * 0.5f + i; }
    const float w = 1.0001f + lane * 1e-6f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (MODE == 0) a[i] = fmaf(t[i], w, a[i]);
                else if (MODE == 1)
                    a[i] = fmaf(__builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, t[i]), 0x101, 0xf, 0xf, false)), w, a[i]);
                else if (MODE == 2)
                    a[i] = fmaf(__builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, t[i]), 0x130, 0xf, 0xf, false)), w, a[i]);
                else if (MODE == 3)
                    asm volatile("v_fmac_f32_dpp %0, %1, %2 row_shl:1 row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(t[i]), "v"(w));
                else if (MODE == 4)
                    asm volatile("v_fmac_f32_dpp %0, %1, %2 wave_shl:1 row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(t[i]), "v"(w));
                else
                    asm volatile("v_fmac_f32_dpp %0, %1, %2 quad_perm:[1,2,3,3] row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(t[i]), "v"(w));
            }
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += a[i];
    out[blockIdx.x * NT + threadIdx.x] = s;
}

// 4 x ds_read_b128 per iteration whose results are only summed once in a while + NV independent FMAs
template <int NV, int READS, int NT>
__global__ __launch_bounds__(NT) void lds_beside_valu(float* out, int iters, int stride_b) {
    extern __shared__ float lds[];
    for (int i = threadIdx.x; i < 32 * 1024; i += NT) lds[i] = (float)(i & 255) * 0.001f;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const char* base = reinterpret_cast<const char*>(lds);
    float acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = lane * 0.001f + i;
    const float w = 1.0001f, c = 0.001f;
    f4 sink = {0, 0, 0, 0};
    int a = ((lane + wave * 7) & 255) * stride_b;
    for (int it = 0; it < iters; ++it) {
        f4 x[4];
        if (READS) {
            x[0] = *reinterpret_cast<const f4*>(base + a);
            x[1] = *reinterpret_cast<const f4*>(base + a + 16);
            x[2] = *reinterpret_cast<const f4*>(base + a + 40 * 80);
            x[3] = *reinterpret_cast<const f4*>(base + a + 40 * 80 + 16);
        }
#pragma unroll
        for (int r = 0; r < NV / 8; ++r)
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = fmaf(acc[i], w, c);
        if (READS) {
            asm volatile("" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]));
            sink[0] += x[0][0]; sink[1] += x[1][1]; sink[2] += x[2][2]; sink[3] += x[3][3];   // 4 VALU consuming the reads
        }
        a = (a + 5 * stride_b) & 0x7fff;
        a = a - (a % 16);
    }
    float s = sink[0] + sink[1] + sink[2] + sink[3];
#pragma unroll
    for (int i = 0; i < 8; ++i) s += acc[i];
    out[blockIdx.x * NT + threadIdx.x] = s;
}

template <typename F>
static double time_ms(F launch) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(e0);
        launch();
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        (void)hipEventElapsedTime(&ms, e0, e1);
    }
    return ms;
}

template <int WIDE, int NT>
static double lds_bpc(float* out, int iters, int stride_b) {
    auto kern = lds_only<WIDE, NT>;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
    const double ms = time_ms([&] { hipLaunchKernelGGL(kern, dim3(256), dim3(NT), 128 * 1024, 0, out, iters, stride_b); });
    const double bytes_per_cu = (double)iters * (NT / 64) * 64 * 64.0;  // 64 B per lane per iteration
    return bytes_per_cu / (ms * 1e-3 * 2.4e9);
}
template <int MODE, int NT>
static double dpp_cyc(float* out, int iters) {
    const double ms = time_ms([&] { hipLaunchKernelGGL((dpp_only<MODE, NT>), dim3(256), dim3(NT), 0, 0, out, iters); });
    return ms * 1e-3 * 2.4e9 / ((double)iters * 32.0) / (NT / 256);  // cycles per wave-instruction per SIMD
}
template <int NV, int READS, int NT>
static double beside(float* out, int iters, int stride_b) {
    auto kern = lds_beside_valu<NV, READS, NT>;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
    const double ms = time_ms([&] { hipLaunchKernelGGL(kern, dim3(256), dim3(NT), 128 * 1024, 0, out, iters, stride_b); });
    return ms * 1e-3 * 2.4e9 / iters * 4.0 / (NT / 64);  // cycles per iteration per SIMD
}


// part D: the same 4 KB of taps + 24 independent FMAs per iteration, with the reads issued (0) in one burst of four
// ds_read_b128, (1) one ds_read_b128 every 6 FMAs, (2) as eight ds_read_b64, one every 3 FMAs.  Hand-placed (asm):
// does spreading the returns let the VALU stream run beside them?
template <int PAT, int NT>
__global__ __launch_bounds__(NT) void lds_interleave(float* out, int iters, int stride_b) {
    extern __shared__ float lds[];
    for (int i = threadIdx.x; i < 32 * 1024; i += NT) lds[i] = (float)(i & 255) * 0.001f;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = lane * 0.001f + i;
    const float w = 1.0001f, c = 0.001f;
    f4 x0 = {0,0,0,0}, x1 = x0, x2 = x0, x3 = x0;
    float sink = 0.f;
    unsigned a = ((lane + wave * 7) & 255) * stride_b;
#define FMA6(k) _Pragma("unroll") for (int i = 0; i < 6; ++i) acc[(k * 6 + i) & 7] = fmaf(acc[(k * 6 + i) & 7], w, c);
#define FMA3(k) _Pragma("unroll") for (int i = 0; i < 3; ++i) acc[(k * 3 + i) & 7] = fmaf(acc[(k * 3 + i) & 7], w, c);
    for (int it = 0; it < iters; ++it) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        sink += x0[0] + x1[1] + x2[2] + x3[3];
        if (PAT == 0) {
            asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:16\n\tds_read_b128 %2, %4 offset:3200\n\tds_read_b128 %3, %4 offset:3216"
                         : "=v"(x0), "=v"(x1), "=v"(x2), "=v"(x3) : "v"(a) : "memory");
            __builtin_amdgcn_sched_barrier(0);
            FMA6(0) FMA6(1) FMA6(2) FMA6(3)
        } else if (PAT == 1) {
            asm volatile("ds_read_b128 %0, %1" : "=v"(x0) : "v"(a) : "memory"); __builtin_amdgcn_sched_barrier(0); FMA6(0) __builtin_amdgcn_sched_barrier(0);
            asm volatile("ds_read_b128 %0, %1 offset:16" : "=v"(x1) : "v"(a) : "memory"); __builtin_amdgcn_sched_barrier(0); FMA6(1) __builtin_amdgcn_sched_barrier(0);
            asm volatile("ds_read_b128 %0, %1 offset:3200" : "=v"(x2) : "v"(a) : "memory"); __builtin_amdgcn_sched_barrier(0); FMA6(2) __builtin_amdgcn_sched_barrier(0);
            asm volatile("ds_read_b128 %0, %1 offset:3216" : "=v"(x3) : "v"(a) : "memory"); __builtin_amdgcn_sched_barrier(0); FMA6(3)
        } else {
            f2 y[8];
#define RD64(k, off) asm volatile("ds_read_b64 %0, %1 offset:" #off : "=v"(y[k]) : "v"(a) : "memory"); __builtin_amdgcn_sched_barrier(0); FMA3(k) __builtin_amdgcn_sched_barrier(0);
            RD64(0, 0) RD64(1, 8) RD64(2, 16) RD64(3, 24) RD64(4, 3200) RD64(5, 3208) RD64(6, 3216) RD64(7, 3224)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            x0 = (f4){y[0][0], y[1][1], y[2][0], y[3][1]};
            x1 = (f4){y[4][0], y[5][1], y[6][0], y[7][1]};
        }
        __builtin_amdgcn_sched_barrier(0);
        a = (a + 5 * stride_b) & 0x7ff0;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    float s = sink + x0[0] + x1[0] + x2[0] + x3[0];
#pragma unroll
    for (int i = 0; i < 8; ++i) s += acc[i];
    out[blockIdx.x * NT + threadIdx.x] = s;
}
template <int PAT, int NT>
static double interleave(float* out, int iters, int stride_b) {
    auto kern = lds_interleave<PAT, NT>;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
    const double ms = time_ms([&] { hipLaunchKernelGGL(kern, dim3(256), dim3(NT), 128 * 1024, 0, out, iters, stride_b); });
    return ms * 1e-3 * 2.4e9 / iters * 4.0 / (NT / 64);
}

int main() {
    float* out; (void)hipMalloc(&out, 256 * 1024 * sizeof(float));
    const int iters = 40000;
    printf("part A: LDS reads only, bytes per (2.4 GHz) clock per CU\n");
    for (int sb : {80, 16, 64}) {
        printf("  stride %2d B: b128 4/8/16 waves: %6.1f %6.1f %6.1f | b64: %6.1f %6.1f %6.1f\n", sb, lds_bpc<16, 256>(out, iters, sb),
               lds_bpc<16, 512>(out, iters, sb), lds_bpc<16, 1024>(out, iters, sb), lds_bpc<8, 256>(out, iters, sb), lds_bpc<8, 512>(out, iters, sb),
               lds_bpc<8, 1024>(out, iters, sb));
    }
    printf("part B: cycles per wave-instruction per SIMD (1 / 2 / 4 waves per SIMD)\n");
    printf("  v_fmac plain              : %5.2f %5.2f %5.2f\n", dpp_cyc<0, 256>(out, iters), dpp_cyc<0, 512>(out, iters), dpp_cyc<0, 1024>(out, iters));
    printf("  v_mov_dpp row_shl + fmac  : %5.2f %5.2f %5.2f (two instructions unless folded)\n", dpp_cyc<1, 256>(out, iters), dpp_cyc<1, 512>(out, iters), dpp_cyc<1, 1024>(out, iters));
    printf("  v_mov_dpp wave_shl + fmac : %5.2f %5.2f %5.2f\n", dpp_cyc<2, 256>(out, iters), dpp_cyc<2, 512>(out, iters), dpp_cyc<2, 1024>(out, iters));
    printf("  v_fmac_dpp row_shl:1      : %5.2f %5.2f %5.2f\n", dpp_cyc<3, 256>(out, iters), dpp_cyc<3, 512>(out, iters), dpp_cyc<3, 1024>(out, iters));
    printf("  v_fmac_dpp wave_shl:1     : %5.2f %5.2f %5.2f\n", dpp_cyc<4, 256>(out, iters), dpp_cyc<4, 512>(out, iters), dpp_cyc<4, 1024>(out, iters));
    printf("  v_fmac_dpp quad_perm      : %5.2f %5.2f %5.2f\n", dpp_cyc<5, 256>(out, iters), dpp_cyc<5, 512>(out, iters), dpp_cyc<5, 1024>(out, iters));
    printf("part C: cycles per iteration per SIMD: N independent FMAs alone | 4 x ds_read_b128 alone | both (8 and 16 waves per CU)\n");
    printf("  NV=24:  8 waves: %6.1f | %6.1f | %6.1f    16 waves: %6.1f | %6.1f | %6.1f\n", beside<24, 0, 512>(out, iters, 80), beside<0, 1, 512>(out, iters, 80),
           beside<24, 1, 512>(out, iters, 80), beside<24, 0, 1024>(out, iters, 80), beside<0, 1, 1024>(out, iters, 80), beside<24, 1, 1024>(out, iters, 80));
    printf("  NV=48:  8 waves: %6.1f | %6.1f | %6.1f    16 waves: %6.1f | %6.1f | %6.1f\n", beside<48, 0, 512>(out, iters, 80), beside<0, 1, 512>(out, iters, 80),
           beside<48, 1, 512>(out, iters, 80), beside<48, 0, 1024>(out, iters, 80), beside<0, 1, 1024>(out, iters, 80), beside<48, 1, 1024>(out, iters, 80));
    printf("part D: cycles per iteration per SIMD, 4 KB of taps + 24 independent FMAs: burst of 4 x b128 | one b128 per 6 FMAs | 8 x b64, one per 3 FMAs\n");
    printf("   8 waves: %6.1f | %6.1f | %6.1f    12 waves: %6.1f | %6.1f | %6.1f    16 waves: %6.1f | %6.1f | %6.1f\n", interleave<0, 512>(out, iters, 80), interleave<1, 512>(out, iters, 80),
           interleave<2, 512>(out, iters, 80), interleave<0, 768>(out, iters, 80), interleave<1, 768>(out, iters, 80), interleave<2, 768>(out, iters, 80),
           interleave<0, 1024>(out, iters, 80), interleave<1, 1024>(out, iters, 80), interleave<2, 1024>(out, iters, 80));
    return 0;
}
