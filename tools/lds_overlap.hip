// Microbenchmark: do ds_read_b128 traffic and VALU work of the SAME waves overlap on gfx950?
// Models the hot loop of sweep_tiled_kernel: 8 waves per CU (2 per SIMD), per "unit" a wave requests 4 x
// ds_read_b128 (the four bilinear taps of 4 channels, 80-byte position stride) one unit ahead and then runs NV
// dependent-free FMAs on the unit that has landed.  Prints cycles per unit for LDS only, VALU only and both:
// max(...) means the two pipes overlap, the sum means they serialise.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
__device__ int g_scale_q8 = 256;  // source positions per pixel, 8.8 fixed point (skips when > 256, repeats when < 256)

template <int NV, bool USE_LDS, int DEPTH, int MM = 0, int NT = 512>
__global__ __launch_bounds__(NT) void k(float* out, int iters, int stride_f) {
    extern __shared__ float lds[];
    for (int i = threadIdx.x; i < 32 * 1024; i += NT) lds[i] = (float)(i & 255) * 0.001f;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int RW = 40;
    f4 acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
    f4 t[DEPTH + 1][4];
    const f4* lds4 = reinterpret_cast<const f4*>(lds);  // indices in 16-byte units: ds_read_b128 needs provable alignment
    const int s4 = stride_f >> 2;
    const int lpos = (lane * g_scale_q8) >> 8;
    auto addr = [&](int it) { return ((lpos + it * 5 + wave * 97) & 255) * s4 + (it & 3); };
    auto request = [&](int it, f4 (&d)[4]) {
        if (USE_LDS) {
            const int a = addr(it);
            d[0] = lds4[a];
            d[2] = lds4[a + RW * s4];
            if (MM == 0) {
                d[1] = lds4[a + s4];
                d[3] = lds4[a + RW * s4 + s4];
            } else if (MM >= 5) {
                // all reads through inline asm (the consumer waits with s_waitcnt lgkmcnt(n) by hand):
                // 5: four full reads; 6: west full + east by 1 lane in 16 (EXEC narrowed, no branch); 7: west only
                const unsigned b0 = (unsigned)a * 16u, b2 = (unsigned)(a + RW * s4) * 16u;
                const unsigned b1 = (unsigned)(a + s4) * 16u, b3 = (unsigned)(a + RW * s4 + s4) * 16u;
                asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %3" : "=v"(d[0]), "=v"(d[2]) : "v"(b0), "v"(b2) : "memory");
                if (MM == 5) {
                    asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %3" : "=v"(d[1]), "=v"(d[3]) : "v"(b1), "v"(b3) : "memory");
                } else if (MM == 6) {
                    f4 e1 = {0, 0, 0, 0}, e3 = {0, 0, 0, 0};
                    asm volatile("s_mov_b64 s[20:21], exec\n\ts_mov_b32 exec_lo, 0x00010001\n\ts_mov_b32 exec_hi, 0x00010001\n\t"
                                 "ds_read_b128 %0, %2\n\tds_read_b128 %1, %3\n\ts_mov_b64 exec, s[20:21]"
                                 : "+v"(e1), "+v"(e3) : "v"(b1), "v"(b3) : "s20", "s21", "memory");
                    d[1] = e1;
                    d[3] = e3;
                } else {
                    d[1] = (f4){1, 1, 1, 1};
                    d[3] = (f4){2, 2, 2, 2};
                }
            } else if (MM == 4) {
                // east taps fetched only by 1 lane in 16, EXEC narrowed without a branch: are masked-off lanes free?
                const unsigned ab = (unsigned)(a + s4) * 16u, ab2 = (unsigned)(a + RW * s4 + s4) * 16u;
                f4 e1 = d[0], e3 = d[2];
                asm volatile("s_mov_b64 s[20:21], exec\n\ts_mov_b32 exec_lo, 0x00010001\n\ts_mov_b32 exec_hi, 0x00010001\n\t"
                             "ds_read_b128 %0, %2\n\tds_read_b128 %1, %3\n\ts_mov_b64 exec, s[20:21]"
                             : "+v"(e1), "+v"(e3) : "v"(ab), "v"(ab2) : "s20", "s21", "memory");
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                d[1] = e1;
                d[3] = e3;
            } else {
                d[1] = d[0];
                d[3] = d[2];
                if (MM >= 2 && (lane & (MM == 2 ? 15 : 3)) == 0) {  // "broken" lanes fetch their own east taps
                    d[1] = lds4[a + s4];
                    d[3] = lds4[a + RW * s4 + s4];
                }
            }
        } else {
            const float v = (float)(it + lane);
            d[0] = (f4){v, v, v, v}; d[1] = d[0] + 1.0f; d[2] = d[0] + 2.0f; d[3] = d[0] + 3.0f;
        }
    };
#pragma unroll
    for (int j = 0; j < DEPTH; ++j) request(j, t[j]);
    const float w0 = 0.25f + lane * 1e-4f, w1 = 0.26f, w2 = 0.24f, w3 = 0.25f;
    for (int it = 0; it < iters; it += DEPTH + 1) {
#pragma unroll
        for (int j = 0; j <= DEPTH; ++j) {
            request(it + j + DEPTH, t[(j + DEPTH) % (DEPTH + 1)]);
            __builtin_amdgcn_sched_barrier(0);  // keep the prefetch ahead of the blend (the scheduler sinks it otherwise)
            f4 (&c)[4] = t[j];
            if (USE_LDS && MM >= 5) {  // reads of DEPTH later units may stay in flight (LDS returns in order)
                if (MM == 7) asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(2 * DEPTH) : "memory");
                else asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(4 * DEPTH) : "memory");
            }
            asm volatile("" : "+v"(c[0]), "+v"(c[1]), "+v"(c[2]), "+v"(c[3]));      // LDS returns in order: one wait per unit
            // NV wave-instructions: blend (4 per channel) + accumulate, repeated
            f4 v = c[0] * w0;
#pragma unroll
            for (int r = 0; r < NV / 4; ++r) {
                const int s = r & 3;
                if (s == 0) v = c[1] * w1 + v;
                else if (s == 1) v = c[2] * w2 + v;
                else if (s == 2) { v = c[3] * w3 + v; acc0 += v; }
                else { acc1 = v * v + acc1; v = c[0] * w0; }
            }
            acc0 += v;
        }
    }
    f4 s = acc0 + acc1;
    out[blockIdx.x * NT + threadIdx.x] = s[0] + s[1] + s[2] + s[3];
}

template <int NV, bool USE_LDS, int DEPTH, int MM = 0, int NT = 512>
static double run(float* out, int iters, int stride_f) {
    auto kern = k<NV, USE_LDS, DEPTH, MM, NT>;
    hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(kern, dim3(256), dim3(NT), 128 * 1024, 0, out, iters, stride_f);
        hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
    }
    return ms * 1e-3 * 2.4e9 / iters;  // cycles (at 2.4 GHz) per unit per wave, all 8 waves of a CU concurrent
}

template <int NV>
static void row(float* out, int iters, int stride_f) {
    const double both1 = run<NV, true, 1>(out, iters, stride_f), both2 = run<NV, true, 2>(out, iters, stride_f);
    const double valu = run<NV, false, 1>(out, iters, stride_f);
    const double h2 = run<NV, true, 1, 1>(out, iters, stride_f), m16 = run<NV, true, 1, 2>(out, iters, stride_f),
                 m4 = run<NV, true, 1, 3>(out, iters, stride_f);
    printf("NV=%2d  VALU only %6.1f | 4 reads depth1 %6.1f depth2 %6.1f | 2 reads %6.1f | 2 + 2 by 1/16 lanes %6.1f | 2 + 2 by 1/4 lanes %6.1f cycles/unit\n",
           NV, valu, both1, both2, h2, m16, m4);
}

int main() {
    float* out; hipMalloc(&out, 256 * 1024 * sizeof(float));
    const int iters = 60000;
    for (int stride_f : {20, 16}) {
        printf("position stride %d floats (%s)\n", stride_f, stride_f == 20 ? "padded: conflict-free b128" : "unpadded");
        row<0>(out, iters, stride_f);
        row<16>(out, iters, stride_f);
        row<32>(out, iters, stride_f);
        row<48>(out, iters, stride_f);
        row<64>(out, iters, stride_f);
        row<96>(out, iters, stride_f);
    }
    // occupancy: the same loop with 8 / 12 / 16 waves per CU; SIMD throughput = waves per SIMD / cycles per unit
    printf("waves per CU (padded stride, NV=48, 4 reads): cycles per unit and units per 1000 SIMD-cycles\n");
    const double c8 = run<48, true, 1, 0, 512>(out, iters, 20), c12 = run<48, true, 1, 0, 768>(out, iters, 20),
                 c16 = run<48, true, 1, 0, 1024>(out, iters, 20);
    printf("  8 waves %6.1f (%.2f)  12 waves %6.1f (%.2f)  16 waves %6.1f (%.2f)\n", c8, 2000.0 / c8, c12, 3000.0 / c12, c16,
           4000.0 / c16);
    printf("2 full + 2 EXEC-masked reads (1 lane in 16, no branch; waits for them at once), NV=48: %6.1f vs 2 reads %6.1f vs 4 reads %6.1f\n",
           run<48, true, 1, 4, 512>(out, iters, 20), run<48, true, 1, 1, 512>(out, iters, 20), run<48, true, 1, 0, 512>(out, iters, 20));
    printf("hand-scheduled reads, NV=24 / 48: four full %6.1f %6.1f | west full + east by 1 lane in 16 (EXEC) %6.1f %6.1f | west only %6.1f %6.1f\n",
           run<24, true, 1, 5, 512>(out, iters, 20), run<48, true, 1, 5, 512>(out, iters, 20), run<24, true, 1, 6, 512>(out, iters, 20),
           run<48, true, 1, 6, 512>(out, iters, 20), run<24, true, 1, 7, 512>(out, iters, 20), run<48, true, 1, 7, 512>(out, iters, 20));
    printf("source scale (positions per pixel), padded stride, NV=48, 4 reads, 8 waves: cycles per unit\n");
    for (int sc : {256, 243, 230, 269, 282, 320}) {
        hipMemcpyToSymbol(HIP_SYMBOL(g_scale_q8), &sc, sizeof(int));
        printf("  scale %.3f: %6.1f\n", sc / 256.0, run<48, true, 1, 0, 512>(out, iters, 20));
    }
    return 0;
}
