// Microbenchmark (round 2): what sets the speed of the sweep's hot loop -- VALU issue per wave, or LDS?
//
// A gfx950 SIMD issues a wave64 fp32 VALU instruction over 2 cycles, but ONE wave can only issue one VALU
// instruction every 4 cycles; v_pk_fma_f32 (two FMAs per lane) occupies the pipe for 4 cycles, so a lone wave
// reaches the full FMA rate only with packed instructions.  The sweep kernel runs 2 compute waves per SIMD which
// alternate between waiting for their LDS taps and blending them, i.e. each is mostly alone on its SIMD.
//
//   part 1: VALU only -- cycles per (64-lane) FMA for plain and packed streams at 1..4 waves per SIMD
//   part 2: the unit loop of the sweep (4 x ds_read_b128 one or two units ahead, blend 4 taps x 4 channels,
//           accumulate sum and sum of squares, + EXTRA plain VALU standing for the geometry share) with plain
//           and packed arithmetic at 8 / 12 / 16 waves per CU.
// Prints cycles per unit per wave at the measured wall time x 2.4 GHz (so a lower clock shows as more cycles).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ f2 pk_fma(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); }

template <int PK, int NT>
__global__ __launch_bounds__(NT) void valu_only(float* out, int iters) {
    const int lane = threadIdx.x;
    f2 a[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = (f2){lane * 0.001f + i, lane * 0.002f - i};
    const f2 w = {1.0001f, 0.9999f};
    const f2 c = {0.001f, -0.001f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (PK) a[i] = pk_fma(a[i], w, c);
                else { a[i][0] = fmaf(a[i][0], w[0], c[0]); a[i][1] = fmaf(a[i][1], w[1], c[1]); }
            }
    }
    f2 s = a[0];
#pragma unroll
    for (int i = 1; i < 8; ++i) s += a[i];
    out[blockIdx.x * NT + threadIdx.x] = s[0] + s[1];
}

// EXTRA: plain VALU instructions per unit besides the 24 (12 packed) of blend + accumulate
template <int PK, int DEPTH, int EXTRA, int NT>
__global__ __launch_bounds__(NT) void unit_loop(float* out, int iters, int stride_f, int scale_q8) {
    extern __shared__ float lds[];
    for (int i = threadIdx.x; i < 32 * 1024; i += NT) lds[i] = (float)(i & 255) * 0.001f;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int RW = 40;
    const f4* lds4 = reinterpret_cast<const f4*>(lds);
    const int s4 = stride_f >> 2;
    const int lpos = (lane * scale_q8) >> 8;
    auto addr = [&](int it) { return ((lpos + it * 5 + wave * 97) & 255) * s4 + (it & 3); };
    f4 t[DEPTH + 1][4];
    auto request = [&](int it, f4 (&d)[4]) {
        const int a = addr(it);
        d[0] = lds4[a];
        d[1] = lds4[a + s4];
        d[2] = lds4[a + RW * s4];
        d[3] = lds4[a + RW * s4 + s4];
    };
#pragma unroll
    for (int j = 0; j < DEPTH; ++j) request(j, t[j]);
    float nw = 0.25f + lane * 1e-4f, ne = 0.26f, sw = 0.24f, se = 0.25f;
    f4 s = {0, 0, 0, 0}, qq = {0, 0, 0, 0};
    float g = lane * 0.5f;
    for (int it = 0; it < iters; it += DEPTH + 1) {
#pragma unroll
        for (int j = 0; j <= DEPTH; ++j) {
            request(it + j + DEPTH, t[(j + DEPTH) % (DEPTH + 1)]);
            __builtin_amdgcn_sched_barrier(0);
            f4 (&c)[4] = t[j];
            asm volatile("" : "+v"(c[3]));
#pragma unroll
            for (int e = 0; e < EXTRA; ++e) g = fmaf(g, 1.0001f, 0.5f);   // dependent chain like the geometry
            f4 v;
            if (PK) {
#pragma unroll
                for (int hh = 0; hh < 2; ++hh) {
                    f2 x = (f2){c[0][2 * hh], c[0][2 * hh + 1]} * (f2){nw, nw};
                    x = pk_fma((f2){c[1][2 * hh], c[1][2 * hh + 1]}, (f2){ne, ne}, x);
                    x = pk_fma((f2){c[2][2 * hh], c[2][2 * hh + 1]}, (f2){sw, sw}, x);
                    x = pk_fma((f2){c[3][2 * hh], c[3][2 * hh + 1]}, (f2){se, se}, x);
                    f2 ss = (f2){s[2 * hh], s[2 * hh + 1]} + x;
                    f2 q2 = pk_fma(x, x, (f2){qq[2 * hh], qq[2 * hh + 1]});
                    s[2 * hh] = ss[0]; s[2 * hh + 1] = ss[1];
                    qq[2 * hh] = q2[0]; qq[2 * hh + 1] = q2[1];
                }
            } else {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    v[k] = fmaf(c[3][k], se, fmaf(c[2][k], sw, fmaf(c[1][k], ne, c[0][k] * nw)));
                    s[k] += v[k];
                    qq[k] = fmaf(v[k], v[k], qq[k]);
                }
            }
            nw += g * 1e-9f;
        }
    }
    f4 r = s + qq;
    out[blockIdx.x * NT + threadIdx.x] = r[0] + r[1] + r[2] + r[3] + g;
}


// MODE 2: east taps from the neighbouring lane's west taps through DPP (wave_shl:1 -> lane l reads lane l+1),
// west blend + accumulation packed.  Lanes whose neighbour does not hold their east tap (FIXPCT per cent of the
// lanes, pseudo-randomly, plus lanes 31 and 63 when ROWEND) fetch their own east taps in an EXEC-masked pass.
__device__ __forceinline__ float dpp_next(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x130, 0xf, 0xf, false));
}
template <int DEPTH, int EXTRA, int FIXPCT, int ROWEND, int NT>
__global__ __launch_bounds__(NT) void unit_loop_dpp(float* out, int iters, int stride_f, int scale_q8) {
    extern __shared__ float lds[];
    for (int i = threadIdx.x; i < 32 * 1024; i += NT) lds[i] = (float)(i & 255) * 0.001f;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int RW = 40;
    const f4* lds4 = reinterpret_cast<const f4*>(lds);
    const int s4 = stride_f >> 2;
    const int lpos = (lane * scale_q8) >> 8;
    auto addr = [&](int it) { return ((lpos + it * 5 + wave * 97) & 255) * s4 + (it & 3); };
    auto fixlane = [&](int it) -> bool {
        bool f = ROWEND && ((lane & 31) == 31);
        if (FIXPCT > 0) f = f || ((((unsigned)(lane * 2654435761u + (it >> 4) * 40503u) >> 8) % 100u) < (unsigned)FIXPCT);
        return f;
    };
    f4 t[DEPTH + 1][4];
    auto request = [&](int it, f4 (&d)[4]) {
        const int a = addr(it);
        d[0] = lds4[a];
        d[2] = lds4[a + RW * s4];
        if (FIXPCT > 0 || ROWEND) {
            if (fixlane(it)) {
                d[1] = lds4[a + s4];
                d[3] = lds4[a + RW * s4 + s4];
            }
        }
    };
#pragma unroll
    for (int j = 0; j < DEPTH; ++j) request(j, t[j]);
    float nw = 0.25f + lane * 1e-4f, ne = 0.26f + lane * 1e-4f, sw = 0.24f, se = 0.25f;
    f4 s = {0, 0, 0, 0}, qq = {0, 0, 0, 0};
    float g = lane * 0.5f;
    for (int it = 0; it < iters; it += DEPTH + 1) {
#pragma unroll
        for (int j = 0; j <= DEPTH; ++j) {
            request(it + j + DEPTH, t[(j + DEPTH) % (DEPTH + 1)]);
            __builtin_amdgcn_sched_barrier(0);
            f4 (&c)[4] = t[j];
            asm volatile("" : "+v"(c[2]));
#pragma unroll
            for (int e = 0; e < EXTRA; ++e) g = fmaf(g, 1.0001f, 0.5f);
            const bool fx = (FIXPCT > 0 || ROWEND) ? fixlane(it + j) : false;
            const float ne_m = fx ? 0.0f : ne, se_m = fx ? 0.0f : se;
            f4 v;
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) {
                f2 x = (f2){c[0][2 * hh], c[0][2 * hh + 1]} * (f2){nw, nw};
                x = pk_fma((f2){c[2][2 * hh], c[2][2 * hh + 1]}, (f2){sw, sw}, x);
                v[2 * hh] = x[0]; v[2 * hh + 1] = x[1];
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                v[k] = fmaf(dpp_next(c[0][k]), ne_m, v[k]);
                v[k] = fmaf(dpp_next(c[2][k]), se_m, v[k]);
            }
            if (FIXPCT > 0 || ROWEND) {
                if (fx) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) v[k] = fmaf(c[3][k], se, fmaf(c[1][k], ne, v[k]));
                }
            }
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) {
                f2 x = {v[2 * hh], v[2 * hh + 1]};
                f2 ss = (f2){s[2 * hh], s[2 * hh + 1]} + x;
                f2 q2 = pk_fma(x, x, (f2){qq[2 * hh], qq[2 * hh + 1]});
                s[2 * hh] = ss[0]; s[2 * hh + 1] = ss[1];
                qq[2 * hh] = q2[0]; qq[2 * hh + 1] = q2[1];
            }
            nw += g * 1e-9f;
        }
    }
    f4 r = s + qq;
    out[blockIdx.x * NT + threadIdx.x] = r[0] + r[1] + r[2] + r[3] + g;
}

static double time_ms(void (*launch)(void*), void* ctx) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        launch(ctx);
        hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
    }
    hipEventDestroy(e0); hipEventDestroy(e1);
    return ms;
}

struct Ctx { float* out; int iters; int stride_f; int scale; };

template <int PK, int NT>
static double run_valu(Ctx c) {
    auto l = [](void* p) { Ctx* c = (Ctx*)p; hipLaunchKernelGGL((valu_only<PK, NT>), dim3(256), dim3(NT), 0, 0, c->out, c->iters); };
    const double ms = time_ms(l, &c);
    // 64 FMAs per lane per iteration (32 instruction slots x 2 lanes of a pair)
    return ms * 1e-3 * 2.4e9 / ((double)c.iters * 64.0);  // cycles per 64-lane FMA per wave
}

template <int PK, int DEPTH, int EXTRA, int NT>
static double run_unit(Ctx c) {
    auto kern = unit_loop<PK, DEPTH, EXTRA, NT>;
    hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
    auto l = [](void* p) {
        Ctx* c = (Ctx*)p;
        hipLaunchKernelGGL((unit_loop<PK, DEPTH, EXTRA, NT>), dim3(256), dim3(NT), 128 * 1024, 0, c->out, c->iters, c->stride_f, c->scale);
    };
    const double ms = time_ms(l, &c);
    return ms * 1e-3 * 2.4e9 / c.iters;
}

template <int NT>
static void unit_rows(Ctx c) {
    const int wpc = NT / 64;
    printf("  %2d waves/CU | plain: d1 e0 %6.1f  d1 e13 %6.1f  d2 e13 %6.1f | packed: d1 e0 %6.1f  d1 e13 %6.1f  d2 e13 %6.1f  d1 e6 %6.1f | per-SIMD cycles per unit (plain d1 e13 / packed d1 e13 / packed d1 e6): %5.1f / %5.1f / %5.1f\n",
           wpc, run_unit<0, 1, 0, NT>(c), run_unit<0, 1, 13, NT>(c), run_unit<0, 2, 13, NT>(c), run_unit<1, 1, 0, NT>(c),
           run_unit<1, 1, 13, NT>(c), run_unit<1, 2, 13, NT>(c), run_unit<1, 1, 6, NT>(c),
           run_unit<0, 1, 13, NT>(c) * 4.0 / wpc, run_unit<1, 1, 13, NT>(c) * 4.0 / wpc, run_unit<1, 1, 6, NT>(c) * 4.0 / wpc);
}


template <int DEPTH, int EXTRA, int FIXPCT, int ROWEND, int NT>
static double run_dpp(Ctx c) {
    auto kern = unit_loop_dpp<DEPTH, EXTRA, FIXPCT, ROWEND, NT>;
    hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
    auto l = [](void* p) {
        Ctx* c = (Ctx*)p;
        hipLaunchKernelGGL((unit_loop_dpp<DEPTH, EXTRA, FIXPCT, ROWEND, NT>), dim3(256), dim3(NT), 128 * 1024, 0, c->out, c->iters, c->stride_f, c->scale);
    };
    const double ms = time_ms(l, &c);
    return ms * 1e-3 * 2.4e9 / c.iters;
}
template <int NT>
static void dpp_rows(Ctx c) {
    const int wpc = NT / 64;
    printf("  %2d waves/CU | dpp d1: e0 nofix %6.1f  e6 nofix %6.1f  e6 rowend %6.1f  e6 rowend+2%% %6.1f  e6 rowend+10%% %6.1f  e6 rowend+50%% %6.1f | per-SIMD e6: nofix %5.1f rowend+2%% %5.1f\n",
           wpc, run_dpp<1, 0, 0, 0, NT>(c), run_dpp<1, 6, 0, 0, NT>(c), run_dpp<1, 6, 0, 1, NT>(c), run_dpp<1, 6, 2, 1, NT>(c),
           run_dpp<1, 6, 10, 1, NT>(c), run_dpp<1, 6, 50, 1, NT>(c), run_dpp<1, 6, 0, 0, NT>(c) * 4.0 / wpc, run_dpp<1, 6, 2, 1, NT>(c) * 4.0 / wpc);
}

int main() {
    float* out; hipMalloc(&out, 256 * 1024 * sizeof(float));
    Ctx c{out, 40000, 20, 256};
    printf("part 1: VALU only, cycles (at 2.4 GHz) per 64-lane FMA per wave; per SIMD = that / waves per SIMD\n");
    printf("  plain  v_fma_f32   : 1 w/SIMD %5.2f  2 %5.2f  3 %5.2f  4 %5.2f\n", run_valu<0, 256>(c), run_valu<0, 512>(c), run_valu<0, 768>(c), run_valu<0, 1024>(c));
    printf("  packed v_pk_fma_f32: 1 w/SIMD %5.2f  2 %5.2f  3 %5.2f  4 %5.2f\n", run_valu<1, 256>(c), run_valu<1, 512>(c), run_valu<1, 768>(c), run_valu<1, 1024>(c));
    printf("part 2: unit loop, cycles per unit per wave (d = units requested ahead, e = extra plain VALU per unit)\n");
    unit_rows<256>(c);
    unit_rows<512>(c);
    unit_rows<768>(c);
    unit_rows<1024>(c);
    printf("part 3: east taps through DPP from the next lane (2 reads per unit + EXEC-masked fix-up reads)\n");
    dpp_rows<512>(c);
    dpp_rows<768>(c);
    dpp_rows<1024>(c);
    c.scale = 269;
    printf("source scale 1.05 (column skips -> bank conflicts):\n");
    unit_rows<512>(c);
    unit_rows<1024>(c);
    return 0;
}
