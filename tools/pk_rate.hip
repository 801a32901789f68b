// Microbenchmark (round 3): what a SIMD of gfx950 sustains on the sweep kernel's arithmetic -- v_pk_fma_f32 / v_pk_mul_f32 /
// v_pk_add_f32 chains as the blend + variance accumulation issues them -- as a function of
//   * the number of INDEPENDENT dependency chains a wave interleaves (1, 2, 3, 4, 6, 8),
//   * the waves per SIMD (1, 2, 3, 4),
//   * the operand form (plain 64-bit operands, op_sel_hi broadcast of one weight, plain v_fma_f32),
//   * which VGPR banks (register index mod 4) the three source operands come from.
// Cycles are shader cycles from s_memtime inside the kernel (median over the waves of the launch), so the chip's clock
// under load does not enter; s_memrealtime (100 MHz) gives that clock beside it.
//   hipcc --offload-arch=gfx950 -O2 tools/pk_rate.hip -o tools/pk_rate && tools/pk_rate
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>

#define REP4(x) x x x x
#define REP16(x) REP4(REP4(x))

// One instruction of chain c (c = 0..7): acc pair v[16+2c : 17+2c], x pair v[32+2c : 33+2c] (or as given), weight v[0:1]
#define PKFMA(c) "v_pk_fma_f32 v[%c0+" #c "*2:%c0+" #c "*2+1], v[%c1+" #c "*2:%c1+" #c "*2+1], v[%c2:%c2+1], v[%c0+" #c "*2:%c0+" #c "*2+1]\n\t"

enum { OP_PKFMA = 0, OP_PKFMA_BC = 1, OP_FMA = 2, OP_PKADD = 3, OP_PKMUL_FMA = 4 };

// The body is generated with fixed physical registers so that the bank of every operand is known:
//   weights v[4:5], accumulators v[ACC0 + 2c ..], multiplicands v[X0 + 2c ..]
template <int OP, int CHAINS, int ACC0, int X0>
__global__ __launch_bounds__(256) void chain_kernel(unsigned long long* stamps, float* sink, int iters) {
    // initialise every register the asm touches
    float init = threadIdx.x * 1e-6f;
    unsigned long long t0, t1, r0, r1;
    asm volatile(
        "v_mov_b32 v4, 1.0\n\tv_mov_b32 v5, 1.0\n\t"
        "v_mov_b32 v6, %0\n\tv_mov_b32 v7, %0\n\t" ::"v"(init)
        : "v4", "v5", "v6", "v7");
#define INIT(r) asm volatile("v_mov_b32 v" #r ", %0" ::"v"(init) : "v" #r);
    INIT(16) INIT(17) INIT(18) INIT(19) INIT(20) INIT(21) INIT(22) INIT(23) INIT(24) INIT(25) INIT(26) INIT(27) INIT(28) INIT(29) INIT(30) INIT(31)
    INIT(32) INIT(33) INIT(34) INIT(35) INIT(36) INIT(37) INIT(38) INIT(39) INIT(40) INIT(41) INIT(42) INIT(43) INIT(44) INIT(45) INIT(46) INIT(47)
    INIT(48) INIT(49) INIT(50) INIT(51)
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0), "=s"(r0)::"memory");
    for (int it = 0; it < iters; ++it) {
#define ONE(c)                                                                                                                     \
    if (c < CHAINS) {                                                                                                              \
        if (OP == OP_PKFMA)                                                                                                        \
            asm volatile("v_pk_fma_f32 v[%c0:%c0+1], v[%c1:%c1+1], v[4:5], v[%c0:%c0+1]" ::"i"(ACC0 + 2 * c), "i"(X0 + 2 * c));     \
        else if (OP == OP_PKFMA_BC)                                                                                                \
            asm volatile("v_pk_fma_f32 v[%c0:%c0+1], v[%c1:%c1+1], v[4:5], v[%c0:%c0+1] op_sel_hi:[1,0,1]" ::"i"(ACC0 + 2 * c),     \
                         "i"(X0 + 2 * c));                                                                                         \
        else if (OP == OP_FMA)                                                                                                     \
            asm volatile("v_fma_f32 v%c0, v%c1, v4, v%c0\n\tv_fma_f32 v%c2, v%c3, v4, v%c2" ::"i"(ACC0 + 2 * c), "i"(X0 + 2 * c),  \
                         "i"(ACC0 + 2 * c + 1), "i"(X0 + 2 * c + 1));                                                              \
        else if (OP == OP_PKADD)                                                                                                   \
            asm volatile("v_pk_add_f32 v[%c0:%c0+1], v[%c0:%c0+1], v[%c1:%c1+1]" ::"i"(ACC0 + 2 * c), "i"(X0 + 2 * c));             \
        else                                                                                                                       \
            asm volatile("v_pk_mul_f32 v[%c0:%c0+1], v[%c1:%c1+1], v[4:5] op_sel_hi:[1,0]\n\t"                                     \
                         "v_pk_fma_f32 v[%c0:%c0+1], v[%c1:%c1+1], v[6:7], v[%c0:%c0+1] op_sel_hi:[1,0,1]" ::"i"(ACC0 + 2 * c),    \
                         "i"(X0 + 2 * c));                                                                                         \
    }
#define ROUND ONE(0) ONE(1) ONE(2) ONE(3) ONE(4) ONE(5) ONE(6) ONE(7)
        ROUND ROUND ROUND ROUND ROUND ROUND ROUND ROUND
    }
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1), "=s"(r1)::"memory");
    float out;
    asm volatile("v_add_f32 %0, v16, v17" : "=v"(out));
    if (threadIdx.x % 64 == 0) {
        const int w = (blockIdx.x * blockDim.x + threadIdx.x) / 64;
        stamps[2 * w] = t1 - t0;
        stamps[2 * w + 1] = r1 - r0;
    }
    if (out == 123.456f) sink[0] = out;
}

template <int OP, int CHAINS, int ACC0, int X0>
static void run(const char* name, int waves_per_simd, unsigned long long* d_st, float* d_sink) {
    const int iters = 2000;
    const int blocks = 256 * waves_per_simd;   // 256-thread blocks: one wave per SIMD each
    std::vector<unsigned long long> st(2 * blocks * 4);
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL((chain_kernel<OP, CHAINS, ACC0, X0>), dim3(blocks), dim3(256), 0, 0, d_st, d_sink, iters);
        hipDeviceSynchronize();
    }
    hipMemcpy(st.data(), d_st, st.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    std::vector<double> cyc, clk;
    for (int w = 0; w < blocks * 4; ++w) {
        cyc.push_back((double)st[2 * w]);
        clk.push_back((double)st[2 * w] / std::max(1.0, (double)st[2 * w + 1]) * 100.0);
    }
    std::sort(cyc.begin(), cyc.end());
    std::sort(clk.begin(), clk.end());
    const int per_instr = (OP == OP_FMA || OP == OP_PKMUL_FMA) ? 2 : 1;
    const double n = (double)iters * 8 * CHAINS * per_instr;
    const double c = cyc[cyc.size() / 2] / n;
    printf("  %-34s chains %d  waves/SIMD %d : %6.2f cycles per instruction per wave, %6.2f per SIMD   (clock %.0f MHz)\n", name, CHAINS,
           waves_per_simd, c, c / waves_per_simd, clk[clk.size() / 2]);
}

template <int OP, int ACC0, int X0>
static void sweep(const char* name, unsigned long long* d_st, float* d_sink) {
    for (int w = 1; w <= 4; ++w) {
        run<OP, 1, ACC0, X0>(name, w, d_st, d_sink);
        run<OP, 2, ACC0, X0>(name, w, d_st, d_sink);
        run<OP, 3, ACC0, X0>(name, w, d_st, d_sink);
        run<OP, 4, ACC0, X0>(name, w, d_st, d_sink);
        run<OP, 6, ACC0, X0>(name, w, d_st, d_sink);
        run<OP, 8, ACC0, X0>(name, w, d_st, d_sink);
    }
}

int main() {
    unsigned long long* d_st;
    float* d_sink;
    hipMalloc(&d_st, 2 * 256 * 4 * 4 * sizeof(unsigned long long));
    hipMalloc(&d_sink, 16);
    printf("part 1: dependent chains, accumulators v[16..], multiplicands v[32..] (same banks as the accumulators), weight v[4:5]\n");
    sweep<OP_PKFMA, 16, 32>("v_pk_fma_f32", d_st, d_sink);
    sweep<OP_PKFMA_BC, 16, 32>("v_pk_fma_f32 op_sel_hi:[1,0,1]", d_st, d_sink);
    sweep<OP_FMA, 16, 32>("v_fma_f32 (two per chain step)", d_st, d_sink);
    sweep<OP_PKADD, 16, 32>("v_pk_add_f32", d_st, d_sink);
    sweep<OP_PKMUL_FMA, 16, 32>("v_pk_mul + v_pk_fma (blend pair)", d_st, d_sink);
    printf("part 2: operand banks (register index mod 4): multiplicands v[34..] = banks 2,3 against accumulator banks 0,1\n");
    for (int w = 1; w <= 4; ++w) {
        run<OP_PKFMA, 8, 16, 34>("v_pk_fma_f32, x in banks 2,3", w, d_st, d_sink);
        run<OP_PKFMA, 8, 16, 32>("v_pk_fma_f32, x in banks 0,1", w, d_st, d_sink);
        run<OP_FMA, 8, 16, 34>("v_fma_f32, x two banks away", w, d_st, d_sink);
        run<OP_FMA, 8, 16, 33>("v_fma_f32, x one bank away", w, d_st, d_sink);
        run<OP_FMA, 8, 16, 32>("v_fma_f32, x same bank", w, d_st, d_sink);
    }
    return 0;
}
