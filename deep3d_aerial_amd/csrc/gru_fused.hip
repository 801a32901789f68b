// One conv-GRU cell of the slice regularisers as ONE kernel (bf16 matrix-core operands, fp32 state) -- VERDICT r03 item 2.
//
//   adamvs.py:403-427 SliceCostRegNetRED, module.py:5-51 ConvGRUCell.  Per depth slice the reference runs
//       x  = relu(conv3x3(cost))                      (conv1: C -> 8, or conv2: stride 2, 8 -> 16)
//       g  = conv3x3(cat(x, h)) + b_g;  r, u = sigmoid(g)
//       c  = tanh(conv3x3(cat(x, r * h)) + b_c)
//       h' = u * h + (1 - u) * c
//   Round 2/3 ran this as three launches of the tile kernel (csrc/conv2d_zs.hip) that move x, h, r*h, u through HBM five
//   times (72 channel-planes per cell where 24 are compulsory) and are latency-bound at the cascade's image sizes.  Here a
//   workgroup owns output tiles of (16 MG - 8) x TY pixels and keeps everything between `cost` / `h` and `h'` in LDS:
//
//     stage   cost patch (halo 3) and h patch (halo 2): planar fp32 -> channel-last bf16 cells (RNE), zeros outside the image
//     P1      x on the (16 MG) x (TY + 4) region  -> bf16 cells X (zero outside the image: the gates' own zero padding)
//     P2      gates on rows 1 .. TY + 2 of that region: r * h -> bf16 cells R; u stays in the registers of the wave that will
//             also sweep the candidate of the same pixels
//     P3      candidate on rows 2 .. TY + 1, h' = u h + (1 - u) tanh(c) stored for columns 4 .. 16 MG - 5 (whole 16-byte quads:
//             the region starts at a multiple of 4, W % 4 == 0)
//
//   Every phase is the implicit GEMM of conv2d_zs.hip (M = 16 consecutive pixels of a region row, N = 16 output channels,
//   K = (k_y, k_x, c_in) in blocks of 32 = v_mfma_f32_16x16x32_bf16; an A operand is one ds_read_b128 of 8 channels), on the
//   SAME K order and the same packed weights (ops._pack_z2_bf16), with the same epilogue expressions -- so h' is bit-identical to
//   the three-launch form (tests/test_parity_gpu.py::test_gru_cell_fused_*; v_exp / v_rcp forms of sigmoid and tanh were
//   measured -- 411 -> 405 us at stage 3 -- and dropped: a cell is a chain of latencies, not an instruction count).  All phases run on the 16 MG-column grid of the
//   region: the outermost columns of P2 / P3 read one cell beyond the region (row pitch 16 MG + 2 cells) and their results are
//   dropped, which keeps a lane's four pixels the same in P2 and P3 (u never leaves its registers).
//
//   S = 2: the leading convolution is the stride-2 ConvReLU(8, 16) (adamvs.py:411); its 8-channel input patch keeps the even
//   and the odd columns of a row in separate runs (as conv2d_s2_zs_bf16_kernel), so 16 consecutive outputs read 16
//   consecutive cells.
#include <cstdint>
#include "common.h"

namespace d3d {

namespace {

typedef float f4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u4 __attribute__((ext_vector_type(4)));

// D3D_GRU_X (timing experiments only, results wrong; never set in the production build: d3d_build_flags reports it):
//   1 no staging loads | 2 no h loads in the gate epilogue | 4 no exp / tanh | 8 no stores | 16 no matrix-core loops
#ifndef D3D_GRU_X
#define D3D_GRU_X 0
#endif
// The next tile's patches are requested a phase ahead where a workgroup owns its CU (LDS > 80 KB: 256 registers, +3 .. 7 %);
// the 8-channel stride-1 instance shares the CU with a second workgroup at 128 registers, where the prefetched patches spill
// (593 against 411 us at stage 3) -- there the other workgroup is what overlaps the loads.  -DD3D_GRU_PREFETCH=0|1 forces it.
#ifndef D3D_GRU_WAVES2
#define D3D_GRU_WAVES2 4   // waves per SIMD the small-LDS instances are compiled for (4: two workgroups per CU, 128 registers)
#endif
constexpr int GW = 8;            // waves per workgroup
constexpr int GNT = 64 * GW;

struct GruParams {
    const float* cost;   // [CP, HI, WI]: input of the leading convolution (HI, WI = H, W for S = 1; the finer level for S = 2)
    const float* h;      // [HID, H, W] state in
    float* hout;         // [HID, H, W] state out (must not alias h: neighbouring tiles read its halo)
    const u4* w1;        // leading convolution, [NKB1][1][64] B fragments (ops._pack_z2_bf16)
    const u4* wg;        // gates   [NKBG][NTNG][64]
    const u4* wc;        // candidate [NKBG][1][64]
    const float* bg;     // [2 HID]
    const float* bc;     // [HID]
    int H, W;            // the cell's level
    int HI, WI;          // the leading convolution's input level
    int tper;            // tiles per workgroup along y
};

__device__ __forceinline__ unsigned pack_bf16_g(float a, float b) {
    return pack_bf16x2(a, b);   // one v_cvt_pk_bf16_f32 (common.h)
}
__device__ __forceinline__ unsigned short bf16_bits(float a) {
    const __bf16 x = (__bf16)a;
    return __builtin_bit_cast(unsigned short, x);
}

template <int CP, int HID, int S, int MG, int TY>
struct GruGeom {
    static constexpr int RX = 16 * MG, RY = TY + 4;          // region of x / h / r*h
    static constexpr int OX = RX - 8;                        // output tile width: region columns 4 .. RX - 5 (whole aligned quads)
    static constexpr int PITCH = RX + 2;                     // cells per region row (column c lives at c + 1)
    static constexpr int XC = HID == 8 ? 16 : 48;            // bytes per region cell (an odd number of 16-byte slots)
    static constexpr int REG = RY * PITCH * XC;              // one of X | H | R
    static constexpr int CS1 = S == 2 ? 16 : CP * 2 + (CP > 8 ? 16 : 0);   // cost cell
    static constexpr int SPX = S == 2 ? 2 * RX + 1 : RX + 2, SPY = S == 2 ? 2 * RY + 1 : RY + 2;
    static constexpr int NEVEN = RX + 1;                     // S = 2: even columns 0, 2, .. 2 RX first, then the odd ones
    static constexpr int SIMB = ((SPX * SPY * CS1 + 15) / 16) * 16;
    static constexpr int NKB1 = (9 * CP + 31) / 32, NKBG = (18 * HID + 31) / 32, NTNG = HID / 8;
    static constexpr int WB = (NKB1 + NKBG * NTNG + NKBG) * 1024;
    static constexpr int LDS = SIMB + 3 * REG + WB;
};

// A workgroup walks `tper` tiles down the image: the weights are loaded once, and the next tile's cost / state patches are
// in flight (registers) while the current tile is swept -- at these image sizes a cell is a chain of latencies (patch
// loads, three barriers, the state's fp32 reload for the epilogues), not arithmetic: profiles/r04_gru_slice.txt.
template <int CP, int HID, int S, int MG, int TY>
__global__ __launch_bounds__(GNT, (GruGeom<CP, HID, S, MG, TY>::LDS <= 80 * 1024 ? D3D_GRU_WAVES2 : 2)) void gru_cell_fused_kernel(GruParams p) {   // two workgroups per CU where the LDS allows: 128 registers
    using G = GruGeom<CP, HID, S, MG, TY>;
    constexpr int RX = G::RX, RY = G::RY, PITCH = G::PITCH, XC = G::XC, REG = G::REG, CS1 = G::CS1, SPX = G::SPX, SPY = G::SPY;
    constexpr int NEVEN = G::NEVEN, NKB1 = G::NKB1, NKBG = G::NKBG, NTNG = G::NTNG;
#ifdef D3D_GRU_PREFETCH
    constexpr bool PREFETCH = D3D_GRU_PREFETCH != 0;
#else
    constexpr bool PREFETCH = G::LDS > 80 * 1024;
#endif
    static_assert(HID == 8 || HID == 16, "hidden state of 8 or 16 channels");
    static_assert((TY * MG) % GW == 0, "every wave keeps the same number of candidate tasks");
    static_assert(2 * MG <= GW, "one halo-row task per wave at most");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* sim = smem;
    unsigned char* XA = smem + G::SIMB;
    unsigned char* HA = XA + REG;
    unsigned char* RA = HA + REG;
    u4* w1l = reinterpret_cast<u4*>(RA + REG);
    u4* wgl = w1l + NKB1 * 64;
    u4* wcl = wgl + NKBG * NTNG * 64;

    int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int H = p.H, W = p.W;
    const size_t plane = (size_t)H * W, iplane = (size_t)p.HI * p.WI;
    const int rx0 = blockIdx.x * G::OX - 4;   // first region column: a multiple of 4 (the lanes' pixel quads are 16-byte aligned)
    const int nty = (H + TY - 1) / TY;
    const int t0 = blockIdx.y * p.tper, t1 = min(t0 + p.tper, nty);
    int m = lane & 15, kgroup = lane >> 4;

    // ---- weights (once per workgroup) ------------------------------------------------------------------------------------
    for (int i = tid; i < NKB1 * 64; i += GNT) w1l[i] = p.w1[i];
    for (int i = tid; i < NKBG * NTNG * 64; i += GNT) wgl[i] = p.wg[i];
    for (int i = tid; i < NKBG * 64; i += GNT) wcl[i] = p.wc[i];

    // ---- staging: a task = (row, aligned quad of 4 pixels, 4 channels): four dwordx4 loads (W % 4 == 0 and a region that starts
    //      at a multiple of 4: a quad is inside or outside the image as a whole) -> four 8-byte chunks of channel-last bf16
    //      cells; zeros outside the image.  issue_*() leaves the values in registers, commit_*() writes the cells: the loads of
    //      tile t + 1 fly during P1 (cost) and P2 (state) of tile t.
    //      cost patch: columns gxc0 .. gxc0 + SPX - 1 with gxc0 = rx0 - 1 (S = 1) | 2 rx0 - 1 (S = 2); the quads start at
    //      gxc0 - 3 (a multiple of 4) and the 3 + (4 NQC - SPX - 3) columns outside the patch are dropped at the commit.
    constexpr int NQC = (SPX + 3 + 3) / 4, C4C = CP / 4, NTC = SPY * NQC * C4C, RC = (NTC + GNT - 1) / GNT;
    constexpr int NQH = RX / 4, C4H = HID / 4, NTH = RY * NQH * C4H, RH = (NTH + GNT - 1) / GNT;
    f4 sc[RC][4], sh[RH][4];
    auto issue_cost = [&](int ty) {
        const int ry0 = ty * TY - 2;
        const int qx0 = (S == 2 ? 2 * rx0 : rx0) - 4, gy0 = S == 2 ? 2 * ry0 - 1 : ry0 - 1;
#pragma unroll
        for (int r = 0; r < RC; ++r) {
            const int task = r * GNT + tid;
            const int q = task % NQC, rest = task / NQC, c4 = rest % C4C, py = rest / C4C;
            const int gx = qx0 + 4 * q, gy = gy0 + py;
            const bool ok = task < NTC && gx >= 0 && gx < p.WI && gy >= 0 && gy < p.HI;
            const float* __restrict__ src = p.cost + (size_t)(4 * c4) * iplane + (ok ? (size_t)gy * p.WI + gx : 0);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const f4 t = (D3D_GRU_X & 1) ? (f4){0.5f, 0.5f, 0.5f, 0.5f} : *reinterpret_cast<const f4*>(src + (size_t)k * iplane);
                sc[r][k] = ok ? t : (f4){0, 0, 0, 0};
            }
        }
    };
    auto issue_state = [&](int ty) {
        const int ry0 = ty * TY - 2;
#pragma unroll
        for (int r = 0; r < RH; ++r) {
            const int task = r * GNT + tid;
            const int q = task % NQH, rest = task / NQH, c4 = rest % C4H, py = rest / C4H;
            const int gx = rx0 + 4 * q, gy = ry0 + py;
            const bool ok = task < NTH && gx >= 0 && gx < W && gy >= 0 && gy < H;
            const float* __restrict__ src = p.h + (size_t)(4 * c4) * plane + (ok ? (size_t)gy * W + gx : 0);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const f4 t = (D3D_GRU_X & 1) ? (f4){0.25f, 0.25f, 0.25f, 0.25f} : *reinterpret_cast<const f4*>(src + (size_t)k * plane);
                sh[r][k] = ok ? t : (f4){0, 0, 0, 0};
            }
        }
    };
    typedef unsigned u2 __attribute__((ext_vector_type(2)));
    auto commit_cost = [&]() {
#pragma unroll
        for (int r = 0; r < RC; ++r) {
            const int task = r * GNT + tid;
            if (task < NTC) {
                const int q = task % NQC, rest = task / NQC, c4 = rest % C4C, py = rest / C4C;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int px = 4 * q + i - 3;   // column of the patch
                    if (px >= 0 && px < SPX) {
                        const int cell = S == 2 ? py * SPX + ((px & 1) ? NEVEN + (px >> 1) : (px >> 1)) : py * SPX + px;
                        *reinterpret_cast<u2*>(sim + cell * CS1 + c4 * 8) =
                            (u2){pack_bf16_g(sc[r][0][i], sc[r][1][i]), pack_bf16_g(sc[r][2][i], sc[r][3][i])};
                    }
                }
            }
        }
    };
    auto commit_state = [&]() {
#pragma unroll
        for (int r = 0; r < RH; ++r) {
            const int task = r * GNT + tid;
            if (task < NTH) {
                const int q = task % NQH, rest = task / NQH, c4 = rest % C4H, py = rest / C4H;
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    *reinterpret_cast<u2*>(HA + (py * PITCH + 4 * q + i + 1) * XC + c4 * 8) =
                        (u2){pack_bf16_g(sh[r][0][i], sh[r][1][i]), pack_bf16_g(sh[r][2][i], sh[r][3][i])};
            }
        }
    };

    // K group kk = 4 kb + kgroup (8 channels each) of the gates / candidate: tap kk / GPT, part kk % GPT -- the first half of a
    // tap's channels is x, the second h | r*h
    constexpr int GPT = 2 * HID / 8;
    auto a_off = [&](int kb, int& second) {
        const int kk = 4 * kb + kgroup;
        const int t9 = kk / GPT, part = kk - t9 * GPT;
        const int ky = t9 < 9 ? t9 / 3 : 0, kx = t9 < 9 ? t9 % 3 : 0;
        second = part >= GPT / 2 ? 1 : 0;
        return ((ky - 1) * PITCH + kx) * XC + (part % (GPT / 2)) * 16;
    };
    const int hch = m & (HID - 1);
    constexpr int NCT = TY * MG / GW;   // core tasks of a wave: rows 2 .. TY + 1 of the region, gates AND candidate by the same wave
    int crow[NCT], cgrp[NCT];
#pragma unroll
    for (int t = 0; t < NCT; ++t) {
        const int id = wave + GW * t;
        crow[t] = 2 + id / MG; cgrp[t] = id % MG;
    }
    const int hid_ = min(wave, 2 * MG - 1);   // halo task of the gates (waves 0 .. 2 MG - 1): rows 1 and TY + 2
    const int hrow = hid_ < MG ? 1 : RY - 2, hgrp = hid_ % MG;

    if (PREFETCH) {
        issue_cost(t0);
        issue_state(t0);
        commit_cost();
        commit_state();
    }
    __syncthreads();
    for (int ty = t0; ty < t1; ++ty) {
        // (opaque per tile: otherwise every phase's index arithmetic -- tile-invariant -- is hoisted out of this loop and kept in
        //  ~100 registers, and the two-workgroups-per-CU build spills)
        asm volatile("" : "+v"(tid), "+v"(m), "+v"(kgroup));
        const int ry0 = ty * TY - 2;
        const bool more = ty + 1 < t1;
        if (PREFETCH) {
            if (more) issue_cost(ty + 1);   // in flight during P1 (committed behind it: nothing else reads `sim`)
        } else {
            issue_cost(ty);
            issue_state(ty);
            commit_cost();
            commit_state();
            __syncthreads();
        }

        // ---- P1: x = relu(conv(cost)) on the whole region -------------------------------------------------------------------
        {
            constexpr int NT1 = (RY * MG + GW - 1) / GW;
            f4 acc[NT1];
            int base[NT1];
#pragma unroll
            for (int t = 0; t < NT1; ++t) {
                acc[t] = (f4){0, 0, 0, 0};
                const int id = min(wave + GW * t, RY * MG - 1);
                const int r = id / MG, g = id - r * MG;
                base[t] = S == 2 ? (2 * r * SPX + 16 * g + m) * CS1 : (r * SPX + 16 * g + m) * CS1;
            }
#pragma unroll
            for (int kb = 0; kb < ((D3D_GRU_X & 16) ? 1 : NKB1); ++kb) {
                // K index k = 32 kb + 8 kgroup + j -> tap k / CP = (k_y, k_x), channel k % CP (padded taps: zero weights, any valid cell)
                const int k0 = 32 * kb + 8 * kgroup;
                const int t9 = k0 / CP, c = k0 % CP;
                const int ky = t9 < 9 ? t9 / 3 : 0, kx = t9 < 9 ? t9 % 3 : 0;
                const int aoff = S == 2 ? (ky * SPX + ((kx & 1) ? NEVEN : 0) + (kx >> 1)) * CS1 + (t9 < 9 ? c : 0) * 2
                                        : (ky * SPX + kx) * CS1 + (t9 < 9 ? c : 0) * 2;
                const bf16x8 b = __builtin_bit_cast(bf16x8, w1l[kb * 64 + lane]);
#pragma unroll
                for (int t = 0; t < NT1; ++t) {
                    const bf16x8 a = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u4*>(sim + base[t] + aoff));
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[t], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);   // (the scheduler would hoist every K block's operand reads: ~100 registers)
            }
            // D row (pixel) = 4 (lane >> 4) + register, column (channel) = lane & 15
#pragma unroll
            for (int t = 0; t < NT1; ++t) {
                const int id = wave + GW * t;
                if (id < RY * MG && m < HID) {
                    const int r = id / MG, g = id - r * MG;
                    const int gy = ry0 + r;
                    const int cc = 16 * g + 4 * kgroup;
                    const int gx = rx0 + cc;
                    const bool in = gx >= 0 && gx < W && gy >= 0 && gy < H;   // (W % 4 == 0: a quad is inside or outside as a whole)
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const float y = fmaxf(acc[t][k] * 1.0f + 0.0f, 0.0f);
                        *reinterpret_cast<unsigned short*>(XA + (r * PITCH + cc + k + 1) * XC + m * 2) = in ? bf16_bits(y) : (unsigned short)0;
                    }
                }
            }
        }
        __syncthreads();
        if (PREFETCH && more) {
            commit_cost();
            issue_state(ty + 1);        // in flight during P2 (committed behind it: the candidate reads x and r*h only)
        }

        // ---- P2: gates.  Core tasks and one halo task; the fp32 state of the lane's pixel quads is requested first -----------
        f4 ukeep[NCT], hkeep[NCT];
        {
            constexpr int NT2 = NCT + 1;
            f4 acc[NT2][NTNG], hh[NT2];
            int base[NT2], rr[NT2], gg[NT2];
#pragma unroll
            for (int t = 0; t < NT2; ++t) {
#pragma unroll
                for (int nt = 0; nt < NTNG; ++nt) acc[t][nt] = (f4){0, 0, 0, 0};
                rr[t] = t < NCT ? crow[t < NCT ? t : 0] : hrow;
                gg[t] = t < NCT ? cgrp[t < NCT ? t : 0] : hgrp;
                base[t] = (rr[t] * PITCH + 16 * gg[t] + m) * XC;
                const int gy = ry0 + rr[t], gx = rx0 + 16 * gg[t] + 4 * kgroup;
                const bool in = gx >= 0 && gx < W && gy >= 0 && gy < H;
                const f4 v = (D3D_GRU_X & 2) ? (f4){0.125f, 0.125f, 0.125f, 0.125f}
                                             : *reinterpret_cast<const f4*>(p.h + (size_t)hch * plane + (in ? (size_t)gy * W + gx : 0));
                hh[t] = in ? v : (f4){0, 0, 0, 0};
            }
#pragma unroll
            for (int kb = 0; kb < ((D3D_GRU_X & 16) ? 1 : NKBG); ++kb) {
                int second;
                const int aoff = a_off(kb, second);
                const unsigned char* arr = second ? HA : XA;
                bf16x8 b[NTNG];
#pragma unroll
                for (int nt = 0; nt < NTNG; ++nt) b[nt] = __builtin_bit_cast(bf16x8, wgl[(kb * NTNG + nt) * 64 + lane]);
#pragma unroll
                for (int t = 0; t < NT2; ++t) {
                    const bf16x8 a = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u4*>(arr + base[t] + aoff));
#pragma unroll
                    for (int nt = 0; nt < NTNG; ++nt) acc[t][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b[nt], acc[t][nt], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int t = 0; t < NT2; ++t) {
                if (t == NCT && wave >= 2 * MG) continue;   // (no halo task for this wave)
                f4 rgate, ugate;
                if constexpr (HID == 8) {   // one N tile: channels 0-7 reset, 8-15 update
                    const float bs = p.bg[m];
                    f4 y = acc[t][0] * 1.0f + bs;
#pragma unroll
                    for (int k = 0; k < 4; ++k) y[k] = (D3D_GRU_X & 4) ? y[k] : 1.0f / (1.0f + __expf(-y[k]));
                    rgate = y;
#pragma unroll
                    for (int k = 0; k < 4; ++k) ugate[k] = __shfl_down(y[k], 8, 16);   // u of channel m arrives from lane m + 8
                } else {
                    const float shr = p.bg[m], shu = p.bg[16 + m];
                    f4 y = acc[t][0] * 1.0f + shr, z = acc[t][NTNG - 1] * 1.0f + shu;
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        y[k] = (D3D_GRU_X & 4) ? y[k] : 1.0f / (1.0f + __expf(-y[k]));
                        z[k] = (D3D_GRU_X & 4) ? z[k] : 1.0f / (1.0f + __expf(-z[k]));
                    }
                    rgate = y; ugate = z;
                }
                if (m < HID) {
                    const f4 rh = rgate * hh[t];
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                        *reinterpret_cast<unsigned short*>(RA + (rr[t] * PITCH + 16 * gg[t] + 4 * kgroup + k + 1) * XC + m * 2) = bf16_bits(rh[k]);
                }
                if (t < NCT) { ukeep[t < NCT ? t : 0] = ugate; hkeep[t < NCT ? t : 0] = hh[t]; }
            }
        }
        __syncthreads();
        if (PREFETCH && more) commit_state();

        // ---- P3: candidate and state update on the core tasks ---------------------------------------------------------------
        {
            f4 acc[NCT];
            int base[NCT];
#pragma unroll
            for (int t = 0; t < NCT; ++t) {
                acc[t] = (f4){0, 0, 0, 0};
                base[t] = (crow[t] * PITCH + 16 * cgrp[t] + m) * XC;
            }
#pragma unroll
            for (int kb = 0; kb < ((D3D_GRU_X & 16) ? 1 : NKBG); ++kb) {
                int second;
                const int aoff = a_off(kb, second);
                const unsigned char* arr = second ? RA : XA;
                const bf16x8 b = __builtin_bit_cast(bf16x8, wcl[kb * 64 + lane]);
#pragma unroll
                for (int t = 0; t < NCT; ++t) {
                    const bf16x8 a = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u4*>(arr + base[t] + aoff));
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[t], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            if (m < HID) {
                const float bs = p.bc[m];
#pragma unroll
                for (int t = 0; t < NCT; ++t) {
                    const int gy = ry0 + crow[t];
                    const int cc = 16 * cgrp[t] + 4 * kgroup;
                    const int gx = rx0 + cc;
                    f4 y = acc[t] * 1.0f + bs;
                    const f4 u = ukeep[t], hq = hkeep[t];
#pragma unroll
                    for (int k = 0; k < 4; ++k) y[k] = u[k] * hq[k] + (1.0f - u[k]) * ((D3D_GRU_X & 4) ? y[k] : tanhf(y[k]));
                    if (!(D3D_GRU_X & 8) || y[0] == 1234.5f)
                        if (cc >= 4 && cc < RX - 4 && gx < W && gy < H) *reinterpret_cast<f4*>(p.hout + (size_t)m * plane + (size_t)gy * W + gx) = y;
                }
            }
        }
        __syncthreads();   // P3 has read X and R: the next tile's P1 may overwrite X
    }
}

template <int CP, int HID, int S, int MG, int TY>
static int launch_gru(const GruParams& p, hipStream_t stream) {
    using G = GruGeom<CP, HID, S, MG, TY>;
    static_assert(G::LDS <= 160 * 1024, "tile does not fit the LDS");
    auto kern = gru_cell_fused_kernel<CP, HID, S, MG, TY>;
    int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(kern), G::LDS);
    if (rc != D3D_OK) return rc;
    GruParams q = p;
    const int gx = ceil_div(p.W, G::OX), nty = ceil_div(p.H, TY);
    int tper = 8;   // tiles a workgroup walks down the image: enough workgroups for 256 CUs come first
    while (tper > 1 && (long)gx * ceil_div(nty, tper) < 1024) tper >>= 1;
    q.tper = tper;
    const int gy = ceil_div(nty, tper);
    if (gy > 65535) return D3D_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(kern, dim3(gx, gy), dim3(GNT), G::LDS, stream, q);
    D3D_LAUNCH_CHECK("gru_cell_fused_kernel launch");
    return D3D_OK;
}

}  // namespace

// Non-default compile-time knobs of this translation unit (d3d_build_flags): empty for the production build.
const char* gru_build_flags() {
    return ""
#if D3D_GRU_X
           " D3D_GRU_X"
#endif
#ifdef D3D_GRU_PREFETCH
           " D3D_GRU_PREFETCH"
#endif
#if D3D_GRU_WAVES2 != 4
           " D3D_GRU_WAVES2"
#endif
#ifdef D3D_GRU_TY_C8
           " D3D_GRU_TY_C8"
#endif
        ;
}

}  // namespace d3d

using namespace d3d;

// relu(conv3x3(cost)) -> conv-GRU cell, one launch (bf16 matrix-core operands, fp32 accumulation and state).
//   stride 1: cost [CP,H,W] (CP = 8 | 16 | 32), HID = 8 (adamvs.py:409-410 conv1 + conv_gru1)
//   stride 2: cost [8,HI,WI] with H = (HI - 1) / 2 + 1, W = (WI - 1) / 2 + 1, HID = 16 (adamvs.py:411-412 conv2 + conv_gru2)
// w1 / wg / wc: ops._pack_z2_bf16 of the three nn.Conv2d weights; bg [2 HID], bc [HID] their biases (conv1 / conv2 have none).
extern "C" int d3d_gru_cell_fused_bf16(const float* cost, int CP, int HI, int WI, int stride, const float* h, int HID, int H, int W,
                                       const void* w1, const void* wg, const float* bg, const void* wc, const float* bc, float* hout,
                                       d3d_stream_t stream) {
    D3D_REQUIRE(cost && h && hout && w1 && wg && wc && bg && bc, "null pointer");
    D3D_REQUIRE(H > 0 && W > 0 && HI > 0 && WI > 0, "bad dims");
    D3D_REQUIRE(h != hout, "the state is updated out of place (neighbouring tiles read the old halo)");
    D3D_REQUIRE(stride == 1 || stride == 2, "bad stride %d", stride);
    if (stride == 1) D3D_REQUIRE(HI == H && WI == W, "stride 1: the cost map has the state's size");
    else D3D_REQUIRE(H == (HI - 1) / 2 + 1 && W == (WI - 1) / 2 + 1, "stride 2: state %dx%d does not belong to a %dx%d input", H, W, HI, WI);
    if (W % 4 != 0 || WI % 4 != 0 ||
        ((reinterpret_cast<uintptr_t>(h) | reinterpret_cast<uintptr_t>(hout) | reinterpret_cast<uintptr_t>(cost)) & 15)) {
        set_error("d3d_gru_cell_fused_bf16: widths %d / %d (multiples of 4) with 16-byte aligned tensors needed", WI, W);
        return D3D_ERR_UNSUPPORTED;
    }
    GruParams p = {};
    p.cost = cost; p.h = h; p.hout = hout; p.w1 = reinterpret_cast<const u4*>(w1); p.wg = reinterpret_cast<const u4*>(wg);
    p.wc = reinterpret_cast<const u4*>(wc); p.bg = bg; p.bc = bc; p.H = H; p.W = W; p.HI = HI; p.WI = WI;
    hipStream_t st = (hipStream_t)stream;
    if (stride == 1 && HID == 8) {
#ifdef D3D_GRU_TY_C8
        if (CP == 8) return launch_gru<8, 8, 1, 4, D3D_GRU_TY_C8>(p, st);
#else
        if (CP == 8) return launch_gru<8, 8, 1, 4, 8>(p, st);
#endif
        if (CP == 16) return launch_gru<16, 8, 1, 4, 8>(p, st);
        if (CP == 32) return launch_gru<32, 8, 1, 4, 8>(p, st);
    }
    if (stride == 2 && HID == 16 && CP == 8) return launch_gru<8, 16, 2, 4, 4>(p, st);
    set_error("d3d_gru_cell_fused_bf16: C = %d, hidden = %d, stride = %d not taken (8 | 16 | 32 -> 8 at stride 1; 8 -> 16 at stride 2)", CP, HID, stride);
    return D3D_ERR_UNSUPPORTED;
}
