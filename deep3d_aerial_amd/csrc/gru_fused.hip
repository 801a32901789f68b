// One conv-GRU cell of the slice regularisers as ONE kernel (bf16 matrix-core operands, fp32 state) -- VERDICT r03 item 2.
//
//   adamvs.py:403-427 SliceCostRegNetRED, module.py:5-51 ConvGRUCell.  Per depth slice the reference runs
//       x  = relu(conv3x3(cost))                      (conv1: C -> 8, or conv2: stride 2, 8 -> 16)
//       g  = conv3x3(cat(x, h)) + b_g;  r, u = sigmoid(g)
//       c  = tanh(conv3x3(cat(x, r * h)) + b_c)
//       h' = u * h + (1 - u) * c
//   Round 2/3 ran this as three launches of the tile kernel (csrc/conv2d_zs.hip) that move x, h, r*h, u through HBM five
//   times (72 channel-planes per cell where 24 are compulsory).  Here a
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
//   SAME K order and the same packed weights (ops._pack_z2_bf16), with the same epilogue expressions (sigmoid / tanh of common.h in
//   both) -- so h' is bit-identical to the three-launch form (tests/test_parity_gpu.py::test_gru_cell_fused_*).  P1 runs the
//   transposed GEMM (weights as the A operand: a lane then holds four channels of one pixel, one 8-byte cell write); P2 / P3 run
//   on the 16 MG-column grid of the
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
typedef unsigned u4 __attribute__((ext_vector_type(4)));

// The cell is bound by instruction issue (profiles/r04_gru_slice.txt, second part): a tile of the first build was 1650 vector
// instructions per wave for 63 MFMAs -- per-tile index arithmetic recomputed from opaque lane ids, selects behind every load,
// IEEE divisions and tanhf, 2-byte LDS writes.  This build keeps everything that does not change from tile to tile as per-lane
// state, reads and writes global memory through buffer instructions (a 32-bit per-lane offset that is OOB for pixels outside the
// image -- the hardware returns zeros / drops the store -- plus scalar tile and channel offsets: no address arithmetic and no
// select on the vector unit), writes x channel-last from the transposed GEMM (8 bytes per lane), and takes sigmoid / tanh from
// common.h.  -DD3D_GRU_PREFETCH=0|1 forces the patch prefetch off / on; -DD3D_GRU_WAVES2 the register budget of the small instance.
#ifndef D3D_GRU_WAVES2
#define D3D_GRU_WAVES2 4   // waves per SIMD the small-LDS instances are compiled for (4: two workgroups per CU, 128 registers)
#endif
constexpr int GW = 8;            // waves per workgroup
constexpr int GNT = 64 * GW;
constexpr unsigned OOB = 0xffffffffu;   // a buffer offset outside every tensor: loads return zeros, stores are dropped

struct GruParams {
    const void* cost;    // [CP, HI, WI] fp32: input of the leading convolution (HI, WI = H, W for S = 1; the finer level for S = 2);
                         // CL instances: [CP / 8, HI, WI, 8] cells of the library's 16-bit format (the sweep's CL8 plane, sweep_params.h)
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

__device__ __forceinline__ unsigned pack_h16_g(float a, float b) {
    return pack_h16x2(a, b);   // one packed conversion (common.h: pack_h16x2)
}
// raw buffer over a tensor from `origin` on (which may lie before the tensor: lanes that would read there carry OOB offsets)
__device__ __forceinline__ __amdgpu_buffer_rsrc_t tensor_rsrc(const void* origin) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(origin), 0, (int)0xfffffffeu, 0x00020000);
}
__device__ __forceinline__ f4 buf_load4(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    return __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
}
__device__ __forceinline__ float dpp_row_shl8(float v) {   // lane m of a 16-lane row takes lane m + 8
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x108, 0xf, 0xf, true));
}

// CL (S = 1 only): the cost plane arrives as 16-bit channel-last cells in 8-channel groups -- what d3d_weighted_corr_cl8_h16 writes:
// the values the planar form rounds while it stages, rounded once by the sweep instead.  A staging task is then (row, pixel, group):
// ONE 16-byte load and ONE 16-byte LDS write, no conversion (planar fp32: four 16-byte loads, eight packed conversions and four
// 8-byte writes per 16 values) and half the bytes.
template <int CP, int HID, int S, int MG, int TY>
struct GruGeom {
    static constexpr int RX = 16 * MG, RY = TY + 4;          // region of x / h / r*h
    static constexpr int OX = RX - 8;                        // output tile width: region columns 4 .. RX - 5 (whole aligned quads)
    static constexpr int PITCH = RX + 2;                     // cells per region row (column c lives at c + 1)
    static constexpr int XC = bf16_cell_bytes(HID);          // bytes per region cell (bank-conflict-free pitch: common.h)
    static constexpr int REG = ((RY * PITCH * XC + 255) / 256) * 256;   // one of X | H | R: a K group of x and one of h / r*h share a lane group -- whole bank rows apart
    static constexpr int CS1 = S == 2 ? 16 : bf16_cell_bytes(CP);   // cost cell
    static constexpr int SPX = S == 2 ? 2 * RX + 1 : RX + 2, SPY = S == 2 ? 2 * RY + 1 : RY + 2;
    static constexpr int NEVEN = RX + 1;                     // S = 2: even columns 0, 2, .. 2 RX first, then the odd ones
    static constexpr int SIMB = ((SPX * SPY * CS1 + 15) / 16) * 16;
    static constexpr int NKB1 = (9 * CP + 31) / 32, NKBG = (18 * HID + 31) / 32, NTNG = HID / 8;
    static constexpr int WB = (NKB1 + NKBG * NTNG + NKBG) * 1024;
    static constexpr int LDS = SIMB + 3 * REG + WB;
};

// A workgroup walks `tper` tiles down the image: the weights are loaded once, and (PREFETCH) the next tile's cost / state
// patches are in flight -- raw, in registers -- while the current tile is swept.
template <int CP, int HID, int S, int MG, int TY, bool CL = false>
__global__ __launch_bounds__(GNT, (GruGeom<CP, HID, S, MG, TY>::LDS <= 80 * 1024 ? D3D_GRU_WAVES2 : 2)) void gru_cell_fused_kernel(GruParams p) {
    static_assert(!CL || S == 1, "channel-last cost planes feed the stride-1 cell (the stride-2 cell reads the fp32 state of the first)");
    using G = GruGeom<CP, HID, S, MG, TY>;
    constexpr int RX = G::RX, RY = G::RY, PITCH = G::PITCH, XC = G::XC, REG = G::REG, CS1 = G::CS1, SPX = G::SPX, SPY = G::SPY;
    constexpr int NEVEN = G::NEVEN, NKB1 = G::NKB1, NKBG = G::NKBG, NTNG = G::NTNG;
#ifdef D3D_GRU_PREFETCH
    constexpr bool PREFETCH = D3D_GRU_PREFETCH != 0;
#else
    constexpr bool PREFETCH = true;
#endif
    static_assert(HID == 8 || HID == 16, "hidden state of 8 or 16 channels");
    static_assert(MG == 4 && GW == 8, "a wave's tasks: rows (wave >> 2) + 2 t of pixel group wave & 3");
    static_assert((TY * MG) % GW == 0 && (RY * MG) % GW == 0, "every wave keeps the same number of tasks");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // LDS: X | H | R regions first (their cell offsets fit the DS instructions' 16-bit immediates), cost patch, weights
    unsigned char* const XA = smem;
    unsigned char* const sim = smem + 3 * REG;
    u4* const w1l = reinterpret_cast<u4*>(sim + G::SIMB);
    u4* const wgl = w1l + NKB1 * 64;
    u4* const wcl = wgl + NKBG * NTNG * 64;

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int H = p.H, W = p.W, HI = p.HI, WI = p.WI;
    const unsigned plane4 = (unsigned)H * W * 4, iplane4 = (unsigned)HI * WI * 4;   // bytes of a channel plane (host: tensors < 2^31 bytes)
    const int rx0 = blockIdx.x * G::OX - 4;   // first region column: a multiple of 4 (the lanes' pixel quads are 16-byte aligned)
    const int nty = (H + TY - 1) / TY;
    const int t0 = blockIdx.y * p.tper, t1 = min(t0 + p.tper, nty);
    const int m = lane & 15, kg = lane >> 4;

    // ---- weights (once per workgroup) ------------------------------------------------------------------------------------
    for (int i = tid; i < NKB1 * 64; i += GNT) w1l[i] = p.w1[i];
    for (int i = tid; i < NKBG * NTNG * 64; i += GNT) wgl[i] = p.wg[i];
    for (int i = tid; i < NKBG * 64; i += GNT) wcl[i] = p.wc[i];

    // ---- staging: a task = (row, aligned quad of 4 pixels, 4 channels): four 16-byte loads (W % 4 == 0 and a region that starts
    //      at a multiple of 4: a quad is inside or outside the image as a whole) -> four 8-byte chunks of channel-last bf16 cells.
    //      cost patch: columns gxc0 .. gxc0 + SPX - 1 with gxc0 = rx0 - 1 (S = 1) | 2 rx0 - 1 (S = 2); the quads start at
    //      gxc0 - 3 (a multiple of 4) and the columns outside the patch are dropped at the commit.
    //      Per task and for good: its byte offset from the patch origin (OOB where the column is outside the image), its patch
    //      row, its first cell; per tile: rows outside the image turn the offset OOB (a uniform test skips that for inner tiles).
    constexpr int NQC = (SPX + 3 + 3) / 4, C4C = CP / 4, NTC = CL ? 1 : SPY * NQC * C4C, RC = CL ? 1 : (NTC + GNT - 1) / GNT;
    constexpr int NQH = RX / 4, C4H = HID / 4, NTH = RY * NQH * C4H, RH = (NTH + GNT - 1) / GNT;
    const int qx0 = (S == 2 ? 2 * rx0 : rx0) - 4;
    unsigned cvo[RC];    // cost task: offset | OOB
    int cpy[RC], ccell[RC];
    bool ckeep[RC][4];   // the pixel's column lies inside the patch
    if constexpr (!CL) {
#pragma unroll
        for (int r = 0; r < RC; ++r) {
            const int task = r * GNT + tid;
            const int q = task % NQC, rest = task / NQC, c4 = rest % C4C, py = rest / C4C;
            const int gx = qx0 + 4 * q;
            cvo[r] = task < NTC && gx >= 0 && gx < WI ? (unsigned)(4 * c4) * iplane4 + (unsigned)(py * WI + 4 * q) * 4 : OOB;
            cpy[r] = py;
            const int px0 = 4 * q - 3;   // patch column of the quad's first pixel (odd)
            ccell[r] = task < NTC ? (S == 2 ? py * SPX + NEVEN + ((px0 - 1) >> 1) : py * SPX + px0) * CS1 + c4 * 8 : -1;
#pragma unroll
            for (int i = 0; i < 4; ++i) ckeep[r][i] = task < NTC && px0 + i >= 0 && px0 + i < SPX;
        }
    }
    // channel-last cost plane: task = (patch row py, patch column px, 8-channel group g); patch column px is image column rx0 - 1 + px
    constexpr int G8 = CP / 8, NTL = CL ? SPY * SPX * G8 : 1, RL = CL ? (NTL + GNT - 1) / GNT : 1;
    unsigned lvo[RL];    // offset from the patch origin (row 0, column 0 of the patch, group 0) | OOB
    int lpy[RL], lcell[RL];
    if constexpr (CL) {
#pragma unroll
        for (int r = 0; r < RL; ++r) {
            const int task = r * GNT + tid;
            const int px = task % SPX, rest = task / SPX, g = rest % G8, py = rest / G8;
            const int gx = rx0 - 1 + px;
            lvo[r] = task < NTL && gx >= 0 && gx < WI ? ((unsigned)g * (unsigned)(HI * WI) + (unsigned)(py * WI + px)) * 16u : OOB;
            lpy[r] = py;
            lcell[r] = task < NTL ? (py * SPX + px) * CS1 + g * 16 : -1;
        }
    }
    unsigned hvo[RH];
    int hpy[RH], hcell[RH];
#pragma unroll
    for (int r = 0; r < RH; ++r) {
        const int task = r * GNT + tid;
        const int q = task % NQH, rest = task / NQH, c4 = rest % C4H, py = rest / C4H;
        const int gx = rx0 + 4 * q;
        hvo[r] = task < NTH && gx >= 0 && gx < W ? (unsigned)(4 * c4) * plane4 + (unsigned)(py * W + 4 * q) * 4 : OOB;
        hpy[r] = py;
        hcell[r] = task < NTH ? REG + (py * PITCH + 4 * q + 1) * XC + c4 * 8 : -1;   // (H region = XA + REG)
    }
    f4 sc[RC][4], sh[RH][4];
    u4 sl[RL];
    auto issue_cost = [&](int ty) {
        if constexpr (CL) {
            const int gy0 = ty * TY - 3;
            // (the origin may lie before the tensor: the lanes that would read there carry OOB offsets)
            const __amdgpu_buffer_rsrc_t rs = tensor_rsrc(static_cast<const unsigned char*>(p.cost) + ((long)gy0 * WI + (rx0 - 1)) * 16);
            const bool inner = gy0 >= 0 && gy0 + SPY <= HI;
#pragma unroll
            for (int r = 0; r < RL; ++r) {
                unsigned vo = lvo[r];
                if (!inner) vo = (unsigned)(lpy[r] + gy0) < (unsigned)HI ? vo : OOB;
                sl[r] = __builtin_bit_cast(u4, __builtin_amdgcn_raw_buffer_load_b128(rs, vo, 0, 0));
            }
            return;
        }
        const int gy0 = S == 2 ? 2 * (ty * TY - 2) - 1 : ty * TY - 3;
        const __amdgpu_buffer_rsrc_t rs = tensor_rsrc(static_cast<const float*>(p.cost) + ((long)gy0 * WI + qx0));
        const bool inner = gy0 >= 0 && gy0 + SPY <= HI;
#pragma unroll
        for (int r = 0; r < RC; ++r) {
            unsigned vo = cvo[r];
            if (!inner) vo = (unsigned)(cpy[r] + gy0) < (unsigned)HI ? vo : OOB;
#pragma unroll
            for (int k = 0; k < 4; ++k) sc[r][k] = buf_load4(rs, vo, k * iplane4);
        }
    };
    auto issue_state = [&](int ty) {
        const int ry0 = ty * TY - 2;
        const __amdgpu_buffer_rsrc_t rs = tensor_rsrc(p.h + ((long)ry0 * W + rx0));
        const bool inner = ry0 >= 0 && ry0 + RY <= H;
#pragma unroll
        for (int r = 0; r < RH; ++r) {
            unsigned vo = hvo[r];
            if (!inner) vo = (unsigned)(hpy[r] + ry0) < (unsigned)H ? vo : OOB;
#pragma unroll
            for (int k = 0; k < 4; ++k) sh[r][k] = buf_load4(rs, vo, k * plane4);
        }
    };
    typedef unsigned u2 __attribute__((ext_vector_type(2)));
    auto commit_cost = [&]() {
        if constexpr (CL) {
#pragma unroll
            for (int r = 0; r < RL; ++r)
                if (lcell[r] >= 0) *reinterpret_cast<u4*>(sim + lcell[r]) = sl[r];   // (outside the image the load returned zeros)
            return;
        }
#pragma unroll
        for (int r = 0; r < RC; ++r) {
            if (ccell[r] < 0) continue;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                // S = 1: consecutive cells; S = 2: pixel i of the quad is odd, even, odd, even -> the odd run, the even run
                constexpr int dummy = 0; (void)dummy;
                const int off = S == 2 ? ((i & 1) ? ((i + 1) >> 1) - NEVEN : (i >> 1)) * CS1 : i * CS1;
                if (ckeep[r][i])
                    *reinterpret_cast<u2*>(sim + ccell[r] + off) = (u2){pack_h16_g(sc[r][0][i], sc[r][1][i]), pack_h16_g(sc[r][2][i], sc[r][3][i])};
            }
        }
    };
    auto commit_state = [&]() {
#pragma unroll
        for (int r = 0; r < RH; ++r) {
            if (hcell[r] < 0) continue;
#pragma unroll
            for (int i = 0; i < 4; ++i)
                *reinterpret_cast<u2*>(XA + hcell[r] + i * XC) = (u2){pack_h16_g(sh[r][0][i], sh[r][1][i]), pack_h16_g(sh[r][2][i], sh[r][3][i])};
        }
    };

    // ---- P1 (x on the whole region): this wave's tasks are region rows (wave >> 2) + 2 t, t < RY / 2, pixel group wave & 3.
    //      Transposed GEMM (weights as the A operand): D row = channel 4 kg + register, column = pixel m -- a lane leaves with four
    //      consecutive channels of one pixel = ONE 8-byte write into the pixel's cell.
    constexpr int NT1 = RY * MG / GW;
    const int wrow = wave >> 2, wgrp = wave & 3;
    const int a1base = (S == 2 ? (2 * wrow * SPX + 16 * wgrp + m) : (wrow * SPX + 16 * wgrp + m)) * CS1;   // + t * (S == 2 ? 4 : 2) * SPX * CS1
    int a1off[NKB1];   // K index k = 32 kb + 8 kg + j -> tap k / CP = (k_y, k_x), channel k % CP (padded taps: zero weights, any valid cell)
#pragma unroll
    for (int kb = 0; kb < NKB1; ++kb) {
        const int k0 = 32 * kb + 8 * kg;
        const int t9 = k0 / CP, c = k0 % CP;
        const int ky = t9 < 9 ? t9 / 3 : 0, kx = t9 < 9 ? t9 % 3 : 0;
        a1off[kb] = (S == 2 ? (ky * SPX + ((kx & 1) ? NEVEN : 0) + (kx >> 1)) : (ky * SPX + kx)) * CS1 + (t9 < 9 ? c : 0) * 2;
    }
    const int xwr = (wrow * PITCH + 16 * wgrp + m + 1) * XC + kg * 8;      // x cell of task 0: + t * 2 * PITCH * XC
    const bool xcol = 4 * kg < HID;                                        // the lane holds real channels
    const bool xin = rx0 + 16 * wgrp + m >= 0 && rx0 + 16 * wgrp + m < W;  // its pixel's column is inside the image

    // ---- P2 / P3: core tasks = region rows 2 + (wave >> 2) + 2 t (t < NCT), pixel group wave & 3 -- gates AND candidate of
    //      the same pixels by the same wave (u and the fp32 state stay in registers); one halo task of the gates per wave
    //      (rows 1 and RY - 2).  Plain GEMM: D row = pixel 4 kg + register, column = channel m.
    constexpr int NCT = TY * MG / GW;
    constexpr int GPT = 2 * HID / 8;   // K groups of 8 channels per tap: the first half is x, the second h | r*h
    const int hid_ = min(wave, 2 * MG - 1);
    const int hrow = hid_ < MG ? 1 : RY - 2, hgrp = hid_ % MG;
    const int gbase = ((2 + wrow) * PITCH + 16 * wgrp + m) * XC;           // A operands of the core tasks: + t * 2 * PITCH * XC
    const int hbase = (hrow * PITCH + 16 * hgrp + m) * XC;                 // ... of the halo task
    int goff[NKBG], csec[NKBG];   // K group 4 kb + kg -> (tap, part): offset from the pixel's x cell; part in the second half: + REG (h), + 2 REG (r*h)
#pragma unroll
    for (int kb = 0; kb < NKBG; ++kb) {
        const int kk = 4 * kb + kg;
        const int t9 = kk / GPT, part = kk - t9 * GPT;
        const int ky = t9 < 9 ? t9 / 3 : 0, kx = t9 < 9 ? t9 % 3 : 0;
        const int second = part >= GPT / 2 ? 1 : 0;
        goff[kb] = ((ky - 1) * PITCH + kx) * XC + (part % (GPT / 2)) * 16 + second * REG;
        csec[kb] = second * REG;
    }
    const int hch = m & (HID - 1);
    // fp32 state of the lane's pixel quads (channel hch) and where h' goes: offsets from the region origin
    const bool lane_h = m < HID;
    const int ccq = 16 * wgrp + 4 * kg, hcq = 16 * hgrp + 4 * kg;         // first column of the lane's quad: core tasks, halo task
    const bool cin = rx0 + ccq >= 0 && rx0 + ccq < W, hin = rx0 + hcq >= 0 && rx0 + hcq < W;
    const unsigned hq_c = lane_h && cin ? (unsigned)hch * plane4 + (unsigned)((2 + wrow) * W + ccq) * 4 : OOB;   // + t * 2 * W * 4
    const unsigned hq_h = lane_h && hin ? (unsigned)hch * plane4 + (unsigned)(hrow * W + hcq) * 4 : OOB;
    const unsigned st_c = lane_h && cin && ccq >= 4 && ccq < RX - 4 ? hq_c : OOB;                                 // stored columns 4 .. RX - 5
    const int rwr_c = 2 * REG + ((2 + wrow) * PITCH + ccq + 1) * XC + m * 2;   // r*h cells of the lane's quad (R region = XA + 2 REG): + k * XC
    const int rwr_h = 2 * REG + (hrow * PITCH + hcq + 1) * XC + m * 2;
    const float bgr = p.bg[m], bgu = HID == 16 ? p.bg[16 + m] : 0.0f, bcm = lane_h ? p.bc[hch] : 0.0f;

    if (PREFETCH) {
        issue_cost(t0);
        issue_state(t0);
        commit_cost();
        commit_state();
    }
    __syncthreads();
    for (int ty = t0; ty < t1; ++ty) {
        const int ry0 = ty * TY - 2;
        const bool more = ty + 1 < t1;
        if (PREFETCH) {
            if (more) issue_cost(ty + 1);   // in flight during P1 (committed behind it: nothing else reads `sim`)
        } else {
            issue_cost(ty);
            issue_state(ty);
            commit_cost();
            commit_state();
            lds_barrier();
        }

        // ---- P1: x = relu(conv(cost)) on the whole region -------------------------------------------------------------------
        {
            f4 acc[NT1];
#pragma unroll
            for (int t = 0; t < NT1; ++t) acc[t] = (f4){0, 0, 0, 0};
#pragma unroll
            for (int kb = 0; kb < NKB1; ++kb) {
                const unsigned char* ap = sim + a1base + a1off[kb];
                const h16x8 b = __builtin_bit_cast(h16x8, w1l[kb * 64 + lane]);
#pragma unroll
                for (int t = 0; t < NT1; ++t) {
                    const h16x8 a = __builtin_bit_cast(h16x8, *reinterpret_cast<const u4*>(ap + t * ((S == 2 ? 4 : 2) * SPX * CS1)));
                    acc[t] = mfma_h16(b, a, acc[t]);
                }
                __builtin_amdgcn_sched_barrier(0);   // (the scheduler would hoist every K block's operand reads: ~100 registers)
            }
            if (xcol) {
#pragma unroll
                for (int t = 0; t < NT1; ++t) {
                    const int gy = ry0 + wrow + 2 * t;
                    const bool in = xin && gy >= 0 && gy < H;   // zero outside the image: the gates' own zero padding
                    f4 y = acc[t] * 1.0f + 0.0f;
                    y = __builtin_elementwise_max(y, (f4){0, 0, 0, 0});
                    u2 pk = {pack_h16_g(y[0], y[1]), pack_h16_g(y[2], y[3])};
                    if (!in) pk = (u2){0, 0};
                    *reinterpret_cast<u2*>(XA + xwr + t * (2 * PITCH * XC)) = pk;
                }
            }
        }
        lds_barrier();
        if (PREFETCH && more) {
            commit_cost();
            issue_state(ty + 1);        // in flight during P2 (committed behind it: the candidate reads x and r*h only)
        }

        // ---- P2: gates.  The fp32 state of the lane's pixel quads is requested first ----------------------------------------
        f4 ukeep[NCT], hkeep[NCT];
        {
            constexpr int NT2 = NCT + 1;
            f4 acc[NT2][NTNG], hh[NT2];
            const __amdgpu_buffer_rsrc_t rs = tensor_rsrc(p.h + ((long)ry0 * W + rx0));
#pragma unroll
            for (int t = 0; t < NT2; ++t) {
#pragma unroll
                for (int nt = 0; nt < NTNG; ++nt) acc[t][nt] = (f4){0, 0, 0, 0};
                const int gy = ry0 + (t < NCT ? 2 + wrow + 2 * t : hrow);   // (uniform)
                const unsigned vo = t < NCT ? hq_c : hq_h;
                hh[t] = buf_load4(rs, gy >= 0 && gy < H ? vo : OOB, t < NCT ? t * 2 * W * 4 : 0);
            }
#pragma unroll
            for (int kb = 0; kb < NKBG; ++kb) {
                const unsigned char* apc = XA + gbase + goff[kb];
                const unsigned char* aph = XA + hbase + goff[kb];
                h16x8 b[NTNG];
#pragma unroll
                for (int nt = 0; nt < NTNG; ++nt) b[nt] = __builtin_bit_cast(h16x8, wgl[(kb * NTNG + nt) * 64 + lane]);
#pragma unroll
                for (int t = 0; t < NT2; ++t) {
                    const h16x8 a = __builtin_bit_cast(h16x8, *reinterpret_cast<const u4*>(t < NCT ? apc + t * (2 * PITCH * XC) : aph));
#pragma unroll
                    for (int nt = 0; nt < NTNG; ++nt) acc[t][nt] = mfma_h16(a, b[nt], acc[t][nt]);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int t = 0; t < NT2; ++t) {
                if (t == NCT && wave >= 2 * MG) continue;   // (no halo task for this wave)
                f4 rgate, ugate;
                if constexpr (HID == 8) {   // one N tile: channels 0-7 reset, 8-15 update
                    f4 y = acc[t][0] * 1.0f + bgr;
#pragma unroll
                    for (int k = 0; k < 4; ++k) y[k] = gru_sigmoid(y[k]);
                    rgate = y;
#pragma unroll
                    for (int k = 0; k < 4; ++k) ugate[k] = dpp_row_shl8(y[k]);   // u of channel m arrives from lane m + 8
                } else {
                    f4 y = acc[t][0] * 1.0f + bgr, z = acc[t][NTNG - 1] * 1.0f + bgu;
#pragma unroll
                    for (int k = 0; k < 4; ++k) { y[k] = gru_sigmoid(y[k]); z[k] = gru_sigmoid(z[k]); }
                    rgate = y; ugate = z;
                }
                if (lane_h) {
                    const f4 rh = rgate * hh[t];
                    const unsigned p01 = pack_h16_g(rh[0], rh[1]), p23 = pack_h16_g(rh[2], rh[3]);
                    unsigned char* dst = XA + (t < NCT ? rwr_c + t * (2 * PITCH * XC) : rwr_h);
                    *reinterpret_cast<unsigned short*>(dst) = (unsigned short)p01;
                    *reinterpret_cast<unsigned short*>(dst + XC) = (unsigned short)(p01 >> 16);
                    *reinterpret_cast<unsigned short*>(dst + 2 * XC) = (unsigned short)p23;
                    *reinterpret_cast<unsigned short*>(dst + 3 * XC) = (unsigned short)(p23 >> 16);
                }
                if (t < NCT) { ukeep[t < NCT ? t : 0] = ugate; hkeep[t < NCT ? t : 0] = hh[t]; }
            }
        }
        lds_barrier();
        if (PREFETCH && more) commit_state();

        // ---- P3: candidate and state update on the core tasks ---------------------------------------------------------------
        {
            f4 acc[NCT];
#pragma unroll
            for (int t = 0; t < NCT; ++t) acc[t] = (f4){0, 0, 0, 0};
#pragma unroll
            for (int kb = 0; kb < NKBG; ++kb) {
                const unsigned char* ap = XA + gbase + goff[kb] + csec[kb];   // second half of a tap: r*h instead of h
                const h16x8 b = __builtin_bit_cast(h16x8, wcl[kb * 64 + lane]);
#pragma unroll
                for (int t = 0; t < NCT; ++t) {
                    const h16x8 a = __builtin_bit_cast(h16x8, *reinterpret_cast<const u4*>(ap + t * (2 * PITCH * XC)));
                    acc[t] = mfma_h16(a, b, acc[t]);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            const __amdgpu_buffer_rsrc_t ro = tensor_rsrc(p.hout + ((long)ry0 * W + rx0));
#pragma unroll
            for (int t = 0; t < NCT; ++t) {
                const int gy = ry0 + 2 + wrow + 2 * t;   // (uniform; >= 0 always)
                f4 y = acc[t] * 1.0f + bcm;
                const f4 u = ukeep[t], hq = hkeep[t];
#pragma unroll
                for (int k = 0; k < 4; ++k) y[k] = u[k] * hq[k] + (1.0f - u[k]) * gru_tanh(y[k]);
                buffer_store_b128_guarded(__builtin_bit_cast(u4, y), ro, gy < H ? st_c : OOB, t * 2 * W * 4);   // (common.h: gfx950 store-data hazard)
            }
        }
        lds_barrier();   // P3 has read X and R: the next tile's P1 may overwrite X
    }
}

template <int CP, int HID, int S, int MG, int TY, bool CL = false>
static int launch_gru(const GruParams& p, hipStream_t stream) {
    using G = GruGeom<CP, HID, S, MG, TY>;
    static_assert(G::LDS <= 160 * 1024, "tile does not fit the LDS");
    auto kern = gru_cell_fused_kernel<CP, HID, S, MG, TY, CL>;
    int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(kern), G::LDS);
    if (rc != D3D_OK) return rc;
    GruParams q = p;
    const int gx = ceil_div(p.W, G::OX), nty = ceil_div(p.H, TY);
    // tiles a workgroup walks down the image (common.h pick_tper: whole rounds of resident workgroups, the prologue -- weights and
    // per-lane task state -- amortised; with two tiles or more the next patch is in flight under the sweep)
    const int tper = pick_tper(gx, nty, G::LDS, G::WB + 16 * 1024, G::SIMB + 2 * G::REG, G::LDS <= 80 * 1024 ? 2 : 1);
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
#ifdef D3D_GRU_PREFETCH
           " D3D_GRU_PREFETCH"
#endif
#if D3D_GRU_WAVES2 != 4
           " D3D_GRU_WAVES2"
#endif
#if defined(D3D_GRU2_TY) && D3D_GRU2_TY != 8
           " D3D_GRU2_TY"
#endif
        ;
}

}  // namespace d3d

#ifndef D3D_GRU1_TY
#define D3D_GRU1_TY 8   // tile rows of the 8-channel stride-1 cell (the last cascade stage); 10 rows fit the LDS of two workgroups per CU but
                        // spill 116 bytes per lane inside 128 registers: 243 -> 524 us
#endif

using namespace d3d;

// relu(conv3x3(cost)) -> conv-GRU cell, one launch (bf16 matrix-core operands, fp32 accumulation and state).
//   stride 1: cost [CP,H,W] (CP = 8 | 16 | 32), HID = 8 (adamvs.py:409-410 conv1 + conv_gru1)
//   stride 2: cost [8,HI,WI] with H = (HI - 1) / 2 + 1, W = (WI - 1) / 2 + 1, HID = 16 (adamvs.py:411-412 conv2 + conv_gru2)
// w1 / wg / wc: ops._pack_z2_bf16 of the three nn.Conv2d weights; bg [2 HID], bc [HID] their biases (conv1 / conv2 have none).
extern "C" int d3d_gru_cell_fused_h16(const float* cost, int CP, int HI, int WI, int stride, const float* h, int HID, int H, int W,
                                       const void* w1, const void* wg, const float* bg, const void* wc, const float* bc, float* hout,
                                       d3d_stream_t stream) {
    D3D_REQUIRE(cost && h && hout && w1 && wg && wc && bg && bc, "null pointer");
    D3D_REQUIRE(H > 0 && W > 0 && HI > 0 && WI > 0, "bad dims");
    D3D_REQUIRE(h != hout, "the state is updated out of place (neighbouring tiles read the old halo)");
    D3D_REQUIRE(stride == 1 || stride == 2, "bad stride %d", stride);
    if (stride == 1) D3D_REQUIRE(HI == H && WI == W, "stride 1: the cost map has the state's size");
    else D3D_REQUIRE(H == (HI - 1) / 2 + 1 && W == (WI - 1) / 2 + 1, "stride 2: state %dx%d does not belong to a %dx%d input", H, W, HI, WI);
    if (W % 4 != 0 || WI % 4 != 0 || (long)CP * HI * WI * 4 >= (1L << 31) || (long)HID * H * W * 4 >= (1L << 31) ||
        ((reinterpret_cast<uintptr_t>(h) | reinterpret_cast<uintptr_t>(hout) | reinterpret_cast<uintptr_t>(cost)) & 15)) {
        set_error("d3d_gru_cell_fused_h16: widths %d / %d (multiples of 4) with 16-byte aligned tensors below 2 GiB needed", WI, W);
        return D3D_ERR_UNSUPPORTED;
    }
    GruParams p = {};
    p.cost = cost; p.h = h; p.hout = hout; p.w1 = reinterpret_cast<const u4*>(w1); p.wg = reinterpret_cast<const u4*>(wg);
    p.wc = reinterpret_cast<const u4*>(wc); p.bg = bg; p.bc = bc; p.H = H; p.W = W; p.HI = HI; p.WI = WI;
    hipStream_t st = (hipStream_t)stream;
    if (stride == 1 && HID == 8) {
        if (CP == 8) return launch_gru<8, 8, 1, 4, D3D_GRU1_TY>(p, st);   // (12- / 16-row tiles at one workgroup per CU: 306 / 283 against 244-254 us at stage 3)
        if (CP == 16) return launch_gru<16, 8, 1, 4, 8>(p, st);
        if (CP == 32) return launch_gru<32, 8, 1, 4, 8>(p, st);
    }
#ifndef D3D_GRU2_TY
#define D3D_GRU2_TY 8   // tile rows of the stride-2 cell: the 2 x stride input patch makes the halo expensive (129 x 17 input pixels per 56 x 4
                        // outputs = 2.45 x, 129 x 25 per 56 x 8 = 1.8 x): 8 rows are 10 - 24 % faster than 4 (20.1 -> 15.3 / 57.2 -> 49.4 /
                        // 175.5 -> 158.5 us at the three stages).  Rounds 4-5 ran 4 rows because 8 were "not bit-identical, a little
                        // differently from run to run" in output channels 12-15: the store-data hazard of gfx950 that LLVM does not
                        // guard when the store's soffset is a register (common.h buffer_store_b128_guarded) -- with more than two core
                        // tasks per wave the next task's first packed add overwrote the registers of the store just issued.
#endif
    if (stride == 2 && HID == 16 && CP == 8) return launch_gru<8, 16, 2, 4, D3D_GRU2_TY>(p, st);
    set_error("d3d_gru_cell_fused_h16: C = %d, hidden = %d, stride = %d not taken (8 | 16 | 32 -> 8 at stride 1; 8 -> 16 at stride 2)", CP, HID, stride);
    return D3D_ERR_UNSUPPORTED;
}

// The stride-1 cell with its cost plane as channel-last 16-bit cells in 8-channel groups, cost [CP / 8, H, W, 8] in the library's h16
// format: one plane of the volume d3d_weighted_corr_cl8_h16 writes.  The planar entry rounds the fp32 plane to that format while it
// stages; here the sweep has done the same rounding once, so the new state is bit for bit the planar entry's on the planar form of
// the same volume.  CP = 8 | 16 | 32, HID = 8; W a multiple of 4, 16-byte aligned tensors, else D3D_ERR_UNSUPPORTED.
extern "C" int d3d_gru_cell_fused_cl8_h16(const void* cost_cl8, int CP, const float* h, int HID, int H, int W, const void* w1, const void* wg,
                                          const float* bg, const void* wc, const float* bc, float* hout, d3d_stream_t stream) {
    D3D_REQUIRE(cost_cl8 && h && hout && w1 && wg && wc && bg && bc, "null pointer");
    D3D_REQUIRE(H > 0 && W > 0, "bad dims");
    D3D_REQUIRE(h != hout, "the state is updated out of place (neighbouring tiles read the old halo)");
    if (W % 4 != 0 || (long)CP * H * W * 2 >= (1L << 31) || (long)HID * H * W * 4 >= (1L << 31) ||
        ((reinterpret_cast<uintptr_t>(h) | reinterpret_cast<uintptr_t>(hout) | reinterpret_cast<uintptr_t>(cost_cl8)) & 15)) {
        set_error("d3d_gru_cell_fused_cl8_h16: width %d (a multiple of 4) with 16-byte aligned tensors below 2 GiB needed", W);
        return D3D_ERR_UNSUPPORTED;
    }
    GruParams p = {};
    p.cost = cost_cl8; p.h = h; p.hout = hout; p.w1 = reinterpret_cast<const u4*>(w1); p.wg = reinterpret_cast<const u4*>(wg);
    p.wc = reinterpret_cast<const u4*>(wc); p.bg = bg; p.bc = bc; p.H = H; p.W = W; p.HI = H; p.WI = W;
    hipStream_t st = (hipStream_t)stream;
    if (HID == 8) {
        if (CP == 8) return launch_gru<8, 8, 1, 4, D3D_GRU1_TY, true>(p, st);
        if (CP == 16) return launch_gru<16, 8, 1, 4, 8, true>(p, st);
        if (CP == 32) return launch_gru<32, 8, 1, 4, 8, true>(p, st);
    }
    set_error("d3d_gru_cell_fused_cl8_h16: C = %d, hidden = %d not taken (8 | 16 | 32 -> 8)", CP, HID);
    return D3D_ERR_UNSUPPORTED;
}
