// One conv-GRU cell of the slice regularisers as ONE kernel (bf16 matrix-core operands, fp32 state) -- VERDICT r03 item 2.
//
//   adamvs.py:403-427 SliceCostRegNetRED, module.py:5-51 ConvGRUCell.  Per depth slice the reference runs
//       x  = relu(conv3x3(cost))                      (conv1: C -> 8, or conv2: stride 2, 8 -> 16)
//       g  = conv3x3(cat(x, h)) + b_g;  r, u = sigmoid(g)
//       c  = tanh(conv3x3(cat(x, r * h)) + b_c)
//       h' = u * h + (1 - u) * c
//   Round 2/3 ran this as three launches of the tile kernel (csrc/conv2d_zs.hip) that move x, h, r*h, u through HBM five
//   times (72 channel-planes per cell where 24 are compulsory) and are latency-bound at the cascade's image sizes.  Here a
//   workgroup owns an output tile of (16 MG - 4) x TY pixels and keeps everything between `cost` / `h` and `h'` in LDS:
//
//     stage   cost patch (halo 3) and h patch (halo 2): planar fp32 -> channel-last bf16 cells (RNE), zeros outside the image
//     P1      x on the (16 MG) x (TY + 4) region  -> bf16 cells X (zero outside the image: the gates' own zero padding)
//     P2      gates on rows 1 .. TY + 2 of that region: r * h -> bf16 cells R; u stays in the registers of the wave that will
//             also sweep the candidate of the same pixels
//     P3      candidate on rows 2 .. TY + 1, h' = u h + (1 - u) tanh(c) stored for columns 2 .. 16 MG - 3
//
//   Every phase is the implicit GEMM of conv2d_zs.hip (M = 16 consecutive pixels of a region row, N = 16 output channels,
//   K = (k_y, k_x, c_in) in blocks of 32 = v_mfma_f32_16x16x32_bf16; an A operand is one ds_read_b128 of 8 channels), on the
//   SAME K order and the same packed weights (ops._pack_z2_bf16), with the same epilogue expressions -- so h' is bit-identical to
//   the three-launch form (tests/test_parity_gpu.py::test_gru_cell_fused_*).  All phases run on the 16 MG-column grid of the
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
};

__device__ __forceinline__ unsigned pack_bf16_g(float a, float b) {
    const __bf16 x = (__bf16)a, y = (__bf16)b;   // v_cvt_pk_bf16_f32: RNE
    return (unsigned)__builtin_bit_cast(unsigned short, x) | ((unsigned)__builtin_bit_cast(unsigned short, y) << 16);
}
__device__ __forceinline__ unsigned short bf16_bits(float a) {
    const __bf16 x = (__bf16)a;
    return __builtin_bit_cast(unsigned short, x);
}

template <int CP, int HID, int S, int MG, int TY>
struct GruGeom {
    static constexpr int RX = 16 * MG, RY = TY + 4;          // region of x / h / r*h
    static constexpr int OX = RX - 4;                        // output tile width
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

template <int CP, int HID, int S, int MG, int TY>
__global__ __launch_bounds__(GNT, 1) void gru_cell_fused_kernel(GruParams p) {
    using G = GruGeom<CP, HID, S, MG, TY>;
    constexpr int RX = G::RX, RY = G::RY, PITCH = G::PITCH, XC = G::XC, REG = G::REG, CS1 = G::CS1, SPX = G::SPX, SPY = G::SPY;
    constexpr int NEVEN = G::NEVEN, NKB1 = G::NKB1, NKBG = G::NKBG, NTNG = G::NTNG;
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

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int H = p.H, W = p.W;
    const size_t plane = (size_t)H * W, iplane = (size_t)p.HI * p.WI;
    const int ox0 = blockIdx.x * G::OX, oy0 = blockIdx.y * TY;   // output tile origin; the region starts 2 pixels up / left
    const int m = lane & 15, kgroup = lane >> 4;

    // ---- weights ------------------------------------------------------------------------------------------------------
    for (int i = tid; i < NKB1 * 64; i += GNT) w1l[i] = p.w1[i];
    for (int i = tid; i < NKBG * NTNG * 64; i += GNT) wgl[i] = p.wg[i];
    for (int i = tid; i < NKBG * 64; i += GNT) wcl[i] = p.wc[i];

    // ---- stage the cost patch and the state patch: a task = (pixel, 8 channels), eight dword loads -> one 16-byte cell chunk ---
    {
        constexpr int G8 = CP / 8, NTASK = SPX * SPY * G8;
        const int gx0 = S == 2 ? 2 * (ox0 - 2) - 1 : ox0 - 3, gy0 = S == 2 ? 2 * (oy0 - 2) - 1 : oy0 - 3;
#pragma unroll 2
        for (int t0 = 0; t0 < NTASK; t0 += GNT) {
            const int task = t0 + tid;
            const int pix = task / G8, g8 = task - pix * G8;
            const int py = pix / SPX, px = pix - py * SPX;
            const int gx = gx0 + px, gy = gy0 + py;
            const bool ok = task < NTASK && gx >= 0 && gx < p.WI && gy >= 0 && gy < p.HI;
            const float* __restrict__ src = p.cost + (size_t)(8 * g8) * iplane + (ok ? (size_t)gy * p.WI + gx : 0);
            float v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const float t = src[(size_t)k * iplane];
                v[k] = ok ? t : 0.0f;
            }
            if (task < NTASK) {
                const int cell = S == 2 ? py * SPX + ((px & 1) ? NEVEN + (px >> 1) : (px >> 1)) : py * SPX + px;
                *reinterpret_cast<u4*>(sim + cell * CS1 + g8 * 16) =
                    (u4){pack_bf16_g(v[0], v[1]), pack_bf16_g(v[2], v[3]), pack_bf16_g(v[4], v[5]), pack_bf16_g(v[6], v[7])};
            }
        }
    }
    {
        constexpr int G8 = HID / 8, NTASK = RX * RY * G8;
#pragma unroll 2
        for (int t0 = 0; t0 < NTASK; t0 += GNT) {
            const int task = t0 + tid;
            const int pix = task / G8, g8 = task - pix * G8;
            const int py = pix / RX, px = pix - py * RX;
            const int gx = ox0 - 2 + px, gy = oy0 - 2 + py;
            const bool ok = task < NTASK && gx >= 0 && gx < W && gy >= 0 && gy < H;
            const float* __restrict__ src = p.h + (size_t)(8 * g8) * plane + (ok ? (size_t)gy * W + gx : 0);
            float v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const float t = src[(size_t)k * plane];
                v[k] = ok ? t : 0.0f;
            }
            if (task < NTASK)
                *reinterpret_cast<u4*>(HA + (py * PITCH + px + 1) * XC + g8 * 16) =
                    (u4){pack_bf16_g(v[0], v[1]), pack_bf16_g(v[2], v[3]), pack_bf16_g(v[4], v[5]), pack_bf16_g(v[6], v[7])};
        }
    }
    __syncthreads();

    // ---- P1: x = relu(conv(cost)) on the whole region ----------------------------------------------------------------------
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
        for (int kb = 0; kb < NKB1; ++kb) {
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
        }
        // D row (pixel) = 4 (lane >> 4) + register, column (channel) = lane & 15
#pragma unroll
        for (int t = 0; t < NT1; ++t) {
            const int id = wave + GW * t;
            if (id < RY * MG && m < HID) {
                const int r = id / MG, g = id - r * MG;
                const int gy = oy0 - 2 + r;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int cc = 16 * g + 4 * kgroup + k;
                    const int gx = ox0 - 2 + cc;
                    const bool in = gx >= 0 && gx < W && gy >= 0 && gy < H;
                    const float y = fmaxf(acc[t][k] * 1.0f + 0.0f, 0.0f);
                    *reinterpret_cast<unsigned short*>(XA + (r * PITCH + cc + 1) * XC + m * 2) = in ? bf16_bits(y) : (unsigned short)0;
                }
            }
        }
    }
    __syncthreads();

    // ---- P2: gates.  Core tasks (rows 2 .. TY + 1: swept again by the SAME wave in P3) and halo tasks (rows 1 and TY + 2) ------
    constexpr int NCT = TY * MG / GW;
    f4 ukeep[NCT], hkeep[NCT];
    // K group kk = 4 kb + kgroup (8 channels each): tap kk / GPT, part kk % GPT -- the first half of a tap's channels is x, the second h | r*h
    constexpr int GPT = 2 * HID / 8;
    auto a_off = [&](int kb, int& second) {
        const int kk = 4 * kb + kgroup;
        const int t9 = kk / GPT, part = kk - t9 * GPT;
        const int ky = t9 < 9 ? t9 / 3 : 0, kx = t9 < 9 ? t9 % 3 : 0;
        second = part >= GPT / 2 ? 1 : 0;
        return ((ky - 1) * PITCH + kx) * XC + (part % (GPT / 2)) * 16;
    };
    const int hch = m & (HID - 1);
    auto load_h4 = [&](int r, int g) {   // h[hch] at the lane's four pixels (zeros outside the image)
        f4 v;
        const int gy = oy0 - 2 + r;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int gx = ox0 - 2 + 16 * g + 4 * kgroup + k;
            const bool in = gx >= 0 && gx < W && gy >= 0 && gy < H;
            const float t = p.h[(size_t)hch * plane + (in ? (size_t)gy * W + gx : 0)];
            v[k] = in ? t : 0.0f;
        }
        return v;
    };
    {
        constexpr int NT2 = NCT + 1;   // the last one is the halo task (wave < 2 MG)
        f4 acc[NT2][NTNG];
        int base[NT2], rr[NT2], gg[NT2];
#pragma unroll
        for (int t = 0; t < NT2; ++t) {
#pragma unroll
            for (int nt = 0; nt < NTNG; ++nt) acc[t][nt] = (f4){0, 0, 0, 0};
            if (t < NCT) {
                const int id = wave + GW * t;
                rr[t] = 2 + id / MG; gg[t] = id % MG;
            } else {
                const int id = min(wave, 2 * MG - 1);
                rr[t] = id < MG ? 1 : RY - 2; gg[t] = id % MG;
            }
            base[t] = (rr[t] * PITCH + 16 * gg[t] + m) * XC;
        }
#pragma unroll
        for (int kb = 0; kb < NKBG; ++kb) {
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
        }
#pragma unroll
        for (int t = 0; t < NT2; ++t) {
            if (t == NCT && wave >= 2 * MG) continue;   // (no halo task for this wave)
            const f4 hh = load_h4(rr[t], gg[t]);
            f4 rgate, ugate;
            if constexpr (HID == 8) {   // one N tile: channels 0-7 reset, 8-15 update
                const float sh = p.bg[m];
                f4 y = acc[t][0] * 1.0f + sh;
#pragma unroll
                for (int k = 0; k < 4; ++k) y[k] = 1.0f / (1.0f + __expf(-y[k]));
                rgate = y;
#pragma unroll
                for (int k = 0; k < 4; ++k) ugate[k] = __shfl_down(y[k], 8, 16);   // u of channel m arrives from lane m + 8
            } else {
                const float shr = p.bg[m], shu = p.bg[16 + m];
                f4 y = acc[t][0] * 1.0f + shr, z = acc[t][NTNG - 1] * 1.0f + shu;
#pragma unroll
                for (int k = 0; k < 4; ++k) { y[k] = 1.0f / (1.0f + __expf(-y[k])); z[k] = 1.0f / (1.0f + __expf(-z[k])); }
                rgate = y; ugate = z;
            }
            if (m < HID) {
                const f4 rh = rgate * hh;
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    *reinterpret_cast<unsigned short*>(RA + (rr[t] * PITCH + 16 * gg[t] + 4 * kgroup + k + 1) * XC + m * 2) = bf16_bits(rh[k]);
            }
            if (t < NCT) { ukeep[t] = ugate; hkeep[t] = hh; }
        }
    }
    __syncthreads();

    // ---- P3: candidate and state update on the core tasks ---------------------------------------------------------------------
    {
        f4 acc[NCT];
        int base[NCT];
#pragma unroll
        for (int t = 0; t < NCT; ++t) {
            acc[t] = (f4){0, 0, 0, 0};
            const int id = wave + GW * t;
            base[t] = ((2 + id / MG) * PITCH + 16 * (id % MG) + m) * XC;
        }
#pragma unroll
        for (int kb = 0; kb < NKBG; ++kb) {
            int second;
            const int aoff = a_off(kb, second);
            const unsigned char* arr = second ? RA : XA;
            const bf16x8 b = __builtin_bit_cast(bf16x8, wcl[kb * 64 + lane]);
#pragma unroll
            for (int t = 0; t < NCT; ++t) {
                const bf16x8 a = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u4*>(arr + base[t] + aoff));
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[t], 0, 0, 0);
            }
        }
        if (m < HID) {
            const float sh = p.bc[m];
#pragma unroll
            for (int t = 0; t < NCT; ++t) {
                const int id = wave + GW * t;
                const int r = 2 + id / MG, g = id % MG;
                const int gy = oy0 - 2 + r;
                f4 y = acc[t] * 1.0f + sh;
                const f4 u = ukeep[t], hh = hkeep[t];
#pragma unroll
                for (int k = 0; k < 4; ++k) y[k] = u[k] * hh[k] + (1.0f - u[k]) * tanhf(y[k]);
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int cc = 16 * g + 4 * kgroup + k;
                    const int gx = ox0 - 2 + cc;
                    if (cc >= 2 && cc < RX - 2 && gx < W && gy < H) p.hout[(size_t)m * plane + (size_t)gy * W + gx] = y[k];
                }
            }
        }
    }
}

template <int CP, int HID, int S, int MG, int TY>
static int launch_gru(const GruParams& p, hipStream_t stream) {
    using G = GruGeom<CP, HID, S, MG, TY>;
    static_assert(G::LDS <= 160 * 1024, "tile does not fit the LDS");
    auto kern = gru_cell_fused_kernel<CP, HID, S, MG, TY>;
    int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(kern), G::LDS);
    if (rc != D3D_OK) return rc;
    const int gx = ceil_div(p.W, G::OX), gy = ceil_div(p.H, TY);
    if (gy > 65535) return D3D_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(kern, dim3(gx, gy), dim3(GNT), G::LDS, stream, p);
    D3D_LAUNCH_CHECK("gru_cell_fused_kernel launch");
    return D3D_OK;
}

}  // namespace

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
    GruParams p = {};
    p.cost = cost; p.h = h; p.hout = hout; p.w1 = reinterpret_cast<const u4*>(w1); p.wg = reinterpret_cast<const u4*>(wg);
    p.wc = reinterpret_cast<const u4*>(wc); p.bg = bg; p.bc = bc; p.H = H; p.W = W; p.HI = HI; p.WI = WI;
    hipStream_t st = (hipStream_t)stream;
    if (stride == 1 && HID == 8) {
        if (CP == 8) return launch_gru<8, 8, 1, 4, 8>(p, st);
        if (CP == 16) return launch_gru<16, 8, 1, 4, 8>(p, st);
        if (CP == 32) return launch_gru<32, 8, 1, 4, 8>(p, st);
    }
    if (stride == 2 && HID == 16 && CP == 8) return launch_gru<8, 16, 2, 2, 8>(p, st);
    set_error("d3d_gru_cell_fused_bf16: C = %d, hidden = %d, stride = %d not taken (8 | 16 | 32 -> 8 at stride 1; 8 -> 16 at stride 2)", CP, HID, stride);
    return D3D_ERR_UNSUPPORTED;
}
