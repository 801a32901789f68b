// 3x3 stride-1 2-D convolution over WIDE channel counts (C_in = 64 | 128 as cat(x, x2), C_out = 32 | 64 | 128) on the bf16 matrix
// cores -- the two coarse conv-GRU levels of the RED-Net slice regulariser (msrednet.py:337-370: conv_gru3 32 + 32 -> 64 / -> 32,
// conv_gru4 64 + 64 -> 128 / -> 64; module.py:53-99 ConvGRUCell2, whose GroupNorm keeps the convolutions plain).
//
// Until round 4 these four layers per slice ran -- in bf16 mode too -- on round 1's generic fp32 matrix-core kernel
// (csrc/conv_mfma.hip, `d3d_conv_gemm_f32`: 31 TFLOP/s on 1.34 TFLOP per view = 43 of a 145 ms RED-Net view), because the tile
// kernel of csrc/conv2d_zs.hip keeps a whole layer's cells and weights in LDS, which ends at 48 input channels.  Here the K
// dimension is walked in CHUNKS of 32 input channels:
//   * a workgroup (8 waves) owns 64 x 8 output pixels x NW 16-channel output tiles (NW <= 4: blockIdx.z takes the next group of
//     output channels, so the small coarse images of the cascade's first stage still fill the chip);
//   * per chunk the 66 x 10 patch of 32 channels is staged as channel-last bf16 cells (RNE, 80-byte cells: odd 16-byte slots)
//     beside the 9 x NW weight fragments of that chunk (the host packing of conv2d_zs.hip, ops._pack_z2_bf16: K = (k_y, k_x,
//     c_in), so chunk j of tap t is K block t * (C_in / 32) + j -- no second packing); the next chunk's patch is requested
//     into registers before the sweep of the current one and committed behind it;
//   * the sweep is the implicit GEMM of the tile kernels: M = 16 consecutive pixels of a row, N = 16 output channels, K = 32
//     = (tap, 32 channels of the chunk): one ds_read_b128 A operand per (M group, K block) feeds NW MFMAs, one B fragment
//     feeds the wave's four M groups;
//   * epilogue: folded affine / bias, optional ReLU, optional skip (added last), 16-byte stores of 4 pixels of a channel.
// Operands are rounded to bf16 exactly as conv2d_zs.hip rounds them; accumulation is fp32 in K-block order (tap-major within a
// chunk, chunks in channel order).
#include <cstdint>
#include "common.h"
#include "gn_stats.h"

namespace d3d {

namespace {

typedef float f4 __attribute__((ext_vector_type(4)));
typedef unsigned u4 __attribute__((ext_vector_type(4)));

constexpr int WW = 8;                 // waves = output rows of a tile
constexpr int WNT = 64 * WW;
constexpr int WTX = 64, WPX = WTX + 2, WPY = WW + 2;
constexpr int WCS = 80;               // bytes per 32-channel cell
constexpr int WPATCH = WPX * WPY * WCS;

struct WideParams {
    const float* in;      // [C1, H, W]
    const float* in2;     // [CI - C1, H, W] or null
    int C1, CI;
    const u4* wpk;        // [9 * CI / 32][NTN][64] B fragments (ops._pack_z2_bf16)
    const float* scale;   // [CO] or null
    const float* shift;   // [CO] or null
    const float* skip;    // [CO, H, W] or null (added after the activation)
    float* out;           // [CO, H, W]
    int H, W, CO, NTN;    // NTN = 16-channel output tiles of the layer
    int act;              // 0 none | 1 ReLU
    double* gn;           // GN form: GroupNorm statistics of the output (gn_stats.h), zeroed by the caller; channels >= gn_split are group 1
    int gn_split;
    int stuffed;          // 1: `in` is [CI, H/2, W/2] and stands for its zero-stuffed image (in at the even rows / columns, zeros elsewhere):
                          // the stride-2 transposed convolution as a stride-1 one, without the stuffed tensor (d3d_convtranspose2d_k3s2_wide_h16)
};

__device__ __forceinline__ unsigned pack_h16_w(float a, float b) {
    return pack_h16x2(a, b);   // one packed conversion (common.h: pack_h16x2)
}

template <int NW, bool GN = false>
__global__ __launch_bounds__(WNT, 2) void conv2d_wide_bf16_kernel(WideParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* patch = smem;
    u4* wlds = reinterpret_cast<u4*>(smem + WPATCH);   // [9][NW][64]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int H = p.H, W = p.W;
    const size_t plane = (size_t)H * W;
    const size_t iplane = p.stuffed ? (size_t)(H >> 1) * (W >> 1) : plane;   // a channel plane of the input
    const int x0 = blockIdx.x * WTX, y0 = blockIdx.y * WW;
    const int nt0 = blockIdx.z * NW;            // first output tile of this workgroup
    const int nchunk = p.CI / 32, nkb_tap = nchunk;   // K blocks per tap in the packed weights

    // ---- staging of one 32-channel chunk: a task = (pixel of the 66 x 10 patch, 8 channels) -> eight dword loads, one 16-byte chunk
    constexpr int NTASK = WPX * WPY * 4, ROUNDS = (NTASK + WNT - 1) / WNT;
    float stg[ROUNDS][8];
    // per-lane state of a task, the same for every chunk: element offset of its pixel inside a channel plane (clamped into the
    // image), whether it is inside, its cell -- and the values stay raw until they are committed: a select behind the loads would
    // wait for them before the sweep they are meant to fly under
    size_t poff[ROUNDS];
    int pdst[ROUNDS], pg8[ROUNDS];
    bool pok[ROUNDS];
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {
        const int task = tid + r * WNT;
        const int pix = task >> 2, g = task & 3;
        const int py = pix / WPX, px = pix - py * WPX;
        const int gx = x0 + px - 1, gy = y0 + py - 1;
        pok[r] = task < NTASK && gx >= 0 && gx < W && gy >= 0 && gy < H;
        poff[r] = task < NTASK ? (size_t)min(max(gy, 0), H - 1) * W + min(max(gx, 0), W - 1) : 0;
        if (p.stuffed) {   // odd rows / columns of the stuffed image are zeros; an even one is input pixel (gy / 2, gx / 2)
            pok[r] = pok[r] && !((gx | gy) & 1);
            poff[r] = task < NTASK ? (size_t)(min(max(gy, 0), H - 1) >> 1) * (W >> 1) + (min(max(gx, 0), W - 1) >> 1) : 0;
        }
        pdst[r] = task < NTASK ? pix * WCS + g * 16 : -1;
        pg8[r] = 8 * g;
    }
    auto issue = [&](int j) {
        const int cbase = 32 * j;
#pragma unroll
        for (int r = 0; r < ROUNDS; ++r) {
            const int c = cbase + pg8[r];
            const float* __restrict__ src = (c < p.C1 ? p.in + (size_t)c * iplane : p.in2 + (size_t)(c - p.C1) * iplane) + poff[r];
#pragma unroll
            for (int k = 0; k < 8; ++k) stg[r][k] = src[(size_t)k * iplane];
        }
    };
    auto commit = [&]() {
#pragma unroll
        for (int r = 0; r < ROUNDS; ++r) {
            if (pdst[r] >= 0) {
                float x[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) x[k] = pok[r] ? stg[r][k] : 0.0f;
                *reinterpret_cast<u4*>(patch + pdst[r]) =
                    (u4){pack_h16_w(x[0], x[1]), pack_h16_w(x[2], x[3]), pack_h16_w(x[4], x[5]), pack_h16_w(x[6], x[7])};
            }
        }
    };
    auto load_weights = [&](int j) {   // the chunk's 9 x NW fragments: K block t * nkb_tap + j, output tiles nt0 ..
        for (int i = tid; i < 9 * NW * 64; i += WNT) {
            const int l = i & 63, rest = i >> 6, nt = rest % NW, t = rest / NW;
            wlds[i] = p.wpk[((size_t)(t * nkb_tap + j) * p.NTN + nt0 + nt) * 64 + l];
        }
    };

    f4 acc[4][NW];
#pragma unroll
    for (int mg = 0; mg < 4; ++mg)
#pragma unroll
        for (int nt = 0; nt < NW; ++nt) acc[mg][nt] = (f4){0, 0, 0, 0};
    const int abase = (wave * WPX + (lane & 15)) * WCS + (lane >> 4) * 16;   // K group = 8 channels of the chunk

    issue(0);
    for (int j = 0; j < nchunk; ++j) {
        commit();
        load_weights(j);
        __syncthreads();
        if (j + 1 < nchunk) issue(j + 1);   // in flight during the sweep
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int aoff = ((t / 3) * WPX + (t % 3)) * WCS;
            h16x8 b[NW];
#pragma unroll
            for (int nt = 0; nt < NW; ++nt) b[nt] = __builtin_bit_cast(h16x8, wlds[(t * NW + nt) * 64 + lane]);
#pragma unroll
            for (int mg = 0; mg < 4; ++mg) {
                const h16x8 a = __builtin_bit_cast(h16x8, *reinterpret_cast<const u4*>(patch + abase + mg * 16 * WCS + aoff));
#pragma unroll
                for (int nt = 0; nt < NW; ++nt) acc[mg][nt] = mfma_h16(a, b[nt], acc[mg][nt]);
            }
        }
        __syncthreads();   // every wave has read the patch and the weights of this chunk
    }

    // ---- epilogue: D row (pixel) = 4 (lane >> 4) + register, column (channel) = lane & 15 ------------------------------------
    const int oy = y0 + wave;
    GnAcc gacc;
    gn_zero(gacc);
    if (oy < H) {
#pragma unroll
        for (int nt = 0; nt < NW; ++nt) {
            const int co = (nt0 + nt) * 16 + (lane & 15);
            if (co >= p.CO) continue;
            const float sc = p.scale ? p.scale[co] : 1.0f, sh = p.shift ? p.shift[co] : 0.0f;
#pragma unroll
            for (int mg = 0; mg < 4; ++mg) {
                const int ox = x0 + mg * 16 + (lane >> 4) * 4;
                if (ox >= W) continue;
                const size_t o = (size_t)co * plane + (size_t)oy * W + ox;
                f4 y = acc[mg][nt] * sc + sh;
                if (p.act == 1) y = __builtin_elementwise_max(y, (f4){0, 0, 0, 0});
                if ((W & 3) == 0) {                                 // a quad is inside or outside as a whole, rows are 16-byte aligned
                    if (p.skip) y = *reinterpret_cast<const f4*>(p.skip + o) + y;
                    *reinterpret_cast<f4*>(p.out + o) = y;
                    if constexpr (GN) gn_add(gacc, co >= p.gn_split, y);
                } else {                                            // (the coarsest level of the first cascade stage: 86 x 58)
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                        if (ox + k < W) p.out[o + k] = p.skip ? p.skip[o + k] + y[k] : y[k];
                    if constexpr (GN) {   // (act 0, no skip: the stored values are y's)
#pragma unroll
                        for (int k = 0; k < 4; ++k)
                            if (ox + k >= W) y[k] = 0.0f;   // a zero adds nothing to either sum
                        gn_add(gacc, co >= p.gn_split, y);
                    }
                }
            }
        }
    }
    if constexpr (GN) gn_flush(gacc, p.gn, p.gn_split < p.CO ? 2 : 1, reinterpret_cast<double*>(smem), tid, WNT / 64);
}

template <int NW, bool GN = false>
static int launch_wide(const WideParams& p, hipStream_t stream) {
    constexpr int lds = WPATCH + 9 * NW * 64 * 16;
    static_assert(lds <= 160 * 1024, "tile does not fit the LDS");
    auto kern = conv2d_wide_bf16_kernel<NW, GN>;
    int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(kern), lds);
    if (rc != D3D_OK) return rc;
    const int gy = ceil_div(p.H, WW);
    if (gy > 65535) return D3D_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(kern, dim3(ceil_div(p.W, WTX), gy, p.NTN / NW), dim3(WNT), lds, stream, p);
    D3D_LAUNCH_CHECK("conv2d_wide_bf16_kernel launch");
    return D3D_OK;
}

}  // namespace

}  // namespace d3d

using namespace d3d;

// out [Co,H,W] = act(conv3x3(cat(in, in2)) * scale + shift) (+ skip, added last); bf16 matrix-core operands, fp32 accumulation.
// C1, C2 multiples of 32 with C1 + C2 = 64 | 128; Co = 32 | 64 | 128; wpacked = ops._pack_z2_bf16(weight).
static int conv2d_k3_wide_bf16(const float* in, int C1, const float* in2, int C2, const void* wpacked, const float* scale,
                              const float* shift, const float* skip, int act, int Co, int H, int W, float* out, double* gn_stats,
                              int gn_split, d3d_stream_t stream) {
    D3D_REQUIRE(in && wpacked && out, "null pointer");
    D3D_REQUIRE(H > 0 && W > 0 && C1 > 0 && C2 >= 0 && Co > 0, "bad dims");
    D3D_REQUIRE(act == 0 || act == 1, "bad act %d", act);
    D3D_REQUIRE(C2 == 0 || in2, "second input missing");
    const int Ci = C1 + C2;
    if ((Ci != 64 && Ci != 128) || C1 % 32 || C2 % 32 || (Co != 32 && Co != 64 && Co != 128) ||
        ((reinterpret_cast<uintptr_t>(out) | reinterpret_cast<uintptr_t>(skip)) & 15)) {
        set_error("d3d_conv2d_k3_wide_h16: C_in = %d + %d (64 | 128 in parts of 32), C_out = %d (32 | 64 | 128) with 16-byte aligned tensors not taken",
                  C1, C2, Co);
        return D3D_ERR_UNSUPPORTED;
    }
    WideParams p = {};
    p.in = in; p.in2 = in2; p.C1 = C1; p.CI = Ci; p.wpk = reinterpret_cast<const u4*>(wpacked); p.scale = scale; p.shift = shift;
    p.skip = skip; p.out = out; p.H = H; p.W = W; p.CO = Co; p.NTN = Co / 16; p.act = act;
    p.gn = gn_stats; p.gn_split = gn_split;
    hipStream_t st = (hipStream_t)stream;
    // four output tiles per workgroup; two where that leaves fewer than ~2 workgroups per CU (the coarse levels of the first stages)
    const long tiles = (long)ceil_div(W, WTX) * ceil_div(H, WW);
    const bool four = p.NTN % 4 == 0 && tiles * (p.NTN / 4) >= 512;
    if (gn_stats) return four ? launch_wide<4, true>(p, st) : launch_wide<2, true>(p, st);
    return four ? launch_wide<4>(p, st) : launch_wide<2>(p, st);
}

// out [Co,H,W] = act(conv3x3(cat(in, in2)) * scale + shift) (+ skip, added last); bf16 matrix-core operands, fp32 accumulation.
// C1, C2 multiples of 32 with C1 + C2 = 64 | 128; Co = 32 | 64 | 128; wpacked = ops._pack_z2_bf16(weight).
extern "C" int d3d_conv2d_k3_wide_h16(const float* in, int C1, const float* in2, int C2, const void* wpacked, const float* scale,
                                       const float* shift, const float* skip, int act, int Co, int H, int W, float* out,
                                       d3d_stream_t stream) {
    return conv2d_k3_wide_bf16(in, C1, in2, C2, wpacked, scale, shift, skip, act, Co, H, W, out, nullptr, 0, stream);
}
// The same layer (act 0, no skip) + the GroupNorm(1, C) statistics of its output (see d3d_conv2d_k3_zs_h16_gn).
extern "C" int d3d_conv2d_k3_wide_h16_gn(const float* in, int C1, const float* in2, int C2, const void* wpacked, const float* shift,
                                          int Co, int H, int W, float* out, double* gn_stats, int gn_split, d3d_stream_t stream) {
    D3D_REQUIRE(gn_stats && gn_split > 0 && gn_split <= Co, "bad statistics arguments");
    return conv2d_k3_wide_bf16(in, C1, in2, C2, wpacked, nullptr, shift, nullptr, 0, Co, H, W, out, gn_stats, gn_split, stream);
}

// module.py:287-294 with 64 input channels (msrednet.py:348 upconv3): out [Co,2H,2W] = act(convT3x3_s2(in) * scale + shift) (+ skip, added
// last) as the stride-1 convolution of the zero-stuffed input with the flipped, transposed kernel (wpacked = ops._pack_z2_bf16 of it)
// -- the stuffed image exists only in the kernel's staging.  in [64,H,W]; Co = 32 | 64.
extern "C" int d3d_convtranspose2d_k3s2_wide_h16(const float* in, const void* wpacked, const float* scale, const float* shift,
                                                  const float* skip, int act, int Ci, int Co, int H, int W, float* out, d3d_stream_t stream) {
    D3D_REQUIRE(in && wpacked && out, "null pointer");
    D3D_REQUIRE(H > 0 && W > 0 && Co > 0, "bad dims");
    D3D_REQUIRE(act == 0 || act == 1, "bad act %d", act);
    if (Ci != 64 || (Co != 32 && Co != 64) || ((reinterpret_cast<uintptr_t>(out) | reinterpret_cast<uintptr_t>(skip)) & 15)) {
        set_error("d3d_convtranspose2d_k3s2_wide_h16: C_in = %d (64), C_out = %d (32 | 64) with 16-byte aligned tensors not taken", Ci, Co);
        return D3D_ERR_UNSUPPORTED;
    }
    WideParams p = {};
    p.in = in; p.in2 = nullptr; p.C1 = Ci; p.CI = Ci; p.wpk = reinterpret_cast<const u4*>(wpacked); p.scale = scale; p.shift = shift;
    p.skip = skip; p.out = out; p.H = 2 * H; p.W = 2 * W; p.CO = Co; p.NTN = Co / 16; p.act = act; p.stuffed = 1;
    const long tiles = (long)ceil_div(p.W, WTX) * ceil_div(p.H, WW);
    const bool four = p.NTN % 4 == 0 && tiles * (p.NTN / 4) >= 512;
    return four ? launch_wide<4>(p, (hipStream_t)stream) : launch_wide<2>(p, (hipStream_t)stream);
}
