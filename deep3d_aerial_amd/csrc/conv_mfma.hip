// Implicit-GEMM 3x3(x3) convolution family on the gfx950 matrix cores, exact fp32.
//
// The regularisers of the path (CostRegNet 3D UNet, cas_mvsnet.py:81-121; pair UNet and slice
// conv-GRU, adamvs.py:198-238,403-427) are small-channel k=3 convolutions: C_out 1..64,
// C_in 8..64.  As a GEMM:  out[co, pix] = SUM_k  W[co, k] * X[k, pix],  k = (tap, ci).
// v_mfma_f32_16x16x4_f32 (f32 in, f32 accumulate; bit-for-bit a k-ordered fmaf chain, so parity
// with the fp32 reference is kept) computes a 16(co) x 16(pix) tile per instruction, K = 4:
//     A[i = lane&15][k = lane>>4] = W[co0 + i][k0 + k]      one VGPR per lane
//     B[k = lane>>4][j = lane&15] = X[k0 + k][pix0 + j]     one VGPR per lane
//     D[row = 4*(lane>>4) + r][col = lane&15]               four VGPRs per lane
//
// One generic kernel, driven by a TAP TABLE (input offset + packed-weight slot per tap), covers
//   * Conv2d / Conv3d, k = 3, pad 1, stride 1 | 2 (taps: offsets -1..1 around out*stride),
//   * ConvTranspose2d / 3d, k = 3, stride 2, pad 1, output_pad 1, as one launch per output-parity
//     class (even outputs use kernel index 1 at input o/2; odd outputs use index 0 at (o+1)/2 and
//     index 2 at (o-1)/2): each class is an ordinary convolution with 1..8 taps at offsets 0/+1,
//   * the channel concatenation of two inputs (GRU: cat(x, h), module.py:30,41).
//
// Workgroup = 4 waves = 4 output rows x 64 output columns of one output slice; wave w owns row w:
// 4 pixel groups (N tiles) x MT channel groups (M tiles) of accumulators.  Input channels are
// processed in chunks of CK: the chunk's input patch (with halo, zero-filled outside the image)
// and the chunk's packed weights are staged in LDS, then the taps are swept with ds_read_b32
// operand fetches (channel stride = 16 mod 32 floats: the four k-groups of a wave hit disjoint
// bank halves).  Epilogue: per-channel affine (folded BatchNorm or bias), ReLU, skip add.
#include "common.h"

namespace d3d {

namespace {

constexpr int CV_ROWS = 4;    // output rows per workgroup (one per wave)
constexpr int CV_COLS = 64;   // output columns per wave (4 N tiles)
constexpr int MAX_TAPS = 27;

struct ConvMParams {
    const float* in0;
    const float* in1;
    const float* wpack;  // [ntaps*Ci][Mpad] packed weights, Mpad = 16*MT (zero padded)
    const float* scale;
    const float* shift;
    const float* skip;
    float* out;
    int Ci0, Ci1, Co;
    int D, H, W;        // input dims
    int Dg, Hg, Wg;     // grid of outputs iterated by this launch
    int Do, Ho, Wo;     // full output dims (addressing)
    int istride;        // input index = g*istride + tap offset
    int ostride, oz, oy, ox;  // output index = g*ostride + o{z,y,x}
    int act, skip_after_act;
    int ntaps;
    int zmin, zspan, ymin, yspan, xmin, xspan;  // tap offset ranges: patch = span + (tile-1)*istride
    signed char tz[MAX_TAPS], ty[MAX_TAPS], tx[MAX_TAPS];
};

typedef float f4v __attribute__((ext_vector_type(4)));

template <int MT, int CK>
__global__ __launch_bounds__(256) void conv_mfma_kernel(ConvMParams p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, j = lane & 15;
    constexpr int MP = 16 * MT;

    // patch geometry (per channel): PZ x PY x PX floats, channel stride CS = 16 (mod 32)
    const int PZ = p.zspan, PY = p.yspan + (CV_ROWS - 1) * p.istride, PX = p.xspan + (CV_COLS - 1) * p.istride;
    int CS = PZ * PY * PX;
    CS += (16 - (CS & 31) + 32) & 31;
    float* xin = lds;                 // [CK][CS]
    float* wl = lds + CK * CS;        // [ntaps*CK][MP]

    const int gx0 = blockIdx.x * CV_COLS;
    const int gy0 = blockIdx.y * CV_ROWS;
    const int gz = blockIdx.z;
    const int Ci = p.Ci0 + p.Ci1;
    const long in_plane = (long)p.H * p.W, in_vol = in_plane * p.D;

    f4v acc[MT][4];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < 4; ++n) acc[m][n] = (f4v){0, 0, 0, 0};

    // input coordinates of patch origin
    const int iz0 = gz * p.istride + p.zmin, iy0 = gy0 * p.istride + p.ymin, ix0 = gx0 * p.istride + p.xmin;
    // this lane's B base: channel g, row `wave`, column j (per N tile: + 16*istride)
    const int bbase = g * CS + (wave * p.istride) * PX + j * p.istride;

    for (int c0 = 0; c0 < Ci; c0 += CK) {
        __syncthreads();  // previous chunk fully consumed
        // ---- stage the input patch of channels c0..c0+CK-1 (zeros outside the image / beyond Ci):
        // a wave takes patch rows (channel, z, y) round-robin, lanes run along x (coalesced)
        const int nrows = CK * PZ * PY;
        for (int rr = wave; rr < nrows; rr += 4) {
            const int c = rr / (PZ * PY), zy = rr - c * (PZ * PY);
            const int z = zy / PY, y = zy - z * PY;
            const int ci = c0 + c;
            const int sy = iy0 + y, sz = iz0 + z;
            const bool rowok = (ci < Ci) && (unsigned)sy < (unsigned)p.H && (unsigned)sz < (unsigned)p.D;
            // rows beyond Ci are read from a valid dummy location (channel 0 of in0) and zeroed
            const float* __restrict__ src = (ci < p.Ci0 || ci >= Ci) ? p.in0 + (long)(ci < p.Ci0 ? ci : 0) * in_vol
                                                                       : p.in1 + (long)(ci - p.Ci0) * in_vol;
            const float* __restrict__ row = src + (long)(rowok ? sz : 0) * in_plane + (long)(rowok ? sy : 0) * p.W;
            float* dst = xin + c * CS + (z * PY + y) * PX;
            for (int x = lane; x < PX; x += 64) {
                const int sx = ix0 + x;
                const bool ok = rowok && (unsigned)sx < (unsigned)p.W;
                const float v = row[ok ? sx : 0];
                dst[x] = ok ? v : 0.0f;
            }
        }
        // ---- stage the packed weights of this chunk: rows k = t*Ci + ci, ci in [c0, c0+CK)
        const int nvalid = min(CK, Ci - c0) * MP;
        for (int t = 0; t < p.ntaps; ++t) {
            const float* __restrict__ wsrc = p.wpack + ((long)t * Ci + c0) * MP;
            for (int e = tid; e < CK * MP; e += 256) wl[t * CK * MP + e] = (e < nvalid) ? wsrc[e] : 0.0f;
        }
        __syncthreads();

        // ---- sweep the taps: K steps of 4 channels
        for (int t = 0; t < p.ntaps; ++t) {
            const int toff = ((p.tz[t] - p.zmin) * PY + (p.ty[t] - p.ymin)) * PX + (p.tx[t] - p.xmin);
            const float* __restrict__ xb = xin + bbase + toff;
#pragma unroll
            for (int kk = 0; kk < CK / 4; ++kk) {
                float b[4];
#pragma unroll
                for (int n = 0; n < 4; ++n) b[n] = xb[kk * 4 * CS + n * 16 * p.istride];
#pragma unroll
                for (int m = 0; m < MT; ++m) {
                    // A[i = lane&15][k = lane>>4] = wl[(t*CK + kk*4 + g) * MP + m*16 + (lane&15)]
                    const float av = wl[(t * CK + kk * 4 + g) * MP + m * 16 + j];
#pragma unroll
                    for (int n = 0; n < 4; ++n)
                        acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b[n], acc[m][n], 0, 0, 0);
                }
            }
        }
    }

    // ---- epilogue: D[row = 4*g + r][col = j] -> channel m*16 + 4*g + r, pixel column n*16 + j
    const int gy = gy0 + wave;
    if (gy >= p.Hg) return;
    const int oy_ = gy * p.ostride + p.oy, oz_ = gz * p.ostride + p.oz;
    const long out_plane = (long)p.Ho * p.Wo;
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int co = m * 16 + 4 * g + r;
            if (co >= p.Co) continue;
            const float sc = p.scale ? p.scale[co] : 1.0f;
            const float sh = p.shift ? p.shift[co] : 0.0f;
#pragma unroll
            for (int n = 0; n < 4; ++n) {
                const int gx = gx0 + n * 16 + j;
                if (gx >= p.Wg) continue;
                const long oidx = ((long)co * p.Do + oz_) * out_plane + (long)oy_ * p.Wo + (gx * p.ostride + p.ox);
                float y = acc[m][n][r];
                if (p.scale) y *= sc;
                if (p.shift) y += sh;
                if (p.skip && !p.skip_after_act) y += p.skip[oidx];
                if (p.act == 1) y = fmaxf(y, 0.0f);
                if (p.skip && p.skip_after_act) y = p.skip[oidx] + y;
                p.out[oidx] = y;
            }
        }
}

static int lds_floats(const ConvMParams& p, int MT, int CK) {
    const int PZ = p.zspan, PY = p.yspan + (CV_ROWS - 1) * p.istride, PX = p.xspan + (CV_COLS - 1) * p.istride;
    int CS = PZ * PY * PX;
    CS += (16 - (CS & 31) + 32) & 31;
    return CK * CS + p.ntaps * CK * 16 * MT;
}

template <int MT, int CK>
static int launch_cfg(const ConvMParams& p, hipStream_t stream) {
    const int bytes = lds_floats(p, MT, CK) * 4;
    if (bytes > 160 * 1024) return D3D_ERR_UNSUPPORTED;
    auto kern = conv_mfma_kernel<MT, CK>;
    static int attr_bytes = 0;
    if (bytes > attr_bytes) {
        int rc = hip_status(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024),
                            "hipFuncSetAttribute(MaxDynamicSharedMemorySize)");
        if (rc != D3D_OK) return rc;
        attr_bytes = 160 * 1024;
    }
    dim3 grid(ceil_div(p.Wg, CV_COLS), ceil_div(p.Hg, CV_ROWS), p.Dg);
    if (grid.y > 65535u || grid.z > 65535u) return D3D_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(kern, grid, dim3(256), bytes, stream, p);
    D3D_LAUNCH_CHECK("conv_mfma_kernel launch");
    return D3D_OK;
}

template <int MT>
static int launch_mt(const ConvMParams& p, hipStream_t stream) {
    // largest channel chunk (no larger than the channel count needs) whose patch + weights leave room for
    // two workgroups per CU: one stages while the other feeds the matrix cores
    const int Ci = p.Ci0 + p.Ci1;
    if (Ci > 8 && lds_floats(p, MT, 16) * 4 <= 80 * 1024) return launch_cfg<MT, 16>(p, stream);
    if (Ci > 4 && lds_floats(p, MT, 8) * 4 <= 80 * 1024) return launch_cfg<MT, 8>(p, stream);
    if (Ci > 8 && lds_floats(p, MT, 16) * 4 <= 150 * 1024 && lds_floats(p, MT, 4) * 4 > 80 * 1024) return launch_cfg<MT, 16>(p, stream);
    if (Ci > 4 && lds_floats(p, MT, 8) * 4 <= 150 * 1024 && lds_floats(p, MT, 4) * 4 > 80 * 1024) return launch_cfg<MT, 8>(p, stream);
    return launch_cfg<MT, 4>(p, stream);
}

static int launch_conv_mfma(const ConvMParams& p, hipStream_t stream) {
    const int MT = (p.Co + 15) / 16;
    switch (MT) {
        case 1: return launch_mt<1>(p, stream);
        case 2: return launch_mt<2>(p, stream);
        case 3: return launch_mt<4>(p, stream);
        case 4: return launch_mt<4>(p, stream);
    }
    set_error("conv_mfma: C_out=%d unsupported (max 64)", p.Co);
    return D3D_ERR_UNSUPPORTED;
}

}  // namespace

}  // namespace d3d

using namespace d3d;

extern "C" {

int d3d_conv_gemm_f32(const float* in0, int Ci0, const float* in1, int Ci1, const float* wpack, int mpad,
                      const float* scale, const float* shift, const float* skip, int skip_after_act, int act,
                      int Co, int D, int H, int W, int Dg, int Hg, int Wg, int Do, int Ho, int Wo, int istride,
                      int ostride, int oz, int oy, int ox, int ntaps, const signed char* taps_zyx, float* out,
                      d3d_stream_t stream) {
    D3D_REQUIRE(in0 && wpack && out && taps_zyx, "null pointer");
    D3D_REQUIRE(Ci0 > 0 && Ci1 >= 0 && (Ci1 == 0 || in1), "bad input channel split %d+%d", Ci0, Ci1);
    D3D_REQUIRE(Co > 0 && Co <= 64 && mpad == 16 * ((Co + 15) / 16 == 3 ? 4 : (Co + 15) / 16), "bad Co=%d / mpad=%d", Co, mpad);
    D3D_REQUIRE(D > 0 && H > 0 && W > 0 && Dg > 0 && Hg > 0 && Wg > 0, "bad dims");
    D3D_REQUIRE(ntaps > 0 && ntaps <= MAX_TAPS, "bad ntaps %d", ntaps);
    D3D_REQUIRE((istride == 1 || istride == 2) && (ostride == 1 || ostride == 2), "bad strides");
    D3D_REQUIRE(act == 0 || act == 1, "bad act %d", act);
    D3D_REQUIRE(oz >= 0 && oy >= 0 && ox >= 0 && (Dg - 1) * ostride + oz < Do && (Hg - 1) * ostride + oy < Ho &&
                    (Wg - 1) * ostride + ox < Wo,
                "output grid %dx%dx%d (stride %d, offset %d,%d,%d) exceeds output %dx%dx%d", Dg, Hg, Wg, ostride, oz,
                oy, ox, Do, Ho, Wo);
    ConvMParams p = {};
    p.in0 = in0; p.in1 = in1; p.wpack = wpack; p.scale = scale; p.shift = shift; p.skip = skip; p.out = out;
    p.Ci0 = Ci0; p.Ci1 = Ci1; p.Co = Co; p.D = D; p.H = H; p.W = W;
    p.Dg = Dg; p.Hg = Hg; p.Wg = Wg; p.Do = Do; p.Ho = Ho; p.Wo = Wo;
    p.istride = istride; p.ostride = ostride; p.oz = oz; p.oy = oy; p.ox = ox;
    p.act = act; p.skip_after_act = skip_after_act ? 1 : 0; p.ntaps = ntaps;
    int zlo = 127, zhi = -128, ylo = 127, yhi = -128, xlo = 127, xhi = -128;
    for (int t = 0; t < ntaps; ++t) {
        p.tz[t] = taps_zyx[3 * t]; p.ty[t] = taps_zyx[3 * t + 1]; p.tx[t] = taps_zyx[3 * t + 2];
        zlo = p.tz[t] < zlo ? p.tz[t] : zlo; zhi = p.tz[t] > zhi ? p.tz[t] : zhi;
        ylo = p.ty[t] < ylo ? p.ty[t] : ylo; yhi = p.ty[t] > yhi ? p.ty[t] : yhi;
        xlo = p.tx[t] < xlo ? p.tx[t] : xlo; xhi = p.tx[t] > xhi ? p.tx[t] : xhi;
    }
    p.zmin = zlo; p.zspan = zhi - zlo + 1; p.ymin = ylo; p.yspan = yhi - ylo + 1; p.xmin = xlo; p.xspan = xhi - xlo + 1;
    return launch_conv_mfma(p, (hipStream_t)stream);
}

}  // extern "C"
