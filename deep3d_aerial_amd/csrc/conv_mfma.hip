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
    int vec, sh, PX, CS;       // 16-byte staging (W % 4 == 0, aligned tensors), origin shift, LDS row length, channel stride
    int mg_row, mg_c, mg_y;    // reciprocal multipliers of the staging index arithmetic
    signed char tz[MAX_TAPS], ty[MAX_TAPS], tx[MAX_TAPS];
};

typedef float f4v __attribute__((ext_vector_type(4)));

typedef unsigned u4v __attribute__((ext_vector_type(4)));

template <int MT, int CK>
__global__ __launch_bounds__(256) void conv_mfma_kernel(ConvMParams p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, j = lane & 15;
    constexpr int MP = 16 * MT;
    constexpr int WS = (MP == 16) ? 16 : MP + 16;  // weight row stride: k-groups g, g+1 on disjoint bank halves

    // patch geometry (per channel): PZ x PY x PX floats, channel stride CS = 16 (mod 32).  vec: rows are staged
    // with 16-byte loads from an origin moved left by sh columns to a multiple of 4 (PX includes sh, rounded to 4)
    const int PZ = p.zspan, PY = p.yspan + (CV_ROWS - 1) * p.istride, PX = p.PX, CS = p.CS;
    float* xin = lds;                                      // [CK][CS]
    float* wl = lds + CK * CS;                             // [ntaps*CK][WS]
    int* tofft = reinterpret_cast<int*>(wl + p.ntaps * CK * WS);  // [ntaps]

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
    const int iz0 = gz * p.istride + p.zmin, iy0 = gy0 * p.istride + p.ymin;
    const int ix0 = gx0 * p.istride + p.xmin - p.sh;
    // this lane's B base: channel g, row `wave`, column j (per N tile: + 16*istride)
    const int bbase = g * CS + (wave * p.istride) * PX + j * p.istride;
    if (tid < p.ntaps)
        tofft[tid] = ((p.tz[tid] - p.zmin) * PY + (p.ty[tid] - p.ymin)) * PX + (p.tx[tid] - p.xmin) + p.sh;

    // staging items: one per (patch row, x segment); vec: a 16-byte load of 4 columns, else one float
    const int rows_per_c = PZ * PY;
    const int nrows = CK * rows_per_c;
    const int per_row = p.vec ? (PX >> 2) : PX;
    const int nitems = nrows * per_row;

    for (int c0 = 0; c0 < Ci; c0 += CK) {
        __syncthreads();  // previous chunk fully consumed
        // ---- stage the input patch of channels c0..c0+CK-1: raw buffer loads, zeros (range check) outside the
        // image / beyond Ci; eight loads in flight per thread.  A chunk never straddles the two inputs.
        {
            const float* cb = (c0 < p.Ci0) ? p.in0 + (long)c0 * in_vol : p.in1 + (long)(c0 - p.Ci0) * in_vol;
            const long span = (long)(CK - 1) * in_vol + in_vol;
            const auto rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(cb), 0, (int)(span * 4), 0x00020000);
            const int cmax = Ci - c0;
            for (int e0 = tid; e0 < nitems; e0 += 256 * 8) {
                unsigned boff[8];
                int dst[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int e = min(e0 + u * 256, nitems - 1);  // (tail slots repeat the last item)
                    const int row = (int)(((long)e * p.mg_row) >> 24), q = e - row * per_row;
                    const int c = (row * p.mg_c) >> 16, zy = row - c * rows_per_c;
                    const int z = (zy * p.mg_y) >> 16, y = zy - z * PY;
                    const int x = p.vec ? 4 * q : q;
                    const int sx = ix0 + x, sy = iy0 + y, sz = iz0 + z;
                    const bool ok = (c < cmax) & ((unsigned)sx < (unsigned)p.W) & ((unsigned)sy < (unsigned)p.H) &
                                    ((unsigned)sz < (unsigned)p.D);
                    boff[u] = ok ? (unsigned)(((long)c * in_vol + (long)sz * in_plane + (long)sy * p.W + sx) << 2)
                                 : 0x80000000u;
                    dst[u] = c * CS + zy * PX + x;
                }
                if (p.vec) {
                    f4v v[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const u4v qv = __builtin_amdgcn_raw_buffer_load_b128(rs, boff[u], 0, 0);
                        v[u] = __builtin_bit_cast(f4v, qv);
                    }
#pragma unroll
                    for (int u = 0; u < 8; ++u) *reinterpret_cast<f4v*>(xin + dst[u]) = v[u];
                } else {
                    float v[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u)
                        v[u] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, boff[u], 0, 0));
#pragma unroll
                    for (int u = 0; u < 8; ++u) xin[dst[u]] = v[u];
                }
            }
        }
        // ---- stage the packed weights of this chunk: rows k = t*Ci + ci, ci in [c0, c0+CK); 16-byte loads,
        // four in flight per thread
        {
            constexpr int R4 = CK * MP / 4;  // float4s per tap
            const int n4 = p.ntaps * R4;
            const int nvalid4 = min(CK, Ci - c0) * (MP / 4);
            for (int e0 = tid; e0 < n4; e0 += 256 * 4) {
                float4 v[4];
                int dst[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int e = min(e0 + u * 256, n4 - 1);
                    const int t = e / R4, r = e - t * R4;
                    const bool ok = r < nvalid4;
                    v[u] = *(reinterpret_cast<const float4*>(p.wpack + ((long)t * Ci + c0) * MP) + (ok ? r : 0));
                    if (!ok) v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
                    const int k = r / (MP / 4), q = r - k * (MP / 4);
                    dst[u] = (t * CK + k) * WS + 4 * q;
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) *reinterpret_cast<float4*>(wl + dst[u]) = v[u];
            }
        }
        __syncthreads();

        // ---- sweep the taps: K steps of 4 channels
        int toff = tofft[0];
        for (int t = 0; t < p.ntaps; ++t) {
            const int toff_next = tofft[min(t + 1, p.ntaps - 1)];
            const float* __restrict__ xb = xin + bbase + toff;
#pragma unroll
            for (int kk = 0; kk < CK / 4; ++kk) {
                float b[4];
#pragma unroll
                for (int n = 0; n < 4; ++n) b[n] = xb[kk * 4 * CS + n * 16 * p.istride];
#pragma unroll
                for (int m = 0; m < MT; ++m) {
                    // A[i = lane&15][k = lane>>4] = wl[(t*CK + kk*4 + g) * WS + m*16 + (lane&15)]
                    const float av = wl[(t * CK + kk * 4 + g) * WS + m * 16 + j];
#pragma unroll
                    for (int n = 0; n < 4; ++n)
                        acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b[n], acc[m][n], 0, 0, 0);
                }
            }
            toff = toff_next;
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

static int patch_px(const ConvMParams& p) {
    const int px = p.xspan + (CV_COLS - 1) * p.istride;
    return p.vec ? ((px + p.sh + 3) & ~3) : px;
}

static int patch_cs(const ConvMParams& p) {
    const int PZ = p.zspan, PY = p.yspan + (CV_ROWS - 1) * p.istride;
    int CS = PZ * PY * patch_px(p);
    CS += (16 - (CS & 31) + 32) & 31;
    return CS;
}

static int lds_floats(const ConvMParams& p, int MT, int CK) {
    const int MP = 16 * MT, WS = (MP == 16) ? 16 : MP + 16;
    return CK * patch_cs(p) + p.ntaps * CK * WS + ((p.ntaps + 3) & ~3);
}

template <int MT, int CK>
static int launch_cfg(const ConvMParams& pin, hipStream_t stream) {
    ConvMParams p = pin;
    const int bytes = lds_floats(p, MT, CK) * 4;
    if (bytes > 160 * 1024) return D3D_ERR_UNSUPPORTED;
    const int Ci = p.Ci0 + p.Ci1;
    if (p.Ci1 > 0 && p.Ci0 % CK != 0) {
        set_error("conv_mfma: input split %d+%d is not aligned to the %d-channel chunk", p.Ci0, p.Ci1, CK);
        return D3D_ERR_UNSUPPORTED;
    }
    (void)Ci;
    const int PZ = p.zspan, PY = p.yspan + (CV_ROWS - 1) * p.istride;
    p.PX = patch_px(p);
    p.CS = patch_cs(p);
    const int per_row = p.vec ? p.PX / 4 : p.PX;
    const long nitems = (long)CK * PZ * PY * per_row;
    p.mg_row = (int)((1L << 24) / per_row + 1);
    p.mg_c = 65536 / (PZ * PY) + 1;
    p.mg_y = 65536 / PY + 1;
    // exactness of the reciprocal arithmetic (e * mg >> 24 with e < nitems, row * mg >> 16 with row < CK*PZ*PY)
    if (nitems >= (1L << 24) / (per_row + 1) || (long)CK * PZ * PY >= 65536 / (PZ * PY + 1) + 1) {
        set_error("conv_mfma: patch too large for the staging index arithmetic");
        return D3D_ERR_UNSUPPORTED;
    }
    const long in_plane = (long)p.H * p.W, in_vol = in_plane * p.D;
    if ((long)CK * in_vol >= (1L << 29)) {
        set_error("conv_mfma: %d-channel block exceeds 32-bit buffer offsets", CK);
        return D3D_ERR_UNSUPPORTED;
    }
    auto kern = conv_mfma_kernel<MT, CK>;
    static int attr_bytes = 0;
    if (bytes > attr_bytes) {
        int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(kern), 160 * 1024);
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
    // largest channel chunk (no larger than the channel count needs, and dividing the first input when two are
    // concatenated) whose patch + weights leave room for two workgroups per CU
    const int Ci = p.Ci0 + p.Ci1;
    auto fits = [&](int ck) { return p.Ci1 == 0 || p.Ci0 % ck == 0; };
    if (Ci > 8 && fits(16) && lds_floats(p, MT, 16) * 4 <= 80 * 1024) return launch_cfg<MT, 16>(p, stream);
    if (Ci > 4 && fits(8) && lds_floats(p, MT, 8) * 4 <= 80 * 1024) return launch_cfg<MT, 8>(p, stream);
    if (Ci > 8 && fits(16) && lds_floats(p, MT, 16) * 4 <= 150 * 1024 && lds_floats(p, MT, 4) * 4 > 80 * 1024)
        return launch_cfg<MT, 16>(p, stream);
    if (Ci > 4 && fits(8) && lds_floats(p, MT, 8) * 4 <= 150 * 1024 && lds_floats(p, MT, 4) * 4 > 80 * 1024)
        return launch_cfg<MT, 8>(p, stream);
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
    p.vec = (W % 4 == 0) && ((reinterpret_cast<uintptr_t>(in0) | reinterpret_cast<uintptr_t>(in1)) & 15) == 0;
    p.sh = p.vec ? (p.xmin & 3) : 0;
    return launch_conv_mfma(p, (hipStream_t)stream);
}

}  // extern "C"
