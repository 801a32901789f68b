// Convolution family of the cost regularisers (fp32, direct form) for gfx950.
//
// One templated kernel covers nn.Conv2d / nn.Conv3d (k=3, pad=1, stride 1|2) and one
// covers the stride-2 transposed forms; the 2D case is the 3D kernel with a depth-1
// kernel (KZ = 1).  A lane owns one output pixel and COT output channels, input rows are
// read coalesced along x and the weights come in through wave-uniform (scalar) loads.
// Epilogue: per-channel affine (folded eval-mode BatchNorm or bias), ReLU, skip add.
// Reference citations: include/deep3d_planesweep.h.
#include "common.h"

namespace d3d {

struct ConvParams {
    const float* in0;
    const float* in1;  // second tensor of a channel concat (2D GRU inputs), may be null
    const float* weight;
    const float* scale;  // null = 1
    const float* shift;  // null = 0
    const float* skip;   // null = none
    float* out;
    int Ci0, Ci1, Co;
    int D, H, W;     // input dims
    int Do, Ho, Wo;  // output dims
    int stride;
    int act;             // 0 none, 1 relu
    int skip_after_act;  // 1: out = skip + act(y) ; 0: out = act(y + skip)
};

__device__ __forceinline__ float epilogue(float acc, int co, long oidx, const ConvParams& p) {
    float y = acc;
    if (p.scale) y *= p.scale[co];
    if (p.shift) y += p.shift[co];
    if (p.skip && !p.skip_after_act) y += p.skip[oidx];
    if (p.act == 1) y = fmaxf(y, 0.0f);
    if (p.skip && p.skip_after_act) y = p.skip[oidx] + y;
    return y;
}

template <int KZ, int COT>
__global__ __launch_bounds__(256) void conv_k3_kernel(ConvParams p) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int yz = blockIdx.y * 4 + (threadIdx.x >> 6);
    const int co0 = blockIdx.z * COT;
    if (x >= p.Wo || yz >= p.Ho * p.Do) return;
    const int z = yz / p.Ho, y = yz - z * p.Ho;
    const int Ci = p.Ci0 + p.Ci1;
    const int sz = (KZ == 1) ? 1 : p.stride;
    const long in_plane = (long)p.H * p.W;
    const long in_vol = in_plane * p.D;
    float acc[COT];
#pragma unroll
    for (int c = 0; c < COT; ++c) acc[c] = 0.0f;

    for (int ci = 0; ci < Ci; ++ci) {
        const float* __restrict__ src = (ci < p.Ci0) ? p.in0 + (long)ci * in_vol : p.in1 + (long)(ci - p.Ci0) * in_vol;
#pragma unroll
        for (int kz = 0; kz < KZ; ++kz) {
            const int iz = z * sz - (KZ / 2) + kz;
            if (iz < 0 || iz >= p.D) continue;
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                const int iy = y * p.stride - 1 + ky;
                if (iy < 0 || iy >= p.H) continue;
                const float* __restrict__ row = src + (long)iz * in_plane + (long)iy * p.W;
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    const int ix = x * p.stride - 1 + kx;
                    const float v = (ix >= 0 && ix < p.W) ? row[ix] : 0.0f;
#pragma unroll
                    for (int c = 0; c < COT; ++c) {
                        const int co = co0 + c;
                        const float wv = (co < p.Co) ? p.weight[(((long)co * Ci + ci) * KZ + kz) * 9 + ky * 3 + kx] : 0.0f;
                        acc[c] = fmaf(v, wv, acc[c]);
                    }
                }
            }
        }
    }
    const long out_plane = (long)p.Ho * p.Wo;
#pragma unroll
    for (int c = 0; c < COT; ++c) {
        const int co = co0 + c;
        if (co >= p.Co) break;
        const long oidx = ((long)co * p.Do + z) * out_plane + (long)y * p.Wo + x;
        p.out[oidx] = epilogue(acc[c], co, oidx, p);
    }
}

// Transposed conv, k=3, stride 2, padding 1, output_padding 1: out dims = 2x in dims
// (only H,W doubled when KZ == 1).  out[o] = sum over taps with (o + 1 - k) even.
template <int KZ, int COT>
__global__ __launch_bounds__(256) void convT_k3s2_kernel(ConvParams p) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int yz = blockIdx.y * 4 + (threadIdx.x >> 6);
    const int co0 = blockIdx.z * COT;
    if (x >= p.Wo || yz >= p.Ho * p.Do) return;
    const int z = yz / p.Ho, y = yz - z * p.Ho;
    const int Ci = p.Ci0;
    const long in_plane = (long)p.H * p.W;
    const long in_vol = in_plane * p.D;
    float acc[COT];
#pragma unroll
    for (int c = 0; c < COT; ++c) acc[c] = 0.0f;

    for (int ci = 0; ci < Ci; ++ci) {
        const float* __restrict__ src = p.in0 + (long)ci * in_vol;
#pragma unroll
        for (int kz = 0; kz < KZ; ++kz) {
            int iz = 0;
            if (KZ == 3) {
                const int tz = z + 1 - kz;
                if (tz < 0 || (tz & 1) || (tz >> 1) >= p.D) continue;
                iz = tz >> 1;
            }
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                const int ty = y + 1 - ky;
                if (ty < 0 || (ty & 1) || (ty >> 1) >= p.H) continue;
                const float* __restrict__ row = src + (long)iz * in_plane + (long)(ty >> 1) * p.W;
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    const int tx = x + 1 - kx;
                    const bool ok = (tx >= 0) && !(tx & 1) && ((tx >> 1) < p.W);
                    const float v = ok ? row[tx >> 1] : 0.0f;
#pragma unroll
                    for (int c = 0; c < COT; ++c) {
                        const int co = co0 + c;
                        const float wv = (co < p.Co) ? p.weight[(((long)ci * p.Co + co) * KZ + kz) * 9 + ky * 3 + kx] : 0.0f;
                        acc[c] = fmaf(v, wv, acc[c]);
                    }
                }
            }
        }
    }
    const long out_plane = (long)p.Ho * p.Wo;
#pragma unroll
    for (int c = 0; c < COT; ++c) {
        const int co = co0 + c;
        if (co >= p.Co) break;
        const long oidx = ((long)co * p.Do + z) * out_plane + (long)y * p.Wo + x;
        p.out[oidx] = epilogue(acc[c], co, oidx, p);
    }
}

template <int KZ>
static int launch_conv(const ConvParams& p, bool transposed, hipStream_t stream) {
    const long rows = (long)p.Ho * p.Do;
    const int cot = p.Co >= 8 ? 8 : (p.Co >= 4 ? 4 : 1);
    dim3 grid(ceil_div(p.Wo, 64), ceil_div(rows, 4), ceil_div(p.Co, cot));
    D3D_REQUIRE(grid.y <= 65535u && grid.z <= 65535u, "output too tall for one launch (%ld rows)", rows);
    dim3 block(256);
    if (!transposed) {
        if (cot == 8) hipLaunchKernelGGL((conv_k3_kernel<KZ, 8>), grid, block, 0, stream, p);
        else if (cot == 4) hipLaunchKernelGGL((conv_k3_kernel<KZ, 4>), grid, block, 0, stream, p);
        else hipLaunchKernelGGL((conv_k3_kernel<KZ, 1>), grid, block, 0, stream, p);
    } else {
        if (cot == 8) hipLaunchKernelGGL((convT_k3s2_kernel<KZ, 8>), grid, block, 0, stream, p);
        else if (cot == 4) hipLaunchKernelGGL((convT_k3s2_kernel<KZ, 4>), grid, block, 0, stream, p);
        else hipLaunchKernelGGL((convT_k3s2_kernel<KZ, 1>), grid, block, 0, stream, p);
    }
    D3D_LAUNCH_CHECK("conv kernel launch");
    return D3D_OK;
}

}  // namespace d3d

using namespace d3d;

extern "C" {

int d3d_conv3d_k3(const float* in, const float* weight, const float* scale, const float* shift, const float* skip,
                  int relu, int Ci, int Co, int D, int H, int W, int stride, float* out, d3d_stream_t stream) {
    D3D_REQUIRE(in && weight && out, "null pointer");
    D3D_REQUIRE(Ci > 0 && Co > 0 && D > 0 && H > 0 && W > 0, "bad dims");
    D3D_REQUIRE(stride == 1 || stride == 2, "stride must be 1 or 2 (got %d)", stride);
    ConvParams p = {};
    p.in0 = in; p.weight = weight; p.scale = scale; p.shift = shift; p.skip = skip; p.out = out;
    p.Ci0 = Ci; p.Co = Co; p.D = D; p.H = H; p.W = W;
    p.Do = (D - 1) / stride + 1; p.Ho = (H - 1) / stride + 1; p.Wo = (W - 1) / stride + 1;
    p.stride = stride; p.act = relu ? 1 : 0; p.skip_after_act = 1;
    return launch_conv<3>(p, false, (hipStream_t)stream);
}

int d3d_convtranspose3d_k3s2(const float* in, const float* weight, const float* scale, const float* shift,
                             const float* skip, int relu, int Ci, int Co, int D, int H, int W, float* out,
                             d3d_stream_t stream) {
    D3D_REQUIRE(in && weight && out, "null pointer");
    D3D_REQUIRE(Ci > 0 && Co > 0 && D > 0 && H > 0 && W > 0, "bad dims");
    ConvParams p = {};
    p.in0 = in; p.weight = weight; p.scale = scale; p.shift = shift; p.skip = skip; p.out = out;
    p.Ci0 = Ci; p.Co = Co; p.D = D; p.H = H; p.W = W;
    p.Do = 2 * D; p.Ho = 2 * H; p.Wo = 2 * W;
    p.stride = 2; p.act = relu ? 1 : 0; p.skip_after_act = 1;
    return launch_conv<3>(p, true, (hipStream_t)stream);
}

int d3d_conv2d_k3(const float* in0, int Ci0, const float* in1, int Ci1, const float* weight, const float* scale,
                  const float* shift, const float* skip, int act, int Co, int H, int W, int stride, float* out,
                  d3d_stream_t stream) {
    D3D_REQUIRE(in0 && weight && out, "null pointer");
    D3D_REQUIRE(Ci0 > 0 && Ci1 >= 0 && (Ci1 == 0 || in1), "bad input channel split %d+%d", Ci0, Ci1);
    D3D_REQUIRE(Co > 0 && H > 0 && W > 0, "bad dims");
    D3D_REQUIRE(stride == 1 || stride == 2, "stride must be 1 or 2 (got %d)", stride);
    D3D_REQUIRE(act == 0 || act == 1, "bad act %d", act);
    ConvParams p = {};
    p.in0 = in0; p.in1 = in1; p.weight = weight; p.scale = scale; p.shift = shift; p.skip = skip; p.out = out;
    p.Ci0 = Ci0; p.Ci1 = Ci1; p.Co = Co; p.D = 1; p.H = H; p.W = W;
    p.Do = 1; p.Ho = (H - 1) / stride + 1; p.Wo = (W - 1) / stride + 1;
    p.stride = stride; p.act = act; p.skip_after_act = 1;
    return launch_conv<1>(p, false, (hipStream_t)stream);
}

int d3d_convtranspose2d_k3s2(const float* in, const float* weight, const float* scale, const float* shift,
                             const float* skip, int skip_after_act, int act, int Ci, int Co, int H, int W,
                             float* out, d3d_stream_t stream) {
    D3D_REQUIRE(in && weight && out, "null pointer");
    D3D_REQUIRE(Ci > 0 && Co > 0 && H > 0 && W > 0, "bad dims");
    D3D_REQUIRE(act == 0 || act == 1, "bad act %d", act);
    ConvParams p = {};
    p.in0 = in; p.weight = weight; p.scale = scale; p.shift = shift; p.skip = skip; p.out = out;
    p.Ci0 = Ci; p.Co = Co; p.D = 1; p.H = H; p.W = W;
    p.Do = 1; p.Ho = 2 * H; p.Wo = 2 * W;
    p.stride = 2; p.act = act; p.skip_after_act = skip_after_act ? 1 : 0;
    return launch_conv<1>(p, true, (hipStream_t)stream);
}

}  // extern "C"
