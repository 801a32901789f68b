// Output side of the path: the reference writes each depth / confidence map as a PFM file whose
// rows run bottom-up (mvs/mvs_cas/datasets/data_io.py:196-223 save_pfm_utf8: np.flipud, then tofile; read side
// data_io.py:150-193 / IO/pfm.py:19-60: fromfile, reshape, np.flipud).  Here the flip is one streaming pass on the
// device into a staging buffer that a single asynchronous D2H copy moves to pinned host memory in file order
// (deep3d_aerial_amd/predict.py PfmWriter); the same kernel un-flips a file payload uploaded for the fusion step.
// 8 B per pixel (one read, one write), 16-byte accesses when rows are 16-byte aligned.
#include "common.h"

namespace d3d {

struct FlipMaps {
    const float* in[8];
};

template <int VEC>
__global__ __launch_bounds__(256) void flip_rows_kernel(FlipMaps m, int H, int Wv, float* __restrict__ out) {
    // Wv = row length in units of VEC floats
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y, k = blockIdx.z;
    if (x >= Wv) return;
    const long row = (long)Wv;
    if constexpr (VEC == 4) {
        const float4* src = reinterpret_cast<const float4*>(m.in[k]) + (long)y * row + x;
        float4* dst = reinterpret_cast<float4*>(out) + ((long)k * H + (H - 1 - y)) * row + x;
        *dst = *src;
    } else {
        out[((long)k * H + (H - 1 - y)) * row + x] = m.in[k][(long)y * row + x];
    }
}

}  // namespace d3d

using namespace d3d;

extern "C" int d3d_flip_rows(const float* const* maps, int n, int H, int W, float* out, d3d_stream_t stream) {
    D3D_REQUIRE(maps && out, "null pointer");
    D3D_REQUIRE(n >= 1 && n <= 8, "n=%d maps (1..8)", n);
    D3D_REQUIRE(H > 0 && W > 0 && H <= 65535, "bad dims %dx%d", H, W);
    FlipMaps m = {};
    bool vec = (W % 4 == 0) && (((uintptr_t)out & 15) == 0);
    for (int k = 0; k < n; ++k) {
        D3D_REQUIRE(maps[k] && maps[k] != out, "null or aliased map %d", k);
        m.in[k] = maps[k];
        vec = vec && (((uintptr_t)maps[k] & 15) == 0);
    }
    if (vec) {
        const int Wv = W / 4;
        hipLaunchKernelGGL(flip_rows_kernel<4>, dim3(ceil_div(Wv, 256), H, n), dim3(256), 0, (hipStream_t)stream, m, H, Wv,
                           out);
    } else {
        hipLaunchKernelGGL(flip_rows_kernel<1>, dim3(ceil_div(W, 256), H, n), dim3(256), 0, (hipStream_t)stream, m, H, W,
                           out);
    }
    D3D_LAUNCH_CHECK("flip_rows_kernel launch");
    return D3D_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// Input side: a decoded 8-bit image [h,w,C] interleaved -> centre-crop window -> per-image,
// per-channel normalisation -> planar float32 [C,H,W], as the dataset item builder does on the host for every view
// of every item (mvs/mvs_cas/datasets/preprocess.py:60-88 crop_input, :92-117 center_image; cas_normal_eval.py:
// 112-147).  Two streaming passes over the crop window: integer sums (exact for 8-bit data), then normalise.
// ---------------------------------------------------------------------------------------------------------------
namespace d3d {

__global__ __launch_bounds__(256) void image_stats_kernel(const unsigned char* __restrict__ img, int w, int C, int y0,
                                                          int x0, int H, int W, unsigned long long* __restrict__ sums) {
    // one block per crop row segment of 256*4 samples; sums[2c] = sum, sums[2c+1] = sum of squares of channel c
    const int y = blockIdx.y;
    const long rowbase = ((long)(y0 + y) * w + x0) * C;
    const int n = W * C;  // interleaved samples in this crop row
    unsigned int s[4] = {0, 0, 0, 0}, q[4] = {0, 0, 0, 0};
    for (int i = blockIdx.x * 1024 + threadIdx.x; i < n && i < (blockIdx.x + 1) * 1024; i += 256) {
        const unsigned v = img[rowbase + i];
        const int cc = i % C;
#pragma unroll
        for (int c = 0; c < 4; ++c) {  // (static indices keep s / q in registers)
            s[c] += cc == c ? v : 0u;
            q[c] += cc == c ? v * v : 0u;
        }
    }
    __shared__ unsigned int red[8][256 / 64];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        unsigned a = s[c], b = q[c];
        for (int o = 32; o > 0; o >>= 1) {
            a += __shfl_down(a, o);
            b += __shfl_down(b, o);
        }
        if ((threadIdx.x & 63) == 0) {
            red[2 * c][threadIdx.x >> 6] = a;
            red[2 * c + 1][threadIdx.x >> 6] = b;
        }
    }
    __syncthreads();
    if (threadIdx.x < 2 * C) {
        unsigned long long t = 0;
        for (int k = 0; k < 4; ++k) t += red[threadIdx.x][k];
        atomicAdd(&sums[threadIdx.x], t);
    }
}

__global__ __launch_bounds__(256) void image_center_kernel(const unsigned char* __restrict__ img, int w, int C, int y0,
                                                           int x0, int H, int W, int mode,
                                                           const unsigned long long* __restrict__ sums,
                                                           float* __restrict__ out) {
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (x >= W) return;
    const unsigned char* p = img + ((long)(y0 + y) * w + x0 + x) * C;
    const double N = (double)H * W;
    for (int c = 0; c < C; ++c) {
        const float v = (float)p[c];
        float r;
        if (mode == 0) {  // 'standard': /255 (preprocess.py:95-96)
            r = v / 255.0f;
        } else if (mode == 2) {  // 'vit': fixed per-channel mean / std (preprocess.py:105-110), float32 arithmetic
            const float pm = c == 0 ? 123.675f : c == 1 ? 116.28f : 103.53f;
            const float ps = c == 0 ? 58.395f : c == 1 ? 57.12f : 57.375f;
            r = (v - pm) / (ps + 0.00000001f);
        } else {  // 'mean': (x - mean) / (sqrt(var) + 1e-8) (preprocess.py:98-103)
            const double sm = (double)sums[2 * c], sq = (double)sums[2 * c + 1];
            const double mean = sm / N;
            // N*sq - sm^2 is an exact integer below 2^63 up to 11.9 M pixels of 8-bit data
            const double var = (N * sq - sm * sm) / (N * N);
            r = (v - (float)mean) / ((float)sqrt(var) + 0.00000001f);
        }
        out[((long)c * H + y) * W + x] = r;
    }
}

}  // namespace d3d

extern "C" int d3d_center_image_u8(const unsigned char* img, int h, int w, int channels, int y0, int x0, int H, int W,
                                   int mode, unsigned long long* sums, float* out, d3d_stream_t stream) {
    D3D_REQUIRE(img && sums && out, "null pointer");
    D3D_REQUIRE(channels >= 1 && channels <= 4, "channels=%d (1..4)", channels);
    D3D_REQUIRE(h > 0 && w > 0 && H > 0 && W > 0 && H <= 65535, "bad dims %dx%d -> %dx%d", h, w, H, W);
    D3D_REQUIRE(y0 >= 0 && x0 >= 0 && y0 + H <= h && x0 + W <= w, "crop window (%d,%d)+%dx%d outside %dx%d", y0, x0, H,
                W, h, w);
    D3D_REQUIRE(mode == 0 || mode == 1 || (mode == 2 && channels == 3), "mode %d (0 standard, 1 mean, 2 vit: 3 channels)", mode);
    hipStream_t st = (hipStream_t)stream;
    if (mode == 1) {
        int rc = hip_status(hipMemsetAsync(sums, 0, 8 * sizeof(unsigned long long), st), "hipMemsetAsync(sums)");
        if (rc != D3D_OK) return rc;
        hipLaunchKernelGGL(image_stats_kernel, dim3(ceil_div((long)W * channels, 1024), H), dim3(256), 0, st, img, w,
                           channels, y0, x0, H, W, sums);
        D3D_LAUNCH_CHECK("image_stats_kernel launch");
    }
    hipLaunchKernelGGL(image_center_kernel, dim3(ceil_div(W, 256), H), dim3(256), 0, st, img, w, channels, y0, x0, H, W,
                       mode, sums, out);
    D3D_LAUNCH_CHECK("image_center_kernel launch");
    return D3D_OK;
}
