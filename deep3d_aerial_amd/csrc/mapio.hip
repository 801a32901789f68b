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
