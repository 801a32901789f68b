// Pooled-context heads of the AdaMVS feature pyramid (adamvs.py:75-101, 116-151 of the reference):
//     out = head(cat(up(branch_4(f)), up(branch_8(f)), f)),   branch_p = AvgPool2d(p) -> 1x1 conv + BN + ReLU,
//     up = bilinear resize to the size of f (align_corners=False), head = 1x1 conv without bias.
// The reference materialises both upsampled branches and the concat at the resolution of f (at stage 3 that is 16 channels
// at full image resolution written and read again).  The head is linear and a 1x1 convolution commutes with a resize, so
//     out = W_f f + up(W_a a) + up(W_b b)
// with a, b the low-resolution branch outputs: the products W_a a, W_b b are taken at 1/16 and 1/64 of the pixels (plain
// GEMMs on the host side), and ONE streaming kernel reads f once, samples the two small maps and writes out -- no
// upsampled tensor, no concat.  The two pools read f once as well.  HBM bound; exact fp32.
#include <cstdint>

#include "common.h"

namespace d3d {

namespace {

typedef float f4 __attribute__((ext_vector_type(4)));
typedef const float __attribute__((address_space(4))) cfloat;

// One thread = one 8 x 8 block of the input: four 4 x 4 means and (where the block is whole) the 8 x 8 mean, each summed
// row-major like a sequential pooling loop.
__global__ __launch_bounds__(256) void avgpool_4_8_kernel(const float* __restrict__ in, int C, int H, int W,
                                                          float* __restrict__ out4, float* __restrict__ out8) {
    const int H4 = H / 4, W4 = W / 4, H8 = H / 8, W8 = W / 8;
    const int bx = blockIdx.x * 64 + (threadIdx.x & 63), by = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (2 * bx >= W4 || 2 * by >= H4) return;
    const bool right = 2 * bx + 1 < W4, low = 2 * by + 1 < H4;
    for (int c = blockIdx.z; c < C; c += gridDim.z) {
        const float* __restrict__ src = in + ((size_t)c * H + 8 * by) * W + 8 * bx;
        f4 v[8][2];
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const bool rok = r < 4 || low;
            v[r][0] = rok ? *reinterpret_cast<const f4*>(src + (size_t)r * W) : (f4){0, 0, 0, 0};
            v[r][1] = rok && right ? *reinterpret_cast<const f4*>(src + (size_t)r * W + 4) : (f4){0, 0, 0, 0};
        }
        float s4[2][2] = {{0, 0}, {0, 0}}, s8 = 0.0f;
#pragma unroll
        for (int r = 0; r < 8; ++r)
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    s4[r >> 2][h] += v[r][h][k];
                    s8 += v[r][h][k];
                }
        float* __restrict__ o4 = out4 + ((size_t)c * H4 + 2 * by) * W4 + 2 * bx;
        o4[0] = s4[0][0] * 0.0625f;
        if (right) o4[1] = s4[0][1] * 0.0625f;
        if (low) {
            o4[W4] = s4[1][0] * 0.0625f;
            if (right) o4[W4 + 1] = s4[1][1] * 0.0625f;
        }
        if (bx < W8 && by < H8) out8[((size_t)c * H8 + by) * W8 + bx] = s8 * 0.015625f;
    }
}

// align_corners=False source coordinate of output index `dst` (negative coordinates clamp to 0, as F.interpolate does)
__device__ __forceinline__ void lin_coord_ctx(int dst, float scale, int in_size, int& i0, int& i1, float& l1) {
    float s = ((float)dst + 0.5f) * scale - 0.5f;
    s = s < 0.0f ? 0.0f : s;
    i0 = min((int)s, in_size - 1);
    i1 = i0 + (i0 < in_size - 1 ? 1 : 0);
    l1 = s - (float)i0;
}

// + bilinear sample of the small map t [CO, Ht, Wt] at the four pixels (X..X+3, Y).  The map is >= 3 x coarser than the
// output, so the four pixels' taps lie in three adjacent columns: six loads per channel instead of sixteen.
template <int CO>
__device__ __forceinline__ void add_branch(f4 (&acc)[CO], const float* __restrict__ t, int Ht, int Wt, int H, int W, int X, int Y) {
    int y0, y1, x0[4], x1[4];
    float ly, lx[4];
    lin_coord_ctx(Y, (float)Ht / (float)H, Ht, y0, y1, ly);
#pragma unroll
    for (int k = 0; k < 4; ++k) lin_coord_ctx(X + k, (float)Wt / (float)W, Wt, x0[k], x1[k], lx[k]);
    const int xb = x0[0], c1 = min(xb + 1, Wt - 1), c2 = min(xb + 2, Wt - 1);
    const float hy = 1.0f - ly;
    const size_t tp = (size_t)Ht * Wt;
#pragma unroll
    for (int co = 0; co < CO; ++co) {
        const float* __restrict__ r0 = t + co * tp + (size_t)y0 * Wt;
        const float* __restrict__ r1 = t + co * tp + (size_t)y1 * Wt;
        const float a0 = r0[xb], a1 = r0[c1], a2 = r0[c2], b0 = r1[xb], b1 = r1[c1], b2 = r1[c2];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int j0 = x0[k] - xb, j1 = x1[k] - xb;   // 0 | 1 and 0 | 1 | 2
            const float t00 = j0 ? a1 : a0, t01 = j1 == 0 ? a0 : j1 == 1 ? a1 : a2;
            const float t10 = j0 ? b1 : b0, t11 = j1 == 0 ? b0 : j1 == 1 ? b1 : b2;
            const float hx = 1.0f - lx[k];
            acc[co][k] += hy * (hx * t00 + lx[k] * t01) + ly * (hx * t10 + lx[k] * t11);
        }
    }
}

// A thread owns four pixels of a row and CO = 8 output channels (blockIdx.z picks the block of 8: 32 accumulator registers
// + the six taps per channel and branch; wider blocks spill -- at 16 | 32 input channels f is re-read per block, from L2).
template <int CI, int CO>
__global__ __launch_bounds__(256) void conv1x1_context_kernel(const float* __restrict__ f, const float* __restrict__ weight,
                                                              const float* __restrict__ a, int Ha, int Wa,
                                                              const float* __restrict__ b, int Hb, int Wb, int H, int W,
                                                              float* __restrict__ out) {
    const int X = (blockIdx.x * 64 + (threadIdx.x & 63)) * 4, Y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (X >= W || Y >= H) return;
    const size_t plane = (size_t)H * W, o = (size_t)Y * W + X;
    const int co0 = blockIdx.z * CO;
    cfloat* w = (cfloat*)weight + (size_t)co0 * CI;   // [C_out][CI]: uniform indices -> scalar loads
    a += (size_t)co0 * Ha * Wa;
    b += (size_t)co0 * Hb * Wb;
    out += (size_t)co0 * plane;
    f4 acc[CO];
#pragma unroll
    for (int co = 0; co < CO; ++co) acc[co] = (f4){0, 0, 0, 0};
#pragma unroll 8
    for (int ci = 0; ci < CI; ++ci) {   // (eight channels' loads in flight: unrolling all 32 takes every register)
        const f4 v = *reinterpret_cast<const f4*>(f + ci * plane + o);
#pragma unroll
        for (int co = 0; co < CO; ++co) acc[co] += w[co * CI + ci] * v;
    }
    add_branch<CO>(acc, a, Ha, Wa, H, W, X, Y);
    add_branch<CO>(acc, b, Hb, Wb, H, W, X, Y);
#pragma unroll
    for (int co = 0; co < CO; ++co) *reinterpret_cast<f4*>(out + co * plane + o) = acc[co];
}

// A bias b added to the INPUT of a zero-padded 3 x 3 convolution reaches output pixel p through the taps whose source lies inside
// the image: sum of taps[co][ky][kx] over them (taps = W . b).  In the interior that is the constant sum the caller adds as an
// ordinary bias; on the one-pixel border this kernel takes the outside taps back out.  One thread per (channel, border pixel).
__global__ __launch_bounds__(256) void conv3x3_bias_border_kernel(float* __restrict__ out, const float* __restrict__ taps, int Co,
                                                                  int H, int W) {
    const int per = 2 * W + 2 * max(H - 2, 0);
    const int i = blockIdx.x * 256 + threadIdx.x, co = blockIdx.y;
    if (i >= per) return;
    int y, x;
    if (i < W) { y = 0; x = i; }
    else if (i < 2 * W) { y = H - 1; x = i - W; }
    else { const int j = i - 2 * W; y = 1 + (j >> 1); x = (j & 1) ? W - 1 : 0; }
    if (H == 1 && i >= W) return;                       // a single row is its own top and bottom
    if (W == 1 && i >= 2 * W && (i & 1)) return;        // a single column its own left and right
    float miss = 0.0f;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const int sy = y + ky - 1, sx = x + kx - 1;
            if (sy < 0 || sy >= H || sx < 0 || sx >= W) miss += taps[(co * 3 + ky) * 3 + kx];
        }
    out[((size_t)co * H + y) * W + x] -= miss;
}

// Plain 1 x 1 convolution in exact fp32 (the output layers of the feature pyramids, module.py:677-679 / 701-703 and their UNet
// siblings): a streaming kernel -- a thread owns four consecutive pixels of the flattened plane and a block of 8 output channels
// (blockIdx.y), reads the CI input planes with 16-byte loads (re-read per block of 8, from L2) and keeps 32 accumulators; weights
// [CI][CO8 blocks] arrive through scalar loads.  out = act(scale * conv + shift) (+ skip, added last).  Round 5: these three layers
// per image were the last users of round 1's conv_stream_kernel in a RED-Net / CasMVSNet view (0.9 - 2.2 TB/s there).
template <int CI>
__global__ __launch_bounds__(256) void conv1x1_f32_kernel(const float* __restrict__ in, const float* __restrict__ wt,
                                                          const float* __restrict__ scale, const float* __restrict__ shift,
                                                          const float* __restrict__ skip, int act, int Co, long plane,
                                                          float* __restrict__ out) {
    const long o = ((long)blockIdx.x * 256 + threadIdx.x) * 4;
    if (o >= plane) return;
    const int co0 = blockIdx.y * 8;
    cfloat* w = (cfloat*)wt + (size_t)blockIdx.y * CI * 8;   // [block][CI][8]: uniform indices -> scalar loads
    f4 acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = (f4){0, 0, 0, 0};
#pragma unroll 8
    for (int ci = 0; ci < CI; ++ci) {
        const f4 v = *reinterpret_cast<const f4*>(in + ci * plane + o);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float wj = w[ci * 8 + j];
#pragma unroll
            for (int k = 0; k < 4; ++k) acc[j][k] = fmaf(wj, v[k], acc[j][k]);
        }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int co = co0 + j;
        if (co >= Co) break;
        f4 y = acc[j] * (scale ? scale[co] : 1.0f) + (shift ? shift[co] : 0.0f);
        if (act == 1) y = __builtin_elementwise_max(y, (f4){0, 0, 0, 0});
        if (skip) y = y + *reinterpret_cast<const f4*>(skip + co * plane + o);
        *reinterpret_cast<f4*>(out + co * plane + o) = y;
    }
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace

}  // namespace d3d

using namespace d3d;

extern "C" int d3d_avgpool2d_4_8(const float* in, int C, int H, int W, float* out4, float* out8, d3d_stream_t stream) {
    D3D_REQUIRE(in && out4 && out8, "null pointer");
    D3D_REQUIRE(C > 0 && H >= 8 && W >= 8, "bad dims C=%d H=%d W=%d (needs H, W >= 8)", C, H, W);
    if (W % 4 != 0 || !aligned16(in)) {
        set_error("d3d_avgpool2d_4_8: W = %d (multiple of 4) and a 16-byte aligned input are required", W);
        return D3D_ERR_UNSUPPORTED;
    }
    const int nbx = ceil_div(W / 4, 2), nby = ceil_div(H / 4, 2);
    dim3 grid(ceil_div(nbx, 64), ceil_div(nby, 4), C < 64 ? C : 64);
    D3D_REQUIRE(grid.y <= 65535, "H=%d too large", H);
    hipLaunchKernelGGL(avgpool_4_8_kernel, grid, dim3(256), 0, (hipStream_t)stream, in, C, H, W, out4, out8);
    D3D_LAUNCH_CHECK("avgpool_4_8_kernel launch");
    return D3D_OK;
}

extern "C" int d3d_conv1x1_context(const float* f, int Ci, const float* weight, const float* a, int Ha, int Wa, const float* b,
                                   int Hb, int Wb, int Co, int H, int W, float* out, d3d_stream_t stream) {
    D3D_REQUIRE(f && weight && a && b && out, "null pointer");
    D3D_REQUIRE(H > 0 && W > 0 && Ha > 0 && Wa > 0 && Hb > 0 && Wb > 0, "bad dims");
    const bool shape = Ci == Co && (Ci == 8 || Ci == 16 || Ci == 32) && W % 4 == 0 && 3 * Wa <= W && 3 * Wb <= W &&
                       aligned16(f) && aligned16(out);
    if (!shape) {
        set_error("d3d_conv1x1_context: C_in = %d, C_out = %d (8 | 16 | 32, equal), W = %d (multiple of 4), branch widths %d, %d "
                  "(at most W / 3), 16-byte aligned tensors: not taken", Ci, Co, W, Wa, Wb);
        return D3D_ERR_UNSUPPORTED;
    }
    dim3 grid(ceil_div(W / 4, 64), ceil_div(H, 4), Co / 8);
    D3D_REQUIRE(grid.y <= 65535, "H=%d too large", H);
    hipStream_t st = (hipStream_t)stream;
    if (Ci == 8) hipLaunchKernelGGL((conv1x1_context_kernel<8, 8>), grid, dim3(256), 0, st, f, weight, a, Ha, Wa, b, Hb, Wb, H, W, out);
    else if (Ci == 16) hipLaunchKernelGGL((conv1x1_context_kernel<16, 8>), grid, dim3(256), 0, st, f, weight, a, Ha, Wa, b, Hb, Wb, H, W, out);
    else hipLaunchKernelGGL((conv1x1_context_kernel<32, 8>), grid, dim3(256), 0, st, f, weight, a, Ha, Wa, b, Hb, Wb, H, W, out);
    D3D_LAUNCH_CHECK("conv1x1_context_kernel launch");
    return D3D_OK;
}

// out [Co,H,W] = act(scale * Conv1x1(in) + shift) (+ skip, added last) in exact fp32.  in [Ci,H,W], Ci = 8 | 16 | 32; wt = the weight
// as [ceil(Co / 8)][Ci][8] (ops._pack_k1: blocks of 8 output channels, zero-padded); H * W a multiple of 4, 16-byte aligned tensors,
// else D3D_ERR_UNSUPPORTED (nothing launched).
extern "C" int d3d_conv2d_k1_f32(const float* in, const float* wt, const float* scale, const float* shift, const float* skip, int act,
                                 int Ci, int Co, int H, int W, float* out, d3d_stream_t stream) {
    D3D_REQUIRE(in && wt && out, "null pointer");
    D3D_REQUIRE(H > 0 && W > 0 && Co > 0, "bad dims");
    D3D_REQUIRE(act == 0 || act == 1, "bad act %d", act);
    const long plane = (long)H * W;
    if ((Ci != 8 && Ci != 16 && Ci != 32) || Co > 64 * 8 || plane % 4 != 0 || !aligned16(in) || !aligned16(out) || (skip && !aligned16(skip))) {
        set_error("d3d_conv2d_k1_f32: C_in = %d (8 | 16 | 32), H * W = %ld (a multiple of 4), 16-byte aligned tensors: not taken", Ci, plane);
        return D3D_ERR_UNSUPPORTED;
    }
    dim3 grid((unsigned)ceil_div(plane / 4, 256), (unsigned)ceil_div(Co, 8));
    hipStream_t st = (hipStream_t)stream;
    if (Ci == 8) hipLaunchKernelGGL((conv1x1_f32_kernel<8>), grid, dim3(256), 0, st, in, wt, scale, shift, skip, act, Co, plane, out);
    else if (Ci == 16) hipLaunchKernelGGL((conv1x1_f32_kernel<16>), grid, dim3(256), 0, st, in, wt, scale, shift, skip, act, Co, plane, out);
    else hipLaunchKernelGGL((conv1x1_f32_kernel<32>), grid, dim3(256), 0, st, in, wt, scale, shift, skip, act, Co, plane, out);
    D3D_LAUNCH_CHECK("conv1x1_f32_kernel launch");
    return D3D_OK;
}

extern "C" int d3d_conv3x3_bias_border(float* out, const float* taps, int Co, int H, int W, d3d_stream_t stream) {
    D3D_REQUIRE(out && taps, "null pointer");
    D3D_REQUIRE(Co > 0 && Co <= 65535 && H > 0 && W > 0, "bad dims Co=%d H=%d W=%d", Co, H, W);
    const int per = 2 * W + 2 * (H > 2 ? H - 2 : 0);
    hipLaunchKernelGGL(conv3x3_bias_border_kernel, dim3(ceil_div(per, 256), Co), dim3(256), 0, (hipStream_t)stream, out, taps, Co, H, W);
    D3D_LAUNCH_CHECK("conv3x3_bias_border_kernel launch");
    return D3D_OK;
}
