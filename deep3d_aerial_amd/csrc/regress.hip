// Depth regression family for gfx950: softmax / soft-argmin / confidence, the online
// exp-sum regression of the slice-recurrent models, depth-hypothesis generation and the
// bilinear resampling that sits between stages.  All of these are HBM-bound streaming
// kernels: one lane per output pixel, planes walked in registers, every global access
// coalesced along x.  Reference citations are in include/deep3d_planesweep.h.
#include "common.h"

#include <cstdint>
#include <type_traits>

namespace d3d {

// VEC consecutive pixels per thread: 16-byte loads / stores when VEC = 4 (the streaming kernels below are bound by the
// number of memory instructions in flight, not by arithmetic).  Per element the arithmetic is the same in either form.
template <int VEC> struct VecT;
template <> struct VecT<1> { typedef float type; };
template <> struct VecT<4> { typedef float type __attribute__((ext_vector_type(4))); };
template <int VEC> __device__ __forceinline__ float vget(const typename VecT<VEC>::type& v, int k);
template <> __device__ __forceinline__ float vget<1>(const float& v, int) { return v; }
template <> __device__ __forceinline__ float vget<4>(const VecT<4>::type& v, int k) { return v[k]; }
template <int VEC> __device__ __forceinline__ void vset(typename VecT<VEC>::type& v, int k, float x);
template <> __device__ __forceinline__ void vset<1>(float& v, int, float x) { v = x; }
template <> __device__ __forceinline__ void vset<4>(VecT<4>::type& v, int k, float x) { v[k] = x; }

static bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// depth of plane d at pixel i: [D] | [D,plane] | affine (lo, step) maps [2,plane] (include/deep3d_planesweep.h)
__device__ __forceinline__ float depth_at(const float* __restrict__ depth, int depth_mode, int d, long plane, long i) {
    if (depth_mode == D3D_DEPTH_PER_PIXEL) return depth[d * plane + i];
    if (depth_mode == D3D_DEPTH_AFFINE) return __fadd_rn(depth[i], __fmul_rn((float)d, depth[plane + i]));
    return depth[d];
}

// cas_mvsnet.py:69-76 (and, WITH_VAR, ucsnet.py:137-151: the spread of the distribution around the regressed depth,
// exp_variance = lamb * sqrt(sum_d p_d (dv_d - depth)^2) -- one more sweep over D).  Two sweeps over D (max, then
// exp-sums); the 4-plane window is re-read from cache at the end.
template <int VEC, bool WITH_VAR>
__global__ __launch_bounds__(256) void softargmin_conf4_kernel(const float* __restrict__ cost,
                                                                const float* __restrict__ depth, int depth_mode,
                                                                int D, long plane, float lamb, float* __restrict__ depth_out,
                                                                float* __restrict__ conf_out, float* __restrict__ var_out) {
    typedef typename VecT<VEC>::type V;
    const long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * VEC;
    if (i >= plane) return;
    float mx[VEC], den[VEC], dep[VEC], idx[VEC];
#pragma unroll
    for (int k = 0; k < VEC; ++k) { mx[k] = -INFINITY; den[k] = 0.0f; dep[k] = 0.0f; idx[k] = 0.0f; }
    for (int d = 0; d < D; ++d) {
        const V c = *reinterpret_cast<const V*>(cost + d * plane + i);
#pragma unroll
        for (int k = 0; k < VEC; ++k) mx[k] = fmaxf(mx[k], vget<VEC>(c, k));
    }
    for (int d = 0; d < D; ++d) {
        const V c = *reinterpret_cast<const V*>(cost + d * plane + i);
        V dv;
        if (depth_mode == D3D_DEPTH_PER_PIXEL) dv = *reinterpret_cast<const V*>(depth + d * plane + i);
#pragma unroll
        for (int k = 0; k < VEC; ++k) {
            const float e = __expf(vget<VEC>(c, k) - mx[k]);
            const float dvk = depth_mode == D3D_DEPTH_PER_PIXEL ? vget<VEC>(dv, k) : depth_at(depth, depth_mode, d, plane, i + k);
            den[k] += e;
            dep[k] = fmaf(e, dvk, dep[k]);
            idx[k] = fmaf(e, (float)d, idx[k]);
        }
    }
    V dout, cout;
    float inv[VEC];
#pragma unroll
    for (int k = 0; k < VEC; ++k) {
        inv[k] = 1.0f / den[k];
        int kk0 = (int)(idx[k] * inv[k]);  // .long() truncation (value >= 0)
        kk0 = min(max(kk0, 0), D - 1);
        float conf = 0.0f;
#pragma unroll
        for (int j = -1; j <= 2; ++j) {
            const int kk = kk0 + j;
            if (kk >= 0 && kk < D) conf += __expf(cost[kk * plane + i + k] - mx[k]);
        }
        vset<VEC>(dout, k, dep[k] * inv[k]);
        vset<VEC>(cout, k, conf * inv[k]);
    }
    *reinterpret_cast<V*>(depth_out + i) = dout;
    *reinterpret_cast<V*>(conf_out + i) = cout;
    if constexpr (WITH_VAR) {
        float var[VEC];
#pragma unroll
        for (int k = 0; k < VEC; ++k) var[k] = 0.0f;
        for (int d = 0; d < D; ++d) {
            const V c = *reinterpret_cast<const V*>(cost + d * plane + i);
            V dv;
            if (depth_mode == D3D_DEPTH_PER_PIXEL) dv = *reinterpret_cast<const V*>(depth + d * plane + i);
#pragma unroll
            for (int k = 0; k < VEC; ++k) {
                const float e = __expf(vget<VEC>(c, k) - mx[k]);
                const float dvk = depth_mode == D3D_DEPTH_PER_PIXEL ? vget<VEC>(dv, k) : depth_at(depth, depth_mode, d, plane, i + k);
                const float t = dvk - vget<VEC>(dout, k);
                var[k] = fmaf(t * t, e * inv[k], var[k]);
            }
        }
        V vout;
#pragma unroll
        for (int k = 0; k < VEC; ++k) vset<VEC>(vout, k, lamb * sqrtf(var[k]));
        *reinterpret_cast<V*>(var_out + i) = vout;
    }
}

// The same for D <= DC with the cost column of the pixel held in registers: every plane is read ONCE, all D loads are in
// flight together, and the arithmetic per element -- max, then exp-sums in plane order, window, variance -- is exactly
// that of the streaming form above (the cascade depths 8 / 32 / 48 / 64 all take this path).
template <int DC, bool WITH_VAR>
__global__ __launch_bounds__(256) void softargmin_conf4_cached_kernel(const float* __restrict__ cost,
                                                                       const float* __restrict__ depth, int depth_mode,
                                                                       int D, long plane, float lamb,
                                                                       float* __restrict__ depth_out,
                                                                       float* __restrict__ conf_out,
                                                                       float* __restrict__ var_out) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= plane) return;
    float c[DC], dvv[DC];
#pragma unroll
    for (int d = 0; d < DC; ++d) {
        c[d] = d < D ? cost[d * plane + i] : -INFINITY;
        dvv[d] = d < D ? depth_at(depth, depth_mode, d, plane, i) : 0.0f;
    }
    float mx = -INFINITY;
#pragma unroll
    for (int d = 0; d < DC; ++d) mx = fmaxf(mx, c[d]);
    float den = 0.0f, dep = 0.0f, idx = 0.0f;
#pragma unroll
    for (int d = 0; d < DC; ++d) {
        if (d < D) {
            c[d] = __expf(c[d] - mx);
            den += c[d];
            dep = fmaf(c[d], dvv[d], dep);
            idx = fmaf(c[d], (float)d, idx);
        }
    }
    const float inv = 1.0f / den;
    int k = (int)(idx * inv);
    k = min(max(k, 0), D - 1);
    float conf = 0.0f;
#pragma unroll
    for (int d = 0; d < DC; ++d)    // planes k-1 .. k+2, in plane order (as the streaming form adds them)
        if (d < D && d >= k - 1 && d <= k + 2) conf += c[d];
    const float depth_v = dep * inv;
    depth_out[i] = depth_v;
    conf_out[i] = conf * inv;
    if constexpr (WITH_VAR) {
        float var = 0.0f;
#pragma unroll
        for (int d = 0; d < DC; ++d) {
            if (d < D) {
                const float t = dvv[d] - depth_v;
                var = fmaf(t * t, c[d] * inv, var);
            }
        }
        var_out[i] = lamb * sqrtf(var);
    }
}

template <bool WITH_VAR>
static bool launch_softargmin_cached(const float* cost, const float* depth, int depth_mode, int D, long plane, float lamb,
                                     float* depth_out, float* conf_out, float* var_out, hipStream_t stream) {
    const dim3 grid(ceil_div(plane, 256)), block(256);
#define D3D_SA_CASE(DC)                                                                                                      \
    if (D <= DC) {                                                                                                           \
        hipLaunchKernelGGL((softargmin_conf4_cached_kernel<DC, WITH_VAR>), grid, block, 0, stream, cost, depth, depth_mode, D, \
                           plane, lamb, depth_out, conf_out, var_out);                                                       \
        return true;                                                                                                         \
    }
    D3D_SA_CASE(8)
    D3D_SA_CASE(16)
    D3D_SA_CASE(32)
    D3D_SA_CASE(48)
    D3D_SA_CASE(64)
#undef D3D_SA_CASE
    return false;
}

// ucsnet.py:42-51 (uncertainty_aware_samples, later stages): per pixel D hypotheses low + step * i + 1e-12 between
// cur - var and cur + var, step = (high - low) / (D - 1).
template <int VEC>
__global__ __launch_bounds__(256) void uncertainty_samples_kernel(const float* __restrict__ cur, const float* __restrict__ var,
                                                                   int D, long plane, float* __restrict__ out) {
    typedef typename VecT<VEC>::type V;
    const long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * VEC;
    if (i >= plane) return;
    const V c = *reinterpret_cast<const V*>(cur + i), v = *reinterpret_cast<const V*>(var + i);
    float low[VEC], step[VEC];
#pragma unroll
    for (int k = 0; k < VEC; ++k) {
        low[k] = vget<VEC>(c, k) - vget<VEC>(v, k);
        const float high = vget<VEC>(c, k) + vget<VEC>(v, k);
        step[k] = (high - low[k]) / ((float)D - 1.0f);
    }
    for (int d = 0; d < D; ++d) {
        V o;
#pragma unroll
        for (int k = 0; k < VEC; ++k) vset<VEC>(o, k, __fadd_rn(__fadd_rn(low[k], __fmul_rn(step[k], (float)d)), 1e-12f));
        *reinterpret_cast<V*>(out + d * plane + i) = o;
    }
}

// adamvs.py:478-486: softmax over D, max prob, expected depth.
__global__ __launch_bounds__(256) void pair_softmax_max_kernel(const float* __restrict__ score,
                                                                const float* __restrict__ depth, int depth_mode,
                                                                int D, long plane, float* __restrict__ view_weight,
                                                                float* __restrict__ pair_depth) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= plane) return;
    float mx = -INFINITY;
    for (int d = 0; d < D; ++d) mx = fmaxf(mx, score[d * plane + i]);
    float den = 0.0f, dep = 0.0f;
    for (int d = 0; d < D; ++d) {
        float e = __expf(score[d * plane + i] - mx);
        float dv = depth_mode == D3D_DEPTH_PER_PIXEL ? depth[d * plane + i] : depth[d];
        den += e;
        dep = fmaf(e, dv, dep);
    }
    float inv = 1.0f / den;
    view_weight[i] = inv;  // max_d exp(s_d - mx)/den = 1/den
    pair_depth[i] = dep * inv;
}

// F.interpolate(bilinear, align_corners=False) source index / weights for one axis.
__device__ __forceinline__ void lin_coord(int dst, float scale, int in_size, int& i0, int& i1, float& l1) {
    float s = ((float)dst + 0.5f) * scale - 0.5f;
    s = s < 0.0f ? 0.0f : s;
    i0 = (int)s;
    i0 = min(i0, in_size - 1);
    i1 = i0 + (i0 < in_size - 1 ? 1 : 0);
    l1 = s - (float)i0;
}

__device__ __forceinline__ float bilerp(const float* __restrict__ in, int w, int y0, int y1, int x0, int x1,
                                        float ly, float lx) {
    float hy = 1.0f - ly, hx = 1.0f - lx;
    return hy * (hx * in[(long)y0 * w + x0] + lx * in[(long)y0 * w + x1]) +
           ly * (hx * in[(long)y1 * w + x0] + lx * in[(long)y1 * w + x1]);
}

template <int VEC>   // VEC consecutive output columns per thread (one 16-byte store when VEC = 4)
__global__ __launch_bounds__(256) void resize_bilinear_kernel(const float* __restrict__ in, int n, int h, int w,
                                                               int H, int W, float* __restrict__ out) {
    typedef typename VecT<VEC>::type V;
    const int X = (blockIdx.x * 64 + (threadIdx.x & 63)) * VEC;
    const int Y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (X >= W || Y >= H) return;
    int y0, y1, x0[VEC], x1[VEC];
    float ly, lx[VEC];
    lin_coord(Y, (float)h / (float)H, h, y0, y1, ly);
#pragma unroll
    for (int k = 0; k < VEC; ++k) lin_coord(X + k, (float)w / (float)W, w, x0[k], x1[k], lx[k]);
    for (int z = blockIdx.z; z < n; z += gridDim.z) {
        V o;
#pragma unroll
        for (int k = 0; k < VEC; ++k) vset<VEC>(o, k, bilerp(in + (long)z * h * w, w, y0, y1, x0[k], x1[k], ly, lx[k]));
        *reinterpret_cast<V*>(out + ((long)z * H + Y) * W + X) = o;
    }
}

// adamvs.py:514-525 with the optional depth-plane resample of adamvs.py:519-520 fused in.
__global__ __launch_bounds__(256) void online_regress_update_kernel(const float* __restrict__ reg,
                                                                     const float* __restrict__ dplane, int hd,
                                                                     int wd, int H, int W, float* __restrict__ max_p,
                                                                     float* __restrict__ sum_d,
                                                                     float* __restrict__ sum_p) {
    int X = blockIdx.x * 64 + (threadIdx.x & 63);
    int Y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (X >= W || Y >= H) return;
    long i = (long)Y * W + X;
    float dv;
    if (hd == H && wd == W) {
        dv = dplane[i];
    } else {
        int y0, y1, x0, x1;
        float ly, lx;
        lin_coord(Y, (float)hd / (float)H, hd, y0, y1, ly);
        lin_coord(X, (float)wd / (float)W, wd, x0, x1, lx);
        dv = bilerp(dplane, wd, y0, y1, x0, x1, ly, lx);
    }
    float p = __expf(reg[i]);
    max_p[i] = fmaxf(max_p[i], p);
    sum_d[i] = fmaf(dv, p, sum_d[i]);
    sum_p[i] = sum_p[i] + p;
}

// The head of a slice regulariser and the online regression in ONE streaming kernel (bf16 mode): adamvs.py:417-418 `reg =
// upconv2d(up)` -- ConvTranspose2d(8, 1, k 3, s 2, p 1, output_pad 1) at stages 1-2, Conv2d(8, 1, 3, pad 1) at stage 3, both with
// a bias -- followed by adamvs.py:514-525 (the kernel above).  As two launches the 8 -> 1 layer ran on the matrix-core tile
// kernel with 15 of its 16 output columns idle and four storing lanes per wave, and `reg` made a round trip through HBM; here
// a thread owns one input pixel (= 2 x 2 outputs) resp. one output pixel, the 72 weights are wave-uniform, operands are
// rounded to bf16 (RNE) as the matrix cores' are, products and sums are fp32.
__device__ __forceinline__ float bf16_round(float v) { return round_h16(v); }   // (the library's 16-bit format: common.h)
// four consecutive outputs of one row (X % 4 == 0, W % 4 == 0): the accumulators move as 16-byte accesses
__device__ __forceinline__ void regress_row4(const float (&reg)[4], long i, int X, int Y, const float* __restrict__ dplane, int hd, int wd,
                                             int H, int W, float* __restrict__ max_p, float* __restrict__ sum_d, float* __restrict__ sum_p) {
    typedef float f4r __attribute__((ext_vector_type(4)));
    f4r dv;
    if (hd == H && wd == W) {
        dv = *reinterpret_cast<const f4r*>(dplane + i);
    } else {
        int y0, y1;
        float ly;
        lin_coord(Y, (float)hd / (float)H, hd, y0, y1, ly);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            int x0, x1;
            float lx;
            lin_coord(X + k, (float)wd / (float)W, wd, x0, x1, lx);
            dv[k] = bilerp(dplane, wd, y0, y1, x0, x1, ly, lx);
        }
    }
    f4r mp = *reinterpret_cast<const f4r*>(max_p + i), sd = *reinterpret_cast<const f4r*>(sum_d + i), sp = *reinterpret_cast<const f4r*>(sum_p + i);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float p = __expf(reg[k]);
        mp[k] = fmaxf(mp[k], p);
        sd[k] = fmaf(dv[k], p, sd[k]);
        sp[k] = sp[k] + p;
    }
    *reinterpret_cast<f4r*>(max_p + i) = mp;
    *reinterpret_cast<f4r*>(sum_d + i) = sd;
    *reinterpret_cast<f4r*>(sum_p + i) = sp;
}

// wt: the 72 weights ALREADY rounded to bf16 (host: ops.slice_head_regress), [c][k_y][k_x]; wave-uniform scalar loads.
// TRANSPOSED: a thread owns the input pixels (y, 2 q), (y, 2 q + 1) = outputs rows 2 y, 2 y + 1 x columns 4 q .. 4 q + 3 (w % 2 == 0);
// otherwise the outputs (y, 4 q .. 4 q + 3) (w % 4 == 0).
template <bool TRANSPOSED>
__global__ __launch_bounds__(256) void slice_head_regress_kernel(const float* __restrict__ up, const float* __restrict__ wt, const float* __restrict__ bias,
                                                                  const float* __restrict__ dplane, int hd, int wd, int h, int w,
                                                                  float* __restrict__ max_p, float* __restrict__ sum_d, float* __restrict__ sum_p) {
    typedef float f4r __attribute__((ext_vector_type(4)));
    typedef float f2r __attribute__((ext_vector_type(2)));
    const int q = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    const long plane = (long)h * w;
    const float b0 = bias[0];
    if constexpr (TRANSPOSED) {
        // output 2 i + p per axis: p = 0 takes kernel tap 1 of input i; p = 1 takes tap 2 of input i and tap 0 of input i + 1
        const int x = 2 * q;
        if (x >= w || y >= h) return;
        const bool xin = x + 2 < w, yin = y + 1 < h;
        float e[4] = {0, 0, 0, 0}, o[4] = {0, 0, 0, 0};   // rows 2 y and 2 y + 1
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const float* __restrict__ r0 = up + c * plane + (long)y * w + x;
            const f2r a = *reinterpret_cast<const f2r*>(r0);
            const float a2 = xin ? r0[2] : 0.0f;
            const f2r bq = yin ? *reinterpret_cast<const f2r*>(r0 + w) : (f2r){0, 0};
            const float b2 = (xin && yin) ? r0[w + 2] : 0.0f;
            const float x0 = bf16_round(a[0]), x1 = bf16_round(a[1]), x2 = bf16_round(a2);
            const float z0 = bf16_round(bq[0]), z1 = bf16_round(bq[1]), z2 = bf16_round(b2);
            const float* __restrict__ k = wt + c * 9;   // [k_y][k_x]
            e[0] = fmaf(x0, k[4], e[0]);
            e[1] = fmaf(x1, k[3], fmaf(x0, k[5], e[1]));
            e[2] = fmaf(x1, k[4], e[2]);
            e[3] = fmaf(x2, k[3], fmaf(x1, k[5], e[3]));
            o[0] = fmaf(z0, k[1], fmaf(x0, k[7], o[0]));
            o[1] = fmaf(z1, k[0], fmaf(z0, k[2], fmaf(x1, k[6], fmaf(x0, k[8], o[1]))));
            o[2] = fmaf(z1, k[1], fmaf(x1, k[7], o[2]));
            o[3] = fmaf(z2, k[0], fmaf(z1, k[2], fmaf(x2, k[6], fmaf(x1, k[8], o[3]))));
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) { e[k] = e[k] * 1.0f + b0; o[k] = o[k] * 1.0f + b0; }
        const int H = 2 * h, W = 2 * w;
        const long i0 = (long)(2 * y) * W + 2 * x;
        regress_row4(e, i0, 2 * x, 2 * y, dplane, hd, wd, H, W, max_p, sum_d, sum_p);
        regress_row4(o, i0 + W, 2 * x, 2 * y + 1, dplane, hd, wd, H, W, max_p, sum_d, sum_p);
    } else {
        const int lane = threadIdx.x & 63;
        const bool live = 4 * q < w && y < h;   // (lanes past the row end stay for the shuffles; they read a valid quad and store nothing)
        const int x = live ? 4 * q : 0;
        if (y >= h) return;                      // (wave-uniform: a wave is one row)
        float o[4] = {0, 0, 0, 0};
        // four channels at a time: their twelve row quads are requested together, THEN shuffled and multiplied (a shuffle right
        // behind its load would wait for it: 24 memory latencies in a row per thread)
#pragma unroll
        for (int c0 = 0; c0 < 8; c0 += 4) {
            f4r v[4][3];
            float el[4][3], er[4][3];
#pragma unroll
            for (int cc = 0; cc < 4; ++cc)
#pragma unroll
                for (int ky = 0; ky < 3; ++ky) {
                    const int yy = y + ky - 1;
                    const bool rin = yy >= 0 && yy < h;
                    const float* __restrict__ r = up + (c0 + cc) * plane + (long)(rin ? yy : 0) * w + x;
                    v[cc][ky] = rin ? *reinterpret_cast<const f4r*>(r) : (f4r){0, 0, 0, 0};
                    // (the first / last lane of a wave has no neighbour lane: it loads the edge pixel itself)
                    el[cc][ky] = (lane == 0 && rin && x > 0) ? r[-1] : 0.0f;
                    er[cc][ky] = ((lane == 63 || x + 4 >= w) && rin && x + 4 < w) ? r[4] : 0.0f;
                }
#pragma unroll
            for (int cc = 0; cc < 4; ++cc)
#pragma unroll
                for (int ky = 0; ky < 3; ++ky) {
                    float lft = __shfl_up(v[cc][ky][3], 1), rgt = __shfl_down(v[cc][ky][0], 1);
                    if (lane == 0) lft = el[cc][ky];
                    if (lane == 63 || x + 4 >= w) rgt = er[cc][ky];
                    const float a[6] = {bf16_round(lft), bf16_round(v[cc][ky][0]), bf16_round(v[cc][ky][1]), bf16_round(v[cc][ky][2]),
                                        bf16_round(v[cc][ky][3]), bf16_round(rgt)};
                    const float* __restrict__ k = wt + (c0 + cc) * 9 + ky * 3;
#pragma unroll
                    for (int j = 0; j < 4; ++j) o[j] = fmaf(a[j + 2], k[2], fmaf(a[j + 1], k[1], fmaf(a[j], k[0], o[j])));
                }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) o[k] = o[k] * 1.0f + b0;
        if (live) regress_row4(o, (long)y * w + x, x, y, dplane, hd, wd, h, w, max_p, sum_d, sum_p);
    }
}

// ---------------------------------------------------------------------------------------------
// Tail of a depth slice of SliceCostRegNetRED at the stages whose head up-samples (adamvs.py:413-418, 423-425, 514-525), ONE
// kernel instead of two launches and a full-resolution 8-channel tensor:
//     up  = relu(upconv1(state2) + b_up + state1)          ConvTranspose2d(16 -> 8, k 3, stride 2, pad 1, output_pad 1)
//     reg = upconv2d(up) + b_head                           ConvTranspose2d(8 -> 1, k 3, stride 2, pad 1, output_pad 1)
//     max_p, sum_d, sum_p <- online regression update with exp(reg) at the slice's depth plane
// A workgroup (8 waves) takes 32 x 8 pixels of state2 per step: the 33 x 9 patch is staged as in conv2d_zs.hip's transposed
// kernel (buffer loads, out-of-range offsets outside the image, bf16 cells), the four parity classes run on
// v_mfma_f32_16x16x32_bf16 with the same packed weights (ops._pack_t2d_bf16) and the same epilogue expressions, and `up` --
// 64 x 16 pixels, rounded to bf16 exactly as the head rounds its operands -- goes into LDS instead of HBM (zeros outside the
// image).  Then a thread owns two neighbouring `up` pixels of the first 62 columns x 14 rows (the head reaches one pixel to the
// right and below: tiles step by 31 x 7 state2 pixels) and runs slice_head_regress_kernel<true>'s arithmetic in its order:
// the regression maps receive bit for bit what the two launches give them (tests/test_parity_gpu.py::
// test_slice_tail_fused_is_the_two_launches).
// ---------------------------------------------------------------------------------------------
namespace tail {
typedef float f4t __attribute__((ext_vector_type(4)));
typedef unsigned u4t __attribute__((ext_vector_type(4)));
constexpr int CI = 16, TXI = 32, TYI = 8, PXI = TXI + 1, PYI = TYI + 1, CS = 32;   // state2 tile, staged patch, bytes per cell
constexpr int NT = 64 * TYI;
constexpr int SX = TXI - 1, SY = TYI - 1;                    // tile step in state2 pixels
// SAME-resolution head (round 5: adamvs.py's last stage, msrednet.py:361-363 -- Conv2d(8, 1, 3, pad 1) on `up`): the head reaches one `up`
// pixel to every side, so the patch starts one state2 pixel up and to the left of the tile and a tile's outputs are the `up` pixels
// [60 bx, 60 bx + 60) x [14 by - 1, 14 by + 13) -- whole aligned quads per row (tiles step by 30 x 7 state2 pixels).
constexpr int SXS = 30;
constexpr int PATCH = PXI * PYI * CS;
constexpr int UW = 2 * TXI, UH = 2 * TYI;                    // `up` region in LDS: 16-byte cells (8 channels)
constexpr int UBYTES = UW * UH * 16;
constexpr int ntaps(int py, int px) { return (1 + py) * (1 + px); }
constexpr int nkb(int py, int px) { return (ntaps(py, px) * CI + 31) / 32; }
constexpr int frag_base(int c) { int s = 0; for (int q = 0; q < c; ++q) s += nkb(q >> 1, q & 1); return s; }
constexpr int NFRAG = frag_base(4);                          // 5
constexpr int LDS_BYTES = 2 * PATCH + UBYTES + NFRAG * 64 * 16;
constexpr unsigned OOB = 0xffffffffu;
struct Params {
    const float* s2;      // [16, h, w]
    const u4t* wup;       // [NFRAG][64] B fragments (ops._pack_t2d_bf16)
    const float* bup;     // [8]
    const float* s1;      // [8, 2h, 2w]: added before the ReLU
    const float* wh;      // [8][3][3] head weights, already rounded to bf16 (host)
    const float* bh;      // [1]
    const float* dplane;
    float *max_p, *sum_d, *sum_p;   // [4h, 4w] (transposed head) | [2h, 2w] (same-resolution head)
    int hd, wd, h, w, tper;
    int skip_after;       // 0: up = relu(upconv1 + b + s1) (adamvs.py:413-416); 1: up = relu(upconv1 + b) + s1 (module.py:287-294 ConvTransReLU + skip)
};
__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc(const void* origin) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(origin), 0, (int)0xfffffffeu, 0x00020000);
}
}  // namespace tail

template <bool TH>   // TH: the head is the stride-2 ConvTranspose2d (stages that up-sample); else Conv2d(8, 1, 3, pad 1) at `up`'s resolution
__global__ __launch_bounds__(tail::NT, 2) void slice_tail_kernel(tail::Params p) {
    using namespace tail;
    constexpr int STX = TH ? SX : SXS, ORG = TH ? 0 : 1;   // tile step along x; patch origin = tile origin - ORG state2 pixels
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* const ubuf = smem;                         // first: its cell offsets fit the DS instructions' immediates
    unsigned char* const pbuf = smem + UBYTES;
    u4t* const wlds = reinterpret_cast<u4t*>(pbuf + 2 * PATCH);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = p.h, w = p.w, H = 2 * h, W = 2 * w;
    const unsigned plane4 = (unsigned)h * w * 4, uplane4 = (unsigned)H * W * 4;   // (host: tensors < 2^31 bytes)
    const int ix0 = blockIdx.x * STX - ORG;
    const int nty = TH ? (h + SY - 1) / SY : (h + SY) / SY;   // (same-resolution head: the last tile's outputs end at `up` row 14 by + 12)
    const int t0 = blockIdx.y * p.tper, t1 = min(t0 + p.tper, nty);
    for (int i = tid; i < NFRAG * 64; i += NT) wlds[i] = p.wup[i];

    // ---- staging of the state2 patch: task = (patch pixel, 8-channel half) -> eight dword loads, one 16-byte slot ---------------
    constexpr int NTASK = PXI * PYI * 2, ROUNDS = (NTASK + NT - 1) / NT;
    float stg[ROUNDS][8];
    unsigned svo[ROUNDS];
    int spy[ROUNDS], scell[ROUNDS];
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {
        const int task = tid + r * NT;
        const int pix = task >> 1, g = task & 1;
        const int py = pix / PXI, px = pix - py * PXI;
        svo[r] = task < NTASK && ix0 + px >= 0 && ix0 + px < w ? (unsigned)(8 * g) * plane4 + (unsigned)(py * w + px) * 4 : OOB;
        spy[r] = py;
        scell[r] = task < NTASK ? pix * CS + g * 16 : -1;
    }
    auto issue = [&](int ty) {
        const int iy0 = ty * SY - ORG;
        const bool inner = iy0 >= 0 && iy0 + PYI <= h;
        const __amdgpu_buffer_rsrc_t rs = rsrc(p.s2 + ((long)iy0 * w + ix0));   // (may lie before the tensor: those lanes carry OOB offsets)
#pragma unroll
        for (int r = 0; r < ROUNDS; ++r) {
            unsigned vo = svo[r];
            if (!inner) vo = (unsigned)(spy[r] + iy0) < (unsigned)h ? vo : OOB;
#pragma unroll
            for (int k = 0; k < 8; ++k) stg[r][k] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, vo, k * plane4, 0));
        }
    };
    auto commit = [&](unsigned char* dst) {
#pragma unroll
        for (int r = 0; r < ROUNDS; ++r)
            if (scell[r] >= 0)
                *reinterpret_cast<u4t*>(dst + scell[r]) = (u4t){pack_h16x2(stg[r][0], stg[r][1]), pack_h16x2(stg[r][2], stg[r][3]),
                                                                 pack_h16x2(stg[r][4], stg[r][5]), pack_h16x2(stg[r][6], stg[r][7])};
    };

    // ---- upconv1 on the matrix cores: wave = state2 row, two 16-pixel groups; K offsets per (parity class, K block) once -------
    const int m = lane & 15, kg = lane >> 4;
    int aoff[NFRAG];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int kb = 0; kb < nkb(c >> 1, c & 1); ++kb) {
            const int px = c & 1;
            const int k0 = 32 * kb + 8 * kg, t = k0 / CI, ch = k0 % CI;
            const bool real = t < ntaps(c >> 1, px);
            const int dx = real ? t % (1 + px) : 0, dy = real ? t / (1 + px) : 0;
            aoff[frag_base(c) + kb] = ((wave + dy) * PXI + m + dx) * CS + (real ? ch : 0) * 2;
        }
    // D row = state2 pixel 4 kg + register, column = channel m: the lane's eight `up` pixels 2 (16 mg + 4 kg) .. + 7 of rows 2 wave + PY
    const bool chan = m < 8;
    const float bup = chan && p.bup ? p.bup[m] : 0.0f;
    // state1 / `up` of the lane's pixels in PAIRS (tiles step by 31 state2 pixels, so a lane's eight pixels may straddle the right
    // edge; W is even): byte offset of pair j from the tile origin of an `up` row pair, OOB where the pair does not exist
    typedef float f2t __attribute__((ext_vector_type(2)));
    unsigned kvo[2][4];
    int uwr[2];         // the first pixel's cell in the LDS region (row 2 wave): + PY * UW * 16, + j * 16 per pixel
#pragma unroll
    for (int mg = 0; mg < 2; ++mg) {
        const int xl = 2 * (16 * mg + 4 * kg);
#pragma unroll
        for (int j = 0; j < 4; ++j)
            kvo[mg][j] = chan && 2 * ix0 + xl + 2 * j >= 0 && 2 * ix0 + xl + 2 * j < W ? (unsigned)m * uplane4 + (unsigned)(2 * wave * W + xl + 2 * j) * 4 : OOB;
        uwr[mg] = ((2 * wave) * UW + xl) * 16 + m * 2;
    }
    // ---- head: thread = two `up` pixels (yy, 2 q), (yy, 2 q + 1) of the first 62 columns x 14 rows -----------------------------
    const int hq = TH ? tid % (SX) : tid % 15, hy = TH ? tid / SX : tid / 15;   // 31 pairs | 15 quads per row
    const bool htask = TH ? tid < SX * 2 * SY : tid < 15 * 2 * SY;                 // 31 x 14 = 434 | 15 x 14 = 210 tasks
    const int urd = TH ? (hy * UW + 2 * hq) * 16 : (hy * UW + 4 * hq + 1) * 16;     // same-resolution head: the cell left of the quad's first pixel, one row up
    const float bh = p.bh[0];

    auto tile = [&](int ty, const unsigned char* buf) {
        const int iy0 = ty * SY - ORG, Y0 = 2 * iy0;
        const bool rowin = (unsigned)(iy0 + wave) < (unsigned)h;   // this wave's state2 row (= two `up` rows) exists
        const __amdgpu_buffer_rsrc_t rk = rsrc(p.s1 + ((long)Y0 * W + 2 * ix0));
        f2t sk[2][2][4];                                      // [PY][mg][pair]: state1 of the lane's pixels, requested before the GEMMs
#pragma unroll
        for (int PY = 0; PY < 2; ++PY)
#pragma unroll
            for (int mg = 0; mg < 2; ++mg)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    sk[PY][mg][j] = __builtin_bit_cast(f2t, __builtin_amdgcn_raw_buffer_load_b64(rk, rowin ? kvo[mg][j] : OOB, PY * W * 4, 0));
        auto rows = [&](auto pyc) {
            constexpr int PY = decltype(pyc)::value;
            f4t acc[2][2];
#pragma unroll
            for (int px = 0; px < 2; ++px)
#pragma unroll
                for (int mg = 0; mg < 2; ++mg) acc[px][mg] = (f4t){0, 0, 0, 0};
#pragma unroll
            for (int px = 0; px < 2; ++px) {
                constexpr int dummy = 0; (void)dummy;
                const int NKB = nkb(PY, px), FB = frag_base(PY * 2 + px);
#pragma unroll
                for (int kb = 0; kb < 2; ++kb) {
                    if (kb >= NKB) break;
                    const h16x8 bw = __builtin_bit_cast(h16x8, wlds[(FB + kb) * 64 + lane]);
#pragma unroll
                    for (int mg = 0; mg < 2; ++mg) {
                        const h16x8 a = __builtin_bit_cast(h16x8, *reinterpret_cast<const u4t*>(buf + aoff[FB + kb] + mg * 16 * CS));
                        acc[px][mg] = mfma_h16(a, bw, acc[px][mg]);
                    }
                }
            }
#pragma unroll
            for (int mg = 0; mg < 2; ++mg) {
                const f4t e = acc[0][mg] * 1.0f + bup, od = acc[1][mg] * 1.0f + bup;
                f4t lo = {e[0], od[0], e[1], od[1]}, hi = {e[2], od[2], e[3], od[3]};
                const f4t slo = {sk[PY][mg][0][0], sk[PY][mg][0][1], sk[PY][mg][1][0], sk[PY][mg][1][1]};
                const f4t shi = {sk[PY][mg][2][0], sk[PY][mg][2][1], sk[PY][mg][3][0], sk[PY][mg][3][1]};
                if (!p.skip_after) { lo += slo; hi += shi; }
                lo = __builtin_elementwise_max(lo, (f4t){0, 0, 0, 0}); hi = __builtin_elementwise_max(hi, (f4t){0, 0, 0, 0});
                if (p.skip_after) { lo = slo + lo; hi = shi + hi; }   // (ConvTransReLU, then the skip: the tile kernel's `skip + y`)
                // zeros outside the image: what the head's bounds tests read there
                unsigned q0 = pack_h16x2(lo[0], lo[1]), q1 = pack_h16x2(lo[2], lo[3]), q2 = pack_h16x2(hi[0], hi[1]), q3 = pack_h16x2(hi[2], hi[3]);
                if (!rowin || kvo[mg][0] == OOB) q0 = 0;
                if (!rowin || kvo[mg][1] == OOB) q1 = 0;
                if (!rowin || kvo[mg][2] == OOB) q2 = 0;
                if (!rowin || kvo[mg][3] == OOB) q3 = 0;
                if (chan) {
                    unsigned char* dst = ubuf + uwr[mg] + PY * (UW * 16);
                    *reinterpret_cast<unsigned short*>(dst) = (unsigned short)q0;
                    *reinterpret_cast<unsigned short*>(dst + 16) = (unsigned short)(q0 >> 16);
                    *reinterpret_cast<unsigned short*>(dst + 32) = (unsigned short)q1;
                    *reinterpret_cast<unsigned short*>(dst + 48) = (unsigned short)(q1 >> 16);
                    *reinterpret_cast<unsigned short*>(dst + 64) = (unsigned short)q2;
                    *reinterpret_cast<unsigned short*>(dst + 80) = (unsigned short)(q2 >> 16);
                    *reinterpret_cast<unsigned short*>(dst + 96) = (unsigned short)q3;
                    *reinterpret_cast<unsigned short*>(dst + 112) = (unsigned short)(q3 >> 16);
                }
            }
        };
        rows(std::integral_constant<int, 0>{});
        rows(std::integral_constant<int, 1>{});
        lds_barrier();                                        // the `up` region is complete
        if constexpr (!TH) {
            if (htask) {
                // outputs (Y, X .. X + 3) with X = 2 ix0 + 2 + 4 hq (a multiple of 4), Y = Y0 + 1 + hy; slice_head_regress_kernel<false>'s
                // arithmetic in its order: channel, kernel row, then fma(right, k2, fma(mid, k1, fma(left, k0, o))).  (As 30 pairs per row
                // on 420 threads with 8-byte accumulator accesses the kernel is slower: 187 against 175 us at the last stage.)
                const int X = 2 * ix0 + 2 + 4 * hq, Y = Y0 + 1 + hy;
                if (X < W && Y >= 0 && Y < H) {
                    float o[4] = {0, 0, 0, 0};
                    u4t c[3][6];
#pragma unroll
                    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                        for (int cc = 0; cc < 6; ++cc) c[ky][cc] = *reinterpret_cast<const u4t*>(ubuf + urd + (ky * UW + cc) * 16);
#pragma unroll
                    for (int ch = 0; ch < 8; ++ch) {
#pragma unroll
                        for (int ky = 0; ky < 3; ++ky) {
                            float a[6];
#pragma unroll
                            for (int cc = 0; cc < 6; ++cc) {
                                const unsigned d = c[ky][cc][ch >> 1];
                                a[cc] = (ch & 1) ? h16_hi(d) : h16_lo(d);
                            }
                            const float* __restrict__ k = p.wh + ch * 9 + ky * 3;
#pragma unroll
                            for (int j = 0; j < 4; ++j) o[j] = fmaf(a[j + 2], k[2], fmaf(a[j + 1], k[1], fmaf(a[j], k[0], o[j])));
                        }
                    }
#pragma unroll
                    for (int k = 0; k < 4; ++k) o[k] = o[k] * 1.0f + bh;
                    regress_row4(o, (long)Y * W + X, X, Y, p.dplane, p.hd, p.wd, H, W, p.max_p, p.sum_d, p.sum_p);
                }
            }
            return;
        }
        if (htask) {
            const int X = 2 * ix0 + 2 * hq, Y = Y0 + hy;      // the task's first `up` pixel
            if (X < W && Y < H) {
                // output 2 i + p per axis: p = 0 takes kernel tap 1 of input i; p = 1 takes tap 2 of input i and tap 0 of input i + 1
                u4t c[2][3];
#pragma unroll
                for (int rr = 0; rr < 2; ++rr)
#pragma unroll
                    for (int cc = 0; cc < 3; ++cc) c[rr][cc] = *reinterpret_cast<const u4t*>(ubuf + urd + (rr * UW + cc) * 16);
                float e[4] = {0, 0, 0, 0}, o[4] = {0, 0, 0, 0};   // rows 2 Y and 2 Y + 1
#pragma unroll
                for (int ch = 0; ch < 8; ++ch) {
                    auto val = [&](const u4t& v) {
                        const unsigned d = v[ch >> 1];
                        return (ch & 1) ? h16_hi(d) : h16_lo(d);
                    };
                    const float x0 = val(c[0][0]), x1 = val(c[0][1]), x2 = val(c[0][2]);
                    const float z0 = val(c[1][0]), z1 = val(c[1][1]), z2 = val(c[1][2]);
                    const float* __restrict__ k = p.wh + ch * 9;   // [k_y][k_x]
                    e[0] = fmaf(x0, k[4], e[0]);
                    e[1] = fmaf(x1, k[3], fmaf(x0, k[5], e[1]));
                    e[2] = fmaf(x1, k[4], e[2]);
                    e[3] = fmaf(x2, k[3], fmaf(x1, k[5], e[3]));
                    o[0] = fmaf(z0, k[1], fmaf(x0, k[7], o[0]));
                    o[1] = fmaf(z1, k[0], fmaf(z0, k[2], fmaf(x1, k[6], fmaf(x0, k[8], o[1]))));
                    o[2] = fmaf(z1, k[1], fmaf(x1, k[7], o[2]));
                    o[3] = fmaf(z2, k[0], fmaf(z1, k[2], fmaf(x2, k[6], fmaf(x1, k[8], o[3]))));
                }
#pragma unroll
                for (int k = 0; k < 4; ++k) { e[k] = e[k] * 1.0f + bh; o[k] = o[k] * 1.0f + bh; }
                const int HH = 2 * H, WW = 2 * W;
                const long i0 = (long)(2 * Y) * WW + 2 * X;
                regress_row4(e, i0, 2 * X, 2 * Y, p.dplane, p.hd, p.wd, HH, WW, p.max_p, p.sum_d, p.sum_p);
                regress_row4(o, i0 + WW, 2 * X, 2 * Y + 1, p.dplane, p.hd, p.wd, HH, WW, p.max_p, p.sum_d, p.sum_p);
            }
        }
    };

    issue(t0);
    commit(pbuf);
    __syncthreads();
    int cur = 0;
    for (int ty = t0; ty < t1; ++ty) {
        const bool more = ty + 1 < t1;
        if (more) issue(ty + 1);
        tile(ty, pbuf + cur * PATCH);
        if (more) commit(pbuf + (cur ^ 1) * PATCH);
        lds_barrier();                                        // the `up` region is free, the next patch is there
        cur ^= 1;
    }
}

__global__ __launch_bounds__(256) void online_regress_finalize_kernel(const float* __restrict__ max_p,
                                                                       const float* __restrict__ sum_d,
                                                                       const float* __restrict__ sum_p, long n,
                                                                       float* __restrict__ depth_out,
                                                                       float* __restrict__ conf_out) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float e = sum_p[i] + 1e-10f;
    depth_out[i] = sum_d[i] / e;
    conf_out[i] = max_p[i] / e;
}

// module.py:616-650.
template <int VEC>
__global__ __launch_bounds__(256) void depth_samples_pixel_kernel(const float* __restrict__ cur, int D,
                                                                   float interval, long plane,
                                                                   float* __restrict__ out) {
    typedef typename VecT<VEC>::type V;
    const long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * VEC;
    if (i >= plane) return;
    const V c = *reinterpret_cast<const V*>(cur + i);
    const float half = (float)D / 2.0f * interval;
    float lo[VEC], step[VEC];
#pragma unroll
    for (int k = 0; k < VEC; ++k) {
        lo[k] = vget<VEC>(c, k) - half;
        const float hi = vget<VEC>(c, k) + half;
        step[k] = (hi - lo[k]) / (float)(D - 1);
    }
    for (int d = 0; d < D; ++d) {
        V o;
#pragma unroll
        for (int k = 0; k < VEC; ++k) vset<VEC>(o, k, __fadd_rn(lo[k], __fmul_rn((float)d, step[k])));
        *reinterpret_cast<V*>(out + d * plane + i) = o;
    }
}

// the two maps the D planes above are generated from (D3D_DEPTH_AFFINE): out[0] = lo, out[1] = step
__global__ __launch_bounds__(256) void depth_samples_affine_kernel(const float* __restrict__ cur, int D, float interval, long plane,
                                                                    float* __restrict__ out) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= plane) return;
    const float half = (float)D / 2.0f * interval;
    const float lo = cur[i] - half, hi = cur[i] + half;
    out[i] = lo;
    out[plane + i] = (hi - lo) / (float)(D - 1);
}

__global__ void depth_samples_plane_kernel(const float* __restrict__ minmax, int D, float* __restrict__ out) {
    int d = blockIdx.x * blockDim.x + threadIdx.x;
    if (d >= D) return;
    float lo = minmax[0], hi = minmax[1];
    float step = (hi - lo) / (float)(D - 1);
    out[d] = __fadd_rn(lo, __fmul_rn((float)d, step));
}

// module.py:24-51 gate math.
__global__ __launch_bounds__(256) void gru_gates_kernel(const float* __restrict__ gates,
                                                         const float* __restrict__ h, int Hc, long plane,
                                                         float* __restrict__ rh, float* __restrict__ u) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    long n = (long)Hc * plane;
    if (i >= n) return;
    float r = gru_sigmoid_as<false>(gates[i]);
    float uu = gru_sigmoid_as<false>(gates[n + i]);
    rh[i] = r * h[i];
    u[i] = uu;
}

__global__ __launch_bounds__(256) void gru_update_kernel(const float* __restrict__ u, const float* __restrict__ h,
                                                          const float* __restrict__ convc, long n,
                                                          float* __restrict__ h_out) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float uu = u[i];
    h_out[i] = uu * h[i] + (1.0f - uu) * gru_tanh_as<false>(convc[i]);
}

// ---------------------------------------------------------------------------------------------
// ConvGRUCell2 (module.py:53-99, the msrednet recurrent cell): the gate / candidate convolutions are each
// followed by nn.GroupNorm(1, C): mean and biased variance over ALL C*H*W elements, then a per-channel
// affine.  Stage 1 reduces sum and sum of squares (fp64 partials, one atomic pair per workgroup); the
// elementwise stages read the two sums, so nothing synchronises with the host.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gn_stats_kernel(const float* __restrict__ xall, long n, double* __restrict__ stats_all) {
    // blockIdx.y = group: x = n consecutive elements, its own (sum, sum of squares) pair
    const float* __restrict__ x = xall + (long)blockIdx.y * n;
    double* __restrict__ stats = stats_all + 2 * blockIdx.y;
    double s = 0.0, q = 0.0;
    const long stride = (long)gridDim.x * blockDim.x * 4;
    for (long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < n; i += stride) {
        if (i + 3 < n) {
            const float4 v = *reinterpret_cast<const float4*>(x + i);
            s += (double)v.x + (double)v.y + (double)v.z + (double)v.w;
            q += (double)v.x * v.x + (double)v.y * v.y + (double)v.z * v.z + (double)v.w * v.w;
        } else {
            for (long k = i; k < n; ++k) { s += x[k]; q += (double)x[k] * x[k]; }
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        s += __shfl_down(s, o);
        q += __shfl_down(q, o);
    }
    __shared__ double ws[4], wq[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) { ws[wave] = s; wq[wave] = q; }
    __syncthreads();
    if (threadIdx.x == 0) {
        atomicAdd(stats, ws[0] + ws[1] + ws[2] + ws[3]);
        atomicAdd(stats + 1, wq[0] + wq[1] + wq[2] + wq[3]);
    }
}

struct GnAffine {  // y = x*a + b with the group statistics folded in
    float mean, rstd;
};
__device__ __forceinline__ GnAffine gn_fold(const double* __restrict__ stats, long n, float eps) {
    const double m = stats[0] / (double)n;
    double var = stats[1] / (double)n - m * m;
    var = var < 0.0 ? 0.0 : var;
    GnAffine g;
    g.mean = (float)m;
    g.rstd = (float)(1.0 / sqrt(var + (double)eps));
    return g;
}

// gates [2Hc,plane] -> r = sigmoid(gn_r(gates[:Hc])), u = sigmoid(gn_u(gates[Hc:])); rh = r*h (module.py:71-82,85)
template <bool FAST>
__global__ __launch_bounds__(256) void gru2_gates_kernel(const float* __restrict__ gates, const double* __restrict__ st_r,
                                                          const double* __restrict__ st_u, const float* __restrict__ g_r,
                                                          const float* __restrict__ b_r, const float* __restrict__ g_u,
                                                          const float* __restrict__ b_u, const float* __restrict__ h,
                                                          int Hc, long plane, float eps, float* __restrict__ rh,
                                                          float* __restrict__ u) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long n = (long)Hc * plane;
    if (i >= n) return;
    const int c = (int)(i / plane);
    const GnAffine ar = gn_fold(st_r, n, eps), au = gn_fold(st_u, n, eps);
    const float rn = (gates[i] - ar.mean) * ar.rstd * g_r[c] + b_r[c];
    const float un = (gates[n + i] - au.mean) * au.rstd * g_u[c] + b_u[c];
    const float r = gru_sigmoid_as<FAST>(rn);
    rh[i] = r * h[i];
    u[i] = gru_sigmoid_as<FAST>(un);
}

// h' = u*h + (1-u)*tanh(gn_o(o))   (module.py:84-98)
template <bool FAST>
__global__ __launch_bounds__(256) void gru2_update_kernel(const float* __restrict__ o, const double* __restrict__ st_o,
                                                           const float* __restrict__ g_o, const float* __restrict__ b_o,
                                                           const float* __restrict__ u, const float* __restrict__ h,
                                                           int Hc, long plane, float eps, float* __restrict__ h_out) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long n = (long)Hc * plane;
    if (i >= n) return;
    const int c = (int)(i / plane);
    const GnAffine a = gn_fold(st_o, n, eps);
    const float on = (o[i] - a.mean) * a.rstd * g_o[c] + b_o[c];
    const float uu = u[i];
    h_out[i] = uu * h[i] + (1.0f - uu) * gru_tanh_as<FAST>(on);
}

// The two kernels above, four pixels of one channel per thread (plane % 4 == 0, 16-byte aligned tensors; grid.y = channel: no
// division), the folded statistics -- an fp64 division and a square root -- evaluated by ONE thread of the workgroup instead of
// every thread.  The same fp32 operations per element, in the same order.
typedef float f4g __attribute__((ext_vector_type(4)));
template <bool FAST>
__global__ __launch_bounds__(256) void gru2_gates_kernel4(const float* __restrict__ gates, const double* __restrict__ st_r,
                                                           const double* __restrict__ st_u, const float* __restrict__ g_r,
                                                           const float* __restrict__ b_r, const float* __restrict__ g_u,
                                                           const float* __restrict__ b_u, const float* __restrict__ h,
                                                           int Hc, long plane, float eps, float* __restrict__ rh,
                                                           float* __restrict__ u) {
    __shared__ float aff[4];
    const long n = (long)Hc * plane;
    if (threadIdx.x == 0) {
        const GnAffine ar = gn_fold(st_r, n, eps), au = gn_fold(st_u, n, eps);
        aff[0] = ar.mean; aff[1] = ar.rstd; aff[2] = au.mean; aff[3] = au.rstd;
    }
    __syncthreads();
    const long q = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (q >= plane) return;
    const int c = blockIdx.y;
    const long i = (long)c * plane + q;
    const float gr = g_r[c], br = b_r[c], gu = g_u[c], bu = b_u[c];
    const f4g vr = *reinterpret_cast<const f4g*>(gates + i), vu = *reinterpret_cast<const f4g*>(gates + n + i);
    const f4g hv = *reinterpret_cast<const f4g*>(h + i);
    f4g orh, ou;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float rn = (vr[k] - aff[0]) * aff[1] * gr + br;
        const float un = (vu[k] - aff[2]) * aff[3] * gu + bu;
        orh[k] = gru_sigmoid_as<FAST>(rn) * hv[k];
        ou[k] = gru_sigmoid_as<FAST>(un);
    }
    *reinterpret_cast<f4g*>(rh + i) = orh;
    *reinterpret_cast<f4g*>(u + i) = ou;
}

template <bool FAST>
__global__ __launch_bounds__(256) void gru2_update_kernel4(const float* __restrict__ o, const double* __restrict__ st_o,
                                                            const float* __restrict__ g_o, const float* __restrict__ b_o,
                                                            const float* __restrict__ u, const float* __restrict__ h,
                                                            int Hc, long plane, float eps, float* __restrict__ h_out) {
    __shared__ float aff[2];
    if (threadIdx.x == 0) {
        const GnAffine a = gn_fold(st_o, (long)Hc * plane, eps);
        aff[0] = a.mean; aff[1] = a.rstd;
    }
    __syncthreads();
    const long q = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (q >= plane) return;
    const int c = blockIdx.y;
    const long i = (long)c * plane + q;
    const float go = g_o[c], bo = b_o[c];
    const f4g ov = *reinterpret_cast<const f4g*>(o + i), uv = *reinterpret_cast<const f4g*>(u + i), hv = *reinterpret_cast<const f4g*>(h + i);
    f4g out;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float on = (ov[k] - aff[0]) * aff[1] * go + bo;
        out[k] = uv[k] * hv[k] + (1.0f - uv[k]) * gru_tanh_as<FAST>(on);
    }
    *reinterpret_cast<f4g*>(h_out + i) = out;
}


// Round 5: the cell in three passes over 104 channel planes instead of 128.  The reset half alone -- rh = sigmoid(gn_r(gates[:Hc])) * h
// -- and the update gate evaluated where it is used: h' = u*h + (1-u)*tanh(gn_o(o)) with u = sigmoid(gn_u(gates[Hc:])) computed from
// the pre-norm gate in the update pass (one plane read either way) instead of being written by one kernel and read by the next.
// Per element the fp32 operations of the two kernels above, in their order: the same bits.  A thread takes up to GRU2_F4 quads a
// workgroup-stride apart, so the folded statistics (an fp64 division and a square root on one thread, a barrier) are paid once
// per 4096 elements instead of once per 1024.
constexpr int GRU2_F4 = 4;
template <bool FAST>
__global__ __launch_bounds__(256) void gru2_reset_kernel4(const float* __restrict__ gates, const double* __restrict__ st_r,
                                                           const float* __restrict__ g_r, const float* __restrict__ b_r,
                                                           const float* __restrict__ h, int Hc, long plane, float eps,
                                                           float* __restrict__ rh) {
    __shared__ float aff[2];
    if (threadIdx.x == 0) {
        const GnAffine ar = gn_fold(st_r, (long)Hc * plane, eps);
        aff[0] = ar.mean; aff[1] = ar.rstd;
    }
    const int c = blockIdx.y;
    const float gr = g_r[c], br = b_r[c];
    const long q0 = ((long)blockIdx.x * (256 * GRU2_F4) + threadIdx.x) * 4;
    f4g vr[GRU2_F4], hv[GRU2_F4];
#pragma unroll
    for (int j = 0; j < GRU2_F4; ++j) {
        const long q = q0 + j * 1024;
        if (q < plane) {
            vr[j] = *reinterpret_cast<const f4g*>(gates + (long)c * plane + q);
            hv[j] = *reinterpret_cast<const f4g*>(h + (long)c * plane + q);
        }
    }
    __syncthreads();
    const float mean = aff[0], rstd = aff[1];
#pragma unroll
    for (int j = 0; j < GRU2_F4; ++j) {
        const long q = q0 + j * 1024;
        if (q < plane) {
            f4g orh;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float rn = (vr[j][k] - mean) * rstd * gr + br;
                orh[k] = gru_sigmoid_as<FAST>(rn) * hv[j][k];
            }
            *reinterpret_cast<f4g*>(rh + (long)c * plane + q) = orh;
        }
    }
}

template <bool FAST>
__global__ __launch_bounds__(256) void gru2_update_gates_kernel4(const float* __restrict__ o, const double* __restrict__ st_o,
                                                                  const float* __restrict__ g_o, const float* __restrict__ b_o,
                                                                  const float* __restrict__ gates_u, const double* __restrict__ st_u,
                                                                  const float* __restrict__ g_u, const float* __restrict__ b_u,
                                                                  const float* __restrict__ h, int Hc, long plane, float eps,
                                                                  float* __restrict__ h_out) {
    __shared__ float aff[4];
    if (threadIdx.x == 0) {
        const GnAffine a = gn_fold(st_o, (long)Hc * plane, eps), au = gn_fold(st_u, (long)Hc * plane, eps);
        aff[0] = a.mean; aff[1] = a.rstd; aff[2] = au.mean; aff[3] = au.rstd;
    }
    const int c = blockIdx.y;
    const float go = g_o[c], bo = b_o[c], gu = g_u[c], bu = b_u[c];
    const long q0 = ((long)blockIdx.x * (256 * GRU2_F4) + threadIdx.x) * 4;
    f4g ov[GRU2_F4], uv[GRU2_F4], hv[GRU2_F4];
#pragma unroll
    for (int j = 0; j < GRU2_F4; ++j) {
        const long q = q0 + j * 1024;
        if (q < plane) {
            ov[j] = *reinterpret_cast<const f4g*>(o + (long)c * plane + q);
            uv[j] = *reinterpret_cast<const f4g*>(gates_u + (long)c * plane + q);
            hv[j] = *reinterpret_cast<const f4g*>(h + (long)c * plane + q);
        }
    }
    __syncthreads();
    const float mo = aff[0], ro = aff[1], mu = aff[2], ru = aff[3];
#pragma unroll
    for (int j = 0; j < GRU2_F4; ++j) {
        const long q = q0 + j * 1024;
        if (q < plane) {
            f4g out;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float on = (ov[j][k] - mo) * ro * go + bo;
                const float un = (uv[j][k] - mu) * ru * gu + bu;
                const float uu = gru_sigmoid_as<FAST>(un);
                out[k] = uu * hv[j][k] + (1.0f - uu) * gru_tanh_as<FAST>(on);
            }
            *reinterpret_cast<f4g*>(h_out + (long)c * plane + q) = out;
        }
    }
}

}  // namespace d3d

using namespace d3d;

extern "C" {

int d3d_softargmin_conf4(const float* cost, const float* depth, int depth_mode, int D, int h, int w,
                         float* depth_out, float* conf_out, d3d_stream_t stream) {
    D3D_REQUIRE(cost && depth && depth_out && conf_out, "null pointer");
    D3D_REQUIRE(D > 0 && h > 0 && w > 0, "bad dims D=%d h=%d w=%d", D, h, w);
    D3D_REQUIRE(depth_mode >= 0 && depth_mode <= 2, "bad depth_mode %d", depth_mode);
    long plane = (long)h * w;
    if (launch_softargmin_cached<false>(cost, depth, depth_mode, D, plane, 0.0f, depth_out, conf_out, nullptr, (hipStream_t)stream)) {
    } else if (plane % 4 == 0 && aligned16(cost) && aligned16(depth) && aligned16(depth_out) && aligned16(conf_out))
        hipLaunchKernelGGL((softargmin_conf4_kernel<4, false>), dim3(ceil_div(plane / 4, 256)), dim3(256), 0, (hipStream_t)stream,
                           cost, depth, depth_mode, D, plane, 0.0f, depth_out, conf_out, (float*)nullptr);
    else
        hipLaunchKernelGGL((softargmin_conf4_kernel<1, false>), dim3(ceil_div(plane, 256)), dim3(256), 0, (hipStream_t)stream,
                           cost, depth, depth_mode, D, plane, 0.0f, depth_out, conf_out, (float*)nullptr);
    D3D_LAUNCH_CHECK("softargmin_conf4_kernel launch");
    return D3D_OK;
}

int d3d_softargmin_conf4_var(const float* cost, const float* depth, int depth_mode, int D, int h, int w, float lamb,
                             float* depth_out, float* conf_out, float* var_out, d3d_stream_t stream) {
    D3D_REQUIRE(cost && depth && depth_out && conf_out && var_out, "null pointer");
    D3D_REQUIRE(D > 0 && h > 0 && w > 0, "bad dims D=%d h=%d w=%d", D, h, w);
    D3D_REQUIRE(depth_mode >= 0 && depth_mode <= 2, "bad depth_mode %d", depth_mode);
    long plane = (long)h * w;
    if (launch_softargmin_cached<true>(cost, depth, depth_mode, D, plane, lamb, depth_out, conf_out, var_out, (hipStream_t)stream)) {
    } else if (plane % 4 == 0 && aligned16(cost) && aligned16(depth) && aligned16(depth_out) && aligned16(conf_out) && aligned16(var_out))
        hipLaunchKernelGGL((softargmin_conf4_kernel<4, true>), dim3(ceil_div(plane / 4, 256)), dim3(256), 0, (hipStream_t)stream,
                           cost, depth, depth_mode, D, plane, lamb, depth_out, conf_out, var_out);
    else
        hipLaunchKernelGGL((softargmin_conf4_kernel<1, true>), dim3(ceil_div(plane, 256)), dim3(256), 0, (hipStream_t)stream,
                           cost, depth, depth_mode, D, plane, lamb, depth_out, conf_out, var_out);
    D3D_LAUNCH_CHECK("softargmin_conf4_kernel (variance) launch");
    return D3D_OK;
}

int d3d_uncertainty_samples(const float* cur_depth, const float* exp_var, int D, int h, int w, float* out,
                            d3d_stream_t stream) {
    D3D_REQUIRE(cur_depth && exp_var && out, "null pointer");
    D3D_REQUIRE(D > 1 && h > 0 && w > 0, "need D > 1 and positive dims (D=%d h=%d w=%d)", D, h, w);
    long plane = (long)h * w;
    if (plane % 4 == 0 && aligned16(cur_depth) && aligned16(exp_var) && aligned16(out))
        hipLaunchKernelGGL(uncertainty_samples_kernel<4>, dim3(ceil_div(plane / 4, 256)), dim3(256), 0, (hipStream_t)stream,
                           cur_depth, exp_var, D, plane, out);
    else
        hipLaunchKernelGGL(uncertainty_samples_kernel<1>, dim3(ceil_div(plane, 256)), dim3(256), 0, (hipStream_t)stream,
                           cur_depth, exp_var, D, plane, out);
    D3D_LAUNCH_CHECK("uncertainty_samples_kernel launch");
    return D3D_OK;
}

int d3d_pair_softmax_max(const float* score, const float* depth, int depth_mode, int D, int h, int w,
                         float* view_weight, float* pair_depth, d3d_stream_t stream) {
    D3D_REQUIRE(score && depth && view_weight && pair_depth, "null pointer");
    D3D_REQUIRE(D > 0 && h > 0 && w > 0, "bad dims D=%d h=%d w=%d", D, h, w);
    D3D_REQUIRE(depth_mode == 0 || depth_mode == 1, "bad depth_mode %d", depth_mode);
    long plane = (long)h * w;
    hipLaunchKernelGGL(pair_softmax_max_kernel, dim3(ceil_div(plane, 256)), dim3(256), 0, (hipStream_t)stream, score,
                       depth, depth_mode, D, plane, view_weight, pair_depth);
    D3D_LAUNCH_CHECK("pair_softmax_max_kernel launch");
    return D3D_OK;
}

int d3d_resize_bilinear(const float* in, int n, int h, int w, int H, int W, float* out, d3d_stream_t stream) {
    D3D_REQUIRE(in && out, "null pointer");
    D3D_REQUIRE(n > 0 && h > 0 && w > 0 && H > 0 && W > 0, "bad dims n=%d %dx%d -> %dx%d", n, h, w, H, W);
    const bool vec = W % 4 == 0 && aligned16(out);
    dim3 grid(ceil_div(vec ? W / 4 : W, 64), ceil_div(H, 4), n < 64 ? n : 64);
    D3D_REQUIRE(grid.y <= 65535, "H=%d too large", H);
    if (vec) hipLaunchKernelGGL(resize_bilinear_kernel<4>, grid, dim3(256), 0, (hipStream_t)stream, in, n, h, w, H, W, out);
    else hipLaunchKernelGGL(resize_bilinear_kernel<1>, grid, dim3(256), 0, (hipStream_t)stream, in, n, h, w, H, W, out);
    D3D_LAUNCH_CHECK("resize_bilinear_kernel launch");
    return D3D_OK;
}

int d3d_online_regress_update(const float* reg, const float* dplane, int hd, int wd, int H, int W, float* max_p,
                              float* sum_d, float* sum_p, d3d_stream_t stream) {
    D3D_REQUIRE(reg && dplane && max_p && sum_d && sum_p, "null pointer");
    D3D_REQUIRE(hd > 0 && wd > 0 && H > 0 && W > 0, "bad dims %dx%d -> %dx%d", hd, wd, H, W);
    dim3 grid(ceil_div(W, 64), ceil_div(H, 4));
    D3D_REQUIRE(grid.y <= 65535, "H=%d too large", H);
    hipLaunchKernelGGL(online_regress_update_kernel, grid, dim3(256), 0, (hipStream_t)stream, reg, dplane, hd, wd, H,
                       W, max_p, sum_d, sum_p);
    D3D_LAUNCH_CHECK("online_regress_update_kernel launch");
    return D3D_OK;
}

int d3d_slice_head_regress_h16(const float* up, const float* weight, const float* bias, int transposed, const float* dplane, int hd,
                                int wd, int h, int w, float* max_p, float* sum_d, float* sum_p, d3d_stream_t stream) {
    D3D_REQUIRE(up && weight && bias && dplane && max_p && sum_d && sum_p, "null pointer");
    D3D_REQUIRE(hd > 0 && wd > 0 && h > 0 && w > 0, "bad dims %dx%d / %dx%d", hd, wd, h, w);
    // (16-byte accesses to the accumulators and, at the maps' own resolution, to the depth plane; 8- / 16-byte ones to `up`)
    const int Ho = transposed ? 2 * h : h, Wo = transposed ? 2 * w : w;
    const bool dquad = hd == Ho && wd == Wo;
    if (w % (transposed ? 2 : 4) != 0 ||
        ((reinterpret_cast<uintptr_t>(up) | reinterpret_cast<uintptr_t>(max_p) | reinterpret_cast<uintptr_t>(sum_d) | reinterpret_cast<uintptr_t>(sum_p) |
          (dquad ? reinterpret_cast<uintptr_t>(dplane) : 0)) & 15)) {
        set_error("d3d_slice_head_regress_h16: w = %d (a multiple of %d) and 16-byte aligned maps needed", w, transposed ? 2 : 4);
        return D3D_ERR_UNSUPPORTED;
    }
    dim3 grid(ceil_div(w / (transposed ? 2 : 4), 64), ceil_div(h, 4));
    D3D_REQUIRE(grid.y <= 65535, "h=%d too large", h);
    if (transposed)
        hipLaunchKernelGGL(slice_head_regress_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, up, weight, bias, dplane, hd, wd, h, w,
                           max_p, sum_d, sum_p);
    else
        hipLaunchKernelGGL(slice_head_regress_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, up, weight, bias, dplane, hd, wd, h, w,
                           max_p, sum_d, sum_p);
    D3D_LAUNCH_CHECK("slice_head_regress_kernel launch");
    return D3D_OK;
}

int d3d_slice_tail_regress_h16(const float* state2, const void* wup_packed, const float* bup, const float* state1, const float* whead,
                                const float* bhead, const float* dplane, int hd, int wd, int h, int w, float* max_p, float* sum_d,
                                float* sum_p, d3d_stream_t stream) {
    D3D_REQUIRE(state2 && wup_packed && bup && state1 && whead && bhead && dplane && max_p && sum_d && sum_p, "null pointer");
    D3D_REQUIRE(hd > 0 && wd > 0 && h > 0 && w > 0, "bad dims %dx%d / %dx%d", hd, wd, h, w);
    // (the depth plane is read in 16-byte quads only when it has the maps' resolution; a [1,1] plane is a 4-byte view of a table)
    const bool dquad = hd == 4 * h && wd == 4 * w;
    if (w % 4 != 0 || (long)h * w * 64 * 4 >= (1L << 31) ||
        ((reinterpret_cast<uintptr_t>(max_p) | reinterpret_cast<uintptr_t>(sum_d) | reinterpret_cast<uintptr_t>(sum_p) |
          (dquad ? reinterpret_cast<uintptr_t>(dplane) : 0)) & 15)) {
        set_error("d3d_slice_tail_regress_h16: w = %d (a multiple of 4), 16-byte aligned maps and tensors below 2 GiB needed", w);
        return D3D_ERR_UNSUPPORTED;
    }
    int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(slice_tail_kernel<true>), tail::LDS_BYTES);
    if (rc != D3D_OK) return rc;
    tail::Params p = {};
    p.s2 = state2; p.wup = reinterpret_cast<const tail::u4t*>(wup_packed); p.bup = bup; p.s1 = state1; p.wh = whead; p.bh = bhead;
    p.dplane = dplane; p.max_p = max_p; p.sum_d = sum_d; p.sum_p = sum_p; p.hd = hd; p.wd = wd; p.h = h; p.w = w;
    const int gx = ceil_div(w, tail::SX), nty = ceil_div(h, tail::SY);
    const int tper = pick_tper(gx, nty, tail::LDS_BYTES, tail::NFRAG * 64 * 16 + 8 * 1024, 2 * tail::PATCH);
    p.tper = tper;
    const int gy = ceil_div(nty, tper);
    if (gy > 65535) return D3D_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(slice_tail_kernel<true>, dim3(gx, gy), dim3(tail::NT), tail::LDS_BYTES, (hipStream_t)stream, p);
    D3D_LAUNCH_CHECK("slice_tail_kernel launch");
    return D3D_OK;
}

// The same tail with the head at `up`'s own resolution (adamvs.py:413-418 at the last stage; msrednet.py:361-363: upconv1 + skip,
// upconv2d, and the update of msrednet.py:418-437): up = relu(ConvTranspose2d_16->8(state2) + bup + state1) or, with
// skip_after_act, relu(ConvTranspose2d_16->8(state2) + bup) + state1 (module.py:287-294 ConvTransReLU followed by the skip);
// reg = Conv2d(8, 1, 3, pad 1)(up) + bhead; accumulators [2h, 2w].  bup may be null.  whead: the nn.Conv2d [1,8,3,3] weights
// [c][k_y][k_x] rounded to the 16-bit format.  Bit-identical to d3d_convtranspose2d_k3s2_zs_h16 followed by
// d3d_slice_head_regress_h16(transposed = 0).  w % 4 == 0, 16-byte aligned maps; D3D_ERR_UNSUPPORTED otherwise (nothing launched).
int d3d_slice_tail_regress_same_h16(const float* state2, const void* wup_packed, const float* bup, const float* state1, int skip_after_act,
                                    const float* whead, const float* bhead, const float* dplane, int hd, int wd, int h, int w,
                                    float* max_p, float* sum_d, float* sum_p, d3d_stream_t stream) {
    D3D_REQUIRE(state2 && wup_packed && state1 && whead && bhead && dplane && max_p && sum_d && sum_p, "null pointer");
    D3D_REQUIRE(hd > 0 && wd > 0 && h > 0 && w > 0, "bad dims %dx%d / %dx%d", hd, wd, h, w);
    const bool dquad = hd == 2 * h && wd == 2 * w;
    // (w % 4: as the two launches it replaces -- the transposed tile kernel takes rows of whole quads only, and narrower maps go through
    //  kernels that are not bit-identical to it)
    if (w % 4 != 0 || (long)h * w * 64 * 4 >= (1L << 31) ||
        ((reinterpret_cast<uintptr_t>(max_p) | reinterpret_cast<uintptr_t>(sum_d) | reinterpret_cast<uintptr_t>(sum_p) |
          (dquad ? reinterpret_cast<uintptr_t>(dplane) : 0) | reinterpret_cast<uintptr_t>(state1)) & 15)) {
        set_error("d3d_slice_tail_regress_same_h16: w = %d (a multiple of 4), 16-byte aligned maps and tensors below 2 GiB needed", w);
        return D3D_ERR_UNSUPPORTED;
    }
    int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(slice_tail_kernel<false>), tail::LDS_BYTES);
    if (rc != D3D_OK) return rc;
    tail::Params p = {};
    p.s2 = state2; p.wup = reinterpret_cast<const tail::u4t*>(wup_packed); p.bup = bup; p.s1 = state1; p.wh = whead; p.bh = bhead;
    p.dplane = dplane; p.max_p = max_p; p.sum_d = sum_d; p.sum_p = sum_p; p.hd = hd; p.wd = wd; p.h = h; p.w = w;
    p.skip_after = skip_after_act ? 1 : 0;
    const int gx = ceil_div(w, tail::SXS), nty = (h + tail::SY) / tail::SY;
    const int tper = pick_tper(gx, nty, tail::LDS_BYTES, tail::NFRAG * 64 * 16 + 8 * 1024, 2 * tail::PATCH);
    p.tper = tper;
    const int gy = ceil_div(nty, tper);
    if (gy > 65535) return D3D_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(slice_tail_kernel<false>, dim3(gx, gy), dim3(tail::NT), tail::LDS_BYTES, (hipStream_t)stream, p);
    D3D_LAUNCH_CHECK("slice_tail_kernel<same> launch");
    return D3D_OK;
}

int d3d_online_regress_finalize(const float* max_p, const float* sum_d, const float* sum_p, int64_t n,
                                float* depth_out, float* conf_out, d3d_stream_t stream) {
    D3D_REQUIRE(max_p && sum_d && sum_p && depth_out && conf_out, "null pointer");
    D3D_REQUIRE(n > 0, "bad n");
    hipLaunchKernelGGL(online_regress_finalize_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, (hipStream_t)stream,
                       max_p, sum_d, sum_p, (long)n, depth_out, conf_out);
    D3D_LAUNCH_CHECK("online_regress_finalize_kernel launch");
    return D3D_OK;
}

int d3d_depth_range_samples(const float* cur_depth, int mode, int D, float interval, int h, int w, float* out,
                            d3d_stream_t stream) {
    D3D_REQUIRE(cur_depth && out, "null pointer");
    D3D_REQUIRE(D > 1, "need D > 1 (got %d)", D);
    if (mode == D3D_DEPTH_PER_PLANE) {
        hipLaunchKernelGGL(depth_samples_plane_kernel, dim3(ceil_div(D, 64)), dim3(64), 0, (hipStream_t)stream,
                           cur_depth, D, out);
    } else if (mode == D3D_DEPTH_PER_PIXEL) {
        D3D_REQUIRE(h > 0 && w > 0, "bad dims h=%d w=%d", h, w);
        long plane = (long)h * w;
        if (plane % 4 == 0 && aligned16(cur_depth) && aligned16(out))
            hipLaunchKernelGGL(depth_samples_pixel_kernel<4>, dim3(ceil_div(plane / 4, 256)), dim3(256), 0,
                               (hipStream_t)stream, cur_depth, D, interval, plane, out);
        else
            hipLaunchKernelGGL(depth_samples_pixel_kernel<1>, dim3(ceil_div(plane, 256)), dim3(256), 0,
                               (hipStream_t)stream, cur_depth, D, interval, plane, out);
    } else if (mode == D3D_DEPTH_AFFINE) {
        D3D_REQUIRE(h > 0 && w > 0, "bad dims h=%d w=%d", h, w);
        const long plane = (long)h * w;
        hipLaunchKernelGGL(depth_samples_affine_kernel, dim3(ceil_div(plane, 256)), dim3(256), 0, (hipStream_t)stream, cur_depth, D,
                           interval, plane, out);
    } else {
        set_error("bad mode %d", mode);
        return D3D_ERR_INVALID_ARG;
    }
    D3D_LAUNCH_CHECK("depth_samples kernel launch");
    return D3D_OK;
}

int d3d_gru_gates(const float* gates, const float* h, int Hc, int64_t plane, float* rh, float* u,
                  d3d_stream_t stream) {
    D3D_REQUIRE(gates && h && rh && u, "null pointer");
    D3D_REQUIRE(Hc > 0 && plane > 0, "bad dims");
    long n = (long)Hc * plane;
    hipLaunchKernelGGL(gru_gates_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, (hipStream_t)stream, gates, h, Hc,
                       (long)plane, rh, u);
    D3D_LAUNCH_CHECK("gru_gates_kernel launch");
    return D3D_OK;
}

int d3d_gru_update(const float* u, const float* h, const float* convc, int64_t n, float* h_out,
                   d3d_stream_t stream) {
    D3D_REQUIRE(u && h && convc && h_out, "null pointer");
    D3D_REQUIRE(n > 0, "bad n");
    hipLaunchKernelGGL(gru_update_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, (hipStream_t)stream, u, h, convc,
                       (long)n, h_out);
    D3D_LAUNCH_CHECK("gru_update_kernel launch");
    return D3D_OK;
}

int d3d_groupnorm_stats(const float* x, int64_t n, int ngroups, double* stats, d3d_stream_t stream) {
    D3D_REQUIRE(x && stats, "null pointer");
    D3D_REQUIRE(n > 0 && ngroups > 0 && ngroups <= 65535, "bad n / ngroups");
    D3D_REQUIRE((reinterpret_cast<uintptr_t>(x) & 15) == 0 && (ngroups == 1 || n % 4 == 0),
                "x must be 16-byte aligned (every group)");
    int rc = hip_status(hipMemsetAsync(stats, 0, 2 * ngroups * sizeof(double), (hipStream_t)stream),
                        "hipMemsetAsync(stats)");
    if (rc != D3D_OK) return rc;
    long blocks = ceil_div(n, 256 * 4 * 8);
    blocks = blocks < 1 ? 1 : (blocks > 2048 ? 2048 : blocks);
    hipLaunchKernelGGL(gn_stats_kernel, dim3((unsigned)blocks, (unsigned)ngroups), dim3(256), 0, (hipStream_t)stream, x,
                       (long)n, stats);
    D3D_LAUNCH_CHECK("gn_stats_kernel launch");
    return D3D_OK;
}

int d3d_gru_gates_gn(const float* gates, const double* stats_r, const double* stats_u, const float* gamma_r,
                     const float* beta_r, const float* gamma_u, const float* beta_u, const float* h, int Hc,
                     int64_t plane, float eps, int fast, float* rh, float* u, d3d_stream_t stream) {
    D3D_REQUIRE(gates && stats_r && stats_u && gamma_r && beta_r && gamma_u && beta_u && h && rh && u, "null pointer");
    D3D_REQUIRE(Hc > 0 && plane > 0 && eps >= 0.0f, "bad dims");
    const long n = (long)Hc * plane;
    if (plane % 4 == 0 && Hc <= 65535 && aligned16(gates) && aligned16(h) && aligned16(rh) && aligned16(u))
        hipLaunchKernelGGL(fast ? gru2_gates_kernel4<true> : gru2_gates_kernel4<false>, dim3(ceil_div(plane / 4, 256), Hc), dim3(256), 0,
                           (hipStream_t)stream, gates, stats_r, stats_u, gamma_r, beta_r, gamma_u, beta_u, h, Hc, (long)plane, eps, rh, u);
    else
        hipLaunchKernelGGL(fast ? gru2_gates_kernel<true> : gru2_gates_kernel<false>, dim3(ceil_div(n, 256)), dim3(256), 0, (hipStream_t)stream,
                           gates, stats_r, stats_u, gamma_r, beta_r, gamma_u, beta_u, h, Hc, (long)plane, eps, rh, u);
    D3D_LAUNCH_CHECK("gru2_gates_kernel launch");
    return D3D_OK;
}

int d3d_gru_update_gn(const float* o, const double* stats_o, const float* gamma, const float* beta, const float* u,
                      const float* h, int Hc, int64_t plane, float eps, int fast, float* h_out, d3d_stream_t stream) {
    D3D_REQUIRE(o && stats_o && gamma && beta && u && h && h_out, "null pointer");
    D3D_REQUIRE(Hc > 0 && plane > 0 && eps >= 0.0f, "bad dims");
    const long n = (long)Hc * plane;
    if (plane % 4 == 0 && Hc <= 65535 && aligned16(o) && aligned16(u) && aligned16(h) && aligned16(h_out))
        hipLaunchKernelGGL(fast ? gru2_update_kernel4<true> : gru2_update_kernel4<false>, dim3(ceil_div(plane / 4, 256), Hc), dim3(256), 0,
                           (hipStream_t)stream, o, stats_o, gamma, beta, u, h, Hc, (long)plane, eps, h_out);
    else
        hipLaunchKernelGGL(fast ? gru2_update_kernel<true> : gru2_update_kernel<false>, dim3(ceil_div(n, 256)), dim3(256), 0, (hipStream_t)stream,
                           o, stats_o, gamma, beta, u, h, Hc, (long)plane, eps, h_out);
    D3D_LAUNCH_CHECK("gru2_update_kernel launch");
    return D3D_OK;
}

int d3d_gru_reset_gn(const float* gates, const double* stats_r, const float* gamma_r, const float* beta_r, const float* h, int Hc,
                     int64_t plane, float eps, int fast, float* rh, d3d_stream_t stream) {
    D3D_REQUIRE(gates && stats_r && gamma_r && beta_r && h && rh, "null pointer");
    D3D_REQUIRE(Hc > 0 && plane > 0 && eps >= 0.0f, "bad dims");
    if (plane % 4 != 0 || Hc > 65535 || !aligned16(gates) || !aligned16(h) || !aligned16(rh)) return D3D_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(fast ? gru2_reset_kernel4<true> : gru2_reset_kernel4<false>, dim3(ceil_div(plane / 4, 256 * GRU2_F4), Hc), dim3(256), 0,
                       (hipStream_t)stream, gates, stats_r, gamma_r, beta_r, h, Hc, (long)plane, eps, rh);
    D3D_LAUNCH_CHECK("gru2_reset_kernel4 launch");
    return D3D_OK;
}

int d3d_gru_update_gates_gn(const float* o, const double* stats_o, const float* gamma, const float* beta, const float* gates,
                            const double* stats_u, const float* gamma_u, const float* beta_u, const float* h, int Hc, int64_t plane,
                            float eps, int fast, float* h_out, d3d_stream_t stream) {
    D3D_REQUIRE(o && stats_o && gamma && beta && gates && stats_u && gamma_u && beta_u && h && h_out, "null pointer");
    D3D_REQUIRE(Hc > 0 && plane > 0 && eps >= 0.0f, "bad dims");
    const float* gates_u = gates + (long)Hc * plane;
    if (plane % 4 != 0 || Hc > 65535 || !aligned16(o) || !aligned16(gates_u) || !aligned16(h) || !aligned16(h_out)) return D3D_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(fast ? gru2_update_gates_kernel4<true> : gru2_update_gates_kernel4<false>, dim3(ceil_div(plane / 4, 256 * GRU2_F4), Hc),
                       dim3(256), 0, (hipStream_t)stream, o, stats_o, gamma, beta, gates_u, stats_u, gamma_u, beta_u, h, Hc, (long)plane, eps, h_out);
    D3D_LAUNCH_CHECK("gru2_update_gates_kernel4 launch");
    return D3D_OK;
}

// One ConvGRUCell2 step (module.py:53-99) as ONE call of the library: the four launches of the cell -- gate convolution with the
// GroupNorm statistics in its epilogue, the reset pass, candidate convolution with its statistics, the state update with the update
// gate evaluated in place -- issued back to back from C.  Same kernels, same operands as the four entry points called one by one
// (d3d_conv2d_k3_{zs|wide}_h16_gn, d3d_gru_reset_gn, d3d_gru_update_gates_gn); what goes away is the host's work between them: a
// RED-Net view is 352 cells, and issued from Python its four free-running level chains are paced by the host, not the card.
//   x [Cx,H,W], h [Hc,H,W] -> hout [Hc,H,W];  wg / wc = ops._pack_z2_bf16 of the gate [2Hc, Cx+Hc, 3, 3] / candidate [Hc, Cx+Hc, 3, 3]
//   weights; stats_g [2][2] and stats_o [2] fp64 ZEROED by the caller; gates [2Hc,H,W], rh [Hc,H,W], o [Hc,H,W]: scratch.
// Shapes: Cx + Hc = 16 | 24 | 32 | 40 (tile kernel: W % 4 == 0) or 64 | 128 (wide kernel: parts of 32); H * W % 4 == 0; 16-byte aligned
// tensors.  D3D_ERR_UNSUPPORTED otherwise, with nothing launched.
int d3d_gru2_cell_gn_h16(const float* x, int Cx, const float* h, int Hc, int H, int W, const void* wg, const float* bg, const void* wc,
                         const float* bc, const float* gamma_r, const float* beta_r, const float* gamma_u, const float* beta_u,
                         const float* gamma_o, const float* beta_o, float eps, int fast, double* stats_g, double* stats_o, float* gates,
                         float* rh, float* o, float* hout, d3d_stream_t stream) {
    D3D_REQUIRE(x && h && wg && bg && wc && bc && gamma_r && beta_r && gamma_u && beta_u && gamma_o && beta_o && stats_g && stats_o && gates &&
                rh && o && hout, "null pointer");
    D3D_REQUIRE(Cx > 0 && Hc > 0 && H > 0 && W > 0 && eps >= 0.0f, "bad dims");
    const int Ci = Cx + Hc;
    const long plane = (long)H * W;
    const bool tile = (Ci == 16 || Ci == 24 || Ci == 32 || Ci == 40) && Cx % 8 == 0 && Hc % 8 == 0 && W % 4 == 0 &&
                      2 * Hc <= (Ci == 24 || Ci == 40 ? 16 : 32);
    const bool wide = (Ci == 64 || Ci == 128) && Cx % 32 == 0 && Hc % 32 == 0 && (Hc == 32 || Hc == 64);
    const uintptr_t al = reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(h) | reinterpret_cast<uintptr_t>(gates) |
                         reinterpret_cast<uintptr_t>(rh) | reinterpret_cast<uintptr_t>(o) | reinterpret_cast<uintptr_t>(hout);
    if (!(tile || wide) || plane % 4 != 0 || Hc > 65535 || (al & 15) || (long)(Ci > 2 * Hc ? Ci : 2 * Hc) * plane * 4 >= (1L << 31)) {
        set_error("d3d_gru2_cell_gn_h16: %d + %d channels at %d x %d not taken", Cx, Hc, H, W);
        return D3D_ERR_UNSUPPORTED;
    }
    int rc = tile ? d3d_conv2d_k3_zs_h16_gn(x, Cx, h, Hc, wg, bg, 2 * Hc, H, W, gates, stats_g, Hc, stream)
                  : d3d_conv2d_k3_wide_h16_gn(x, Cx, h, Hc, wg, bg, 2 * Hc, H, W, gates, stats_g, Hc, stream);
    if (rc != D3D_OK) return rc;   // (shape rules checked above: an error here is a real one)
    rc = d3d_gru_reset_gn(gates, stats_g, gamma_r, beta_r, h, Hc, plane, eps, fast, rh, stream);
    if (rc != D3D_OK) return rc == D3D_ERR_UNSUPPORTED ? D3D_ERR_INVALID_ARG : rc;
    rc = tile ? d3d_conv2d_k3_zs_h16_gn(x, Cx, rh, Hc, wc, bc, Hc, H, W, o, stats_o, Hc, stream)
              : d3d_conv2d_k3_wide_h16_gn(x, Cx, rh, Hc, wc, bc, Hc, H, W, o, stats_o, Hc, stream);
    if (rc != D3D_OK) return rc == D3D_ERR_UNSUPPORTED ? D3D_ERR_INVALID_ARG : rc;
    rc = d3d_gru_update_gates_gn(o, stats_o, gamma_o, beta_o, gates, stats_g + 2, gamma_u, beta_u, h, Hc, plane, eps, fast, hout, stream);
    return rc == D3D_ERR_UNSUPPORTED ? D3D_ERR_INVALID_ARG : rc;
}

}  // extern "C"

