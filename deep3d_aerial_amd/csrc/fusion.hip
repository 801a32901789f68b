// Geometric consistency check between a reference and a source depth map, and the fusion accumulators that consume
// it.  Follows fuse/consistency_check_n.py:29-138 (ConsistencyChecker.check_cupy) and
// fuse/fusion_3d_normal.py:452-527 (the per-reference-view body of Fuse_Depth_Map.fuse_depths) of the reference.
//
// The reference runs this as ≈ 40 CuPy array operations per (ref, src) pair with an H2D copy of every input and
// a D2H copy of every output; here one pair is ONE launch, one lane per reference pixel, all maps resident in
// HBM: 20 B read + a 16 B gather + (fused form) 28 B read-modify-write per pixel, so it is a streaming kernel
// bound by HBM, and the arithmetic (≈ 150 fp64 + 40 fp32 operations per pixel) rides along.
//
// Numerics: the reference mixes precisions by NumPy/CuPy promotion rules and this file keeps every rounding point:
// the 3x3 / 4x4 camera matrices and their inverses are float32 (np.fromstring(dtype=float32), linalg.inv keeps the
// dtype) -- the host passes them widened to double, which is exact; the per-pixel chain is float64 because the
// integer pixel grid times the float32 depth promotes to float64; depth_reprojected, x/y_reprojected and the world
// point are rounded to float32 where the reference calls .astype(float32); the normal test is float32 throughout.
// Contraction is off (csrc/Makefile: -ffp-contract=off) so that products and sums round separately as in the
// oracle (oracle/fusion_oracle.c).
#include "common.h"

namespace d3d {

struct FusionCams {
    double Kri[9];   // inv(K_ref)
    double M1[12];   // (E_src @ inv(E_ref))[:3, :4]
    double Ks[9];    // K_src
    double Ksi[9];   // inv(K_src)
    double Esi[16];  // inv(E_src)
    double Er[12];   // E_ref[:3, :4]
    double Kr[9];    // K_ref
    float Rsi[9];    // inv(E_src[:3,:3])
    float Rri[9];    // inv(E_ref[:3,:3])
};

struct FusionArgs {
    const float* depth_ref;   // [H,W]
    const float* normal_ref;  // [H,W,3]
    const float* prob_ref;    // [H,W]
    const float* depth_src;   // [Hs,Ws]
    const float* normal_src;  // [Hs,Ws,3]
    int H, W, Hs, Ws;
    double pos_thr;
    float depth_thr, normal_thr, conf_thr;
    // pair outputs (consistency_check_n.py:138); any may be null
    unsigned char* mask;       // [H,W]
    float* depth_reprojected;  // [H,W]
    float* depth_src_out;      // [Hs,Ws], preset to depth_src by the caller; consistent samples are zeroed
    float* xyz_world_src;      // [3,H,W]
    float* angle_conf;         // [3,H,W]
    // fused accumulators (fusion_3d_normal.py:513-518); null in the plain check
    int* geo_mask_sum;    // [H,W]   += mask
    float* all_xyz_world;  // [3,H,W] += float32(angle * xyz_world_src)
    float* conf_sum;      // [H,W]   += angle   (the reference keeps three identical planes)
    int* vis;             // [H,W]   = mask * src_idx
    int src_idx;
};

__device__ __forceinline__ long trunc_i64(double v) {
    // float64 -> int64 as NumPy/CuPy .astype(int) does it on the hardware the reference runs on: truncation toward
    // zero; NaN, infinities and out-of-range values give the "integer indefinite" value INT64_MIN.
    return (fabs(v) < 9.2e18) ? (long)v : (-9223372036854775807L - 1);
}

__device__ __forceinline__ long wrap_index(long i, int n) {
    // CuPy integer-array indexing wraps out-of-range indices around (i mod n, sign of the divisor)
    if ((unsigned long)i < (unsigned long)n) return i;
    long r = i % n;
    return r < 0 ? r + n : r;
}

__device__ __forceinline__ void mat3(const double* __restrict__ m, double a, double b, double c, double& x, double& y,
                                     double& z) {
    x = m[0] * a + m[1] * b + m[2] * c;
    y = m[3] * a + m[4] * b + m[5] * c;
    z = m[6] * a + m[7] * b + m[8] * c;
}

__device__ __forceinline__ void mat3f(const float* __restrict__ m, float a, float b, float c, float& x, float& y,
                                      float& z) {
    x = m[0] * a + m[1] * b + m[2] * c;
    y = m[3] * a + m[4] * b + m[5] * c;
    z = m[6] * a + m[7] * b + m[8] * c;
}

template <bool FUSED>
__global__ __launch_bounds__(256) void consistency_kernel(const FusionCams c, const FusionArgs a) {
    const long plane = (long)a.H * a.W;
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= plane) return;
    const int y = (int)(idx / a.W), x = (int)(idx - (long)y * a.W);
    const float dref = a.depth_ref[idx];
    const double d = (double)dref;

    // reference pixel -> reference camera space -> source camera space -> source pixel (check_cupy:53-65)
    double rx, ry, rz;
    mat3(c.Kri, (double)x * d, (double)y * d, d, rx, ry, rz);
    const double sx = c.M1[0] * rx + c.M1[1] * ry + c.M1[2] * rz + c.M1[3];
    const double sy = c.M1[4] * rx + c.M1[5] * ry + c.M1[6] * rz + c.M1[7];
    const double sz = c.M1[8] * rx + c.M1[9] * ry + c.M1[10] * rz + c.M1[11];
    double kx, ky, kz;
    mat3(c.Ks, sx, sy, sz, kx, ky, kz);
    const long xi = trunc_i64(kx / kz + 0.5), yi = trunc_i64(ky / kz + 0.5);  // nearest pixel (check_cupy:70-71)
    const long xw = wrap_index(xi, a.Ws), yw = wrap_index(yi, a.Hs);
    const long sidx = yw * a.Ws + xw;
    const float dsrc = a.depth_src[sidx];
    const float n0 = a.normal_src[sidx * 3], n1 = a.normal_src[sidx * 3 + 1], n2 = a.normal_src[sidx * 3 + 2];

    // sampled source pixel -> source camera -> world -> reference camera -> reference pixel (check_cupy:77-91)
    const double ds = (double)dsrc;
    double bx, by, bz;
    mat3(c.Ksi, (double)xi * ds, (double)yi * ds, ds, bx, by, bz);
    double wv[4];
#pragma unroll
    for (int r = 0; r < 4; ++r)
        wv[r] = c.Esi[4 * r] * bx + c.Esi[4 * r + 1] * by + c.Esi[4 * r + 2] * bz + c.Esi[4 * r + 3];
    double px[3];
#pragma unroll
    for (int r = 0; r < 3; ++r)
        px[r] = c.Er[4 * r] * wv[0] + c.Er[4 * r + 1] * wv[1] + c.Er[4 * r + 2] * wv[2] + c.Er[4 * r + 3] * wv[3];
    const float drep = (float)px[2];
    double qx, qy, qz;
    mat3(c.Kr, px[0], px[1], px[2], qx, qy, qz);
    const float xr = (float)(qx / qz), yr = (float)(qy / qz);

    // position, relative depth, normal angle, confidence (check_cupy:94-119)
    const double ex = (double)xr - (double)x, ey = (double)yr - (double)y;
    const double dist = sqrt(ex * ex + ey * ey);
    const float rel = fabsf(drep - dref) / dref;
    float sw0, sw1, sw2, rw0, rw1, rw2;
    mat3f(c.Rsi, n0, n1, n2, sw0, sw1, sw2);
    const float* nr = a.normal_ref + idx * 3;
    mat3f(c.Rri, nr[0], nr[1], nr[2], rw0, rw1, rw2);
    float cs = rw0 * sw0 + rw1 * sw1 + rw2 * sw2;
    cs = cs / (sqrtf(rw0 * rw0 + rw1 * rw1 + rw2 * rw2) * sqrtf(sw0 * sw0 + sw1 * sw1 + sw2 * sw2));
    const bool m = (dist < a.pos_thr) & (rel < a.depth_thr) & (a.prob_ref[idx] > a.conf_thr) & (cs > a.normal_thr) &
                   (dref > 0.0f);
    const float ang = m ? (cs < 0.0f ? 0.0f : cs) : 0.0f;  // check_cupy:113, 133-136
    const float w0 = m ? (float)wv[0] : 0.0f, w1 = m ? (float)wv[1] : 0.0f, w2 = m ? (float)wv[2] : 0.0f;

    if (a.mask) a.mask[idx] = m ? 1 : 0;
    if (a.depth_reprojected) a.depth_reprojected[idx] = m ? drep : 0.0f;
    if (a.xyz_world_src) {
        a.xyz_world_src[idx] = w0;
        a.xyz_world_src[plane + idx] = w1;
        a.xyz_world_src[2 * plane + idx] = w2;
    }
    if (a.angle_conf) {
        a.angle_conf[idx] = ang;
        a.angle_conf[plane + idx] = ang;
        a.angle_conf[2 * plane + idx] = ang;
    }
    if (a.depth_src_out && m) {
        // check_cupy:123-126: the indices go through "+0.5, truncate" a second time (negative ones move up by one)
        const long xs2 = wrap_index(trunc_i64((double)xi + 0.5), a.Ws), ys2 = wrap_index(trunc_i64((double)yi + 0.5), a.Hs);
        a.depth_src_out[ys2 * a.Ws + xs2] = 0.0f;  // every writer stores the same value: no ordering needed
    }
    if constexpr (FUSED) {
        a.geo_mask_sum[idx] += m ? 1 : 0;
        a.all_xyz_world[idx] += ang * w0;
        a.all_xyz_world[plane + idx] += ang * w1;
        a.all_xyz_world[2 * plane + idx] += ang * w2;
        a.conf_sum[idx] += ang;
        if (a.vis) a.vis[idx] = m ? a.src_idx : 0;
    }
}

// fusion_3d_normal.py:452-474: the reference view's own world points, unit confidence, count 1, world normals
__global__ __launch_bounds__(256) void fusion_ref_init_kernel(const float* __restrict__ depth_ref,
                                                              const float* __restrict__ normal_ref, FusionCams c, int H,
                                                              int W, float* __restrict__ all_xyz_world,
                                                              float* __restrict__ conf_sum, int* __restrict__ geo_mask_sum,
                                                              float* __restrict__ normal_world) {
    const long plane = (long)H * W;
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= plane) return;
    const int y = (int)(idx / W), x = (int)(idx - (long)y * W);
    const double d = (double)depth_ref[idx];
    double rx, ry, rz;
    mat3(c.Kri, (double)x * d, (double)y * d, d, rx, ry, rz);
    // Esi holds inv(E_ref) here
#pragma unroll
    for (int r = 0; r < 3; ++r)
        all_xyz_world[r * plane + idx] =
            (float)(c.Esi[4 * r] * rx + c.Esi[4 * r + 1] * ry + c.Esi[4 * r + 2] * rz + c.Esi[4 * r + 3]);
    conf_sum[idx] = 1.0f;
    geo_mask_sum[idx] = 1;
    if (normal_world) {
        float w0, w1, w2;
        mat3f(c.Rri, normal_ref[idx * 3], normal_ref[idx * 3 + 1], normal_ref[idx * 3 + 2], w0, w1, w2);
        const float n = sqrtf(w0 * w0 + w1 * w1 + w2 * w2);
        normal_world[idx * 3] = w0 / n;
        normal_world[idx * 3 + 1] = w1 / n;
        normal_world[idx * 3 + 2] = w2 / n;
    }
}

// fusion_3d_normal.py:522-527: avg = all_xyz / confidence, final mask = count >= min_geo_consist_num
__global__ __launch_bounds__(256) void fusion_finalize_kernel(const float* __restrict__ all_xyz_world,
                                                              const float* __restrict__ conf_sum,
                                                              const int* __restrict__ geo_mask_sum, long plane,
                                                              int min_num, float* __restrict__ avg_xyz_world,
                                                              unsigned char* __restrict__ final_mask) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= plane) return;
    const float cf = conf_sum[idx];
#pragma unroll
    for (int r = 0; r < 3; ++r) avg_xyz_world[r * plane + idx] = all_xyz_world[r * plane + idx] / cf;
    final_mask[idx] = geo_mask_sum[idx] >= min_num ? 1 : 0;
}

static void fill_cams(FusionCams& c, const double* cam) {
    const double* p = cam;
    auto take = [&](double* dst, int n) {
        for (int i = 0; i < n; ++i) dst[i] = *p++;
    };
    take(c.Kri, 9);
    take(c.M1, 12);
    take(c.Ks, 9);
    take(c.Ksi, 9);
    take(c.Esi, 16);
    take(c.Er, 12);
    take(c.Kr, 9);
    for (int i = 0; i < 9; ++i) c.Rsi[i] = (float)*p++;
    for (int i = 0; i < 9; ++i) c.Rri[i] = (float)*p++;
}

}  // namespace d3d

using namespace d3d;

extern "C" {

int d3d_consistency_check(const float* depth_ref, const float* normal_ref, const float* prob_ref,
                          const float* depth_src, const float* normal_src, const double* cam, int H, int W, int Hs,
                          int Ws, double position_threshold, float depth_threshold, float normal_cos_threshold,
                          float confidence_threshold, unsigned char* mask, float* depth_reprojected,
                          float* depth_src_out, float* xyz_world_src, float* angle_conf, d3d_stream_t stream) {
    D3D_REQUIRE(depth_ref && normal_ref && prob_ref && depth_src && normal_src && cam, "null pointer");
    D3D_REQUIRE(H > 0 && W > 0 && Hs > 0 && Ws > 0, "bad dims %dx%d / %dx%d", H, W, Hs, Ws);
    D3D_REQUIRE(depth_src_out != depth_src, "depth_src_out must be a separate copy of depth_src");
    FusionCams c;
    fill_cams(c, cam);
    FusionArgs a = {};
    a.depth_ref = depth_ref, a.normal_ref = normal_ref, a.prob_ref = prob_ref, a.depth_src = depth_src,
    a.normal_src = normal_src;
    a.H = H, a.W = W, a.Hs = Hs, a.Ws = Ws;
    a.pos_thr = position_threshold, a.depth_thr = depth_threshold, a.normal_thr = normal_cos_threshold,
    a.conf_thr = confidence_threshold;
    a.mask = mask, a.depth_reprojected = depth_reprojected, a.depth_src_out = depth_src_out,
    a.xyz_world_src = xyz_world_src, a.angle_conf = angle_conf;
    hipLaunchKernelGGL(consistency_kernel<false>, dim3(ceil_div((long)H * W, 256)), dim3(256), 0, (hipStream_t)stream, c,
                       a);
    D3D_LAUNCH_CHECK("consistency_kernel launch");
    return D3D_OK;
}

int d3d_fusion_ref_init(const float* depth_ref, const float* normal_ref, const double* cam, int H, int W,
                        float* all_xyz_world, float* conf_sum, int* geo_mask_sum, float* normal_world,
                        d3d_stream_t stream) {
    D3D_REQUIRE(depth_ref && cam && all_xyz_world && conf_sum && geo_mask_sum, "null pointer");
    D3D_REQUIRE(!normal_world || normal_ref, "normal_world needs normal_ref");
    D3D_REQUIRE(H > 0 && W > 0, "bad dims %dx%d", H, W);
    FusionCams c;
    fill_cams(c, cam);
    hipLaunchKernelGGL(fusion_ref_init_kernel, dim3(ceil_div((long)H * W, 256)), dim3(256), 0, (hipStream_t)stream,
                       depth_ref, normal_ref, c, H, W, all_xyz_world, conf_sum, geo_mask_sum, normal_world);
    D3D_LAUNCH_CHECK("fusion_ref_init_kernel launch");
    return D3D_OK;
}

int d3d_fusion_accumulate(const float* depth_ref, const float* normal_ref, const float* prob_ref,
                          const float* depth_src, const float* normal_src, const double* cam, int H, int W, int Hs,
                          int Ws, double position_threshold, float depth_threshold, float normal_cos_threshold,
                          float confidence_threshold, int src_idx, int* geo_mask_sum, float* all_xyz_world,
                          float* conf_sum, int* vis, float* depth_src_out, d3d_stream_t stream) {
    D3D_REQUIRE(depth_ref && normal_ref && prob_ref && depth_src && normal_src && cam, "null pointer");
    D3D_REQUIRE(geo_mask_sum && all_xyz_world && conf_sum, "null accumulator");
    D3D_REQUIRE(H > 0 && W > 0 && Hs > 0 && Ws > 0, "bad dims %dx%d / %dx%d", H, W, Hs, Ws);
    D3D_REQUIRE(depth_src_out != depth_src, "depth_src_out must be a separate copy of depth_src");
    FusionCams c;
    fill_cams(c, cam);
    FusionArgs a = {};
    a.depth_ref = depth_ref, a.normal_ref = normal_ref, a.prob_ref = prob_ref, a.depth_src = depth_src,
    a.normal_src = normal_src;
    a.H = H, a.W = W, a.Hs = Hs, a.Ws = Ws;
    a.pos_thr = position_threshold, a.depth_thr = depth_threshold, a.normal_thr = normal_cos_threshold,
    a.conf_thr = confidence_threshold;
    a.depth_src_out = depth_src_out;
    a.geo_mask_sum = geo_mask_sum, a.all_xyz_world = all_xyz_world, a.conf_sum = conf_sum, a.vis = vis,
    a.src_idx = src_idx;
    hipLaunchKernelGGL(consistency_kernel<true>, dim3(ceil_div((long)H * W, 256)), dim3(256), 0, (hipStream_t)stream, c,
                       a);
    D3D_LAUNCH_CHECK("consistency_kernel (fused) launch");
    return D3D_OK;
}

int d3d_fusion_finalize(const float* all_xyz_world, const float* conf_sum, const int* geo_mask_sum, int H, int W,
                        int min_geo_consist_num, float* avg_xyz_world, unsigned char* final_mask, d3d_stream_t stream) {
    D3D_REQUIRE(all_xyz_world && conf_sum && geo_mask_sum && avg_xyz_world && final_mask, "null pointer");
    D3D_REQUIRE(H > 0 && W > 0, "bad dims %dx%d", H, W);
    const long plane = (long)H * W;
    hipLaunchKernelGGL(fusion_finalize_kernel, dim3(ceil_div(plane, 256)), dim3(256), 0, (hipStream_t)stream,
                       all_xyz_world, conf_sum, geo_mask_sum, plane, min_geo_consist_num, avg_xyz_world, final_mask);
    D3D_LAUNCH_CHECK("fusion_finalize_kernel launch");
    return D3D_OK;
}

}  // extern "C"
