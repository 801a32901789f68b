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


// ---------------------------------------------------------------------------------------------------------------
// Tail of a reference view (fuse/fusion_3d_normal.py:545-570): the confirmed pixels become point-cloud vertices.
// The reference compacts with boolean indexing (row-major order of final_mask), then walks every skip_line-th valid
// point in a Python loop, keeps those inside the scene block in x and y, and lists the views that see the point.
// Here: two flag scans (valid pixels -> ordinal among the valid ones; kept points -> output slot) and one gather.
// Order and content of the output equal the reference's lists.
// ---------------------------------------------------------------------------------------------------------------
struct VisPlanes { const int* p[64]; };     // the visibility planes of a reference view, by value (<= 1 + fusion_num)
constexpr int SCAN_ITEMS = 16;              // flags per thread
constexpr int SCAN_BLOCK = 256 * SCAN_ITEMS;

// block totals of a byte-flag array
__global__ __launch_bounds__(256) void flag_count_kernel(const unsigned char* __restrict__ flags, long n, unsigned* __restrict__ block_sums) {
    __shared__ unsigned wsum[4];
    const long base = (long)blockIdx.x * SCAN_BLOCK + (long)threadIdx.x * SCAN_ITEMS;
    unsigned c = 0;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k) c += (base + k < n && flags[base + k]) ? 1u : 0u;
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) c += __shfl_xor(c, m);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) block_sums[blockIdx.x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}

// exclusive scan of the block totals in place (one workgroup; a 2752 x 1856 map has 1247 blocks); total -> *total
__global__ __launch_bounds__(256) void block_scan_kernel(unsigned* __restrict__ block_sums, int nblocks, unsigned* __restrict__ total) {
    __shared__ unsigned part[256];
    const int per = (nblocks + 255) / 256;
    const int b0 = threadIdx.x * per, b1 = min(b0 + per, nblocks);
    unsigned s = 0;
    for (int b = b0; b < b1; ++b) s += block_sums[b];
    part[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned run = 0;
        for (int i = 0; i < 256; ++i) { const unsigned t = part[i]; part[i] = run; run += t; }
        *total = run;
    }
    __syncthreads();
    unsigned run = part[threadIdx.x];
    for (int b = b0; b < b1; ++b) { const unsigned t = block_sums[b]; block_sums[b] = run; run += t; }
}

// exclusive prefix of this thread's first flag within the grid (block offset + in-block scan)
__device__ __forceinline__ unsigned thread_prefix(const unsigned char* __restrict__ flags, long n, const unsigned* __restrict__ block_offs,
                                                  unsigned (&mine)[SCAN_ITEMS]) {
    __shared__ unsigned wsum[4];
    const long base = (long)blockIdx.x * SCAN_BLOCK + (long)threadIdx.x * SCAN_ITEMS;
    unsigned c = 0;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k) { mine[k] = (base + k < n && flags[base + k]) ? 1u : 0u; c += mine[k]; }
    unsigned incl = c;
#pragma unroll
    for (int sft = 1; sft < 64; sft <<= 1) { const unsigned o = __shfl_up(incl, sft); if ((int)(threadIdx.x & 63) >= sft) incl += o; }
    if ((threadIdx.x & 63) == 63) wsum[threadIdx.x >> 6] = incl;
    __syncthreads();
    unsigned woff = 0;
    for (int wv = 0; wv < (int)(threadIdx.x >> 6); ++wv) woff += wsum[wv];
    return block_offs[blockIdx.x] + woff + incl - c;
}

// valid pixel with ordinal i (among the valid ones) is kept iff i % skip == 0 and it lies inside the block in x and y
__global__ __launch_bounds__(256) void points_mark_kernel(const unsigned char* __restrict__ final_mask, const float* __restrict__ avg_xyz,
                                                          long plane, const unsigned* __restrict__ block_offs, int skip, double x0,
                                                          double x1, double y0, double y1, unsigned char* __restrict__ keep) {
    unsigned mine[SCAN_ITEMS];
    unsigned ord = thread_prefix(final_mask, plane, block_offs, mine);
    const long base = (long)blockIdx.x * SCAN_BLOCK + (long)threadIdx.x * SCAN_ITEMS;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k) {
        const long i = base + k;
        if (i >= plane) break;
        unsigned char kp = 0;
        if (mine[k]) {
            const double x = (double)avg_xyz[i], y = (double)avg_xyz[plane + i];
            kp = (ord % (unsigned)skip == 0u) && (x0 < x) && (x < x1) && (y0 < y) && (y < y1);   // NaN fails, as in Python
            ++ord;
        }
        keep[i] = kp;
    }
}

__global__ __launch_bounds__(256) void points_gather_kernel(const unsigned char* __restrict__ keep, long plane,
                                                            const unsigned* __restrict__ block_offs, const float* __restrict__ avg_xyz,
                                                            const float* __restrict__ color, const float* __restrict__ normal,
                                                            VisPlanes vis, int n_vis, float* __restrict__ out_xyz,
                                                            int* __restrict__ out_color, float* __restrict__ out_normal,
                                                            int* __restrict__ out_views, int* __restrict__ out_nviews) {
    unsigned mine[SCAN_ITEMS];
    unsigned slot = thread_prefix(keep, plane, block_offs, mine);
    const long base = (long)blockIdx.x * SCAN_BLOCK + (long)threadIdx.x * SCAN_ITEMS;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k) {
        const long i = base + k;
        if (i >= plane || !mine[k]) continue;
        const size_t o = slot++;
#pragma unroll
        for (int c = 0; c < 3; ++c) out_xyz[o * 3 + c] = avg_xyz[(size_t)c * plane + i];
        if (color)
#pragma unroll
            for (int c = 0; c < 3; ++c) out_color[o * 3 + c] = (int)(color[(size_t)i * 3 + c] * 255.0f);   // (color * 255).astype(int)
        if (normal)
#pragma unroll
            for (int c = 0; c < 3; ++c) out_normal[o * 3 + c] = normal[(size_t)i * 3 + c];
        // views = sorted(vis[vis > 0] - 1): insertion into the output row (n_vis <= 1 + fusion_num, ~11)
        int nv = 0;
        int* row = out_views + o * n_vis;
        for (int v = 0; v < n_vis; ++v) {
            const int id = vis.p[v][i];
            if (id > 0) {
                int j = nv++;
                while (j > 0 && row[j - 1] > id - 1) { row[j] = row[j - 1]; --j; }
                row[j] = id - 1;
            }
        }
        for (int v = nv; v < n_vis; ++v) row[v] = -1;
        out_nviews[o] = nv;
    }
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

size_t d3d_fusion_points_scratch_bytes(int H, int W) {
    if (H <= 0 || W <= 0) return 0;
    const long blocks = ((long)H * W + SCAN_BLOCK - 1) / SCAN_BLOCK;
    return (size_t)(2 * blocks + 4) * sizeof(unsigned);
}

int d3d_fusion_mark_points(const float* avg_xyz_world, const unsigned char* final_mask, int H, int W, int skip_line,
                           const double* scene_range_xy, void* scratch, unsigned char* keep, unsigned* counts,
                           d3d_stream_t stream) {
    D3D_REQUIRE(avg_xyz_world && final_mask && scene_range_xy && scratch && keep && counts, "null pointer");
    D3D_REQUIRE(H > 0 && W > 0 && skip_line >= 1, "bad dims %dx%d / skip_line %d", H, W, skip_line);
    const long plane = (long)H * W;
    const int blocks = (int)((plane + SCAN_BLOCK - 1) / SCAN_BLOCK);
    unsigned* offs_valid = reinterpret_cast<unsigned*>(scratch);
    unsigned* offs_keep = offs_valid + blocks;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(flag_count_kernel, dim3(blocks), dim3(256), 0, st, final_mask, plane, offs_valid);
    hipLaunchKernelGGL(block_scan_kernel, dim3(1), dim3(256), 0, st, offs_valid, blocks, counts);
    hipLaunchKernelGGL(points_mark_kernel, dim3(blocks), dim3(256), 0, st, final_mask, avg_xyz_world, plane, offs_valid, skip_line,
                       scene_range_xy[0], scene_range_xy[1], scene_range_xy[2], scene_range_xy[3], keep);
    hipLaunchKernelGGL(flag_count_kernel, dim3(blocks), dim3(256), 0, st, keep, plane, offs_keep);
    hipLaunchKernelGGL(block_scan_kernel, dim3(1), dim3(256), 0, st, offs_keep, blocks, counts + 1);
    D3D_LAUNCH_CHECK("fusion point marking launch");
    return D3D_OK;
}

int d3d_fusion_gather_points(const float* avg_xyz_world, const unsigned char* keep, const int* const* vis, int n_vis,
                             const float* color, const float* normal_world, int H, int W, void* scratch, float* out_xyz,
                             int* out_color, float* out_normal, int* out_views, int* out_nviews, d3d_stream_t stream) {
    D3D_REQUIRE(avg_xyz_world && keep && vis && scratch && out_xyz && out_views && out_nviews, "null pointer");
    D3D_REQUIRE(!color || out_color, "colour input without output");
    D3D_REQUIRE(!normal_world || out_normal, "normal input without output");
    D3D_REQUIRE(H > 0 && W > 0 && n_vis >= 1 && n_vis <= 64, "bad dims %dx%d / %d visibility planes (max 64)", H, W, n_vis);
    const long plane = (long)H * W;
    const int blocks = (int)((plane + SCAN_BLOCK - 1) / SCAN_BLOCK);
    unsigned* offs_keep = reinterpret_cast<unsigned*>(scratch) + blocks;     // left by d3d_fusion_mark_points
    VisPlanes vp = {};
    for (int v = 0; v < n_vis; ++v) {
        D3D_REQUIRE(vis[v], "vis[%d] is null", v);
        vp.p[v] = vis[v];
    }
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(points_gather_kernel, dim3(blocks), dim3(256), 0, st, keep, plane, offs_keep, avg_xyz_world, color, normal_world,
                       vp, n_vis, out_xyz, out_color, out_normal, out_views, out_nviews);
    D3D_LAUNCH_CHECK("points_gather_kernel launch");
    return D3D_OK;
}

}  // extern "C"
