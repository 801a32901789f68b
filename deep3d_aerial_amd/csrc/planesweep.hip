// Plane-sweep warp + aggregation kernels for gfx950 (MI355X), and the C-ABI entry points
// that launch them.  See include/deep3d_planesweep.h for the contract and the reference
// citations; DESIGN.md for the data layout and the roofline of each kernel.
//
// Two families live here:
//   * sweep_direct_kernel  -- one lane per reference pixel, taps gathered straight from
//     the planar [C,h,w] source maps through the vector L1/L2.  Handles any geometry,
//     any C, any V.  It is the general path and the reference point for the tiled one.
//   * sweep_tiled_kernel   -- (planesweep_tiled.hip) LDS-staged source footprints for
//     the HBM-write-bound cost-volume shapes.
#include <hip/hip_fp16.h>

#include "common.h"

#include <array>
#include <atomic>
#include <cstring>
#include <mutex>
#include <unordered_map>

namespace d3d {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int hip_status(hipError_t e, const char* what) {
    if (e == hipSuccess) return D3D_OK;
    set_error("%s: %s", what, hipGetErrorString(e));
    return D3D_ERR_HIP;
}

int ensure_dynamic_lds(const void* kernel, int bytes) {
    constexpr int MAXDEV = 16;
    static std::mutex mu;
    static std::unordered_map<const void*, std::array<int, MAXDEV>> granted;   // per-device cache, nothing else global
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) dev = -1;
    if (dev >= 0 && dev < MAXDEV) {
        std::lock_guard<std::mutex> lock(mu);
        auto it = granted.find(kernel);
        if (it != granted.end() && it->second[dev] >= bytes) return D3D_OK;
    }
    const int rc = hip_status(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes),
                              "hipFuncSetAttribute(MaxDynamicSharedMemorySize)");
    if (rc == D3D_OK && dev >= 0 && dev < MAXDEV) {
        std::lock_guard<std::mutex> lock(mu);
        auto& slot = granted[kernel];   // (value-initialised to zeros)
        if (slot[dev] < bytes) slot[dev] = bytes;
    }
    return rc;
}

// defined in planesweep_tiled.hip: returns D3D_ERR_UNSUPPORTED when the shape is outside
// what the tiled kernel handles, in which case the caller uses the direct kernel.
int launch_tiled(int mode, const struct SweepParams& p, hipStream_t stream);
// defined in planesweep_window.hip: the shallow sweeps of the cascades (one window per patch, two workgroups per CU);
// D3D_ERR_UNSUPPORTED outside its shapes
int launch_window(int mode, const struct SweepParams& p, hipStream_t stream, bool forced);
size_t tiled_workspace_bytes(int n_src, int C, int D, int h, int w, int elem_bytes);
size_t window_workspace_bytes(int n_src, int C, int D, int h, int w, int elem_bytes);
// non-default compile-time knobs of the translation units that have any (d3d_build_flags)
const char* tiled_build_flags();
const char* window_build_flags();
const char* gru_build_flags();

}  // namespace d3d

#include "sweep_params.h"

namespace d3d {

// ---------------------------------------------------------------------------------------
// module.py:528-530 on the device: fp64 cofactor inverse of ref_proj, product with each
// src_proj, one rounding to fp32.  One thread per source view.
// ---------------------------------------------------------------------------------------
__global__ void compose_kernel(const float* __restrict__ proj44, int n_views, float* __restrict__ out34) {
    int i = blockIdx.x * blockDim.x + threadIdx.x + 1;
    if (i >= n_views) return;
    double m[16], inv[16];
    for (int k = 0; k < 16; ++k) m[k] = (double)proj44[k];
    inv[0] = m[5] * m[10] * m[15] - m[5] * m[11] * m[14] - m[9] * m[6] * m[15] + m[9] * m[7] * m[14] +
             m[13] * m[6] * m[11] - m[13] * m[7] * m[10];
    inv[4] = -m[4] * m[10] * m[15] + m[4] * m[11] * m[14] + m[8] * m[6] * m[15] - m[8] * m[7] * m[14] -
             m[12] * m[6] * m[11] + m[12] * m[7] * m[10];
    inv[8] = m[4] * m[9] * m[15] - m[4] * m[11] * m[13] - m[8] * m[5] * m[15] + m[8] * m[7] * m[13] +
             m[12] * m[5] * m[11] - m[12] * m[7] * m[9];
    inv[12] = -m[4] * m[9] * m[14] + m[4] * m[10] * m[13] + m[8] * m[5] * m[14] - m[8] * m[6] * m[13] -
              m[12] * m[5] * m[10] + m[12] * m[6] * m[9];
    inv[1] = -m[1] * m[10] * m[15] + m[1] * m[11] * m[14] + m[9] * m[2] * m[15] - m[9] * m[3] * m[14] -
             m[13] * m[2] * m[11] + m[13] * m[3] * m[10];
    inv[5] = m[0] * m[10] * m[15] - m[0] * m[11] * m[14] - m[8] * m[2] * m[15] + m[8] * m[3] * m[14] +
             m[12] * m[2] * m[11] - m[12] * m[3] * m[10];
    inv[9] = -m[0] * m[9] * m[15] + m[0] * m[11] * m[13] + m[8] * m[1] * m[15] - m[8] * m[3] * m[13] -
             m[12] * m[1] * m[11] + m[12] * m[3] * m[9];
    inv[13] = m[0] * m[9] * m[14] - m[0] * m[10] * m[13] - m[8] * m[1] * m[14] + m[8] * m[2] * m[13] +
              m[12] * m[1] * m[10] - m[12] * m[2] * m[9];
    inv[2] = m[1] * m[6] * m[15] - m[1] * m[7] * m[14] - m[5] * m[2] * m[15] + m[5] * m[3] * m[14] +
             m[13] * m[2] * m[7] - m[13] * m[3] * m[6];
    inv[6] = -m[0] * m[6] * m[15] + m[0] * m[7] * m[14] + m[4] * m[2] * m[15] - m[4] * m[3] * m[14] -
             m[12] * m[2] * m[7] + m[12] * m[3] * m[6];
    inv[10] = m[0] * m[5] * m[15] - m[0] * m[7] * m[13] - m[4] * m[1] * m[15] + m[4] * m[3] * m[13] +
              m[12] * m[1] * m[7] - m[12] * m[3] * m[5];
    inv[14] = -m[0] * m[5] * m[14] + m[0] * m[6] * m[13] + m[4] * m[1] * m[14] - m[4] * m[2] * m[13] -
              m[12] * m[1] * m[6] + m[12] * m[2] * m[5];
    inv[3] = -m[1] * m[6] * m[11] + m[1] * m[7] * m[10] + m[5] * m[2] * m[11] - m[5] * m[3] * m[10] -
             m[9] * m[2] * m[7] + m[9] * m[3] * m[6];
    inv[7] = m[0] * m[6] * m[11] - m[0] * m[7] * m[10] - m[4] * m[2] * m[11] + m[4] * m[3] * m[10] +
             m[8] * m[2] * m[7] - m[8] * m[3] * m[6];
    inv[11] = -m[0] * m[5] * m[11] + m[0] * m[7] * m[9] + m[4] * m[1] * m[11] - m[4] * m[3] * m[9] -
              m[8] * m[1] * m[7] + m[8] * m[3] * m[5];
    inv[15] = m[0] * m[5] * m[10] - m[0] * m[6] * m[9] - m[4] * m[1] * m[10] + m[4] * m[2] * m[9] +
              m[8] * m[1] * m[6] - m[8] * m[2] * m[5];
    double det = m[0] * inv[0] + m[1] * inv[4] + m[2] * inv[8] + m[3] * inv[12];
    double idet = 1.0 / det;  // singular ref_proj -> inf/nan propagate, as torch.inverse would raise
    const float* s = proj44 + 16 * i;
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 4; ++c) {
            double acc = 0.0;
            for (int k = 0; k < 4; ++k) acc += (double)s[r * 4 + k] * (inv[k * 4 + c] * idet);
            out34[(i - 1) * 12 + r * 4 + c] = (float)acc;
        }
}

// ---------------------------------------------------------------------------------------
// Direct kernel.  Block = 64 x 4 reference pixels, each lane walks d_chunk planes.
// CC = channels kept in registers per pass.
// ---------------------------------------------------------------------------------------
struct TapD {
    int off;      // clamped north-west tap offset (y0c*w + x0c)
    int dx, dyw;  // 0/1 and 0/w steps to the east / south taps (clamped at the border)
    float nw, ne, sw, se;  // weights, zero for taps outside the image
};

__device__ __forceinline__ TapD make_tap_direct(float u, float v, int h, int w) {
    TapD t;
    float fu = floorf(u), fv = floorf(v);
    int x0 = (int)fu, y0 = (int)fv;
    float ax = u - fu, ay = v - fv;
    float bx = (fu + 1.0f) - u, by = (fv + 1.0f) - v;
    bool vx0 = (x0 >= 0) && (x0 < w), vx1 = (x0 >= -1) && (x0 < w - 1);
    bool vy0 = (y0 >= 0) && (y0 < h), vy1 = (y0 >= -1) && (y0 < h - 1);
    t.nw = (vx0 && vy0) ? bx * by : 0.0f;
    t.ne = (vx1 && vy0) ? ax * by : 0.0f;
    t.sw = (vx0 && vy1) ? bx * ay : 0.0f;
    t.se = (vx1 && vy1) ? ax * ay : 0.0f;
    int x0c = min(max(x0, 0), w - 1), x1c = min(max(x0 + 1, 0), w - 1);
    int y0c = min(max(y0, 0), h - 1), y1c = min(max(y0 + 1, 0), h - 1);
    t.off = y0c * w + x0c;
    t.dx = x1c - x0c;
    t.dyw = (y1c - y0c) * w;
    return t;
}

// element access of the two storage types: fp32, or fp16 storage with fp32 arithmetic (BASELINE config 5)
__device__ __forceinline__ float ldf(const float* p) { return *p; }
__device__ __forceinline__ float ldf(const __half* p) { return __half2float(*p); }
__device__ __forceinline__ void stf(float* p, float v) { *p = v; }
__device__ __forceinline__ void stf(__half* p, float v) { *p = __float2half_rn(v); }

template <typename T>
__device__ __forceinline__ float gather4(const T* __restrict__ f, const TapD& t) {
    const T* q = f + t.off;
    // same summation order as grid_sample: nw, ne, sw, se
    float acc = ldf(q) * t.nw;
    acc = fmaf(ldf(q + t.dx), t.ne, acc);
    acc = fmaf(ldf(q + t.dyw), t.sw, acc);
    acc = fmaf(ldf(q + t.dyw + t.dx), t.se, acc);
    return acc;
}

// T = float: tensors as declared in SweepParams.  T = __half: feats[] and out point to fp16 tensors of the same
// shapes (depth, weights and projections stay fp32); every product and sum is fp32, the result is rounded once.
template <int MODE, int CC, typename T = float>
__global__ __launch_bounds__(256) void sweep_direct_kernel(SweepParams p) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= p.w || y >= p.h) return;
    const int h = p.h, w = p.w, C = p.C, D = p.D;
    const size_t plane = (size_t)h * w;
    const size_t pix = (size_t)y * w + x;
    const float xf = (float)x, yf = (float)y;
    const int d0 = blockIdx.z * p.d_chunk;
    const int d1 = min(d0 + p.d_chunk, D);
    const float invV = 1.0f / (float)(p.n_src + 1);

    for (int d = d0; d < d1; ++d) {
        const float dv = p.depth_mode == D3D_DEPTH_PER_PIXEL ? p.depth[(size_t)d * plane + pix]
                         : p.depth_mode == D3D_DEPTH_AFFINE ? __fadd_rn(p.depth[pix], __fmul_rn((float)d, p.depth[plane + pix]))
                                                            : p.depth[d];
        float den = 1e-5f;
        if (MODE == MODE_WEIGHTED)
            for (int i = 0; i < p.n_src; ++i) den += p.weights[(size_t)i * plane + pix];
        float pair_acc = 0.0f;
        for (int c0 = 0; c0 < C; c0 += CC) {
            float s[CC], q[CC], r[CC];
#pragma unroll
            for (int c = 0; c < CC; ++c) {
                r[c] = (MODE == MODE_WARP) ? 0.0f : ldf(reinterpret_cast<const T*>(p.feats[0]) + (size_t)(c0 + c) * plane + pix);
                s[c] = (MODE == MODE_VARIANCE) ? r[c] : 0.0f;
                q[c] = r[c] * r[c];
            }
            for (int i = 0; i < p.n_src; ++i) {
                const float* __restrict__ P = p.proj34 + 12 * i;
                Ray ray = make_ray(P, xf, yf);
                float u, v;
                project(ray, P[3], P[7], P[11], dv, h, w, u, v);
                TapD t = make_tap_direct(u, v, h, w);
                const T* __restrict__ f = reinterpret_cast<const T*>(p.feats[i + 1]) + (size_t)c0 * plane;
                float vw = (MODE == MODE_WEIGHTED) ? p.weights[(size_t)i * plane + pix] : 0.0f;
#pragma unroll
                for (int c = 0; c < CC; ++c) {
                    float val = gather4(f + (size_t)c * plane, t);
                    if (MODE == MODE_VARIANCE) {
                        s[c] += val;
                        q[c] = fmaf(val, val, q[c]);
                    } else if (MODE == MODE_WEIGHTED) {
                        s[c] = fmaf(val * r[c], vw, s[c]);
                    } else if (MODE == MODE_PAIR) {
                        pair_acc = fmaf(r[c], val, pair_acc);
                    } else {
                        s[c] = val;
                    }
                }
            }
            if (MODE != MODE_PAIR) {
#pragma unroll
                for (int c = 0; c < CC; ++c) {
                    float o;
                    if (MODE == MODE_VARIANCE) {
                        float m = s[c] * invV;
                        o = fmaf(q[c], invV, -(m * m));
                    } else if (MODE == MODE_WEIGHTED) {
                        o = s[c] / den;
                    } else {
                        o = s[c];
                    }
                    stf(reinterpret_cast<T*>(p.out) + (p.plane_major ? (size_t)d * C + (c0 + c) : (size_t)(c0 + c) * D + d) * plane + pix, o);
                }
            }
        }
        if (MODE == MODE_PAIR) stf(reinterpret_cast<T*>(p.out) + (size_t)d * plane + pix, pair_acc / (float)C);
    }
}


// ---------------------------------------------------------------------------------------
// module.py:560-601 homo_warping_double: the same warp with the coordinate chain in fp64 -- rot @ [x,y,1], times
// depth, plus trans, the perspective divide and the normalisation to [-1,1] are double; the grid is rounded to fp32
// (`.float()`, module.py:590) and F.grid_sample un-normalises it in fp32.  Unused by the reference's models (it needs
// fp64 projection matrices, which its datasets never produce); built because SURVEY.md 8(a) lists it.  One lane per
// (pixel, plane), channels in a loop: a direct-gather kernel, not a tuned one.
// ---------------------------------------------------------------------------------------
__global__ void compose_f64_kernel(const double* __restrict__ proj44, int n_views, double* __restrict__ out34) {
    int i = blockIdx.x * blockDim.x + threadIdx.x + 1;
    if (i >= n_views) return;
    // Gauss-Jordan with partial pivoting on the reference view's 4x4 (fp64)
    double a[4][8];
    for (int r = 0; r < 4; ++r)
        for (int c = 0; c < 4; ++c) { a[r][c] = proj44[r * 4 + c]; a[r][4 + c] = r == c ? 1.0 : 0.0; }
    for (int col = 0; col < 4; ++col) {
        int piv = col;
        for (int r = col + 1; r < 4; ++r) if (fabs(a[r][col]) > fabs(a[piv][col])) piv = r;
        for (int c = 0; c < 8; ++c) { double t = a[col][c]; a[col][c] = a[piv][c]; a[piv][c] = t; }
        const double inv = 1.0 / a[col][col];
        for (int c = 0; c < 8; ++c) a[col][c] *= inv;
        for (int r = 0; r < 4; ++r) {
            if (r == col) continue;
            const double f = a[r][col];
            for (int c = 0; c < 8; ++c) a[r][c] -= f * a[col][c];
        }
    }
    const double* s = proj44 + 16 * i;
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 4; ++c) {
            double acc = 0.0;
            for (int k = 0; k < 4; ++k) acc += s[r * 4 + k] * a[k][4 + c];
            out34[(i - 1) * 12 + r * 4 + c] = acc;
        }
}

__global__ __launch_bounds__(256) void warp_f64coord_kernel(const float* __restrict__ src, const double* __restrict__ P,
                                                            const float* __restrict__ depth, int depth_mode, int C, int D,
                                                            int h, int w, float* __restrict__ out) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    const int d = blockIdx.z;
    if (x >= w || y >= h) return;
    const size_t plane = (size_t)h * w, pix = (size_t)y * w + x;
    const double dv = (double)(depth_mode == D3D_DEPTH_PER_PIXEL ? depth[(size_t)d * plane + pix] : depth[d]);
    const double xd = (double)x, yd = (double)y;
    // torch.matmul(rot, xyz): a 3-term dot product per row, accumulated left to right
    const double rx = P[0] * xd + P[1] * yd + P[2];
    const double ry = P[4] * xd + P[5] * yd + P[6];
    const double rz = P[8] * xd + P[9] * yd + P[10];
    const double px = rx * dv + P[3], py = ry * dv + P[7], pz = rz * dv + P[11];
    const float gx = (float)((px / pz) / ((double)(w - 1) / 2.0) - 1.0);
    const float gy = (float)((py / pz) / ((double)(h - 1) / 2.0) - 1.0);
    // grid_sampler_unnormalize, align_corners=True: ((coord + 1) / 2) * (size - 1), in fp32
    const float ix = ((gx + 1.0f) / 2.0f) * (float)(w - 1);
    const float iy = ((gy + 1.0f) / 2.0f) * (float)(h - 1);
    const bool ok = (ix > -2.0f) && (ix < (float)w + 1.0f) && (iy > -2.0f) && (iy < (float)h + 1.0f);  // also rejects NaN
    const TapD t = make_tap_direct(ok ? ix : -2.0f, ok ? iy : -2.0f, h, w);
    for (int c = 0; c < C; ++c)
        out[((size_t)c * D + d) * plane + pix] = gather4(src + (size_t)c * plane, t);
}

template <int MODE, typename T = float>
static int launch_direct(const SweepParams& p, hipStream_t stream) {
    SweepParams q = p;
    q.d_chunk = 8;
    dim3 grid(ceil_div(p.w, 64), ceil_div(p.h, 4), ceil_div(p.D, q.d_chunk));
    dim3 block(256);
    if (p.C % 16 == 0)
        hipLaunchKernelGGL((sweep_direct_kernel<MODE, 16, T>), grid, block, 0, stream, q);
    else if (p.C % 8 == 0)
        hipLaunchKernelGGL((sweep_direct_kernel<MODE, 8, T>), grid, block, 0, stream, q);
    else if (p.C % 4 == 0)
        hipLaunchKernelGGL((sweep_direct_kernel<MODE, 4, T>), grid, block, 0, stream, q);
    else
        hipLaunchKernelGGL((sweep_direct_kernel<MODE, 1, T>), grid, block, 0, stream, q);
    D3D_LAUNCH_CHECK("sweep_direct_kernel launch");
    return D3D_OK;
}

static int check_dims(int C, int D, int h, int w) {
    D3D_REQUIRE(C > 0 && D > 0 && h > 1 && w > 1, "bad dims C=%d D=%d h=%d w=%d (need C,D>0 and h,w>1)", C, D, h, w);
    D3D_REQUIRE((long)h * w < (1L << 30), "feature map too large: %d x %d", h, w);
    D3D_REQUIRE(ceil_div(h, 4) <= 65535 && ceil_div(D, 8) <= 65535, "grid too large (h=%d D=%d)", h, D);
    return D3D_OK;
}

// Test hook (d3d_debug_force_path): pins the kernel family so that the parity suite can run both on the same inputs.
// Every other experiment switch needs a -DD3D_EXPERIMENTS build; nothing here reads the environment.
static std::atomic<int> g_force_path{0};   // 0 dispatcher's choice | 1 direct-gather kernel | 2 LDS-ring kernel | 3 window kernel
static int forced_path() { return g_force_path.load(std::memory_order_relaxed); }

// Which kernel family served the sweep calls so far (d3d_debug_dispatch_counts): [1] direct gather, [2] LDS rings, [3] windows.
// The model-level parity tests read it to prove that the production kernels -- not a fallback -- produced what they compare.
static std::atomic<unsigned long long> g_dispatched[4];
static int counted(int family, int rc) {
    if (rc == D3D_OK) g_dispatched[family].fetch_add(1, std::memory_order_relaxed);
    return rc;
}

static int sweep_dispatch(int mode, const SweepParams& p, hipStream_t stream) {
    const int force = forced_path();
    if (force == 0 || force == 3) {
        int rc = counted(3, launch_window(mode, p, stream, force == 3));
        if (rc != D3D_ERR_UNSUPPORTED || force == 3) return rc;
    }
    if (force != 1) {
        int rc = counted(2, launch_tiled(mode, p, stream));
        if (rc != D3D_ERR_UNSUPPORTED) return rc;
        if (force == 2) return rc;
    }
    if (p.elem_bytes == 2) return counted(1, launch_direct<MODE_VARIANCE, __half>(p, stream));
    switch (mode) {
        case MODE_WARP: return counted(1, launch_direct<MODE_WARP>(p, stream));
        case MODE_VARIANCE: return counted(1, launch_direct<MODE_VARIANCE>(p, stream));
        case MODE_WEIGHTED: return counted(1, launch_direct<MODE_WEIGHTED>(p, stream));
        case MODE_PAIR: return counted(1, launch_direct<MODE_PAIR>(p, stream));
    }
    set_error("internal: bad mode %d", mode);
    return D3D_ERR_INVALID_ARG;
}

}  // namespace d3d

using namespace d3d;

extern "C" {

int d3d_version(void) { return D3D_ABI_VERSION; }

const char* d3d_last_error(void) { return g_err; }

int d3d_compose_projections(const float* proj44, int n_views, float* out34, d3d_stream_t stream) {
    D3D_REQUIRE(proj44 && out34, "null pointer");
    D3D_REQUIRE(n_views >= 2 && n_views <= D3D_MAX_VIEWS, "n_views=%d out of range [2,%d]", n_views, D3D_MAX_VIEWS);
    hipLaunchKernelGGL(compose_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, proj44, n_views, out34);
    D3D_LAUNCH_CHECK("compose_kernel launch");
    return D3D_OK;
}

int d3d_debug_dispatch_counts(unsigned long long* out4, int reset) {
    D3D_REQUIRE(out4, "null pointer");
    for (int i = 0; i < 4; ++i) out4[i] = reset ? g_dispatched[i].exchange(0, std::memory_order_relaxed) : g_dispatched[i].load(std::memory_order_relaxed);
    return D3D_OK;
}

const char* d3d_h16_format(void) { return D3D_H16_FORMAT; }

const char* d3d_build_flags(void) {
    static char buf[512];
    static std::atomic<int> done{0};
    if (!done.load(std::memory_order_acquire)) {
        char tmp[512];
        snprintf(tmp, sizeof(tmp), "%s%s%s", tiled_build_flags(), window_build_flags(), gru_build_flags());
        const char* s = tmp;
        while (*s == ' ') ++s;
        snprintf(buf, sizeof(buf), "%s", s);   // (idempotent: concurrent first calls write the same bytes)
        done.store(1, std::memory_order_release);
    }
    return buf;
}

int d3d_debug_force_path(int path) {
    D3D_REQUIRE(path >= 0 && path <= 3, "path must be 0 (auto), 1 (direct), 2 (tiled) or 3 (window)");
    g_force_path.store(path, std::memory_order_relaxed);
    return D3D_OK;
}

size_t d3d_sweep_workspace_bytes_for(int n_views, int C, int D, int h, int w, int elem_bytes, int depth_mode) {
    if (n_views < 2 || C <= 0 || D <= 0 || h <= 1 || w <= 1 || (elem_bytes != 4 && elem_bytes != 2)) return 0;
    const size_t a = tiled_workspace_bytes(n_views - 1, C, D, h, w, elem_bytes);
    // the window kernel's channel-last copy serves its gather path, which exists for [D,h,w] hypothesis volumes only
    const size_t b = depth_mode == D3D_DEPTH_PER_PIXEL ? window_workspace_bytes(n_views - 1, C, D, h, w, elem_bytes) : 0;
    return a > b ? a : b;
}

size_t d3d_sweep_workspace_bytes(int n_views, int C, int D, int h, int w, int elem_bytes) {
    return d3d_sweep_workspace_bytes_for(n_views, C, D, h, w, elem_bytes, D3D_DEPTH_PER_PIXEL);   // the largest any depth mode uses
}

int d3d_homo_warp(const float* src, const float* proj34, const float* depth, int depth_mode, int C, int D, int h,
                  int w, float* out, void* workspace, size_t workspace_bytes, d3d_stream_t stream) {
    D3D_REQUIRE(src && proj34 && depth && out, "null pointer");
    D3D_REQUIRE(depth_mode == 0 || depth_mode == 1, "bad depth_mode %d", depth_mode);
    int rc = check_dims(C, D, h, w);
    if (rc) return rc;
    SweepParams p = {};
    p.feats[0] = src;  // unused as reference in MODE_WARP
    p.feats[1] = src;
    p.proj34 = proj34;
    p.depth = depth;
    p.out = out;
    p.n_src = 1;
    p.C = C; p.D = D; p.h = h; p.w = w;
    p.depth_mode = depth_mode;
    p.elem_bytes = 4;
    p.workspace = workspace;
    p.workspace_bytes = workspace ? workspace_bytes : 0;
    return sweep_dispatch(MODE_WARP, p, (hipStream_t)stream);
}

int d3d_compose_projections_f64(const double* proj44, int n_views, double* out34, d3d_stream_t stream) {
    D3D_REQUIRE(proj44 && out34, "null pointer");
    D3D_REQUIRE(n_views >= 2 && n_views <= D3D_MAX_VIEWS, "n_views=%d out of range [2,%d]", n_views, D3D_MAX_VIEWS);
    hipLaunchKernelGGL(compose_f64_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, proj44, n_views, out34);
    D3D_LAUNCH_CHECK("compose_f64_kernel launch");
    return D3D_OK;
}

int d3d_homo_warp_f64coord(const float* src, const double* proj34, const float* depth, int depth_mode, int C, int D,
                           int h, int w, float* out, d3d_stream_t stream) {
    D3D_REQUIRE(src && proj34 && depth && out, "null pointer");
    D3D_REQUIRE(depth_mode == 0 || depth_mode == 1, "bad depth_mode %d", depth_mode);
    int rc = check_dims(C, D, h, w);
    if (rc) return rc;
    D3D_REQUIRE(D <= 65535, "D=%d too large for one launch", D);
    hipLaunchKernelGGL(warp_f64coord_kernel, dim3(ceil_div(w, 64), ceil_div(h, 4), D), dim3(256), 0, (hipStream_t)stream, src,
                       proj34, depth, depth_mode, C, D, h, w, out);
    D3D_LAUNCH_CHECK("warp_f64coord_kernel launch");
    return D3D_OK;
}

static int fill_multi(SweepParams& p, const float* const* feats, const float* proj34, const float* depth,
                      int depth_mode, int n_views, int C, int D, int h, int w, float* out, void* workspace,
                      size_t workspace_bytes, int elem_bytes = 4) {
    D3D_REQUIRE(feats && proj34 && depth && out, "null pointer");
    D3D_REQUIRE(n_views >= 2 && n_views <= D3D_MAX_VIEWS, "n_views=%d out of range [2,%d]", n_views, D3D_MAX_VIEWS);
    D3D_REQUIRE(depth_mode >= 0 && depth_mode <= 2, "bad depth_mode %d", depth_mode);
    int rc = check_dims(C, D, h, w);
    if (rc) return rc;
    for (int i = 0; i < n_views; ++i) {
        D3D_REQUIRE(feats[i], "feats[%d] is null", i);
        p.feats[i] = feats[i];
    }
    p.proj34 = proj34;
    p.depth = depth;
    p.out = out;
    p.n_src = n_views - 1;
    p.C = C; p.D = D; p.h = h; p.w = w;
    p.depth_mode = depth_mode;
    p.elem_bytes = elem_bytes;
    p.workspace = workspace;
    p.workspace_bytes = workspace ? workspace_bytes : 0;
    return D3D_OK;
}

int d3d_variance_volume(const float* const* feats, const float* proj34, const float* depth, int depth_mode,
                        int n_views, int C, int D, int h, int w, float* out, void* workspace, size_t workspace_bytes,
                        d3d_stream_t stream) {
    SweepParams p = {};
    int rc = fill_multi(p, feats, proj34, depth, depth_mode, n_views, C, D, h, w, out, workspace, workspace_bytes);
    if (rc) return rc;
    return sweep_dispatch(MODE_VARIANCE, p, (hipStream_t)stream);
}

// The same volume with plane d as one contiguous [C,h,w] block -- out [D,C,h,w] -- for the models that walk the depth slices one at a
// time (msrednet.py:400-437: the recurrent regulariser then reads a slice where it lies, no per-slice copy).
int d3d_variance_volume_planes(const float* const* feats, const float* proj34, const float* depth, int depth_mode,
                               int n_views, int C, int D, int h, int w, float* out, void* workspace, size_t workspace_bytes,
                               d3d_stream_t stream) {
    SweepParams p = {};
    int rc = fill_multi(p, feats, proj34, depth, depth_mode, n_views, C, D, h, w, out, workspace, workspace_bytes);
    if (rc) return rc;
    p.plane_major = 1;
    return sweep_dispatch(MODE_VARIANCE, p, (hipStream_t)stream);
}

static int variance_volume_cl_any(int layout, const float* const* feats, const float* proj34, const float* depth, int depth_mode,
                                int n_views, int C, int D, int h, int w, void* out, void* workspace, size_t workspace_bytes,
                                d3d_stream_t stream) {
    SweepParams p = {};
    int rc = fill_multi(p, feats, proj34, depth, depth_mode, n_views, C, D, h, w, reinterpret_cast<float*>(out), workspace,
                        workspace_bytes);
    if (rc) return rc;
    p.out_cl = layout;
    if (forced_path() == 1) {
        set_error("d3d_variance_volume_cl_h16: only the LDS-ring and window kernels write channel-last volumes");
        return D3D_ERR_UNSUPPORTED;
    }
    if (forced_path() != 2) {
        rc = counted(3, launch_window(MODE_VARIANCE, p, (hipStream_t)stream, forced_path() == 3));
        if (rc != D3D_ERR_UNSUPPORTED || forced_path() == 3) return rc;
    }
    return counted(2, launch_tiled(MODE_VARIANCE, p, (hipStream_t)stream));
}

int d3d_variance_volume_cl_h16(const float* const* feats, const float* proj34, const float* depth, int depth_mode,
                                int n_views, int C, int D, int h, int w, void* out, void* workspace, size_t workspace_bytes,
                                d3d_stream_t stream) {
    return variance_volume_cl_any(1, feats, proj34, depth, depth_mode, n_views, C, D, h, w, out, workspace, workspace_bytes, stream);
}

int d3d_variance_volume_cl8_h16(const float* const* feats, const float* proj34, const float* depth, int depth_mode,
                                 int n_views, int C, int D, int h, int w, void* out, void* workspace, size_t workspace_bytes,
                                 d3d_stream_t stream) {
    return variance_volume_cl_any(2, feats, proj34, depth, depth_mode, n_views, C, D, h, w, out, workspace, workspace_bytes, stream);
}

int d3d_variance_volume_f16(const void* const* feats, const float* proj34, const float* depth, int depth_mode,
                            int n_views, int C, int D, int h, int w, void* out, void* workspace, size_t workspace_bytes,
                            d3d_stream_t stream) {
    SweepParams p = {};
    // same argument checks; the pointers are carried in the fp32-typed slots and reinterpreted by the kernels
    int rc = fill_multi(p, reinterpret_cast<const float* const*>(feats), proj34, depth, depth_mode, n_views, C, D, h, w,
                        reinterpret_cast<float*>(out), workspace, workspace_bytes, 2);
    if (rc) return rc;
    return sweep_dispatch(MODE_VARIANCE, p, (hipStream_t)stream);
}

int d3d_weighted_corr(const float* const* feats, const float* proj34, const float* weights, const float* depth,
                      int depth_mode, int n_views, int C, int D, int h, int w, int plane_major, float* out, void* workspace,
                      size_t workspace_bytes, d3d_stream_t stream) {
    SweepParams p = {};
    D3D_REQUIRE(weights, "null weights");
    D3D_REQUIRE(plane_major == 0 || plane_major == 1, "bad plane_major %d", plane_major);
    int rc = fill_multi(p, feats, proj34, depth, depth_mode, n_views, C, D, h, w, out, workspace, workspace_bytes);
    if (rc) return rc;
    p.weights = weights;
    p.plane_major = plane_major;
    return sweep_dispatch(MODE_WEIGHTED, p, (hipStream_t)stream);
}

// adamvs.py:492-509 with the volume leaving as 16-bit cells in planes of 8-channel groups [D, C/8, h, w, 8] (the library's h16
// format, RNE of the fp32 value d3d_weighted_corr stores): plane d is one contiguous block whose cells the fused conv-GRU cell
// stages with 16-byte loads (d3d_gru_cell_fused_cl8_h16).  Window kernel only (C % 8 == 0, at most 4 source views, D <= 48: every
// stage of the cascades); D3D_ERR_UNSUPPORTED otherwise -- the caller then takes d3d_weighted_corr.
int d3d_weighted_corr_cl8_h16(const float* const* feats, const float* proj34, const float* weights, const float* depth,
                              int depth_mode, int n_views, int C, int D, int h, int w, void* out, void* workspace,
                              size_t workspace_bytes, d3d_stream_t stream) {
    SweepParams p = {};
    D3D_REQUIRE(weights, "null weights");
    int rc = fill_multi(p, feats, proj34, depth, depth_mode, n_views, C, D, h, w, reinterpret_cast<float*>(out), workspace, workspace_bytes);
    if (rc) return rc;
    p.weights = weights;
    p.out_cl = 2;
    if (forced_path() == 1 || forced_path() == 2) {
        set_error("d3d_weighted_corr_cl8_h16: only the window kernel writes the channel-last correlation volume");
        return D3D_ERR_UNSUPPORTED;
    }
    return counted(3, launch_window(MODE_WEIGHTED, p, (hipStream_t)stream, forced_path() == 3));
}

int d3d_pair_corr_mean(const float* ref, const float* src, const float* proj34, const float* depth, int depth_mode,
                       int C, int D, int h, int w, float* out, void* workspace, size_t workspace_bytes,
                       d3d_stream_t stream) {
    const float* feats[2] = {ref, src};
    SweepParams p = {};
    int rc = fill_multi(p, feats, proj34, depth, depth_mode, 2, C, D, h, w, out, workspace, workspace_bytes);
    if (rc) return rc;
    return sweep_dispatch(MODE_PAIR, p, (hipStream_t)stream);
}

}  // extern "C"
