// z-streaming implicit-GEMM convolution on the gfx950 matrix cores (exact fp32, v_mfma_f32_16x16x4_f32).
//
// Second-generation kernel for the regularisers' k=3 convolutions (cas_mvsnet.py:84-121,
// adamvs.py:198-238,403-427, module.py:5-51,297-304).  conv_mfma.hip restages a 3-plane input patch
// and the packed weights for every output slice; here a workgroup owns an (x,y) tile of output
// COLUMNS and streams the input volume through LDS one z-plane at a time:
//
//   * every input plane is staged ONCE per tile (input-stationary in z): plane zi contributes to
//     the open output columns g_z with zi = g_z*cz + tz through the taps of that tz, accumulated in
//     up to NS live accumulator sets (3 for a stride-1 conv, 2 for stride 2 / transposed, 1 in 2D);
//     a set is flushed (affine, ReLU, skip, store) when its last input plane has passed;
//   * the packed weights of ALL taps stay resident in LDS for the whole stream;
//   * "folding": the GEMM row index m is (fold position, c_out), so a column of the GEMM produces
//     fz*fy*fx neighbouring outputs.  C_out = 8 layers fill the 16 MFMA rows with two x-neighbours,
//     C_out = 1 layers with a 4x4 patch, and a stride-2 transposed convolution IS a fold over its
//     2x2(x2) output parities with input offsets {0,1}: one launch instead of 4/8 parity launches.
//     The host (ops.py) builds the tap list and the zero-padded packed weights for each case.
//
// Geometry per dimension d:  input index  = g_d*c_d + tap_d,   output index = g_d*s_d + b_d + fold_d.
// GEMM: D[m][col] += A[m][k] * B[k][col], k = (tap, ci); A = wpack[t][ci][m] (LDS), B = input patch (LDS).
// Workgroup = 4 waves = 4 column rows (g_y) x 16*NT columns (g_x); lanes: k-group g = lane>>4, j = lane&15.
#include "common.h"

namespace d3d {

namespace {

constexpr int ZS_MAXTAPS = 128;

struct ConvZParams {
    const float* in0;
    const float* in1;
    const float* wpack;  // [ntaps][Ci][MP]
    const float* scale;
    const float* shift;
    const float* skip;
    float* out;
    int Ci0, Ci1, Co, M;
    int D, H, W;
    int Do, Ho, Wo;
    int Gz, Gy, Gx;
    int cz, cy, cx;
    int sz, sy, sx;
    int bz, by, bx;
    int fz, fy, fx;
    int act, skip_after_act;
    int ntaps, zmin, zspan, ymin, yspan, xmin, xspan;
    int PY, PX, CS, CiP;
    int nseg, mg_nseg, mg_py;  // staging: 16-lane segments per patch row, 16-bit reciprocal multipliers
    int zseg;                  // output columns in z per workgroup
    int tzstart[5];            // taps sorted by tz: taps of tz = zmin+i are [tzstart[i], tzstart[i+1])
    signed char ty[ZS_MAXTAPS], tx[ZS_MAXTAPS];
};

typedef float f4v __attribute__((ext_vector_type(4)));

template <int MT, int NT, int NS, int CK>
__global__ __launch_bounds__(256) void conv_stream_kernel(ConvZParams p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int MP = 16 * MT;
    constexpr int WS = (MP == 16) ? 16 : MP + 16;  // weight row stride: k-groups g, g+1 on disjoint bank halves
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, j = lane & 15;
    const int Ci = p.Ci0 + p.Ci1, CiP = p.CiP, PX = p.PX, PY = p.PY, CS = p.CS;

    float* wl = lds;                                          // [ntaps*CiP][WS]
    int* tofft = reinterpret_cast<int*>(lds + p.ntaps * CiP * WS);  // [ntaps] patch offset of each tap
    float* xin = reinterpret_cast<float*>(tofft + ((p.ntaps + 3) & ~3));  // [CK][CS] one input plane chunk

    // ---- one-time: resident weights (rows beyond Ci are zero), tap offsets
    {
        const int rows = p.ntaps * CiP;
        const int n4 = rows * (MP / 4);
        for (int e0 = tid; e0 < n4; e0 += 256 * 4) {
            float4 v[4];
            int dsto[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int e = e0 + u * 256;
                const bool in = e < n4;
                const int row = in ? e / (MP / 4) : 0, q = in ? e - row * (MP / 4) : 0;
                const int t = row / CiP, c = row - t * CiP;
                const bool ok = in && c < Ci;
                const float4* src = reinterpret_cast<const float4*>(p.wpack + ((long)t * Ci + (ok ? c : 0)) * MP) + q;
                v[u] = *src;
                if (!ok) v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
                dsto[u] = in ? row * WS + q * 4 : -1;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (dsto[u] >= 0) *reinterpret_cast<float4*>(wl + dsto[u]) = v[u];
        }
        if (tid < p.ntaps) tofft[tid] = (p.ty[tid] - p.ymin) * PX + (p.tx[tid] - p.xmin);
    }

    // ---- GEMM row -> (c_out, fold position) of this lane's accumulator rows m = mt*16 + 4g + r
    int rowinfo[MT][4];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int m = mt * 16 + 4 * g + r;
            int info = -1;
            if (m < p.M) {
                const int q0 = m / p.Co, co = m - q0 * p.Co;
                const int q1 = q0 / p.fx, fxv = q0 - q1 * p.fx;
                const int fzv = q1 / p.fy, fyv = q1 - fzv * p.fy;
                info = co | (fxv << 8) | (fyv << 16) | (fzv << 24);
            }
            rowinfo[mt][r] = info;
        }

    const int gx0 = blockIdx.x * (16 * NT);
    const int gy = blockIdx.y * 4 + wave;
    const int gz_lo = blockIdx.z * p.zseg;
    const int gz_hi = min(p.Gz, gz_lo + p.zseg);
    const int iy0 = blockIdx.y * 4 * p.cy + p.ymin, ix0 = gx0 * p.cx + p.xmin;
    const int bbase = g * CS + (wave * p.cy) * PX + j * p.cx;
    const long in_plane = (long)p.H * p.W, in_vol = in_plane * p.D;
    const long out_plane = (long)p.Ho * p.Wo;
    const int zmax = p.zmin + p.zspan - 1;

    f4v acc[NS][MT][NT];
#pragma unroll
    for (int s = 0; s < NS; ++s)
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int n = 0; n < NT; ++n) acc[s][m][n] = (f4v){0, 0, 0, 0};

    // staging constants of this thread: 16-lane group `grp` walks patch rows grp, grp+16, ...; lanes run along x
    const int grp = tid >> 4, xs = tid & 15;
    const int srows = CK * PY;
    const int nitems = ((srows + 15) >> 4) * p.nseg;

    auto flush = [&](int gz) {
        if (gy < p.Gy) {
            const int oyb = gy * p.sy + p.by, ozb = gz * p.sz + p.bz;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    int info = rowinfo[mt][r];
                    // opaque to loop-invariant code motion: otherwise every row's 64-bit output address is
                    // precomputed outside the z loop and held in registers (32*MT VGPRs)
                    asm volatile("" : "+v"(info));
                    if (info < 0) continue;
                    const int co = info & 255, fxv = (info >> 8) & 255, fyv = (info >> 16) & 255, fzv = info >> 24;
                    const int oz = ozb + fzv, oy = oyb + fyv;
                    if (oz >= p.Do || oy >= p.Ho) continue;
                    const float sc = p.scale ? p.scale[co] : 1.0f;
                    const float sh = p.shift ? p.shift[co] : 0.0f;
                    const long obase = ((long)co * p.Do + oz) * out_plane + (long)oy * p.Wo;
#pragma unroll
                    for (int n = 0; n < NT; ++n) {
                        const int gx = gx0 + n * 16 + j;
                        const int ox = gx * p.sx + p.bx + fxv;
                        if (gx >= p.Gx || ox >= p.Wo) continue;
                        const long oidx = obase + ox;
                        float y = acc[0][mt][n][r];
                        if (p.scale) y *= sc;
                        if (p.shift) y += sh;
                        if (p.skip && !p.skip_after_act) y += p.skip[oidx];
                        if (p.act == 1) y = fmaxf(y, 0.0f);
                        if (p.skip && p.skip_after_act) y = p.skip[oidx] + y;
                        p.out[oidx] = y;
                    }
                    // keep the unrolled rows sequential: hoisting every row's skip loads and addresses
                    // together costs >100 VGPRs and an occupancy step
                    __builtin_amdgcn_sched_barrier(0);
                }
        }
        // rotate the live sets: slot s <- slot s+1, last slot cleared
#pragma unroll
        for (int s = 0; s + 1 < NS; ++s)
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int n = 0; n < NT; ++n) acc[s][m][n] = acc[s + 1][m][n];
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int n = 0; n < NT; ++n) acc[NS - 1][m][n] = (f4v){0, 0, 0, 0};
    };

    int gbase = gz_lo;  // output column (z) held by accumulator slot 0
    const int zi_first = gz_lo * p.cz + p.zmin, zi_last = (gz_hi - 1) * p.cz + zmax;
    const int nchunks = CiP / CK;

    for (int zi = zi_first; zi <= zi_last; ++zi) {
        while (gbase < gz_hi && zi > gbase * p.cz + zmax) {
            flush(gbase);
            ++gbase;
        }
        if ((unsigned)zi >= (unsigned)p.D) continue;  // plane outside the volume: zero padding
        for (int c = 0; c < nchunks; ++c) {
            __syncthreads();  // previous plane chunk consumed (and, first time, weights/tap table written)
            // ---- stage channels [c*CK, c*CK+CK) of plane zi: zeros outside the image / beyond Ci
            for (int i0 = 0; i0 < nitems; i0 += 8) {
                float v[8];
                int dsto[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int i = i0 + u;
                    const int k = (i * p.mg_nseg) >> 16, seg = i - k * p.nseg;
                    const int r = grp + 16 * k;
                    const int cc = (r * p.mg_py) >> 16, y = r - cc * PY;
                    const int x = seg * 16 + xs;
                    const int ci = c * CK + cc;
                    const int sy = iy0 + y, sx = ix0 + x;
                    const bool slot = (i < nitems) && (r < srows) && (x < PX);
                    const bool ok = slot && (ci < Ci) && (unsigned)sy < (unsigned)p.H && (unsigned)sx < (unsigned)p.W;
                    const float* __restrict__ src = (ci < p.Ci0 || !ok) ? p.in0 + (long)(ok ? ci : 0) * in_vol
                                                                         : p.in1 + (long)(ci - p.Ci0) * in_vol;
                    const float val = src[ok ? (long)zi * in_plane + (long)sy * p.W + sx : 0];
                    v[u] = ok ? val : 0.0f;
                    dsto[u] = slot ? cc * CS + y * PX + x : -1;
                }
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    if (dsto[u] >= 0) xin[dsto[u]] = v[u];
            }
            __syncthreads();

            // ---- every open output column takes the taps whose tz links it to this plane
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                const int gzs = gbase + s;
                const int tzv = zi - gzs * p.cz;
                if (gzs >= gz_hi || tzv < p.zmin || tzv > zmax) continue;
                const int tb = p.tzstart[tzv - p.zmin], te = p.tzstart[tzv - p.zmin + 1];
                if (tb >= te) continue;
                int toff = tofft[tb];
                for (int t = tb; t < te; ++t) {
                    const int toff_next = tofft[min(t + 1, p.ntaps - 1)];
                    const float* __restrict__ xb = xin + bbase + toff;
                    const float* __restrict__ wa = wl + ((t * CiP + c * CK + g) * WS + j);
#pragma unroll
                    for (int kk = 0; kk < CK / 4; ++kk) {
                        float b[NT];
#pragma unroll
                        for (int n = 0; n < NT; ++n) b[n] = xb[kk * 4 * CS + n * 16 * p.cx];
#pragma unroll
                        for (int m = 0; m < MT; ++m) {
                            const float av = wa[kk * 4 * WS + m * 16];
#pragma unroll
                            for (int n = 0; n < NT; ++n)
                                acc[s][m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b[n], acc[s][m][n], 0, 0, 0);
                        }
                    }
                    toff = toff_next;
                }
            }
        }
    }
    while (gbase < gz_hi) {
        flush(gbase);
        ++gbase;
    }
}

struct StreamCfg {
    int MT, NT, NS, CK;
    int lds_bytes;
};

static int patch_stride(int PY, int PX, int cx) {
    int cs = PY * PX;
    if (cx == 1) cs += (16 - (cs & 31) + 32) & 31;  // CS = 16 mod 32
    else if (cx == 2) cs |= 1;                       // odd
    else if (cx == 4) cs += (2 - (cs & 3) + 4) & 3;  // CS = 2 mod 4
    return cs;
}

static int lds_bytes_for(const ConvZParams& p, int MT, int NT, int CK) {
    const int MP = 16 * MT, WS = (MP == 16) ? 16 : MP + 16;
    const int Ci = p.Ci0 + p.Ci1, CiP = (Ci + CK - 1) / CK * CK;
    const int PY = p.yspan + 3 * p.cy, PX = p.xspan + (16 * NT - 1) * p.cx;
    return 4 * (p.ntaps * CiP * WS + ((p.ntaps + 3) & ~3) + CK * patch_stride(PY, PX, p.cx));
}

template <int MT, int NT, int NS, int CK>
static int launch_stream(ConvZParams& p, hipStream_t stream) {
    const int Ci = p.Ci0 + p.Ci1;
    p.CiP = (Ci + CK - 1) / CK * CK;
    p.PY = p.yspan + 3 * p.cy;
    p.PX = p.xspan + (16 * NT - 1) * p.cx;
    p.CS = patch_stride(p.PY, p.PX, p.cx);
    p.nseg = (p.PX + 15) / 16;
    p.mg_nseg = 65536 / p.nseg + 1;
    p.mg_py = 65536 / p.PY + 1;
    const int bytes = lds_bytes_for(p, MT, NT, CK);
    auto kern = conv_stream_kernel<MT, NT, NS, CK>;
    static bool attr_set = false;
    if (!attr_set) {
        int rc = hip_status(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024),
                            "hipFuncSetAttribute(MaxDynamicSharedMemorySize)");
        if (rc != D3D_OK) return rc;
        attr_set = true;
    }
    const int nx = ceil_div(p.Gx, 16 * NT), ny = ceil_div(p.Gy, 4);
    // enough workgroups to fill 256 CUs several times over, at the price of 2 halo planes per z segment
    int nz = ceil_div(4096, (long)nx * ny);
    nz = nz < 1 ? 1 : (nz > p.Gz ? p.Gz : nz);
    p.zseg = ceil_div(p.Gz, nz);
    nz = ceil_div(p.Gz, p.zseg);
    if (ny > 65535 || nz > 65535) {
        set_error("conv_stream: grid %dx%dx%d too large", nx, ny, nz);
        return D3D_ERR_UNSUPPORTED;
    }
    hipLaunchKernelGGL(kern, dim3(nx, ny, nz), dim3(256), bytes, stream, p);
    D3D_LAUNCH_CHECK("conv_stream_kernel launch");
    return D3D_OK;
}

template <int MT, int NT, int CK>
static int launch_ns(ConvZParams& p, int NS, hipStream_t stream) {
    if (NS == 1) return launch_stream<MT, NT, 1, CK>(p, stream);
    if (NS == 2) return launch_stream<MT, NT, 2, CK>(p, stream);
    return launch_stream<MT, NT, 3, CK>(p, stream);
}

template <int MT, int NT>
static int launch_ck(ConvZParams& p, int NS, int CK, hipStream_t stream) {
    if (CK == 16) return launch_ns<MT, NT, 16>(p, NS, stream);
    return launch_ns<MT, NT, 8>(p, NS, stream);
}

}  // namespace

}  // namespace d3d

using namespace d3d;

extern "C" {

int d3d_conv_fold_f32(const float* in0, int Ci0, const float* in1, int Ci1, const float* wpack, int mpad, int M,
                      const float* scale, const float* shift, const float* skip, int skip_after_act, int act,
                      int Co, int D, int H, int W, int Do, int Ho, int Wo, const int* geom, int ntaps,
                      const signed char* taps_zyx, float* out, d3d_stream_t stream) {
    D3D_REQUIRE(in0 && wpack && out && taps_zyx && geom, "null pointer");
    D3D_REQUIRE(Ci0 > 0 && Ci1 >= 0 && (Ci1 == 0 || in1), "bad input channel split %d+%d", Ci0, Ci1);
    D3D_REQUIRE(D > 0 && H > 0 && W > 0 && Do > 0 && Ho > 0 && Wo > 0, "bad dims");
    D3D_REQUIRE(ntaps > 0 && ntaps <= ZS_MAXTAPS, "bad ntaps %d", ntaps);
    D3D_REQUIRE(act == 0 || act == 1, "bad act %d", act);
    ConvZParams p = {};
    p.in0 = in0; p.in1 = in1; p.wpack = wpack; p.scale = scale; p.shift = shift; p.skip = skip; p.out = out;
    p.Ci0 = Ci0; p.Ci1 = Ci1; p.Co = Co; p.M = M; p.D = D; p.H = H; p.W = W; p.Do = Do; p.Ho = Ho; p.Wo = Wo;
    p.Gz = geom[0]; p.Gy = geom[1]; p.Gx = geom[2];
    p.cz = geom[3]; p.cy = geom[4]; p.cx = geom[5];
    p.sz = geom[6]; p.sy = geom[7]; p.sx = geom[8];
    p.bz = geom[9]; p.by = geom[10]; p.bx = geom[11];
    p.fz = geom[12]; p.fy = geom[13]; p.fx = geom[14];
    p.act = act; p.skip_after_act = skip_after_act ? 1 : 0; p.ntaps = ntaps;
    D3D_REQUIRE(p.Gz > 0 && p.Gy > 0 && p.Gx > 0, "bad column grid %dx%dx%d", p.Gz, p.Gy, p.Gx);
    D3D_REQUIRE(p.cz > 0 && p.cy > 0 && p.cx > 0 && p.sz > 0 && p.sy > 0 && p.sx > 0, "bad steps");
    D3D_REQUIRE(p.bz >= 0 && p.by >= 0 && p.bx >= 0, "bad output base");
    D3D_REQUIRE(p.fz > 0 && p.fy > 0 && p.fx > 0 && p.fz < 128 && p.fy < 256 && p.fx < 256, "bad fold");
    D3D_REQUIRE(Co > 0 && Co <= 64 && M == Co * p.fz * p.fy * p.fx && M <= 64, "bad rows: Co=%d fold=%dx%dx%d M=%d", Co,
                p.fz, p.fy, p.fx, M);
    const int MT = M <= 16 ? 1 : (M <= 32 ? 2 : 4);
    D3D_REQUIRE(mpad == 16 * MT, "mpad=%d, expected %d", mpad, 16 * MT);
    // the last column must start inside the output (individual fold positions may overhang and are clipped)
    D3D_REQUIRE((p.Gz - 1) * p.sz + p.bz < Do && (p.Gy - 1) * p.sy + p.by < Ho && (p.Gx - 1) * p.sx + p.bx < Wo,
                "column grid exceeds the output tensor");
    int lo[3] = {127, 127, 127}, hi[3] = {-128, -128, -128};
    int prev_tz = -128;
    for (int t = 0; t < ntaps; ++t) {
        const int tz = taps_zyx[3 * t];
        D3D_REQUIRE(tz >= prev_tz, "taps must be sorted by z offset");
        prev_tz = tz;
        for (int d = 0; d < 3; ++d) {
            const int v = taps_zyx[3 * t + d];
            lo[d] = v < lo[d] ? v : lo[d];
            hi[d] = v > hi[d] ? v : hi[d];
        }
        p.ty[t] = taps_zyx[3 * t + 1];
        p.tx[t] = taps_zyx[3 * t + 2];
    }
    p.zmin = lo[0]; p.zspan = hi[0] - lo[0] + 1;
    p.ymin = lo[1]; p.yspan = hi[1] - lo[1] + 1;
    p.xmin = lo[2]; p.xspan = hi[2] - lo[2] + 1;
    D3D_REQUIRE(p.zspan <= 4, "z tap span %d too large", p.zspan);
    {
        int t = 0;
        for (int i = 0; i <= p.zspan; ++i) {
            while (t < ntaps && taps_zyx[3 * t] < p.zmin + i) ++t;
            p.tzstart[i] = t;
        }
    }
    const int open_cols = (p.zspan + p.cz - 1) / p.cz;
    D3D_REQUIRE(open_cols <= 3, "more than 3 open output columns in z (span %d, step %d)", p.zspan, p.cz);
    const int NS = open_cols;
    const int Ci = Ci0 + Ci1;
    // tile / chunk choice: narrow tiles when the folded patch is wide; 16-channel chunks when they still leave
    // room for two workgroups per CU
    int NT = 4, CK = (Ci > 8) ? 16 : 8;
    if (MT == 1 && lds_bytes_for(p, MT, 4, 8) > 72 * 1024) NT = 1;
    if (CK == 16 && lds_bytes_for(p, MT, NT, 16) > 72 * 1024) CK = 8;
    const int bytes = lds_bytes_for(p, MT, NT, CK);
    if (bytes > 156 * 1024) {
        set_error("conv_stream: resident weights + patch need %d B of LDS (ntaps=%d Ci=%d M=%d)", bytes, ntaps, Ci, M);
        return D3D_ERR_UNSUPPORTED;
    }
    {
        // staging index arithmetic uses 16-bit reciprocals: check their exact range on the host
        const int PY = p.yspan + 3 * p.cy, PX = p.xspan + (16 * NT - 1) * p.cx;
        const int nseg = (PX + 15) / 16, rows = CK * PY;
        const int items = ((rows + 15) / 16) * nseg + 8;
        D3D_REQUIRE(items < 1024 && rows + 16 < 1024 && nseg <= 64 && PY <= 64, "patch %dx%d too large", PY, PX);
    }
    hipStream_t st = (hipStream_t)stream;
    if (NT == 1) return launch_ck<1, 1>(p, NS, CK, st);
    switch (MT) {
        case 1: return launch_ck<1, 4>(p, NS, CK, st);
        case 2: return launch_ck<2, 4>(p, NS, CK, st);
        default: return launch_ck<4, 4>(p, NS, CK, st);
    }
}

}  // extern "C"
