// Convolution family of the cost regularisers (fp32, direct form) for gfx950.
//
// One templated kernel covers nn.Conv2d / nn.Conv3d (k=3, pad=1, stride 1|2) and one
// covers the stride-2 transposed forms; the 2D case is the 3D kernel with a depth-1
// kernel (KZ = 1).  A lane owns one output pixel and COT output channels, input rows are
// read coalesced along x and the weights come in through wave-uniform (scalar) loads.
// Epilogue: per-channel affine (folded eval-mode BatchNorm or bias), ReLU, skip add.
// Reference citations: include/deep3d_planesweep.h.
#include "common.h"

#include <cstdlib>

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
    int act;             // 0 none, 1 relu; conv2d_stream_kernel also 2 | 3 = ConvGRUCell gates / state update (see there)
    int skip_after_act;  // 1: out = skip + act(y) ; 0: out = act(y + skip)
    const float* aux1;   // act 3: the update gate u
    int ep_split;        // act 2: channels < ep_split are the reset gate (multiplied by h = skip)
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

// ---------------------------------------------------------------------------------------------------------------
// C_out = 1, stride 1 (the probability layer of CostRegNet, cas_mvsnet.py:110: nn.Conv3d(8, 1, 3, padding=1)): a single
// output channel wastes 15 of the 16 rows of a matrix-core tile (the 4x4-patch fold reaches 25 % useful work), and
// the layer is pure streaming: 8 input planes per output plane.  This kernel streams the input volume through LDS one
// z-plane at a time (each input element is fetched from memory once per 64x4 tile); a lane owns one (x, y) column and
// keeps the three output planes a staged input plane contributes to in registers.  Per lane and plane: 9*Ci LDS
// reads, 27*Ci FMAs with wave-uniform (scalar-loaded) weights; the next plane's loads are in flight during the FMAs.
// ---------------------------------------------------------------------------------------------------------------
constexpr int C1_TX = 64, C1_TY = 4, C1_PW = C1_TX + 2, C1_PH = C1_TY + 2, C1_PS = C1_PW + 1;  // patch row stride (pad)

template <int CI>
__global__ __launch_bounds__(256) void conv3d_co1_kernel(ConvParams p, int zseg) {
    constexpr int PLANE = C1_PH * C1_PS;      // floats per channel in a patch
    constexpr int NLD = (CI * C1_PH * C1_PW + 255) / 256;  // staging loads per lane and plane
    __shared__ float patch[2][CI * PLANE + 1];
    const int tid = threadIdx.x;
    const int tx = tid & 63, ty = tid >> 6;
    const int x0 = blockIdx.x * C1_TX, y0 = blockIdx.y * C1_TY;
    const int z_lo = blockIdx.z * zseg, z_hi = min(z_lo + zseg, p.D);  // output planes [z_lo, z_hi)
    const long in_plane = (long)p.H * p.W, in_vol = in_plane * p.D;
    const int x = x0 + tx, y = y0 + ty;
    const bool valid = (x < p.W) && (y < p.H);

    // staging element e (0 .. CI*PH*PW): channel, patch row, patch col -> 32-bit byte offset from the plane's base
    // (raw buffer loads: out-of-image elements get an out-of-range offset and read as the zero padding) and LDS slot
    unsigned voff[NLD];
    int lslot[NLD];
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
        const int e = tid + 256 * i;
        const int c = e / (C1_PH * C1_PW), r = e - c * (C1_PH * C1_PW);
        const int py = r / C1_PW, px = r - py * C1_PW;
        const int gy = y0 - 1 + py, gx = x0 - 1 + px;
        const bool live = e < CI * C1_PH * C1_PW;
        const bool ok = live && gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
        voff[i] = ok ? (unsigned)(((long)c * in_vol + (long)gy * p.W + gx) * 4) : 0x80000000u;
        lslot[i] = live ? c * PLANE + py * C1_PS + px : CI * PLANE;  // dead elements land in a spare word
    }
    const int span_bytes = (int)((((long)CI - 1) * in_vol + in_plane) * 4);  // (host: < 2^31)
    float pv[NLD];
    auto issue = [&](int zi) {
        const bool zin = zi >= 0 && zi < p.D;
        const float* base = p.in0 + (long)(zin ? zi : 0) * in_plane;
        const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, zin ? span_bytes : 0, 0x00020000);
#pragma unroll
        for (int i = 0; i < NLD; ++i)
            pv[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, (int)voff[i], 0, 0));
    };
    auto land = [&](int buf) {
#pragma unroll
        for (int i = 0; i < NLD; ++i) patch[buf][lslot[i]] = pv[i];
    };

    float a_prev = 0.0f, a_cur = 0.0f, a_next = 0.0f;  // output planes zi-1, zi, zi+1 while input plane zi is swept
    // weights [1][CI][3][3][3] through the constant address space: wave-uniform addresses become scalar loads, and the
    // per-plane laundering below keeps them from being hoisted out of the plane loop into 27*CI registers
    typedef const float __attribute__((address_space(4))) cfloat;
    cfloat* wt = (cfloat*)p.weight;
    issue(z_lo - 1);
    land(0);
    __syncthreads();
    for (int zi = z_lo - 1, it = 0; zi <= z_hi; ++zi, ++it) {
        const int buf = it & 1;
        if (zi + 1 <= z_hi) issue(zi + 1);  // next plane's loads fly during the FMAs
        const float* __restrict__ pl = patch[buf] + ty * C1_PS + tx;
#pragma unroll 1  // (a rolled channel loop bounds the live weights to 27 scalar registers)
        for (int c = 0; c < CI; ++c) {
            cfloat* wc = wt + c * 27;
            asm volatile("" : "+s"(wc));
#pragma unroll
            for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) {
                    const float v = pl[c * PLANE + dy * C1_PS + dx];
                    cfloat* w = wc + dy * 3 + dx;
                    a_next = fmaf(v, w[0], a_next);   // kz = 0: this plane is the z-1 neighbour of output zi+1
                    a_cur = fmaf(v, w[9], a_cur);     // kz = 1
                    a_prev = fmaf(v, w[18], a_prev);  // kz = 2: ... the z+1 neighbour of output zi-1
                }
        }
        // output plane zi-1 is complete (it has seen input planes zi-2, zi-1, zi)
        const int zo = zi - 1;
        if (valid && zo >= z_lo && zo < z_hi) {
            const long oidx = (long)zo * in_plane + (long)y * p.W + x;
            p.out[oidx] = epilogue(a_prev, 0, oidx, p);
        }
        a_prev = a_cur;
        a_cur = a_next;
        a_next = 0.0f;
        if (zi + 1 <= z_hi) land(buf ^ 1);
        __syncthreads();
    }
}

static int launch_conv3d_co1(const ConvParams& p, hipStream_t stream) {
    const int gx = ceil_div(p.W, C1_TX), gy = ceil_div(p.H, C1_TY);
    // depth segments: enough workgroups to fill the chip, at least 8 planes each (two halo planes per segment)
    int nz = 1;
    while ((long)gx * gy * nz < 2048 && p.D / (nz * 2) >= 8) nz *= 2;
    const int zseg = ceil_div(p.D, nz);
    nz = ceil_div(p.D, zseg);
    if (gy > 65535 || nz > 65535) return D3D_ERR_UNSUPPORTED;
    dim3 grid(gx, gy, nz);
    const long in_plane = (long)p.H * p.W;
    if (p.Ci0 != 8 || ((long)7 * in_plane * p.D + in_plane) * 4 >= (1L << 31)) return D3D_ERR_UNSUPPORTED;  // 32-bit offsets
    hipLaunchKernelGGL(conv3d_co1_kernel<8>, grid, dim3(256), 0, stream, p, zseg);
    D3D_LAUNCH_CHECK("conv3d_co1_kernel launch");
    return D3D_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// C_out = 8, stride 1 (conv0 of every CostRegNet, cas_mvsnet.py:84: ConvBnReLU3D(C_in, 8)): the fp32 vector units of
// this chip have the same peak as its fp32 matrix cores (157 TFLOP/s), and an 8-row GEMM fills half an MFMA tile at
// best, so these three layers -- 14 ms of a 53 ms CasMVSNet view on the matrix-core kernel -- go through the same
// z-streaming form as the single-channel kernel above: a lane owns one (x, y) column, eight output channels and the
// three open output planes (24 accumulators); the input arrives in chunks of 8 channels per LDS stage; weights are
// pre-packed [C_in][ky][kx][kz][8] so that the 72 weights of a (channel, row) are contiguous scalar loads feeding
// v_pk_fma_f32 (two output channels per instruction).  Per tap value: 1 LDS read, 24 FMAs.
// ---------------------------------------------------------------------------------------------------------------
template <int XP>  // output pixels per lane along x (tile = 64*XP x 4): XP = 2 halves the scalar weight loads per FMA
__global__ __launch_bounds__(256) void conv3d_co8_kernel(ConvParams p, int zseg) {
    constexpr int CK = 8, CO = 8;
    constexpr int TX = 64 * XP, PW = TX + 2, PS = PW + 1, PH = C1_TY + 2;
    constexpr int PLANE = PH * PS;
    constexpr int NLD = (CK * PH * PW + 255) / 256;
    __shared__ float patch[2][CK * PLANE + 1];
    const int tid = threadIdx.x;
    const int tx = tid & 63, ty = tid >> 6;
    const int x0 = blockIdx.x * TX, y0 = blockIdx.y * C1_TY;
    const int z_lo = blockIdx.z * zseg, z_hi = min(z_lo + zseg, p.D);
    const long in_plane = (long)p.H * p.W, in_vol = in_plane * p.D;
    const int y = y0 + ty;
    const int nchunk = p.Ci0 / CK;

    unsigned voff[NLD];
    int lslot[NLD];
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
        const int e = tid + 256 * i;
        const int c = e / (PH * PW), r = e - c * (PH * PW);
        const int py = r / PW, px = r - py * PW;
        const int gy = y0 - 1 + py, gx = x0 - 1 + px;
        const bool live = e < CK * PH * PW;
        const bool ok = live && gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
        voff[i] = ok ? (unsigned)(((long)c * in_vol + (long)gy * p.W + gx) * 4) : 0x80000000u;
        lslot[i] = live ? c * PLANE + py * PS + px : CK * PLANE;
    }
    const int span_bytes = (int)((((long)CK - 1) * in_vol + in_plane) * 4);  // (host: < 2^31)
    float pv[NLD];
    // step s = (input plane, channel chunk)
    auto issue = [&](int zi, int ch) {
        const bool zin = zi >= 0 && zi < p.D;
        const float* base = p.in0 + (long)ch * CK * in_vol + (long)(zin ? zi : 0) * in_plane;
        const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, zin ? span_bytes : 0, 0x00020000);
#pragma unroll
        for (int i = 0; i < NLD; ++i)
            pv[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, (int)voff[i], 0, 0));
    };
    auto land = [&](int buf) {
#pragma unroll
        for (int i = 0; i < NLD; ++i) patch[buf][lslot[i]] = pv[i];
    };

    float acc[XP][3][CO];  // [.][0] output plane zi-1, [1] zi, [2] zi+1 while input plane zi is swept
#pragma unroll
    for (int q = 0; q < XP; ++q)
#pragma unroll
        for (int k = 0; k < 3; ++k)
#pragma unroll
            for (int o = 0; o < CO; ++o) acc[q][k][o] = 0.0f;
    typedef const float __attribute__((address_space(4))) cfloat;
    cfloat* wt = (cfloat*)p.weight;  // packed [C_in][3 ky][3 kx][3 kz][8]

    issue(z_lo - 1, 0);
    land(0);
    __syncthreads();
    int it = 0;
    for (int zi = z_lo - 1; zi <= z_hi; ++zi) {
        for (int ch = 0; ch < nchunk; ++ch, ++it) {
            const int buf = it & 1;
            // next step: next chunk of this plane, or the first chunk of the next plane
            const bool more = (ch + 1 < nchunk) || (zi + 1 <= z_hi);
            if (more) issue(ch + 1 < nchunk ? zi : zi + 1, ch + 1 < nchunk ? ch + 1 : 0);
            const float* __restrict__ pl = patch[buf] + ty * PS + tx;
#pragma unroll 1
            for (int c = 0; c < CK; ++c) {
#pragma unroll 1
                for (int dy = 0; dy < 3; ++dy) {
                    cfloat* w = wt + ((ch * CK + c) * 3 + dy) * 72;
                    asm volatile("" : "+s"(w));  // (reloaded per row: 72 live scalars, not 216 * C_in)
#pragma unroll
                    for (int dx = 0; dx < 3; ++dx) {
#pragma unroll
                        for (int q = 0; q < XP; ++q) {
                            const float v = pl[c * PLANE + dy * PS + dx + 64 * q];
#pragma unroll
                            for (int o = 0; o < CO; ++o) {
                                acc[q][2][o] = fmaf(v, w[dx * 24 + 0 * 8 + o], acc[q][2][o]);  // kz = 0 -> output plane zi+1
                                acc[q][1][o] = fmaf(v, w[dx * 24 + 1 * 8 + o], acc[q][1][o]);
                                acc[q][0][o] = fmaf(v, w[dx * 24 + 2 * 8 + o], acc[q][0][o]);  // kz = 2 -> output plane zi-1
                            }
                        }
                    }
                }
            }
            if (more) land(buf ^ 1);
            __syncthreads();
        }
        const int zo = zi - 1;
#pragma unroll
        for (int q = 0; q < XP; ++q) {
            const int x = x0 + tx + 64 * q;
            if (x < p.W && y < p.H && zo >= z_lo && zo < z_hi) {
                const long obase = (long)zo * in_plane + (long)y * p.W + x;
#pragma unroll
                for (int o = 0; o < CO; ++o) {
                    const long oidx = (long)o * in_vol + obase;
                    p.out[oidx] = epilogue(acc[q][0][o], o, oidx, p);
                }
            }
#pragma unroll
            for (int o = 0; o < CO; ++o) {
                acc[q][0][o] = acc[q][1][o];
                acc[q][1][o] = acc[q][2][o];
                acc[q][2][o] = 0.0f;
            }
        }
    }
}

static int launch_conv3d_co8(const ConvParams& p, hipStream_t stream) {
    // two pixels per lane (128-wide tiles) unless the wider tiles pad the rows by over 4 % more (measured: 2752 and
    // 1376 wide +13 % / +8 % faster with two, 688 wide 10 % slower)
    const double pad1 = ceil_div(p.W, 64) * 64.0 / p.W, pad2 = ceil_div(p.W, 128) * 128.0 / p.W;
    int xp = (pad2 - pad1 <= 0.04) ? 2 : 1;
#ifdef D3D_EXPERIMENTS
    if (const char* e = getenv("D3D_CONV_CO8_XP")) xp = atoi(e) == 1 ? 1 : 2;
#endif
    const int gx = ceil_div(p.W, 64 * xp), gy = ceil_div(p.H, C1_TY);
    int nz = 1;
    while ((long)gx * gy * nz < 2048 && p.D / (nz * 2) >= 8) nz *= 2;
    const int zseg = ceil_div(p.D, nz);
    nz = ceil_div(p.D, zseg);
    const long in_plane = (long)p.H * p.W;
    if (gy > 65535 || nz > 65535 || p.Ci0 % 8 != 0 || ((long)7 * in_plane * p.D + in_plane) * 4 >= (1L << 31))
        return D3D_ERR_UNSUPPORTED;
    if (xp == 2)
        hipLaunchKernelGGL(conv3d_co8_kernel<2>, dim3(gx, gy, nz), dim3(256), 0, stream, p, zseg);
    else
        hipLaunchKernelGGL(conv3d_co8_kernel<1>, dim3(gx, gy, nz), dim3(256), 0, stream, p, zseg);
    D3D_LAUNCH_CHECK("conv3d_co8_kernel launch");
    return D3D_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// Transposed convolution k = 3, stride 2, pad 1, output_pad 1 with C_out = 8 (conv11 of every CostRegNet,
// cas_mvsnet.py:103: Deconv3d(16, 8) + BN + ReLU, then the skip add of :118): 35 GFLOP but 2.9 GB of output / skip
// traffic per stage -- a streaming layer.  Same z-streaming vector-unit form as above, seen from the INPUT: a lane owns
// one input (x, y) column = a 2x2 block of output columns; input plane z feeds output planes 2z-1 (k_z = 0), 2z
// (k_z = 1) and 2z+1 (k_z = 2), every kernel tap exactly once:
//   out(2y+py, 2x+px): py = 0 takes k_y = 1 from row y; py = 1 takes k_y = 2 from row y and k_y = 0 from row y+1
// (same along x).  96 accumulators (3 planes x 4 parities x 8 channels); the two x parities of a row leave as one
// 8-byte store per lane (512 contiguous bytes per wave).  Weights packed [C_in][kz][ky][kx][8].
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void convT3d_co8_kernel(ConvParams p, int zseg) {
    constexpr int CK = 8, CO = 8;
    constexpr int PW = C1_TX + 1, PS = PW + 2, PH = C1_TY + 1;  // +1 halo on the high side
    constexpr int PLANE = PH * PS;
    constexpr int NLD = (CK * PH * PW + 255) / 256;
    __shared__ float patch[2][CK * PLANE + 1];
    const int tid = threadIdx.x;
    const int tx = tid & 63, ty = tid >> 6;
    const int x0 = blockIdx.x * C1_TX, y0 = blockIdx.y * C1_TY;
    const int z_lo = blockIdx.z * zseg, z_hi = min(z_lo + zseg, p.D);  // input planes [z_lo, z_hi) -> output planes [2 z_lo, 2 z_hi)
    const long in_plane = (long)p.H * p.W, in_vol = in_plane * p.D;
    const long out_plane = (long)p.Ho * p.Wo, out_vol = out_plane * p.Do;
    const int x = x0 + tx, y = y0 + ty;
    const bool valid = (x < p.W) && (y < p.H);
    const int nchunk = p.Ci0 / CK;

    unsigned voff[NLD];
    int lslot[NLD];
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
        const int e = tid + 256 * i;
        const int c = e / (PH * PW), r = e - c * (PH * PW);
        const int py = r / PW, px = r - py * PW;
        const int gy = y0 + py, gx = x0 + px;
        const bool live = e < CK * PH * PW;
        const bool ok = live && gy < p.H && gx < p.W;
        voff[i] = ok ? (unsigned)(((long)c * in_vol + (long)gy * p.W + gx) * 4) : 0x80000000u;
        lslot[i] = live ? c * PLANE + py * PS + px : CK * PLANE;
    }
    const int span_bytes = (int)((((long)CK - 1) * in_vol + in_plane) * 4);  // (host: < 2^31)
    float pv[NLD];
    auto issue = [&](int zi, int ch) {
        const bool zin = zi >= 0 && zi < p.D;
        const float* base = p.in0 + (long)ch * CK * in_vol + (long)(zin ? zi : 0) * in_plane;
        const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, zin ? span_bytes : 0, 0x00020000);
#pragma unroll
        for (int i = 0; i < NLD; ++i)
            pv[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, (int)voff[i], 0, 0));
    };
    auto land = [&](int buf) {
#pragma unroll
        for (int i = 0; i < NLD; ++i) patch[buf][lslot[i]] = pv[i];
    };

    // acc[kz]: [0] output plane 2zi-1 (opened by the previous input plane), [1] 2zi, [2] 2zi+1; [parity py*2+px][co]
    float acc[3][4][CO];
#pragma unroll
    for (int k = 0; k < 3; ++k)
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int o = 0; o < CO; ++o) acc[k][q][o] = 0.0f;
    typedef const float __attribute__((address_space(4))) cfloat;
    cfloat* wt = (cfloat*)p.weight;  // packed [C_in][3 kz][3 ky][3 kx][8]

    auto store_plane = [&](int oz, float (&a)[4][CO]) {  // output plane oz of this lane's 2x2 columns
        if (!valid) return;
#pragma unroll
        for (int py = 0; py < 2; ++py) {
            const long rowbase = (long)oz * out_plane + (long)(2 * y + py) * p.Wo + 2 * x;
#pragma unroll
            for (int o = 0; o < CO; ++o) {
                const long oidx = (long)o * out_vol + rowbase;
                float2 r;
                r.x = a[py * 2 + 0][o];
                r.y = a[py * 2 + 1][o];
                if (p.scale) { r.x *= p.scale[o]; r.y *= p.scale[o]; }
                if (p.shift) { r.x += p.shift[o]; r.y += p.shift[o]; }
                float2 sk = {0.0f, 0.0f};
                if (p.skip) sk = *reinterpret_cast<const float2*>(p.skip + oidx);
                if (p.skip && !p.skip_after_act) { r.x += sk.x; r.y += sk.y; }
                if (p.act == 1) { r.x = fmaxf(r.x, 0.0f); r.y = fmaxf(r.y, 0.0f); }
                if (p.skip && p.skip_after_act) { r.x = sk.x + r.x; r.y = sk.y + r.y; }
                *reinterpret_cast<float2*>(p.out + oidx) = r;
            }
        }
    };

    issue(z_lo, 0);
    land(0);
    __syncthreads();
    int it = 0;
    for (int zi = z_lo; zi <= z_hi; ++zi) {  // the plane at z_hi only closes output plane 2 z_hi - 1 (zeros beyond the volume)
        for (int ch = 0; ch < nchunk; ++ch, ++it) {
            const int buf = it & 1;
            const bool more = (ch + 1 < nchunk) || (zi + 1 <= z_hi);
            if (more) issue(ch + 1 < nchunk ? zi : zi + 1, ch + 1 < nchunk ? ch + 1 : 0);
            const float* __restrict__ pl = patch[buf] + ty * PS + tx;
#pragma unroll 1
            for (int c = 0; c < CK; ++c) {
                const float v00 = pl[c * PLANE], v01 = pl[c * PLANE + 1], v10 = pl[c * PLANE + PS], v11 = pl[c * PLANE + PS + 1];
#pragma unroll
                for (int kz = 0; kz < 3; ++kz) {
                    cfloat* w = wt + ((ch * CK + c) * 3 + kz) * 72;  // [ky][kx][co]
                    asm volatile("" : "+s"(w));
#pragma unroll
                    for (int o = 0; o < CO; ++o) {
                        float (&a)[4][CO] = acc[kz];
                        a[0][o] = fmaf(v00, w[(1 * 3 + 1) * 8 + o], a[0][o]);
                        a[1][o] = fmaf(v00, w[(1 * 3 + 2) * 8 + o], a[1][o]);
                        a[1][o] = fmaf(v01, w[(1 * 3 + 0) * 8 + o], a[1][o]);
                        a[2][o] = fmaf(v00, w[(2 * 3 + 1) * 8 + o], a[2][o]);
                        a[2][o] = fmaf(v10, w[(0 * 3 + 1) * 8 + o], a[2][o]);
                        a[3][o] = fmaf(v00, w[(2 * 3 + 2) * 8 + o], a[3][o]);
                        a[3][o] = fmaf(v01, w[(2 * 3 + 0) * 8 + o], a[3][o]);
                        a[3][o] = fmaf(v10, w[(0 * 3 + 2) * 8 + o], a[3][o]);
                        a[3][o] = fmaf(v11, w[(0 * 3 + 0) * 8 + o], a[3][o]);
                    }
                    __builtin_amdgcn_sched_barrier(0);  // one k_z group of 72 scalar weights live at a time
                }
            }
            if (more) land(buf ^ 1);
            __syncthreads();
        }
        if (zi > z_lo) store_plane(2 * zi - 1, acc[0]);
        if (zi < z_hi) store_plane(2 * zi, acc[1]);
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int o = 0; o < CO; ++o) {
                acc[0][q][o] = acc[2][q][o];
                acc[1][q][o] = 0.0f;
                acc[2][q][o] = 0.0f;
            }
    }
}

static int launch_convT3d_co8(const ConvParams& p, hipStream_t stream) {
    const int gx = ceil_div(p.W, C1_TX), gy = ceil_div(p.H, C1_TY);
    int nz = 1;
    while ((long)gx * gy * nz < 2048 && p.D / (nz * 2) >= 4) nz *= 2;
    const int zseg = ceil_div(p.D, nz);
    nz = ceil_div(p.D, zseg);
    const long in_plane = (long)p.H * p.W;
    if (gy > 65535 || nz > 65535 || p.Ci0 % 8 != 0 || ((long)7 * in_plane * p.D + in_plane) * 4 >= (1L << 31))
        return D3D_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(convT3d_co8_kernel, dim3(gx, gy, nz), dim3(256), 0, stream, p, zseg);
    D3D_LAUNCH_CHECK("convT3d_co8_kernel launch");
    return D3D_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// 2D 3x3 stride-1 convolution with C_out = 8 | 16 on the vector units (the full- and half-resolution layers of the
// feature pyramids, module.py:653-755: 3 -> 8, 8 -> 8, 32 -> 8, 16 -> 16, 32 -> 16): these are streaming layers (a few
// GFLOP over hundreds of MB) that a 16-row matrix-core tile half fills.  A workgroup takes a 64 x 8 tile, a lane two
// pixels four rows apart; the input arrives in chunks of 8 channels (patch 10 x 66 per channel, the next chunk's
// loads in flight during the FMAs); weights packed [C_in padded to 8][ky][kx][C_out] feed v_pk_fma_f32 from scalar
// registers.  Channels beyond C_in (the 3-channel image layer) are staged as zeros.
// ---------------------------------------------------------------------------------------------------------------
template <int CO, int CK = 8>   // CK: channels per staged chunk (4 for the 3-channel image layer: half the zero work)
__global__ __launch_bounds__(256) void conv2d_stream_kernel(ConvParams p) {
    constexpr int YP = 2, TY = 4 * YP;
    constexpr int PW = C1_TX + 2, PS = PW + 1, PH = TY + 2;
    constexpr int PLANE = PH * PS;
    constexpr int NLD = (CK * PH * PW + 255) / 256;
    __shared__ float patch[2][CK * PLANE + 1];
    const int tid = threadIdx.x;
    const int tx = tid & 63, ty = tid >> 6;
    const int x0 = blockIdx.x * C1_TX, y0 = blockIdx.y * TY;
    const long in_plane = (long)p.H * p.W;
    const int x = x0 + tx;
    const int nchunk = (p.Ci0 + p.Ci1 + CK - 1) / CK;

    unsigned voff[NLD];
    int lslot[NLD], lch[NLD];
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
        const int e = tid + 256 * i;
        const int c = e / (PH * PW), r = e - c * (PH * PW);
        const int py = r / PW, px = r - py * PW;
        const int gy = y0 - 1 + py, gx = x0 - 1 + px;
        const bool live = e < CK * PH * PW;
        const bool ok = live && gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
        voff[i] = ok ? (unsigned)(((long)c * in_plane + (long)gy * p.W + gx) * 4) : 0x80000000u;
        lslot[i] = live ? c * PLANE + py * PS + px : CK * PLANE;
        lch[i] = c;
    }
    float pv[NLD];
    auto issue = [&](int ch) {
        const int c0 = ch * CK;
        // a chunk comes from one tensor: the first input, or (two-input form, C_in0 % 8 == 0) the second
        const bool second = c0 >= p.Ci0;
        const int nc = second ? min(CK, p.Ci0 + p.Ci1 - c0) : min(CK, p.Ci0 - c0);  // channels of this chunk that exist
        const float* base = second ? p.in1 + (long)(c0 - p.Ci0) * in_plane : p.in0 + (long)c0 * in_plane;
        const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, (int)((long)nc * in_plane * 4), 0x00020000);
#pragma unroll
        for (int i = 0; i < NLD; ++i)
            pv[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, (int)voff[i], 0, 0));
    };
    auto land = [&](int buf) {
#pragma unroll
        for (int i = 0; i < NLD; ++i) patch[buf][lslot[i]] = pv[i];
    };

    float acc[YP][CO];
#pragma unroll
    for (int q = 0; q < YP; ++q)
#pragma unroll
        for (int o = 0; o < CO; ++o) acc[q][o] = 0.0f;
    typedef const float __attribute__((address_space(4))) cfloat;
    cfloat* wt = (cfloat*)p.weight;  // packed [C_in padded][3 ky][3 kx][CO]

    issue(0);
    land(0);
    __syncthreads();
    for (int ch = 0; ch < nchunk; ++ch) {
        const int buf = ch & 1;
        if (ch + 1 < nchunk) issue(ch + 1);
        const float* __restrict__ pl = patch[buf] + ty * PS + tx;
#pragma unroll 1
        for (int c = 0; c < CK; ++c) {
#pragma unroll 1
            for (int dy = 0; dy < 3; ++dy) {
                cfloat* w = wt + ((ch * CK + c) * 3 + dy) * (3 * CO);
                asm volatile("" : "+s"(w));
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) {
#pragma unroll
                    for (int q = 0; q < YP; ++q) {
                        const float v = pl[c * PLANE + (dy + 4 * q) * PS + dx];
#pragma unroll
                        for (int o = 0; o < CO; ++o) acc[q][o] = fmaf(v, w[dx * CO + o], acc[q][o]);
                    }
                }
            }
        }
        if (ch + 1 < nchunk) land(buf ^ 1);
        __syncthreads();
    }
#pragma unroll
    for (int q = 0; q < YP; ++q) {
        const int y = y0 + ty + 4 * q;
        if (x < p.W && y < p.H) {
#pragma unroll
            for (int o = 0; o < CO; ++o) {
                const long oidx = (long)o * in_plane + (long)y * p.W + x;
                if (p.act <= 1) {
                    p.out[oidx] = epilogue(acc[q][o], o, oidx, p);
                } else {
                    // ConvGRUCell (module.py:24-51) fused as in conv_stream.hip: act 2: y = sigmoid(y), reset-gate channels
                    // (< ep_split) times the state h = skip -> out = [r*h | u]; act 3: h' = u*h + (1-u)*tanh(y)
                    float yv = acc[q][o];
                    if (p.scale) yv *= p.scale[o];
                    if (p.shift) yv += p.shift[o];
                    if (p.act == 2) {
                        yv = gru_sigmoid_as<false>(yv);
                        if (o < p.ep_split) yv *= p.skip[oidx];
                    } else {
                        const float u = p.aux1[oidx], hv = p.skip[oidx];
                        yv = u * hv + (1.0f - u) * gru_tanh_as<false>(yv);
                    }
                    p.out[oidx] = yv;
                }
            }
        }
    }
    (void)lch;
}

static int launch_conv2d_stream(const ConvParams& p, hipStream_t stream) {
    const int gx = ceil_div(p.W, C1_TX), gy = ceil_div(p.H, 8);
    const long in_plane = (long)p.H * p.W;
    if (gy > 65535 || (long)8 * in_plane * 4 >= (1L << 31)) return D3D_ERR_UNSUPPORTED;
    if (p.Co == 8 && p.Ci0 + p.Ci1 <= 4)
        hipLaunchKernelGGL((conv2d_stream_kernel<8, 4>), dim3(gx, gy), dim3(256), 0, stream, p);
    else if (p.Co == 8)
        hipLaunchKernelGGL(conv2d_stream_kernel<8>, dim3(gx, gy), dim3(256), 0, stream, p);
    else if (p.Co == 16)
        hipLaunchKernelGGL(conv2d_stream_kernel<16>, dim3(gx, gy), dim3(256), 0, stream, p);
    else
        return D3D_ERR_UNSUPPORTED;
    D3D_LAUNCH_CHECK("conv2d_stream_kernel launch");
    return D3D_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// Lateral connection of the feature pyramid (module.py:736-747, FeatureNet_mvsnet "fpn"): out = conv1x1(x) + bias +
// nearest-x2 upsampling of the coarser level -- `F.interpolate(f, scale_factor=2, mode="nearest") + self.inner(x)`.
// The reference materialises the upsampled tensor (32 channels at full resolution: 653 MB written and read again); here
// the coarse level is read in place at (y/2, x/2).  Pure streaming: C_in + C_out/4 + C_out floats per pixel.
// ---------------------------------------------------------------------------------------------------------------
template <int CI, int CO>
__global__ __launch_bounds__(256) void conv1x1_upskip_kernel(const float* __restrict__ in, const float* wpacked,
                                                             const float* __restrict__ bias,
                                                             const float* __restrict__ coarse, int H, int W,
                                                             float* __restrict__ out) {
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (x >= W) return;
    const long plane = (long)H * W, idx = (long)y * W + x;
    const int hw = W >> 1;
    const long cplane = (long)(H >> 1) * hw, cidx = (long)(y >> 1) * hw + (x >> 1);
    float v[CI];
#pragma unroll
    for (int c = 0; c < CI; ++c) v[c] = in[c * plane + idx];
    float acc[CO];
#pragma unroll
    for (int o = 0; o < CO; ++o) acc[o] = bias ? bias[o] : 0.0f;
    typedef const float __attribute__((address_space(4))) cfloat;
    cfloat* wt = (cfloat*)wpacked;  // [CI][CO]
#pragma unroll
    for (int c = 0; c < CI; ++c) {
        cfloat* w = wt + c * CO;
        asm volatile("" : "+s"(w));
#pragma unroll
        for (int o = 0; o < CO; ++o) acc[o] = fmaf(v[c], w[o], acc[o]);
        __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int o = 0; o < CO; ++o) out[o * plane + idx] = acc[o] + coarse[o * cplane + cidx];
}

}  // namespace d3d

using namespace d3d;

extern "C" {

int d3d_conv1x1_upskip(const float* in, int Ci, const float* wpacked, const float* bias, const float* coarse, int Co, int H,
                       int W, float* out, d3d_stream_t stream) {
    D3D_REQUIRE(in && wpacked && coarse && out, "null pointer");
    D3D_REQUIRE(H > 0 && W > 0 && H % 2 == 0 && W % 2 == 0 && H <= 65535, "bad dims %dx%d (even sizes)", H, W);
    dim3 grid(ceil_div(W, 256), H);
    hipStream_t st = (hipStream_t)stream;
    if (Ci == 8 && Co == 32)
        hipLaunchKernelGGL((conv1x1_upskip_kernel<8, 32>), grid, dim3(256), 0, st, in, wpacked, bias, coarse, H, W, out);
    else if (Ci == 16 && Co == 32)
        hipLaunchKernelGGL((conv1x1_upskip_kernel<16, 32>), grid, dim3(256), 0, st, in, wpacked, bias, coarse, H, W, out);
    else {
        set_error("d3d_conv1x1_upskip: unsupported channels %d -> %d", Ci, Co);
        return D3D_ERR_UNSUPPORTED;
    }
    D3D_LAUNCH_CHECK("conv1x1_upskip_kernel launch");
    return D3D_OK;
}

int d3d_conv2d_k3_stream(const float* in0, int Ci0, const float* in1, int Ci1, const float* wpacked, const float* scale,
                         const float* shift, const float* skip, const float* aux1, int ep_split, int act, int Co, int H,
                         int W, float* out, d3d_stream_t stream) {
    D3D_REQUIRE(in0 && wpacked && out, "null pointer");
    D3D_REQUIRE(Ci0 > 0 && Ci1 >= 0 && (Ci1 == 0 || (in1 && Ci0 % 8 == 0)), "bad input channel split %d+%d", Ci0, Ci1);
    D3D_REQUIRE(H > 0 && W > 0, "bad dims");
    D3D_REQUIRE(act >= 0 && act <= 3, "bad act %d", act);
    D3D_REQUIRE(act < 2 || skip, "act %d needs the state h in `skip`", act);
    D3D_REQUIRE(act != 3 || aux1, "act 3 needs the update gate in `aux1`");
    ConvParams p = {};
    p.in0 = in0; p.in1 = in1; p.weight = wpacked; p.scale = scale; p.shift = shift; p.skip = skip; p.out = out;
    p.aux1 = aux1; p.ep_split = ep_split;
    p.Ci0 = Ci0; p.Ci1 = Ci1; p.Co = Co; p.D = 1; p.H = H; p.W = W; p.Do = 1; p.Ho = H; p.Wo = W;
    p.stride = 1; p.act = act; p.skip_after_act = 1;
    int rc = launch_conv2d_stream(p, (hipStream_t)stream);
    if (rc == D3D_ERR_UNSUPPORTED) set_error("d3d_conv2d_k3_stream: unsupported shape Ci=%d+%d Co=%d %dx%d", Ci0, Ci1, Co, H, W);
    return rc;
}

int d3d_convtranspose3d_k3s2_co8(const float* in, const float* wpacked, const float* scale, const float* shift,
                                 const float* skip, int relu, int Ci, int D, int H, int W, float* out,
                                 d3d_stream_t stream) {
    D3D_REQUIRE(in && wpacked && out, "null pointer");
    D3D_REQUIRE(Ci > 0 && D > 0 && H > 0 && W > 0, "bad dims");
    ConvParams p = {};
    p.in0 = in; p.weight = wpacked; p.scale = scale; p.shift = shift; p.skip = skip; p.out = out;
    p.Ci0 = Ci; p.Co = 8; p.D = D; p.H = H; p.W = W; p.Do = 2 * D; p.Ho = 2 * H; p.Wo = 2 * W;
    p.stride = 2; p.act = relu ? 1 : 0; p.skip_after_act = 1;
    int rc = launch_convT3d_co8(p, (hipStream_t)stream);
    if (rc == D3D_ERR_UNSUPPORTED) set_error("d3d_convtranspose3d_k3s2_co8: unsupported shape Ci=%d %dx%dx%d", Ci, D, H, W);
    return rc;
}

int d3d_conv3d_k3_co8(const float* in, const float* wpacked, const float* scale, const float* shift, const float* skip,
                      int relu, int Ci, int D, int H, int W, float* out, d3d_stream_t stream) {
    D3D_REQUIRE(in && wpacked && out, "null pointer");
    D3D_REQUIRE(Ci > 0 && D > 0 && H > 0 && W > 0, "bad dims");
    ConvParams p = {};
    p.in0 = in; p.weight = wpacked; p.scale = scale; p.shift = shift; p.skip = skip; p.out = out;
    p.Ci0 = Ci; p.Co = 8; p.D = D; p.H = H; p.W = W; p.Do = D; p.Ho = H; p.Wo = W;
    p.stride = 1; p.act = relu ? 1 : 0; p.skip_after_act = 1;
    int rc = launch_conv3d_co8(p, (hipStream_t)stream);
    if (rc == D3D_ERR_UNSUPPORTED) set_error("d3d_conv3d_k3_co8: unsupported shape Ci=%d %dx%dx%d", Ci, D, H, W);
    return rc;
}

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
    if (Co == 1 && stride == 1 && Ci == 8) {  // streaming single-channel form
        bool co1 = true;
#ifdef D3D_EXPERIMENTS
        if (const char* e = getenv("D3D_CONV_CO1")) co1 = atoi(e) != 0;   // 0: generic kernel
#endif
        if (co1) {
            int rc = launch_conv3d_co1(p, (hipStream_t)stream);
            if (rc != D3D_ERR_UNSUPPORTED) return rc;
        }
    }
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
