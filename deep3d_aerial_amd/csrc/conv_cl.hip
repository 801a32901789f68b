// Channel-last bf16 activations for the CostRegNet layers in bf16 mode (BASELINE config 3):
//   * 3x3x3 STRIDE-2 convolution + folded BN + ReLU (conv1 / conv3 of CostRegNet, cas_mvsnet.py:86,89: 8 -> 16, 16 -> 32) on
//     v_mfma_f32_16x16x32_bf16, z-streaming, channel-last bf16 in and out ([D][H][W][C], see conv_c8.hip);
//   * the two format conversions (planar fp32 <-> channel-last bf16) for the layers that stay on the planar kernels
//     (conv5 / conv6, 1/64 .. 1/512 of the voxels) and for tests.
//
// Stride-2 kernel.  Output (z, y, x) reads inputs (2z + k_z - 1, 2y + k_y - 1, 2x + k_x - 1):
//   * a workgroup (8 waves) owns 32 x 8 OUTPUT pixels and walks the INPUT planes of its z segment, each staged once: an
//     even plane 2a feeds output plane a through k_z = 1, an odd plane 2a + 1 feeds plane a (k_z = 2, which completes it)
//     and plane a + 1 (k_z = 0): two accumulator sets alternate;
//   * the staged patch (65 x 17 input pixels) keeps the even and the odd columns of a row in separate runs, so the 16
//     pixels of an A operand (input columns 2m + k_x - 1) are 16 CONSECUTIVE cells: one ds_read_b128 per lane, cell pitch
//     an odd number of 16-byte slots, no bank conflicts;
//   * weights are the A operand (the fragment layouts of A and B are the same), so D row = channel, column = pixel: a lane
//     leaves with four consecutive channels of one pixel = one 8-byte store into the pixel's cell.
#include "common.h"

namespace d3d {

namespace {

typedef float f4 __attribute__((ext_vector_type(4)));
typedef unsigned u4 __attribute__((ext_vector_type(4)));

constexpr int TYO = 8;                 // output rows of a workgroup = waves (output columns: 16 * MGK)
constexpr int PYI = 2 * TYO + 1;       // staged input patch rows
constexpr int NT = 64 * TYO;

struct S2Params {
    const void* in;       // [D, H, W, CI] bf16
    const u4* wpk;        // [3 (kz)][NKB][N tiles][64 lanes] fragments (ops._pack_c8_bf16)
    const float* scale;   // [CO] or null
    const float* shift;   // [CO] or null
    const void* skip;     // [Do, Ho, Wo, CO] bf16 or null (added after the activation)
    void* out;            // [Do, Ho, Wo, CO] bf16
    int D, H, W;          // input dims
    int Do, Ho, Wo, CO;
    int relu;
    int zper;             // output planes per workgroup
};

__device__ __forceinline__ unsigned pack_h16_cl(float a, float b) {
    return pack_h16x2(a, b);   // one packed conversion (common.h: pack_h16x2)
}
__device__ __forceinline__ f4 unpack_h16x4_cl(uint2 u) {
    return (f4){h16_lo(u.x), h16_hi(u.x),
                h16_lo(u.y), h16_hi(u.y)};
}

// MGK: 16-pixel groups per wave (output tile width 16 * MGK); WG: weight fragments read from global memory (L2) per use
// (32 -> 64: 110 KB of weights do not fit beside the double-buffered patch)
template <int CI, int NTN, int MGK = 2, bool WG = false>
__global__ __launch_bounds__(NT) void conv3d_s2_cl_kernel(S2Params p) {
    constexpr int TXO = 16 * MGK, MG = MGK;          // output tile width, 16-pixel groups per wave (= output row)
    constexpr int PXI = 2 * TXO + 1, NEVEN = TXO + 1;   // staged patch columns; the even ones come first in a row, then the TXO odd ones
    constexpr int NKB = (9 * CI + 31) / 32;
    constexpr int CS = bf16_cell_bytes(CI);   // (bank-conflict-free pitch: common.h)
    constexpr int G = CI / 8;
    constexpr int PATCH = PXI * PYI * CS;
    constexpr int AW = MG * NTN;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    u4* wlds = reinterpret_cast<u4*>(smem + 2 * PATCH);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int bxi = blockIdx.x, byi = blockIdx.y, bzi = blockIdx.z;   // (XCD-contiguous tile order: see conv_c8.hip)
#ifdef D3D_S2_XCD   // (measured: no change -- off)
    {
        const int nx = gridDim.x, ny = gridDim.y, n = nx * ny * gridDim.z;
        if (n % 8 == 0) {
            int lin = (bzi * ny + byi) * nx + bxi;
            lin = (lin % 8) * (n / 8) + lin / 8;
            bxi = lin % nx; byi = (lin / nx) % ny; bzi = lin / (nx * ny);
        }
    }
#endif
    const int xo0 = bxi * TXO, yo0 = byi * TYO;
    const int zo0 = bzi * p.zper, zo1 = min(zo0 + p.zper, p.Do);
    const int D = p.D, H = p.H, W = p.W;

    if constexpr (!WG)
        for (int i = tid; i < 3 * NKB * NTN * 64; i += NT) wlds[i] = p.wpk[i];
    const u4* __restrict__ wsrc = WG ? p.wpk : wlds;

    // ---- staging: task = (patch pixel, 8-channel group), one 16-byte load and one 16-byte LDS write ----------------
    constexpr int NTASK = PXI * PYI * G;
    constexpr int ROUNDS = (NTASK + NT - 1) / NT;
    u4 stc[ROUNDS];
    // What a task reads and writes is the same for every plane: per-lane state computed once (these kernels are bound by
    // instruction issue -- profiles/r04_t2p.txt -- and the per-plane address arithmetic was most of their vector instructions).
    // Loads are clamped into the image and always issued; the value is zeroed outside.
    unsigned stoff[ROUNDS];   // byte offset of the task's 16 bytes inside an input plane (host: a plane is < 2^31 bytes)
    int stdst[ROUNDS];        // its cell in the patch, -1: no task
    bool stok[ROUNDS];
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {
        const int task = tid + r * NT;
        const int pix = task / G, g = task - pix * G;
        const int py = pix / PXI, px = pix - py * PXI;
        const int gx = 2 * xo0 - 1 + px, gy = 2 * yo0 - 1 + py;
        stok[r] = task < NTASK && gx >= 0 && gx < W && gy >= 0 && gy < H;
        stoff[r] = task >= NTASK ? 0 : ((unsigned)min(max(gy, 0), H - 1) * W + min(max(gx, 0), W - 1)) * (CI * 2) + g * 16;   // (no task: every such lane reads the plane's first bytes -- one line -- instead of real cells beyond the patch)
        const int cell = py * PXI + ((px & 1) ? NEVEN + (px >> 1) : (px >> 1));
        stdst[r] = task < NTASK ? cell * CS + g * 16 : -1;
    }
    const size_t iplane = (size_t)H * W * (CI * 2);
    bool stzin = false;   // the plane in the staging registers is inside the volume
    auto issue = [&](int zi) {
        const bool zin = zi >= 0 && zi < D;
        stzin = zin;
        if (!zin) return;   // (uniform) a plane outside the volume: nothing loaded, the commit writes zeros
        const unsigned char* __restrict__ src = static_cast<const unsigned char*>(p.in) + (size_t)min(max(zi, 0), D - 1) * iplane;
#pragma unroll
        for (int r = 0; r < ROUNDS; ++r) {
            const u4 v = *reinterpret_cast<const u4*>(src + stoff[r]);
            stc[r] = v;   // raw: zeroed for cells outside the volume when it is committed -- a select here would wait for the load
        }
    };
    auto commit = [&](unsigned char* dst) {
#pragma unroll
        for (int r = 0; r < ROUNDS; ++r)
            if (stdst[r] >= 0) *reinterpret_cast<u4*>(dst + stdst[r]) = stzin && stok[r] ? stc[r] : (u4){0, 0, 0, 0};
    };

    // K index k = 32 kb + 8 (lane >> 4) + j -> tap t = k / CI = (k_y, k_x), channel k % CI.  Output pixel m of the row reads
    // patch row 2 wave + k_y, patch column 2 m + k_x: even run index m (k_x = 0) | m + 1 (k_x = 2), odd run index m (k_x = 1)
    auto a_offset = [&](int kb, int kgroup) {
        const int k0 = 32 * kb + 8 * kgroup;
        const int t = k0 / CI, c = k0 % CI;
        const int ky = t < 9 ? t / 3 : 0, kx = t < 9 ? t % 3 : 0;   // padded taps read a valid cell; their weights are zero
        const int col = kx == 1 ? NEVEN : (kx >> 1);
        return (ky * PXI + col) * CS + (t < 9 ? c : 0) * 2;
    };
    const int abase = (2 * wave * PXI + (lane & 15)) * CS;
    int aoffs[WG ? 1 : NKB];   // one register per K block, computed once (WG: tiny volumes, per use)
    if constexpr (!WG) {
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb) aoffs[kb] = a_offset(kb, lane >> 4);
    }

    f4 acc[2][AW];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int i = 0; i < AW; ++i) acc[s][i] = (f4){0, 0, 0, 0};

    const int oy = yo0 + wave;
    // plane-invariant state of the epilogue: affine per channel tile, offsets inside an output plane and store masks per pixel group
    f4 esc[NTN], esh[NTN];
    bool est[NTN][MG];
    unsigned eoff[MG];   // bf16 elements inside an output plane, channel tile 0 (host: a plane is < 2^31 elements)
#pragma unroll
    for (int nt = 0; nt < NTN; ++nt) {
        const int cb = nt * 16 + (lane >> 4) * 4;
        const bool cin = cb < p.CO;
        esc[nt] = p.scale && cin ? *reinterpret_cast<const f4*>(p.scale + cb) : (f4){1, 1, 1, 1};
        esh[nt] = p.shift && cin ? *reinterpret_cast<const f4*>(p.shift + cb) : (f4){0, 0, 0, 0};
#pragma unroll
        for (int mg = 0; mg < MG; ++mg) est[nt][mg] = cin && xo0 + mg * 16 + (lane & 15) < p.Wo;
    }
#pragma unroll
    for (int mg = 0; mg < MG; ++mg)
        eoff[mg] = ((unsigned)min(oy, p.Ho - 1) * p.Wo + min(xo0 + mg * 16 + (lane & 15), p.Wo - 1)) * p.CO + (lane >> 4) * 4;
    const size_t oplane = (size_t)p.Ho * p.Wo * p.CO;
    auto store_plane = [&](int zo, f4 (&a)[AW]) {
        if (oy < p.Ho && zo >= zo0) {
            const unsigned short* __restrict__ sk = static_cast<const unsigned short*>(p.skip) + (size_t)zo * oplane;
            unsigned short* __restrict__ dst = static_cast<unsigned short*>(p.out) + (size_t)zo * oplane;
#pragma unroll
            for (int nt = 0; nt < NTN; ++nt) {
#pragma unroll
                for (int mg = 0; mg < MG; ++mg) {
                    if (est[nt][mg]) {
                        const unsigned o = eoff[mg] + nt * 16;
                        f4 v = a[mg * NTN + nt] * esc[nt] + esh[nt];
                        if (p.relu) v = __builtin_elementwise_max(v, (f4){0, 0, 0, 0});
                        if (p.skip) v += unpack_h16x4_cl(*reinterpret_cast<const uint2*>(sk + o));
                        const uint2 pk = {pack_h16_cl(v[0], v[1]), pack_h16_cl(v[2], v[3])};
                        *reinterpret_cast<uint2*>(dst + o) = pk;
                    }
                }
            }
        }
#pragma unroll
        for (int i = 0; i < AW; ++i) a[i] = (f4){0, 0, 0, 0};
    };

    // one k_z slice (KZ2 < 0) or two (k_z = KZ1 -> d1, k_z = KZ2 -> d2) of the weights over the staged plane
    auto sweep = [&](const unsigned char* buf, int kz1, f4 (&d1)[AW], int kz2, f4 (&d2)[AW], bool two) {
        int kgroup = lane >> 4;
        asm volatile("" : "+v"(kgroup));
auto kb_body = [&](int kb) {
            const int aoffk = WG ? a_offset(kb, kgroup) : aoffs[WG ? 0 : kb];
            h16x8 w1[NTN], w2[NTN];
#pragma unroll
            for (int nt = 0; nt < NTN; ++nt) {
                w1[nt] = __builtin_bit_cast(h16x8, wsrc[((kz1 * NKB + kb) * NTN + nt) * 64 + lane]);
                if (two) w2[nt] = __builtin_bit_cast(h16x8, wsrc[((kz2 * NKB + kb) * NTN + nt) * 64 + lane]);
            }
#pragma unroll
            for (int mg = 0; mg < MG; ++mg) {
                const h16x8 a = __builtin_bit_cast(h16x8, *reinterpret_cast<const u4*>(buf + abase + mg * 16 * CS + aoffk));
#pragma unroll
                for (int nt = 0; nt < NTN; ++nt) {
                    d1[mg * NTN + nt] = mfma_h16(w1[nt], a, d1[mg * NTN + nt]);
                    if (two) d2[mg * NTN + nt] = mfma_h16(w2[nt], a, d2[mg * NTN + nt]);
                }
            }
        };
        if constexpr (WG) {   // fragments come from L2: one K block's loads in flight at a time (registers)
#pragma unroll 1
            for (int kb = 0; kb < NKB; ++kb) kb_body(kb);
        } else {
#pragma unroll
            for (int kb = 0; kb < NKB; ++kb) kb_body(kb);
        }
    };

    // ---- walk the input planes 2 zo0 - 1 .. 2 zo1 - 1 (the last one is odd) -----------------------------------------
    const int zlast = 2 * zo1 - 1;
    int zi = 2 * zo0 - 1, cur = 0;
    issue(zi);
    commit(smem);
    __syncthreads();
    auto odd_step = [&](f4 (&lo)[AW], f4 (&hi)[AW]) {   // zi = 2a + 1: completes output plane a (lo), opens plane a + 1 (hi)
        const bool more = zi < zlast, live = zi >= 0 && zi < D;
        if (more) issue(zi + 1);
        if (live) sweep(smem + cur * PATCH, 2, lo, 0, hi, true);
        if (more) commit(smem + (cur ^ 1) * PATCH);   // before the stores: waiting for the staged loads must not wait for them too
        store_plane((zi - 1) >> 1, lo);
        lds_barrier();                                // (LDS only: the stores stay in flight, common.h)
        cur ^= 1;
    };
    auto even_step = [&](f4 (&mid)[AW]) {               // zi = 2a: k_z = 1 of output plane a; never the last plane
        const bool live = zi < D;
        issue(zi + 1);
        if (live) sweep(smem + cur * PATCH, 1, mid, 1, mid, false);
        commit(smem + (cur ^ 1) * PATCH);
        lds_barrier();
        cur ^= 1;
    };
    while (true) {
        odd_step(acc[1], acc[0]);
        if (zi == zlast) break;
        ++zi;
        even_step(acc[0]);
        ++zi;
        odd_step(acc[0], acc[1]);
        if (zi == zlast) break;
        ++zi;
        even_step(acc[1]);
        ++zi;
    }
}

template <int CI, int NTN, int MGK = 2, bool WG = false>
static int launch_s2(const S2Params& p, hipStream_t stream) {
    constexpr int NKB = (9 * CI + 31) / 32;
    constexpr int CS = bf16_cell_bytes(CI);
    constexpr int TXOk = 16 * MGK, PXIk = 2 * TXOk + 1;
    const int lds = 2 * PXIk * PYI * CS + (WG ? 0 : 3 * NKB * NTN * 64 * 16);
    auto kern = conv3d_s2_cl_kernel<CI, NTN, MGK, WG>;
    int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(kern), lds);
    if (rc != D3D_OK) return rc;
    S2Params q = p;
    const int gx = ceil_div(p.Wo, TXOk), gy = ceil_div(p.Ho, TYO);
    q.zper = pick_zper((long)gx * gy, p.Do, 2, 1, lds);   // (every z segment re-reads one halo plane)
    hipLaunchKernelGGL(kern, dim3(gx, gy, ceil_div(p.Do, q.zper)), dim3(NT), lds, stream, q);
    D3D_LAUNCH_CHECK("conv3d_s2_cl_kernel launch");
    return D3D_OK;
}

// ---- format conversions: thread = (voxel, 8-channel group) ---------------------------------------------------------
__global__ __launch_bounds__(256) void planar_to_cl_kernel(const float* __restrict__ in, int C, size_t n, u4* __restrict__ out) {
    const size_t v = (size_t)blockIdx.x * 256 + threadIdx.x;
    const int g = blockIdx.y, G = C >> 3;
    if (v >= n) return;
    const float* src = in + (size_t)(8 * g) * n + v;
    float f[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) f[k] = src[(size_t)k * n];
    out[v * G + g] = (u4){pack_h16_cl(f[0], f[1]), pack_h16_cl(f[2], f[3]), pack_h16_cl(f[4], f[5]), pack_h16_cl(f[6], f[7])};
}

__global__ __launch_bounds__(256) void cl_to_planar_kernel(const u4* __restrict__ in, int C, size_t n, float* __restrict__ out) {
    const size_t v = (size_t)blockIdx.x * 256 + threadIdx.x;
    const int g = blockIdx.y, G = C >> 3;
    if (v >= n) return;
    const u4 u = in[v * G + g];
    float* dst = out + (size_t)(8 * g) * n + v;
    const unsigned w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        dst[(size_t)(2 * k) * n] = h16_lo(w[k]);
        dst[(size_t)(2 * k + 1) * n] = h16_hi(w[k]);
    }
}

}  // namespace

}  // namespace d3d

using namespace d3d;

extern "C" int d3d_conv3d_k3s2_cl_h16(const void* in, const void* wpacked, const float* scale, const float* shift, const void* skip,
                                       int relu, int Ci, int Co, int D, int H, int W, void* out, d3d_stream_t stream) {
    D3D_REQUIRE(in && wpacked && out, "null pointer");
    D3D_REQUIRE(D > 0 && H > 0 && W > 0, "bad dims %dx%dx%d", D, H, W);
    S2Params p = {};
    p.in = in; p.wpk = reinterpret_cast<const u4*>(wpacked); p.scale = scale; p.shift = shift; p.skip = skip; p.out = out;
    p.D = D; p.H = H; p.W = W; p.Do = (D - 1) / 2 + 1; p.Ho = (H - 1) / 2 + 1; p.Wo = (W - 1) / 2 + 1; p.CO = Co; p.relu = relu;
    const bool shape = (Ci == 8 && Co == 16) || (Ci == 16 && Co == 32) || (Ci == 8 && Co == 8) || (Ci == 16 && Co == 16) ||
                       (Ci == 32 && Co == 64);
    if (!shape || ceil_div(p.Ho, TYO) > 65535 || p.Do > 65535 || (long)H * W * Ci * 2 >= (1L << 31)) {   // (32-bit offsets inside a plane)
        set_error("d3d_conv3d_k3s2_cl_h16: %d -> %d channels not taken (8->8, 8->16, 16->16, 16->32, 32->64)", Ci, Co);
        return D3D_ERR_UNSUPPORTED;
    }
    hipStream_t st = (hipStream_t)stream;
    if (Ci == 32) return launch_s2<32, 4, 1, true>(p, st);   // conv5: 16-wide tiles, weights streamed from L2
    if (Ci == 8) return launch_s2<8, 1>(p, st);
    return Co > 16 ? launch_s2<16, 2>(p, st) : launch_s2<16, 1>(p, st);
}

extern "C" int d3d_volume_planar_to_cl_h16(const float* in, int C, size_t n, void* out, d3d_stream_t stream) {
    D3D_REQUIRE(in && out, "null pointer");
    D3D_REQUIRE(C > 0 && C % 8 == 0 && C <= 8 * 65535 && n > 0 && n < ((size_t)1 << 39), "C = %d (multiple of 8), n = %zu", C, n);
    hipLaunchKernelGGL(planar_to_cl_kernel, dim3((unsigned)((n + 255) / 256), C / 8), dim3(256), 0, (hipStream_t)stream, in, C, n,
                       static_cast<u4*>(out));
    D3D_LAUNCH_CHECK("planar_to_cl_kernel launch");
    return D3D_OK;
}

extern "C" int d3d_volume_cl_h16_to_planar(const void* in, int C, size_t n, float* out, d3d_stream_t stream) {
    D3D_REQUIRE(in && out, "null pointer");
    D3D_REQUIRE(C > 0 && C % 8 == 0 && C <= 8 * 65535 && n > 0 && n < ((size_t)1 << 39), "C = %d (multiple of 8), n = %zu", C, n);
    hipLaunchKernelGGL(cl_to_planar_kernel, dim3((unsigned)((n + 255) / 256), C / 8), dim3(256), 0, (hipStream_t)stream,
                       static_cast<const u4*>(in), C, n, out);
    D3D_LAUNCH_CHECK("cl_to_planar_kernel launch");
    return D3D_OK;
}
