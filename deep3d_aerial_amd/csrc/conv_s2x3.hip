// 3x3x3 STRIDE-2 convolution + folded BN + ReLU in fp32 accuracy on the bf16 matrix cores: conv1 / conv3 / conv5 of every CostRegNet
// (cas_mvsnet.py:86,89,92: 8 -> 16, 16 -> 32, 32 -> 64) in fp32 mode -- the reference's own precision -- VERDICT r03 item 7.
// Until round 4 these three layers ran on the vector-unit stream kernels: 5.2 ms of a 31 ms CasMVSNet view.
//
// The scheme is conv_cl.hip's stride-2 kernel (bf16 mode) with conv_c8.hip's split operands:
//   * planar fp32 tensors in and out ([C][D][H][W]); every input value is staged as the exact sum of three bf16 numbers
//     (hi = rne(v), mid = rne(v - hi), lo = rne(v - hi - mid)), the host packs the weights' three parts (ops._pack_c8_bf16x3),
//     and a K block of 32 takes the six products whose weight is >= 2^-16 of the leading one, small terms first: the error of
//     the fp32 instruction at 6 x 16 instead of 8 x 32 cycles per K block;
//   * a workgroup (8 waves) owns 16 MGK x 8 OUTPUT pixels and walks the INPUT planes of its z segment, each staged once into
//     ONE patch buffer (split cells are three times the bf16 ones): an even plane 2a feeds output plane a through k_z = 1, an
//     odd plane 2a + 1 completes plane a (k_z = 2) and opens plane a + 1 (k_z = 0) -- two accumulator sets alternate;
//   * the staged patch keeps the even and the odd columns of a row in separate runs, so the 16 pixels of an A operand (input
//     columns 2 m + k_x - 1) are 16 consecutive cells; cell pitches 48 | 96 | 224 bytes (8 | 16 | 32 channels: the hi | mid |
//     lo runs of a cell, conflict-free for gfx950's ds_read_b128 lane groups, tools/conv_bank_sim.py);
//   * pixels are the A operand: D row = pixel (lane >> 4) * 4 + register, column = channel -- a lane leaves with four
//     consecutive pixels of one channel = one 16-byte store into the planar output;
//   * everything plane-invariant (staging offsets, cells, K-block operand offsets, epilogue offsets) is per-lane state
//     computed once; staged values are zeroed when they are committed, not behind the load.
#include <cstdint>
#include "common.h"

namespace d3d {

namespace {

typedef float f4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u4 __attribute__((ext_vector_type(4)));

constexpr int TYO = 8;                 // output rows of a workgroup = waves (output columns: 16 * MGK)
constexpr int PYI = 2 * TYO + 1;       // staged input patch rows
constexpr int NT = 64 * TYO;

struct S2XParams {
    const float* in;      // [CI, D, H, W]
    const u4* wpk;        // [3 parts][3 (kz)][NKB][N tiles][64 lanes] fragments (ops._pack_c8_bf16x3)
    const float* scale;   // [CO] or null
    const float* shift;   // [CO] or null
    const float* skip;    // [CO, Do, Ho, Wo] or null (added after the activation)
    float* out;           // [CO, Do, Ho, Wo]
    int D, H, W;          // input dims
    int Do, Ho, Wo, CO;
    int relu;
    int zper;             // output planes per workgroup
};

constexpr int s2x_cell_bytes(int CI) { return CI == 8 ? 48 : CI == 16 ? 96 : 224; }

// MGK: 16-pixel groups per wave (output tile width 16 * MGK); WG: weight fragments read from global memory (L2) per use
template <int CI, int NTN, int MGK, bool WG>
__global__ __launch_bounds__(NT) void conv3d_s2_x3_kernel(S2XParams p) {
    constexpr int TXO = 16 * MGK;
    constexpr int PXI = 2 * TXO + 1, NEVEN = TXO + 1;   // staged patch columns; the even ones come first in a row, then the TXO odd ones
    constexpr int NKB = (9 * CI + 31) / 32;
    constexpr int CS = s2x_cell_bytes(CI);
    constexpr int G = CI / 8;
    constexpr int PATCH = PXI * PYI * CS;
    constexpr int AW = MGK * NTN;
    constexpr int WS = 3 * NKB * NTN * 64;              // fragments per weight part
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    u4* wlds = reinterpret_cast<u4*>(smem + PATCH);

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int xo0 = blockIdx.x * TXO, yo0 = blockIdx.y * TYO;
    const int zo0 = blockIdx.z * p.zper, zo1 = min(zo0 + p.zper, p.Do);
    const int D = p.D, H = p.H, W = p.W;
    const size_t plane = (size_t)H * W, vol = (size_t)D * plane;
    const size_t oplane = (size_t)p.Ho * p.Wo, ovol = (size_t)p.Do * oplane;

    if constexpr (!WG)
        for (int i = tid; i < 3 * WS; i += NT) wlds[i] = p.wpk[i];
    const u4* __restrict__ wsrc = WG ? p.wpk : wlds;

    // ---- staging: task = (patch pixel, 8-channel group): eight dword loads (channel stride = the volume), three 16-byte writes --
    constexpr int NTASK = PXI * PYI * G;
    constexpr int ROUNDS = (NTASK + NT - 1) / NT;
    float stg[ROUNDS][8];
    size_t poff[ROUNDS];   // element offset of the task's first channel inside an input plane (clamped into the image)
    int pdst[ROUNDS];      // its cell (hi run) in the patch, -1: no task
    bool pok[ROUNDS];
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {
        const int task = tid + r * NT;
        const int pix = task / G, g = task - pix * G;
        const int py = pix / PXI, px = pix - py * PXI;
        const int gx = 2 * xo0 - 1 + px, gy = 2 * yo0 - 1 + py;
        pok[r] = task < NTASK && gx >= 0 && gx < W && gy >= 0 && gy < H;
        poff[r] = task < NTASK ? (size_t)(8 * g) * vol + (size_t)min(max(gy, 0), H - 1) * W + min(max(gx, 0), W - 1) : 0;
        const int cell = py * PXI + ((px & 1) ? NEVEN + (px >> 1) : (px >> 1));
        pdst[r] = task < NTASK ? cell * CS + g * 16 : -1;
    }
    bool stzin = false;   // the plane in the staging registers is inside the volume
    auto issue = [&](int zi) {
        stzin = zi >= 0 && zi < D;
        if (!stzin) return;   // (uniform) nothing loaded, the commit writes zeros
        const float* __restrict__ srcp = p.in + (size_t)zi * plane;
#pragma unroll
        for (int r = 0; r < ROUNDS; ++r) {
            const float* __restrict__ src = srcp + poff[r];
#pragma unroll
            for (int k = 0; k < 8; ++k) stg[r][k] = src[(size_t)k * vol];   // raw: zeroed when committed
        }
    };
    auto commit = [&]() {
#pragma unroll
        for (int r = 0; r < ROUNDS; ++r) {
            if (pdst[r] < 0) continue;
            const bool ok = stzin && pok[r];
            unsigned hi[4], mi[4], lo[4];
#pragma unroll
            for (int k = 0; k < 8; k += 2) {   // hi = bf16(x), mid = bf16(x - hi), lo = bf16(x - hi - mid): the differences are exact in fp32
                const float a = ok ? stg[r][k] : 0.0f, b = ok ? stg[r][k + 1] : 0.0f;
                hi[k >> 1] = pack_bf16x2(a, b);
                const float a1 = a - __builtin_bit_cast(float, hi[k >> 1] << 16), b1 = b - __builtin_bit_cast(float, hi[k >> 1] & 0xffff0000u);
                mi[k >> 1] = pack_bf16x2(a1, b1);
                const float a2 = a1 - __builtin_bit_cast(float, mi[k >> 1] << 16), b2 = b1 - __builtin_bit_cast(float, mi[k >> 1] & 0xffff0000u);
                lo[k >> 1] = pack_bf16x2(a2, b2);
            }
            unsigned char* cell = smem + pdst[r];
            *reinterpret_cast<u4*>(cell) = (u4){hi[0], hi[1], hi[2], hi[3]};
            *reinterpret_cast<u4*>(cell + CI * 2) = (u4){mi[0], mi[1], mi[2], mi[3]};
            *reinterpret_cast<u4*>(cell + CI * 4) = (u4){lo[0], lo[1], lo[2], lo[3]};
        }
    };

    // K index k = 32 kb + 8 (lane >> 4) + j -> tap t = k / CI = (k_y, k_x), channel k % CI.  Output pixel m of the row reads
    // patch row 2 wave + k_y, patch column 2 m + k_x: even run index m (k_x = 0) | m + 1 (k_x = 2), odd run index m (k_x = 1)
    int aoffs[NKB];
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb) {
        const int k0 = 32 * kb + 8 * (lane >> 4);
        const int t = k0 / CI, c = k0 % CI;
        const int ky = t < 9 ? t / 3 : 0, kx = t < 9 ? t % 3 : 0;   // padded taps read a valid cell; their weights are zero
        const int col = kx == 1 ? NEVEN : (kx >> 1);
        aoffs[kb] = ((2 * wave + ky) * PXI + col + (lane & 15)) * CS + (t < 9 ? c : 0) * 2;
    }

    f4 acc[2][AW];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int i = 0; i < AW; ++i) acc[s][i] = (f4){0, 0, 0, 0};

    // ---- epilogue state: D row (pixel) = (lane >> 4) * 4 + register, column (channel) = lane & 15 ------------------------------
    const int oy = yo0 + wave;
    float esc[NTN], esh[NTN];
    size_t eoff[NTN];     // element offset of the lane's first pixel quad inside an output plane + channel * volume
    bool eok[NTN][MGK];
#pragma unroll
    for (int nt = 0; nt < NTN; ++nt) {
        const int co = nt * 16 + (lane & 15);
        const bool cin = co < p.CO;
        esc[nt] = p.scale && cin ? p.scale[co] : 1.0f;
        esh[nt] = p.shift && cin ? p.shift[co] : 0.0f;
        eoff[nt] = (size_t)min(co, p.CO - 1) * ovol + (size_t)min(oy, p.Ho - 1) * p.Wo + xo0 + (lane >> 4) * 4;
#pragma unroll
        for (int mg = 0; mg < MGK; ++mg) eok[nt][mg] = cin && oy < p.Ho && xo0 + mg * 16 + (lane >> 4) * 4 < p.Wo;   // Wo % 4 == 0
    }
    auto store_plane = [&](int zo, f4 (&a)[AW]) {
        if (zo >= zo0) {
            const float* __restrict__ sk = p.skip ? p.skip + (size_t)zo * oplane : nullptr;
            float* __restrict__ dst = p.out + (size_t)zo * oplane;
#pragma unroll
            for (int nt = 0; nt < NTN; ++nt)
#pragma unroll
                for (int mg = 0; mg < MGK; ++mg) {
                    if (eok[nt][mg]) {
                        const size_t o = eoff[nt] + mg * 16;
                        f4 v = a[mg * NTN + nt] * esc[nt] + esh[nt];
                        if (p.relu) v = __builtin_elementwise_max(v, (f4){0, 0, 0, 0});
                        if (sk) v += *reinterpret_cast<const f4*>(sk + o);
                        *reinterpret_cast<f4*>(dst + o) = v;
                    }
                }
        }
#pragma unroll
        for (int i = 0; i < AW; ++i) a[i] = (f4){0, 0, 0, 0};
    };

    // one k_z slice or two (k_z = kz1 -> d1, k_z = kz2 -> d2) of the weights over the staged plane
    auto sweep = [&](int kz1, f4 (&d1)[AW], int kz2, f4 (&d2)[AW], bool two) {
        auto kb_body = [&](int kb) {
            const unsigned char* ap = smem + aoffs[kb];
#pragma unroll
            for (int mg = 0; mg < MGK; ++mg) {
                bf16x8 a[3];
#pragma unroll
                for (int sp = 0; sp < 3; ++sp) a[sp] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u4*>(ap + mg * 16 * CS + sp * CI * 2));
#pragma unroll
                for (int nt = 0; nt < NTN; ++nt) {
#pragma unroll
                    for (int half = 0; half < 2; ++half) {
                        if (half == 1 && !two) break;
                        const int wi = (((half ? kz2 : kz1) * NKB + kb) * NTN + nt) * 64 + lane;
                        const bf16x8 bh = __builtin_bit_cast(bf16x8, wsrc[wi]);
                        const bf16x8 bm = __builtin_bit_cast(bf16x8, wsrc[WS + wi]);
                        const bf16x8 bl = __builtin_bit_cast(bf16x8, wsrc[2 * WS + wi]);
                        f4 c = half ? d2[mg * NTN + nt] : d1[mg * NTN + nt];   // small terms first
                        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[2], bh, c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], bl, c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1], bm, c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1], bh, c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], bm, c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], bh, c, 0, 0, 0);
                        if (half) d2[mg * NTN + nt] = c; else d1[mg * NTN + nt] = c;
                    }
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

    // ---- walk the input planes 2 zo0 - 1 .. 2 zo1 - 1 (the last one is odd): ONE patch buffer -- the next plane waits in
    //      registers during the sweep and is committed once every wave has read the patch ---------------------------------------
    const int zlast = 2 * zo1 - 1;
    int zi = 2 * zo0 - 1;
    issue(zi);
    commit();
    __syncthreads();
    auto odd_step = [&](f4 (&lo)[AW], f4 (&hi)[AW]) {   // zi = 2a + 1: completes output plane a (lo), opens plane a + 1 (hi)
        const bool more = zi < zlast, live = zi >= 0 && zi < D;
        if (more) issue(zi + 1);
        if (live) sweep(2, lo, 0, hi, true);
        lds_barrier();                                 // every wave has read the patch
        if (more) commit();
        store_plane((zi - 1) >> 1, lo);
        lds_barrier();
    };
    auto even_step = [&](f4 (&mid)[AW]) {               // zi = 2a: k_z = 1 of output plane a; never the last plane
        const bool live = zi < D;
        issue(zi + 1);
        if (live) sweep(1, mid, 1, mid, false);
        lds_barrier();
        commit();
        lds_barrier();
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

template <int CI, int NTN, int MGK, bool WG>
static int launch_s2x(const S2XParams& p, hipStream_t stream) {
    constexpr int NKB = (9 * CI + 31) / 32;
    constexpr int TXOk = 16 * MGK, PXIk = 2 * TXOk + 1;
    constexpr int lds = PXIk * PYI * s2x_cell_bytes(CI) + (WG ? 0 : 3 * 3 * NKB * NTN * 64 * 16);
    static_assert(lds <= 160 * 1024, "tile does not fit the LDS");
    auto kern = conv3d_s2_x3_kernel<CI, NTN, MGK, WG>;
    int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(kern), lds);
    if (rc != D3D_OK) return rc;
    S2XParams q = p;
    const int gx = ceil_div(p.Wo, TXOk), gy = ceil_div(p.Ho, TYO);
    q.zper = pick_zper((long)gx * gy, p.Do, 2, 1, lds);   // (every z segment re-reads one halo plane)
    hipLaunchKernelGGL(kern, dim3(gx, gy, ceil_div(p.Do, q.zper)), dim3(NT), lds, stream, q);
    D3D_LAUNCH_CHECK("conv3d_s2_x3_kernel launch");
    return D3D_OK;
}

}  // namespace

}  // namespace d3d

using namespace d3d;

extern "C" int d3d_conv3d_k3s2_zs_bf16x3(const float* in, const void* wpacked, const float* scale, const float* shift,
                                         const float* skip, int relu, int Ci, int Co, int D, int H, int W, float* out,
                                         d3d_stream_t stream) {
    D3D_REQUIRE(in && wpacked && out, "null pointer");
    D3D_REQUIRE(D > 0 && H > 0 && W > 0, "bad dims %dx%dx%d", D, H, W);
    S2XParams p = {};
    p.in = in; p.wpk = reinterpret_cast<const u4*>(wpacked); p.scale = scale; p.shift = shift; p.skip = skip; p.out = out;
    p.D = D; p.H = H; p.W = W; p.Do = (D - 1) / 2 + 1; p.Ho = (H - 1) / 2 + 1; p.Wo = (W - 1) / 2 + 1; p.CO = Co; p.relu = relu;
    const bool shape = (Ci == 8 && Co == 16) || (Ci == 16 && Co == 32) || (Ci == 32 && Co == 64);
    if (!shape || p.Wo % 4 != 0 || ceil_div(p.Ho, TYO) > 65535 || p.Do > 65535 ||
        ((reinterpret_cast<uintptr_t>(out) | reinterpret_cast<uintptr_t>(skip)) & 15)) {
        set_error("d3d_conv3d_k3s2_zs_bf16x3: %d -> %d channels (8 -> 16, 16 -> 32, 32 -> 64), output width %d (a multiple of 4), "
                  "16-byte aligned output not taken", Ci, Co, p.Wo);
        return D3D_ERR_UNSUPPORTED;
    }
    hipStream_t st = (hipStream_t)stream;
    if (Ci == 8) return launch_s2x<8, 1, 2, false>(p, st);
    if (Ci == 16) return launch_s2x<16, 2, 1, false>(p, st);   // 16-wide tiles: 54 KB of split cells beside 90 KB of weights
    return launch_s2x<32, 4, 1, true>(p, st);                   // conv5: weights streamed from L2
}
