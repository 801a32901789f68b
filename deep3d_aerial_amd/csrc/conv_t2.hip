// nn.ConvTranspose3d(k = 3, stride 2, pad 1, output_pad 1) + folded BN + ReLU + skip -- the three decoder layers of every
// CostRegNet (cas_mvsnet.py:97-103,116-118: 64 -> 32, 32 -> 16, 16 -> 8; module.py:307-314 Deconv3d) -- on the bf16 matrix
// cores (v_mfma_f32_16x16x32_bf16, fp32 accumulation), z-streaming like conv_c8.hip.  BASELINE config 3 (bf16 operands).
//
// Output voxel o = 2 i + p per dimension (p = parity): an even coordinate takes kernel tap k = 1 of input i; an odd one
// takes k = 2 of input i and k = 0 of input i + 1.  So the layer is eight small dense convolutions (one per output
// parity, 1 .. 8 taps) over the SAME input, and nothing is multiplied by an inserted zero:
//   * a workgroup (8 waves) owns a 32 x 8 tile of INPUT pixels (= 64 x 16 outputs per plane) and walks the OUTPUT planes;
//     an even plane needs input plane z/2, an odd one planes (z-1)/2 and (z+1)/2: two staged planes are resident in LDS
//     (fp32 planar -> bf16 channel-last cells, RNE), the next one is loaded while an even plane is computed;
//   * per output plane and wave (= input row): for each row parity both column parities are accumulated (M = 16 input
//     pixels, N = 16 output channels, K = taps x C_in in blocks of 32: an A operand is one ds_read_b128 of 8 channels of
//     the cell (x + dx, y + dy) of plane z + dz), then interleaved in registers so that a lane stores 8 consecutive
//     output pixels (two 16-byte stores; the skip tensor is read the same way);
//   * the weights of all 27 taps sit in LDS, packed by the host per parity class in the B-operand lane order.
#include "common.h"

#include <type_traits>

namespace d3d {

namespace {

typedef float f4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u4 __attribute__((ext_vector_type(4)));

constexpr int TYI = 8;   // input rows per workgroup = waves

// CL = true: input, skip and output are channel-last bf16 volumes ([D][H][W][C], see conv_c8.hip) instead of planar fp32.
struct T2Params {
    const void* in;      // [CI, D, H, W] fp32 | [D, H, W, CI] bf16
    const u4* wpk;       // per parity class (pz, py, px): [K blocks][N tiles][64 lanes] 16-byte B fragments
    const float* scale;  // [CO] or null
    const float* shift;  // [CO] or null
    const void* skip;    // [CO, 2D, 2H, 2W] fp32 | [2D, 2H, 2W, CO] bf16, or null (added after the activation)
    void* out;           // [CO, 2D, 2H, 2W] fp32 | [2D, 2H, 2W, CO] bf16
    int D, H, W, CO;
    int relu;
    int ozper;           // output planes per workgroup (even)
};

__device__ __forceinline__ unsigned pack_bf16_t2(float a, float b) {   // the split (fp32-mode) operands: three bf16 pieces
    return pack_bf16x2(a, b);   // one v_cvt_pk_bf16_f32 (common.h)
}
__device__ __forceinline__ unsigned pack_h16_t2(float a, float b) {    // the single 16-bit operand / stored activation (common.h)
    return pack_h16x2(a, b);
}

constexpr int ntaps(int pz, int py, int px) { return (1 + pz) * (1 + py) * (1 + px); }
constexpr int nkb(int CI, int pz, int py, int px) { return (ntaps(pz, py, px) * CI + 31) / 32; }
// first fragment (in units of NTN * 64 lanes) of parity class p = pz * 4 + py * 2 + px
constexpr int frag_base(int CI, int p) {
    int s = 0;
    for (int q = 0; q < p; ++q) s += nkb(CI, q >> 2, (q >> 1) & 1, q & 1);
    return s;
}

__device__ __forceinline__ f4 unpack_h16x4_t2(uint2 u) {
    return (f4){h16_lo(u.x), h16_hi(u.x),
                h16_lo(u.y), h16_hi(u.y)};
}

// first fragment of the x-folded packing: class (pz, py) holds the K blocks of its px = 1 tap set
constexpr int fold_base(int CI, int c) {
    int s = 0;
    for (int q = 0; q < c; ++q) s += nkb(CI, q >> 1, q & 1, 1);
    return s;
}

// NTN: 16-channel output tiles; MG: 16-pixel groups per wave (input tile width = 16 * MG); CL: channel-last bf16 volumes.
// FOLD (CL, C_out = 8): both column parities in ONE GEMM -- rows 0..7 of the weight operand are the 8 channels of the even
// output column (taps with dx = 1 zeroed), rows 8..15 those of the odd one, so no half of the tile idles, the four lane
// groups of a pixel leave with the 32 contiguous bytes of the output cells (2 ix, 2 ix + 1) and every lane stores.
// X3 (planar fp32 tensors, fp32 accuracy): cells hold the hi | mid | lo runs of the exact three-way bf16 split of their channels,
// the weight fragments come in three parts, six products per K block (see conv_c8.hip).
// WG: the weight fragments are read from global memory (L2) per use instead of living in LDS -- 64 -> 32 in split operands:
// 324 KB of fragments, tiny volumes (as conv_c8.hip's 64 -> 64 layer).
template <int CI, int NTN, int MG, bool CL, bool FOLD = false, bool X3 = false, bool WG = false>
__global__ __launch_bounds__(64 * TYI) void convt3d_zs_bf16_kernel(T2Params p) {
    static_assert(!FOLD || (CL && NTN == 1), "x-folded form: channel-last, C_out = 8");
    static_assert(!X3 || (!CL && !FOLD), "split operands: planar fp32 tensors");
    constexpr int NT = 64 * TYI;
    constexpr int TXI = 16 * MG;
    constexpr int PXI = TXI + 1, PYI = TYI + 1;         // input patch: one more column / row for the d = 1 taps
    constexpr int CS = X3 ? 6 * CI + 16 : CI == 64 ? 144 : bf16_cell_bytes(CI);   // bytes per cell (bank-conflict-free pitch: common.h; 64 channels: 160-byte cells measured 0.17 -> 0.21 ms per view)
    constexpr int G = CI / 8;
    constexpr int PATCH = PXI * PYI * CS;
    constexpr int NFRAG = FOLD ? fold_base(CI, 4) : frag_base(CI, 8);
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    u4* wlds = reinterpret_cast<u4*>(smem + 2 * PATCH);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // each XCD takes a contiguous run of the logical tile order (neighbouring tiles share halo cells: one L2), see conv_c8.hip
    int bxi = blockIdx.x, byi = blockIdx.y, bzi = blockIdx.z;
#ifdef D3D_T2_XCD   // (measured: regulariser leg 5.96 -> 6.08 ms -- off)
    {
        const int nx = gridDim.x, ny = gridDim.y, n = nx * ny * gridDim.z;
        if (n % 8 == 0) {
            int lin = (bzi * ny + byi) * nx + bxi;
            lin = (lin % 8) * (n / 8) + lin / 8;
            bxi = lin % nx; byi = (lin / nx) % ny; bzi = lin / (nx * ny);
        }
    }
#endif
    const int ix0 = bxi * TXI, iy0 = byi * TYI;
    const int D = p.D, H = p.H, W = p.W;
    const int oz0 = bzi * p.ozper, oz1 = min(oz0 + p.ozper, 2 * D);
    const size_t iplane = (size_t)H * W, ivol = (size_t)D * iplane;
    const size_t oplane = 4 * iplane, ovol = (size_t)(2 * D) * oplane;
    const int OW = 2 * W;

    if constexpr (!WG)
        for (int i = tid; i < (X3 ? 3 : 1) * NFRAG * NTN * 64; i += NT) wlds[i] = p.wpk[i];   // (split: [part][fragment])
    const u4* __restrict__ wsrc = WG ? p.wpk : wlds;

    constexpr int NTASK = PXI * PYI * G;
    constexpr int ROUNDS = (NTASK + NT - 1) / NT;
    float stg[CL ? 1 : ROUNDS][8];
    u4 stc[CL ? ROUNDS : 1];
    // channel-last input: a task's offset inside an input plane (clamped: always loaded, zeroed outside), its cell and whether it is
    // inside are per-lane state computed once (instruction issue bounds these kernels, see conv_c8.hip)
    unsigned stoff[CL ? ROUNDS : 1];
    int stdst[CL ? ROUNDS : 1];
    bool stok[CL ? ROUNDS : 1];
    if constexpr (CL) {
#pragma unroll
        for (int r = 0; r < ROUNDS; ++r) {
            const int task = tid + r * NT;
            const int pix = task / G, g = task - pix * G;
            const int py = pix / PXI, px = pix - py * PXI;
            const int gx = ix0 + px, gy = iy0 + py;
            stok[r] = task < NTASK && gx < W && gy < H;
            stoff[r] = task >= NTASK ? 0 : ((unsigned)min(gy, H - 1) * W + min(gx, W - 1)) * (CI * 2) + g * 16;   // (host: a plane is < 2^31 bytes; no task: the plane's first bytes, one line)
            stdst[r] = task < NTASK ? pix * CS + g * 16 : -1;
        }
    }
    // planar fp32 input: the same per-lane state (element offset of the task's first channel inside a plane, clamped)
    size_t poff[CL ? 1 : ROUNDS];
    int pdst[CL ? 1 : ROUNDS];
    bool pok[CL ? 1 : ROUNDS];
    if constexpr (!CL) {
#pragma unroll
        for (int r = 0; r < ROUNDS; ++r) {
            const int task = tid + r * NT;
            const int pix = task / G, g = task - pix * G;
            const int py = pix / PXI, px = pix - py * PXI;
            const int gx = ix0 + px, gy = iy0 + py;
            pok[r] = task < NTASK && gx < W && gy < H;
            poff[r] = task < NTASK ? (size_t)(8 * g) * ivol + (size_t)min(gy, H - 1) * W + min(gx, W - 1) : 0;
            pdst[r] = task < NTASK ? pix * CS + g * 16 : -1;
        }
    }
    bool stzin = false;   // the plane in the staging registers is inside the volume
    auto issue = [&](int zi) {
        const bool zin = zi >= 0 && zi < D;
        stzin = zin;
        if constexpr (CL) {
            if (!zin) return;   // (uniform) a plane outside the volume: nothing loaded, the commit writes zeros
            const unsigned char* __restrict__ src = static_cast<const unsigned char*>(p.in) + (size_t)min(max(zi, 0), D - 1) * iplane * (CI * 2);
#pragma unroll
            for (int r = 0; r < ROUNDS; ++r) {
                const u4 v = *reinterpret_cast<const u4*>(src + stoff[r]);
                stc[r] = v;   // raw: zeroed for cells outside the volume when it is committed -- a select here would wait for the load
            }
            return;
        }
        if (!zin) return;   // (uniform) a plane outside the volume: nothing loaded, the commit writes zeros
        const float* __restrict__ srcp = static_cast<const float*>(p.in) + (size_t)zi * iplane;
#pragma unroll
        for (int r = 0; r < ROUNDS; ++r) {
            const float* __restrict__ src = srcp + poff[CL ? 0 : r];
#pragma unroll
            for (int k = 0; k < 8; ++k) stg[r][k] = src[(size_t)k * ivol];   // raw: zeroed when committed
        }
    };
    auto commit = [&](unsigned char* dst) {
        if constexpr (CL) {
#pragma unroll
            for (int r = 0; r < ROUNDS; ++r)
                if (stdst[r] >= 0) *reinterpret_cast<u4*>(dst + stdst[r]) = stzin && stok[r] ? stc[r] : (u4){0, 0, 0, 0};
            return;
        }
#pragma unroll
        for (int r = 0; r < ROUNDS; ++r) {
            if (pdst[CL ? 0 : r] >= 0) {
                const bool ok = stzin && pok[CL ? 0 : r];
                float x[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) x[k] = ok ? stg[r][k] : 0.0f;
                unsigned char* cell = dst + pdst[CL ? 0 : r];
                if constexpr (X3) {   // hi = bf16(x), mid = bf16(x - hi), lo = bf16(x - hi - mid): the differences are exact in fp32
                    unsigned hi[4], mi[4], lo[4];
#pragma unroll
                    for (int k = 0; k < 8; k += 2) {
                        hi[k >> 1] = pack_bf16_t2(x[k], x[k + 1]);
                        const float a1 = x[k] - __builtin_bit_cast(float, hi[k >> 1] << 16);
                        const float b1 = x[k + 1] - __builtin_bit_cast(float, hi[k >> 1] & 0xffff0000u);
                        mi[k >> 1] = pack_bf16_t2(a1, b1);
                        const float a2 = a1 - __builtin_bit_cast(float, mi[k >> 1] << 16);
                        const float b2 = b1 - __builtin_bit_cast(float, mi[k >> 1] & 0xffff0000u);
                        lo[k >> 1] = pack_bf16_t2(a2, b2);
                    }
                    *reinterpret_cast<u4*>(cell) = (u4){hi[0], hi[1], hi[2], hi[3]};
                    *reinterpret_cast<u4*>(cell + CI * 2) = (u4){mi[0], mi[1], mi[2], mi[3]};
                    *reinterpret_cast<u4*>(cell + CI * 4) = (u4){lo[0], lo[1], lo[2], lo[3]};
                    continue;
                }
                *reinterpret_cast<u4*>(cell) = (u4){pack_h16_t2(x[0], x[1]), pack_h16_t2(x[2], x[3]), pack_h16_t2(x[4], x[5]), pack_h16_t2(x[6], x[7])};
            }
        }
    };

    const int n = lane & 15;                       // output channel within a 16-channel tile
    const int iy = iy0 + wave;                     // this wave's input row
    const int abase = (wave * PXI + (lane & 15)) * CS;

    // plane-invariant state of the channel-last epilogues: affine of the lane's four channels, offsets (bf16 elements) inside
    // an output plane of output row 2 iy (+ PY rows of 2 W cells) and store masks per pixel group
    f4 esc[CL ? NTN : 1], esh[CL ? NTN : 1];
    unsigned eoff[CL ? MG : 1];
    bool est[CL ? NTN : 1][CL ? MG : 1];
    if constexpr (CL) {
        const int g = lane >> 4;
#pragma unroll
        for (int nt = 0; nt < NTN; ++nt) {
            const int cb = FOLD ? (g & 1) * 4 : nt * 16 + g * 4;
            const bool cin = cb < p.CO;
            esc[nt] = p.scale && cin ? *reinterpret_cast<const f4*>(p.scale + cb) : (f4){1, 1, 1, 1};
            esh[nt] = p.shift && cin ? *reinterpret_cast<const f4*>(p.shift + cb) : (f4){0, 0, 0, 0};
#pragma unroll
            for (int mg = 0; mg < MG; ++mg) est[nt][mg] = cin && iy < H && ix0 + mg * 16 + (lane & 15) < W;
        }
#pragma unroll
        for (int mg = 0; mg < MG; ++mg) {
            const int ix = min(ix0 + mg * 16 + (lane & 15), W - 1);
            eoff[mg] = ((unsigned)(2 * min(iy, H - 1)) * OW + 2 * ix + (FOLD ? g >> 1 : 0)) * p.CO + (FOLD ? (g & 1) * 4 : g * 4);   // (host: a plane is < 2^31 elements)
        }
    }
    const size_t oplane_cl = (size_t)4 * iplane * p.CO;   // bf16 elements of a channel-last output plane
    // one (pz, py) row of outputs: both column parities, all pixel groups and channel tiles
    auto row = [&](auto pzc, auto pyc, int oz, const unsigned char* b0, const unsigned char* b1) {
        constexpr int PZ = decltype(pzc)::value, PY = decltype(pyc)::value;
        if constexpr (FOLD) {
            f4 accf[MG];
#pragma unroll
            for (int mg = 0; mg < MG; ++mg) accf[mg] = (f4){0, 0, 0, 0};
            const int kgroup = lane >> 4;   // (not opaque: the K offsets are plane-invariant, the compiler keeps or folds them)
            constexpr int NKB = nkb(CI, PZ, PY, 1);
            constexpr int FB = fold_base(CI, PZ * 2 + PY);
#pragma unroll
            for (int kb = 0; kb < NKB; ++kb) {
                const int k0 = 32 * kb + 8 * kgroup;
                const int t = k0 / CI, c = k0 % CI;            // taps (dz, dy, dx) of the odd column, dz-major
                const bool real = t < ntaps(PZ, PY, 1);
                const int dx = real ? (t & 1) : 0, dy = real ? (t >> 1) % (1 + PY) : 0, dz = real ? (t >> 1) / (1 + PY) : 0;
                const unsigned char* buf = dz ? b1 : b0;
                const int aoff = (dy * PXI + dx) * CS + (real ? c : 0) * 2;
                const h16x8 wf = __builtin_bit_cast(h16x8, wsrc[(FB + kb) * 64 + lane]);
#pragma unroll
                for (int mg = 0; mg < MG; ++mg) {
                    const h16x8 a = __builtin_bit_cast(h16x8, *reinterpret_cast<const u4*>(buf + abase + mg * 16 * CS + aoff));
                    accf[mg] = mfma_h16(wf, a, accf[mg]);
                }
            }
            // D row (lane >> 4) * 4 + r = (column parity, channel), column = input pixel lane & 15
            {
                const unsigned short* __restrict__ sk = static_cast<const unsigned short*>(p.skip) + (size_t)oz * oplane_cl + PY * (OW * 8);
                unsigned short* __restrict__ dst = static_cast<unsigned short*>(p.out) + (size_t)oz * oplane_cl + PY * (OW * 8);
#pragma unroll
                for (int mg = 0; mg < MG; ++mg) {
                    if (est[0][mg]) {
                        f4 v = accf[mg] * esc[0] + esh[0];
                        if (p.relu) v = __builtin_elementwise_max(v, (f4){0, 0, 0, 0});
                        if (p.skip) v += unpack_h16x4_t2(*reinterpret_cast<const uint2*>(sk + eoff[mg]));
                        const uint2 pk = {pack_h16_t2(v[0], v[1]), pack_h16_t2(v[2], v[3])};
                        *reinterpret_cast<uint2*>(dst + eoff[mg]) = pk;
                    }
                }
            }
            return;
        }
        f4 acc[2][MG][NTN];
#pragma unroll
        for (int px = 0; px < 2; ++px)
#pragma unroll
            for (int mg = 0; mg < MG; ++mg)
#pragma unroll
                for (int nt = 0; nt < NTN; ++nt) acc[px][mg][nt] = (f4){0, 0, 0, 0};
        const int kgroup = lane >> 4;
#pragma unroll
        for (int px = 0; px < 2; ++px) {
            const int NKB = nkb(CI, PZ, PY, px);
            const int FB = frag_base(CI, PZ * 4 + PY * 2 + px);
#pragma unroll
            for (int kb = 0; kb < NKB; ++kb) {
                // K index k = 32 kb + 8 kgroup + j  ->  tap t = k / CI over (dz, dy, dx), channel k % CI
                const int k0 = 32 * kb + 8 * kgroup;
                const int t = k0 / CI, c = k0 % CI;
                const bool real = t < ntaps(PZ, PY, px);
                const int dx = real ? t % (1 + px) : 0, dy = real ? (t / (1 + px)) % (1 + PY) : 0;
                const int dz = real ? t / ((1 + px) * (1 + PY)) : 0;
                const unsigned char* buf = dz ? b1 : b0;
                const int aoff = (dy * PXI + dx) * CS + (real ? c : 0) * 2;
                if constexpr (X3) {
                    constexpr int WS = NFRAG * NTN * 64;   // fragments per weight part
#pragma unroll
                    for (int mg = 0; mg < MG; ++mg) {
                        bf16x8 a[3];
#pragma unroll
                        for (int sp = 0; sp < 3; ++sp)
                            a[sp] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u4*>(buf + abase + mg * 16 * CS + aoff + sp * CI * 2));
#pragma unroll
                        for (int nt = 0; nt < NTN; ++nt) {
                            const int wi = ((FB + kb) * NTN + nt) * 64 + lane;
                            const bf16x8 bh = __builtin_bit_cast(bf16x8, wsrc[wi]);
                            const bf16x8 bm = __builtin_bit_cast(bf16x8, wsrc[WS + wi]);
                            const bf16x8 bl = __builtin_bit_cast(bf16x8, wsrc[2 * WS + wi]);
                            f4 c = acc[px][mg][nt];   // small terms first
                            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[2], bh, c, 0, 0, 0);
                            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], bl, c, 0, 0, 0);
                            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1], bm, c, 0, 0, 0);
                            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1], bh, c, 0, 0, 0);
                            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], bm, c, 0, 0, 0);
                            acc[px][mg][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], bh, c, 0, 0, 0);
                        }
                    }
                    continue;
                }
                h16x8 bfrag[NTN];
#pragma unroll
                for (int nt = 0; nt < NTN; ++nt) bfrag[nt] = __builtin_bit_cast(h16x8, wsrc[((FB + kb) * NTN + nt) * 64 + lane]);
#pragma unroll
                for (int mg = 0; mg < MG; ++mg) {
                    const h16x8 a = __builtin_bit_cast(h16x8, *reinterpret_cast<const u4*>(buf + abase + mg * 16 * CS + aoff));
#pragma unroll
                    for (int nt = 0; nt < NTN; ++nt) {
                        if constexpr (CL) acc[px][mg][nt] = mfma_h16(bfrag[nt], a, acc[px][mg][nt]);
                        else acc[px][mg][nt] = mfma_h16(a, bfrag[nt], acc[px][mg][nt]);
                    }
                }
            }
        }
        // epilogue: D row = input pixel (lane >> 4) * 4 + r, column = channel; outputs x = 2 ix + px interleaved
        const int oy = 2 * iy + PY;
        if constexpr (CL) {
            // weights were the A operand: D row = channel (lane >> 4) * 4 + register, column = input pixel lane & 15; a lane
            // stores four consecutive channels of the output pixels (2 ix, 2 ix + 1) = one 8-byte store each
            const unsigned short* __restrict__ sk = static_cast<const unsigned short*>(p.skip) + (size_t)oz * oplane_cl + (size_t)PY * OW * p.CO;
            unsigned short* __restrict__ dst = static_cast<unsigned short*>(p.out) + (size_t)oz * oplane_cl + (size_t)PY * OW * p.CO;
#pragma unroll
            for (int nt = 0; nt < NTN; ++nt) {
#pragma unroll
                for (int mg = 0; mg < MG; ++mg) {
                    if (est[nt][mg]) {
#pragma unroll
                        for (int px = 0; px < 2; ++px) {
                            const unsigned o = eoff[mg] + px * p.CO + nt * 16;
                            f4 v = acc[px][mg][nt] * esc[nt] + esh[nt];
                            if (p.relu) v = __builtin_elementwise_max(v, (f4){0, 0, 0, 0});
                            if (p.skip) v += unpack_h16x4_t2(*reinterpret_cast<const uint2*>(sk + o));
                            const uint2 pk = {pack_h16_t2(v[0], v[1]), pack_h16_t2(v[2], v[3])};
                            *reinterpret_cast<uint2*>(dst + o) = pk;
                        }
                    }
                }
            }
        } else if (iy < H) {
            const float* __restrict__ skipf = static_cast<const float*>(p.skip);
            float* __restrict__ outf = static_cast<float*>(p.out);
#pragma unroll
            for (int nt = 0; nt < NTN; ++nt) {
                const int co = nt * 16 + n;
                if (co < p.CO) {
                    const float sc = p.scale ? p.scale[co] : 1.0f, sh = p.shift ? p.shift[co] : 0.0f;
#pragma unroll
                    for (int mg = 0; mg < MG; ++mg) {
                        const int ix = ix0 + mg * 16 + (lane >> 4) * 4;
                        if (ix < W) {   // W need not be a multiple of 4: guard each pair below
                            const size_t o = (size_t)co * ovol + (size_t)oz * oplane + (size_t)oy * OW + 2 * ix;
                            f4 e = acc[0][mg][nt] * sc + sh, od = acc[1][mg][nt] * sc + sh;
                            if (p.relu) { e = __builtin_elementwise_max(e, (f4){0, 0, 0, 0}); od = __builtin_elementwise_max(od, (f4){0, 0, 0, 0}); }
                            f4 lo = {e[0], od[0], e[1], od[1]}, hi = {e[2], od[2], e[3], od[3]};
                            if (ix + 3 < W && (W & 1) == 0) {   // 16-byte aligned rows
                                if (skipf) { lo += *reinterpret_cast<const f4*>(skipf + o); hi += *reinterpret_cast<const f4*>(skipf + o + 4); }
                                *reinterpret_cast<f4*>(outf + o) = lo;
                                *reinterpret_cast<f4*>(outf + o + 4) = hi;
                            } else {   // ragged right edge
                                const float v[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                                for (int j = 0; j < 8; ++j)
                                    if (ix + (j >> 1) < W) outf[o + j] = v[j] + (skipf ? skipf[o + j] : 0.0f);
                            }
                        }
                    }
                }
            }
        }
    };

    // ---- walk the output planes: plane iz lives in buffer iz & 1 ------------------------------------------------
    const int izf = oz0 >> 1;                     // oz0 is even
    issue(izf);
    commit(smem + (izf & 1) * PATCH);
    __syncthreads();
    for (int oz = oz0; oz < oz1; ++oz) {
        const int iz = oz >> 1;
        const unsigned char* b0 = smem + (iz & 1) * PATCH;
        const unsigned char* b1 = smem + ((iz + 1) & 1) * PATCH;
        if ((oz & 1) == 0) {
            const bool more = oz + 1 < oz1;       // the odd plane that follows needs input plane iz + 1
            if (more) issue(iz + 1);
            row(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, oz, b0, b1);
            row(std::integral_constant<int, 0>{}, std::integral_constant<int, 1>{}, oz, b0, b1);
            if (more) commit(smem + ((iz + 1) & 1) * PATCH);   // that buffer held plane iz - 1, last read by plane oz - 1
        } else {
            row(std::integral_constant<int, 1>{}, std::integral_constant<int, 0>{}, oz, b0, b1);
            row(std::integral_constant<int, 1>{}, std::integral_constant<int, 1>{}, oz, b0, b1);
        }
        lds_barrier();   // (LDS only: the stores of the plane stay in flight, common.h)
    }
}

template <int CI, int NTN, int MG, bool CL, bool FOLD = false, bool X3 = false, bool WG = false>
static int launch(const T2Params& p, hipStream_t stream) {
    constexpr int TXI = 16 * MG;
    constexpr int CS = X3 ? 6 * CI + 16 : CI == 64 ? 144 : bf16_cell_bytes(CI);
    constexpr int lds = 2 * (TXI + 1) * (TYI + 1) * CS + (WG ? 0 : (X3 ? 3 : 1) * (FOLD ? fold_base(CI, 4) : frag_base(CI, 8)) * NTN * 64 * 16);
    static_assert(lds <= 160 * 1024, "tile does not fit the LDS");
    auto kern = convt3d_zs_bf16_kernel<CI, NTN, MG, CL, FOLD, X3, WG>;
    int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(kern), lds);
    if (rc != D3D_OK) return rc;
    T2Params q = p;
    const int gx = ceil_div(p.W, TXI), gy = ceil_div(p.H, TYI);
    q.ozper = 2 * pick_zper((long)gx * gy, p.D, 2, 1, lds);
    hipLaunchKernelGGL(kern, dim3(gx, gy, ceil_div(2 * p.D, q.ozper)), dim3(64 * TYI), lds, stream, q);
    D3D_LAUNCH_CHECK("convt3d_zs_bf16_kernel launch");
    return D3D_OK;
}

}  // namespace

}  // namespace d3d

using namespace d3d;

extern "C" int d3d_convtranspose3d_k3s2_cl_h16(const void* in, const void* wpacked, const float* scale, const float* shift,
                                                const void* skip, int relu, int Ci, int Co, int D, int H, int W, void* out,
                                                int channel_last, d3d_stream_t stream) {
    D3D_REQUIRE(in && wpacked && out, "null pointer");
    D3D_REQUIRE(D > 0 && H > 0 && W > 0, "bad dims %dx%dx%d", D, H, W);
    const bool shape = (Ci == 16 && Co == 8) || (Ci == 16 && Co == 16) || (Ci == 32 && Co == 16) || (Ci == 64 && Co == 32);
    if (!shape || ceil_div(H, TYI) > 65535 || 2 * D > 65535 || (channel_last && (long)H * W * 4 * (Ci > Co ? Ci : Co) * 2 >= (1L << 31))) {   // (32-bit offsets inside a plane)
        set_error("d3d_convtranspose3d_k3s2_cl_h16: %d -> %d channels not taken (16->8, 16->16, 32->16, 64->32)", Ci, Co);
        return D3D_ERR_UNSUPPORTED;
    }
    T2Params p = {};
    p.in = in; p.wpk = reinterpret_cast<const u4*>(wpacked); p.scale = scale; p.shift = shift; p.skip = skip; p.out = out;
    p.D = D; p.H = H; p.W = W; p.CO = Co; p.relu = relu;
    hipStream_t st = (hipStream_t)stream;
    if (channel_last == 2) {   // x-folded weight packing
        if (Ci == 16 && Co == 8) return launch<16, 1, 2, true, true>(p, st);
        set_error("d3d_convtranspose3d_k3s2_cl_h16: the x-folded form takes 16 -> 8 channels only");
        return D3D_ERR_UNSUPPORTED;
    }
    if (channel_last) {
        if (Ci == 16) return launch<16, 1, 2, true>(p, st);
        if (Ci == 32) return launch<32, 1, 2, true>(p, st);
        return launch<64, 2, 1, true>(p, st);
    }
    if (Ci == 16) return launch<16, 1, 2, false>(p, st);
    if (Ci == 32) return launch<32, 1, 2, false>(p, st);
    return launch<64, 2, 1, false>(p, st);
}

extern "C" int d3d_convtranspose3d_k3s2_zs_bf16x3(const float* in, const void* wpacked, const float* scale, const float* shift,
                                                  const float* skip, int relu, int Ci, int Co, int D, int H, int W, float* out,
                                                  d3d_stream_t stream) {
    D3D_REQUIRE(in && wpacked && out, "null pointer");
    D3D_REQUIRE(D > 0 && H > 0 && W > 0, "bad dims %dx%dx%d", D, H, W);
    const bool shape = (Ci == 16 && (Co == 8 || Co == 16)) || (Ci == 32 && Co == 16) || (Ci == 64 && Co == 32);
    if (!shape || ceil_div(H, TYI) > 65535 || 2 * D > 65535) {
        set_error("d3d_convtranspose3d_k3s2_zs_bf16x3: %d -> %d channels not taken (16 -> 8, 16 -> 16, 32 -> 16, 64 -> 32)", Ci, Co);
        return D3D_ERR_UNSUPPORTED;
    }
    T2Params p = {};
    p.in = in; p.wpk = reinterpret_cast<const u4*>(wpacked); p.scale = scale; p.shift = shift; p.skip = skip; p.out = out;
    p.D = D; p.H = H; p.W = W; p.CO = Co; p.relu = relu;
    if (Ci == 64) return launch<64, 2, 1, false, false, true, true>(p, (hipStream_t)stream);   // conv7 (round 4): fragments from L2
    if (Ci == 32) return launch<32, 1, 1, false, false, true>(p, (hipStream_t)stream);   // conv9 (round 4): 16-wide tiles, two 17 x 9 patches of 208-byte split cells beside 81 KB of weights
    return launch<16, 1, 2, false, false, true>(p, (hipStream_t)stream);
}

extern "C" int d3d_convtranspose3d_k3s2_zs_h16(const float* in, const void* wpacked, const float* scale, const float* shift,
                                                const float* skip, int relu, int Ci, int Co, int D, int H, int W, float* out,
                                                d3d_stream_t stream) {
    return d3d_convtranspose3d_k3s2_cl_h16(in, wpacked, scale, shift, skip, relu, Ci, Co, D, H, W, out, 0, stream);
}
