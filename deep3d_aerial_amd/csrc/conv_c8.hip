// 3x3x3 stride-1 convolution with EIGHT output channels on the bf16 matrix cores of gfx950 (v_mfma_f32_16x16x32_bf16):
// conv0 of every CostRegNet (cas_mvsnet.py:84 ConvBnReLU3D(C_in, 8), module.py:297-304) on the full stage volumes --
// 32 -> 8 at [48,688,464], 16 -> 8 at [32,1376,928], 8 -> 8 at [8,2752,1856] -- the layers that carry most of a
// CasMVSNet view's activation traffic.  BASELINE config 3 asks for bf16 operands with fp32 accumulation.
//
// Why its own kernel: with 8 output channels the GEMM is [pixels] x [27 * C_in] x [8] -- the matrix cores finish it in a
// fraction of the time HBM needs to stream the volume once (8 -> 8 at stage 3: 141 GFLOP, 2.6 GB), so everything is
// arranged around reading each input plane ONCE and writing each output plane ONCE:
//   * A workgroup (8 waves) owns a 64 x 8 tile of (x, y) and walks z.  Input plane z is staged once (fp32 planar
//     [C_in][z][y][x] -> bf16 channel-last cells, RNE) and feeds the three output planes z-1, z, z+1 through the k_z
//     slices of the weights; three accumulator sets rotate, a plane is finished after its third input plane.
//   * GEMM orientation: M = 16 consecutive pixels of a row, N = 16 = the 8 output channels (upper half zero),
//     K = (k_y, k_x, c_in) in blocks of 32.  An A operand is ONE ds_read_b128 per lane: 8 consecutive channels of the
//     pixel (m + k_x, y + k_y) -- the im2col matrix is never materialised.  The B operands (weights, packed by the host in
//     exactly the lane order of the instruction) stay in registers (C_in <= 16) or LDS (C_in = 32).
//   * One wave = one row of the tile = four 16-pixel groups; D[m][n] leaves as 16-byte stores of 4 consecutive pixels
//     per (lane, channel) after the folded-BN affine, ReLU and skip.
//   * The next plane's global loads are issued before the MFMA sweep of the current one and written to the other LDS
//     buffer after it: one barrier per plane.
#include "common.h"

namespace d3d {

namespace {

typedef float f4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u4 __attribute__((ext_vector_type(4)));

constexpr int TY = 8;                      // output tile rows of a workgroup (columns: 16 * MGN, 64 by default)
constexpr int PY = TY + 2;                 // staged patch rows with the 1-pixel halo (columns: 16 * MGN + 2)
constexpr int NT = 64 * TY;                // one wave per tile row

// Activation formats.  PLANAR: fp32 [C][D][H][W] (what the cost-volume kernels write and the regression kernels read).
// CL ("channel-last"): bf16 [D][H][W][C] -- the form the layers of a CostRegNet hand to each other in bf16 mode: the
// operands are rounded to bf16 at staging anyway, so storing them rounded halves the activation traffic, and a staging
// task becomes ONE 16-byte load of 8 channels instead of 8 loads from 8 planes plus four conversions.
struct C8Params {
    const void* in;       // PLANAR [CI, D, H, W] fp32 | CL [D, H, W, CI] bf16
    const u4* wpk;        // [3 (kz)][NKB][N tiles][64 lanes] 16-byte B fragments (bf16 pairs), host-packed
    const float* scale;   // [CO] or null
    const float* shift;   // [CO] or null
    const void* skip;     // in the OUTPUT's format, or null (added after the activation)
    void* out;            // PLANAR [CO, D, H, W] fp32 | CL [D, H, W, CO] bf16
    int D, H, W;
    int relu;
    int zper;             // output planes per workgroup along z
    int CO;               // real output channels (<= 16 * NTN)
    int in_cl8;           // CL input in planes of 8-channel groups [D, CI/8, H, W, 8] (the sweep kernels' CL8 volume) instead of [D, H, W, CI]
};

__device__ __forceinline__ unsigned pack_bf16(float a, float b) {   // the split (fp32-mode) operands: three bf16 pieces
    return pack_bf16x2(a, b);   // one v_cvt_pk_bf16_f32 (common.h)
}
__device__ __forceinline__ unsigned pack_h16(float a, float b) {    // the single 16-bit operand / stored activation (common.h)
    return pack_h16x2(a, b);
}

// waves per SIMD the register budget is sized for: two workgroups per CU (the second one's loads and stores fly while the
// first one sweeps) where the staging registers allow it
__device__ __forceinline__ f4 unpack_h16x4(uint2 u) {
    return (f4){h16_lo(u.x), h16_hi(u.x),
                h16_lo(u.y), h16_hi(u.y)};
}

// NTN: 16-channel output tiles (1: C_out <= 16, 2: C_out = 32, 4: C_out = 64); INCL / OUTCL: channel-last bf16 input / output;
// MGN: 16-pixel groups per wave (tile width 16 * MGN); WG: the weight fragments are read from global memory (L2) per use
// instead of living in LDS -- the 64 -> 64 layer's 221 KB of weights do not fit beside a patch, its volumes are tiny.
// KZF (C_out = 1, planar output: the probability layer): the three k_z slices are COLUMNS 0, 1, 2 of one weight tile, so one
// MFMA per K block serves the three open output planes; after every input plane column 2 is complete (it is stored) and
// the columns move one to the right (v_mov_dpp row_shr:1 within the 16-lane rows) -- a third of the MFMAs and weight reads.
__device__ __forceinline__ float dpp_row_shr1(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x111, 0xf, 0xf, true));
}

// X3 (fp32 accuracy on the bf16 matrix cores, the 3-D form of conv2d_zs.hip's split operands; planar fp32 in and out): every
// input value and every weight is the exact sum of three bf16 numbers (hi + mid + lo); a staged cell holds the hi, mid and lo
// runs of its channels, the host packs the weights' three parts, and per K block the six products whose error terms matter --
// lo*hi, hi*lo, mid*mid, mid*hi, hi*mid, hi*hi, small terms first -- are accumulated in fp32: the error is that of the fp32
// instruction v_mfma_f32_16x16x4_f32, at six eighths of its time per product and with the z-streaming traffic of this kernel.
template <int CI, int NTN, bool INCL, bool OUTCL, int MGN = 4, bool WG = false, bool KZF = false, bool CO8 = false, bool X3 = false>
__global__ __launch_bounds__(NT, X3 ? (CI <= 16 ? 4 : 2) : NTN > 1 ? 1 : (CI <= 8 || MGN < 4) ? 4 : 2) void conv3d_c8_bf16_kernel(C8Params p) {
    static_assert(!X3 || (!INCL && !OUTCL && !CO8), "split operands: planar fp32 tensors");
    static_assert(!CO8 || (OUTCL && NTN == 1), "CO8: the channel-last epilogue for exactly 8 output channels");
    static_assert(!KZF || (NTN == 1 && !OUTCL && !WG), "k_z-folded form: one output channel, planar output");
    constexpr int TX = 16 * MGN, PX = TX + 2;          // (shadow the 64-wide defaults)
    constexpr int NKB = (9 * CI + 31) / 32;            // K blocks of 32 per k_z slice
    // bytes per pixel cell: an ODD number of 16-byte slots (1 | 3 | 5) -- split operands: 3 slots, 6 (16 channels: unpadded, so that
    // patch + weights stay under half the LDS and two workgroups share a CU; the A reads are a quarter of the LDS reads) or 13
    // bf16 cells: 16 | 32 bytes (8 | 16 channels); 32 | 64 channels as PLANES of 16-channel cells -- the bank-conflict-free pitch
    // of common.h without the padding of a 96- | 160-byte cell (the 32 -> 32 layer's patch + weights would not fit)
    constexpr bool PL = !X3 && CI == 32;              // (64 -> 64 at 1/512 of the voxels keeps its 144-byte cells: planes measured 0.27 -> 0.30 ms per view)
    constexpr int CS = X3 ? (CI == 16 ? 96 : 6 * CI + (CI > 8 ? 16 : 0)) : PL ? 32 : CI == 64 ? 144 : bf16_cell_bytes(CI);
    constexpr int NBUF = X3 && CI >= 16 ? 1 : 2;   // split cells of 16 | 32 channels: ONE patch buffer (commit after every wave has read it)
    constexpr int G = CI / 8;                          // 8-channel groups per pixel
    constexpr int PLANE = PX * PY * CS + 64;          // (planes: 64 mod 128 bytes apart, so a 16-byte staging store's eight lanes -- two pixels x two planes -- spread over the banks)
    constexpr int PATCH = PL ? (CI / 16) * PLANE : PX * PY * CS;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    u4* wlds = reinterpret_cast<u4*>(smem + NBUF * PATCH);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // Workgroups are dealt to the 8 XCDs round-robin in launch order (x fastest): give each XCD a CONTIGUOUS run of the logical
    // order instead, x then y fastest, so that neighbouring tiles -- which share their halo rows and columns -- share an L2.
    int bxi = blockIdx.x, byi = blockIdx.y, bzi = blockIdx.z;
#ifndef D3D_CONV_NO_XCD   // (regulariser leg 6.03 -> 5.97 ms)
    {
        const int nx = gridDim.x, ny = gridDim.y, n = nx * ny * gridDim.z;
        if (n % 8 == 0) {
            int lin = (bzi * ny + byi) * nx + bxi;
            lin = (lin % 8) * (n / 8) + lin / 8;
            bxi = lin % nx; byi = (lin / nx) % ny; bzi = lin / (nx * ny);
        }
    }
#endif
    const int x0 = bxi * TX, y0 = byi * TY;
    const int z0 = bzi * p.zper, z1 = min(z0 + p.zper, p.D);
    const int D = p.D, H = p.H, W = p.W;
    const size_t plane = (size_t)H * W, vol = (size_t)D * plane;

    // ---- weights: resident in LDS (a K block's three k_z fragments are read once per wave and plane and reused by the
    //      four pixel groups; keeping all 3 * NKB fragments in registers would cost the second workgroup per CU) -------
    if constexpr (!WG)
        for (int i = tid; i < (X3 ? 3 : 1) * (KZF ? 1 : 3) * NKB * NTN * 64; i += NT) wlds[i] = p.wpk[i];   // (split: [part][k_z][K block][tile][lane])
    const u4* __restrict__ wsrc = WG ? p.wpk : wlds;

    // ---- per-lane A offsets: K index k = 32 kb + 8 (lane >> 4) + j  ->  tap t = k / CI, channel k % CI ----------
    // One register per K block, computed once: recomputed per use (to spare NKB registers) the divisions and products were
    // ~25 vector instructions per K block and plane -- 375 of the 800 of a plane triple at C_in = 16, in kernels bound by
    // instruction issue (profiles/r04_t2p.txt).  WG (64 -> 64: weights from L2, tiny volumes) keeps the per-use form.
    auto a_offset = [&](int kb, int kgroup) {
        const int k0 = 32 * kb + 8 * kgroup;
        const int t = k0 / CI, c = k0 % CI;
        const int ky = t < 9 ? t / 3 : 0, kx = t < 9 ? t % 3 : 0;   // padded taps read a valid cell; their weights are zero
        const int cc = t < 9 ? c : 0;
        return (ky * PX + kx) * CS + (PL ? (cc >> 4) * PLANE + (cc & 15) * 2 : cc * 2);
    };
    int aoffs[WG ? 1 : NKB];
    if constexpr (!WG) {
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb) aoffs[kb] = a_offset(kb, lane >> 4);
    }
    const int abase = (wave * PX + (lane & 15)) * CS;   // pixel (m, row) of the patch: + mg * 16 * CS per pixel group

    // ---- staging: task = (patch pixel, 8-channel group) ------------------------------------------------------
    // The loads of the next plane are issued in two halves around the two halves of the MFMA sweep, so that only half of
    // them sit in registers at a time (C_in = 32: 6 rounds of 8 floats per thread).
    constexpr int NTASK = PX * PY * G;
    constexpr int ROUNDS = (NTASK + NT - 1) / NT;
    // (measured at the cascade shapes: C_in = 32 1.18 ms split vs 2.37 ms unsplit; C_in = 16 1.52 ms unsplit at one workgroup
    //  per CU vs 2.1-2.5 ms split or at two workgroups per CU; C_in = 8 0.76 ms at two workgroups per CU vs 1.54 ms at one)
    constexpr int RH = (CI > 16 && !INCL && NBUF == 2) ? (ROUNDS + 1) / 2 : ROUNDS;   // rounds in the first half
    float stg[INCL ? 1 : RH][8];
    u4 stc[INCL ? RH : 1];
    // Channel-last input: what a staging task reads and writes does not change from plane to plane -- its 16 bytes' offset inside
    // a plane (clamped into the image: the load is always issued, the value zeroed outside), whether it is inside, its cell in the
    // patch -- so it is per-lane state computed once.  These kernels are bound by instruction issue (profiles/r04_t2p.txt), and
    // recomputing (pixel, group, row, column, bounds, 64-bit address) per plane was a third of their vector instructions.
    unsigned stoff[INCL ? ROUNDS : 1];
    int stdst[INCL ? ROUNDS : 1];
    bool stok[INCL ? ROUNDS : 1];
    if constexpr (INCL) {
#pragma unroll
        for (int r = 0; r < ROUNDS; ++r) {
            const int task = tid + r * NT;
            const int pix = task / G, g = task - pix * G;
            const int py = pix / PX, px = pix - py * PX;
            const int gx = x0 + px - 1, gy = y0 + py - 1;
            stok[r] = task < NTASK && gx >= 0 && gx < W && gy >= 0 && gy < H;
            const unsigned cy = min(max(gy, 0), H - 1), cx = min(max(gx, 0), W - 1);
            stoff[r] = task >= NTASK ? 0 : p.in_cl8 ? ((g * H + cy) * W + cx) * 16 : (cy * W + cx) * (CI * 2) + g * 16;   // (host: a plane is < 2^31 bytes; no task: the plane's first bytes, one line)
            stdst[r] = task < NTASK ? (PL ? (g >> 1) * PLANE + pix * CS + (g & 1) * 16 : pix * CS + g * 16) : -1;
        }
    }
    // planar fp32 input (fp32 mode and the first layer of bf16 mode without a CL8 volume): the same per-lane state -- element
    // offset of the task's first channel inside a plane (clamped), whether it is inside the image, its cell
    size_t poff[INCL ? 1 : ROUNDS];
    int pdst[INCL ? 1 : ROUNDS];
    bool pok[INCL ? 1 : ROUNDS];
    if constexpr (!INCL) {
#pragma unroll
        for (int r = 0; r < ROUNDS; ++r) {
            const int task = tid + r * NT;
            const int pix = task / G, g = task - pix * G;
            const int py = pix / PX, px = pix - py * PX;
            const int gx = x0 + px - 1, gy = y0 + py - 1;
            pok[r] = task < NTASK && gx >= 0 && gx < W && gy >= 0 && gy < H;
            poff[r] = task < NTASK ? (size_t)(8 * g) * vol + (size_t)min(max(gy, 0), H - 1) * W + min(max(gx, 0), W - 1) : 0;
            pdst[r] = task < NTASK ? (PL ? (g >> 1) * PLANE + pix * CS + (g & 1) * 16 : pix * CS + g * 16) : -1;
        }
    }
    bool stzin = false;   // the plane in the staging registers is inside the volume
    auto issue = [&](int zi, int r0, int r1) {   // global loads of input plane zi, rounds [r0, r1), into registers (zeros outside the volume)
        const bool zin = zi >= 0 && zi < D;
        stzin = zin;
        if constexpr (INCL) {
            if (!zin) return;   // (uniform) a plane outside the volume: nothing loaded, the commit writes zeros
            const unsigned char* __restrict__ src = static_cast<const unsigned char*>(p.in) + (size_t)min(max(zi, 0), D - 1) * plane * (CI * 2);
#pragma unroll
            for (int rr = 0; rr < RH; ++rr) {
                const int r = r0 + rr;
                if (r >= r1) break;
#ifdef D3D_X_CONV0_NOLOAD   // timing-only build (wrong results): the channel-last input is not read -- what a layer whose input tile was already
                            // in the CU would cost (profiles/r05_sweep_conv0_fusion_bound.txt); reported by d3d_build_flags() (planesweep_window.hip)
                const u4 v = {stoff[r], (unsigned)r, (unsigned)zi, 0x3c003c00u};
                (void)src;
#else
                const u4 v = *reinterpret_cast<const u4*>(src + stoff[r]);
#endif
                stc[rr] = v;   // raw: zeroed for cells outside the volume when it is committed -- a select here would wait for the load
            }
            return;
        }
        if (!zin) return;   // (uniform) a plane outside the volume: nothing loaded, the commit writes zeros
        const float* __restrict__ srcp = static_cast<const float*>(p.in) + (size_t)zi * plane;
#pragma unroll
        for (int rr = 0; rr < RH; ++rr) {
            const int r = r0 + rr;
            if (r >= r1) break;
            const float* __restrict__ src = srcp + poff[INCL ? 0 : r];
#pragma unroll
            for (int k = 0; k < 8; ++k) stg[rr][k] = src[(size_t)k * vol];   // raw: zeroed when committed (a select here would wait for the load)
        }
    };
    auto commit = [&](unsigned char* dst, int r0, int r1) {   // registers -> bf16 cells
        if constexpr (INCL) {
#pragma unroll
            for (int rr = 0; rr < RH; ++rr) {
                const int r = r0 + rr;
                if (r >= r1) break;
                if (stdst[r] >= 0) *reinterpret_cast<u4*>(dst + stdst[r]) = stzin && stok[r] ? stc[rr] : (u4){0, 0, 0, 0};
            }
            return;
        }
#pragma unroll
        for (int rr = 0; rr < RH; ++rr) {
            const int r = r0 + rr;
            if (r >= r1) break;
            if (pdst[INCL ? 0 : r] >= 0) {
                const bool ok = stzin && pok[INCL ? 0 : r];
                float x[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) x[k] = ok ? stg[rr][k] : 0.0f;
                unsigned char* cell = dst + pdst[INCL ? 0 : r];
                if constexpr (X3) {   // hi = bf16(x), mid = bf16(x - hi), lo = bf16(x - hi - mid): the differences are exact in fp32
                    float r1[8], r2[8];
                    unsigned hi[4], mi[4], lo[4];
#pragma unroll
                    for (int k = 0; k < 8; k += 2) {
                        hi[k >> 1] = pack_bf16(x[k], x[k + 1]);
                        r1[k] = x[k] - __builtin_bit_cast(float, hi[k >> 1] << 16);
                        r1[k + 1] = x[k + 1] - __builtin_bit_cast(float, hi[k >> 1] & 0xffff0000u);
                        mi[k >> 1] = pack_bf16(r1[k], r1[k + 1]);
                        r2[k] = r1[k] - __builtin_bit_cast(float, mi[k >> 1] << 16);
                        r2[k + 1] = r1[k + 1] - __builtin_bit_cast(float, mi[k >> 1] & 0xffff0000u);
                        lo[k >> 1] = pack_bf16(r2[k], r2[k + 1]);
                    }
                    *reinterpret_cast<u4*>(cell) = (u4){hi[0], hi[1], hi[2], hi[3]};
                    *reinterpret_cast<u4*>(cell + CI * 2) = (u4){mi[0], mi[1], mi[2], mi[3]};
                    *reinterpret_cast<u4*>(cell + CI * 4) = (u4){lo[0], lo[1], lo[2], lo[3]};
                    continue;
                }
                *reinterpret_cast<u4*>(cell) = (u4){pack_h16(x[0], x[1]), pack_h16(x[2], x[3]), pack_h16(x[4], x[5]), pack_h16(x[6], x[7])};
            }
        }
    };

    constexpr int AW = MGN * NTN;   // accumulators per plane slot: [pixel group][channel tile]
    f4 acc[3][AW];
#pragma unroll
    for (int s = 0; s < 3; ++s)
#pragma unroll
        for (int i = 0; i < AW; ++i) acc[s][i] = (f4){0, 0, 0, 0};

    const int oy = y0 + wave;

    // plane-invariant state of the C_out = 8 channel-last epilogue
    struct { f4 sc, sh; unsigned oo[MGN]; bool st[MGN]; bool live; } c8s;
    if constexpr (CO8) {
        const int cb = (lane >> 4) * 4;
        c8s.live = cb < 8;
        c8s.sc = p.scale && c8s.live ? *reinterpret_cast<const f4*>(p.scale + cb) : (f4){1, 1, 1, 1};
        c8s.sh = p.shift && c8s.live ? *reinterpret_cast<const f4*>(p.shift + cb) : (f4){0, 0, 0, 0};
#pragma unroll
        for (int mg = 0; mg < MGN; ++mg) {
            const int ox = x0 + mg * 16 + (lane & 15);
            c8s.oo[mg] = ((unsigned)min(oy, H - 1) * W + min(ox, W - 1)) * 8 + (c8s.live ? cb : 0);   // bf16 elements inside an output plane
            c8s.st[mg] = cb == 0 && ox < W;
        }
    }
    auto store_plane = [&](int zo, f4 (&a)[AW]) {   // epilogue of a finished output plane
        if constexpr (OUTCL) {
            // operands swapped (weights as A): D row = channel (lane >> 4) * 4 + register, column = pixel lane & 15 -- a
            // lane owns four consecutive channels of one pixel = one 8-byte store into the pixel's cell
            if constexpr (CO8) {
              if (oy < H && zo >= z0 && zo < z1) {
                // C_out = 8: rows 8..15 of the tile are padding and half of the wave has nothing to store.  The lanes of
                // channels 4..7 hand their packed pair to the lanes of channels 0..3 of the same pixel (v_permlane16_swap:
                // row 1 -> row 0), which then write the pixel's whole 16-byte cell -- 16 lanes x 16 contiguous bytes
                // instead of 32 lanes x 8.  (affine, offsets inside the plane and store mask: per-lane state, see c8s)
                const size_t ob = (size_t)zo * plane * 8;
#pragma unroll
                for (int mg = 0; mg < MGN; ++mg) {
                    const size_t o = ob + c8s.oo[mg];
                    f4 v = a[mg] * c8s.sc + c8s.sh;
                    if (p.relu) v = __builtin_elementwise_max(v, (f4){0, 0, 0, 0});
                    if (p.skip && c8s.live) v += unpack_h16x4(*reinterpret_cast<const uint2*>(static_cast<const unsigned short*>(p.skip) + o));
                    const unsigned px = pack_h16(v[0], v[1]), py = pack_h16(v[2], v[3]);
                    const auto qx = __builtin_amdgcn_permlane16_swap(px, px, false, false);
                    const auto qy = __builtin_amdgcn_permlane16_swap(py, py, false, false);
                    if (c8s.st[mg]) {
                        typedef unsigned u4s __attribute__((ext_vector_type(4)));
                        *reinterpret_cast<u4s*>(static_cast<unsigned short*>(p.out) + o) = (u4s){px, py, qx[1], qy[1]};
                    }
                }
              }
            } else if (oy < H && zo >= z0 && zo < z1) {
#pragma unroll
                for (int nt = 0; nt < NTN; ++nt) {
                    const int cb = nt * 16 + (lane >> 4) * 4;
                    if (cb < p.CO) {   // C_out % 4 == 0
                        const f4 sc = p.scale ? *reinterpret_cast<const f4*>(p.scale + cb) : (f4){1, 1, 1, 1};
                        const f4 sh = p.shift ? *reinterpret_cast<const f4*>(p.shift + cb) : (f4){0, 0, 0, 0};
#pragma unroll
                        for (int mg = 0; mg < MGN; ++mg) {
                            const int ox = x0 + mg * 16 + (lane & 15);
                            if (ox < W) {
                                const size_t o = (((size_t)zo * H + oy) * W + ox) * p.CO + cb;   // bf16 element index
                                f4 v = a[mg * NTN + nt] * sc + sh;
                                if (p.relu) v = __builtin_elementwise_max(v, (f4){0, 0, 0, 0});
                                if (p.skip) v += unpack_h16x4(*reinterpret_cast<const uint2*>(static_cast<const unsigned short*>(p.skip) + o));
                                const uint2 pk = {pack_h16(v[0], v[1]), pack_h16(v[2], v[3])};
                                *reinterpret_cast<uint2*>(static_cast<unsigned short*>(p.out) + o) = pk;
                            }
                        }
                    }
                }
            }
        } else if (oy < H && zo >= z0 && zo < z1) {
            const float* __restrict__ skipf = static_cast<const float*>(p.skip);
            float* __restrict__ outf = static_cast<float*>(p.out);
#pragma unroll
            for (int nt = 0; nt < NTN; ++nt) {
                const int co = KZF ? 0 : nt * 16 + (lane & 15);
                if (KZF ? (lane & 15) == 2 : co < p.CO) {   // (KZF: column 2 = the plane that has seen its three input planes)
                    const float sc = p.scale ? p.scale[co] : 1.0f, sh = p.shift ? p.shift[co] : 0.0f;
#pragma unroll
                    for (int mg = 0; mg < MGN; ++mg) {
                        const int ox = x0 + mg * 16 + (lane >> 4) * 4;   // D layout: row (pixel) = (lane >> 4) * 4 + register
                        if (ox < W) {                                     // W % 4 == 0: a quad is inside or outside as a whole
                            const size_t o = (size_t)co * vol + (size_t)zo * plane + (size_t)oy * W + ox;
                            f4 v = a[mg * NTN + nt] * sc + sh;
                            if (p.relu) v = __builtin_elementwise_max(v, (f4){0, 0, 0, 0});
                            if (skipf) v += *reinterpret_cast<const f4*>(skipf + o);
                            *reinterpret_cast<f4*>(outf + o) = v;
                        }
                    }
                }
            }
        }
        if constexpr (KZF) {   // columns move up by one plane; column 0 starts the next plane at zero
#pragma unroll
            for (int i = 0; i < AW; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) a[i][r] = dpp_row_shr1(a[i][r]);
        } else {
#pragma unroll
            for (int i = 0; i < AW; ++i) a[i] = (f4){0, 0, 0, 0};
        }
    };

    // input plane zi feeds output planes zi+1 (k_z = 0), zi (k_z = 1), zi-1 (k_z = 2); slot of output plane zo = phase of zo
    auto sweep = [&](const unsigned char* buf, f4 (&up)[AW], f4 (&mid)[AW], f4 (&down)[AW], int kb0, int kb1) {
        int kgroup = lane >> 4;
        asm volatile("" : "+v"(kgroup));   // keeps the offsets out of long-lived registers
        auto kb_body = [&](int kb) {
            const int aoffk = WG ? a_offset(kb, kgroup) : aoffs[WG ? 0 : kb];
            if constexpr (KZF && X3) {   // one tile set (columns = k_z), three weight parts
                constexpr int WS1 = NKB * 64;
                const bf16x8 bh = __builtin_bit_cast(bf16x8, wsrc[kb * 64 + lane]);
                const bf16x8 bm = __builtin_bit_cast(bf16x8, wsrc[WS1 + kb * 64 + lane]);
                const bf16x8 bl = __builtin_bit_cast(bf16x8, wsrc[2 * WS1 + kb * 64 + lane]);
#pragma unroll
                for (int mg = 0; mg < MGN; ++mg) {
                    bf16x8 a[3];
#pragma unroll
                    for (int sp = 0; sp < 3; ++sp)
                        a[sp] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u4*>(buf + abase + mg * 16 * CS + aoffk + sp * CI * 2));
                    f4 c = up[mg];   // small terms first
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[2], bh, c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], bl, c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1], bm, c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1], bh, c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], bm, c, 0, 0, 0);
                    up[mg] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], bh, c, 0, 0, 0);
                }
                return;
            }
            if constexpr (KZF) {
                const h16x8 bw = __builtin_bit_cast(h16x8, wsrc[kb * 64 + lane]);
#pragma unroll
                for (int mg = 0; mg < MGN; ++mg) {
                    const h16x8 a = __builtin_bit_cast(h16x8, *reinterpret_cast<const u4*>(buf + abase + mg * 16 * CS + aoffk));
                    up[mg] = mfma_h16(a, bw, up[mg]);
                }
                return;
            }
            if constexpr (X3) {
                // weights [part s][k_z][K block][tile][lane]; A parts at + s * CI * 2 inside the cell
                constexpr int WS = 3 * NKB * NTN * 64;   // fragments per weight part
#pragma unroll
                for (int mg = 0; mg < MGN; ++mg) {
                    bf16x8 a[3];
#pragma unroll
                    for (int sp = 0; sp < 3; ++sp)
                        a[sp] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u4*>(buf + abase + mg * 16 * CS + aoffk + sp * CI * 2));
#pragma unroll
                    for (int kz = 0; kz < 3; ++kz) {
                        f4 (&dstacc)[AW] = kz == 0 ? up : kz == 1 ? mid : down;
#pragma unroll
                        for (int nt = 0; nt < NTN; ++nt) {
                            const int wi = ((kz * NKB + kb) * NTN + nt) * 64 + lane;
                            const bf16x8 bh = __builtin_bit_cast(bf16x8, wsrc[wi]);
                            const bf16x8 bm = __builtin_bit_cast(bf16x8, wsrc[WS + wi]);
                            const bf16x8 bl = __builtin_bit_cast(bf16x8, wsrc[2 * WS + wi]);
                            f4 c = dstacc[mg * NTN + nt];   // small terms first
                            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[2], bh, c, 0, 0, 0);
                            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], bl, c, 0, 0, 0);
                            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1], bm, c, 0, 0, 0);
                            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1], bh, c, 0, 0, 0);
                            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], bm, c, 0, 0, 0);
                            dstacc[mg * NTN + nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], bh, c, 0, 0, 0);
                        }
                    }
                }
                return;
            }
            h16x8 b0[NTN], b1[NTN], b2[NTN];
#pragma unroll
            for (int nt = 0; nt < NTN; ++nt) {
                b0[nt] = __builtin_bit_cast(h16x8, wsrc[((0 * NKB + kb) * NTN + nt) * 64 + lane]);
                b1[nt] = __builtin_bit_cast(h16x8, wsrc[((1 * NKB + kb) * NTN + nt) * 64 + lane]);
                b2[nt] = __builtin_bit_cast(h16x8, wsrc[((2 * NKB + kb) * NTN + nt) * 64 + lane]);
            }
#pragma unroll
            for (int mg = 0; mg < MGN; ++mg) {
                const h16x8 a = __builtin_bit_cast(h16x8, *reinterpret_cast<const u4*>(buf + abase + mg * 16 * CS + aoffk));
#pragma unroll
                for (int nt = 0; nt < NTN; ++nt) {
                    // (the A and B fragment layouts are the same, so D^T costs nothing: weights first = channel rows)
                    if constexpr (OUTCL) {
                        up[mg * NTN + nt] = mfma_h16(b0[nt], a, up[mg * NTN + nt]);
                        mid[mg * NTN + nt] = mfma_h16(b1[nt], a, mid[mg * NTN + nt]);
                        down[mg * NTN + nt] = mfma_h16(b2[nt], a, down[mg * NTN + nt]);
                    } else {
                        up[mg * NTN + nt] = mfma_h16(a, b0[nt], up[mg * NTN + nt]);
                        mid[mg * NTN + nt] = mfma_h16(a, b1[nt], mid[mg * NTN + nt]);
                        down[mg * NTN + nt] = mfma_h16(a, b2[nt], down[mg * NTN + nt]);
                    }
                }
            }
        };
        if constexpr (WG) {   // fragments come from L2: keep ONE K block's loads in flight, not all of them (registers)
#pragma unroll 1
            for (int kb = kb0; kb < kb1; ++kb) kb_body(kb);
        } else {
#pragma unroll
            for (int kb = 0; kb < NKB; ++kb) {
                if (kb < kb0 || kb >= kb1) continue;
                kb_body(kb);
            }
        }
    };

    // ---- z walk: input planes z0-1 .. z1 ----------------------------------------------------------------------
    issue(z0 - 1, 0, RH);
    commit(smem, 0, RH);
    if (ROUNDS > RH) {
        issue(z0 - 1, RH, ROUNDS);
        commit(smem, RH, ROUNDS);
    }
    __syncthreads();
    int cur = 0;
    auto step = [&](int zi, f4 (&up)[AW], f4 (&mid)[AW], f4 (&down)[AW]) {   // up: zi+1, mid: zi, down: zi-1
        const bool more = zi + 1 <= z1, live = zi >= 0 && zi < D;
        constexpr int KH = (NKB + 1) / 2;
        if constexpr (NBUF == 1) {
            if (more) issue(zi + 1, 0, ROUNDS);        // the next plane waits in registers
            if (live) sweep(smem, up, mid, down, 0, NKB);
            store_plane(zi - 1, down);
            lds_barrier();                             // every wave has read the patch (LDS only: the plane's stores stay in flight)
            if (more) commit(smem, 0, ROUNDS);
            lds_barrier();
            return;
        }
        if (more) issue(zi + 1, 0, RH);                // the next plane's loads fly during the MFMA sweep
        if (live) sweep(smem + cur * PATCH, up, mid, down, 0, KH);
        if (more) commit(smem + (cur ^ 1) * PATCH, 0, RH);
        if (more && ROUNDS > RH) issue(zi + 1, RH, ROUNDS);
        if (live) sweep(smem + cur * PATCH, up, mid, down, KH, NKB);
        store_plane(zi - 1, down);                     // complete: it has seen input planes zi-2, zi-1, zi
        if (more && ROUNDS > RH) commit(smem + (cur ^ 1) * PATCH, RH, ROUNDS);
        lds_barrier();                                 // (LDS only: the stores of the finished plane stay in flight, common.h)
        cur ^= 1;
    };
    if constexpr (KZF) {
        for (int zi = z0 - 1; zi <= z1; ++zi) step(zi, acc[0], acc[0], acc[0]);   // one accumulator set: columns = planes
    } else {
        for (int zi = z0 - 1; zi <= z1; zi += 3) {
            step(zi, acc[1], acc[0], acc[2]);
            if (zi + 1 <= z1) step(zi + 1, acc[2], acc[1], acc[0]);
            if (zi + 2 <= z1) step(zi + 2, acc[0], acc[2], acc[1]);
        }
    }
}

template <int CI, int NTN, bool INCL, bool OUTCL, int MGN = 4, bool WG = false, bool KZF = false, bool CO8 = false, bool X3 = false>
static int launch(const C8Params& p, hipStream_t stream) {
    if constexpr (OUTCL && NTN == 1 && !KZF && !WG && !CO8)
        if (p.CO == 8) return launch<CI, NTN, INCL, OUTCL, MGN, WG, KZF, true>(p, stream);   // whole-cell stores (see store_plane)
    constexpr int NKB = (9 * CI + 31) / 32;
    constexpr bool PL = !X3 && CI == 32;
    constexpr int CS = X3 ? (CI == 16 ? 96 : 6 * CI + (CI > 8 ? 16 : 0)) : PL ? 32 : CI == 64 ? 144 : bf16_cell_bytes(CI);
    constexpr int TXk = 16 * MGN, PXk = TXk + 2;
    constexpr int lds = (X3 && CI >= 16 ? 1 : 2) * (PL ? (CI / 16) * (PXk * PY * CS + 64) : PXk * PY * CS) + (WG ? 0 : (X3 ? 3 : 1) * (KZF ? 1 : 3) * NKB * NTN * 64 * 16);
    static_assert(lds <= 160 * 1024, "tile does not fit the LDS");
    auto kern = conv3d_c8_bf16_kernel<CI, NTN, INCL, OUTCL, MGN, WG, KZF, CO8, X3>;
    int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(kern), lds);
    if (rc != D3D_OK) return rc;
    C8Params q = p;
    const int gx = ceil_div(p.W, TXk), gy = ceil_div(p.H, TY);
    q.zper = pick_zper((long)gx * gy, p.D, 4, 2, lds);   // (every z segment re-reads two halo planes; common.h)
    hipLaunchKernelGGL(kern, dim3(gx, gy, ceil_div(p.D, q.zper)), dim3(NT), lds, stream, q);
    D3D_LAUNCH_CHECK("conv3d_c8_bf16_kernel launch");
    return D3D_OK;
}

}  // namespace

}  // namespace d3d

using namespace d3d;

template <bool INCL, bool OUTCL>
static int launch_fmt(const C8Params& p, int Ci, int Co, hipStream_t st) {
    if (Co > 16) return launch<32, 2, INCL, OUTCL>(p, st);
    switch (Ci) {
        case 8: return launch<8, 1, INCL, OUTCL>(p, st);   // (32-wide tiles: 0.63 -> 0.74 ms at 8 x 1856 x 2752 -- four workgroups per CU either way)
        case 16:
            // channel-last in and out: 32-wide tiles keep the kernel inside 128 registers (102), so two workgroups share a CU
            // (16 -> 8 at 32 x 928 x 1376: 1.09 -> 0.95 ms on the same box)
            if constexpr (INCL && OUTCL) return launch<16, 1, true, true, 2>(p, st);   // (64-wide tiles, round 4 again: 0.675 -> 0.70-0.72 ms at stage 2)
            else return launch<16, 1, INCL, OUTCL>(p, st);
        default: return launch<32, 1, INCL, OUTCL>(p, st);
    }
}

extern "C" int d3d_conv3d_k3_cl_h16(const void* in, int in_cl, const void* wpacked, const float* scale, const float* shift,
                                     const void* skip, int relu, int Ci, int Co, int D, int H, int W, void* out, int out_cl,
                                     d3d_stream_t stream) {
    D3D_REQUIRE(in && wpacked && out, "null pointer");
    D3D_REQUIRE(D > 0 && H > 0 && W > 0, "bad dims %dx%dx%d", D, H, W);
    const bool wide = Ci == 64 && Co == 64 && in_cl && out_cl;   // conv6: channel-last only, weights streamed from L2
    const bool shape = ((Ci == 8 || Ci == 16 || Ci == 32) && Co >= 1 && (Co <= 16 || (Co == 32 && Ci == 32))) || wide;
    if (!shape || (out_cl ? Co % 4 != 0 : W % 4 != 0) || ceil_div(H, TY) > 65535 || D > 65535) {
        set_error("d3d_conv3d_k3_cl_h16: C_in = %d (8 | 16 | 32), C_out = %d (<= 16, or 32 with C_in = 32; a multiple of 4 for "
                  "channel-last output; 64 -> 64 channel-last), W = %d (a multiple of 4 for planar output) not taken", Ci, Co, W);
        return D3D_ERR_UNSUPPORTED;
    }
    C8Params p = {};
    p.in = in; p.wpk = reinterpret_cast<const u4*>(wpacked); p.scale = scale; p.shift = shift; p.skip = skip; p.out = out;
    p.D = D; p.H = H; p.W = W; p.relu = relu; p.CO = Co;
    p.in_cl8 = in_cl == 2;   // (in_cl: 0 planar fp32 | 1 channel-last bf16 [D,H,W,Ci] | 2 CL8 [D,Ci/8,H,W,8])
    hipStream_t st = (hipStream_t)stream;
    if (wide) return launch<64, 4, true, true, 2, true>(p, st);
    if (in_cl) return out_cl ? launch_fmt<true, true>(p, Ci, Co, st) : launch_fmt<true, false>(p, Ci, Co, st);
    return out_cl ? launch_fmt<false, true>(p, Ci, Co, st) : launch_fmt<false, false>(p, Ci, Co, st);
}

extern "C" int d3d_conv3d_k3_c1_cl_h16(const void* in, int in_cl, const void* wpacked, const float* scale, const float* shift,
                                        const float* skip, int relu, int Ci, int D, int H, int W, float* out, d3d_stream_t stream) {
    D3D_REQUIRE(in && wpacked && out, "null pointer");
    D3D_REQUIRE(D > 0 && H > 0 && W > 0, "bad dims %dx%dx%d", D, H, W);
    if ((Ci != 8 && Ci != 16 && Ci != 32) || W % 4 != 0 || ceil_div(H, TY) > 65535 || D > 65535) {
        set_error("d3d_conv3d_k3_c1_cl_h16: C_in = %d (8 | 16 | 32), W = %d (a multiple of 4) not taken", Ci, W);
        return D3D_ERR_UNSUPPORTED;
    }
    C8Params p = {};
    p.in = in; p.wpk = reinterpret_cast<const u4*>(wpacked); p.scale = scale; p.shift = shift; p.skip = skip; p.out = out;
    p.D = D; p.H = H; p.W = W; p.relu = relu; p.CO = 1;
    p.in_cl8 = in_cl == 2;
    hipStream_t st = (hipStream_t)stream;
    if (in_cl) {
        switch (Ci) {
            case 8: return launch<8, 1, true, false, 4, false, true>(p, st);
            case 16: return launch<16, 1, true, false, 4, false, true>(p, st);
            default: return launch<32, 1, true, false, 4, false, true>(p, st);
        }
    }
    switch (Ci) {
        case 8: return launch<8, 1, false, false, 4, false, true>(p, st);
        case 16: return launch<16, 1, false, false, 4, false, true>(p, st);
        default: return launch<32, 1, false, false, 4, false, true>(p, st);
    }
}

extern "C" int d3d_conv3d_k3_c1_bf16x3(const float* in, const void* wpacked, const float* scale, const float* shift, const float* skip,
                                        int relu, int Ci, int D, int H, int W, float* out, d3d_stream_t stream) {
    D3D_REQUIRE(in && wpacked && out, "null pointer");
    D3D_REQUIRE(D > 0 && H > 0 && W > 0, "bad dims %dx%dx%d", D, H, W);
    if (Ci != 8 || W % 4 != 0 || ceil_div(H, TY) > 65535 || D > 65535) {
        set_error("d3d_conv3d_k3_c1_bf16x3: C_in = %d (8), W = %d (a multiple of 4) not taken", Ci, W);
        return D3D_ERR_UNSUPPORTED;
    }
    C8Params p = {};
    p.in = in; p.wpk = reinterpret_cast<const u4*>(wpacked); p.scale = scale; p.shift = shift; p.skip = skip; p.out = out;
    p.D = D; p.H = H; p.W = W; p.relu = relu; p.CO = 1;
    return launch<8, 1, false, false, 4, false, true, false, true>(p, (hipStream_t)stream);
}

extern "C" int d3d_conv3d_k3_zs_bf16x3(const float* in, const void* wpacked, const float* scale, const float* shift,
                                        const float* skip, int relu, int Ci, int Co, int D, int H, int W, float* out,
                                        d3d_stream_t stream) {
    D3D_REQUIRE(in && wpacked && out, "null pointer");
    D3D_REQUIRE(D > 0 && H > 0 && W > 0, "bad dims %dx%dx%d", D, H, W);
    const bool wide = (Ci == 32 && Co == 32) || (Ci == 64 && Co == 64);   // conv4 / conv6 (round 4): weight fragments from L2
    if (((Ci != 8 && Ci != 16 && Ci != 32) || Co < 1 || Co > 16) && !wide) {
        set_error("d3d_conv3d_k3_zs_bf16x3: C_in = %d (8 | 16 | 32), C_out = %d (<= 16; 32 -> 32, 64 -> 64) not taken", Ci, Co);
        return D3D_ERR_UNSUPPORTED;
    }
    if (W % 4 != 0 || ceil_div(H, TY) > 65535 || D > 65535) {
        set_error("d3d_conv3d_k3_zs_bf16x3: W = %d (a multiple of 4) not taken", W);
        return D3D_ERR_UNSUPPORTED;
    }
    C8Params p = {};
    p.in = in; p.wpk = reinterpret_cast<const u4*>(wpacked); p.scale = scale; p.shift = shift; p.skip = skip; p.out = out;
    p.D = D; p.H = H; p.W = W; p.relu = relu; p.CO = Co;
    hipStream_t st = (hipStream_t)stream;
    // 32-wide tiles (split cells are three times as wide); 16 | 32 input channels: one patch buffer beside the weights
    if (Ci == 32 && Co == 32) return launch<32, 2, false, false, 2, true, false, false, true>(p, st);
    if (Ci == 64) return launch<64, 4, false, false, 1, true, false, false, true>(p, st);
    switch (Ci) {
        case 8: return launch<8, 1, false, false, 2, false, false, false, true>(p, st);
        case 16: return launch<16, 1, false, false, 2, false, false, false, true>(p, st);
        default: return launch<32, 1, false, false, 2, false, false, false, true>(p, st);
    }
}

extern "C" int d3d_conv3d_k3_zs_h16(const float* in, const void* wpacked, const float* scale, const float* shift,
                                     const float* skip, int relu, int Ci, int Co, int D, int H, int W, float* out,
                                     d3d_stream_t stream) {
    return d3d_conv3d_k3_cl_h16(in, 0, wpacked, scale, shift, skip, relu, Ci, Co, D, H, W, out, 0, stream);
}

extern "C" int d3d_conv3d_k3_c8_h16(const float* in, const void* wpacked, const float* scale, const float* shift,
                                     const float* skip, int relu, int Ci, int D, int H, int W, float* out,
                                     d3d_stream_t stream) {
    return d3d_conv3d_k3_cl_h16(in, 0, wpacked, scale, shift, skip, relu, Ci, 8, D, H, W, out, 0, stream);
}
