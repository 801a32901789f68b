// 3x3 stride-1 2-D convolution on the bf16 matrix cores (v_mfma_f32_16x16x32_bf16, fp32 accumulation) for the large image
// layers of the slice regularisers in bf16 mode (BASELINE config 3): ConvReLU(C, 8) and the two convolutions of a conv-GRU
// cell with their fused epilogues (adamvs.py:403-427 SliceCostRegNetRED, module.py:5-51 ConvGRUCell) --
//     gates:      g = conv(cat(x, h)) + b;  out = [sigmoid(g_r) * h | sigmoid(g_u)]
//     candidate:  c = conv(cat(x, r*h)) + b;  h' = u * h + (1 - u) * tanh(c)
// Tensors stay planar fp32 in HBM (the recurrent state keeps its precision; only the MFMA operands are bf16, as in every
// bf16-mode kernel), the channel concat is never materialised (two input pointers).
//
// Why its own kernel: the round-1 stream kernel walks an image row by row (one barrier and one epilogue per row and tile
// column); at these shapes its waves wait 64 % of their cycles and the matrix cores are 7 % busy
// (profiles/r02_gru_conv_counters.txt).  Here a workgroup (8 waves) takes 64 x 8 (or 32 x 8) output pixels per step: the
// whole 66 x 10 patch is staged at once (fp32 planar -> bf16 channel-last cells, RNE), all nine taps are one sweep
// (M = 16 consecutive pixels of a row, N = 16 output channels, K = (k_y, k_x, c_in) in blocks of 32, an A operand = one
// ds_read_b128 of 8 channels), and the workgroup walks `tper` tiles down the image with the next patch's loads in flight
// during the sweep.  D[pixel][channel]: a lane leaves with four consecutive pixels of one channel = 16-byte loads of
// h / u and 16-byte stores.
#include <cstdint>
#include "common.h"
#include "gn_stats.h"

#include <type_traits>

namespace d3d {

namespace {

typedef float f4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u4 __attribute__((ext_vector_type(4)));
typedef unsigned u2 __attribute__((ext_vector_type(2)));

constexpr int TYZ = 8;              // output rows of a tile = waves
constexpr int PYZ = TYZ + 2;        // staged rows
constexpr int NTZ = 64 * TYZ;

struct Z2Params {
    const float* in;      // [C1, H, W]
    const float* in2;     // [CI - C1, H, W] or null: second part of the channel concat
    int C1;
    const u4* wpk;        // [NKB][N tiles][64 lanes] B fragments (ops._pack_z2_bf16)
    const float* scale;   // [CO] or null
    const float* shift;   // [CO] or null (the bias)
    const float* skip;    // act 0 | 1: [CO,H,W] added before / after the activation, or null; act 2: h [ep_split,H,W]; act 3: h [CO,H,W]
    const float* aux1;    // act 3: the update gate u [CO,H,W]
    float* out;           // [CO, H, W]
    int H, W, CO;
    int act;              // 0 none | 1 ReLU | 2 GRU gates | 3 GRU update
    int ep_split;         // act 2: channels below it are multiplied by h
    int skip_after_act;
    int tper;             // tiles per workgroup along y
    int sub2;             // stride-1 kernel: keep the even rows / columns only = the stride-2 layer (act 0 | 1; out, skip [CO,(H+1)/2,W/2])
    // IMG3 form (d3d_conv2d_k3_pair3_bf16x3): `in` is not read -- the 8-channel input is the 3 -> 8 layer of the 3-channel image below
    const float* img;     // [3, H, W]
    const float* w0;      // [4][3][3][8] fp32: the 3 -> 8 layer's weights as d3d_conv2d_k3_stream packs them (channel 3 = zeros)
    const float* scale0;  // [8] or null
    const float* shift0;  // [8] or null
    int act0;             // 0 | 1 (ReLU)
    // GroupNorm statistics of the output (gn_stats.h; act 0, no skip): fp64 pairs, zeroed by the caller; channels >= gn_split are group 1
    double* gn;
    int gn_split;
    // batched launch (round 5: RED-Net's encoder for every depth slice of a stage at once -- blockIdx.z is the item): element strides of
    // `in` and `out` between items; 0 for the single-item launches.  (in2 / skip / aux1 / gn are not batched.)
    long in_bstride, out_bstride;
    int nbatch;
};

__device__ __forceinline__ unsigned pack_bf16_z2(float a, float b) {   // the split (fp32-mode) operands: three bf16 pieces
    return pack_bf16x2(a, b);   // one v_cvt_pk_bf16_f32 (common.h)
}
__device__ __forceinline__ unsigned pack_h16_z2(float a, float b) {    // the single 16-bit operand (common.h)
    return pack_h16x2(a, b);
}
// Global memory through buffer instructions: a raw buffer from `origin` on (which may lie before the tensor) + a 32-bit per-lane
// byte offset + a scalar offset.  A lane whose pixels are outside the image carries OOBZ: the load returns zeros, the store is
// dropped -- no address arithmetic and no select on the vector unit.
constexpr unsigned OOBZ = 0xffffffffu;
__device__ __forceinline__ __amdgpu_buffer_rsrc_t z2_rsrc(const void* origin) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(origin), 0, (int)0xfffffffeu, 0x00020000);
}

// F32: the same tile scheme with exact fp32 operands (v_mfma_f32_16x16x4_f32, the default precision of the models): cells
// of C_in floats + 16 bytes (20 | 36-dword pitches: the 16 pixels of an A operand, one dword each, fall in 16 different
// banks, and so do the four K groups), K blocks of 4 channels of one tap, weights fp32 in LDS.
//
// X3: fp32 accuracy on the bf16 matrix cores.  An fp32 value is the exact sum of three bf16 numbers (hi = rne(v), mid =
// rne(v - hi), lo = rne(v - hi - mid): 3 x 8 significand bits); activations are split while they are staged (a cell holds
// the hi | mid | lo runs of its C_in channels), weights on the host, and a K block of 32 takes the six products whose weight
// is >= 2^-16 of the leading one (hi*hi, hi*mid, mid*hi, mid*mid, hi*lo, lo*hi: what is dropped is <= 2^-24 of the product,
// fp32's own rounding) -- 6 x 16 cycles against the 8 x 32 cycles of the fp32 instruction for the same K.
template <bool F32, bool X3>
__host__ __device__ constexpr int z2_cell_bytes(int CI) {
    // bf16: an odd number of 16-byte slots (24 and 40 channels -- RED-Net's finest conv-GRU level at stages 2 / 1 -- are 3 and 5 already)
    return F32 ? CI * 4 + 16 : (X3 ? 3 : 1) * CI * 2 + ((CI > 8 && CI != 24 && CI != 40) || X3 ? (CI > 8 ? 16 : 0) : 0);
}

__device__ __forceinline__ void split3_bf16(float a, float b, unsigned& hi, unsigned& mid, unsigned& lo) {
    hi = pack_bf16_z2(a, b);
    const float ra = a - __builtin_bit_cast(float, hi << 16), rb = b - __builtin_bit_cast(float, hi & 0xffff0000u);   // exact
    mid = pack_bf16_z2(ra, rb);
    lo = pack_bf16_z2(ra - __builtin_bit_cast(float, mid << 16), rb - __builtin_bit_cast(float, mid & 0xffff0000u));
}

// eight channels of one pixel -> the hi | mid | lo runs of its split cell (`cell` points at the group's slot of the hi run)
__device__ __forceinline__ void put8_split3(unsigned char* cell, int CI, const float (&v)[8]) {
    unsigned hi[4], mid[4], lo[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) split3_bf16(v[2 * k], v[2 * k + 1], hi[k], mid[k], lo[k]);
    *reinterpret_cast<u4*>(cell) = (u4){hi[0], hi[1], hi[2], hi[3]};
    *reinterpret_cast<u4*>(cell + CI * 2) = (u4){mid[0], mid[1], mid[2], mid[3]};
    *reinterpret_cast<u4*>(cell + CI * 4) = (u4){lo[0], lo[1], lo[2], lo[3]};
}

// the six products of a split K block, small terms first
__device__ __forceinline__ f4 mfma_split3(const bf16x8 (&a)[3], bf16x8 bh, bf16x8 bm, bf16x8 bl, f4 c) {
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[2], bh, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], bl, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1], bm, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1], bh, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], bm, c, 0, 0, 0);
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], bh, c, 0, 0, 0);
}

// IMG3 (CI = 8, split operands): the first two layers of a feature trunk in one launch (module.py:663-666 conv0 = ConvBnReLU(3, 8)
// + ConvBnReLU(8, 8) at full resolution).  The 8-channel input never exists in memory: the staging phase loads the 3-channel
// image patch (12 x 68 per tile), the workgroup evaluates the first layer at the 10 x 66 patch positions with the vector-unit
// arithmetic of conv2d_stream_kernel (conv.hip: same order c, k_y, k_x of the fused multiply-adds, scale, shift, ReLU) and writes
// the split cells the matrix-core sweep reads -- what the two launches compute, bit for bit, without the 8 x H x W tensor's write
// and read (326 of the pair's 550 MB at 2752 x 1856).
// (GN instance of 40 channels: 124 registers without the statistics' fp64 partial sums -- held to 128, i.e. to two workgroups per CU)
template <int CI, int NTN, int MGN, bool F32 = false, bool X3 = false, bool IMG3 = false, bool GN = false>
__global__ __launch_bounds__(NTZ, GN && CI == 40 ? 4 : 2) void conv2d_zs_bf16_kernel(Z2Params p) {
    static_assert(!(F32 && X3), "one operand format");
    static_assert(!IMG3 || (CI == 8 && X3 && NTN == 1), "image form: 3 -> 8 -> CO <= 16 on split operands");
    constexpr int TX = 16 * MGN, PX = TX + 2;
    constexpr int NKB = F32 ? 9 * CI / 4 : (9 * CI + 31) / 32;
    constexpr int NSPL = X3 ? 3 : 1;
    constexpr int CS = z2_cell_bytes<F32, X3>(CI);
    constexpr int G = CI / 8;
    constexpr int PATCH = PX * PYZ * CS;
    // X3: one patch buffer (commit after the sweep, two barriers per tile) -- the split cells are 1.4 x the fp32 ones and
    // a second buffer would leave one workgroup per CU; two or three resident workgroups overlap their phases instead
    constexpr int NBUF = X3 || (F32 && CI > 32) ? 1 : 2;   // (fp32 cells of 48 channels: 71 KB of patch beside 83 KB of weights)
    constexpr int AW = MGN * NTN;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    u4* wlds = reinterpret_cast<u4*>(smem + NBUF * PATCH);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (p.nbatch > 1) { p.in += (size_t)blockIdx.z * p.in_bstride; p.out += (size_t)blockIdx.z * p.out_bstride; }   // (batched launch: this workgroup's item)
    const int H = p.H, W = p.W;
    const int x0 = blockIdx.x * TX;
    const int nty = (H + TYZ - 1) / TYZ;
    const int t0 = blockIdx.y * p.tper, t1 = min(t0 + p.tper, nty);

    if constexpr (F32) {
        float* wf = reinterpret_cast<float*>(wlds);
        const float* wg = reinterpret_cast<const float*>(p.wpk);
        for (int i = tid; i < NKB * NTN * 64; i += NTZ) wf[i] = wg[i];
    } else {
        for (int i = tid; i < NSPL * NKB * NTN * 64; i += NTZ) wlds[i] = p.wpk[i];   // X3: [hi | mid | lo][NKB][N tiles][lane]
    }

    // ---- staging with 16-byte loads.  A quad task = (4 channels, patch row, 4 aligned columns x0 + 4q ..): four dwordx4 loads,
    // the lanes of a wave reading whole 128-byte runs (W % 4 == 0: a quad is inside or outside the image as a whole); an edge
    // task = (4 channels, patch row, halo column x0 - 1 | x0 + TX): four dword loads, given to the threads the quads leave idle.
    // A channel quad comes from `in` or from `in2` (C1 % 8 == 0).  Task order: channel quad fastest, then column quad, then row:
    // a load instruction still reads whole 128-byte runs (one per channel and row), and the lanes of a cell write of the
    // commit cover the C4 adjacent 8-byte (fp32: 16-byte) slots of one cell before moving 4 cells on.
    // The kernel is bound by instruction issue (profiles/r04_t2p.txt, r04_gru_slice.txt), so what a task reads and writes is
    // per-lane state computed once: its byte offset from the patch origin inside its source tensor (OOB where its column is
    // outside the image: a buffer load then returns zeros), its patch row, its cell.  Per tile only the rows outside the image
    // turn offsets OOB (inner tiles skip even that), and the values stay raw in registers until the commit -- a select behind
    // the load would wait for it before the sweep it is meant to hide behind.
    constexpr int C4 = CI / 4, QX = TX / 4;
    constexpr int NQT = C4 * PYZ * QX, RQ = (NQT + NTZ - 1) / NTZ;
    constexpr int NST = C4 * PYZ * 2, RS = (NST + NTZ - 1) / NTZ;
    (void)G;
    const unsigned plane4 = (unsigned)H * W * 4;   // bytes of a channel plane (host: every tensor < 2^31 bytes)
    const int c14 = p.in2 ? p.C1 / 4 : C4;         // channel quads below it come from `in`
    f4 stq[RQ][4];      // [channel of the quad][4 pixels]
    float sts[RS][4];   // [channel of the quad]
    unsigned qvo[RQ], svo[RS];
    int qpy[RQ], spy[RS], qcell[RQ], scell[RS];
    bool q2[RQ], s2[RS];   // the task's channels come from in2
    const int tid_e = NTZ - 1 - tid;
#pragma unroll
    for (int r = 0; r < RQ; ++r) {
        const int task = tid + r * NTZ;
        const int c4 = task % C4, rest = task / C4, q = rest % QX, py = rest / QX;
        q2[r] = c4 >= c14;
        qvo[r] = task < NQT && x0 + 4 * q < W ? (unsigned)(4 * (q2[r] ? c4 - c14 : c4)) * plane4 + (unsigned)(py * W + 4 * q) * 4 : OOBZ;
        qpy[r] = py;
        qcell[r] = task < NQT ? (py * PX + 1 + 4 * q) * CS : -1;
    }
#pragma unroll
    for (int r = 0; r < RS; ++r) {
        const int task = tid_e + r * NTZ;
        const int c4 = task % C4, rest = task / C4, side = rest & 1, py = rest >> 1;
        const int gx = side ? x0 + TX : x0 - 1;
        s2[r] = c4 >= c14;
        svo[r] = task < NST && gx >= 0 && gx < W ? (unsigned)(4 * (s2[r] ? c4 - c14 : c4)) * plane4 + (unsigned)(py * W + gx - x0 + 1) * 4 : OOBZ;
        spy[r] = py;
        scell[r] = task < NST ? (py * PX + (side ? PX - 1 : 0)) * CS : -1;
    }
    int qc4[RQ], sc4[RS];
#pragma unroll
    for (int r = 0; r < RQ; ++r) qc4[r] = (tid + r * NTZ) % C4;
#pragma unroll
    for (int r = 0; r < RS; ++r) sc4[r] = (tid_e + r * NTZ) % C4;
    auto issue = [&](int ty) {
        const int gy0 = ty * TYZ - 1;                                  // patch origin: row gy0, column x0 - 1 (edge tasks) | x0 (quads)
        const bool inner = gy0 >= 0 && gy0 + PYZ <= H;
        const __amdgpu_buffer_rsrc_t rq1 = z2_rsrc(p.in + ((long)gy0 * W + x0)), rs1 = z2_rsrc(p.in + ((long)gy0 * W + x0 - 1));
        const __amdgpu_buffer_rsrc_t rq2 = z2_rsrc((p.in2 ? p.in2 : p.in) + ((long)gy0 * W + x0)), rs2 = z2_rsrc((p.in2 ? p.in2 : p.in) + ((long)gy0 * W + x0 - 1));
        // Two sources: the descriptor of a buffer instruction is scalar, and a per-lane choice between two of them is compiled
        // into a loop over the distinct descriptors around every load (two passes when a wave's tasks straddle the tensors).  That
        // is what the two-tensor layers get (asking both tensors with an OOB offset for the wrong one and OR-ing the answers was
        // measured: 725 against 562 us for the three launches of an unfused stage-3 cell); a layer with ONE input must not pay
        // for it, hence the uniform branch.
        if (p.in2) {
#pragma unroll
            for (int r = 0; r < RQ; ++r) {
                unsigned vo = qvo[r];
                if (!inner) vo = (unsigned)(qpy[r] + gy0) < (unsigned)H ? vo : OOBZ;
                const __amdgpu_buffer_rsrc_t rs = q2[r] ? rq2 : rq1;
#pragma unroll
                for (int k = 0; k < 4; ++k) stq[r][k] = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(rs, vo, k * plane4, 0));
            }
#pragma unroll
            for (int r = 0; r < RS; ++r) {
                unsigned vo = svo[r];
                if (!inner) vo = (unsigned)(spy[r] + gy0) < (unsigned)H ? vo : OOBZ;
                const __amdgpu_buffer_rsrc_t rs = s2[r] ? rs2 : rs1;
#pragma unroll
                for (int k = 0; k < 4; ++k) sts[r][k] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, vo, k * plane4, 0));
            }
            return;
        }
#pragma unroll
        for (int r = 0; r < RQ; ++r) {
            unsigned vo = qvo[r];
            if (!inner) vo = (unsigned)(qpy[r] + gy0) < (unsigned)H ? vo : OOBZ;
#pragma unroll
            for (int k = 0; k < 4; ++k) stq[r][k] = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(rq1, vo, k * plane4, 0));
        }
#pragma unroll
        for (int r = 0; r < RS; ++r) {
            unsigned vo = svo[r];
            if (!inner) vo = (unsigned)(spy[r] + gy0) < (unsigned)H ? vo : OOBZ;
#pragma unroll
            for (int k = 0; k < 4; ++k) sts[r][k] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs1, vo, k * plane4, 0));
        }
    };
    // one pixel's four channels 4 c4 .. + 3 into its cell
    auto put4 = [&](unsigned char* cell, int c4, float v0, float v1, float v2, float v3) {
        if constexpr (F32) {
            *reinterpret_cast<f4*>(cell + c4 * 16) = (f4){v0, v1, v2, v3};
        } else if constexpr (X3) {
            unsigned h0, m0, l0, h1, m1, l1;
            split3_bf16(v0, v1, h0, m0, l0);
            split3_bf16(v2, v3, h1, m1, l1);
            *reinterpret_cast<u2*>(cell + c4 * 8) = (u2){h0, h1};
            *reinterpret_cast<u2*>(cell + CI * 2 + c4 * 8) = (u2){m0, m1};
            *reinterpret_cast<u2*>(cell + CI * 4 + c4 * 8) = (u2){l0, l1};
        } else {
            *reinterpret_cast<u2*>(cell + c4 * 8) = (u2){pack_h16_z2(v0, v1), pack_h16_z2(v2, v3)};
        }
    };
    auto commit = [&](unsigned char* dst) {
#pragma unroll
        for (int r = 0; r < RQ; ++r) {
            if (qcell[r] >= 0) {
                unsigned char* cell = dst + qcell[r];
#pragma unroll
                for (int i = 0; i < 4; ++i) put4(cell + i * CS, qc4[r], stq[r][0][i], stq[r][1][i], stq[r][2][i], stq[r][3][i]);
            }
        }
#pragma unroll
        for (int r = 0; r < RS; ++r)
            if (scell[r] >= 0) put4(dst + scell[r], sc4[r], sts[r][0], sts[r][1], sts[r][2], sts[r][3]);
    };

    // K index k = 32 kb + 8 (lane >> 4) + j -> tap t = k / CI = (k_y, k_x), channel k % CI: one register per K block, computed once
    const int abase = (wave * PX + (lane & 15)) * CS + (F32 ? (lane >> 4) * 4 : 0);   // F32: K group = channel within the block of 4
    int aoffs[F32 ? 1 : NKB];
    if constexpr (!F32) {
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb) {
            const int k0 = 32 * kb + 8 * (lane >> 4);
            const int t = k0 / CI, c = k0 % CI;
            const int ky = t < 9 ? t / 3 : 0, kx = t < 9 ? t % 3 : 0;   // padded taps read a valid cell; their weights are zero
            aoffs[kb] = abase + (ky * PX + kx) * CS + (t < 9 ? c : 0) * 2;
        }
    }

    // ---- epilogue state: D row (pixel) = (lane >> 4) * 4 + register, column (channel) = lane & 15.  Per lane and for good: the
    //      affine of its channel per N tile, the byte offset of its pixel quad from the tile origin per (N tile, pixel group) --
    //      OOB where the channel or the column does not exist, so loads return zeros and stores are dropped.
    float esc[NTN], esh[NTN];
    unsigned eoff[NTN][MGN];
#pragma unroll
    for (int nt = 0; nt < NTN; ++nt) {
        const int co = nt * 16 + (lane & 15);
        esc[nt] = p.scale && co < p.CO ? p.scale[co] : 1.0f;
        esh[nt] = p.shift && co < p.CO ? p.shift[co] : 0.0f;
#pragma unroll
        for (int mg = 0; mg < MGN; ++mg) {
            const int ox = x0 + mg * 16 + (lane >> 4) * 4;      // W % 4 == 0: a quad is inside or outside as a whole
            eoff[nt][mg] = co < p.CO && ox < W ? (unsigned)co * plane4 + (unsigned)(wave * W + mg * 16 + (lane >> 4) * 4) * 4 : OOBZ;
        }
    }

    GnAcc gacc;
    gn_zero(gacc);
    auto tile = [&](int ty, const unsigned char* buf) {
        f4 acc[AW];
#pragma unroll
        for (int i = 0; i < AW; ++i) acc[i] = (f4){0, 0, 0, 0};
        if constexpr (F32) {
            const float* wf = reinterpret_cast<const float*>(wlds);
#pragma unroll
            for (int kb = 0; kb < NKB; ++kb) {   // K block = channels 4 kb % CI .. + 3 of tap 4 kb / CI
                const int t = (4 * kb) / CI, c = (4 * kb) % CI;
                const int aoffk = ((t / 3) * PX + (t % 3)) * CS + c * 4;
                float b[NTN];
#pragma unroll
                for (int nt = 0; nt < NTN; ++nt) b[nt] = wf[(kb * NTN + nt) * 64 + lane];
#pragma unroll
                for (int mg = 0; mg < MGN; ++mg) {
                    const float a = *reinterpret_cast<const float*>(buf + abase + mg * 16 * CS + aoffk);
#pragma unroll
                    for (int nt = 0; nt < NTN; ++nt)
                        acc[mg * NTN + nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b[nt], acc[mg * NTN + nt], 0, 0, 0);
                }
            }
        } else if constexpr (X3) {
#pragma unroll
            for (int kb = 0; kb < NKB; ++kb) {
                const unsigned char* ap = buf + aoffs[kb];
                bf16x8 b[3][NTN];
#pragma unroll
                for (int s = 0; s < 3; ++s)
#pragma unroll
                    for (int nt = 0; nt < NTN; ++nt)
                        b[s][nt] = __builtin_bit_cast(bf16x8, wlds[((s * NKB + kb) * NTN + nt) * 64 + lane]);
#pragma unroll
                for (int mg = 0; mg < MGN; ++mg) {
                    bf16x8 a[3];
#pragma unroll
                    for (int s = 0; s < 3; ++s)
                        a[s] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u4*>(ap + mg * 16 * CS + s * CI * 2));
#pragma unroll
                    for (int nt = 0; nt < NTN; ++nt) {   // small terms first
                        f4 c = acc[mg * NTN + nt];
                        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[2], b[0][nt], c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[2][nt], c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1], b[1][nt], c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1], b[0][nt], c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[1][nt], c, 0, 0, 0);
                        acc[mg * NTN + nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[0][nt], c, 0, 0, 0);
                    }
                }
            }
        } else
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb) {
            const unsigned char* ap = buf + aoffs[kb];
            h16x8 b[NTN];
#pragma unroll
            for (int nt = 0; nt < NTN; ++nt) b[nt] = __builtin_bit_cast(h16x8, wlds[(kb * NTN + nt) * 64 + lane]);
#pragma unroll
            for (int mg = 0; mg < MGN; ++mg) {
                const h16x8 a = __builtin_bit_cast(h16x8, *reinterpret_cast<const u4*>(ap + mg * 16 * CS));
#pragma unroll
                for (int nt = 0; nt < NTN; ++nt)
                    acc[mg * NTN + nt] = mfma_h16(a, b[nt], acc[mg * NTN + nt]);
            }
        }
        // ---- epilogue ------------------------------------------------------------------------------------------------------
        const int oy = ty * TYZ + wave;
        if (oy >= H) return;
        if (p.sub2) {   // the stride-2 layer of a shape the stride-2 kernel has no room for: three quarters of the tile are dropped
            if (oy & 1) return;
#pragma unroll
            for (int nt = 0; nt < NTN; ++nt) {
                const int co = nt * 16 + (lane & 15);
                if (co >= p.CO) continue;
#pragma unroll
                for (int mg = 0; mg < MGN; ++mg) {
                    const int ox = x0 + mg * 16 + (lane >> 4) * 4;
                    if (ox >= W) continue;
                    const f4 y = acc[mg * NTN + nt] * esc[nt] + esh[nt];
                    const size_t o2 = ((size_t)co * ((H + 1) >> 1) + (oy >> 1)) * (W >> 1) + (ox >> 1);
                    float y0 = y[0], y2 = y[2];
                    if (p.skip && !p.skip_after_act) { y0 += p.skip[o2]; y2 += p.skip[o2 + 1]; }
                    if (p.act == 1) { y0 = fmaxf(y0, 0.0f); y2 = fmaxf(y2, 0.0f); }
                    if (p.skip && p.skip_after_act) { y0 = p.skip[o2] + y0; y2 = p.skip[o2 + 1] + y2; }
                    p.out[o2] = y0;
                    p.out[o2 + 1] = y2;
                }
            }
            return;
        }
        const long torg = (long)ty * TYZ * W + x0;   // tile origin (elements inside a channel plane)
        const __amdgpu_buffer_rsrc_t ro = z2_rsrc(p.out + torg);
        const __amdgpu_buffer_rsrc_t rk = z2_rsrc((p.skip ? p.skip : p.out) + torg), ra = z2_rsrc((p.aux1 ? p.aux1 : p.out) + torg);
        // the epilogue's operands first (h, u, the skip tensor): all in flight at once, not one round trip per pixel quad
        f4 ek[AW], ea[AW];
#pragma unroll
        for (int i = 0; i < AW; ++i) {
            const unsigned o = eoff[i % NTN][i / NTN];
            const bool gate_h = p.act == 2 && (i % NTN) * 16 + (lane & 15) < p.ep_split;
            ek[i] = (f4){0, 0, 0, 0};
            ea[i] = (f4){0, 0, 0, 0};
            if (p.act == 2 ? true : p.skip != nullptr)
                ek[i] = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(rk, p.act == 2 && !gate_h ? OOBZ : o, 0, 0));
            if (p.act == 3) ea[i] = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(ra, o, 0, 0));
        }
#pragma unroll
        for (int i = 0; i < AW; ++i) {
            const int nt = i % NTN, mg = i / NTN;
            f4 y = acc[i] * esc[nt] + esh[nt];
            if (p.act == 2) {
                const bool gate_h = nt * 16 + (lane & 15) < p.ep_split;
#pragma unroll
                for (int k = 0; k < 4; ++k) y[k] = gru_sigmoid_as<!(F32 || X3)>(y[k]);   // (fp32 mode: torch's own expression, common.h)
                if (gate_h) y *= ek[i];
            } else if (p.act == 3) {
                const f4 u = ea[i], hh = ek[i];
#pragma unroll
                for (int k = 0; k < 4; ++k) y[k] = u[k] * hh[k] + (1.0f - u[k]) * gru_tanh_as<!(F32 || X3)>(y[k]);
            } else {
                if (p.skip && !p.skip_after_act) y += ek[i];
                if (p.act == 1) y = __builtin_elementwise_max(y, (f4){0, 0, 0, 0});
                if (p.skip && p.skip_after_act) y = ek[i] + y;
                if constexpr (GN) { if (eoff[nt][mg] != OOBZ) gn_add(gacc, nt * 16 + (lane & 15) >= p.gn_split, y); }   // (what is stored, and only that)
            }
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4, y), ro, eoff[nt][mg], 0, 0);
        }
    };

    if constexpr (IMG3) {
        constexpr int IPX = TX + 4, IPY = PYZ + 2, IPL = IPY * IPX;            // image patch: origin (tile row - 2, x0 - 2)
        constexpr int NIL = (3 * IPL + NTZ - 1) / NTZ;
        float* ipatch = reinterpret_cast<float*>(smem + NBUF * PATCH + NSPL * NKB * NTN * 64 * 16);
        unsigned ivo[NIL];
        int ipy[NIL], islot[NIL];
        float iv[NIL];
#pragma unroll
        for (int i = 0; i < NIL; ++i) {
            const int e = tid + NTZ * i;
            const int c = e / IPL, r = e - c * IPL, iy = r / IPX, ix = r - iy * IPX;
            const int gx = x0 - 2 + ix;
            ivo[i] = e < 3 * IPL && gx >= 0 && gx < W ? (unsigned)c * plane4 + (unsigned)(iy * W + ix) * 4 : OOBZ;
            ipy[i] = iy;
            islot[i] = e < 3 * IPL ? e : -1;
        }
        auto issue_img = [&](int ty) {
            const int gyo = ty * TYZ - 2;
            const __amdgpu_buffer_rsrc_t ri = z2_rsrc(p.img + ((long)gyo * W + x0 - 2));
#pragma unroll
            for (int i = 0; i < NIL; ++i) {
                const unsigned vo = (unsigned)(ipy[i] + gyo) < (unsigned)H ? ivo[i] : OOBZ;
                iv[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(ri, vo, 0, 0));
            }
        };
        auto commit_img = [&]() {
#pragma unroll
            for (int i = 0; i < NIL; ++i)
                if (islot[i] >= 0) ipatch[islot[i]] = iv[i];
        };
        // items of the first layer: (64 patch positions, channel quad); the quad is the wave's (its weights are scalar operands)
        constexpr int NPOS = PYZ * PX, NITEM = 2 * ((NPOS + 63) / 64), NRND = (NITEM + TYZ - 1) / TYZ;
        typedef const float __attribute__((address_space(4))) cfloat;
        const int wu = __builtin_amdgcn_readfirstlane(wave), hq = wu & 1;   // (scalar: the quad's weights are scalar loads)
        float sc0[4], sh0[4];
#pragma unroll
        for (int o = 0; o < 4; ++o) {
            sc0[o] = p.scale0 ? p.scale0[4 * hq + o] : 1.0f;
            sh0[o] = p.shift0 ? p.shift0[4 * hq + o] : 0.0f;
        }
        int ppy[NRND], ppx[NRND];
#pragma unroll
        for (int r = 0; r < NRND; ++r) {
            const int pos = (r * (TYZ / 2) + (wu >> 1)) * 64 + lane;
            ppy[r] = pos < NPOS ? pos / PX : -1;
            ppx[r] = pos - (pos / PX) * PX;
        }
        auto produce = [&](int ty) {
            const int gy0 = ty * TYZ - 1;
            // the rounds of a wave side by side: a (c, k_y) row of the weights is loaded (scalar) once for all of them
            float acc[NRND][4];
            const float* __restrict__ pl[NRND];
#pragma unroll
            for (int r = 0; r < NRND; ++r) {
                pl[r] = ipatch + max(ppy[r], 0) * IPX + ppx[r];
#pragma unroll
                for (int o = 0; o < 4; ++o) acc[r][o] = 0.0f;
            }
            cfloat* wt = (cfloat*)p.w0 + 4 * hq;
#pragma unroll 1
            for (int c = 0; c < 3; ++c) {
                cfloat* w = wt + c * 72;
                asm volatile("" : "+s"(w));
#pragma unroll
                for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                    for (int dx = 0; dx < 3; ++dx)
#pragma unroll
                        for (int r = 0; r < NRND; ++r) {
                            const float v = pl[r][c * IPL + dy * IPX + dx];
#pragma unroll
                            for (int o = 0; o < 4; ++o) acc[r][o] = fmaf(v, w[dy * 24 + dx * 8 + o], acc[r][o]);
                        }
            }
#pragma unroll
            for (int r = 0; r < NRND; ++r) {
                const int py = max(ppy[r], 0), px = ppx[r];
                const bool inside = ppy[r] >= 0 && (unsigned)(gy0 + py) < (unsigned)H && (unsigned)(x0 - 1 + px) < (unsigned)W;
                float y[4];
#pragma unroll
                for (int o = 0; o < 4; ++o) {
                    float t = acc[r][o];
                    if (p.scale0) t *= sc0[o];
                    if (p.shift0) t += sh0[o];
                    if (p.act0 == 1) t = fmaxf(t, 0.0f);
                    y[o] = inside ? t : 0.0f;   // outside the image: the second layer's zero padding
                }
                if (ppy[r] >= 0) put4(smem + (py * PX + px) * CS, hq, y[0], y[1], y[2], y[3]);
            }
        };
        issue_img(t0);
        commit_img();
        __syncthreads();
        produce(t0);
        lds_barrier();
        for (int ty = t0; ty < t1; ++ty) {
            const bool more = ty + 1 < t1;
            if (more) issue_img(ty + 1);
            tile(ty, smem);
            lds_barrier();   // every wave has read the cells
            if (more) {
                commit_img();
                lds_barrier();
                produce(ty + 1);
                lds_barrier();
            }
        }
        return;
    }
    // ---- walk the tiles of this workgroup: the next patch's loads fly during the sweep -------------------------------
    issue(t0);
    commit(smem);
    __syncthreads();
    int cur = 0;
    for (int ty = t0; ty < t1; ++ty) {
        const bool more = ty + 1 < t1;
        if (more) issue(ty + 1);
        tile(ty, smem + cur * PATCH);
        if constexpr (NBUF == 2) {
            if (more) commit(smem + (cur ^ 1) * PATCH);
            lds_barrier();
            cur ^= 1;
        } else {
            lds_barrier();   // every wave has read the patch
            if (more) commit(smem);
            lds_barrier();
        }
    }
    if constexpr (GN) gn_flush(gacc, p.gn, p.gn_split < p.CO ? 2 : 1, reinterpret_cast<double*>(smem), tid, TYZ);
}

template <int CI, int NTN, int MGN, bool F32 = false, bool X3 = false, bool IMG3 = false, bool GN = false>
static int launch_z2(const Z2Params& p, hipStream_t stream) {
    constexpr int TX = 16 * MGN, PX = TX + 2;
    constexpr int NKB = F32 ? 9 * CI / 4 : (9 * CI + 31) / 32;
    constexpr int PATCH = PX * PYZ * z2_cell_bytes<F32, X3>(CI);
    constexpr int WBYTES = NKB * NTN * 64 * (F32 ? 4 : 16) * (X3 ? 3 : 1);
    constexpr int lds = (X3 || (F32 && CI > 32) ? 1 : 2) * PATCH + WBYTES + (IMG3 ? 3 * (PYZ + 2) * (TX + 4) * 4 : 0);
    static_assert(lds <= 160 * 1024, "tile does not fit the LDS");
    if ((reinterpret_cast<uintptr_t>(p.in) | reinterpret_cast<uintptr_t>(p.in2) | reinterpret_cast<uintptr_t>(p.skip) |
         reinterpret_cast<uintptr_t>(p.aux1) | reinterpret_cast<uintptr_t>(p.out)) & 15) {
        set_error("conv2d tile kernel: tensors must be 16-byte aligned (dwordx4 loads and stores)");
        return D3D_ERR_UNSUPPORTED;
    }
    if ((long)(CI > p.CO ? CI : p.CO) * p.H * p.W * 4 * (1) >= (1L << 31)) return D3D_ERR_UNSUPPORTED;   // (32-bit buffer offsets)
    auto kern = conv2d_zs_bf16_kernel<CI, NTN, MGN, F32, X3, IMG3, GN>;
    int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(kern), lds);
    if (rc != D3D_OK) return rc;
    Z2Params q = p;
    const int gx = ceil_div(p.W, TX), nty = ceil_div(p.H, TYZ), nb = p.nbatch > 1 ? p.nbatch : 1;
    const int tper = pick_tper(gx * nb, nty, lds, WBYTES, PATCH, GN && CI == 40 ? 4 : 2);
    q.tper = tper;
    const int gy = ceil_div(nty, tper);
    if (gy > 65535 || nb > 65535) return D3D_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(kern, dim3(gx, gy, nb), dim3(NTZ), lds, stream, q);
    D3D_LAUNCH_CHECK("conv2d_zs_bf16_kernel launch");
    return D3D_OK;
}

// ---- stride 2 (adamvs.py:411 ConvReLU(8, 16, 3, 2, 1) of the slice regulariser): out (y, x) reads in (2y + k_y - 1, 2x + k_x - 1) --
// 32 x 8 OUTPUT pixels per step; the staged 65 x 17 patch keeps the even and the odd columns of a row in separate runs, so
// the 16 pixels of an A operand (input columns 2m + k_x - 1) are 16 consecutive cells.  p.H, p.W: INPUT size.
// KS = 5 (split operands only): the 5 x 5 stride-2 layers of the feature trunks (module.py:669, 675, padding 2) on the same scheme --
// the patch is 2 T + 3 wide / high, tap column k_x reads the even run at m + k_x / 2 (k_x even) or the odd run at m + (k_x - 1) / 2.
template <int CI, int NTN, bool F32 = false, bool X3 = false, int KS = 3, int MG = 2>
__global__ __launch_bounds__(NTZ, 2) void conv2d_s2_zs_bf16_kernel(Z2Params p) {
    static_assert(!(F32 && X3) && (KS == 3 || (KS == 5 && X3)), "one operand format; 5 x 5 with split operands only");
    constexpr int TXO = 16 * MG, PXI = 2 * TXO + KS - 2, PYI = 2 * TYZ + KS - 2, NEVEN = (PXI + 1) / 2, ORG = KS / 2;
    constexpr int NKB = F32 ? KS * KS * CI / 4 : (KS * KS * CI + 31) / 32;
    constexpr int CS = z2_cell_bytes<F32, X3>(CI);
    constexpr int G = CI / 8;
    constexpr int PATCH = PXI * PYI * CS;
    constexpr int NBUF = X3 ? 1 : 2;   // (split operands: one patch buffer, as in the stride-1 kernel)
    constexpr int AW = MG * NTN;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    u4* wlds = reinterpret_cast<u4*>(smem + NBUF * PATCH);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (p.nbatch > 1) { p.in += (size_t)blockIdx.z * p.in_bstride; p.out += (size_t)blockIdx.z * p.out_bstride; }   // (batched launch: this workgroup's item)
    const int H = p.H, W = p.W, Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
    const int xo0 = blockIdx.x * TXO;
    const int nty = (Ho + TYZ - 1) / TYZ;
    const int t0 = blockIdx.y * p.tper, t1 = min(t0 + p.tper, nty);

    if constexpr (F32) {
        float* wf = reinterpret_cast<float*>(wlds);
        const float* wg = reinterpret_cast<const float*>(p.wpk);
        for (int i = tid; i < NKB * NTN * 64; i += NTZ) wf[i] = wg[i];
    } else {
        for (int i = tid; i < (X3 ? 3 : 1) * NKB * NTN * 64; i += NTZ) wlds[i] = p.wpk[i];   // X3: [hi | mid | lo][NKB][N tiles][lane]
    }

    // task = (patch pixel, 8-channel group): eight dword loads, one cell slot.  Per lane and for good (see the stride-1 kernel):
    // the task's byte offset from the patch origin (OOBZ where its column is outside the image), its patch row, its cell.
    constexpr int NTASK = PXI * PYI * G;
    constexpr int ROUNDS = (NTASK + NTZ - 1) / NTZ;
    const unsigned plane4 = (unsigned)H * W * 4, oplane4 = (unsigned)Ho * Wo * 4;   // (host: every tensor < 2^31 bytes)
    float stg[ROUNDS][8];
    unsigned svo[ROUNDS];
    int spy[ROUNDS], scell[ROUNDS];
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {
        const int task = tid + r * NTZ;
        const int pix = task / G, g = task - pix * G;
        const int py = pix / PXI, px = pix - py * PXI;
        const int gx = 2 * xo0 - ORG + px;
        svo[r] = task < NTASK && gx >= 0 && gx < W ? (unsigned)(8 * g) * plane4 + (unsigned)(py * W + px) * 4 : OOBZ;
        spy[r] = py;
        const int cell = py * PXI + ((px & 1) ? NEVEN + (px >> 1) : (px >> 1));
        scell[r] = task < NTASK ? cell * CS + g * (F32 ? 32 : 16) : -1;
    }
    auto issue = [&](int ty) {
        const int gy0 = 2 * ty * TYZ - ORG;
        const bool inner = gy0 >= 0 && gy0 + PYI <= H;
        const __amdgpu_buffer_rsrc_t rs = z2_rsrc(p.in + ((long)gy0 * W + 2 * xo0 - ORG));
#pragma unroll
        for (int r = 0; r < ROUNDS; ++r) {
            unsigned vo = svo[r];
            if (!inner) vo = (unsigned)(spy[r] + gy0) < (unsigned)H ? vo : OOBZ;
#pragma unroll
            for (int k = 0; k < 8; ++k) stg[r][k] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, vo, k * plane4, 0));
        }
    };
    auto commit = [&](unsigned char* dst) {
#pragma unroll
        for (int r = 0; r < ROUNDS; ++r) {
            if (scell[r] >= 0) {
                unsigned char* cell = dst + scell[r];
                if constexpr (F32) {
                    *reinterpret_cast<f4*>(cell) = (f4){stg[r][0], stg[r][1], stg[r][2], stg[r][3]};
                    *reinterpret_cast<f4*>(cell + 16) = (f4){stg[r][4], stg[r][5], stg[r][6], stg[r][7]};
                } else if constexpr (X3) {
                    put8_split3(cell, CI, stg[r]);
                } else {
                    const u4 v = {pack_h16_z2(stg[r][0], stg[r][1]), pack_h16_z2(stg[r][2], stg[r][3]),
                                  pack_h16_z2(stg[r][4], stg[r][5]), pack_h16_z2(stg[r][6], stg[r][7])};
                    *reinterpret_cast<u4*>(cell) = v;
                }
            }
        }
    };
    auto a_offset = [&](int kb, int kgroup) {
        const int k0 = 32 * kb + 8 * kgroup;
        const int t = k0 / CI, c = k0 % CI;
        const int ky = t < KS * KS ? t / KS : 0, kx = t < KS * KS ? t % KS : 0;
        const int col = (kx & 1) ? NEVEN + (kx >> 1) : (kx >> 1);   // even run index m + k_x / 2, odd run index m + (k_x - 1) / 2
        return (ky * PXI + col) * CS + (t < KS * KS ? c : 0) * 2;
    };
    const int abase = (2 * wave * PXI + (lane & 15)) * CS + (F32 ? (lane >> 4) * 4 : 0);
    int aoffs[F32 ? 1 : NKB];   // one register per K block, computed once
    if constexpr (!F32) {
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb) aoffs[kb] = abase + a_offset(kb, lane >> 4);
    }
    // epilogue state (see the stride-1 kernel): affine per N tile, byte offset of the lane's pixel quad from the tile origin
    float esc[NTN], esh[NTN];
    unsigned eoff[NTN][MG];
#pragma unroll
    for (int nt = 0; nt < NTN; ++nt) {
        const int co = nt * 16 + (lane & 15);
        esc[nt] = p.scale && co < p.CO ? p.scale[co] : 1.0f;
        esh[nt] = p.shift && co < p.CO ? p.shift[co] : 0.0f;
#pragma unroll
        for (int mg = 0; mg < MG; ++mg) {
            const int ox = xo0 + mg * 16 + (lane >> 4) * 4;     // Wo % 4 == 0: a quad is inside or outside as a whole
            eoff[nt][mg] = co < p.CO && ox < Wo ? (unsigned)co * oplane4 + (unsigned)(wave * Wo + mg * 16 + (lane >> 4) * 4) * 4 : OOBZ;
        }
    }

    auto tile = [&](int ty, const unsigned char* buf) {
        f4 acc[AW];
#pragma unroll
        for (int i = 0; i < AW; ++i) acc[i] = (f4){0, 0, 0, 0};
        if constexpr (F32) {
            const float* wf = reinterpret_cast<const float*>(wlds);
#pragma unroll
            for (int kb = 0; kb < NKB; ++kb) {
                const int t = (4 * kb) / CI, c = (4 * kb) % CI;
                const int ky = t / 3, kx = t % 3;
                const int aoffk = (ky * PXI + (kx == 1 ? NEVEN : (kx >> 1))) * CS + c * 4;
                float b[NTN];
#pragma unroll
                for (int nt = 0; nt < NTN; ++nt) b[nt] = wf[(kb * NTN + nt) * 64 + lane];
#pragma unroll
                for (int mg = 0; mg < MG; ++mg) {
                    const float a = *reinterpret_cast<const float*>(buf + abase + mg * 16 * CS + aoffk);
#pragma unroll
                    for (int nt = 0; nt < NTN; ++nt)
                        acc[mg * NTN + nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b[nt], acc[mg * NTN + nt], 0, 0, 0);
                }
            }
        } else if constexpr (X3) {
#pragma unroll
            for (int kb = 0; kb < NKB; ++kb) {
                const unsigned char* ap = buf + aoffs[F32 ? 0 : kb];
                bf16x8 b[3][NTN];
#pragma unroll
                for (int sp = 0; sp < 3; ++sp)
#pragma unroll
                    for (int nt = 0; nt < NTN; ++nt)
                        b[sp][nt] = __builtin_bit_cast(bf16x8, wlds[((sp * NKB + kb) * NTN + nt) * 64 + lane]);
#pragma unroll
                for (int mg = 0; mg < MG; ++mg) {
                    bf16x8 a[3];
#pragma unroll
                    for (int sp = 0; sp < 3; ++sp)
                        a[sp] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u4*>(ap + mg * 16 * CS + sp * CI * 2));
#pragma unroll
                    for (int nt = 0; nt < NTN; ++nt) acc[mg * NTN + nt] = mfma_split3(a, b[0][nt], b[1][nt], b[2][nt], acc[mg * NTN + nt]);
                }
            }
        } else
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb) {
            const unsigned char* ap = buf + aoffs[F32 ? 0 : kb];
            h16x8 b[NTN];
#pragma unroll
            for (int nt = 0; nt < NTN; ++nt) b[nt] = __builtin_bit_cast(h16x8, wlds[(kb * NTN + nt) * 64 + lane]);
#pragma unroll
            for (int mg = 0; mg < MG; ++mg) {
                const h16x8 a = __builtin_bit_cast(h16x8, *reinterpret_cast<const u4*>(ap + mg * 16 * CS));
#pragma unroll
                for (int nt = 0; nt < NTN; ++nt)
                    acc[mg * NTN + nt] = mfma_h16(a, b[nt], acc[mg * NTN + nt]);
            }
        }
        const int oy = ty * TYZ + wave;
        if (oy >= Ho) return;
        const long torg = (long)ty * TYZ * Wo + xo0;   // tile origin inside an output channel plane
        const __amdgpu_buffer_rsrc_t ro = z2_rsrc(p.out + torg), rk = z2_rsrc((p.skip ? p.skip : p.out) + torg);
        f4 ek[AW];
#pragma unroll
        for (int i = 0; i < AW; ++i) {
            ek[i] = (f4){0, 0, 0, 0};
            if (p.skip) ek[i] = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(rk, eoff[i % NTN][i / NTN], 0, 0));
        }
#pragma unroll
        for (int i = 0; i < AW; ++i) {
            const int nt = i % NTN, mg = i / NTN;
            f4 y = acc[i] * esc[nt] + esh[nt];
            if (p.skip && !p.skip_after_act) y += ek[i];
            if (p.act == 1) y = __builtin_elementwise_max(y, (f4){0, 0, 0, 0});
            if (p.skip && p.skip_after_act) y = ek[i] + y;
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4, y), ro, eoff[nt][mg], 0, 0);
        }
    };

    issue(t0);
    commit(smem);
    __syncthreads();
    int cur = 0;
    for (int ty = t0; ty < t1; ++ty) {
        const bool more = ty + 1 < t1;
        if (more) issue(ty + 1);
        tile(ty, smem + cur * PATCH);
        if constexpr (NBUF == 2) {
            if (more) commit(smem + (cur ^ 1) * PATCH);
            lds_barrier();
            cur ^= 1;
        } else {
            lds_barrier();
            if (more) commit(smem);
            lds_barrier();
        }
    }
}

// ---- transposed, k = 3, stride 2, pad 1, output_pad 1 (adamvs.py:413-417 upconv1 16 -> 8 with the skip before the ReLU, upconv2d
// 8 -> 1): four per-parity dense convolutions over one staged 33 x 9 input patch (32 x 8 input pixels = 64 x 16 outputs per
// step); the two column parities of an output row are interleaved in registers, a lane stores 8 consecutive pixels.
// K4: ConvTranspose2d(k 4, stride 2, pad 1) -> [Co,2H,2W] on the same scheme: every parity class has 2 x 2 taps, output
// (2i + p) reads inputs i - 1 + p + d, d = 0 | 1 (kernel index 3 - p - 2 d), so the patch carries a halo on both sides (34 x 10).
// conv3x3(nearest_x2(f)) is such a layer with summed weights (ops.upsampled_conv_weight): the FPN output level.
constexpr int ntaps2(int py, int px, bool k4 = false) { return k4 ? 4 : (1 + py) * (1 + px); }
constexpr int nkb2(int CI, int py, int px, bool f32 = false, bool k4 = false) {
    return f32 ? ntaps2(py, px, k4) * CI / 4 : (ntaps2(py, px, k4) * CI + 31) / 32;
}
constexpr int frag_base2(int CI, int c, bool f32 = false, bool k4 = false) {
    int s = 0;
    for (int q = 0; q < c; ++q) s += nkb2(CI, q >> 1, q & 1, f32, k4);
    return s;
}

// XF (K4, split operands, C_out <= 8): both column parities of an output row as ONE 16-column tile -- columns 0..7 the channels
// of the even output column, 8..15 those of the odd one, over the three patch columns m, m + 1, m + 2 the two parities touch
// (weights zero where a parity does not use a column): K = 2 x 3 x C_in per row parity, 36 instead of 48 MFMAs, 36 instead of
// 48 KB of weights at C_in = 32 -- with 16-wide tiles the kernel then fits two workgroups per CU.  The odd half reaches the
// storing lanes with one v_mov_dpp row_ror:8 per register.
__device__ __forceinline__ float dpp_row_ror8(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xf, 0xf, true));
}

template <int CI, bool F32 = false, bool X3 = false, bool K4 = false, bool XF = false>
__global__ __launch_bounds__(NTZ, 2) void convt2d_zs_bf16_kernel(Z2Params p) {
    static_assert(!(F32 && X3) && (!K4 || !F32) && (!XF || (K4 && X3)), "one operand format; the k = 4 form has bf16 / split operands");
    constexpr int MG = XF ? 1 : 2, TXI = 16 * MG, PXI = TXI + (K4 ? 2 : 1), PYI = TYZ + (K4 ? 2 : 1), ORG = K4 ? 1 : 0;
    constexpr int CS = z2_cell_bytes<F32, X3>(CI);
    constexpr int G = CI / 8;
    constexpr int PATCH = PXI * PYI * CS;
    constexpr int NKBF = (6 * CI + 31) / 32;   // XF: K blocks per row parity
    constexpr int NFRAG = XF ? 2 * NKBF : frag_base2(CI, 4, F32, K4);
    constexpr int NBUF = X3 && CI > 8 ? 1 : 2;   // (split cells of 16 | 32 channels: one patch buffer, more workgroups per CU)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    u4* wlds = reinterpret_cast<u4*>(smem + NBUF * PATCH);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int H = p.H, W = p.W, OW = 2 * W;
    const int ix0 = blockIdx.x * TXI;
    const int nty = (H + TYZ - 1) / TYZ;
    const int t0 = blockIdx.y * p.tper, t1 = min(t0 + p.tper, nty);

    if constexpr (F32) {
        float* wf = reinterpret_cast<float*>(wlds);
        const float* wg = reinterpret_cast<const float*>(p.wpk);
        for (int i = tid; i < NFRAG * 64; i += NTZ) wf[i] = wg[i];
    } else {
        for (int i = tid; i < (X3 ? 3 : 1) * NFRAG * 64; i += NTZ) wlds[i] = p.wpk[i];   // X3: [hi | mid | lo][fragments][lane]
    }

    // task = (patch pixel, 8-channel group): eight dword loads, one cell slot; offsets, rows and cells per lane and for good,
    // OOBZ offsets for what lies outside the image (see the stride-1 kernel)
    constexpr int NTASK = PXI * PYI * G;
    constexpr int ROUNDS = (NTASK + NTZ - 1) / NTZ;
    const unsigned plane4 = (unsigned)H * W * 4, oplane4 = 4 * plane4;   // (host: every tensor < 2^31 bytes)
    float stg[ROUNDS][8];
    unsigned svo[ROUNDS];
    int spy[ROUNDS], scell[ROUNDS];
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {
        const int task = tid + r * NTZ;
        const int pix = task / G, g = task - pix * G;
        const int py = pix / PXI, px = pix - py * PXI;
        const int gx = ix0 + px - ORG;
        svo[r] = task < NTASK && gx >= 0 && gx < W ? (unsigned)(8 * g) * plane4 + (unsigned)(py * W + px) * 4 : OOBZ;
        spy[r] = py;
        scell[r] = task < NTASK ? pix * CS + g * (F32 ? 32 : 16) : -1;
    }
    auto issue = [&](int ty) {
        const int gy0 = ty * TYZ - ORG;
        const bool inner = gy0 >= 0 && gy0 + PYI <= H;
        const __amdgpu_buffer_rsrc_t rs = z2_rsrc(p.in + ((long)gy0 * W + ix0 - ORG));
#pragma unroll
        for (int r = 0; r < ROUNDS; ++r) {
            unsigned vo = svo[r];
            if (!inner) vo = (unsigned)(spy[r] + gy0) < (unsigned)H ? vo : OOBZ;
#pragma unroll
            for (int k = 0; k < 8; ++k) stg[r][k] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, vo, k * plane4, 0));
        }
    };
    auto commit = [&](unsigned char* dst) {
#pragma unroll
        for (int r = 0; r < ROUNDS; ++r) {
            if (scell[r] >= 0) {
                unsigned char* cell = dst + scell[r];
                if constexpr (F32) {
                    *reinterpret_cast<f4*>(cell) = (f4){stg[r][0], stg[r][1], stg[r][2], stg[r][3]};
                    *reinterpret_cast<f4*>(cell + 16) = (f4){stg[r][4], stg[r][5], stg[r][6], stg[r][7]};
                } else if constexpr (X3) {
                    put8_split3(cell, CI, stg[r]);
                } else {
                    const u4 v = {pack_h16_z2(stg[r][0], stg[r][1]), pack_h16_z2(stg[r][2], stg[r][3]),
                                  pack_h16_z2(stg[r][4], stg[r][5]), pack_h16_z2(stg[r][6], stg[r][7])};
                    *reinterpret_cast<u4*>(cell) = v;
                }
            }
        }
    };
    const int abase = (wave * PXI + (lane & 15)) * CS + (F32 ? (lane >> 4) * 4 : 0);
    // epilogue state: channel lane & 15 (XF: < 8), four input pixels = eight outputs from column 2 ix on; byte offset of output
    // row 2 (ty TYZ + wave) from the tile origin, OOBZ where the channel or the columns do not exist
    const int eco = lane & 15;
    const bool ecin = eco < p.CO && (!XF || eco < 8);
    const float esc = p.scale && ecin ? p.scale[eco] : 1.0f, esh = p.shift && ecin ? p.shift[eco] : 0.0f;
    unsigned eoff[MG];
#pragma unroll
    for (int mg = 0; mg < MG; ++mg) {
        const int ix = ix0 + mg * 16 + (lane >> 4) * 4;             // W % 4 == 0: inside or outside as a whole
        eoff[mg] = ecin && ix < W ? (unsigned)eco * oplane4 + (unsigned)(2 * wave * OW + 2 * (mg * 16 + (lane >> 4) * 4)) * 4 : OOBZ;
    }
    // both halves of the lane's eight outputs: skip operand in, result out
    auto finish = [&](int ty, int PY, int mg, f4 e, f4 od) {
        const long torg = ((long)(2 * ty * TYZ + PY)) * OW + 2 * ix0;   // tile origin inside an output channel plane
        const __amdgpu_buffer_rsrc_t ro = z2_rsrc(p.out + torg), rk = z2_rsrc((p.skip ? p.skip : p.out) + torg);
        const unsigned o = ty * TYZ + wave < H ? eoff[mg] : OOBZ;
        f4 lo = {e[0], od[0], e[1], od[1]}, hi = {e[2], od[2], e[3], od[3]};
        if (p.skip) {
            const f4 k0 = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(rk, o, 0, 0));
            const f4 k1 = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(rk, o, 16, 0));
            if (!p.skip_after_act) { lo += k0; hi += k1; }
            if (p.act == 1) { lo = __builtin_elementwise_max(lo, (f4){0, 0, 0, 0}); hi = __builtin_elementwise_max(hi, (f4){0, 0, 0, 0}); }
            if (p.skip_after_act) { lo = k0 + lo; hi = k1 + hi; }
        } else if (p.act == 1) {
            lo = __builtin_elementwise_max(lo, (f4){0, 0, 0, 0}); hi = __builtin_elementwise_max(hi, (f4){0, 0, 0, 0});
        }
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4, lo), ro, o, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4, hi), ro, o, 16, 0);
    };

    auto row = [&](auto pyc, int ty, const unsigned char* buf) {   // output row 2 iy + PY of this wave's input row: both column parities
        constexpr int PY = decltype(pyc)::value;
        if constexpr (XF) {
            f4 accf = {0, 0, 0, 0};
            const int kg = lane >> 4;   // (not opaque: the K offsets are tile-invariant, the compiler keeps or folds them)
#pragma unroll
            for (int kb = 0; kb < NKBF; ++kb) {
                const int k0 = 32 * kb + 8 * kg;
                const int t = k0 / CI, c = k0 % CI;
                const bool real = t < 6;
                const int dxx = real ? t % 3 : 0, dy = real ? t / 3 : 0;
                const int aoff = ((dy + PY) * PXI + dxx) * CS + (real ? c : 0) * 2;
                bf16x8 bw[3], a[3];
#pragma unroll
                for (int sp = 0; sp < 3; ++sp) {
                    bw[sp] = __builtin_bit_cast(bf16x8, wlds[(sp * NFRAG + PY * NKBF + kb) * 64 + lane]);
                    a[sp] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u4*>(buf + abase + aoff + sp * CI * 2));
                }
                accf = mfma_split3(a, bw[0], bw[1], bw[2], accf);
            }
            f4 oddf;
#pragma unroll
            for (int r = 0; r < 4; ++r) oddf[r] = dpp_row_ror8(accf[r]);   // columns 8..15 -> the lanes of columns 0..7
            finish(ty, PY, 0, accf * esc + esh, oddf * esc + esh);
            return;
        }
        f4 acc[2][MG];
#pragma unroll
        for (int px = 0; px < 2; ++px)
#pragma unroll
            for (int mg = 0; mg < MG; ++mg) acc[px][mg] = (f4){0, 0, 0, 0};
        const int kgroup = lane >> 4;
#pragma unroll
        for (int px = 0; px < 2; ++px) {
            const int NKB = nkb2(CI, PY, px, F32, K4);
            const int FB = frag_base2(CI, PY * 2 + px, F32, K4);
            if constexpr (F32) {
                const float* wf = reinterpret_cast<const float*>(wlds);
#pragma unroll
                for (int kb = 0; kb < NKB; ++kb) {
                    const int t = (4 * kb) / CI, c = (4 * kb) % CI;
                    const int dx = t % (1 + px), dy = t / (1 + px);
                    const int aoff = (dy * PXI + dx) * CS + c * 4;
                    const float bw = wf[(FB + kb) * 64 + lane];
#pragma unroll
                    for (int mg = 0; mg < MG; ++mg) {
                        const float a = *reinterpret_cast<const float*>(buf + abase + mg * 16 * CS + aoff);
                        acc[px][mg] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bw, acc[px][mg], 0, 0, 0);
                    }
                }
                continue;
            }
#pragma unroll
            for (int kb = 0; kb < NKB; ++kb) {
                const int k0 = 32 * kb + 8 * kgroup;
                const int t = k0 / CI, c = k0 % CI;
                const bool real = t < ntaps2(PY, px, K4);
                const int dx = !real ? 0 : K4 ? (t & 1) + px : t % (1 + px), dy = !real ? 0 : K4 ? (t >> 1) + PY : t / (1 + px);
                const int aoff = (dy * PXI + dx) * CS + (real ? c : 0) * 2;
                if constexpr (X3) {
                    bf16x8 bw[3];
#pragma unroll
                    for (int sp = 0; sp < 3; ++sp) bw[sp] = __builtin_bit_cast(bf16x8, wlds[(sp * NFRAG + FB + kb) * 64 + lane]);
#pragma unroll
                    for (int mg = 0; mg < MG; ++mg) {
                        bf16x8 a[3];
#pragma unroll
                        for (int sp = 0; sp < 3; ++sp)
                            a[sp] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u4*>(buf + abase + mg * 16 * CS + aoff + sp * CI * 2));
                        acc[px][mg] = mfma_split3(a, bw[0], bw[1], bw[2], acc[px][mg]);
                    }
                    continue;
                }
                const h16x8 bw = __builtin_bit_cast(h16x8, wlds[(FB + kb) * 64 + lane]);
#pragma unroll
                for (int mg = 0; mg < MG; ++mg) {
                    const h16x8 a = __builtin_bit_cast(h16x8, *reinterpret_cast<const u4*>(buf + abase + mg * 16 * CS + aoff));
                    acc[px][mg] = mfma_h16(a, bw, acc[px][mg]);
                }
            }
        }
#pragma unroll
        for (int mg = 0; mg < MG; ++mg) finish(ty, PY, mg, acc[0][mg] * esc + esh, acc[1][mg] * esc + esh);
    };

    issue(t0);
    commit(smem);
    __syncthreads();
    int cur = 0;
    for (int ty = t0; ty < t1; ++ty) {
        const bool more = ty + 1 < t1;
        if (more) issue(ty + 1);
        row(std::integral_constant<int, 0>{}, ty, smem + cur * PATCH);
        row(std::integral_constant<int, 1>{}, ty, smem + cur * PATCH);
        if constexpr (NBUF == 2) {
            if (more) commit(smem + (cur ^ 1) * PATCH);
            lds_barrier();
            cur ^= 1;
        } else {
            lds_barrier();
            if (more) commit(smem);
            lds_barrier();
        }
    }
}

template <int CI, int NTN, bool F32 = false, bool X3 = false, int KS = 3, int MG = 2>
static int launch_s2z(const Z2Params& p, hipStream_t stream) {
    constexpr int NKB = F32 ? KS * KS * CI / 4 : (KS * KS * CI + 31) / 32;
    constexpr int CS = z2_cell_bytes<F32, X3>(CI);
    constexpr int lds = (X3 ? 1 : 2) * (32 * MG + KS - 2) * (2 * TYZ + KS - 2) * CS + NKB * NTN * 64 * (F32 ? 4 : 16) * (X3 ? 3 : 1);
    static_assert(lds <= 160 * 1024, "tile does not fit the LDS");
    if ((long)(CI > p.CO ? CI : p.CO) * p.H * p.W * 4 * (1) >= (1L << 31)) return D3D_ERR_UNSUPPORTED;   // (32-bit buffer offsets)
    auto kern = conv2d_s2_zs_bf16_kernel<CI, NTN, F32, X3, KS, MG>;
    int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(kern), lds);
    if (rc != D3D_OK) return rc;
    Z2Params q = p;
    const int Ho = (p.H - 1) / 2 + 1, Wo = (p.W - 1) / 2 + 1;
    const int gx = ceil_div(Wo, 16 * MG), nty = ceil_div(Ho, TYZ), nb = p.nbatch > 1 ? p.nbatch : 1;
    constexpr int PATCH_S2 = (32 * MG + KS - 2) * (2 * TYZ + KS - 2) * CS;
    const int tper = pick_tper(gx * nb, nty, lds, lds - (X3 ? 1 : 2) * PATCH_S2, PATCH_S2);
    q.tper = tper;
    const int gy = ceil_div(nty, tper);
    if (gy > 65535 || nb > 65535) return D3D_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(kern, dim3(gx, gy, nb), dim3(NTZ), lds, stream, q);
    D3D_LAUNCH_CHECK("conv2d_s2_zs_bf16_kernel launch");
    return D3D_OK;
}

template <int CI, bool F32 = false, bool X3 = false, bool K4 = false, bool XF = false>
static int launch_tz(const Z2Params& p, hipStream_t stream) {
    constexpr int CS = z2_cell_bytes<F32, X3>(CI);
    constexpr int TXI = XF ? 16 : 32;
    constexpr int NFRAG = XF ? 2 * ((6 * CI + 31) / 32) : frag_base2(CI, 4, F32, K4);
    constexpr int lds = (X3 && CI > 8 ? 1 : 2) * (K4 ? (TXI + 2) * 10 : 33 * 9) * CS + NFRAG * 64 * (F32 ? 4 : 16) * (X3 ? 3 : 1);
    static_assert(lds <= 160 * 1024, "tile does not fit the LDS");
    if ((long)(CI > p.CO ? CI : p.CO) * p.H * p.W * 4 * (4) >= (1L << 31)) return D3D_ERR_UNSUPPORTED;   // (32-bit buffer offsets)
    auto kern = convt2d_zs_bf16_kernel<CI, F32, X3, K4, XF>;
    int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(kern), lds);
    if (rc != D3D_OK) return rc;
    Z2Params q = p;
    const int gx = ceil_div(p.W, TXI), nty = ceil_div(p.H, TYZ);
    constexpr int PATCH_T = (K4 ? (TXI + 2) * 10 : 33 * 9) * CS;
    const int tper = pick_tper(gx, nty, lds, lds - (X3 && CI > 8 ? 1 : 2) * PATCH_T, PATCH_T);
    q.tper = tper;
    const int gy = ceil_div(nty, tper);
    if (gy > 65535) return D3D_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(kern, dim3(gx, gy), dim3(NTZ), lds, stream, q);
    D3D_LAUNCH_CHECK("convt2d_zs_bf16_kernel launch");
    return D3D_OK;
}

}  // namespace

}  // namespace d3d

using namespace d3d;

extern "C" int d3d_conv2d_k3_zs_f32(const float* in, int C1, const float* in2, int C2, const void* wpacked, const float* scale,
                                    const float* shift, const float* skip, const float* aux1, int act, int ep_split,
                                    int skip_after_act, int Co, int H, int W, float* out, d3d_stream_t stream) {
    D3D_REQUIRE(in && wpacked && out, "null pointer");
    D3D_REQUIRE(H > 0 && W > 0 && C1 > 0 && C2 >= 0 && Co > 0, "bad dims");
    D3D_REQUIRE(act >= 0 && act <= 3, "bad act %d", act);
    D3D_REQUIRE(C2 == 0 || in2, "second input missing");
    D3D_REQUIRE(act < 2 || skip, "GRU epilogue (act %d) needs the state h in `skip`", act);
    D3D_REQUIRE(act != 2 || (ep_split > 0 && ep_split <= Co), "GRU gate epilogue: bad ep_split %d", ep_split);
    D3D_REQUIRE(act != 3 || aux1, "GRU update epilogue needs the update gate u in `aux1`");
    const int Ci = C1 + C2;
    const bool shape = (((Ci == 8 || Ci == 16 || Ci == 32) && Co <= 32) || (Ci == 48 && Co <= 48)) && C1 % 8 == 0 && C2 % 8 == 0 &&
                       W % 4 == 0;
    if (!shape) {
        set_error("d3d_conv2d_k3_zs_f32: C_in = %d + %d (8 | 16 | 32 in groups of 8 with C_out <= 32, or 48 with C_out <= 48), "
                  "C_out = %d, W = %d (multiple of 4) not taken", C1, C2, Co, W);
        return D3D_ERR_UNSUPPORTED;
    }
    Z2Params p = {};
    p.in = in; p.in2 = in2; p.C1 = C1; p.wpk = reinterpret_cast<const u4*>(wpacked); p.scale = scale; p.shift = shift;
    p.skip = skip; p.aux1 = aux1; p.out = out; p.H = H; p.W = W; p.CO = Co; p.act = act; p.ep_split = ep_split;
    p.skip_after_act = skip_after_act;
    hipStream_t st = (hipStream_t)stream;
    if (Ci == 48) {   // the pair-visibility UNet (adamvs.py:198-238) in fp32 mode: one patch buffer, the K loop is the bound
        if (Co > 32) return launch_z2<48, 3, 2, true>(p, st);
        if (Co > 16) return launch_z2<48, 2, 2, true>(p, st);
        return launch_z2<48, 1, 2, true>(p, st);
    }
    if (Co > 16) {
        if (Ci == 32) return launch_z2<32, 2, 2, true>(p, st);
        if (Ci == 16) return launch_z2<16, 2, 2, true>(p, st);
        return launch_z2<8, 2, 4, true>(p, st);
    }
    if (Ci == 32) return launch_z2<32, 1, 2, true>(p, st);
    if (Ci == 16) return launch_z2<16, 1, 2, true>(p, st);   // (64-wide tiles, one workgroup per CU: 14.7 -> 16.8 ms per AdaMVS view)
    return launch_z2<8, 1, 4, true>(p, st);
}

extern "C" int d3d_conv2d_k3_zs_bf16x3(const float* in, int C1, const float* in2, int C2, const void* wpacked, const float* scale,
                                       const float* shift, const float* skip, const float* aux1, int act, int ep_split,
                                       int skip_after_act, int Co, int H, int W, float* out, d3d_stream_t stream) {
    D3D_REQUIRE(in && wpacked && out, "null pointer");
    D3D_REQUIRE(H > 0 && W > 0 && C1 > 0 && C2 >= 0 && Co > 0, "bad dims");
    D3D_REQUIRE(act >= 0 && act <= 3, "bad act %d", act);
    D3D_REQUIRE(C2 == 0 || in2, "second input missing");
    D3D_REQUIRE(act < 2 || skip, "GRU epilogue (act %d) needs the state h in `skip`", act);
    D3D_REQUIRE(act != 2 || (ep_split > 0 && ep_split <= Co), "GRU gate epilogue: bad ep_split %d", ep_split);
    D3D_REQUIRE(act != 3 || aux1, "GRU update epilogue needs the update gate u in `aux1`");
    const int Ci = C1 + C2;
    const bool shape = (Ci == 8 || Ci == 16 || Ci == 32) && C1 % 8 == 0 && C2 % 8 == 0 && Co <= 32 && W % 4 == 0;
    if (!shape) {
        set_error("d3d_conv2d_k3_zs_bf16x3: C_in = %d + %d (8 | 16 | 32 in groups of 8), C_out = %d (<= 32), W = %d (multiple of 4) not taken",
                  C1, C2, Co, W);
        return D3D_ERR_UNSUPPORTED;
    }
    Z2Params p = {};
    p.in = in; p.in2 = in2; p.C1 = C1; p.wpk = reinterpret_cast<const u4*>(wpacked); p.scale = scale; p.shift = shift;
    p.skip = skip; p.aux1 = aux1; p.out = out; p.H = H; p.W = W; p.CO = Co; p.act = act; p.ep_split = ep_split;
    p.skip_after_act = skip_after_act;
    hipStream_t st = (hipStream_t)stream;
    if (Co > 16) {
        if (Ci == 32) return launch_z2<32, 2, 2, false, true>(p, st);
        if (Ci == 16) return launch_z2<16, 2, 2, false, true>(p, st);
        return launch_z2<8, 2, 4, false, true>(p, st);
    }
    if (Ci == 32) return launch_z2<32, 1, 1, false, true>(p, st);   // 16-wide tiles: 65 KB, two workgroups per CU
    if (Ci == 16) return launch_z2<16, 1, 2, false, true>(p, st);
    return launch_z2<8, 1, 4, false, true>(p, st);
}

// conv0 of a feature trunk (3 -> 8 -> CO <= 16, both 3 x 3 stride 1 with affine + ReLU | none) in one launch: see IMG3 above.
extern "C" int d3d_conv2d_k3_pair3_bf16x3(const float* img, const float* w0packed, const float* scale0, const float* shift0, int act0,
                                          const void* wpacked, const float* scale, const float* shift, int act, int Co, int H, int W,
                                          float* out, d3d_stream_t stream) {
    D3D_REQUIRE(img && w0packed && wpacked && out, "null pointer");
    D3D_REQUIRE(H > 0 && W > 0 && Co > 0, "bad dims");
    D3D_REQUIRE((act == 0 || act == 1) && (act0 == 0 || act0 == 1), "bad act %d / %d", act0, act);
    if (Co > 16 || W % 4 != 0 || (reinterpret_cast<uintptr_t>(img) & 3)) {
        set_error("d3d_conv2d_k3_pair3_bf16x3: C_out = %d (<= 16), W = %d (multiple of 4) not taken", Co, W);
        return D3D_ERR_UNSUPPORTED;
    }
    Z2Params p = {};
    p.in = out;   // (not read; the launcher checks its alignment)
    p.C1 = 8; p.wpk = reinterpret_cast<const u4*>(wpacked); p.scale = scale; p.shift = shift;
    p.out = out; p.H = H; p.W = W; p.CO = Co; p.act = act;
    p.img = img; p.w0 = w0packed; p.scale0 = scale0; p.shift0 = shift0; p.act0 = act0;
    return launch_z2<8, 1, 4, false, true, true>(p, (hipStream_t)stream);
}

enum { PREC_BF16 = 0, PREC_F32 = 1, PREC_X3 = 2 };   // operand format of the stride-2 / transposed tile kernels

static int conv2d_k3s2_zs(int prec, const float* in, const void* wpacked, const float* scale, const float* shift, const float* skip,
                          int act, int skip_after_act, int Ci, int Co, int H, int W, float* out, d3d_stream_t stream, int nbatch = 1,
                          long in_bstride = 0, long out_bstride = 0);

extern "C" int d3d_conv2d_k5s2_zs_bf16x3(const float* in, const void* wpacked, const float* scale, const float* shift,
                                         const float* skip, int act, int skip_after_act, int Ci, int Co, int H, int W, float* out,
                                         d3d_stream_t stream) {
    D3D_REQUIRE(in && wpacked && out, "null pointer");
    D3D_REQUIRE(H > 0 && W > 0 && Ci > 0 && Co > 0, "bad dims");
    D3D_REQUIRE(act == 0 || act == 1, "bad act %d", act);
    const int Wo = (W - 1) / 2 + 1;
    if (!((Ci == 8 && Co <= 16) || (Ci == 16 && Co <= 32)) || Wo % 4 != 0) {
        set_error("d3d_conv2d_k5s2_zs_bf16x3: C_in = %d, C_out = %d (8 -> <= 16 | 16 -> <= 32), output width %d (multiple of 4) not taken", Ci, Co, Wo);
        return D3D_ERR_UNSUPPORTED;
    }
    Z2Params p = {};
    p.in = in; p.C1 = Ci; p.wpk = reinterpret_cast<const u4*>(wpacked); p.scale = scale; p.shift = shift; p.skip = skip; p.out = out;
    p.H = H; p.W = W; p.CO = Co; p.act = act; p.skip_after_act = skip_after_act;
    hipStream_t st = (hipStream_t)stream;
    if (Ci == 8) return launch_s2z<8, 1, false, true, 5, 1>(p, st);
    return Co > 16 ? launch_s2z<16, 2, false, true, 5, 1>(p, st) : launch_s2z<16, 1, false, true, 5, 1>(p, st);
}

extern "C" int d3d_conv2d_k3s2_zs_bf16x3(const float* in, const void* wpacked, const float* scale, const float* shift,
                                         const float* skip, int act, int skip_after_act, int Ci, int Co, int H, int W, float* out,
                                         d3d_stream_t stream) {
    return conv2d_k3s2_zs(PREC_X3, in, wpacked, scale, shift, skip, act, skip_after_act, Ci, Co, H, W, out, stream);
}

extern "C" int d3d_conv2d_k3s2_zs_h16(const float* in, const void* wpacked, const float* scale, const float* shift,
                                       const float* skip, int act, int skip_after_act, int Ci, int Co, int H, int W, float* out,
                                       d3d_stream_t stream) {
    return conv2d_k3s2_zs(PREC_BF16, in, wpacked, scale, shift, skip, act, skip_after_act, Ci, Co, H, W, out, stream);
}

// The same layer over `nbatch` images in ONE launch (RED-Net's encoder, msrednet.py:352-356, for every depth slice of a stage before the
// recurrent loop: the stride-2 ConvReLUs depend on the cost slices only): in [nbatch][Ci,H,W] and out [nbatch][Co,Ho,Wo] with the given
// element strides between items; no skip.  Per item bit for bit d3d_conv2d_k3s2_zs_h16.
extern "C" int d3d_conv2d_k3s2_zs_h16_batched(const float* in, const void* wpacked, const float* scale, const float* shift, int act, int Ci,
                                               int Co, int H, int W, int nbatch, int64_t in_bstride, int64_t out_bstride, float* out,
                                               d3d_stream_t stream) {
    D3D_REQUIRE(nbatch >= 1 && in_bstride >= (int64_t)Ci * H * W && out_bstride > 0, "bad batch arguments");
    return conv2d_k3s2_zs(PREC_BF16, in, wpacked, scale, shift, nullptr, act, 1, Ci, Co, H, W, out, stream, nbatch, (long)in_bstride,
                          (long)out_bstride);
}

extern "C" int d3d_conv2d_k3s2_zs_f32(const float* in, const void* wpacked, const float* scale, const float* shift,
                                      const float* skip, int act, int skip_after_act, int Ci, int Co, int H, int W, float* out,
                                      d3d_stream_t stream) {
    return conv2d_k3s2_zs(PREC_F32, in, wpacked, scale, shift, skip, act, skip_after_act, Ci, Co, H, W, out, stream);
}

static int conv2d_k3s2_zs(int prec, const float* in, const void* wpacked, const float* scale, const float* shift, const float* skip,
                          int act, int skip_after_act, int Ci, int Co, int H, int W, float* out, d3d_stream_t stream, int nbatch,
                          long in_bstride, long out_bstride) {
    const bool f32 = prec == PREC_F32;
    D3D_REQUIRE(in && wpacked && out, "null pointer");
    D3D_REQUIRE(H > 0 && W > 0 && Ci > 0 && Co > 0, "bad dims");
    D3D_REQUIRE(act == 0 || act == 1, "bad act %d", act);
    const int Wo = (W - 1) / 2 + 1;
    if (Ci == 48 && Co <= 48 && W % 4 == 0 && prec != PREC_X3) {
        // the pair-visibility UNet (adamvs.py:198-238): 48-channel cells leave no room for a stride-2 patch, and at its image
        // sizes the layer is latency, not work -- the stride-1 tile kernel computes every position and keeps the even ones
        Z2Params p = {};
        p.in = in; p.C1 = Ci; p.wpk = reinterpret_cast<const u4*>(wpacked); p.scale = scale; p.shift = shift; p.skip = skip; p.out = out;
        p.H = H; p.W = W; p.CO = Co; p.act = act; p.skip_after_act = skip_after_act; p.sub2 = 1; p.nbatch = nbatch; p.in_bstride = in_bstride; p.out_bstride = out_bstride;
        hipStream_t st = (hipStream_t)stream;
        if (f32) return Co > 32 ? launch_z2<48, 3, 2, true>(p, st) : Co > 16 ? launch_z2<48, 2, 2, true>(p, st) : launch_z2<48, 1, 2, true>(p, st);
        return Co > 32 ? launch_z2<48, 3, 2>(p, st) : Co > 16 ? launch_z2<48, 2, 2>(p, st) : launch_z2<48, 1, 2>(p, st);
    }
    if (Ci == 32 && Co <= 64 && W % 4 == 0 && prec == PREC_BF16) {
        // RED-Net's conv3 (msrednet.py:346: ConvReLU(32, 64, stride 2) at the third level of a slice -- 172 x 116 pixels and up): 32-channel
        // cells leave no room for a stride-2 patch either; the stride-1 tile kernel with a subsampled store again (three quarters of its
        // products are dropped, at these image sizes the layer is latency: 87 us on the round-1 stream kernel, 88 calls per view)
        Z2Params p = {};
        p.in = in; p.C1 = Ci; p.wpk = reinterpret_cast<const u4*>(wpacked); p.scale = scale; p.shift = shift; p.skip = skip; p.out = out;
        p.H = H; p.W = W; p.CO = Co; p.act = act; p.skip_after_act = skip_after_act; p.sub2 = 1; p.nbatch = nbatch; p.in_bstride = in_bstride; p.out_bstride = out_bstride;
        hipStream_t st = (hipStream_t)stream;
        return Co > 48 ? launch_z2<32, 4, 2>(p, st) : Co > 32 ? launch_z2<32, 3, 2>(p, st) : Co > 16 ? launch_z2<32, 2, 2>(p, st) : launch_z2<32, 1, 2>(p, st);
    }
    // (two 65 x 17 patches must fit the LDS: bf16 cells up to C_in = 16, fp32 cells C_in = 8)
    if ((Ci != 8 && (Ci != 16 || f32)) || Co > 32 || Wo % 4 != 0) {
        set_error("d3d_conv2d_k3s2_zs_%s: C_in = %d (8%s), C_out = %d (<= 32), output width %d (multiple of 4) not taken",
                  f32 ? "f32" : prec == PREC_X3 ? "bf16x3" : "bf16", Ci, f32 ? "" : " | 16", Co, Wo);
        return D3D_ERR_UNSUPPORTED;
    }
    Z2Params p = {};
    p.in = in; p.C1 = Ci; p.wpk = reinterpret_cast<const u4*>(wpacked); p.scale = scale; p.shift = shift; p.skip = skip; p.out = out;
    p.H = H; p.W = W; p.CO = Co; p.act = act; p.skip_after_act = skip_after_act; p.nbatch = nbatch; p.in_bstride = in_bstride; p.out_bstride = out_bstride;
    hipStream_t st = (hipStream_t)stream;
    if (f32) return Co > 16 ? launch_s2z<8, 2, true>(p, st) : launch_s2z<8, 1, true>(p, st);
    if (prec == PREC_X3) {
        if (Co > 16) return Ci == 8 ? launch_s2z<8, 2, false, true>(p, st) : launch_s2z<16, 2, false, true>(p, st);
        return Ci == 8 ? launch_s2z<8, 1, false, true>(p, st) : launch_s2z<16, 1, false, true>(p, st);
    }
    if (Co > 16) return Ci == 8 ? launch_s2z<8, 2>(p, st) : launch_s2z<16, 2>(p, st);
    return Ci == 8 ? launch_s2z<8, 1>(p, st) : launch_s2z<16, 1>(p, st);
}

static int convtranspose2d_k3s2_zs(int prec, const float* in, const void* wpacked, const float* scale, const float* shift,
                                   const float* skip, int act, int skip_after_act, int Ci, int Co, int H, int W, float* out,
                                   d3d_stream_t stream);

extern "C" int d3d_convtranspose2d_k4s2_zs_bf16x3(const float* in, const void* wpacked, const float* scale, const float* shift,
                                                  const float* skip, int act, int skip_after_act, int Ci, int Co, int H, int W,
                                                  float* out, d3d_stream_t stream) {
    D3D_REQUIRE(in && wpacked && out, "null pointer");
    D3D_REQUIRE(H > 0 && W > 0 && Ci > 0 && Co > 0, "bad dims");
    D3D_REQUIRE(act == 0 || act == 1, "bad act %d", act);
    if ((Ci != 8 && Ci != 16 && Ci != 32) || Co > 16 || W % 4 != 0) {
        set_error("d3d_convtranspose2d_k4s2_zs_bf16x3: C_in = %d (8 | 16 | 32), C_out = %d (<= 16), W = %d (multiple of 4) not taken", Ci, Co, W);
        return D3D_ERR_UNSUPPORTED;
    }
    Z2Params p = {};
    p.in = in; p.C1 = Ci; p.wpk = reinterpret_cast<const u4*>(wpacked); p.scale = scale; p.shift = shift; p.skip = skip; p.out = out;
    p.H = H; p.W = W; p.CO = Co; p.act = act; p.skip_after_act = skip_after_act;
    hipStream_t st = (hipStream_t)stream;
    if (Co <= 8)   // both column parities in one tile (wpacked: ops._pack_t2d_k4fold_bf16 x 3; round 4 again: 160 us folded, 188 us unfolded at 32 -> 8, 928 x 1376)
        return Ci == 8 ? launch_tz<8, false, true, true, true>(p, st) : Ci == 16 ? launch_tz<16, false, true, true, true>(p, st)
                                                                                   : launch_tz<32, false, true, true, true>(p, st);
    return Ci == 8 ? launch_tz<8, false, true, true>(p, st) : Ci == 16 ? launch_tz<16, false, true, true>(p, st)
                                                                         : launch_tz<32, false, true, true>(p, st);
}

extern "C" int d3d_convtranspose2d_k3s2_zs_bf16x3(const float* in, const void* wpacked, const float* scale, const float* shift,
                                                  const float* skip, int act, int skip_after_act, int Ci, int Co, int H, int W,
                                                  float* out, d3d_stream_t stream) {
    return convtranspose2d_k3s2_zs(PREC_X3, in, wpacked, scale, shift, skip, act, skip_after_act, Ci, Co, H, W, out, stream);
}

extern "C" int d3d_convtranspose2d_k3s2_zs_h16(const float* in, const void* wpacked, const float* scale, const float* shift,
                                                const float* skip, int act, int skip_after_act, int Ci, int Co, int H, int W,
                                                float* out, d3d_stream_t stream) {
    return convtranspose2d_k3s2_zs(PREC_BF16, in, wpacked, scale, shift, skip, act, skip_after_act, Ci, Co, H, W, out, stream);
}

extern "C" int d3d_convtranspose2d_k3s2_zs_f32(const float* in, const void* wpacked, const float* scale, const float* shift,
                                               const float* skip, int act, int skip_after_act, int Ci, int Co, int H, int W,
                                               float* out, d3d_stream_t stream) {
    return convtranspose2d_k3s2_zs(PREC_F32, in, wpacked, scale, shift, skip, act, skip_after_act, Ci, Co, H, W, out, stream);
}

static int convtranspose2d_k3s2_zs(int prec, const float* in, const void* wpacked, const float* scale, const float* shift,
                                   const float* skip, int act, int skip_after_act, int Ci, int Co, int H, int W, float* out,
                                   d3d_stream_t stream) {
    const bool f32 = prec == PREC_F32;
    D3D_REQUIRE(in && wpacked && out, "null pointer");
    D3D_REQUIRE(H > 0 && W > 0 && Ci > 0 && Co > 0, "bad dims");
    D3D_REQUIRE(act == 0 || act == 1, "bad act %d", act);
    if ((Ci != 8 && Ci != 16 && Ci != 32) || Co > 16 || W % 4 != 0) {
        set_error("d3d_convtranspose2d_k3s2_zs_h16: C_in = %d (8 | 16 | 32), C_out = %d (<= 16), W = %d (multiple of 4) not taken", Ci, Co, W);
        return D3D_ERR_UNSUPPORTED;
    }
    Z2Params p = {};
    p.in = in; p.C1 = Ci; p.wpk = reinterpret_cast<const u4*>(wpacked); p.scale = scale; p.shift = shift; p.skip = skip; p.out = out;
    p.H = H; p.W = W; p.CO = Co; p.act = act; p.skip_after_act = skip_after_act;
    hipStream_t st = (hipStream_t)stream;
    if (f32) return Ci == 8 ? launch_tz<8, true>(p, st) : Ci == 16 ? launch_tz<16, true>(p, st) : launch_tz<32, true>(p, st);
    if (prec == PREC_X3)
        return Ci == 8 ? launch_tz<8, false, true>(p, st) : Ci == 16 ? launch_tz<16, false, true>(p, st) : launch_tz<32, false, true>(p, st);
    return Ci == 8 ? launch_tz<8>(p, st) : Ci == 16 ? launch_tz<16>(p, st) : launch_tz<32>(p, st);
}

static int conv2d_k3_zs_bf16(const float* in, int C1, const float* in2, int C2, const void* wpacked, const float* scale,
                            const float* shift, const float* skip, const float* aux1, int act, int ep_split,
                            int skip_after_act, int Co, int H, int W, float* out, double* gn_stats, int gn_split, d3d_stream_t stream);
extern "C" int d3d_conv2d_k3_zs_h16(const float* in, int C1, const float* in2, int C2, const void* wpacked, const float* scale,
                                     const float* shift, const float* skip, const float* aux1, int act, int ep_split,
                                     int skip_after_act, int Co, int H, int W, float* out, d3d_stream_t stream) {
    return conv2d_k3_zs_bf16(in, C1, in2, C2, wpacked, scale, shift, skip, aux1, act, ep_split, skip_after_act, Co, H, W, out, nullptr, 0, stream);
}
// The same layer (act 0, no skip) + the GroupNorm(1, C) statistics of its output for the normalisation that follows (module.py:62-67,
// 71-99 ConvGRUCell2): gn_stats [ngroups][2] fp64 (sum, sum of squares), ZEROED by the caller; channels >= gn_split form the second
// group (gn_split = Co: one group).  What d3d_groupnorm_stats computes from the stored tensor, without the pass over it.
extern "C" int d3d_conv2d_k3_zs_h16_gn(const float* in, int C1, const float* in2, int C2, const void* wpacked, const float* shift, int Co,
                                        int H, int W, float* out, double* gn_stats, int gn_split, d3d_stream_t stream) {
    D3D_REQUIRE(gn_stats && gn_split > 0 && gn_split <= Co, "bad statistics arguments");
    return conv2d_k3_zs_bf16(in, C1, in2, C2, wpacked, nullptr, shift, nullptr, nullptr, 0, 0, 0, Co, H, W, out, gn_stats, gn_split, stream);
}
static int conv2d_k3_zs_bf16(const float* in, int C1, const float* in2, int C2, const void* wpacked, const float* scale,
                            const float* shift, const float* skip, const float* aux1, int act, int ep_split,
                            int skip_after_act, int Co, int H, int W, float* out, double* gn_stats, int gn_split, d3d_stream_t stream) {
    D3D_REQUIRE(in && wpacked && out, "null pointer");
    D3D_REQUIRE(H > 0 && W > 0 && C1 > 0 && C2 >= 0 && Co > 0, "bad dims");
    D3D_REQUIRE(act >= 0 && act <= 3, "bad act %d", act);
    D3D_REQUIRE(C2 == 0 || in2, "second input missing");
    D3D_REQUIRE(act < 2 || skip, "GRU epilogue (act %d) needs the state h in `skip`", act);
    D3D_REQUIRE(act != 2 || (ep_split > 0 && ep_split <= Co), "GRU gate epilogue: bad ep_split %d", ep_split);
    D3D_REQUIRE(act != 3 || aux1, "GRU update epilogue needs the update gate u in `aux1`");
    const int Ci = C1 + C2;
    // (48 -> <= 48: the pair-visibility UNet of AdaMVS, adamvs.py:198-238)
    // (24 | 40 -> <= 16: conv_gru1 of the RED-Net slice regulariser at stages 2 / 1, msrednet.py:340: 16 | 32 cost channels + 8 state channels)
    const bool shape = (((Ci == 8 || Ci == 16 || Ci == 32) && Co <= 32) || (Ci == 48 && Co <= 48) || ((Ci == 24 || Ci == 40) && Co <= 16)) &&
                       C1 % 8 == 0 && C2 % 8 == 0 && W % 4 == 0;
    if (!shape) {
        set_error("d3d_conv2d_k3_zs_h16: C_in = %d + %d (8 | 16 | 32 | 48 in groups of 8; 24 | 40 with C_out <= 16), C_out = %d (<= 32; <= 48 with C_in = 48), "
                  "W = %d (multiple of 4) not taken", C1, C2, Co, W);
        return D3D_ERR_UNSUPPORTED;
    }
    Z2Params p = {};
    p.in = in; p.in2 = in2; p.C1 = C1; p.wpk = reinterpret_cast<const u4*>(wpacked); p.scale = scale; p.shift = shift;
    p.skip = skip; p.aux1 = aux1; p.out = out; p.H = H; p.W = W; p.CO = Co; p.act = act; p.ep_split = ep_split;
    p.skip_after_act = skip_after_act;
    p.gn = gn_stats; p.gn_split = gn_split;
    hipStream_t st = (hipStream_t)stream;
    if (gn_stats) {   // the layers of ConvGRUCell2's finest levels (msrednet.py:337-370): 16 | 24 | 32 | 40 -> 8 | 16 | 32
        if (Ci == 40 && Co <= 16) return launch_z2<40, 1, 2, false, false, false, true>(p, st);
        if (Ci == 24 && Co <= 16) return launch_z2<24, 1, 4, false, false, false, true>(p, st);
        if (Ci == 32) return Co > 16 ? launch_z2<32, 2, 2, false, false, false, true>(p, st) : launch_z2<32, 1, 2, false, false, false, true>(p, st);
        if (Ci == 16) return Co > 16 ? launch_z2<16, 2, 4, false, false, false, true>(p, st) : launch_z2<16, 1, 4, false, false, false, true>(p, st);
        set_error("d3d_conv2d_k3_zs_h16_gn: C_in = %d, C_out = %d not taken", Ci, Co);
        return D3D_ERR_UNSUPPORTED;
    }
    if (Ci == 48) return Co > 32 ? launch_z2<48, 3, 2>(p, st) : Co > 16 ? launch_z2<48, 2, 2>(p, st) : launch_z2<48, 1, 2>(p, st);
    if (Ci == 40) return launch_z2<40, 1, 2>(p, st);
    if (Ci == 24) return launch_z2<24, 1, 4>(p, st);
    if (Co > 16) {
        if (Ci == 32) return launch_z2<32, 2, 2>(p, st);
        if (Ci == 16) return launch_z2<16, 2, 4>(p, st);
        return launch_z2<8, 2, 4>(p, st);
    }
    if (Ci == 32) return launch_z2<32, 1, 2>(p, st);
    if (Ci == 16) return launch_z2<16, 1, 4>(p, st);
    return launch_z2<8, 1, 4>(p, st);
}
