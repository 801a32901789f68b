// The last two layers of every CostRegNet in ONE kernel (bf16 mode, channel-last volumes):
//     y    = conv0 + ReLU(BN(ConvTranspose3d(16 -> 8, k 3, stride 2, pad 1, output_pad 1)(x)))   (cas_mvsnet.py:103,118: conv11)
//     prob = Conv3d(8 -> 1, k 3, pad 1)(y) + bias                                                  (cas_mvsnet.py:105,119)
// Unfused (conv_t2.hip FOLD form, then conv_c8.hip KZF form) the full-resolution 8-channel volume y is written once and
// read once (with the 3 x 3 halo: 1.3 x) -- 37 of the 60 bytes per voxel the two layers move.  Here y never leaves the CU:
//   * a workgroup (8 waves) owns 30 x 10 coarse cells (= 60 x 20 outputs per plane) and walks the output planes; per plane it
//     computes y for 32 x 12 cells (one cell of halo on every side: the 3 x 3 window of the probability layer) with the
//     x-folded GEMM of conv_t2.hip (M = 16 cells, rows of the weight operand = (column parity, channel), K = taps x 16
//     channels), adds conv0's cells (the only full-resolution READ), rounds to bf16 exactly like the unfused layer's store
//     and writes the 64 x 24 cells into LDS (zeros outside the volume: the padding of the probability layer);
//   * the probability layer then runs from that LDS plane in the k_z-folded form of conv_c8.hip: columns 0, 1, 2 of ONE
//     accumulator tile are the three open output planes, the finished one (column 2) is stored as fp32 and the columns move
//     on (v_mov_dpp row_shr:1) -- 10 tiles of 16 pixels per wave;
//   * coarse input planes are staged as in conv_t2.hip (two resident, the next one loaded while an even plane is computed).
// Same K order, same epilogue expressions and the same rounding points as the two kernels it replaces: the result is
// bit-identical to theirs (tests/test_parity_gpu.py::test_conv11_prob_fused_is_the_two_layers).
// HBM per full-resolution voxel: 4 B (x, 1/8 of the voxels x 32 B) + 16 B x 1.28 (conv0, halo) + 4 B (prob) = 28.5 B for 60.
#include <cstdint>
#include "common.h"

#include <type_traits>


namespace d3d {

namespace {

typedef float f4 __attribute__((ext_vector_type(4)));
typedef unsigned u4 __attribute__((ext_vector_type(4)));

constexpr int CI = 16;                          // channels of x
constexpr int NW = 8, NT = 64 * NW;
constexpr int RXI = 32, RYI = 12;               // coarse cells whose y is computed per plane
constexpr int OXI = RXI - 2, OYI = RYI - 2;     // coarse cells whose 2 x 2 outputs the workgroup finishes
constexpr int PXI = RXI + 1, PYI = RYI + 1;     // staged coarse patch: one more column / row for the d = 1 taps
constexpr int CS = bf16_cell_bytes(CI);         // bytes per coarse cell: 32 (bank-conflict-free pitch: common.h)
constexpr int PATCH = PXI * PYI * CS;
constexpr int YP = 2 * RXI + 18, YR = 2 * RYI;  // y plane in LDS: 16-byte cells, rows of 82 -- the last pixel group's lanes 12..15 read past column 63, and
                                                // taps (0, 2) | (1, 0) of one lane group sit a whole bank row apart (YP - 2 = 80 cells; tools/conv_bank_sim.py)
constexpr int YBYTES = YP * YR * 16;
constexpr int TTILES = RYI * (RXI / 16) / NW;   // conv11 tiles (coarse row, 16-cell group) per wave: 3
constexpr int PTILES = 2 * OYI * 4 / NW;        // probability tiles (output row, 16-pixel group) per wave: 10

constexpr int ntaps2(int pz, int py) { return (1 + pz) * (1 + py) * 2; }
constexpr int nkb2(int pz, int py) { return (ntaps2(pz, py) * CI + 31) / 32; }
constexpr int fold_base2(int c) {
    int s = 0;
    for (int q = 0; q < c; ++q) s += nkb2(q >> 1, q & 1);
    return s;
}
constexpr int NFT = fold_base2(4);              // fragments of conv11 (ops._pack_t2_fold_bf16): 9
constexpr int NFP = 3;                          // fragments of the probability layer (ops._pack_c8_kzfold_bf16, C_in = 8): 72 -> 96 rows
constexpr int LDS_BYTES = 2 * PATCH + YBYTES + (NFT + NFP) * 64 * 16;
static_assert(LDS_BYTES <= 80 * 1024, "two workgroups per CU");

struct T2PParams {
    const void* in;       // [D, H, W, 16] bf16
    const u4* wt;         // conv11, x-folded packing
    const float* scale;   // [8] or null
    const float* shift;   // [8] or null
    const void* skip;     // [2D, 2H, 2W, 8] bf16 or null (added after the activation)
    const u4* wp;         // probability layer, k_z-folded packing
    const float* pbias;   // [1] or null
    float* out;           // [2D, 2H, 2W] fp32
    int D, H, W;
    int relu;
    int ozper;            // output planes per workgroup (even)
};

__device__ __forceinline__ unsigned pack_h16_p(float a, float b) {
    return pack_h16x2(a, b);   // one packed conversion (common.h: pack_h16x2)
}

__device__ __forceinline__ f4 unpack_h16x4_p(uint2 u) {
    return (f4){h16_lo(u.x), h16_hi(u.x),
                h16_lo(u.y), h16_hi(u.y)};
}

__device__ __forceinline__ float dpp_row_shr1_p(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x111, 0xf, 0xf, true));
}

// The kernel is bound by instruction issue, not by a latency (profiles/r04_t2p.txt: every part taken out shortens it by its own
// instruction count; a vector or scalar instruction costs the SIMD four cycles of that issue port whatever it computes), so
// everything that does not change from plane to plane -- staging and y-cell addresses, in-volume masks, the K offsets of the
// A operands, store offsets -- is per-lane state computed once, and the plane walk is addresses by immediate offsets:
// 32-bit offsets inside a plane from a scalar plane base.
template <bool SKIP>
__global__ __launch_bounds__(NT, 4) void convt3d_prob_kernel(T2PParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // LDS: y plane first (its cell offsets then fit the 16-bit immediates of the DS instructions), coarse planes, weights
    unsigned char* const ybuf = smem;
    unsigned char* const cbuf = smem + YBYTES;
    u4* const wt = reinterpret_cast<u4*>(cbuf + 2 * PATCH);
    u4* const wp = wt + NFT * 64;

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int D = p.D, H = p.H, W = p.W, OH = 2 * H, OW = 2 * W;
    const int ix0 = blockIdx.x * OXI, iy0 = blockIdx.y * OYI;   // first coarse cell the workgroup finishes
    const int cx0 = ix0 - 1, cy0 = iy0 - 1;                     // first coarse cell it computes
    const int oz0 = blockIdx.z * p.ozper, oz1 = min(oz0 + p.ozper, 2 * D);

    const float pbias = p.pbias ? p.pbias[0] : 0.0f;
    for (int i = tid; i < NFT * 64; i += NT) wt[i] = p.wt[i];
    for (int i = tid; i < NFP * 64; i += NT) wp[i] = p.wp[i];

    // ---- coarse planes: task = (patch cell, 8-channel half); loads clamped into the volume (always issued), zeros outside --
    constexpr int NTASK = PXI * PYI * 2;
    constexpr int ROUNDS = (NTASK + NT - 1) / NT;
    const unsigned iplane = (unsigned)H * W * (CI * 2);          // bytes of a coarse plane (host: < 2^31)
    unsigned stoff[ROUNDS];                                      // byte offset of the task's 16 bytes inside a coarse plane
    int stdst[ROUNDS];                                           // LDS offset inside a patch, -1: no task
    bool stok[ROUNDS];
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {
        const int task = tid + r * NT, pix = task >> 1, g = task & 1;
        const int py = pix / PXI, px = pix - py * PXI;
        const int gx = cx0 + px, gy = cy0 + py;
        stok[r] = task < NTASK && gx >= 0 && gx < W && gy >= 0 && gy < H;
        stoff[r] = task >= NTASK ? 0 : ((unsigned)min(max(gy, 0), H - 1) * W + min(max(gx, 0), W - 1)) * (CI * 2) + g * 16;   // (no task: every such lane reads the plane's first bytes -- one line -- instead of real cells beyond the patch)
        stdst[r] = task < NTASK ? pix * CS + g * 16 : -1;
    }
    u4 stc[ROUNDS];
    bool stzin = false;   // the plane in the staging registers is inside the volume
    auto issue = [&](int zi) {
        const bool zin = zi >= 0 && zi < D;
        stzin = zin;
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

    // ---- conv11: this wave's tiles are coarse rows r0 + 4 t (t < 3), 16-cell group mg ----------------------------------
    // D row (lane >> 4) * 4 + r = (column parity, channel), column = coarse cell lane & 15
    const int g4 = lane >> 4, cb = (g4 & 1) * 4, pxo = g4 >> 1, m = lane & 15;
    const int r0 = wave >> 1, mg = wave & 1;
    const f4 sc = p.scale ? *reinterpret_cast<const f4*>(p.scale + cb) : (f4){1, 1, 1, 1};
    const f4 sh = p.shift ? *reinterpret_cast<const f4*>(p.shift + cb) : (f4){0, 0, 0, 0};
    const int abase = (r0 * PXI + mg * 16 + m) * CS;             // A operands: + t * 4 * PXI * CS + K offset
    const int ywr = ((2 * r0) * YP + 2 * (mg * 16 + m) + pxo) * 16 + cb * 2;   // y cell: + (8 t + PY) * YP * 16
    const unsigned oplane = (unsigned)OH * OW * 16;             // bytes of a full-resolution 8-channel plane (host: < 2^31)
    bool ins[TTILES][2];                                         // the cell is inside the volume
    unsigned skoff[TTILES][2];                                   // conv0's 8 bytes of it inside a plane
    {
        const int X = 2 * (cx0 + mg * 16 + m) + pxo;
#pragma unroll
        for (int t = 0; t < TTILES; ++t)
#pragma unroll
            for (int py = 0; py < 2; ++py) {
                const int Y = 2 * (cy0 + r0 + 4 * t) + py;
                ins[t][py] = X >= 0 && X < OW && Y >= 0 && Y < OH;
                skoff[t][py] = ((unsigned)min(max(Y, 0), OH - 1) * OW + min(max(X, 0), OW - 1)) * 16 + cb * 2;
            }
    }
    // K offsets of the A operands, per (parity class, K block): k = 32 kb + 8 (lane >> 4) -> tap (dz, dy, dx) dz-major, channel.
    // 16 channels: two taps per K block, no padded taps, and dz = kb / (1 + PY) is the same for every lane (the buffer, see yplane).
    int aoff[NFT];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int kb = 0; kb < nkb2(c >> 1, c & 1); ++kb) {
            const int PY = c & 1;
            const int k0 = 32 * kb + 8 * g4, tp = k0 / CI, ch = k0 % CI;
            const int dx = tp & 1, dy = (tp >> 1) % (1 + PY);
            aoff[fold_base2(c) + kb] = (dy * PXI + dx) * CS + ch * 2;
        }
    uint2 sk[TTILES][2];
    auto load_skip = [&](int oz) {   // conv0's cells of y plane oz (clamped into the volume: cells outside are zeroed by `ins`)
        if constexpr (SKIP) {
            const unsigned char* __restrict__ src = static_cast<const unsigned char*>(p.skip) + (size_t)min(max(oz, 0), 2 * D - 1) * oplane;
#pragma unroll
            for (int t = 0; t < TTILES; ++t)
#pragma unroll
                for (int py = 0; py < 2; ++py)
                    sk[t][py] = *reinterpret_cast<const uint2*>(src + skoff[t][py]);
        }
    };
    // par = parity of the coarse plane iz = oz >> 1 the y plane starts from: plane iz lives in buffer par, iz + 1 in the other
    auto yplane = [&](auto pzc, int par) {
        constexpr int PZ = decltype(pzc)::value;
        const unsigned char* const ab0 = cbuf + par * PATCH + abase;          // coarse plane oz >> 1
        const unsigned char* const ab1 = cbuf + (par ^ 1) * PATCH + abase;    // the next one (dz = 1 taps of an odd y plane)
#pragma unroll
        for (int t = 0; t < TTILES; ++t) {
            auto half = [&](auto pyc) {
                constexpr int PY = decltype(pyc)::value;
                constexpr int NKB = nkb2(PZ, PY);
                constexpr int FB = fold_base2(PZ * 2 + PY);
                f4 acc = {0, 0, 0, 0};
#pragma unroll
                for (int kb = 0; kb < NKB; ++kb) {
                    const unsigned char* abuf = kb / (1 + PY) ? ab1 : ab0;
                    const int ao = aoff[FB + kb];
                    const h16x8 wf = __builtin_bit_cast(h16x8, wt[(FB + kb) * 64 + lane]);
                    const h16x8 a = __builtin_bit_cast(h16x8, *reinterpret_cast<const u4*>(abuf + t * (4 * PXI * CS) + ao));
                    acc = mfma_h16(wf, a, acc);
                }
                f4 v = acc * sc + sh;
                if (p.relu) v = __builtin_elementwise_max(v, (f4){0, 0, 0, 0});
                if constexpr (SKIP) v += unpack_h16x4_p(sk[t][PY]);
                uint2 pk = {pack_h16_p(v[0], v[1]), pack_h16_p(v[2], v[3])};
                if (!ins[t][PY]) pk = (uint2){0, 0};                 // the zero padding of the probability layer
                *reinterpret_cast<uint2*>(ybuf + ywr + (8 * t + PY) * (YP * 16)) = pk;
            };
            half(std::integral_constant<int, 0>{});
            half(std::integral_constant<int, 1>{});
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    // ---- probability layer from the LDS plane: this wave's tiles are output rows q0 + 2 j (j < 10), 16-pixel group pg;
    //      columns 0, 1, 2 of a tile = output planes zi + 1, zi, zi - 1 ----------------------------------------------------
    const int q0 = wave >> 2, pg = wave & 3;
    f4 acc[PTILES];
#pragma unroll
    for (int j = 0; j < PTILES; ++j) acc[j] = (f4){0, 0, 0, 0};
    int sbase[NFP];                                              // A operands: K index 32 kb + 8 (lane >> 4) + c -> tap (k_y, k_x), channel c
#pragma unroll
    for (int kb = 0; kb < NFP; ++kb) {
        const int tp = 4 * kb + g4;
        const int ky = tp < 9 ? tp / 3 : 0, kx = tp < 9 ? tp % 3 : 0;   // padded taps read a valid cell; their weights are zero
        sbase[kb] = ((1 + ky + q0) * YP + 1 + kx + m + 16 * pg) * 16;  // + j * 2 * YP * 16
    }
    auto sweep = [&]() {
#pragma unroll
        for (int kb = 0; kb < NFP; ++kb) {
            const h16x8 bw = __builtin_bit_cast(h16x8, wp[kb * 64 + lane]);
#pragma unroll
            for (int j = 0; j < PTILES; ++j) {
                const h16x8 a = __builtin_bit_cast(h16x8, *reinterpret_cast<const u4*>(ybuf + sbase[kb] + j * (2 * YP * 16)));
                acc[j] = mfma_h16(a, bw, acc[j]);
                if (j % 5 == 4) __builtin_amdgcn_sched_barrier(0);   // five operand loads in flight, not thirty (registers)
            }
        }
    };
    // D layout: row (pixel) = (lane >> 4) * 4 + register, column 2 = the plane that has seen its three y planes
    const int xl = 16 * pg + g4 * 4;
    const bool slane = m == 2 && xl < 2 * OXI && 2 * ix0 + xl < OW;   // OW % 4 == 0: a quad is inside or outside as a whole
    const unsigned soff = ((unsigned)(2 * iy0 + q0) * OW + 2 * ix0 + xl) * 4;   // + j * 2 * OW * 4 inside an output plane
    const unsigned splane = (unsigned)OH * OW * 4;
    auto store_shift = [&](int zo) {
        if (zo >= oz0 && zo < oz1) {
            unsigned char* dst = reinterpret_cast<unsigned char*>(p.out) + (size_t)zo * splane;
#pragma unroll
            for (int j = 0; j < PTILES; ++j)
                if (slane && 2 * iy0 + q0 + 2 * j < OH)
                    *reinterpret_cast<f4*>(dst + soff + j * (2 * OW * 4)) = acc[j] + pbias;
        }
#pragma unroll
        for (int j = 0; j < PTILES; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[j][r] = dpp_row_shr1_p(acc[j][r]);
    };

    // ---- walk the y planes oz0 - 1 .. oz1: coarse plane iz lives in buffer iz & 1 ------------------------------------
    // Pairs (odd plane, even plane), unrolled so that the staging registers have ONE live range per pair: issued after the odd
    // plane's conv11, committed after the even plane's -- unconditionally (planes outside the volume stage zeros nobody reads).
    const int zs = oz0 - 1, izs = zs >> 1;                          // oz0 is even: the first y plane is odd, it needs izs and izs + 1
    issue(izs);
    commit(cbuf + (izs & 1) * PATCH);
    issue(izs + 1);
    commit(cbuf + ((izs + 1) & 1) * PATCH);
    load_skip(zs);
    __syncthreads();
    for (int zo = zs; zo < oz1; zo += 2) {
        {
            const int zi = zo, iz = zi >> 1;
            const bool live = zi >= 0 && zi < 2 * D;
            if (live) yplane(std::integral_constant<int, 1>{}, iz & 1);
            issue(iz + 2);                                          // for the odd plane zi + 2
            load_skip(zi + 1);
            lds_barrier();                                          // y plane complete
            if (live) sweep();
            store_shift(zi - 1);
            lds_barrier();                                          // y plane free
        }
        {
            const int zi = zo + 1, iz = zi >> 1;
            const bool live = zi < 2 * D;
            if (live) yplane(std::integral_constant<int, 0>{}, iz & 1);
            commit(cbuf + ((iz + 1) & 1) * PATCH);                  // that buffer held plane iz - 1, last read by y plane zi - 1
            load_skip(zi + 1);
            lds_barrier();
            if (live) sweep();
            store_shift(zi - 1);
            lds_barrier();
        }
    }
}

}  // namespace

}  // namespace d3d

using namespace d3d;

extern "C" int d3d_convtranspose3d_prob_cl_h16(const void* in, const void* wt_folded, const float* scale, const float* shift,
                                                const void* skip, int relu, const void* wprob_kzfolded, const float* prob_bias,
                                                int D, int H, int W, float* out, d3d_stream_t stream) {
    D3D_REQUIRE(in && wt_folded && wprob_kzfolded && out, "null pointer");
    D3D_REQUIRE(D > 0 && H > 0 && W > 0, "bad dims %dx%dx%d", D, H, W);
    const int gx = ceil_div(W, OXI), gy = ceil_div(H, OYI);
    if (W % 2 != 0 || gy > 65535 || 2 * D > 65535 || (long)H * W * 64 >= (1L << 31) ||   // (32-bit offsets inside a plane)
        ((reinterpret_cast<uintptr_t>(in) | reinterpret_cast<uintptr_t>(skip) | reinterpret_cast<uintptr_t>(out)) & 15)) {
        set_error("d3d_convtranspose3d_prob_cl_h16: W = %d (even: rows of 2 W floats in 16-byte quads), plane %d x %d, 16-byte aligned volumes needed", W, H, W);
        return D3D_ERR_UNSUPPORTED;
    }
    auto kern = skip ? convt3d_prob_kernel<true> : convt3d_prob_kernel<false>;
    int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(kern), LDS_BYTES);
    if (rc != D3D_OK) return rc;
    T2PParams p = {};
    p.in = in; p.wt = reinterpret_cast<const u4*>(wt_folded); p.scale = scale; p.shift = shift; p.skip = skip;
    p.wp = reinterpret_cast<const u4*>(wprob_kzfolded); p.pbias = prob_bias; p.out = out;
    p.D = D; p.H = H; p.W = W; p.relu = relu;
    // (every z segment recomputes two y planes; rounds 2-4: doubled until 1536 workgroups -- 1024 | 3072: the same to 3 %, profiles/r04_t2p.txt)
    p.ozper = 2 * pick_zper((long)gx * gy, D, 4, 2, LDS_BYTES, 2, 1536);
    hipLaunchKernelGGL(kern, dim3(gx, gy, ceil_div(2 * D, p.ozper)), dim3(NT), LDS_BYTES, (hipStream_t)stream, p);
    D3D_LAUNCH_CHECK("convt3d_prob_kernel launch");
    return D3D_OK;
}
