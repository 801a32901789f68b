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
#include <climits>

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
    const float* aux1;   // act == 3: the update gate u
    float* out;
    int Ci0, Ci1, Co, M;
    int D, H, W;
    int Do, Ho, Wo;
    int Gz, Gy, Gx;
    int cz, cy, cx;
    int sz, sy, sx;
    int bz, by, bx;
    int fz, fy, fx;
    int act, skip_after_act, ep_split;  // act: 0 none, 1 ReLU, 2 GRU gates, 3 GRU state update (see the header)
    int ntaps, zmin, zspan, ymin, yspan, xmin, xspan;
    int PY, PX, CS, CiP;
    int nseg, mg_nseg, mg_py;  // staging: 16-lane segments per patch row, 16-bit reciprocal multipliers
    float inv_nseg;            // 1 / nseg for the unit -> (row, segment) split
    int zseg;                  // output columns in z per workgroup
    int vec, sh;               // vec: staging by 16-byte loads (W % 4 == 0); patch origin moved left by sh columns to a multiple of 4
    int wyn;                   // waves along y: 4 (tile 4 rows x 16*NT columns) or 1 (1 row x 64*NT columns)
    int tzstart[16];           // taps sorted by tz: taps of tz = zmin+i are [tzstart[i], tzstart[i+1])
    signed char ty[ZS_MAXTAPS], tx[ZS_MAXTAPS];
};

typedef float f4v __attribute__((ext_vector_type(4)));
typedef unsigned u4v __attribute__((ext_vector_type(4)));

#ifdef D3D_CONV_STATS
// debug build only (tools/run_convstats.sh): per-phase cycle sums of wave 0 of every workgroup
__device__ unsigned long long g_conv_stats[8];
#define ZS_T(var) const unsigned long long var = __builtin_amdgcn_s_memtime()
#define ZS_ADD(i, a, b) st[i] += (b) - (a)
#else
#define ZS_T(var)
#define ZS_ADD(i, a, b)
#endif

typedef h16x4 bf4v;   // four operands of the library's 16-bit format (common.h)

// BF16 = true: the same kernel with bf16 MFMA operands (v_mfma_f32_16x16x16_bf16, fp32 accumulate) -- the
// precision BASELINE's config 3 names.  Activations are rounded (RNE, v_cvt_pk_bf16_f32) as they are staged: the
// LDS patch holds, per channel quad and position, the four bf16 channels as one 8-byte entry, so a B operand is
// one ds_read_b64 (the first bf16 version kept the patch fp32 and paid four ds_read_b32 + two conversions per
// operand: LDS-read bound).  The resident weights are stored as bf16.  One MFMA covers K = 16: one tap x 16 channels
// (CK = 16) or two taps x 8 channels (CK = 8).  Needs the 16-byte staging mode (W % 4 == 0).
//
// SPLIT = true: the workgroup has eight waves -- waves 0-3 sweep and flush (the MFMA side), waves 4-7 stage the
// NEXT step's patch into the other half of a double-buffered LDS patch, one barrier per step.  Staging (loads,
// LDS writes, index arithmetic) then never sits in the MFMA waves' instruction stream.
template <int MT, int NT, int NS, int CK, bool BF16, bool SPLIT>
__global__ __launch_bounds__(SPLIT ? 512 : 256) void conv_stream_kernel(ConvZParams p) {
    constexpr int THREADS = SPLIT ? 512 : 256;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int MP = 16 * MT;
    constexpr int WS = (MP == 16) ? 16 : MP + 16;  // weight row stride: k-groups g, g+1 on disjoint bank halves
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = (tid >> 6) & 3;                 // tile role of an MFMA wave (loader waves: unused)
    const bool loader = SPLIT && (tid >> 8) != 0;    // waves 4-7 of a split workgroup
    const int g = lane >> 4, j = lane & 15;
    const int Ci = p.Ci0 + p.Ci1, CiP = p.CiP, PX = p.PX, PY = p.PY, CS = p.CS;
    // staging rows: (chunk channel, patch row) -- bf16: (channel quad, patch row), four channels per 8-byte entry
    const int srows = (BF16 ? CK / 4 : CK) * PY, srows16 = (srows + 15) & ~15;
    constexpr int QENT = 2;  // floats per bf16 quad entry

#ifdef D3D_CONV_STATS
    unsigned long long st[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const unsigned long long t_begin = __builtin_amdgcn_s_memtime();
#endif
    float* wl = lds;                                                     // [ntaps*CiP][WS] resident weights
    // bf16: [ntaps][chunks][4 k-slots][WS] entries of 4 bf16 (8 bytes) = 2 floats per entry
    const int wl_floats = BF16 ? p.ntaps * (CiP / CK) * 4 * WS * 2 : p.ntaps * CiP * WS;
    int* tofft = reinterpret_cast<int*>(lds + wl_floats);                // [ntaps] patch offset of each tap
    float* ssl = reinterpret_cast<float*>(tofft + ((p.ntaps + 3) & ~3)); // [2][64] per-channel scale, shift
    int2* rowt = reinterpret_cast<int2*>(ssl + 128);                     // [srows16] staging row table
    float* xin = reinterpret_cast<float*>(rowt + srows16);               // [CK][CS] one input plane chunk

    // wave arrangement inside the tile: 4 x 1 (volumes) or 1 x 4 (single-row planes: 2D images streamed row by row)
    const int wy = p.wyn == 4 ? wave : 0, wx = p.wyn == 4 ? 0 : wave, wxn = 4 / p.wyn;
    const int tx0 = blockIdx.x * wxn * (16 * NT);  // first column of the workgroup's tile
    const int gx0 = tx0 + wx * (16 * NT);          // first column of this wave
    const int gy = blockIdx.y * p.wyn + wy;
    const int gz_lo = blockIdx.z * p.zseg;
    const int gz_hi = min(p.Gz, gz_lo + p.zseg);
    const int iy0 = blockIdx.y * p.wyn * p.cy + p.ymin, ix0 = tx0 * p.cx + p.xmin;
    const int lcx = p.cx == 4 ? 2 : (p.cx == 2 ? 1 : 0), PXq = PX >> lcx;
    const int bbase = g * CS + (wy * p.cy) * PX + (wx * (16 * NT) + j);
    const long in_plane = (long)p.H * p.W, in_vol = in_plane * p.D;
    const long out_plane = (long)p.Ho * p.Wo;
    const int zmax = p.zmin + p.zspan - 1;

    // ---- one-time: resident weights (rows beyond Ci are zero), tap offsets, epilogue constants, row table
    {
        if constexpr (BF16) {
            // entry (t, chunk c, k-slot g, row m): 4 consecutive k of row m.  CK = 16: channels c*16 + 4g.. of tap t;
            // CK = 8: channels c*8 + 4(g&1).. of tap t + (g>>1) -- zero when that tap belongs to another z group
            // (the sweep pairs taps inside a group) or does not exist.
            const int nch = CiP / CK, nent = p.ntaps * nch * 4 * MP;
            bf4v* wl16 = reinterpret_cast<bf4v*>(wl);
            for (int e = tid; e < nent; e += THREADS) {
                const int m = e % MP, r1 = e / MP;
                const int gg = r1 & 3, r2 = r1 >> 2;
                const int c = r2 % nch, t = r2 / nch;
                int tt = t, cb = c * CK + 4 * gg;
                bool live = true;
                if (CK == 8) {
                    tt = t + (gg >> 1);
                    cb = c * CK + 4 * (gg & 1);
                    int te = p.ntaps;
                    for (int i = 0; i < p.zspan; ++i)
                        if (t >= p.tzstart[i] && t < p.tzstart[i + 1]) te = p.tzstart[i + 1];
                    live = tt < te;
                }
                float v[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int ci = cb + q;
                    const bool ok = live && ci < Ci;
                    const float w = p.wpack[((long)(ok ? tt : 0) * Ci + (ok ? ci : 0)) * MP + m];
                    v[q] = ok ? w : 0.0f;
                }
                wl16[r1 * WS + m] = cvt_h16x4(v[0], v[1], v[2], v[3]);
            }
        }
        const int rows = BF16 ? 0 : p.ntaps * CiP;
        const int n4 = rows * (MP / 4);
        for (int e0 = tid; e0 < n4; e0 += THREADS * 4) {
            float4 v[4];
            int dsto[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int e = e0 + u * THREADS;
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
        // Patch rows are stored de-interleaved by the column step cx: x -> (x % cx) * (PX / cx) + x / cx, so the 16
        // columns of a B tile (x = j*cx + tap offset) are 16 consecutive LDS words whatever the fold / stride
        if (tid < p.ntaps) {
            const int o = (p.tx[tid] - p.xmin) + p.sh;
            tofft[tid] = (p.ty[tid] - p.ymin) * PX + (o & (p.cx - 1)) * (PX >> lcx) + (o >> lcx);
        }
        if (tid < 64) {
            ssl[tid] = (p.scale && tid < p.Co) ? p.scale[tid] : 1.0f;
            ssl[64 + tid] = (p.shift && tid < p.Co) ? p.shift[tid] : 0.0f;
        }
        // row r = (chunk channel cc, patch row y): source element offset inside a (chunk, plane) block, INT_MIN
        // when the row lies outside the image; LDS offset of the row (| cc << 20), or -1 for a padding row
        for (int r = tid; r < srows16; r += THREADS) {
            int2 e = make_int2(INT_MIN, -1);
            if (r < srows) {
                const int cc = r / PY, y = r - cc * PY;  // bf16: cc = channel quad
                const int sy = iy0 + y;
                if ((unsigned)sy < (unsigned)p.H) e.x = (int)((BF16 ? 4 * cc : cc) * in_vol + (long)sy * p.W + ix0 - p.sh);
                e.y = (cc * CS + y * PX) | (cc << 20);
            }
            rowt[r] = e;
        }
    }

    // ---- this lane's accumulator rows m = mt*16 + 4g + r -> output element offset of (c_out, fold position)
    // relative to the column origin, -1 when the row is unused or its y position is outside the output
    int rowoff[MT][4];
    int rowco[MT], rowfz[MT];  // c_out / z fold position of the four rows, one byte each
    unsigned long long colmask = 0;  // bit (mt*4+r)*NT+n: column n of that row lands inside the output
    int coloff[NT];
#pragma unroll
    for (int n = 0; n < NT; ++n) coloff[n] = (gx0 + n * 16 + j) * p.sx + p.bx;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        rowco[mt] = 0;
        rowfz[mt] = 0;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int m = mt * 16 + 4 * g + r;
            int off = -1;
            if (m < p.M && gy < p.Gy) {
                const int q0 = m / p.Co, co = m - q0 * p.Co;
                const int q1 = q0 / p.fx, fxv = q0 - q1 * p.fx;
                const int fzv = q1 / p.fy, fyv = q1 - fzv * p.fy;
                const int oy = gy * p.sy + p.by + fyv;
                if (oy < p.Ho) {
                    off = (int)(((long)co * p.Do + fzv) * out_plane + (long)oy * p.Wo + fxv);
                    rowco[mt] |= co << (8 * r);
                    rowfz[mt] |= fzv << (8 * r);
#pragma unroll
                    for (int n = 0; n < NT; ++n)
                        if (gx0 + n * 16 + j < p.Gx && coloff[n] + fxv < p.Wo)
                            colmask |= 1ull << ((mt * 4 + r) * NT + n);
                }
            }
            rowoff[mt][r] = off;
        }
    }

    f4v acc[NS][MT][NT];
#pragma unroll
    for (int s = 0; s < NS; ++s)
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int n = 0; n < NT; ++n) acc[s][m][n] = (f4v){0, 0, 0, 0};

    // flush accumulator slot 0 as output column gz (affine, skip, ReLU, store), then rotate the slots
    auto flush = [&](int gz) {
        const int ozb = gz * p.sz + p.bz;
        const long ubase = (long)ozb * out_plane;  // uniform part of the output offset
        float* __restrict__ outp = p.out + ubase;
        const float* __restrict__ skp = p.skip ? p.skip + ubase : nullptr;
        // (values laundered through empty asm statements are loop invariant: left visible, the compiler hoists
        // every per-element offset and predicate out of the plane loop and pins them in registers)
        unsigned mlo = (unsigned)colmask, mhi = (unsigned)(colmask >> 32);
        asm volatile("" : "+v"(mlo), "+v"(mhi));
        const unsigned long long cmask = ((unsigned long long)mhi << 32) | mlo;
        // skip values first, all in flight together (clamped addresses, no branches), then the epilogue proper.
        // (The 32-row variants batch per 16-row tile instead: the full batch costs them an occupancy step.)
        constexpr bool BATCH_SKIP = (MT != 2);
        // GRU epilogues (module.py:24-51 fused into the gate / candidate convolutions): act 2: y = sigmoid(y), rows
        // c_out < ep_split (the reset gate) multiplied by h = skip; act 3 (image kernels only): h' = u*h + (1-u)*tanh(y)
        // with h = skip, u = aux1.
        const bool gru_gate = p.act == 2, gru_upd = SPLIT && MT == 1 && p.act == 3;
        const float* __restrict__ a1p = gru_upd ? p.aux1 + ubase : nullptr;
        float sk[BATCH_SKIP ? MT : 1][4][NT];
        float sk2[(SPLIT && MT == 1) ? 4 : 1][NT];
        if (BATCH_SKIP && skp) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    int off = rowoff[mt][r];
                    asm volatile("" : "+v"(off));
                    const bool rowok = !gru_gate || ((rowco[mt] >> (8 * r)) & 255) < p.ep_split;  // h has ep_split channels
#pragma unroll
                    for (int n = 0; n < NT; ++n) {
                        const bool ok = (off >= 0) & rowok & (((cmask >> ((mt * 4 + r) * NT + n)) & 1ull) != 0);
                        sk[mt][r][n] = skp[ok ? (unsigned)(off + coloff[n]) : 0u];
                        if constexpr (SPLIT && MT == 1)
                            if (gru_upd) sk2[r][n] = a1p[ok ? (unsigned)(off + coloff[n]) : 0u];
                    }
                }
        }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            if (!BATCH_SKIP && skp) {  // this tile's skip values, all in flight together
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    int off = rowoff[mt][r];
                    asm volatile("" : "+v"(off));
                    const bool rowok = !gru_gate || ((rowco[mt] >> (8 * r)) & 255) < p.ep_split;
#pragma unroll
                    for (int n = 0; n < NT; ++n) {
                        const bool ok = (off >= 0) & rowok & (((cmask >> ((mt * 4 + r) * NT + n)) & 1ull) != 0);
                        sk[0][r][n] = skp[ok ? (unsigned)(off + coloff[n]) : 0u];
                    }
                }
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                int off = rowoff[mt][r];
                asm volatile("" : "+v"(off));
                if (off >= 0 && ozb + ((rowfz[mt] >> (8 * r)) & 255) < p.Do) {  // (a z fold may overhang the volume)
                    const int co = (rowco[mt] >> (8 * r)) & 255;
                    const float sc = ssl[co], sh = ssl[64 + co];
#pragma unroll
                    for (int n = 0; n < NT; ++n) {
                        if ((cmask >> ((mt * 4 + r) * NT + n)) & 1ull) {
                            const unsigned e = (unsigned)(off + coloff[n]);
                            float y = acc[0][mt][n][r] * sc + sh;
                            float skv = 0.0f;
                            if (skp && (!gru_gate || co < p.ep_split)) skv = sk[BATCH_SKIP ? mt : 0][r][n];
                            if (gru_gate) {
                                y = gru_sigmoid_as<BF16>(y);
                                if (co < p.ep_split) y *= skv;
                            } else if (gru_upd) {
                                const float u = sk2[(SPLIT && MT == 1) ? r : 0][n];
                                y = u * skv + (1.0f - u) * gru_tanh_as<BF16>(y);
                            } else {
                                if (skp && !p.skip_after_act) y += skv;
                                if (p.act == 1) y = fmaxf(y, 0.0f);
                                if (skp && p.skip_after_act) y = skv + y;
                            }
                            outp[e] = y;
                        }
                    }
                }
                if (r == 3) __builtin_amdgcn_sched_barrier(0);  // one 16-row tile at a time: bounded registers
            }
        }
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

    // ---- staging of one step = (input plane zi, channel chunk c), software-pipelined through registers: the
    // loads of step k+1 are issued before the tap sweep of step k and land in LDS after it.  A 16-lane group
    // `grp` takes patch rows grp, grp+16, ...; its lanes run along x in 16-wide segments.  Item i = (row
    // iteration k, segment seg); the row table supplies the row's source / LDS offsets, so an item costs a
    // handful of integer instructions.
    constexpr int PF = BF16 ? 48 : 32;  // floats a thread can hold in flight (larger patches finish synchronously)
    float pv[PF];
    const int grp = (tid & 255) >> 4, xs = tid & 15;
    float* xw = xin;        // patch buffer being written (split: alternates per step)
    const float* xr = xin;  // patch buffer being read
    // staging units = (patch row, segment), dealt round-robin to the sixteen 16-lane groups of the staging waves
    const int nunits = srows * p.nseg;
    const int nitems = (nunits + 15) >> 4;
    // Staging loads are raw buffer loads: base (uniform, per step) + 32-bit byte offset, and the hardware range
    // check returns 0 for offset 0x80000000 -- out-of-image / beyond-C_in elements need no branch and no select.
    // Per item: a row-table read and about ten full-rate integer instructions (no 64-bit or multiply ops).
    auto step_rsrc = [&](int zi, int c) {
        const int c0 = c * CK;  // a chunk never straddles the two inputs (host: Ci0 % CK == 0 when Ci1 > 0)
        const float* b = (c0 < p.Ci0) ? p.in0 + (long)c0 * in_vol : p.in1 + (long)(c0 - p.Ci0) * in_vol;
        b += (long)zi * in_plane;
        const long span = (long)(CK - 1) * in_vol + in_plane;  // bytes reachable from b inside this (chunk, plane)
        return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(b), 0, (int)(span * 4),
                                                 0x00020000);
    };
    // item i -> LDS slot (or -1) and source byte offset (0x80000000 = reads as zero).  Slots beyond the last item
    // repeat it (same load, same LDS word), so batches of 8 carry no branches.
    auto item_row = [&](int i, int nit, int grpo, int& seg) -> int2 {
        const int ic = min(i, nit - 1);                  // uniform (batches are padded by repeating the last item)
        const int u = min(grpo + 16 * ic, nunits - 1);   // this group's unit (tail groups repeat the last unit)
        const int row = (int)(((float)u + 0.5f) * p.inv_nseg);
        seg = u - row * p.nseg;
        return rowt[row];
    };
    const int ix0s = ix0 - p.sh;  // image column of patch column 0
    int dk4[4];                   // LDS offsets of the four columns of an aligned float4 (de-interleaved rows)
#pragma unroll
    for (int k = 0; k < 4; ++k) dk4[k] = (k & (p.cx - 1)) * PXq + (k >> lcx);
    auto item_addr = [&](int2 e, int seg, int xso, int cmax, int& dst, unsigned& boff) {
        // scalar mode: 16 lanes x 1 float per segment; vec mode: 16 lanes x 4 floats (xso = 4 * lane-in-group)
        const int x = (p.vec ? seg * 64 : seg * 16) + xso;
        // (bitwise &, not &&: short-circuit evaluation turns each item into branches with its own LDS waits)
        const bool ok = (e.x != INT_MIN) & ((unsigned)(ix0s + x) < (unsigned)p.W) & ((BF16 ? 4 * (e.y >> 20) : (e.y >> 20)) < cmax);
        boff = ok ? (unsigned)(e.x + x) << 2 : 0x80000000u;  // beyond num_records (< 2 GiB): reads as zero
        dst = ((e.y >= 0) & (x < PX)) ? (e.y & 0xfffff) + (x & (p.cx - 1)) * PXq + (x >> lcx) : -1;  // (vec: x % 4 == 0)
    };
    auto issue = [&](int zi, int c) {
        const auto rs = step_rsrc(zi, c);
        const int cmax = Ci - c * CK;
        int grpo = grp, xso = p.vec ? 4 * xs : xs, nit = nitems;
        // laundered: the item arithmetic is loop invariant; hoisted out of the plane loop it would pin hundreds
        // of registers (and the uniform parts would be spilled SGPRs read back with VALU instructions)
        asm volatile("" : "+v"(grpo), "+v"(xso), "+s"(nit));
        if constexpr (BF16) {
            // item = (channel quad, patch row, 4 columns): four 16-byte loads, one per channel (scalar offset)
#pragma unroll
            for (int b = 0; b < PF / 16; ++b) {
                if (b < nit) {
                    int seg, dst;
                    unsigned boff;
                    const int2 e = item_row(b, nit, grpo, seg);
                    item_addr(e, seg, xso, cmax, dst, boff);
                    const int cq = 4 * (e.y >> 20);
#pragma unroll
                    for (int ch = 0; ch < 4; ++ch) {
                        const unsigned bo = (cq + ch < cmax) ? boff : 0x80000000u;
                        const u4v q = __builtin_amdgcn_raw_buffer_load_b128(rs, bo, (unsigned)((long)ch * in_vol * 4), 0);
                        const f4v f = __builtin_bit_cast(f4v, q);
#pragma unroll
                        for (int k = 0; k < 4; ++k) pv[16 * b + 4 * ch + k] = f[k];
                    }
                }
            }
            return;
        }
        if (p.vec) {  // 16-byte loads: a quarter of the instructions
#pragma unroll
            for (int b4 = 0; b4 < PF / 4; b4 += 4) {
                if (b4 < nit) {
                    int2 e[4];
                    int seg[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) e[u] = item_row(b4 + u, nit, grpo, seg[u]);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        int dst;
                        unsigned boff;
                        item_addr(e[u], seg[u], xso, cmax, dst, boff);
                        // (bit-cast the whole vector: extracting u32 elements and casting each makes this compiler
                        // narrow the load to one dword and splat it)
                        const u4v q = __builtin_amdgcn_raw_buffer_load_b128(rs, boff, 0, 0);
                        const f4v f = __builtin_bit_cast(f4v, q);
#pragma unroll
                        for (int k = 0; k < 4; ++k) pv[4 * (b4 + u) + k] = f[k];
                    }
                }
            }
            return;
        }
#pragma unroll
        for (int b8 = 0; b8 < PF; b8 += 8) {
            if (b8 < nit) {
                int2 e[8];
                int seg[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) e[u] = item_row(b8 + u, nit, grpo, seg[u]);
                __builtin_amdgcn_sched_barrier(0);  // the eight row-table reads in flight together, one wait
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    int dst;
                    unsigned boff;
                    item_addr(e[u], seg[u], xso, cmax, dst, boff);
                    pv[b8 + u] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, boff, 0, 0));
                }
            }
        }
    };
    auto land = [&](int zi, int c) {
        const auto rs = step_rsrc(zi, c);
        const int cmax = Ci - c * CK;
        int grpo = grp, xso = p.vec ? 4 * xs : xs, nit = nitems;
        asm volatile("" : "+v"(grpo), "+v"(xso), "+s"(nit));
        if constexpr (BF16) {
            bf4v* xw8 = reinterpret_cast<bf4v*>(xw);
            auto pack = [](float a0, float a1, float a2, float a3) { return cvt_h16x4(a0, a1, a2, a3); };
#pragma unroll
            for (int b = 0; b < PF / 16; ++b) {
                if (b < nit) {
                    int seg, dst;
                    unsigned boff;
                    const int2 e = item_row(b, nit, grpo, seg);
                    item_addr(e, seg, xso, cmax, dst, boff);
                    if (dst >= 0) {
#pragma unroll
                        for (int k = 0; k < 4; ++k)
                            xw8[dst + dk4[k]] = pack(pv[16 * b + k], pv[16 * b + 4 + k], pv[16 * b + 8 + k], pv[16 * b + 12 + k]);
                    }
                }
            }
            for (int i = PF / 16; i < nit; ++i) {  // beyond the register window
                int seg, dst;
                unsigned boff;
                const int2 e = item_row(i, nit, grpo, seg);
                item_addr(e, seg, xso, cmax, dst, boff);
                const int cq = 4 * (e.y >> 20);
                f4v v[4];
#pragma unroll
                for (int ch = 0; ch < 4; ++ch) {
                    const unsigned bo = (cq + ch < cmax) ? boff : 0x80000000u;
                    const u4v q = __builtin_amdgcn_raw_buffer_load_b128(rs, bo, (unsigned)((long)ch * in_vol * 4), 0);
                    v[ch] = __builtin_bit_cast(f4v, q);
                }
                if (dst >= 0) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) xw8[dst + dk4[k]] = pack(v[0][k], v[1][k], v[2][k], v[3][k]);
                }
            }
            return;
        }
        if (p.vec) {
#pragma unroll
            for (int b4 = 0; b4 < PF / 4; b4 += 4) {
                if (b4 < nit) {
                    int2 e[4];
                    int seg[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) e[u] = item_row(b4 + u, nit, grpo, seg[u]);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        int dst;
                        unsigned boff;
                        item_addr(e[u], seg[u], xso, cmax, dst, boff);
                        if (dst >= 0) {
#pragma unroll
                            for (int k = 0; k < 4; ++k) xw[dst + dk4[k]] = pv[4 * (b4 + u) + k];
                        }
                    }
                }
            }
            for (int i0 = PF / 4; i0 < nit; i0 += 4) {  // beyond the register window
                f4v v[4];
                int dsto[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    unsigned boff;
                    int seg;
                    const int2 e = item_row(i0 + u, nit, grpo, seg);
                    item_addr(e, seg, xso, cmax, dsto[u], boff);
                    const u4v q = __builtin_amdgcn_raw_buffer_load_b128(rs, boff, 0, 0);
                    v[u] = __builtin_bit_cast(f4v, q);
                }
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (dsto[u] >= 0) {
#pragma unroll
                        for (int k = 0; k < 4; ++k) xw[dsto[u] + dk4[k]] = v[u][k];
                    }
            }
            return;
        }
#pragma unroll
        for (int b8 = 0; b8 < PF; b8 += 8) {
            if (b8 < nit) {
                int2 e[8];
                int seg[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) e[u] = item_row(b8 + u, nit, grpo, seg[u]);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    int dst;
                    unsigned boff;
                    item_addr(e[u], seg[u], xso, cmax, dst, boff);
                    if (dst >= 0) xw[dst] = pv[b8 + u];
                }
            }
        }
        for (int i0 = PF; i0 < nit; i0 += 8) {  // beyond the register window
            float v[8];
            int dsto[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                unsigned boff;
                int seg;
                const int2 e = item_row(i0 + u, nit, grpo, seg);
                item_addr(e, seg, xso, cmax, dsto[u], boff);
                v[u] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, boff, 0, 0));
            }
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (dsto[u] >= 0) xw[dsto[u]] = v[u];
        }
    };
    auto sweep = [&](int zi, int c) {
        // every open output column takes the taps whose tz links it to this plane
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const int gzs = gbase + s;
            const int tzv = zi - gzs * p.cz;
            if (gzs >= gz_hi || tzv < p.zmin || tzv > zmax) continue;
            const int tb = p.tzstart[tzv - p.zmin], te = p.tzstart[tzv - p.zmin + 1];
            if (tb >= te) continue;
            if constexpr (BF16) {
                const bf4v* __restrict__ wl16 = reinterpret_cast<const bf4v*>(wl);
                const int nch = CiP / CK;
                const bf4v* __restrict__ xr8 = reinterpret_cast<const bf4v*>(xr);
                const int bb = bbase - g * CS;  // patch entry of this lane's column, quad 0
                constexpr int TSTEP = (CK == 8) ? 2 : 1;
                for (int t = tb; t < te; t += TSTEP) {
                    // this lane's k-slot: channel quad qd at tap t (CK = 16: qd = g) or at tap t + (g>>1) (CK = 8: qd = g&1)
                    const int tl = (CK == 8) ? min(t + (g >> 1), te - 1) : t;
                    const int qd = (CK == 8) ? (g & 1) : g;
                    const bf4v* __restrict__ xb = xr8 + bb + tofft[tl] + qd * CS;
                    const bf4v* __restrict__ wa = wl16 + ((t * nch + c) * 4 + g) * WS + j;
                    bf4v b[NT];
#pragma unroll
                    for (int n = 0; n < NT; ++n) b[n] = xb[n * 16];
#pragma unroll
                    for (int m = 0; m < MT; ++m) {
                        const bf4v av = wa[m * 16];
#pragma unroll
                        for (int n = 0; n < NT; ++n)
                            acc[s][m][n] = mfma_h16_k16(av, b[n], acc[s][m][n]);
                    }
                }
                continue;
            }
            int toff = tofft[tb];
            for (int t = tb; t < te; ++t) {
                const int toff_next = tofft[min(t + 1, p.ntaps - 1)];
                const float* __restrict__ xb = xr + bbase + toff;
                const float* __restrict__ wa = wl + ((t * CiP + c * CK + g) * WS + j);
#pragma unroll
                for (int kk = 0; kk < CK / 4; ++kk) {
                    float b[NT];
#pragma unroll
                    for (int n = 0; n < NT; ++n) b[n] = xb[kk * 4 * CS + n * 16];
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
    };

    int zi = zi_first < 0 ? 0 : zi_first;  // planes outside the volume are zero padding: skipped
    const int zend = zi_last >= p.D ? p.D - 1 : zi_last;
    int c = 0;
    ZS_T(p0);
    __syncthreads();  // row table, weights, tap table, epilogue constants written
    ZS_T(p1);
    ZS_ADD(4, p0, p1);
    if constexpr (SPLIT) {
        // both roles walk the same step sequence (plane zi, chunk c); step s lives in patch buffer s & 1
        auto advance = [&](int& z, int& ch) {
            if (++ch == nchunks) {
                ch = 0;
                ++z;
            }
        };
        if (zi <= zend) {
            if (loader) {
                for (int sidx = 0;; ++sidx) {
                    xw = xin + (sidx & 1) * (BF16 ? (CK / 4) * CS * QENT : CK * CS);
                    issue(zi, c);
                    land(zi, c);
                    __syncthreads();  // step sidx staged; the MFMA waves are done with step sidx - 1
                    advance(zi, c);
                    if (zi > zend) break;
                }
            } else {
                __syncthreads();  // step 0 staged
                for (int sidx = 0;; ++sidx) {
                    if (c == 0) {
                        ZS_T(f0);
                        while (gbase < gz_hi && zi > gbase * p.cz + zmax) {
                            flush(gbase);
                            ++gbase;
                        }
                        ZS_T(f1);
                        ZS_ADD(3, f0, f1);
                    }
                    xr = xin + (sidx & 1) * (BF16 ? (CK / 4) * CS * QENT : CK * CS);
                    ZS_T(c0);
                    sweep(zi, c);
                    ZS_T(c1);
                    ZS_ADD(2, c0, c1);
#ifdef D3D_CONV_STATS
                    st[6] += 1;
#endif
                    advance(zi, c);
                    if (zi > zend) break;
                    __syncthreads();  // next step staged
                    ZS_T(b1);
                    ZS_ADD(4, c1, b1);
                }
            }
        }
        if (loader) return;
    } else if (zi <= zend) {
        issue(zi, 0);
        ZS_T(p2);
        ZS_ADD(0, p1, p2);
        land(zi, 0);
        ZS_T(p3);
        ZS_ADD(1, p2, p3);
        __syncthreads();
        for (;;) {
            if (c == 0) {
                ZS_T(f0);
                while (gbase < gz_hi && zi > gbase * p.cz + zmax) {
                    flush(gbase);
                    ++gbase;
                }
                ZS_T(f1);
                ZS_ADD(3, f0, f1);
            }
            int nzi = zi, nc = c + 1;
            if (nc == nchunks) {
                nc = 0;
                ++nzi;
            }
            const bool has_next = nzi <= zend;
            ZS_T(s0);
            if (has_next) issue(nzi, nc);
            ZS_T(c0);
            ZS_ADD(0, s0, c0);
            sweep(zi, c);
            ZS_T(c1);
            ZS_ADD(2, c0, c1);
#ifdef D3D_CONV_STATS
            st[6] += 1;
#endif
            if (!has_next) break;
            __syncthreads();  // every wave has finished reading the current patch
            ZS_T(b1);
            ZS_ADD(4, c1, b1);
            land(nzi, nc);
            ZS_T(l1);
            ZS_ADD(1, b1, l1);
            __syncthreads();
            zi = nzi;
            c = nc;
        }
    }
    {
        ZS_T(e0);
        while (gbase < gz_hi) {
            flush(gbase);
            ++gbase;
        }
        ZS_T(e1);
        ZS_ADD(3, e0, e1);
    }
#ifdef D3D_CONV_STATS
    if (tid == 0) {
        st[5] = __builtin_amdgcn_s_memtime() - t_begin;
        st[7] = 1;
        for (int i = 0; i < 8; ++i) atomicAdd(&g_conv_stats[i], st[i]);
    }
#endif
}

struct StreamCfg {
    int MT, NT, NS, CK;
    int lds_bytes;
};

static int patch_stride(int PY, int PX, int /*cx*/) {
    int cs = PY * PX;
    cs += (16 - (cs & 31) + 32) & 31;  // CS = 16 mod 32: the k-groups of a half-wave read disjoint bank halves
    return cs;
}

static int patch_rows(const ConvZParams& p) { return p.yspan + (p.wyn - 1) * p.cy; }
static int patch_cols(const ConvZParams& p, int NT) {  // LDS row length (vec mode: origin shifted, whole float4s)
    const int px = p.xspan + (16 * NT * (4 / p.wyn) - 1) * p.cx;
    return p.vec ? ((px + p.sh + 3) & ~3) : (px + p.cx - 1) / p.cx * p.cx;  // a multiple of cx (1, 2 or 4)
}

static int lds_bytes_for(const ConvZParams& p, int MT, int NT, int CK, bool bf16 = false, bool split = false) {
    const int MP = 16 * MT, WS = (MP == 16) ? 16 : MP + 16;
    const int Ci = p.Ci0 + p.Ci1, CiP = (Ci + CK - 1) / CK * CK;
    if (bf16) {
        const int PYb = patch_rows(p), PXb = patch_cols(p, NT);
        const int srows16b = ((CK / 4) * PYb + 15) & ~15;
        return 4 * (p.ntaps * (CiP / CK) * 4 * WS * 2 + ((p.ntaps + 3) & ~3) + 128 + 2 * srows16b +
                    (split ? 2 : 1) * (CK / 4) * patch_stride(PYb, PXb, p.cx) * 2);
    }
    const int PY = patch_rows(p), PX = patch_cols(p, NT);
    const int srows16 = (CK * PY + 15) & ~15;
    return 4 * (p.ntaps * CiP * WS + ((p.ntaps + 3) & ~3) + 128 + 2 * srows16 +
                (split ? 2 : 1) * CK * patch_stride(PY, PX, p.cx));
}

template <int MT, int NT, int NS, int CK, bool BF16, bool SPLIT>
static int launch_stream(ConvZParams& p, hipStream_t stream) {
    const int Ci = p.Ci0 + p.Ci1;
    p.CiP = (Ci + CK - 1) / CK * CK;
    p.PY = patch_rows(p);
    p.PX = patch_cols(p, NT);
    p.CS = patch_stride(p.PY, p.PX, p.cx);
    p.nseg = p.vec ? (p.PX + 63) / 64 : (p.PX + 15) / 16;
    p.mg_nseg = 65536 / p.nseg + 1;
    p.inv_nseg = 1.0f / (float)p.nseg;
    p.mg_py = 65536 / p.PY + 1;
    const int bytes = lds_bytes_for(p, MT, NT, CK, BF16, SPLIT);
    auto kern = conv_stream_kernel<MT, NT, NS, CK, BF16, SPLIT>;
    static bool attr_set = false;
    if (!attr_set) {
        int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(kern), 160 * 1024);
        if (rc != D3D_OK) return rc;
        attr_set = true;
    }
    const int nx = ceil_div(p.Gx, 16 * NT * (4 / p.wyn)), ny = ceil_div(p.Gy, p.wyn);
    // enough workgroups to fill 256 CUs several times over, at the price of the halo planes and the one-time
    // prologue per z segment (row-streamed images: fewer, longer segments)
    int nz = ceil_div(p.wyn == 1 ? 1024 : 4096, (long)nx * ny);
    nz = nz < 1 ? 1 : (nz > p.Gz ? p.Gz : nz);
    p.zseg = ceil_div(p.Gz, nz);
    nz = ceil_div(p.Gz, p.zseg);
    if (ny > 65535 || nz > 65535) {
        set_error("conv_stream: grid %dx%dx%d too large", nx, ny, nz);
        return D3D_ERR_UNSUPPORTED;
    }
    hipLaunchKernelGGL(kern, dim3(nx, ny, nz), dim3(SPLIT ? 512 : 256), bytes, stream, p);
    D3D_LAUNCH_CHECK("conv_stream_kernel launch");
    return D3D_OK;
}

template <int MT, int NT, int CK, bool BF16>
static int launch_sp(ConvZParams& p, int NS, bool split, hipStream_t stream) {
    if (split) {
        if (NS == 1) return launch_stream<MT, NT, 1, CK, BF16, true>(p, stream);
        if (NS == 2) return launch_stream<MT, NT, 2, CK, BF16, true>(p, stream);
        return launch_stream<MT, NT, 3, CK, BF16, true>(p, stream);
    }
    if (NS == 1) return launch_stream<MT, NT, 1, CK, BF16, false>(p, stream);
    if (NS == 2) return launch_stream<MT, NT, 2, CK, BF16, false>(p, stream);
    return launch_stream<MT, NT, 3, CK, BF16, false>(p, stream);
}

template <int MT, int NT, int CK>
static int launch_ns(ConvZParams& p, int NS, bool bf16, bool split, hipStream_t stream) {
    if (bf16) return launch_sp<MT, NT, CK, true>(p, NS, split, stream);
    return launch_sp<MT, NT, CK, false>(p, NS, split, stream);
}

template <int MT, int NT>
static int launch_ck(ConvZParams& p, int NS, int CK, bool bf16, bool split, hipStream_t stream) {
    if (CK == 16) return launch_ns<MT, NT, 16>(p, NS, bf16, split, stream);
    return launch_ns<MT, NT, 8>(p, NS, bf16, split, stream);
}

}  // namespace

}  // namespace d3d

using namespace d3d;

extern "C" {

#ifdef D3D_CONV_STATS
// debug build only: [issue loads, land in LDS (incl. wait), sweep, flush, barrier after sweep, total, steps, workgroups]
int d3d_conv_stream_stats(unsigned long long* out8, int reset) {
    if (hipDeviceSynchronize() != hipSuccess) return D3D_ERR_HIP;
    if (out8 && hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_conv_stats), 64) != hipSuccess) return D3D_ERR_HIP;
    if (reset) {
        unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_conv_stats), z, 64) != hipSuccess) return D3D_ERR_HIP;
    }
    return D3D_OK;
}
#endif

}  // extern "C"

static int conv_fold_impl(bool bf16, const float* in0, int Ci0, const float* in1, int Ci1, const float* wpack, int mpad,
                          int M, const float* scale, const float* shift, const float* skip, int skip_after_act, int act,
                          const float* aux1, int ep_split, int Co, int D, int H, int W, int Do, int Ho, int Wo, const int* geom, int ntaps,
                          const signed char* taps_zyx, float* out, d3d_stream_t stream) {
    D3D_REQUIRE(in0 && wpack && out && taps_zyx && geom, "null pointer");
    D3D_REQUIRE(Ci0 > 0 && Ci1 >= 0 && (Ci1 == 0 || in1), "bad input channel split %d+%d", Ci0, Ci1);
    D3D_REQUIRE(D > 0 && H > 0 && W > 0 && Do > 0 && Ho > 0 && Wo > 0, "bad dims");
    D3D_REQUIRE(ntaps > 0 && ntaps <= ZS_MAXTAPS, "bad ntaps %d", ntaps);
    D3D_REQUIRE(act >= 0 && act <= 3, "bad act %d", act);
    D3D_REQUIRE(act < 2 || skip, "GRU epilogue (act %d) needs the state h in `skip`", act);
    D3D_REQUIRE(act != 2 || (ep_split > 0 && ep_split <= Co), "GRU gate epilogue: bad ep_split %d", ep_split);
    D3D_REQUIRE(act != 3 || aux1, "GRU update epilogue needs the update gate u in `aux1`");
    ConvZParams p = {};
    p.in0 = in0; p.in1 = in1; p.wpack = wpack; p.scale = scale; p.shift = shift; p.skip = skip; p.out = out;
    p.Ci0 = Ci0; p.Ci1 = Ci1; p.Co = Co; p.M = M; p.D = D; p.H = H; p.W = W; p.Do = Do; p.Ho = Ho; p.Wo = Wo;
    p.Gz = geom[0]; p.Gy = geom[1]; p.Gx = geom[2];
    p.cz = geom[3]; p.cy = geom[4]; p.cx = geom[5];
    p.sz = geom[6]; p.sy = geom[7]; p.sx = geom[8];
    p.bz = geom[9]; p.by = geom[10]; p.bx = geom[11];
    p.fz = geom[12]; p.fy = geom[13]; p.fx = geom[14];
    p.act = act; p.skip_after_act = skip_after_act ? 1 : 0; p.ntaps = ntaps;
    p.aux1 = aux1; p.ep_split = ep_split;
    D3D_REQUIRE(p.Gz > 0 && p.Gy > 0 && p.Gx > 0, "bad column grid %dx%dx%d", p.Gz, p.Gy, p.Gx);
    D3D_REQUIRE(p.cz > 0 && p.cy > 0 && (p.cx == 1 || p.cx == 2 || p.cx == 4) && p.sz > 0 && p.sy > 0 && p.sx > 0,
                "bad steps (column step must be 1, 2 or 4)");
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
    D3D_REQUIRE(p.zspan <= 15, "z tap span %d too large", p.zspan);
    {
        int t = 0;
        for (int i = 0; i <= p.zspan; ++i) {
            while (t < ntaps && taps_zyx[3 * t] < p.zmin + i) ++t;
            p.tzstart[i] = t;
        }
    }
    // 16-byte staging loads when every patch row starts on a 16-byte boundary: W % 4 == 0, aligned tensors, and the
    // patch origin moved left to a multiple of 4 columns (tile origins are multiples of 16 columns)
    p.vec = (W % 4 == 0) && ((reinterpret_cast<uintptr_t>(in0) | reinterpret_cast<uintptr_t>(in1)) & 15) == 0;
#ifdef D3D_EXPERIMENTS
    if (getenv("D3D_CONV_SCALAR_STAGING")) p.vec = false;
#endif
    p.sh = p.vec ? (p.xmin & 3) : 0;
    // single-row planes (a 2D image handed over as [C, rows, 1, W]): the four waves sit side by side
    p.wyn = (H == 1 && Ho == 1 && p.yspan == 1 && p.fy == 1 && p.Gy == 1) ? 1 : 4;
    const int open_cols = (p.zspan + p.cz - 1) / p.cz;
    D3D_REQUIRE(open_cols <= 3, "more than 3 open output columns in z (span %d, step %d)", p.zspan, p.cz);
    const int NS = open_cols;
    const int Ci = Ci0 + Ci1;
    // tile / chunk choice: narrow tiles when the folded patch is wide; 16-channel chunks when they still leave
    // room for two workgroups per CU
    int NT = 4, CK = (Ci > 8) ? 16 : 8;
    if (MT == 1 && lds_bytes_for(p, MT, 4, 8, bf16) > 72 * 1024) NT = 1;
    if (MT == 1 && ((8 * patch_rows(p) + 15) / 16) * ((patch_cols(p, 4) + 15) / 16) > 48) NT = 1;  // register window
    if (CK == 16 && lds_bytes_for(p, MT, NT, 16, bf16) > 72 * 1024) CK = 8;
    const long in_plane = (long)H * W, in_vol = in_plane * D;
    // 32-bit element offsets inside one (chunk, plane) block and inside the output; a chunk must not straddle
    // the two concatenated inputs
    if (CK == 16 && (15 * in_vol + in_plane >= (1L << 29) || (Ci1 > 0 && Ci0 % 16 != 0))) CK = 8;
    if (7 * in_vol + in_plane >= (1L << 29) || (Ci1 > 0 && Ci0 % 8 != 0) || (long)Co * Do * Ho * Wo >= (1L << 30)) {
        set_error("conv_stream: tensor too large for 32-bit offsets, or input split %d+%d not chunk aligned", Ci0, Ci1);
        return D3D_ERR_UNSUPPORTED;
    }
    if (bf16 && !p.vec) {
        set_error("conv_stream: the bf16 kernels need 16-byte staging (W %% 4 == 0 and 16-byte aligned inputs)");
        return D3D_ERR_UNSUPPORTED;
    }
    const int bytes = lds_bytes_for(p, MT, NT, CK, bf16);
    if (bytes > 156 * 1024) {
        set_error("conv_stream: resident weights + patch need %d B of LDS (ntaps=%d Ci=%d M=%d)", bytes, ntaps, Ci, M);
        return D3D_ERR_UNSUPPORTED;
    }
    {
        // staging index arithmetic uses 16-bit reciprocals: check their exact range on the host
        const int PY = patch_rows(p), PX = patch_cols(p, NT);
        const int nseg = p.vec ? (PX + 63) / 64 : (PX + 15) / 16, rows = CK * PY;
        const int items = ((rows + 15) / 16) * nseg + 8;
        D3D_REQUIRE(items < 1024 && rows + 16 < 1024 && nseg <= 64 && PY <= 64 && rows * nseg < 8192, "patch %dx%d too large",
                    PY, PX);
    }
    hipStream_t st = (hipStream_t)stream;
    // Loader / MFMA wave split (double-buffered patch): measured 8-11 % faster on row-streamed images (short
    // steps), 5-10 % slower on volumes (fewer MFMA waves per CU) -- so images only.  D3D_CONV_SPLIT=0|1 overrides.
    bool split = p.wyn == 1;
#ifdef D3D_EXPERIMENTS
    if (const char* e = getenv("D3D_CONV_SPLIT")) split = atoi(e) != 0;
#endif
    split = split && lds_bytes_for(p, MT, NT, CK, bf16, true) <= 156 * 1024 && !(MT == 4 && NS == 3);
    if (act == 3 && !(split && MT == 1)) {
        set_error("conv_stream: the GRU update epilogue exists in the image (split) kernels with <= 16 GEMM rows only");
        return D3D_ERR_UNSUPPORTED;
    }
    if (NT == 1) return launch_ck<1, 1>(p, NS, CK, bf16, split, st);
    switch (MT) {
        case 1: return launch_ck<1, 4>(p, NS, CK, bf16, split, st);
        case 2: return launch_ck<2, 4>(p, NS, CK, bf16, split, st);
        default: return launch_ck<4, 4>(p, NS, CK, bf16, split, st);
    }
}

extern "C" {

int d3d_conv_fold_f32(const float* in0, int Ci0, const float* in1, int Ci1, const float* wpack, int mpad, int M,
                      const float* scale, const float* shift, const float* skip, int skip_after_act, int act,
                      const float* aux1, int ep_split, int Co, int D, int H, int W, int Do, int Ho, int Wo,
                      const int* geom, int ntaps, const signed char* taps_zyx, float* out, d3d_stream_t stream) {
    return conv_fold_impl(false, in0, Ci0, in1, Ci1, wpack, mpad, M, scale, shift, skip, skip_after_act, act, aux1, ep_split,
                          Co, D, H, W, Do, Ho, Wo, geom, ntaps, taps_zyx, out, stream);
}

int d3d_conv_fold_h16(const float* in0, int Ci0, const float* in1, int Ci1, const float* wpack, int mpad, int M,
                       const float* scale, const float* shift, const float* skip, int skip_after_act, int act,
                       const float* aux1, int ep_split, int Co, int D, int H, int W, int Do, int Ho, int Wo,
                       const int* geom, int ntaps, const signed char* taps_zyx, float* out, d3d_stream_t stream) {
    return conv_fold_impl(true, in0, Ci0, in1, Ci1, wpack, mpad, M, scale, shift, skip, skip_after_act, act, aux1, ep_split,
                          Co, D, H, W, Do, Ho, Wo, geom, ntaps, taps_zyx, out, stream);
}

}  // extern "C"

