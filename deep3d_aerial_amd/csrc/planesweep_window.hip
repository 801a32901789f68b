// Window plane-sweep kernel for gfx950: the SHALLOW sweeps of the cascades (cas_mvsnet.py:211-226 -- stages 2 and 3 sweep 32 /
// 8 hypotheses around the previous stage's depth, a few source pixels of disparity in all).
//
// The ring kernel (planesweep_tiled.hip) is built for deep sweeps: a planner picks step sizes, loader waves stream window
// deltas into torus rings while compute waves sweep, one 12-wave workgroup owns all of a CU's LDS.  On an 8-plane sweep
// a workgroup of that kernel lives ~38 k cycles of which ~6 k are arithmetic: the planning chain, the first window's
// latency and the step barriers are all exposed because nothing else runs on the CU.  Here instead:
//   * workgroup = a 32 x 8 patch of reference pixels x ALL the planes of its segment (<= 32) x all 8-channel groups; 8 waves
//     (4 pixel waves x 2 plane sub-ranges), HALF a CU's LDS and 128 registers per lane, so TWO workgroups share a CU and
//     one's loads run under the other's arithmetic;
//   * no planner, no rings, no loader waves: per source view ONE window -- the hull of the patch over the depth range of
//     the chunk of planes being swept (the projection is monotone in x, y and d separately, so the hull of the eight box
//     corners bounds every sample; same margins and clamps as the ring kernel's planner) -- staged by all waves with
//     coalesced buffer loads from the planar maps, positions outside the image as zeros (= zero padding per tap);
//   * every wave derives the windows itself from the same inputs (32 lanes = 8 corners x 4 views, DPP reductions,
//     v_readlane): no planning barrier, the only barriers are "window staged" and "window free";
//   * LDS layout: per view and 4-channel quad a dense [rows][cols] array of 16-byte cells, no padding: the 16 lanes a
//     ds_read_b128 services together hold 16 consecutive pixels of a row, whose taps are 16 consecutive cells (64 banks);
//   * if the windows of the whole segment do not fit, the segment is swept in chunks (first try: what fitted last time plus a
//     quarter; then three quarters of that ...); if not even one plane per sub-range fits (p.z <= 0 at a corner, a depth
//     discontinuity inside the patch) the chunk gathers from global memory with the ring kernel's fallback arithmetic;
//   * the channel-last bf16 volume leaves in planes of 8-channel groups (CL8, sweep_params.h): whole-cell stores.
// Arithmetic per sample is the ring kernel's, instruction for instruction (geometry with two roundings + rcp/Newton, packed
// blend in nw, ne, sw, se order, sum / sum of squares in view order): results are bit-identical to it and to the direct kernel.
#include "common.h"

#include "sweep_params.h"
#include "sweep_device.h"

#include <cstdlib>

namespace d3d {

int pack_channel_last_g8(const SweepParams& p, hipStream_t stream);   // planesweep_tiled.hip

namespace {

constexpr int WTW = 32;               // patch width: one 128-byte output row segment per lane row
// Shapes of workgroup on the 32 x 8 patch.  Built: 8 waves (4 pixel waves x 2 plane sub-ranges), 8-channel groups, half a CU's
// LDS, 128 registers.  Behind -DD3D_WINDOW_CG16 (measured, slower: see launch_window_ch): 12 waves (4 x 3), 16-channel groups
// -- the geometry is paid once per 16 channels, four views' geometry stays live (168 registers), reference features in LDS --
// on ALL of a CU's LDS, one workgroup per CU.
#ifndef D3D_WINDOW_WG3   // experiment: 8-channel form as 4-wave workgroups (one plane sub-range), THREE per CU on a third of the LDS each:
                        // stage 3 0.80 -> 0.86 ms, stage 2 1.20 -> 1.46, stage 1 1.87 -> 2.55; inside a CasMVSNet view the stage-3 windows
                        // no longer fit (0.83 -> 3.4 ms)
#define D3D_WINDOW_WG3 0
#endif
constexpr bool window_big(int MODE, int CH) { return CH != 8 && MODE != MODE_PAIR; }   // the 12-wave form (16-channel groups of the multi-view modes)
constexpr int window_waves(int MODE, int CH) { return window_big(MODE, CH) ? 12 : (D3D_WINDOW_WG3 ? 4 : 8); }
constexpr int window_lds_bytes(int MODE, int CH) { return window_big(MODE, CH) ? 160 * 1024 : (D3D_WINDOW_WG3 ? 53 * 1024 : 80 * 1024); }
#ifndef D3D_WINDOW_DSEG
#define D3D_WINDOW_DSEG 32
#endif
#ifndef D3D_WINDOW_RPW
#define D3D_WINDOW_RPW 0   // window rows per wave and view in one staging batch (0: the built-in choice)
#endif
#ifndef D3D_WINDOW_VB
#define D3D_WINDOW_VB 0    // views whose rows are requested together in one staging batch (0 = 1: view by view)
#endif
constexpr int WDSEG_MAX = D3D_WINDOW_DSEG;   // planes per workgroup segment (upper bound)
constexpr int WTAB = 2 * WDSEG_MAX + 32;   // floats: per-plane depth range of the patch (pmin, pmax) + the views' translations (4 each)
constexpr int WTOFF = 2 * WDSEG_MAX;      // where the translations start

struct WindowArgs {
    int tiles_x, tiles_y, nseg, dseg, ngroups;
    int cap_bytes;   // LDS bytes available for windows
    const float* cl; // channel-last copy of the source maps [view][C / 8][h * w][8] (pack_channel_last_g8) for the gather path, or null
    unsigned long long* stats;   // -DD3D_EXPERIMENTS + D3D_WINDOW_STATS: [0] workgroups [1] staged chunks [2] planes in them [3] gathered planes
                                 // [4] staged bytes / 16 [5] cycles staging (wave 0) [6] cycles sweeping (wave 0) [7] cycles total (wave 0)
};

#ifdef D3D_EXPERIMENTS
#define D3D_WSTAT(i, v) do { if (a.stats && tid == 0) atomicAdd(a.stats + (i), (unsigned long long)(v)); } while (0)
#define D3D_WCLOCK() (a.stats ? clock64() : 0ll)
#else
#define D3D_WSTAT(i, v) do { } while (0)
#define D3D_WCLOCK() 0ll
#endif

// wave-uniform description of one view's staged window (scalar registers)
struct WinView {
    int wx0, wy0;   // source coordinates of cell (0, 0)
    int ww, wh;     // cells per row / rows
    int rowb;       // bytes per row of a quad plane (ww * 16)
    int qb;         // bytes per quad plane (ww * wh * 16)
    int base;       // absolute LDS byte address of quad 0, cell (0, 0)
    int basem;      // base - wy0 * rowb - wx0 * 16: the address of source position (0, 0), so that the plane loop adds floor(v) * rowb
                    // + floor(u) * 16 without subtracting the window origin (two instructions and two scalar registers per view less)
};

// geo_ring (planesweep_tiled.hip) without the torus: same projection, clamps and weights; the cell address is relative to the
// window origin.  Clamped samples lie in the clamped hull, so 0 <= column <= ww - 2 and 0 <= row <= wh - 2.
__device__ __forceinline__ TapL geo_win(const Ray& r, float tx, float ty, float tz, float d, float umax, float vmax, const WinView& W) {
    const float px = __fadd_rn(__fmul_rn(r.rx, d), tx);
    const float py = __fadd_rn(__fmul_rn(r.ry, d), ty);
    const float pz = __fadd_rn(__fmul_rn(r.rz, d), tz);
    const float iz = __builtin_amdgcn_rcpf(pz);
    const float u0 = px * iz, v0 = py * iz;
    float u = fmaf(fmaf(-u0, pz, px), iz, u0);
    float v = fmaf(fmaf(-v0, pz, py), iz, v0);
    u = __builtin_amdgcn_fmed3f(u, -1.0f, umax);
    v = __builtin_amdgcn_fmed3f(v, -1.0f, vmax);
    const float fu = floorf(u), fv = floorf(v);
    const float ax = u - fu, ay = v - fv;
    const float bx = (fu + 1.0f) - u, by = (fv + 1.0f) - v;
    TapL t;
    t.nw = bx * by;
    t.ne = ax * by;
    t.sw = bx * ay;
    t.se = ax * ay;
    t.a0 = __mul24((int)fv, W.rowb) + W.basem + ((int)fu << 4);   // floor(v) >= -1: signed 24-bit product
    t.a1 = t.a0 + W.rowb;
    return t;
}

__device__ __forceinline__ void lds_write4_abs(int byte_addr, const f4& v) {
    *(__attribute__((address_space(3))) f4*)(unsigned)byte_addr = v;
}

// Lane exchanges of the reductions below as DPP operands (one vector instruction each; __shfl_xor is a ds_bpermute round trip
// through the LDS pipe plus its address arithmetic): quad_perm [1,0,3,2] / [2,3,0,1], row_half_mirror, row_mirror.
template <int CTRL>
__device__ __forceinline__ float dpp_get(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, v), __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}
__device__ __forceinline__ float lane_value(float v, int l) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l));
}
// min / max over aligned groups of 8 lanes
__device__ __forceinline__ float grp8_min(float v) {
    v = fminf(v, dpp_get<0xB1>(v)); v = fminf(v, dpp_get<0x4E>(v)); v = fminf(v, dpp_get<0x141>(v));
    return v;
}
__device__ __forceinline__ float grp8_max(float v) {
    v = fmaxf(v, dpp_get<0xB1>(v)); v = fmaxf(v, dpp_get<0x4E>(v)); v = fmaxf(v, dpp_get<0x141>(v));
    return v;
}
// min / max over the wave (uniform result): rows of 16 by DPP, the four rows through scalar registers
__device__ __forceinline__ float wave_min_dpp(float v) {
    v = fminf(grp8_min(v), dpp_get<0x140>(grp8_min(v)));
    return fminf(fminf(lane_value(v, 0), lane_value(v, 16)), fminf(lane_value(v, 32), lane_value(v, 48)));
}
__device__ __forceinline__ float wave_max_dpp(float v) {
    v = fmaxf(grp8_max(v), dpp_get<0x140>(grp8_max(v)));
    return fmaxf(fmaxf(lane_value(v, 0), lane_value(v, 16)), fmaxf(lane_value(v, 32), lane_value(v, 48)));
}

}  // namespace

// PH: patch height (PH / 2 pixel waves; the rest of the workgroup's waves are plane sub-ranges).
template <int MODE, int NSRC, int CH, bool OUTCL, int PH>
__global__ __launch_bounds__(64 * window_waves(MODE, CH), !window_big(MODE, CH) && !D3D_WINDOW_WG3 ? 4 : 3) void sweep_window_kernel(SweepParams p, WindowArgs a) {
    constexpr int WWAVES = window_waves(MODE, CH), WTHREADS = 64 * WWAVES;
    static_assert(MODE == MODE_VARIANCE || MODE == MODE_WEIGHTED || (MODE == MODE_PAIR && NSRC == 1),
                  "window kernel: variance, weighted correlation, and the channel mean of one pair's correlation");
    static_assert(!OUTCL || MODE == MODE_VARIANCE || MODE == MODE_WEIGHTED, "channel-last 16-bit output: the variance volume and the weighted correlation");
    static_assert(NSRC <= 8, "one lane per (corner, view): 8 x NSRC <= 64");
    constexpr int NPW = PH / 2, NSUBW = WWAVES / NPW;   // pixel waves, plane sub-ranges
    constexpr int Q = CH / 4;
    // The weighted correlation leaving as CL8 16-bit cells (round 5: the slice regularisers' fast mode): with the four view weights
    // of a pixel and the four views' rays in registers for the whole kernel this instance does not fit 128 registers (18 spills in
    // the plane loop even with the packed accumulate below; 148 and a 6.7 ms sweep instead of 1.7 ms before it).  The weights live in
    // LDS instead ([pixel][4] floats, 4 KB), read where they are used -- one ds_read_b32 per (quad, view, plane) -- and so do the
    // views' matrices, from which a view's ray is rebuilt per plane: 123 registers, no spill.
    constexpr bool VW_LDS = OUTCL && MODE == MODE_WEIGHTED;
    static_assert(!VW_LDS || NSRC <= 4, "four view weights per LDS cell");
    extern __shared__ __attribute__((aligned(16))) float lds[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = rfl(tid >> 6);
    const int pw = wave % NPW, sub = wave / NPW;
    const long long t_begin = D3D_WCLOCK();
    long long t_stage = 0, t_sweep = 0;
    const int h = p.h, w = p.w, D = p.D;
    const size_t plane = (size_t)h * w;
#if defined(D3D_EXPERIMENTS) && defined(D3D_WX_STAGGER)   // do the two workgroups of a CU run in lockstep?  the second wave of the dispatch starts late
    if (blockIdx.x >= 256 && blockIdx.x < 512)
        for (int i = 0; i < D3D_WX_STAGGER; ++i) __builtin_amdgcn_s_sleep(127);
#endif

    // block -> (patch row, segment, patch column): blocks b, b + 8, ... share an XCD (and its L2); within an XCD's run vertically
    // adjacent patches come first (they share most of their source rows), as in the ring kernel
    int b = blockIdx.x;
    {
        const int nblk = gridDim.x, per = nblk / 8;
        if (nblk % 8 == 0) b = (b % 8) * per + b / 8;
    }
    const int ty = b % a.tiles_y;
    const int seg = (b / a.tiles_y) % a.nseg;
    const int tx = b / (a.tiles_y * a.nseg);
    const int x0 = tx * WTW, y0 = ty * PH;
    const int ds = seg * a.dseg, de = min(ds + a.dseg, D), nplanes = de - ds;
    const int x1c = min(x0 + WTW - 1, w - 1), y1c = min(y0 + PH - 1, h - 1);

    // lane -> pixel: row-major over the wave's two patch rows
    const int px = x0 + (lane & 31), py = y0 + 2 * pw + (lane >> 5);
    const bool valid = px < w && py < h;
    const int pix = valid ? py * w + px : 0;
    const unsigned pixb = (unsigned)pix * 4u;
    // channel-last cells: C * 2 bytes per pixel (h*w*C*2 < 2^32, checked at launch), or 16 in a plane of 8-channel groups (CL8)
    const unsigned pixo = OUTCL ? (p.out_cl == 2 ? (unsigned)pix * 16u : (unsigned)pix * (unsigned)p.C * 2u) : pixb;
    const float xf = (float)px, yf = (float)py;

    // ---- per-lane inputs, requested first: their latency runs under the range / window phases ------------------------------
    Ray ray[NSRC];
    float T0[NSRC], T1[NSRC], T2[NSRC];
#pragma unroll
    for (int i = 0; i < NSRC; ++i) {
        const float* __restrict__ M = p.proj34 + 12 * min(i, p.n_src - 1);
        ray[i] = make_ray(M, xf, yf);
        T0[i] = M[3]; T1[i] = M[7]; T2[i] = M[11];
        if (i >= p.n_src) {   // unused view of the template: every sample at (-1, -1), a zero cell (window (-1, -1, 2, 2) below)
            ray[i].rx = 0.0f; ray[i].ry = 0.0f; ray[i].rz = 0.0f;
            T0[i] = -1.0f; T1[i] = -1.0f; T2[i] = 1.0f;
        }
    }
#ifndef D3D_WINDOW_T_SGPR
    // The translations live in LDS and are read back per view and plane (one broadcast ds_read_b128): as scalar registers they
    // were the ones spilled -- 48 v_readlane in a plane loop of 428 instructions.
    if (tid < NSRC) {
        f4 tv = {0, 0, 0, 0};
#pragma unroll
        for (int i = 0; i < NSRC; ++i) if (tid == i) tv = (f4){T0[i], T1[i], T2[i], 0.0f};
        *reinterpret_cast<f4*>(lds + WTOFF + 4 * tid) = tv;
    }
    const int tadr = lds_base_bytes(lds) + WTOFF * 4;
#endif
    float vw[NSRC];
    float rden = 0.0f;
    if (MODE == MODE_WEIGHTED) {
        float den = 1e-5f;
#pragma unroll
        for (int i = 0; i < NSRC; ++i) {
            const float t = p.weights[(size_t)min(i, p.n_src - 1) * plane + pix];
            vw[i] = (valid && i < p.n_src) ? t : 0.0f;
            den += vw[i];
        }
        rden = 1.0f / den;
        if constexpr (VW_LDS) {
            if (sub == 0) {
                f4 wv = {0, 0, 0, 0};
#pragma unroll
                for (int i = 0; i < NSRC; ++i) wv[i] = vw[i];
                lds_write4_abs(lds_base_bytes(lds) + WTAB * 4 + (pw * 64 + lane) * 16, wv);   // (read behind the range table's barrier)
            }
            // ... and so do the views' 3 x 4 matrices (rows [r0 r1 r2 t]; an unused view: every sample at (-1, -1)): the lane's
            // ray of a view is rebuilt from them per plane (make_ray's own expression) instead of living in three registers per view
            if (tid < 3 * NSRC) {
                const int i = tid / 3, row = tid - 3 * i;
                f4 m = {0.0f, 0.0f, 0.0f, row == 2 ? 1.0f : -1.0f};
                if (i < p.n_src) { const float* __restrict__ M = p.proj34 + 12 * i + 4 * row; m = (f4){M[0], M[1], M[2], M[3]}; }
                lds_write4_abs(lds_base_bytes(lds) + WTAB * 4 + WTW * PH * 16 + tid * 16, m);
            }
        }
    }
    float aff_lo = 1.0f, aff_step = 0.0f;
    if (p.depth_mode == D3D_DEPTH_AFFINE && valid) {
        aff_lo = p.depth[pix];
        aff_step = p.depth[plane + pix];
    }
    // corner lanes: lane = 8 * view + corner (x end, y end, depth end)
    const int cview = min(lane >> 3, NSRC - 1), ck = lane & 7;
    Ray cray;
    float cT0, cT1, cT2;
    {
        const float* __restrict__ M = p.proj34 + 12 * min(cview, p.n_src - 1);
        cray = make_ray(M, (ck & 1) ? (float)x1c : (float)x0, (ck & 2) ? (float)y1c : (float)y0);
        cT0 = M[3]; cT1 = M[7]; cT2 = M[11];
    }

    // ---- per-plane depth range of the patch (pmin, pmax) -> LDS table -------------------------------------------------------
    constexpr int PPL = (WTW * PH) / 64;   // patch pixels per lane when one wave covers the patch
    if (p.depth_mode == D3D_DEPTH_PER_PLANE) {
        for (int i = tid; i < nplanes; i += WTHREADS) {
            const float dv = p.depth[ds + i];
            lds[i] = dv;
            lds[WDSEG_MAX + i] = dv;
        }
    } else if (p.depth_mode == D3D_DEPTH_AFFINE) {
        float blo[PPL], bst[PPL];
        bool bok[PPL];
#pragma unroll
        for (int k = 0; k < PPL; ++k) {
            const int q = lane + 64 * k;
            const int qx = x0 + (q & 31), qy = y0 + (q >> 5);
            bok[k] = qx < w && qy < h;
            const size_t qi = bok[k] ? (size_t)qy * w + qx : 0;
            blo[k] = p.depth[qi];
            bst[k] = p.depth[plane + qi];
        }
        for (int i = wave; i < nplanes; i += WWAVES) {
            float lo = INFINITY, hi = -INFINITY;
#pragma unroll
            for (int k = 0; k < PPL; ++k) {
                const float dv = __fadd_rn(blo[k], __fmul_rn((float)(ds + i), bst[k]));
                if (bok[k]) { lo = fminf(lo, dv); hi = fmaxf(hi, dv); }
            }
            lo = wave_min_dpp(lo);
            hi = wave_max_dpp(hi);
            if (lane == 0) { lds[i] = lo; lds[WDSEG_MAX + i] = hi; }
        }
    } else {
        for (int i = wave; i < nplanes; i += WWAVES) {
            float lo = INFINITY, hi = -INFINITY;
#pragma unroll
            for (int k = 0; k < PPL; ++k) {
                const int q = lane + 64 * k;
                const int qx = x0 + (q & 31), qy = y0 + (q >> 5);
                if (qx < w && qy < h) {
                    const float dv = p.depth[(size_t)(ds + i) * plane + (size_t)qy * w + qx];
                    lo = fminf(lo, dv);
                    hi = fmaxf(hi, dv);
                }
            }
            lo = wave_min_dpp(lo);
            hi = wave_max_dpp(hi);
            if (lane == 0) { lds[i] = lo; lds[WDSEG_MAX + i] = hi; }
        }
    }
    __syncthreads();

    // 16-channel groups: the patch's REFERENCE features live in LDS too ([quad][pixel] 16-byte cells, a lane reads its own cell
    // once per plane and quad) -- sixteen registers per lane the 128-register budget of two workgroups per CU does not have
    constexpr bool REF_LDS = Q > 2 && MODE != MODE_PAIR;   // (a pair sweep has one view's geometry live: room for the reference in registers)
    constexpr int NPIX = WTW * PH;
    constexpr int REF_BYTES = REF_LDS ? CH * NPIX * 4 : 0;
    constexpr int VW_BYTES = VW_LDS ? WTW * PH * 16 + 3 * NSRC * 16 : 0;   // view weights per pixel + the views' matrices
    const int my_mat = lds_base_bytes(lds) + WTAB * 4 + WTW * PH * 16;
    const int my_vw = lds_base_bytes(lds) + WTAB * 4 + (pw * 64 + lane) * 16;
    const int ref0 = lds_base_bytes(lds) + WTAB * 4 + VW_BYTES;
    const int my_ref = ref0 + (pw * 64 + lane) * 16;
    const int lds0 = ref0 + REF_BYTES;
    const int cap_bytes = a.cap_bytes - REF_BYTES - VW_BYTES;
    const float umax = (float)w, vmax = (float)h;
    const float invV = 1.0f / (float)(p.n_src + 1);
    const size_t cstride_b = (p.plane_major ? plane : (size_t)D * plane) * 4;   // bytes between channels

    // windows of the planes [c0p, c0p + n) of the segment; returns false when they cannot be bounded or do not fit
    WinView W[NSRC];
    auto windows = [&](int c0p, int n) -> bool {
        float lo = INFINITY, hi = -INFINITY;
        for (int i = c0p; i < c0p + n; ++i) {
            lo = fminf(lo, lds[i]);
            hi = fmaxf(hi, lds[WDSEG_MAX + i]);
        }
        const float dv = (ck & 4) ? hi : lo;
        const float qx = __fadd_rn(__fmul_rn(cray.rx, dv), cT0);
        const float qy = __fadd_rn(__fmul_rn(cray.ry, dv), cT1);
        const float qz = __fadd_rn(__fmul_rn(cray.rz, dv), cT2);
        bool ok = (qz > 1e-20f) && (qz < 1e30f);
        const float iz = 1.0f / qz;
        float u = qx * iz, v = qy * iz;
        ok = ok && (fabsf(u) < 1e30f) && (fabsf(v) < 1e30f);
        u = fminf(fmaxf(u, -1.0f), (float)w);   // the same clamp as the samples' (geo_win)
        v = fminf(fmaxf(v, -1.0f), (float)h);
        const float umn = grp8_min(u), umx = grp8_max(u), vmn = grp8_min(v), vmx = grp8_max(v);
        const float okf = grp8_min(ok ? 1.0f : 0.0f);
        // interior samples differ from the corner hull by fp32 rounding only (<< 1/16 px)
        const int wx0 = max((int)floorf(umn - 0.0625f), -1);
        const int wy0 = max((int)floorf(vmn - 0.0625f), -1);
        const int wx1 = min((int)floorf(umx + 0.0625f) + 1, w + 1);
        const int wy1 = min((int)floorf(vmx + 0.0625f) + 1, h + 1);
        bool good = true;
        int off = 0;
#pragma unroll
        for (int i = 0; i < NSRC; ++i) {
            const int src = 8 * i;
            W[i].wx0 = __builtin_amdgcn_readlane(wx0, src);
            W[i].wy0 = __builtin_amdgcn_readlane(wy0, src);
            W[i].ww = max(__builtin_amdgcn_readlane(wx1, src) - W[i].wx0 + 1, 2);
            W[i].wh = max(__builtin_amdgcn_readlane(wy1, src) - W[i].wy0 + 1, 2);
            good = good && (__builtin_amdgcn_readlane(__float_as_int(okf), src) != 0);
            if (i >= p.n_src) { W[i].wx0 = -1; W[i].wy0 = -1; W[i].ww = 2; W[i].wh = 2; }
            W[i].rowb = W[i].ww * 16;
            W[i].qb = W[i].rowb * W[i].wh;
            W[i].base = lds0 + off;
            W[i].basem = W[i].base - W[i].wy0 * W[i].rowb - W[i].wx0 * 16;
            good = good && W[i].ww < 1024 && W[i].wh < 1024;
            off += good ? W[i].qb * Q : 0;
            good = good && off <= cap_bytes;
        }
        return good;
    };

    f4 r[REF_LDS ? 1 : Q];   // reference features of the group (8-channel groups: registers)
    unsigned long long even_quad = 0;
    auto finalize_store = [&](const f4& s, const f4& qq, unsigned long long& ob, int q) {
        f4 o;
        if (MODE == MODE_VARIANCE) {
            const f2 iv = {invV, invV};
            const f2 ml = lo2(s) * iv, mh = hi2(s) * iv;
            o = cat2(pk_fma(lo2(qq), iv, -(ml * ml)), pk_fma(hi2(qq), iv, -(mh * mh)));
        } else {
            o = s * rden;
        }
        if constexpr (OUTCL && MODE == MODE_WEIGHTED) {
            store_sbase_h16x4(ob, pixo, pack_h16x4(o));   // (quad by quad: sweep_device.h)
            ob += 8;
        } else if constexpr (OUTCL) {
            if ((q & 1) == 0) {
                even_quad = pack_h16x4(o);
            } else {
#ifdef D3D_CL_PARTIAL_DEFAULT_POLICY   // experiment: see store_sbase_h16x8
                if (a.ngroups > 1) store_sbase_h16x8<true>(ob, pixo, even_quad, pack_h16x4(o));
                else
#endif
#ifdef D3D_X_SWEEP_NOSTORE   // timing-only build (wrong results): the channel-last volume is computed and NOT written -- what a sweep whose
                             // output stayed in the CU would cost (profiles/r05_sweep_conv0_fusion_bound.txt); reported by d3d_build_flags()
                {
                    const unsigned long long hq = pack_h16x4(o);
                    if (a.cap_bytes < 0) store_sbase_h16x8(ob, pixo, even_quad, hq);   // (a uniform test that never holds: the values stay live)
                    else asm volatile("" : : "v"(hq), "v"(even_quad));
                }
#else
                store_sbase_h16x8(ob, pixo, even_quad, pack_h16x4(o));
#endif
                ob += 16;
            }
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                store_sbase(ob, pixb, o[k]);
                ob += cstride_b;
            }
        }
    };
    auto accumulate = [&](f4& s, f4& qq, const f4& val, const f4& rq, int i) {
        if (MODE == MODE_VARIANCE) {
            const f2 vl = lo2(val), vh = hi2(val);
            s = cat2(lo2(s) + vl, hi2(s) + vh);
            qq = cat2(pk_fma(vl, vl, lo2(qq)), pk_fma(vh, vh, hi2(qq)));
        } else {
            float wv;
            if constexpr (VW_LDS) wv = *(volatile __attribute__((address_space(3))) float*)(unsigned)(my_vw + 4 * i);
            else wv = vw[i];
            // per channel: the product val * ref rounded, then one fused multiply-add with the view weight -- two channels per
            // instruction (v_pk_mul_f32, v_pk_fma_f32: the scalar form's results bit for bit, half its instructions; written as four
            // scalar chains the channel-last instance spilled 148 registers)
            const f2 w2 = {wv, wv};
            s = cat2(pk_fma(lo2(val) * lo2(rq), w2, lo2(s)), pk_fma(hi2(val) * hi2(rq), w2, hi2(s)));
        }
    };

    // ---- chunks of planes: as many of the planes that are left as have windows that fit; per chunk every channel group ------
    // (the windows depend on the planes only: computed once per chunk, not once per chunk and group -- 23 -> 6 searches per
    //  workgroup at stage 1)
    int guess = nplanes;   // planes to try first for the next chunk
    int done = 0;
    while (done < nplanes) {
        // first try: what fitted last time plus a quarter (a sweep with wide plane spacing settles at a few planes per chunk:
        // searching down from "everything that is left" for every chunk cost seven window computations per chunk at stage 1)
        const int first = min(nplanes - done, guess);
        int n = first;
        bool fit = windows(done, n);
        while (!fit && n > NSUBW) {   // three quarters of the planes, in whole rounds of the sub-ranges
            n = max((n * 3 / 4) / NSUBW * NSUBW, NSUBW);
            fit = windows(done, n);
        }
        for (int gi = 0; gi < a.ngroups; ++gi) {
            const int c0 = gi * CH;
            if constexpr (REF_LDS) {
                for (int idx = tid; idx < Q * NPIX; idx += WTHREADS) {   // (read after the chunk's "windows staged" barrier)
                    const int q = idx / NPIX, pp = idx - q * NPIX;
                    const int qx = x0 + (pp & 31), qy = y0 + (pp >> 5);
                    const bool in = qx < w && qy < h;
                    const float* __restrict__ sq = p.feats[0] + (size_t)(c0 + 4 * q) * plane + (in ? (size_t)qy * w + qx : 0);
                    f4 v;
                    v[0] = sq[0]; v[1] = sq[plane]; v[2] = sq[2 * plane]; v[3] = sq[3 * plane];
                    if (!in) v = (f4){0, 0, 0, 0};
                    lds_write4_abs(ref0 + idx * 16, v);
                }
            } else {
#pragma unroll
                for (int q = 0; q < Q; ++q)
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const float t = p.feats[0][(size_t)(c0 + 4 * q + k) * plane + pix];
                        r[REF_LDS ? 0 : q][k] = valid ? t : 0.0f;
                    }
                // (waited for below, behind the staging loads: ref_wait)
            }
            asm volatile("" : "+v"(aff_lo), "+v"(aff_step), "+v"(rden));
            if (MODE == MODE_WEIGHTED && !VW_LDS) {
#pragma unroll
                for (int i = 0; i < NSRC; ++i) asm volatile("" : "+v"(vw[i]));
            }
            // one plane of the segment for this wave's pixels
            auto sweep_plane = [&](const int dl_, const bool staged) {
                const int d = ds + dl_;
                float dv;
                if (p.depth_mode == D3D_DEPTH_PER_PIXEL) {
                    const float t = p.depth[(size_t)d * plane + pix];
                    dv = valid ? t : 1.0f;
                } else if (p.depth_mode == D3D_DEPTH_AFFINE) {
                    dv = __fadd_rn(aff_lo, __fmul_rn((float)d, aff_step));
                } else {
                    dv = lds[dl_];
                }
                unsigned long long ob = OUTCL ? uniform64(reinterpret_cast<unsigned short*>(p.out) +
                                                          (p.out_cl == 2 ? ((size_t)d * (p.C / 8) + c0 / 8) * plane * 8 : (size_t)d * plane * p.C + c0))
                                              : uniform64(p.out + (p.plane_major ? (size_t)d * p.C + c0 : (size_t)c0 * D + d) * plane);
                if (!valid) return;
                if constexpr (MODE == MODE_PAIR) {
                    // adamvs.py:469-474: mean over the channels of reference x warped source, all C channels in one pass (quads in
                    // order, channels in order: the ring kernel's sum)
                    float pair_acc = 0.0f;
                    if (staged) {
#ifndef D3D_WINDOW_T_SGPR
                        const f4 tv = *(volatile lds_f4_ptr)(unsigned)tadr;
                        const TapL g = geo_win(ray[0], tv[0], tv[1], tv[2], dv, umax, vmax, W[0]);
#else
                        const TapL g = geo_win(ray[0], T0[0], T1[0], T2[0], dv, umax, vmax, W[0]);
#endif
                        f4 tp[2][4];
                        auto request = [&](int q2, f4 (&dst)[4]) {
                            const int n = g.a0 + q2 * W[0].qb, s_ = g.a1 + q2 * W[0].qb;
                            dst[0] = lds_read4_abs(n);
                            dst[1] = lds_read4_abs(n + 16);
                            dst[2] = lds_read4_abs(s_);
                            dst[3] = lds_read4_abs(s_ + 16);
                        };
                        request(0, tp[0]);
#pragma unroll
                        for (int q = 0; q < Q; ++q) {
                            if (q + 1 < Q) request(q + 1, tp[(q + 1) & 1]);
                            f4 (&c)[4] = tp[q & 1];
                            asm volatile("" : "+v"(c[3]));
                            const f4 val = blend(c[0], c[1], c[2], c[3], g.nw, g.ne, g.sw, g.se);
#pragma unroll
                            for (int k = 0; k < 4; ++k) pair_acc = fmaf(r[REF_LDS ? 0 : q][k], val[k], pair_acc);
                        }
                    } else {
                        float u, v;
                        project(ray[0], T0[0], T1[0], T2[0], dv, h, w, u, v);
                        const TapG t = make_tap_glb(u, v, h, w);
                        const float* __restrict__ gsrc = p.feats[1] + t.off;
#pragma unroll
                        for (int q = 0; q < Q; ++q)
#pragma unroll
                            for (int kk = 0; kk < 4; ++kk) {
                                const float* __restrict__ gk = gsrc + (size_t)(4 * q + kk) * plane;
                                const float val = fmaf(gk[t.dyw + t.dx], t.se, fmaf(gk[t.dyw], t.sw, fmaf(gk[t.dx], t.ne, gk[0] * t.nw)));
                                pair_acc = fmaf(r[REF_LDS ? 0 : q][kk], val, pair_acc);
                            }
                    }
                    store_sbase(uniform64(p.out + (size_t)d * plane), pixb, pair_acc / (float)CH);
                    return;
                }
                if constexpr (Q > 2) { if (staged) {
                    // Unit-major (16-channel groups): units u = (quad q, view i) in q-major order, one quad's accumulators live, the
                    // four views' geometry kept for the whole plane (the ring kernel's order)
                    TapL t[NSRC];
#pragma unroll
                    for (int i = 0; i < NSRC; ++i) t[i] = geo_win(ray[i], T0[i], T1[i], T2[i], dv, umax, vmax, W[i]);
                    constexpr int NU = Q * NSRC;
                    f4 tp[2][4];
                    auto request = [&](int u, f4 (&dst)[4]) {
                        const int q2 = u / NSRC, i2 = u % NSRC;
                        const int n = t[i2].a0 + q2 * W[i2].qb, s_ = t[i2].a1 + q2 * W[i2].qb;
                        dst[0] = lds_read4_abs(n);
                        dst[1] = lds_read4_abs(n + 16);
                        dst[2] = lds_read4_abs(s_);
                        dst[3] = lds_read4_abs(s_ + 16);
                    };
                    f4 rq = lds_read4_abs(my_ref);
                    request(0, tp[0]);
                    f4 s, qq;
#pragma unroll
                    for (int u = 0; u < NU; ++u) {
                        const int q = u / NSRC, i = u % NSRC;
                        if (u + 1 < NU) request(u + 1, tp[(u + 1) & 1]);
                        if (i == 0) {
                            if (MODE == MODE_VARIANCE) { s = rq; qq = s * s; }
                            else { s = (f4){0, 0, 0, 0}; qq = s; }
                        }
                        f4 (&c)[4] = tp[u & 1];
                        asm volatile("" : "+v"(c[3]));   // one wait per unit (LDS returns in order)
                        const f4 val = blend(c[0], c[1], c[2], c[3], t[i].nw, t[i].ne, t[i].sw, t[i].se);
                        accumulate(s, qq, val, rq, i);
                        if (i == NSRC - 1 && q + 1 < Q) rq = lds_read4_abs(my_ref + (q + 1) * (NPIX * 16));   // (weighted mode reads rq until here)
                        if (i == NSRC - 1) finalize_store(s, qq, ob, q);
                    }
                } }
                if constexpr (Q <= 2) { if (staged) {
                    // View-major (8-channel groups): ONE view's geometry is live at a time (the next view's is computed while this view's taps are in
                    // flight) and the accumulators of the group's quads stay in registers until the last view -- the unit-major
                    // order of the ring kernel keeps 4 x (2 addresses + 4 weights) alive and does not fit 128 registers.  Per channel
                    // the views are still added in order 0, 1, ...: the same sums.
                    f4 s[Q], qq[Q];
#pragma unroll
                    for (int q = 0; q < Q; ++q) {
                        if (MODE == MODE_VARIANCE) { s[q] = r[REF_LDS ? 0 : q]; qq[q] = s[q] * s[q]; }
                        else { s[q] = (f4){0, 0, 0, 0}; qq[q] = s[q]; }
                    }
                    f4 tp[Q][4];
                    auto request = [&](const TapL& g, int i2, int q2) {
                        const int n = g.a0 + q2 * W[i2].qb, s_ = g.a1 + q2 * W[i2].qb;
                        tp[q2][0] = lds_read4_abs(n);
                        tp[q2][1] = lds_read4_abs(n + 16);
                        tp[q2][2] = lds_read4_abs(s_);
                        tp[q2][3] = lds_read4_abs(s_ + 16);
                    };
#ifndef D3D_WINDOW_T_SGPR
                    auto geo_view = [&](int i) {
                        if constexpr (VW_LDS) {   // (matrix rows from LDS, broadcast reads; the ray as make_ray computes it)
                            const f4 m0 = *(volatile lds_f4_ptr)(unsigned)(my_mat + 48 * i), m1 = *(volatile lds_f4_ptr)(unsigned)(my_mat + 48 * i + 16),
                                     m2 = *(volatile lds_f4_ptr)(unsigned)(my_mat + 48 * i + 32);
                            Ray rr;
                            rr.rx = fmaf(m0[0], xf, fmaf(m0[1], yf, m0[2]));
                            rr.ry = fmaf(m1[0], xf, fmaf(m1[1], yf, m1[2]));
                            rr.rz = fmaf(m2[0], xf, fmaf(m2[1], yf, m2[2]));
                            return geo_win(rr, m0[3], m1[3], m2[3], dv, umax, vmax, W[i]);
                        }
                        const f4 tv = *(volatile lds_f4_ptr)(unsigned)(tadr + 16 * i);   // (volatile: not hoisted out of the plane loop)
                        return geo_win(ray[i], tv[0], tv[1], tv[2], dv, umax, vmax, W[i]);
                    };
#else
                    auto geo_view = [&](int i) { return geo_win(ray[i], T0[i], T1[i], T2[i], dv, umax, vmax, W[i]); };
#endif
                    TapL gc = geo_view(0);
#pragma unroll
                    for (int q = 0; q < Q; ++q) request(gc, 0, q);
#pragma unroll
                    for (int i = 0; i < NSRC; ++i) {
                        __builtin_amdgcn_sched_barrier(0);   // (keeps the scheduler from hoisting every view's geometry to the top)
                        TapL gn = gc;
                        if (i + 1 < NSRC) gn = geo_view(i + 1);
#pragma unroll
                        for (int q = 0; q < Q; ++q) {
                            asm volatile("" : "+v"(tp[q][3]));   // one wait per quad (LDS returns in order)
                            const f4 val = blend(tp[q][0], tp[q][1], tp[q][2], tp[q][3], gc.nw, gc.ne, gc.sw, gc.se);
                            if (i + 1 < NSRC) request(gn, i + 1, q);   // the next view's taps of this quad, into the registers just read
                            accumulate(s[q], qq[q], val, r[REF_LDS ? 0 : q], i);
                        }
                        gc = gn;
                    }
#pragma unroll
                    for (int q = 0; q < Q; ++q) finalize_store(s[q], qq[q], ob, q);
                } }
                if (!staged) {   // taps from the planar maps in global memory (the ring kernel's fallback arithmetic), one view at a time
                    f4 s[Q], qq[Q], rf[Q];
#pragma unroll
                    for (int q = 0; q < Q; ++q) {
                        if constexpr (REF_LDS) {   // (no barrier on this path: the reference features straight from global memory)
#pragma unroll
                            for (int k = 0; k < 4; ++k) rf[q][k] = p.feats[0][(size_t)(c0 + 4 * q + k) * plane + pix];
                        } else {
                            rf[q] = r[REF_LDS ? 0 : q];
                        }
                        if (MODE == MODE_VARIANCE) { s[q] = rf[q]; qq[q] = s[q] * s[q]; }
                        else { s[q] = (f4){0, 0, 0, 0}; qq[q] = s[q]; }
                    }
                    for (int i = 0; i < p.n_src; ++i) {
                        const float* __restrict__ M = p.proj34 + 12 * i;
                        const Ray rr = make_ray(M, xf, yf);
                        float u, v;
                        project(rr, M[3], M[7], M[11], dv, h, w, u, v);
                        const TapG t = make_tap_glb(u, v, h, w);
                        const float vwi = (MODE == MODE_WEIGHTED) ? p.weights[(size_t)i * plane + pix] : 0.0f;
                        const float* __restrict__ g = p.feats[i + 1] + (size_t)c0 * plane + t.off;
                        // with the channel-last copy (hypothesis volumes: the caller did not vouch for a smooth map) a tap's eight channels
                        // are 32 contiguous bytes -- two 16-byte loads instead of eight 4-byte loads in eight planes; same values
                        const f4* __restrict__ gc = CH == 8 && a.cl ? reinterpret_cast<const f4*>(a.cl + (((size_t)i * a.ngroups + gi) * plane + t.off) * 8) : nullptr;
#pragma unroll
                        for (int q = 0; q < Q; ++q) {
                            f4 val;
                            if (CH == 8 && gc) {
                                const f4 nw = gc[q], ne = gc[2 * t.dx + q], sw = gc[2 * t.dyw + q], se = gc[2 * (t.dyw + t.dx) + q];
#pragma unroll
                                for (int kk = 0; kk < 4; ++kk) val[kk] = fmaf(se[kk], t.se, fmaf(sw[kk], t.sw, fmaf(ne[kk], t.ne, nw[kk] * t.nw)));
                            } else
#pragma unroll
                            for (int kk = 0; kk < 4; ++kk) {
                                const float* __restrict__ gk = g + (size_t)(4 * q + kk) * plane;
                                val[kk] = fmaf(gk[t.dyw + t.dx], t.se, fmaf(gk[t.dyw], t.sw, fmaf(gk[t.dx], t.ne, gk[0] * t.nw)));
                            }
                            if (MODE == MODE_VARIANCE) {
                                accumulate(s[q], qq[q], val, rf[q], 0);
                            } else {
#pragma unroll
                                for (int k = 0; k < 4; ++k) s[q][k] = fmaf(val[k] * rf[q][k], vwi, s[q][k]);
                            }
                        }
                    }
#pragma unroll
                    for (int q = 0; q < Q; ++q) finalize_store(s[q], qq[q], ob, q);
                }
            };

                if (fit) {
                    const long long ts0 = D3D_WCLOCK();
                    D3D_WSTAT(1, 1); D3D_WSTAT(2, n);
                    // Stage.  A wave takes whole rows of a view's window -- every quad of the row: the row's scalar arithmetic (range
                    // test, byte offset) is paid once for the group's channels -- the lane is the column.  Loads are buffer loads -- the
                    // view's descriptor and the row's byte offset in scalar registers, the column as a 32-bit lane offset, no 64-bit
                    // address registers -- and a lane or row outside the image gets an offset beyond the buffer, for which the hardware
                    // returns zeros: no clamps, no selects.  The loads of RPW rows (4 x Q x RPW per lane) are issued before the first
                    // 16-byte LDS write; the other workgroup of the CU sweeps meanwhile.
                    // Views per batch (VB): round 5 measured requesting the rows of SEVERAL views' windows before the first LDS write of a
                    // batch (one memory round trip per batch instead of one per view).  The extra descriptors / offsets in flight push the
                    // kernel over its 128 registers (80 - 180 bytes of scratch per lane, reloaded in the plane loop): stage 1 1.68 -> 2.28 ms
                    // with all four views, 1.94 with two; and halving the rows per batch (twice the round trips) costs only 5 % -- the staging
                    // is not a latency chain.  View by view stays (profiles/r05_window_staging.txt).
                    constexpr int VB = D3D_WINDOW_VB > 0 ? (D3D_WINDOW_VB < NSRC ? D3D_WINDOW_VB : NSRC) : 1;
                    constexpr int RPW = D3D_WINDOW_RPW > 0 ? D3D_WINDOW_RPW : (Q > 2 ? 1 : (VB > 2 ? 1 : (PH > 8 ? 3 : 2)));
                    constexpr int OOB = 0x7ffffff0;   // (launch check: C * h * w * 4 < 2^31, so any offset from here on is out of range)
                    const int fbytes = p.C * (int)plane * 4;
#pragma unroll
                    for (int i = 0; i < NSRC; ++i) {
                        if (i >= p.n_src)   // the zero cells of an unused view
                            for (int idx = tid; idx < 4 * Q; idx += WTHREADS) lds_write4_abs(W[i].base + idx * 16, (f4){0, 0, 0, 0});
                    }
#pragma unroll
                    for (int i0 = 0; i0 < NSRC; i0 += VB) {
                        int maxw = 0, maxh = 0;
#pragma unroll
                        for (int i = i0; i < i0 + VB && i < NSRC; ++i) {
                            if (i < p.n_src) {
                                maxw = max(maxw, W[i].ww);
                                maxh = max(maxh, W[i].wh);
                            }
                        }
#if defined(D3D_EXPERIMENTS) && defined(D3D_WX_NOSTAGE)   // timing only (results wrong): the sweep without its staging
                        if (p.n_src < 100) maxw = 0;
#endif
                        for (int col0 = 0; col0 < maxw; col0 += 64) {   // (windows wider than 64 cells: rare)
                            const int col = col0 + lane;
                            for (int row0 = wave; row0 < maxh; row0 += RPW * WWAVES) {
                                f4 v[VB][RPW][Q];
#pragma unroll
                                for (int i = i0; i < i0 + VB && i < NSRC; ++i) {
                                    if (i >= p.n_src) continue;
                                    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.feats[i + 1]), 0, fbytes, 0x00020000);
                                    const int sx = W[i].wx0 + col;
                                    const int voff = (col < W[i].ww && sx >= 0 && sx < w) ? sx * 4 : OOB;
#pragma unroll
                                    for (int r = 0; r < RPW; ++r) {
                                        const int sy = W[i].wy0 + min(row0 + r * WWAVES, W[i].wh - 1);
                                        const bool yin = sy >= 0 && sy < h;
                                        // rows outside the image: the out-of-range marker travels in the VECTOR offset, which the hardware
                                        // range-checks against num_records (zeros come back); the scalar offset is documented as excluded
                                        // from that check, so it always stays inside the buffer
                                        const int roff = yin ? (c0 * (int)plane + sy * w) * 4 : 0;
                                        const int pb = yin ? (int)plane * 4 : 0;
                                        const int vo = yin ? voff : OOB;
#pragma unroll
                                        for (int q = 0; q < Q; ++q)
#pragma unroll
                                            for (int c = 0; c < 4; ++c)
                                                v[i - i0][r][q][c] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, vo, roff + (4 * q + c) * pb, 0));
                                    }
                                }
#pragma unroll
                                for (int i = i0; i < i0 + VB && i < NSRC; ++i) {
                                    if (i >= p.n_src) continue;
                                    const int cadr = W[i].base + col * 16;
#pragma unroll
                                    for (int r = 0; r < RPW; ++r) {
                                        const int row = row0 + r * WWAVES;
                                        if (row < W[i].wh && col < W[i].ww) {
#pragma unroll
                                            for (int q = 0; q < Q; ++q) lds_write4_abs(cadr + q * W[i].qb + row * W[i].rowb, v[i - i0][r][q]);
                                        }
                                    }
                                }
                                // every load of the batch is consumed here, written or not: a load left pending would be waited for
                                // where its register is next overwritten -- inside the plane loop, on the counter that also counts the
                                // planes' stores
#pragma unroll
                                for (int i = i0; i < i0 + VB && i < NSRC; ++i) {
                                    if (i >= p.n_src) continue;
#pragma unroll
                                    for (int r = 0; r < RPW; ++r)
#pragma unroll
                                        for (int q = 0; q < Q; ++q) asm volatile("" : : "v"(v[i - i0][r][q]));
                                }
                            }
                        }
                    }
                    // The reference loads are waited for HERE, once per chunk and group, with the staging loads in front of them: a
                    // wait left to the first use would sit inside the plane loop, where the counter it waits on also counts the
                    // previous plane's stores -- every plane would wait for its predecessor's stores.
                    if constexpr (!REF_LDS) {
#pragma unroll
                        for (int q = 0; q < Q; ++q) asm volatile("" : "+v"(r[q]));
                    }
                    __syncthreads();   // windows staged
                    const long long ts1 = D3D_WCLOCK();
#if defined(D3D_EXPERIMENTS) && defined(D3D_WX_NOSWEEP)   // timing only (results wrong): the staging without the sweep
                    if (p.n_src > 100)
#endif
                    for (int j = sub; j < n; j += NSUBW) sweep_plane(done + j, true);
                    __syncthreads();   // windows free (next chunk / next channel group)
                    t_stage += ts1 - ts0;
                    t_sweep += D3D_WCLOCK() - ts1;
#ifdef D3D_EXPERIMENTS
                    { int cells = 0;
#pragma unroll
                      for (int i = 0; i < NSRC; ++i) cells += W[i].qb / 16 * Q;
                      D3D_WSTAT(4, cells); }
#endif
                } else {
                    D3D_WSTAT(3, n);
                    if constexpr (!REF_LDS) {
#pragma unroll
                        for (int q = 0; q < Q; ++q) asm volatile("" : "+v"(r[q]));
                    }
                    for (int j = sub; j < n; j += NSUBW) sweep_plane(done + j, false);
                }
        }
        done += n;
        if (!fit) guess = NSUBW;
        else if (!(n == first && first < guess))   // (a chunk cut short by the end of the segment says nothing about what fits)
            guess = max((n + n / 4 + NSUBW - 1) / NSUBW * NSUBW, n + NSUBW);
    }
    D3D_WSTAT(0, 1); D3D_WSTAT(5, t_stage); D3D_WSTAT(6, t_sweep); D3D_WSTAT(7, D3D_WCLOCK() - t_begin);
    (void)t_begin; (void)t_stage; (void)t_sweep;
}

template <int MODE, int NSRC, int CH, bool OUTCL, int PH>
static int launch_window_one(const SweepParams& p, hipStream_t stream) {
    auto kern = sweep_window_kernel<MODE, NSRC, CH, OUTCL, PH>;
    constexpr int WLDS_BYTES = window_lds_bytes(MODE, CH);
    if (OUTCL && (size_t)p.h * p.w * p.C * 2 >= ((size_t)1 << 32)) return D3D_ERR_UNSUPPORTED;
    if ((size_t)p.h * p.w * p.C * 4 >= 0x7ffffff0u) return D3D_ERR_UNSUPPORTED;   // buffer-load offsets of the staging (see OOB)
    int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(kern), WLDS_BYTES);
    if (rc != D3D_OK) return rc;
    WindowArgs a;
    a.ngroups = p.C / CH;
    a.tiles_x = ceil_div(p.w, WTW);
    a.tiles_y = ceil_div(p.h, PH);
    // plane segments: at most WDSEG_MAX planes, and short enough that the launch has ~ten rounds of workgroups (two per CU) -- but
    // not below 8 planes: a segment pays its own prologue and restages windows its neighbour segment has in LDS
    a.nseg = ceil_div(p.D, WDSEG_MAX);
    {
        const long tiles = (long)a.tiles_x * a.tiles_y;
#ifndef D3D_WINDOW_ROUNDS
#define D3D_WINDOW_ROUNDS 0
#endif
        const int want = (int)min((long)ceil_div(p.D, 8), (long)ceil_div((long)D3D_WINDOW_ROUNDS * 512, tiles));
        a.nseg = max(a.nseg, want);
    }
    a.dseg = ceil_div(p.D, a.nseg);
    a.nseg = ceil_div(p.D, a.dseg);
    a.cap_bytes = WLDS_BYTES - WTAB * 4;
    a.stats = nullptr;
    a.cl = nullptr;
    if (MODE != MODE_PAIR && CH == 8 && p.depth_mode == D3D_DEPTH_PER_PIXEL && p.workspace &&
        p.workspace_bytes >= (size_t)p.n_src * p.C * p.h * p.w * 4) {
        // Hypothesis VOLUMES (the models that have a smooth map hand over its two generating maps, D3D_DEPTH_AFFINE): patches whose
        // hypotheses no window bounds gather their taps from global memory, and that path wants the channel-last copy
        rc = pack_channel_last_g8(p, stream);
        if (rc != D3D_OK) return rc;
        a.cl = reinterpret_cast<const float*>(p.workspace);
    }
    const long nblk = (long)a.tiles_x * a.tiles_y * a.nseg;
    if (nblk > 0x7fffffffL) return D3D_ERR_UNSUPPORTED;
#ifdef D3D_EXPERIMENTS
    if (getenv("D3D_WINDOW_STATS")) {   // debug only: synchronous, allocates
        (void)hipMalloc(&a.stats, 8 * sizeof(unsigned long long));
        (void)hipMemset(a.stats, 0, 8 * sizeof(unsigned long long));
    }
#endif
    hipLaunchKernelGGL(kern, dim3((unsigned)nblk), dim3(64 * window_waves(MODE, CH)), WLDS_BYTES, stream, p, a);
    D3D_LAUNCH_CHECK("sweep_window_kernel launch");
#ifdef D3D_EXPERIMENTS
    if (a.stats) {
        unsigned long long hs[8];
        (void)hipMemcpy(hs, a.stats, sizeof(hs), hipMemcpyDeviceToHost);
        const double n = (double)hs[0];
        fprintf(stderr, "[d3d window stats] CH=%d waves=%d wgs=%.0f groups=%d | per workgroup: staged chunks %.2f, planes per chunk %.2f, gathered planes %.2f, "
                "staged KB %.1f | cycles (wave 0): staging %.0f, sweeping %.0f, total %.0f\n", CH, window_waves(MODE, CH), n, a.ngroups, hs[1] / n,
                hs[1] ? (double)hs[2] / hs[1] : 0.0, hs[3] / n, hs[4] * 16.0 / 1024.0 / n, hs[5] / n, hs[6] / n, hs[7] / n);
        (void)hipFree(a.stats);
    }
#endif
    return D3D_OK;
}

// -DD3D_WINDOW_PH16=n: sweeps of at most n planes take 32 x 16 patches (8 pixel waves, one plane sub-range) instead of 32 x 8
// (4 x 2): what a wave executes besides its planes -- rays, the depth-range table, the windows, its share of the staging: ~2000 of
// the ~3300 instructions of a wave at stage 3 -- is then spread over twice the planes.  On the synthetic stage-3 sweep (flat
// depth map) 0.77 -> 0.63 ms; inside a CasMVSNet view (random weights: a noisy stage-2 depth map) the taller patch's windows no
// longer fit and the sweep takes 3.7 ms instead of 0.76 -- off (profiles/r04_window_phases.txt).
#ifndef D3D_WINDOW_PH16
#define D3D_WINDOW_PH16 0
#endif
#ifndef D3D_WINDOW_PLANES
#define D3D_WINDOW_PLANES 48   // sweeps of at most this many planes take the window kernel (0: never): every stage of the cascades
#endif

template <int MODE, int NSRC>
static int launch_window_ch(const SweepParams& p, hipStream_t stream) {
    // (-DD3D_WINDOW_CG16: 16-channel groups on 12-wave workgroups that own the CU's LDS, for deep sweeps -- the geometry is paid
    //  once per 16 channels, but the windows are twice as large, nothing hides their staging, and chunks shrink: config 2 7.3 ms
    //  against 6.7 ms for the 8-channel form and 5.8 ms for the ring kernel; stage 2 of the cascade 1.33 against 1.31 ms)
#ifdef D3D_WINDOW_CG16
    const bool cg16 = p.C % 16 == 0 && p.D >= D3D_WINDOW_CG16 && p.out_cl != 2;   // (the CL8 store walks ONE plane of 8-channel groups)
#else
    constexpr bool cg16 = false;
#endif
    if constexpr (MODE == MODE_VARIANCE) {
        if (p.out_cl) {
#ifdef D3D_WINDOW_CG16
            if (cg16) return launch_window_one<MODE, NSRC, 16, true, 8>(p, stream);
#endif
#if D3D_WINDOW_PH16 > 0
            if (p.D <= D3D_WINDOW_PH16) return launch_window_one<MODE, NSRC, 8, true, 16>(p, stream);
#endif
            return launch_window_one<MODE, NSRC, 8, true, 8>(p, stream);
        }
    }
    if constexpr (MODE == MODE_WEIGHTED) {   // (round 5: the slice regularisers' cost planes as CL8 16-bit cells -- d3d_weighted_corr_cl8_h16)
        if (p.out_cl == 2) return launch_window_one<MODE, NSRC, 8, true, 8>(p, stream);
    }
    if (p.out_cl) return D3D_ERR_UNSUPPORTED;
#ifdef D3D_WINDOW_CG16
    if (cg16) return launch_window_one<MODE, NSRC, 16, false, 8>(p, stream);
#endif
    (void)cg16;
#if D3D_WINDOW_PH16 > 0
    if (p.D <= D3D_WINDOW_PH16) return launch_window_one<MODE, NSRC, 8, false, 16>(p, stream);
#endif
    return launch_window_one<MODE, NSRC, 8, false, 8>(p, stream);
}

// scratch of the gather path's channel-last copy (d3d_sweep_workspace_bytes takes the larger of this and the ring kernel's)
size_t window_workspace_bytes(int n_src, int C, int D, int h, int w, int elem_bytes) {
    if (elem_bytes != 4 || C % 8 != 0 || n_src > 4 || n_src < 1 || D3D_WINDOW_PLANES == 0 || D > D3D_WINDOW_PLANES) return 0;
    return (size_t)n_src * C * h * w * 4;
}

// Returns D3D_ERR_UNSUPPORTED for shapes outside what the window kernel is built for (the caller then takes the ring kernel).
// `forced`: the test hook asked for this kernel -- the plane limit of the dispatcher's choice does not apply.
int launch_window(int mode, const SweepParams& p, hipStream_t stream, bool forced) {
    if (p.elem_bytes != 4 || p.C % 8 != 0 || p.n_src > 4 || p.n_src < 1) return D3D_ERR_UNSUPPORTED;
    if (mode != MODE_VARIANCE && mode != MODE_WEIGHTED && mode != MODE_PAIR) return D3D_ERR_UNSUPPORTED;
    if (!forced && (D3D_WINDOW_PLANES == 0 || p.D > D3D_WINDOW_PLANES)) return D3D_ERR_UNSUPPORTED;
    if (mode == MODE_PAIR) {   // one source view, every channel in one pass
        if (p.n_src != 1 || p.out_cl) return D3D_ERR_UNSUPPORTED;
#if D3D_WINDOW_PH16 > 0
        if (p.D <= D3D_WINDOW_PH16)
            switch (p.C) {
                case 8: return launch_window_one<MODE_PAIR, 1, 8, false, 16>(p, stream);
                case 16: return launch_window_one<MODE_PAIR, 1, 16, false, 16>(p, stream);
                case 32: return launch_window_one<MODE_PAIR, 1, 32, false, 16>(p, stream);
            }
#endif
        switch (p.C) {
            case 8: return launch_window_one<MODE_PAIR, 1, 8, false, 8>(p, stream);
            case 16: return launch_window_one<MODE_PAIR, 1, 16, false, 8>(p, stream);
            case 32: return launch_window_one<MODE_PAIR, 1, 32, false, 8>(p, stream);
        }
        return D3D_ERR_UNSUPPORTED;
    }
    if (mode == MODE_VARIANCE)
        return p.n_src <= 2 ? launch_window_ch<MODE_VARIANCE, 2>(p, stream) : launch_window_ch<MODE_VARIANCE, 4>(p, stream);
    return p.n_src <= 2 ? launch_window_ch<MODE_WEIGHTED, 2>(p, stream) : launch_window_ch<MODE_WEIGHTED, 4>(p, stream);
}

// Non-default compile-time knobs of this translation unit (d3d_build_flags): empty for the production build.
const char* window_build_flags() {
    return ""
#ifdef D3D_EXPERIMENTS
           " D3D_EXPERIMENTS(window)"
#endif
#if D3D_WINDOW_WG3
           " D3D_WINDOW_WG3"
#endif
#ifdef D3D_WINDOW_CG16
           " D3D_WINDOW_CG16"
#endif
#if D3D_WINDOW_PH16 != 0
           " D3D_WINDOW_PH16"
#endif
#if D3D_WINDOW_DSEG != 32
           " D3D_WINDOW_DSEG"
#endif
#if D3D_WINDOW_RPW != 0
           " D3D_WINDOW_RPW"
#endif
#if D3D_WINDOW_VB != 0
           " D3D_WINDOW_VB"
#endif
#ifdef D3D_CL_PARTIAL_DEFAULT_POLICY
           " D3D_CL_PARTIAL_DEFAULT_POLICY"
#endif
#ifdef D3D_X_SWEEP_NOSTORE
           " D3D_X_SWEEP_NOSTORE"
#endif
#ifdef D3D_X_CONV0_NOLOAD
           " D3D_X_CONV0_NOLOAD"
#endif
        ;
}

}  // namespace d3d
