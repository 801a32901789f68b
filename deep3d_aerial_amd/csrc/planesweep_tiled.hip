// LDS-tiled plane-sweep kernel for gfx950 (MI355X): the fast path behind d3d_variance_volume,
// d3d_weighted_corr, d3d_pair_corr_mean and d3d_homo_warp.
//
// Why a tiled kernel: per voxel and source view a bilinear sample touches 4 taps x C channels
// (2 KB per voxel at C = 32, V = 5) while only 4*C bytes are written, so the gather must not
// go to L2 -- it is served from LDS.  Design (numbers in DESIGN.md):
//
//   * One workgroup (10 compute + 2 loader waves) = a 32 x 4 patch of reference pixels x a segment
//     of depth planes x a group of CH channels.  It walks the segment in STEPS of SP = NSUB*m
//     planes: compute wave (pw, sub) owns two patch rows and the planes d = step*SP + sub + NSUB*j.
//   * The source footprint of (patch x step) is bounded per view by projecting the 8 corners of
//     the frustum slab (valid because the map is projective and p.z > 0 at every corner).  All
//     step windows of the segment are planned ONCE in the prologue, one lane per step.
//   * Each view owns a RING in LDS: a torus of RW x RH positions, channel-last
//     ([position][CH+4 floats]; the +4 pad spreads ds_read_b128 over all 64 banks), with column
//     RW and row RH duplicating column 0 and row 0 so that the east/south taps never wrap.
//     Moving to the next step only stages the thin L-shaped difference of consecutive windows.
//     RW/RH are at least the span of the UNION of any two consecutive windows, so the slots
//     that difference lands in are never part of the window being read: the loader waves
//     write step k+1's difference while the compute waves work on step k, and one barrier per
//     step is the only synchronisation.  Compute waves never wait on a global load.  Positions
//     outside the source image are staged as zeros (= per-tap zero padding).
//   * Compute: a lane owns one reference pixel and CH channels.  Per plane it does the geometry
//     ONCE per view (3 mul/add, rcp + Newton, floor/fract, window test, ring address) and then,
//     per channel quad and view, four ds_read_b128 feed 4 x (4 interpolation + 2 accumulation)
//     VALU ops; the reads of the next (quad, view) unit are issued before the current one is
//     blended.  Lanes are permuted so that each hardware ds_read_b128 lane group covers 16
//     consecutive pixels (conflict-free for unit-scale sweeps).  Results leave as 128-byte row
//     segments per (channel, plane) -- full-line HBM writes -- through scalar-base stores.
//   * If the rings cannot be made to fit (p.z <= 0 at a corner, extreme scale/rotation) the
//     workgroup runs the same loop nest with taps gathered from global memory instead.
//
// Nothing is reshaped into a GEMM: this is HBM-write-bound streaming work with an LDS-fed
// gather, not MFMA work.
#include "common.h"

#include <hip/hip_fp16.h>

#include "sweep_params.h"
#include "sweep_device.h"

#include <cstdlib>

namespace d3d {

namespace {

constexpr int TW = 32;          // patch width  (one 128-byte output row segment per lane-row)
constexpr int TH = 4;           // patch height
constexpr int NPIXW = 2;        // pixel waves (2 patch rows each)
#ifndef D3D_NSUB
#define D3D_NSUB 4
#endif
#ifndef D3D_NLOADW
#define D3D_NLOADW 4
#endif
constexpr int NSUB = D3D_NSUB;  // depth sub-ranges
constexpr int NCOMP = NPIXW * NSUB;  // compute waves
constexpr int NLOADW = D3D_NLOADW;  // loader waves: stage the next step's window delta into the rings
constexpr int NWAVES = NCOMP + NLOADW;  // 12 waves: 3 per SIMD, 168-VGPR budget
constexpr int THREADS = 64 * NWAVES;
constexpr int DSEG_MAX = 128;   // planes per workgroup segment upper bound (launch_one never asks for more)
constexpr int MAXSTEPS = 64;    // >= DSEG_MAX / NSUB, <= 64 (one lane per step)
constexpr int PFD = 2;          // delta staging items (64 positions x CH channels) a loader wave keeps in flight
#ifndef D3D_LDS_PIPE
#define D3D_LDS_PIPE 1
#endif
// D3D_GRAB: the planes of a step are handed out DYNAMICALLY (one LDS counter per step and pixel wave) instead of plane j
// going to depth sub-range j mod NSUB.  Why: issue arbitration on a SIMD goes by wave age, so the oldest wave of a SIMD
// runs its planes at full single-wave speed, the younger ones on the leftover slots; with a static split the old waves
// then idle at the step barrier while the SIMD drops to the (much lower: tools/plane_loop_rate.hip) throughput of one or
// two active waves.  Handing out planes as waves become free keeps every wave of a SIMD busy until the step runs out.
// (Lowering a wave's priority with s_setprio as it advances was tried instead: priorities are STRICT -- a lower-priority
// wave does not even get the leftover slots -- and the step took 1.3x longer.)
#ifndef D3D_GRAB
#define D3D_GRAB 0
#endif
// D3D_FREERUN: no step barriers.  The compute waves take planes from ONE queue per pixel wave that runs through the whole
// pass, and meet the loader waves only through two LDS counters per step: staged[k] (loader waves that have finished
// delta k) and done[k] (planes of step k that have been swept).  A compute wave enters step k once staged[k] is
// complete; the loaders start delta k once done[k-2] is complete (delta k may reuse slots of window k-2 only: the rings
// hold the union of two consecutive windows), so the compute waves of a workgroup are at most one step apart and never
// wait for EACH OTHER.  Why it matters: a SIMD of gfx950 reaches its vector throughput only with three or more waves
// issuing (tools/plane_loop_rate.hip: 1766 / 1667 / 1125 cycles per plane and channel group at 1 / 2 / 3 waves per SIMD),
// issue arbitration goes by wave age, and with a barrier per step the old waves of a SIMD finish their planes first and
// then idle while the young ones run alone at the one-wave rate.
#ifndef D3D_FREERUN
#define D3D_FREERUN 0
#endif
constexpr int LDS_PIPE = D3D_LDS_PIPE;     // (quad, view) units whose taps are requested ahead of the one being blended
constexpr int NCAND = 4;        // candidate step sizes: 4, 2, 1, 1/2 times NSUB planes
constexpr int MAXRECTS = 512;   // non-empty delta rectangles per workgroup segment (4 words each)
// LDS bank layout of the rings (tools/bank_sim.py models it on config 2; PMC: SQ_LDS_BANK_CONFLICT).  A ds_read_b128 is
// serviced in four 16-lane groups, one 16-byte slot (4 banks) per lane, 16 slots per LDS cycle.  A lane group holds an
// 8 x 2 block of reference pixels (lane -> pixel map below), whose taps span < 8 columns and 2 (rarely 3) rows of a
// source view for scales up to 1.14.  With a position stride of an ODD number of slots (STRIDE = CH + 4 floats) and a
// row pitch of 8 slots mod 16 (32 floats mod 64), two positions share a slot only if they lie (16,0), (8,1) or (0,2)
// apart -- outside such a block.  The torus wrap keeps that property when RW is a multiple of 8 and RH is even.
// Round 1 (16 x 1 pixels per group, unpadded pitch): 6.2 LDS cycles per tap read on config 2; this layout: 4.3.
constexpr int RING_RW_QUANT = 8;
constexpr int RING_RH_QUANT = 2;
template <int STRIDE>
__host__ __device__ __forceinline__ int ring_row_floats(int RW) {
    const int n = (RW + 1) * STRIDE;
    return ((n - 32 + 63) & ~63) + 32;   // smallest pitch >= n that is 32 mod 64
}
// non-empty delta rectangles per workgroup segment: MAXRECTS

// Cycle stamps and plan statistics of the kernel (D3D_TILED_STATS): compiled only into -DD3D_EXPERIMENTS builds -- in the
// production kernel the stamps and their accumulators would be live scalar registers across the plane loop.
#ifdef D3D_EXPERIMENTS
#define D3D_STAMPS 1
#else
#define D3D_STAMPS 0
#endif
#define D3D_CLOCK() (D3D_STAMPS ? clock64() : 0ll)

struct TiledArgs {
    int ngroups;     // channel groups (C / CH), one workgroup pass each
    int dseg;        // planes per workgroup segment
    int cap_floats;  // floats available for the rings
    int nseg;        // segments along depth
    int tiles_x, tiles_y;
    const float* cl;  // channel-last copy of the source maps [view][group][h*w][CH] (pack_channel_last_kernel), or null
    unsigned long long* tstats;  // debug timing (cycles, wave 0): [0] prologue [1] barrierA+write+barrierB [2] issue [3] compute [4] total
    unsigned* stats;  // debug (D3D_TILED_STATS): [0] ring wgs [1] fallback wgs [2] sum m [3] sum nsteps [4] sum ring floats [5] overflow items
};

// LDS map (ints/floats):
//   [zero cell 2*STRIDE][pmin DSEG_MAX][pmax DSEG_MAX][header 32][plan table MAXSTEPS*NSRC*8]
//   [rect starts MAXSTEPS+4][rect descriptors MAXRECTS*8][rings ...]
// header: 0 mode (1 rings, 0 global gather) | 1 planes per step | 2 nsteps | 4+4i.. RW, RH, base, RW | RH << 8 | (base/4) << 16 per view
// plan entry (per step, view): wx0, wy0, ww, wh, ox, oy, ox - wx0, oy - wy0
// rect descriptor (non-empty delta rectangles, grouped by step), 4 words:
//   (rx + 1) | (ry + 1) << 16,  width | view << 12 | first position (cumulative within the step) << 16,
//   positions,  ring col | row << 16 of the rect origin (unwrapped)
template <int CW, int NSRC>   // CW = 4-byte words per ring position (CH fp32 channels, or CH/2 words of fp16 pairs)
struct Lds {
    static constexpr int STRIDE = CW + 4;
    static constexpr int ZERO = 0;
    static constexpr int PMIN = 2 * STRIDE;
    static constexpr int PMAX = PMIN + DSEG_MAX;
    static constexpr int HDR = PMAX + DSEG_MAX;
    static constexpr int VT = HDR + 32;                // per view: T0, T1, T2, -, RW, RH, row bytes, absolute ring base (view-major plane loop)
    static constexpr int CNT = VT + 8 * NSRC;          // per-step hand-off counters: [MAXSTEPS] staged, [MAXSTEPS] done
    static constexpr int PLAN = CNT + 2 * MAXSTEPS;
    static constexpr int SST = PLAN + MAXSTEPS * NSRC * 8;
    static constexpr int STOT = SST + MAXSTEPS + 4;   // positions to stage per step
    static constexpr int RECTS = ((STOT + MAXSTEPS + 3) / 4) * 4;
    static constexpr int DATA = RECTS + (MAXRECTS + 1) * 4;  // 16-byte aligned
};

struct Rects {  // window(k) minus window(k-1): left | right | top | bottom (any may be empty)
    int rx[4], ry[4], rw[4], rh[4], start[5];
};


// Step hand-off between the loader and the compute waves of a workgroup.  D3D_DECOUPLE = 1 (experiment, off):
// two LDS counters per step instead of one s_barrier per step -- a compute wave starts step k as soon as the
// loaders have staged it, and the loaders start staging step k+1 as soon as EVERY compute wave has left step k-1.
// Measured: the 18 % "barrier wait" of the compute waves barely moves (43k -> 39.5k cycles per workgroup) -- they
// wait for the staging of the next window, not for each other -- so the plain barrier stays.  The spin is bounded
// (no hang whatever happens; ~100x longer than any legitimate wait).
#ifndef D3D_DECOUPLE
#define D3D_DECOUPLE 0
#endif
#ifndef STAGE0_ALL
#define STAGE0_ALL 1  // compute waves help to stage the first window of a pass: neutral on deep sweeps (239k cycles per workgroup
                     // either way), -7 % on the cascade's 8-plane stage where that window is the only staging there is
#endif
#define STAGE0_ALL_EFF (STAGE0_ALL && !D3D_DECOUPLE)
__device__ __forceinline__ void step_signal(int* ctr, int lane) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    if (lane == 0) __hip_atomic_fetch_add(ctr, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ void step_wait(int* ctr, int target) {
#ifndef D3D_SPIN_LOG2
#define D3D_SPIN_LOG2 16
#endif
    for (int spin = 0; spin < (1 << D3D_SPIN_LOG2); ++spin) {
        if (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) >= target) break;
        __builtin_amdgcn_s_sleep(2);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}
// Counter hand-off in LDS (D3D_FREERUN).  The LDS executes the DS operations of the CU in arrival order and those of one wave
// in program order, so "ring writes, then counter add" by one wave and "counter read, then ring reads" by another need no
// fence beyond keeping the COMPILER from reordering them (a workgroup-scope fence would also wait for every global store
// the wave has in flight).  Waits are bounded: a wave that gives up sweeps on (wrong results, never a hang).
__device__ __forceinline__ void ctr_add(int* ctr, int lane) {
    asm volatile("" ::: "memory");
    if (lane == 0) __hip_atomic_fetch_add(ctr, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    asm volatile("" ::: "memory");
}
__device__ __forceinline__ void ctr_wait(int* ctr, int target) {
    for (int spin = 0; spin < (1 << 22); ++spin) {
        const int v = __builtin_amdgcn_readfirstlane(__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
        if (v >= target) break;
        __builtin_amdgcn_s_sleep(1);
    }
    asm volatile("" ::: "memory");
}
__device__ __forceinline__ int posmod(int a, int n) {
    int r = a % n;
    return r < 0 ? r + n : r;
}


// Wave-uniform constants of one source view's ring (scalar registers in the plane loop).
struct RingView {
    int RW, RH;  // torus size in positions
    int rowb;    // bytes between ring rows
    int base;    // absolute LDS byte address of ring position (0, 0)
};

// Projection + bilinear weights + ring addresses of one (pixel, plane, view): ~34 VALU instructions, no branches, no
// predicates.  (u, v) are CLAMPED to [-1, w] x [-1, h] instead of being tested against the window: the planner clamps the
// hull of the patch the same way (monotone, so a clamped sample lies in the clamped hull), windows may reach one position
// past the image on the left / top and two on the right / bottom, and every position outside the image is staged as
// zeros -- a sample outside the image reads zero taps (or taps of weight exactly 0), which is per-tap zero padding.
// (kx, ky) = ring offset minus window origin of the current step.  Non-finite projections end at a clamp bound
// (v_med3_f32 returns the smallest operand when one is a NaN).
template <int STRIDE>
__device__ __forceinline__ TapL geo_ring(const Ray& r, float tx, float ty, float tz, float d, float umax, float vmax, int kx, int ky,
                                         const RingView& R) {
    const float px = __fadd_rn(__fmul_rn(r.rx, d), tx);
    const float py = __fadd_rn(__fmul_rn(r.ry, d), ty);
    const float pz = __fadd_rn(__fmul_rn(r.rz, d), tz);
    // quotient = v_rcp_f32 estimate + one Newton correction (correctly rounded in practice: common.h project())
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
    unsigned c = (unsigned)((int)fu + kx), rr = (unsigned)((int)fv + ky);
    c = min(c, c - (unsigned)R.RW);  // one conditional wrap: 0 <= c < 2 RW
    rr = min(rr, rr - (unsigned)R.RH);
    t.a0 = R.base + (int)__umul24(rr, (unsigned)R.rowb) + (int)__umul24(c, (unsigned)(STRIDE * 4));
    t.a1 = t.a0 + R.rowb;
    return t;
}


// fp16 storage: one 2-byte store per lane (a wave writes two 64-byte row segments per channel and plane)
__device__ __forceinline__ void store_sbase_h(unsigned long long sb, unsigned byte_off, float v) {
    const _Float16 hv = (_Float16)v;   // RNE
    const unsigned bits = __builtin_bit_cast(unsigned short, hv);
    asm volatile("global_store_short %0, %1, %2 nt" : : "v"(byte_off), "v"(bits), "s"(sb));
}
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
// fp32 fma whose first factor is the LOW / HIGH fp16 half of a 32-bit word (v_fma_mix_f32): the fp16 tap is widened
// inside the instruction -- exact -- so no conversion instruction and no fp32 copy of the tap exist (hipcc prefers
// v_cvt_f32_f16 + packed fp32 fma here: 32 more instructions and 32 more live registers per 8-channel unit).
__device__ __forceinline__ float fma_mix_lo(float pair, float w, float acc) {
    float d;
    asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel_hi:[1,0,0]" : "=v"(d) : "v"(pair), "v"(w), "v"(acc));
    return d;
}
__device__ __forceinline__ float fma_mix_hi(float pair, float w, float acc) {
    float d;
    asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(d) : "v"(pair), "v"(w), "v"(acc));
    return d;
}
__device__ __forceinline__ float ldf(const float* q) { return *q; }
__device__ __forceinline__ float ldf(const __half* q) { return __half2float(*q); }

}  // namespace

// T = float: tensors as declared in SweepParams.  T = __half (MODE_VARIANCE only; BASELINE config 5): feats[] and out
// are fp16 tensors of the same shapes, the rings hold fp16 cells (half the LDS bytes per tap), every product and sum
// is fp32 (v_fma_mix_f32 reads the fp16 tap directly), the result is rounded once (RNE) at the store.
// OUTCL: the variance volume leaves as a channel-last bf16 volume [D][h][w][C] (RNE) -- the form conv0 of the 3-D
// regulariser stages in bf16 mode (conv_c8.hip); four channels of a voxel = one 8-byte store instead of four 4-byte ones.
// NSUB_T / NLOAD_T: depth sub-ranges and loader waves of a workgroup.  The default (4 + 4: 12 waves, all of a CU's LDS) is
// sized for deep sweeps; SHALLOW sweeps (the last cascade stage: 8 planes) run 2 + 2 = 6 waves on half the LDS, so TWO
// workgroups share a CU and one's planning prologue and first-window latency (most of its life: 47 % + 40 % at 8 planes)
// overlap the other's arithmetic.
template <int MODE, int NSRC, int CH, typename T = float, bool OUTCL = false, int NSUB_T = D3D_NSUB, int NLOAD_T = D3D_NLOADW,
          int NPIXW_T = 2>
__global__ __launch_bounds__(64 * (NPIXW_T * NSUB_T + NLOAD_T), 3) void sweep_tiled_kernel(SweepParams p, TiledArgs a) {   // 3 waves per SIMD: the 168-register budget for either workgroup size
    constexpr int NSUB = NSUB_T, NLOADW = NLOAD_T, NPIXW = NPIXW_T, TH = 2 * NPIXW;   // (shadow the defaults)
    constexpr int NCOMP = NPIXW * NSUB, NWAVES = NCOMP + NLOADW, THREADS = 64 * NWAVES;
    constexpr bool F16 = sizeof(T) == 2;
    static_assert(!OUTCL || (MODE == MODE_VARIANCE && !F16), "channel-last bf16 output is built for the fp32 variance volume");
    constexpr int CW = CH * (int)sizeof(T) / 4;   // words per ring position
    constexpr int CPC = 16 / (int)sizeof(T);      // channels per 16-byte chunk (one ds_read_b128): 4 | 8
    using L = Lds<CW, NSRC>;
    constexpr int STRIDE = L::STRIDE;
    constexpr int Q = CW / 4;                     // 16-byte chunks per position
    static_assert(!F16 || MODE == MODE_VARIANCE, "fp16 storage is built for the variance volume only");
    constexpr bool VIEW_MAJOR = F16;             // unit order of the compute loop (see there)
    constexpr bool FREERUN = D3D_FREERUN && NLOADW > 0 && NPIXW <= 2;   // (the plane queues: one per pixel wave, two header words)
    constexpr int MAXRS = 4 * NSRC;               // delta rectangles a step can have (left | right | top | bottom per view)
    static_assert(NSRC <= 15, "rect descriptors keep the view in 4 bits");
    extern __shared__ __attribute__((aligned(16))) float lds[];
    int* ldsi = reinterpret_cast<int*>(lds);

    const long long t_start = D3D_CLOCK();
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = rfl(tid >> 6);
    const bool loader = wave >= NCOMP;
    const int lw = wave - NCOMP;                 // loader index
    const int pw = (wave % NCOMP) % NPIXW;       // which pair of patch rows
    const int sub = (wave % NCOMP) / NPIXW;      // depth sub-range
    const int h = p.h, w = p.w, D = p.D;
    const size_t plane = (size_t)h * w;

    // block -> (patch row, segment, channel group, patch column).  Blocks b, b+8, ... share an XCD
    // (and its L2): give each XCD a contiguous run of the logical order.
    int b = blockIdx.x;
    {
        const int nblk = gridDim.x;
        const int per = nblk / 8;
        if (nblk % 8 == 0) b = (b % 8) * per + b / 8;
    }
    // Within an XCD's run, vertically adjacent patches come first: they share most of their source
    // rows (same depth segment, same channel group), so the windows of the ~32 workgroups an XCD runs
    // concurrently overlap in its L2 instead of each being fetched from HBM.
    const int ty = b % a.tiles_y;
    const int seg = (b / a.tiles_y) % a.nseg;
    const int tx = b / (a.tiles_y * a.nseg);
    // The workgroup sweeps ALL channel groups of its (patch, depth segment), one after the other: the plan (windows,
    // step size, staging rectangles) does not depend on the channels, so the prologue is paid once.
    const int x0 = tx * TW, y0 = ty * TH;
    const int ds = seg * a.dseg;
    const int de = min(ds + a.dseg, D);
    const int nplanes = de - ds;
    const int x1c = min(x0 + TW - 1, w - 1), y1c = min(y0 + TH - 1, h - 1);

    // Lane -> pixel.  ds_read_b128 is serviced in the lane groups {0-3,12-15,20-27} and
    // {4-11,16-19,28-31} (+32; confirmed with tools/lds_groups.hip): each group takes an 8 x 2 block of
    // pixels (bank layout: see ring_row_floats).  Lanes 0-15 / 16-31 / 32-47 / 48-63 each hold 16
    // consecutive pixels of one row, so a store instruction still writes whole 64-byte row segments.
    const int l5 = lane & 31;
    const int pxl = (l5 < 4) ? l5 : (l5 < 12) ? l5 + 4 : (l5 < 20) ? l5 - 8 : (l5 < 28) ? l5 - 20 : l5 - 16;
    const int px = x0 + pxl + 16 * (lane >> 5);
    const int py = y0 + pw * 2 + (l5 >> 4);
    const bool valid = (px < w) && (py < h);
    const int pix = valid ? py * w + px : 0;
    const unsigned pixb = (unsigned)pix * (unsigned)sizeof(T);  // per-lane byte offset (h*w < 2^30)
    // channel-last cells: C * 2 bytes per pixel (h*w*C*2 < 2^32, checked at launch), or 16 in a plane of 8-channel groups (CL8)
    const unsigned pixo = OUTCL ? (p.out_cl == 2 ? (unsigned)pix * 16u : (unsigned)pix * (unsigned)p.C * 2u) : pixb;
    const float xf = (float)px, yf = (float)py;

    // --- per-lane inputs of the compute waves, requested BEFORE the planning phases below so that their global-memory latency
    //     (projection rows, the reference features of the first channel group, view weights, hypothesis maps) runs under
    //     the planning instead of after it: on the cascade's shallow sweeps the prologue is most of a workgroup's life ----
    //     (Up to four source views.  With six, 36 more registers live across the planning make the plane loop spill.)
    constexpr bool EARLY_INPUTS = NSRC <= 4 && !F16 && CH <= 16;
    int grp = 0, c0 = 0;  // current channel group / its first channel
    Ray ray[NSRC];
    float T0[NSRC], T1[NSRC], T2[NSRC];
    auto projection_rows = [&]() {
#pragma unroll
        for (int i = 0; i < NSRC; ++i) {
            const float* __restrict__ M = p.proj34 + 12 * min(i, p.n_src - 1);
            ray[i] = make_ray(M, xf, yf);
            T0[i] = M[3]; T1[i] = M[7]; T2[i] = M[11];
            // (kept in vector registers: as scalars they would be live across the whole planning prologue, and the scalar
            //  file is what spills into the plane loop -- one v_readlane per use)
            if constexpr (EARLY_INPUTS) asm volatile("" : "+v"(T0[i]), "+v"(T1[i]), "+v"(T2[i]));
        }
    };
    if constexpr (EARLY_INPUTS) projection_rows();
    f4 r[F16 ? 1 : Q];
    float rh[F16 ? Q : 1][8];   // fp16 storage: reference features of the group, 8 channels per chunk
    auto load_reference = [&]() {
        if (F16) {
#pragma unroll
            for (int q = 0; q < Q; ++q)
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    float t = ldf(reinterpret_cast<const T*>(p.feats[0]) + (size_t)(c0 + 8 * q + k) * plane + pix);
                    rh[F16 ? q : 0][k] = valid ? t : 0.0f;
                }
        } else if (MODE != MODE_WARP) {
#pragma unroll
            for (int q = 0; q < Q; ++q)
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    float t = p.feats[0][(size_t)(c0 + 4 * q + k) * plane + pix];
                    r[F16 ? 0 : q][k] = valid ? t : 0.0f;
                }
        }
    };
    float vw[NSRC];
    float rden = 0.0f;
    float aff_lo = 1.0f, aff_step = 0.0f;   // D3D_DEPTH_AFFINE: this lane's pixel
    auto lane_inputs = [&]() {
        if (MODE == MODE_WEIGHTED) {
            float den = 1e-5f;
#pragma unroll
            for (int i = 0; i < NSRC; ++i) {
                float t = p.weights[(size_t)min(i, p.n_src - 1) * plane + pix];
                vw[i] = (valid && i < p.n_src) ? t : 0.0f;
                den += vw[i];
            }
            rden = 1.0f / den;
        }
        load_reference();   // group 0
        if (p.depth_mode == D3D_DEPTH_AFFINE && valid) {
            aff_lo = p.depth[pix];
            aff_step = p.depth[plane + pix];
        }
    };
    if (EARLY_INPUTS && !loader) lane_inputs();

    // --- zero cell and per-plane depth range of this patch -------------------------------------
    for (int i = tid; i < 2 * STRIDE; i += THREADS) lds[L::ZERO + i] = 0.0f;
    if (p.depth_mode == D3D_DEPTH_PER_PLANE) {
        for (int i = tid; i < nplanes; i += THREADS) {
            float dv = p.depth[ds + i];
            lds[L::PMIN + i] = dv;
            lds[L::PMAX + i] = dv;
        }
    } else if (p.depth_mode == D3D_DEPTH_AFFINE) {
        // hypotheses lo + k * step per pixel (two maps): each lane reads its pixels' pair ONCE, the per-plane range of the patch
        // is arithmetic (the per-pixel form below re-reads the patch for every plane: a global-latency chain per plane)
        float blo[(TW * TH) / 64], bst[(TW * TH) / 64];
        bool bok[(TW * TH) / 64];
#pragma unroll
        for (int k = 0; k < (TW * TH) / 64; ++k) {
            const int q = lane + 64 * k;
            const int qx = x0 + (q & 31), qy = y0 + (q >> 5);
            bok[k] = qx < w && qy < h;
            const size_t qi = bok[k] ? (size_t)qy * w + qx : 0;
            blo[k] = p.depth[qi];
            bst[k] = p.depth[plane + qi];
        }
        for (int i = wave; i < nplanes; i += NWAVES) {
            float lo = INFINITY, hi = -INFINITY;
#pragma unroll
            for (int k = 0; k < (TW * TH) / 64; ++k) {
                const float dv = __fadd_rn(blo[k], __fmul_rn((float)(ds + i), bst[k]));
                if (bok[k]) { lo = fminf(lo, dv); hi = fmaxf(hi, dv); }
            }
            lo = wave_min(lo);
            hi = wave_max(hi);
            if (lane == 0) {
                lds[L::PMIN + i] = lo;
                lds[L::PMAX + i] = hi;
            }
        }
    } else {
        for (int i = wave; i < nplanes; i += NWAVES) {
            float lo = INFINITY, hi = -INFINITY;
#pragma unroll
            for (int k = 0; k < (TW * TH) / 64; ++k) {
                int q = lane + 64 * k;
                int qx = x0 + (q & 31), qy = y0 + (q >> 5);
                if (qx < w && qy < h) {
                    float dv = p.depth[(size_t)(ds + i) * plane + (size_t)qy * w + qx];
                    lo = fminf(lo, dv);
                    hi = fmaxf(hi, dv);
                }
            }
            lo = wave_min(lo);
            hi = wave_max(hi);
            if (lane == 0) {
                lds[L::PMIN + i] = lo;
                lds[L::PMAX + i] = hi;
            }
        }
    }
    __syncthreads();
    const long long t_range = D3D_CLOCK();

    // --- plan every step of the segment ---------------------------------------------------------
    // P1: wave (candidate, view) computes, one lane per step, that view's windows for the candidate
    //     step size m in {4, 2, 1}; results go to a scratch table in the (still unused) ring area.
    // P2: wave 0 picks the largest candidate whose rings fit and builds the plan / rectangle tables.
    constexpr int CT = L::DATA;                          // [cand][view][step][4] windows
    constexpr int CHD = CT + NCAND * NSRC * 64 * 4;      // [cand][view][4]: RW, RH, bad, nsteps
    // candidate c steps by NSUB*4, NSUB*2, NSUB, NSUB/2 planes (the last leaves half the sub-waves idle,
    // which is still far better than gathering from global memory)
    auto cand_planes = [](int c) { return c < 3 ? NSUB * (4 >> c) : max(NSUB / 2, 1); };
    auto candidate_windows = [&](int cand, int vsel) {
        {
            const int SPc = cand_planes(cand);
            const int nstepc = (nplanes + SPc - 1) / SPc;
            const bool feasible = nstepc <= MAXSTEPS;
            const bool act = feasible && lane < nstepc;
            const int p0 = min(lane * SPc, nplanes - 1), p1 = min(lane * SPc + SPc, nplanes);
            float lo = INFINITY, hi = -INFINITY;
            for (int i = p0; i < p1; ++i) {
                lo = fminf(lo, lds[L::PMIN + i]);
                hi = fmaxf(hi, lds[L::PMAX + i]);
            }
            const float* __restrict__ M = p.proj34 + 12 * min(vsel, p.n_src - 1);
            float umin = INFINITY, umax = -INFINITY, vmin = INFINITY, vmax = -INFINITY;
            bool ok = true;
#pragma unroll
            for (int ck = 0; ck < 8; ++ck) {
                Ray cr = make_ray(M, (ck & 1) ? (float)x1c : (float)x0, (ck & 2) ? (float)y1c : (float)y0);
                const float dv = (ck & 4) ? hi : lo;
                float qx = __fadd_rn(__fmul_rn(cr.rx, dv), M[3]);
                float qy = __fadd_rn(__fmul_rn(cr.ry, dv), M[7]);
                float qz = __fadd_rn(__fmul_rn(cr.rz, dv), M[11]);
                ok = ok && (qz > 1e-20f) && (qz < 1e30f);
                float iz = 1.0f / qz;
                float u = qx * iz, v = qy * iz;
                ok = ok && (fabsf(u) < 1e30f) && (fabsf(v) < 1e30f);
                // the same clamp as the samples' (geo_ring): windows lie in [-1, w + 1] x [-1, h + 1]
                u = fminf(fmaxf(u, -1.0f), (float)w);
                v = fminf(fmaxf(v, -1.0f), (float)h);
                umin = fminf(umin, u); umax = fmaxf(umax, u);
                vmin = fminf(vmin, v); vmax = fmaxf(vmax, v);
            }
            // interior samples differ from the corner hull by fp32 rounding only (<< 1/16 px)
            int wx0 = max((int)floorf(umin - 0.0625f), -1);
            int wy0 = max((int)floorf(vmin - 0.0625f), -1);
            int wx1 = min((int)floorf(umax + 0.0625f) + 1, w + 1);
            int wy1 = min((int)floorf(vmax + 0.0625f) + 1, h + 1);
            int ww = max(wx1 - wx0 + 1, 0), wh = max(wy1 - wy0 + 1, 0);   // >= 2 x 2: every sample has its four taps in the window
            // unused view of the template (n_src < NSRC): its samples sit at (-1, -1) (geo_ring with a zero ray), weight 1 on
            // the tap at (-1, -1) -- a zero -- and weight 0 on the other three
            if (vsel >= p.n_src) { wx0 = -1; wy0 = -1; ww = 2; wh = 2; ok = true; }
            if (!act) { ww = 0; wh = 0; ok = true; }
            // ring size: the union of consecutive windows must fit (asynchronous delta staging)
            int ux = ww, uy = wh;
            {
                const int qx0 = __shfl_up(wx0, 1), qy0 = __shfl_up(wy0, 1);
                const int qw = __shfl_up(ww, 1), qh = __shfl_up(wh, 1);
                if (lane > 0 && act && qw > 0 && ww > 0) {
                    ux = max(wx0 + ww, qx0 + qw) - min(wx0, qx0);
                    uy = max(wy0 + wh, qy0 + qh) - min(wy0, qy0);
                }
            }
            const int RWc = max(wave_maxi(ux), 1), RHc = max(wave_maxi(uy), 1);
            const int badc = (__any(!ok) || !feasible) ? 1 : 0;  // p.z <= 0 somewhere: cannot bound the frustum
            int* e = ldsi + CT + ((cand * NSRC + vsel) * 64 + lane) * 4;
            e[0] = wx0; e[1] = wy0; e[2] = ww; e[3] = wh;
            if (lane == 0) {
                int* hd = ldsi + CHD + (cand * NSRC + vsel) * 4;
                hd[0] = RWc; hd[1] = RHc; hd[2] = badc; hd[3] = nstepc;
            }
        }
    };
    // the three regular candidates first (one round: NWAVES = 3 * 4 waves); the half-step candidate only when none
    // of them fits (rare: very wide plane spacing)
    constexpr int NCAND1 = NCAND - 1;
    for (int wk = wave; wk < NCAND1 * NSRC; wk += NWAVES) candidate_windows(wk / NSRC, wk % NSRC);
    __syncthreads();
    const long long t_p1 = D3D_CLOCK();
    auto plan_tables = [&](int cbeg, int cend) {  // wave 0: first candidate in [cbeg, cend) whose rings fit
        int mode = 0, sp_sel = NSUB, nsteps_sel = (nplanes + NSUB - 1) / NSUB;
        for (int cand = cbeg; cand < cend && !mode; ++cand) {
            const int nsteps = ldsi[CHD + (cand * NSRC) * 4 + 3];
            const bool act = lane < nsteps;
            int wx0[NSRC], wy0[NSRC], ww[NSRC], wh[NSRC], RW[NSRC], RH[NSRC];
            int bad = 0, total = 0, totalq = 0;
            int RWq[NSRC], RHq[NSRC];   // rounded up for the bank layout (any size >= the union span is valid)
#pragma unroll
            for (int i = 0; i < NSRC; ++i) {
                const int* hd = ldsi + CHD + (cand * NSRC + i) * 4;
                RW[i] = hd[0]; RH[i] = hd[1]; bad |= hd[2];
                RWq[i] = (RW[i] + RING_RW_QUANT - 1) / RING_RW_QUANT * RING_RW_QUANT;
                RHq[i] = (RH[i] + RING_RH_QUANT - 1) / RING_RH_QUANT * RING_RH_QUANT;
                total += ring_row_floats<STRIDE>(RW[i]) * (RH[i] + 1);
                totalq += ring_row_floats<STRIDE>(RWq[i]) * (RHq[i] + 1);
            }
            if (totalq <= a.cap_floats) {   // the conflict-free ring sizes fit: use them
                total = totalq;
#pragma unroll
                for (int i = 0; i < NSRC; ++i) { RW[i] = RWq[i]; RH[i] = RHq[i]; }
            }
            if (bad || total > a.cap_floats) continue;
#pragma unroll
            for (int i = 0; i < NSRC; ++i) {
                const int* e = ldsi + CT + ((cand * NSRC + i) * 64 + lane) * 4;
                wx0[i] = e[0]; wy0[i] = e[1]; ww[i] = e[2]; wh[i] = e[3];
            }
            if (!bad && total <= a.cap_floats) {
                mode = 1;
                sp_sel = cand_planes(cand);
                nsteps_sel = nsteps;
                int base = L::DATA;
                int oxv[NSRC], oyv[NSRC], basev[NSRC];
                bool dims_ok = true;
#pragma unroll
                for (int i = 0; i < NSRC; ++i) {
                    // ring coordinates are taken relative to the first step's window origin
                    const int cx0 = __shfl(wx0[i], 0), cy0 = __shfl(wy0[i], 0);
                    oxv[i] = posmod(wx0[i] - cx0, RW[i]);
                    oyv[i] = posmod(wy0[i] - cy0, RH[i]);
                    basev[i] = base;
                    dims_ok = dims_ok && RW[i] < 256 && RH[i] < 256;
                    if (act) {
                        int* e = ldsi + L::PLAN + (lane * NSRC + i) * 8;
                        e[0] = wx0[i]; e[1] = wy0[i]; e[2] = ww[i]; e[3] = wh[i];
                        e[4] = oxv[i];
                        e[5] = oyv[i];
                        e[6] = oxv[i] - wx0[i];   // ring column of source column 0 (geo_ring's kx, ky)
                        e[7] = oyv[i] - wy0[i];
                    }
                    if (lane == 0) {
                        ldsi[L::HDR + 4 + 4 * i + 0] = RW[i];
                        ldsi[L::HDR + 4 + 4 * i + 1] = RH[i];
                        ldsi[L::HDR + 4 + 4 * i + 2] = base;
                        ldsi[L::HDR + 4 + 4 * i + 3] = RW[i] | (RH[i] << 8) | ((base >> 2) << 16);
                        if constexpr (VIEW_MAJOR) {   // the plane loop reads its per-view constants back from here (see `geometry`)
                            const float* __restrict__ M = p.proj34 + 12 * min(i, p.n_src - 1);
                            const bool dummy = i >= p.n_src;   // unused view of the template: every sample at (-1, -1)
                            lds[L::VT + 8 * i + 0] = dummy ? -1.0f : M[3];
                            lds[L::VT + 8 * i + 1] = dummy ? -1.0f : M[7];
                            lds[L::VT + 8 * i + 2] = dummy ? 1.0f : M[11];
                            lds[L::VT + 8 * i + 3] = 0.0f;
                            ldsi[L::VT + 8 * i + 4] = RW[i];
                            ldsi[L::VT + 8 * i + 5] = RH[i];
                            ldsi[L::VT + 8 * i + 6] = ring_row_floats<STRIDE>(RW[i]) * 4;
                            ldsi[L::VT + 8 * i + 7] = lds_base_bytes(lds) + 4 * base;
                        }
                    }
                    base += ring_row_floats<STRIDE>(RW[i]) * (RH[i] + 1);
                }
                // ---- staging items of every step: window(k) minus window(k-1) as <= 4 rectangles per
                // view, cut into items of 64 positions (all CH channels of the group).
                int cnt[NSRC][4], rx[NSRC][4], ry[NSRC][4], rwid[NSRC][4], nel[NSRC][4];
                int npos = 0;   // positions this step stages
#pragma unroll
                for (int i = 0; i < NSRC; ++i) {
                    const int xa = wx0[i], xb = wx0[i] + ww[i] - 1, ya = wy0[i], yb = wy0[i] + wh[i] - 1;
                    const int pxa = __shfl_up(wx0[i], 1), pya = __shfl_up(wy0[i], 1);
                    const int pxb = pxa + __shfl_up(ww[i], 1) - 1, pyb = pya + __shfl_up(wh[i], 1) - 1;
                    int ix0 = 1, ix1 = 0, iy0 = 1, iy1 = 0;  // intersection with the previous window
                    if (lane > 0) { ix0 = max(xa, pxa); ix1 = min(xb, pxb); iy0 = max(ya, pya); iy1 = min(yb, pyb); }
                    const bool ov = (ix0 <= ix1) && (iy0 <= iy1);
                    // left | right | top | bottom; without overlap "left" is the whole window
                    int rh_[4];
                    rx[i][0] = xa;      ry[i][0] = ya;      rwid[i][0] = (ov ? ix0 - 1 : xb) - xa + 1; rh_[0] = wh[i];
                    rx[i][1] = ix1 + 1; ry[i][1] = ya;      rwid[i][1] = ov ? xb - ix1 : 0;             rh_[1] = wh[i];
                    rx[i][2] = ix0;     ry[i][2] = ya;      rwid[i][2] = ov ? ix1 - ix0 + 1 : 0;        rh_[2] = ov ? iy0 - ya : 0;
                    rx[i][3] = ix0;     ry[i][3] = iy1 + 1; rwid[i][3] = ov ? ix1 - ix0 + 1 : 0;        rh_[3] = ov ? yb - iy1 : 0;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        nel[i][r] = act ? max(rwid[i][r], 0) * max(rh_[r], 0) : 0;
                        cnt[i][r] = (nel[i][r] + 63) >> 6;
                        npos += nel[i][r];
                    }
                }
                int nrect = 0;
#pragma unroll
                for (int i = 0; i < NSRC; ++i)
#pragma unroll
                    for (int r = 0; r < 4; ++r) nrect += cnt[i][r] > 0 ? 1 : 0;
                int incl = nrect;
#pragma unroll
                for (int sft = 1; sft < 64; sft <<= 1) {
                    int o = __shfl_up(incl, sft);
                    if (lane >= sft) incl += o;
                }
                const int total_rects = __shfl(incl, 63);
                // (field widths of the descriptors: a step stages < 65536 positions, windows are < 4096 wide)
                if (total_rects > MAXRECTS || !dims_ok || __any(npos > 0xffff) || w > 0xfff0 || h > 0xfff0) {
                    mode = 0;  // cannot describe the staging work: gather from global memory instead
                } else {
                    int idx = incl - nrect;
                    if (act) ldsi[L::SST + lane] = idx;
                    if (lane == nsteps - 1) ldsi[L::SST + nsteps] = incl;
                    int pstart = 0;
#pragma unroll
                    for (int i = 0; i < NSRC; ++i)
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            if (cnt[i][r] > 0) {
                                int* e = ldsi + L::RECTS + 4 * idx;
                                e[0] = (rx[i][r] + 1) | ((ry[i][r] + 1) << 16);   // windows start at >= -1
                                e[1] = rwid[i][r] | (i << 12) | (pstart << 16);
                                e[2] = nel[i][r];
                                // ring coordinates of the rect origin (unwrapped: < 2*RW, 2*RH)
                                e[3] = (rx[i][r] - wx0[i] + oxv[i]) | ((ry[i][r] - wy0[i] + oyv[i]) << 16);
                                pstart += nel[i][r];
                                ++idx;
                            }
                    if (act) ldsi[L::STOT + lane] = pstart;
                }
            }
        }
        if (lane == 0) {
            ldsi[L::HDR + 0] = mode;
            ldsi[L::HDR + 1] = sp_sel;
            ldsi[L::HDR + 2] = nsteps_sel;
        }
        for (int i = lane; i < 2 * MAXSTEPS; i += 64) ldsi[L::CNT + i] = 0;
        if (lane < 2) ldsi[L::HDR + 28 + lane] = 0;   // plane queues of the pixel waves (D3D_FREERUN)
    };
    if (wave == 0) plan_tables(0, NCAND1);
    __syncthreads();
    if (rfl(ldsi[L::HDR + 0]) == 0) {  // (workgroup-uniform)
        __syncthreads();  // everyone has read the header before wave 0 rewrites it
        for (int wk = wave; wk < NSRC; wk += NWAVES) candidate_windows(NCAND1, wk);
        __syncthreads();
        if (wave == 0) plan_tables(NCAND1, NCAND);
        __syncthreads();
    }
    if (D3D_STAMPS && a.tstats && tid == 0) {  // prologue phases: depth range | windows of every candidate | plan + rectangle tables
        const long long t_p2 = clock64();
        atomicAdd(a.tstats + 16, (unsigned long long)(t_range - t_start));
        atomicAdd(a.tstats + 17, (unsigned long long)(t_p1 - t_range));
        atomicAdd(a.tstats + 18, (unsigned long long)(t_p2 - t_p1));
    }
    const int ring = rfl(ldsi[L::HDR + 0]);
    const int SP = rfl(ldsi[L::HDR + 1]);  // planes per step
    const int nsteps = rfl(ldsi[L::HDR + 2]);
    if (D3D_STAMPS && a.stats && tid == 0) {
        atomicAdd(a.stats + (ring ? 0 : 1), 1u);
        atomicAdd(a.stats + 2, (unsigned)SP);
        atomicAdd(a.stats + 3, (unsigned)nsteps);
        if (ring) {
            unsigned tot = 0;
            for (int i = 0; i < NSRC; ++i) tot += (ldsi[L::HDR + 4 + 4 * i] + 1) * (ldsi[L::HDR + 5 + 4 * i] + 1);
            atomicAdd(a.stats + 4, tot);
        }
    }
    // ---------------------------------------------------------------------------------------
    // Delta staging: step k owns the item descriptors [sst[k], sst[k+1]); wave wv takes items
    // wv, wv + NWAVES, ...; the first PFD of them are prefetched one step ahead.
    // ---------------------------------------------------------------------------------------
    struct Item {
        const float* g;  // global address of the group's first channel (always dereferenceable)
        int meta;        // LDS float index of the ring slot | ok << 20 | dup column << 21 | dup row << 22, or -1
        int geo;         // RW | RH << 8 of the view's ring
    };
    long long fdelta[NSRC];  // byte distance from the first source map to source map i (wave-uniform)
#pragma unroll
    for (int i = 0; i < NSRC; ++i) {
        const long long d0 = reinterpret_cast<long long>(p.feats[1]);
        const long long di = reinterpret_cast<long long>(p.feats[i + 1]);
        fdelta[i] = (i < p.n_src) ? di - d0 : 0;
    }
    // Element `el` (0..T-1) of step k's flattened delta -> source address and ring slot of this lane.
    // Each lane finds its own rectangle (<= 16 per step), so items are densely packed.
    auto item = [&](int el, int Tn, int rb, const int (&pst16)[MAXRS]) -> Item {
        Item I;
        const bool live = el < Tn;
        // rect holding element el: count the rect starts <= el (starts beyond the step's rects are INT_MAX)
        int ridx = rb - 1;
#pragma unroll
        for (int rr = 0; rr < MAXRS; ++rr) ridx += (pst16[rr] <= el) ? 1 : 0;
        ridx = max(ridx, rb);
        const int* e = ldsi + L::RECTS + 4 * ridx;
        const int e0 = e[0], e1 = e[1], e6 = e[3];
        const int rx = (e0 & 0xffff) - 1, ry = (e0 >> 16) - 1;
        const int rwid = e1 & 0xfff, vi = (e1 >> 12) & 0xf, pst = (unsigned)e1 >> 16;
        const int e7 = ldsi[L::HDR + 4 + 4 * vi + 3];   // ring geometry of the rect's view
        const int RWv = e7 & 0xff, RHv = (e7 >> 8) & 0xff, bs = (e7 >> 16) << 2;
        const int pos = el - pst;
        const int cy = (int)(((float)pos + 0.5f) * __builtin_amdgcn_rcpf((float)max(rwid, 1)));
        const int cx = pos - cy * rwid;
        const int sx = rx + cx, sy = ry + cy;
        const bool ok = live && ((unsigned)sx < (unsigned)w) && ((unsigned)sy < (unsigned)h);
        // per-lane source view: feats[1] + (uniform) byte distance to view vi's map.  Kept as integer
        // arithmetic on purpose: a select chain over the pointers is turned back into an indexed load
        // of the argument block, whose s_waitcnt vmcnt(0) would serialise the staging loads.
        long long dsel = 0;
#pragma unroll
        for (int i = 1; i < NSRC; ++i) dsel = (vi == i) ? fdelta[i] : dsel;
        if (a.cl)  // one position's CH channels are contiguous: 16-byte loads
            I.g = a.cl + ((size_t)(vi * a.ngroups + grp) * plane + (size_t)(ok ? sy * w + sx : 0)) * CW;
        else
            I.g = reinterpret_cast<const float*>(reinterpret_cast<const char*>(p.feats[1]) + dsel) + (size_t)c0 * plane +
                  (ok ? sy * w + sx : 0);
        unsigned c = (unsigned)((e6 & 0xffff) + cx), r = (unsigned)((e6 >> 16) + cy);
        c = min(c, c - (unsigned)RWv);
        r = min(r, r - (unsigned)RHv);
        const int dst = bs + (int)r * ring_row_floats<STRIDE>(RWv) + (int)c * STRIDE;
        I.meta = live ? (dst | (ok ? 1 << 20 : 0) | (c == 0 ? 1 << 21 : 0) | (r == 0 ? 1 << 22 : 0)) : -1;
        I.geo = e7 & 0xffff;
        return I;
    };
    auto put = [&](int meta, int geo, int q, f4 v) {
        if (meta >= 0) {
            const int dst = (meta & 0xfffff) + 4 * q;
            const int RWv = geo & 0xff, RHv = geo >> 8;
            const int dupc = (meta >> 21) & 1 ? RWv * STRIDE : 0;
            const int dupr = (meta >> 22) & 1 ? RHv * ring_row_floats<STRIDE>(RWv) : 0;
            f4 x = (meta >> 20) & 1 ? v : (f4){0, 0, 0, 0};
            *reinterpret_cast<f4*>(lds + dst) = x;
            if (dupc) *reinterpret_cast<f4*>(lds + dst + dupc) = x;
            if (dupr) *reinterpret_cast<f4*>(lds + dst + dupr) = x;
            if (dupc && dupr) *reinterpret_cast<f4*>(lds + dst + dupc + dupr) = x;
        }
    };
    f4 pf[PFD][Q];
    int pmeta[PFD], pgeo[PFD];
    auto write_round = [&]() {
#pragma unroll
        for (int j = 0; j < PFD; ++j)
#pragma unroll
            for (int q = 0; q < Q; ++q) put(pmeta[j], pgeo[j], q, pf[j][q]);
    };
    // Loader waves: bring window(k) minus window(k-1) into the rings.
    const bool ltiming = D3D_STAMPS && a.tstats != nullptr && lw == 0;
    long long lt_desc = 0, lt_issue = 0, lt_write = 0, lt_bar = 0;
    // (sw, snw): this wave's index among the snw waves sharing the step's items -- the loaders for the deltas;
    // ALL waves for the initial window, which nothing can overlap (one workgroup per CU): the compute waves
    // would only wait for it.
    auto stage = [&](int k, int sw, int snw) {
        const int rb = rfl(ldsi[L::SST + k]), re = rfl(ldsi[L::SST + k + 1]);
        const int Tn = rfl(ldsi[L::STOT + k]);
        const int nitems = (Tn + 63) >> 6;
        // first positions of the step's (<= 4 per view) rects: one LDS read by MAXRS lanes, then wave-uniform copies
        const int psv = (lane < re - rb) ? (int)((unsigned)ldsi[L::RECTS + 4 * (rb + min(lane, MAXRS - 1)) + 1] >> 16) : 0x7fffffff;
        int pst16[MAXRS];
#pragma unroll
        for (int rr = 0; rr < MAXRS; ++rr) pst16[rr] = __builtin_amdgcn_readlane(psv, rr);
        for (int it0 = sw; it0 < nitems; it0 += snw * PFD) {
            long long tq1 = 0, tq2 = 0;
            if (ltiming) tq1 = clock64();
#pragma unroll
            for (int j = 0; j < PFD; ++j) {
                const int it = it0 + snw * j;
                Item I = item(it < nitems ? it * 64 + lane : Tn, Tn, rb, pst16);
                if (a.cl) {
#pragma unroll
                    for (int q = 0; q < Q; ++q) pf[j][q] = reinterpret_cast<const f4*>(I.g)[q];
                } else if (!F16) {   // planar fp32 maps (no workspace given); fp16 rings are only launched with the channel-last copy
#pragma unroll
                    for (int q = 0; q < Q; ++q) {
                        const float* __restrict__ g = I.g + (size_t)(4 * q) * plane;
                        pf[j][q][0] = g[0];
                        pf[j][q][1] = g[plane];
                        pf[j][q][2] = g[2 * plane];
                        pf[j][q][3] = g[3 * plane];
                    }
                }
                pmeta[j] = I.meta;
                pgeo[j] = I.geo;
            }
            if (ltiming) { tq2 = clock64(); lt_issue += tq2 - tq1; }
            write_round();
            if (ltiming) lt_write += clock64() - tq2;
        }
    };

    // Barrier k separates {compute step k-1, stage delta k} from {compute step k, stage delta k+1}.
    // Both roles execute exactly nsteps barriers in ring mode and none in gather mode.
#if D3D_DECOUPLE
#error "the counter hand-off experiment predates the channel-group loop (its counters are per step, not per pass)"
#endif
    // (In gather mode -- rings do not fit, or p.z <= 0 at a corner -- there is nothing to stage: the loader waves sweep
    //  planes too, see GATHER_ALL below.)
    if (loader && ring) {
        // (Raising the loaders' issue priority with s_setprio was measured: their decode time halves,
        // but the compute waves lose the same slots and the kernel gets 5 % slower -- left at default.)
        for (int gi = 0; gi < a.ngroups; ++gi) {
        grp = gi;
        c0 = gi * CH;
        if (FREERUN && ring) {
            stage(0, wave, NWAVES);
            __syncthreads();   // start of the pass: window 0 is complete, the counters are clear
            for (int k = 1; k < nsteps; ++k) {
                // delta k may land on slots of window k-2 (never of window k-1): every plane of step k-2 has to be swept
                if (k >= 2) ctr_wait(ldsi + L::CNT + (k - 2), min(SP, nplanes - (k - 2) * SP) * NPIXW);
                stage(k, lw, NLOADW);
                ctr_add(ldsi + L::CNT + MAXSTEPS + k, lane);   // (after this wave's ring writes, in program order)
            }
        } else if (ring) {
            stage(0, STAGE0_ALL_EFF ? wave : lw, STAGE0_ALL_EFF ? NCOMP + NLOADW : NLOADW);
#if D3D_DECOUPLE
            step_signal(ldsi + L::CNT + 0, lane);
            for (int k = 0; k + 1 < nsteps; ++k) {
                long long tb = 0;
                if (ltiming) tb = clock64();
                if (k > 0) step_wait(ldsi + L::CNT + MAXSTEPS + (k - 1), NCOMP);  // every compute wave has left step k-1
                if (ltiming) lt_bar += clock64() - tb;
                stage(k + 1, lw, NLOADW);
                step_signal(ldsi + L::CNT + (k + 1), lane);
            }
#else
            for (int k = 0; k < nsteps; ++k) {
                long long tb = 0;
                if (ltiming) tb = clock64();
                __syncthreads();
                if (ltiming) lt_bar += clock64() - tb;
                if (k + 1 < nsteps) stage(k + 1, lw, NLOADW);
            }
#endif
        }
        __syncthreads();  // end of the pass: the rings may be overwritten with the next group's first window
        }
        if (ltiming && lane == 0) {
            atomicAdd(a.tstats + 2, (unsigned long long)lt_bar);
            atomicAdd(a.tstats + 6, (unsigned long long)lt_issue);
            atomicAdd(a.tstats + 7, (unsigned long long)lt_write);
        }
        return;
    }

    // --- per-lane constants: rays, reference features, weights -------------------------------
    if constexpr (!EARLY_INPUTS) {
        projection_rows();
        lane_inputs();
    } else {
        if (loader) lane_inputs();   // (gather mode: the loader waves sweep planes, and skipped their inputs above)
    }
    RingView RV[NSRC];
    const int lds0 = lds_base_bytes(lds);
    const float umax = (float)w, vmax = (float)h;   // geo_ring's clamp bounds
#pragma unroll
    for (int i = 0; i < NSRC; ++i) {
        if (ring && i >= p.n_src) {   // unused view of the template: every sample at (-1, -1), see the planner
            ray[i].rx = 0.0f; ray[i].ry = 0.0f; ray[i].rz = 0.0f;
            T0[i] = -1.0f; T1[i] = -1.0f; T2[i] = 1.0f;
        }
        RV[i].RW = ring ? rfl(ldsi[L::HDR + 4 + 4 * i + 0]) : 1;
        RV[i].RH = ring ? rfl(ldsi[L::HDR + 4 + 4 * i + 1]) : 1;
        RV[i].rowb = ring_row_floats<STRIDE>(RV[i].RW) * 4;
        RV[i].base = lds0 + 4 * (ring ? rfl(ldsi[L::HDR + 4 + 4 * i + 2]) : L::DATA);
#ifdef D3D_RV_VGPR   // experiment: the ring constants as (uniform) vector registers instead of spilled scalar registers
        asm volatile("" : "+v"(RV[i].RW), "+v"(RV[i].RH), "+v"(RV[i].rowb), "+v"(RV[i].base));
#endif
    }
    const float invV = 1.0f / (float)(p.n_src + 1);
    const size_t cstride_b = (p.plane_major ? plane : (size_t)D * plane) * sizeof(T);   // bytes between channels
    const size_t cl_step = p.out_cl == 2 ? plane * 16 : 16;

    // all CH/4.. channels of one quad: accumulators -> output values -> stores
    unsigned long long even_quad = 0;   // OUTCL: the packed even quad waits for the odd one (one 16-byte store per pair)
    auto finalize_store = [&](const f4& s, const f4& qq, unsigned long long& ob, int q) {
        f4 o;
        if (MODE == MODE_VARIANCE) {   // two channels per instruction; each half is the scalar sequence m = s/V, fma(qq, 1/V, -(m*m))
            const f2 iv = {invV, invV};
            const f2 ml = lo2(s) * iv, mh = hi2(s) * iv;
            o = cat2(pk_fma(lo2(qq), iv, -(ml * ml)), pk_fma(hi2(qq), iv, -(mh * mh)));
        } else if (MODE == MODE_WEIGHTED) {
            o = s * rden;
        } else {
            o = s;
        }
        if constexpr (OUTCL) {
            if ((q & 1) == 0) {
                even_quad = pack_h16x4(o);
            } else {
                store_sbase_h16x8(ob, pixo, even_quad, pack_h16x4(o));
                ob += cl_step;   // the next 8 channels: the next 16 bytes of the cell, or the next group plane (CL8)
            }
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                store_sbase(ob, pixb, o[k]);
                ob += cstride_b;
            }
        }
    };
    auto accumulate = [&](f4& s, f4& qq, float& pair_acc, const f4& val, int q, int i) {
        if (MODE == MODE_VARIANCE) {
            const f2 vl = lo2(val), vh = hi2(val);
            s = cat2(lo2(s) + vl, hi2(s) + vh);
            qq = cat2(pk_fma(vl, vl, lo2(qq)), pk_fma(vh, vh, hi2(qq)));
        } else if (MODE == MODE_WEIGHTED) {
#pragma unroll
            for (int k = 0; k < 4; ++k) s[k] = fmaf(val[k] * r[F16 ? 0 : q][k], vw[i], s[k]);
        } else if (MODE == MODE_PAIR) {
#pragma unroll
            for (int k = 0; k < 4; ++k) pair_acc = fmaf(r[F16 ? 0 : q][k], val[k], pair_acc);
        } else {
            s = val;
        }
    };

    // ================================ main loop over steps ======================================
    const bool timing = D3D_STAMPS && a.tstats != nullptr && wave == 0;
    const bool wtiming = D3D_STAMPS && a.tstats != nullptr;  // every compute wave: its own barrier wait
    long long t_ww = 0;
    long long t_w = 0, t_w0 = 0, t_c = 0, t_mark = 0;
    if (timing) { t_mark = clock64(); if (lane == 0) atomicAdd(a.tstats + 0, (unsigned long long)(t_mark - t_start)); }
    // (s_setprio 2 / 3 on ALL compute waves: no effect, 6.49 / 6.46 ms next to 6.51 / 6.47 ms without)
#ifdef D3D_YOUNG_PRIO
    // Issue arbitration is priority, then age: of the two compute waves of a SIMD the older one (waves 0-3) wins every
    // contested slot, finishes its planes of a step early and idles at the step barrier (~95k of 427k cycles per workgroup)
    // while the younger one runs alone.  Static priority for the younger half evens the two out.
    if (wave >= NCOMP / 2) __builtin_amdgcn_s_setprio(D3D_YOUNG_PRIO);
#endif
    // ---- one plane of the segment (dl_ = plane index within the segment) for this wave's pixels and the current group ----
    int kx[NSRC], ky[NSRC];   // ring offset minus window origin of the current step, per view (scalar registers)
    int plan_adr = 0;   // view-major: absolute LDS byte address of the current step's plan entries
    auto load_step = [&](int k) {
        if constexpr (VIEW_MAJOR) {
            plan_adr = lds_base_bytes(lds) + 4 * (L::PLAN + (ring ? k : 0) * NSRC * 8);
            return;
        }
#pragma unroll
        for (int i = 0; i < NSRC; ++i) {
            const int* e = ldsi + L::PLAN + ((ring ? k : 0) * NSRC + i) * 8;
            kx[i] = rfl(e[6]);
            ky[i] = rfl(e[7]);
        }
    };
    auto sweep_plane = [&](const int dl_) {
            const int d = ds + dl_;
            float dv;
            if (p.depth_mode == D3D_DEPTH_PER_PIXEL) {
                float t = p.depth[(size_t)d * plane + pix];
                dv = valid ? t : 1.0f;
            } else if (p.depth_mode == D3D_DEPTH_AFFINE) {
                dv = __fadd_rn(aff_lo, __fmul_rn((float)d, aff_step));   // (no load in the plane loop)
            } else {
                dv = lds[L::PMIN + dl_];
            }
            unsigned long long ob = OUTCL ? uniform64(reinterpret_cast<unsigned short*>(p.out) +
                                                      (p.out_cl == 2 ? ((size_t)d * (p.C / 8) + c0 / 8) * plane * 8 : (size_t)d * plane * p.C + c0))
                                          : uniform64(reinterpret_cast<T*>(p.out) + (p.plane_major ? (size_t)d * p.C + c0 : (size_t)c0 * D + d) * plane);  // scalar base, once per plane
            float pair_acc = 0.0f;
            if (!valid) return;  // one EXEC region per plane instead of one branch per store

            if constexpr (VIEW_MAJOR) {
                // ---- view-major order (fp16 storage, or more than four source views) -----------------------------
                // The geometry of ONE view is live at a time (the next view's is computed while this view's taps are
                // in flight) and the accumulators of all Q chunks stay in registers until the last view: with six
                // views the chunk-major order of the fp32 / four-view path keeps 6 x (2 addresses + 4 weights) alive
                // and spills, and a scratch reload inside the plane loop waits (vmcnt) for every store in flight.
                float sa[Q][CPC], qa[Q][CPC];
#pragma unroll
                for (int q = 0; q < Q; ++q)
#pragma unroll
                    for (int k = 0; k < CPC; ++k) {
                        const float rv = F16 ? rh[F16 ? q : 0][F16 ? k : 0] : r[F16 ? 0 : q][F16 ? 0 : k];
                        sa[q][k] = (MODE == MODE_VARIANCE) ? rv : 0.0f;
                        qa[q][k] = rv * rv;
                    }
                auto accumulate_vm = [&](const float (&val)[CPC], int q, int i) {
#pragma unroll
                    for (int k = 0; k < CPC; k += 2) {
                        const f2 v = {val[k], val[k + 1]};
                        if (MODE == MODE_VARIANCE) {
                            const f2 sn = (f2){sa[q][k], sa[q][k + 1]} + v;
                            const f2 qn = pk_fma(v, v, (f2){qa[q][k], qa[q][k + 1]});
                            sa[q][k] = sn[0]; sa[q][k + 1] = sn[1];
                            qa[q][k] = qn[0]; qa[q][k + 1] = qn[1];
                        } else {   // MODE_WEIGHTED (fp32 only)
                            sa[q][k] = fmaf(v[0] * r[F16 ? 0 : q][F16 ? 0 : k], vw[i], sa[q][k]);
                            sa[q][k + 1] = fmaf(v[1] * r[F16 ? 0 : q][F16 ? 0 : k + 1], vw[i], sa[q][k + 1]);
                        }
                    }
                };
                if (ring) {
                    // Per-view constants (translation, ring size / pitch / base, the step's ring offset) come from LDS -- three
                    // broadcast reads per view and plane.  As scalar registers (17 per view, six views) most of them were spilled:
                    // 104 v_readlane + 85 s_nop in a plane loop of 1093 instructions (config 5's kernel).
                    const int vt_adr = lds0 + 4 * L::VT;
                    auto geometry = [&](int i) -> TapL {
                        typedef int i4v __attribute__((ext_vector_type(4)));
                        typedef int i2v __attribute__((ext_vector_type(2)));
                        const f4 tv = *(volatile lds_f4_ptr)(unsigned)(vt_adr + 32 * i);
                        const i4v rv = *(volatile __attribute__((address_space(3))) const i4v*)(unsigned)(vt_adr + 32 * i + 16);
                        const i2v kk = *(volatile __attribute__((address_space(3))) const i2v*)(unsigned)(plan_adr + 32 * i + 24);
                        RingView R;
                        R.RW = rv[0]; R.RH = rv[1]; R.rowb = rv[2]; R.base = rv[3];
                        return geo_ring<STRIDE>(ray[i], tv[0], tv[1], tv[2], dv, umax, vmax, kk[0], kk[1], R);
                    };
                    auto fetch = [&](const TapL& g, int q, f4 (&dst)[4]) {
                        dst[0] = lds_read4_abs(g.a0 + q * 16);
                        dst[1] = lds_read4_abs(g.a0 + q * 16 + STRIDE * 4);
                        dst[2] = lds_read4_abs(g.a1 + q * 16);
                        dst[3] = lds_read4_abs(g.a1 + q * 16 + STRIDE * 4);
                    };
                    f4 tp[2][4];
                    TapL gc = geometry(0);
                    fetch(gc, 0, tp[0]);
#pragma unroll
                    for (int i = 0; i < NSRC; ++i) {
                        // keep the scheduler from hoisting every view's geometry to the top of the plane (it would: the
                        // views are independent), which is exactly the register pressure this order avoids
                        __builtin_amdgcn_sched_barrier(0);
                        TapL gn = gc;
                        if (i + 1 < NSRC) gn = geometry(i + 1);
#pragma unroll
                        for (int q = 0; q < Q; ++q) {
                            const int u = i * Q + q;
                            if (q + 1 < Q) fetch(gc, q + 1, tp[(u + 1) & 1]);
                            else if (i + 1 < NSRC) fetch(gn, 0, tp[(u + 1) & 1]);
                            f4 (&c)[4] = tp[u & 1];
                            asm volatile("" : "+v"(c[3]));
                            float val[CPC];
                            if constexpr (F16) {
                                // same summation order as grid_sample: nw, ne, sw, se -- tap by tap ACROSS the eight channels, so that
                                // consecutive instructions are independent: the compiler puts an s_nop between an inline-asm
                                // instruction and a consumer of its result (152 of them in a loop of 1140 with the channel-major order)
                                const float tw[4] = {gc.nw, gc.ne, gc.sw, gc.se};
#pragma unroll
                                for (int k = 0; k < CPC; ++k) val[k] = 0.0f;
#pragma unroll
                                for (int tp_ = 0; tp_ < 4; ++tp_) {
#pragma unroll
                                    for (int k = 0; k < CPC; k += 2) {
                                        const int wi = k >> 1;           // word wi of a tap holds channels k (low half) and k + 1
                                        val[k] = fma_mix_lo(c[tp_][wi], tw[tp_], val[k]);
                                        val[k + 1] = fma_mix_hi(c[tp_][wi], tw[tp_], val[k + 1]);
                                    }
                                    if (tp_ < 3) __builtin_amdgcn_sched_barrier(0);   // (keep the order: the scheduler would re-chain them)
                                }
                            } else {
                                const f4 v4 = blend(c[0], c[1], c[2], c[3], gc.nw, gc.ne, gc.sw, gc.se);
#pragma unroll
                                for (int k = 0; k < CPC; ++k) val[k] = v4[k];
                            }
                            accumulate_vm(val, q, i);
                        }
                        gc = gn;
                    }
                } else {   // fallback workgroups: taps from the planar maps in global memory, one channel at a time
                    for (int i = 0; i < p.n_src; ++i) {
                        const float* __restrict__ M = p.proj34 + 12 * i;
                        const Ray rr = make_ray(M, xf, yf);
                        float u, v;
                        project(rr, M[3], M[7], M[11], dv, h, w, u, v);
                        const TapG t = make_tap_glb(u, v, h, w);
                        const T* __restrict__ gp = reinterpret_cast<const T*>(p.feats[i + 1]) + (size_t)c0 * plane + t.off;
#pragma unroll
                        for (int q = 0; q < Q; ++q) {
                            float val[CPC];
#pragma unroll
                            for (int k = 0; k < CPC; ++k) {
                                const T* __restrict__ gk = gp + (size_t)(CPC * q + k) * plane;
                                val[k] = fmaf(ldf(gk + t.dyw + t.dx), t.se, fmaf(ldf(gk + t.dyw), t.sw, fmaf(ldf(gk + t.dx), t.ne, ldf(gk) * t.nw)));
                            }
                            accumulate_vm(val, q, i);
                        }
                    }
                }
#pragma unroll
                for (int q = 0; q < Q; ++q)
#pragma unroll
                    for (int k = 0; k < CPC; ++k) {
                        float o;
                        if (MODE == MODE_VARIANCE) {
                            const float m = sa[q][k] * invV;
                            o = fmaf(qa[q][k], invV, -(m * m));
                        } else {
                            o = sa[q][k] * rden;
                        }
                        if constexpr (F16) store_sbase_h(ob, pixb, o);
                        else store_sbase(ob, pixb, o);
                        ob += cstride_b;
                    }
                return;
            }

            if (ring) {
                TapL t[NSRC];
#pragma unroll
                for (int i = 0; i < NSRC; ++i) {
                    t[i] = geo_ring<STRIDE>(ray[i], T0[i], T1[i], T2[i], dv, umax, vmax, kx[i], ky[i], RV[i]);
                }
                // Units u = (quad q, view i) in q-major order; the four taps of unit u+1 are requested
                // before unit u is blended, so LDS latency overlaps the 24 VALU ops of a unit.
                constexpr int NU = Q * NSRC;
                constexpr int PD = LDS_PIPE;  // units requested ahead of the one being blended
                f4 tp[PD + 1][4];
                auto request = [&](int u, f4 (&dst)[4]) {
                    const int q2 = u / NSRC, i2 = u % NSRC;
                    dst[0] = lds_read4_abs(t[i2].a0 + q2 * 16);
                    dst[1] = lds_read4_abs(t[i2].a0 + q2 * 16 + STRIDE * 4);
                    dst[2] = lds_read4_abs(t[i2].a1 + q2 * 16);
                    dst[3] = lds_read4_abs(t[i2].a1 + q2 * 16 + STRIDE * 4);
                };
#pragma unroll
                for (int u = 0; u < PD && u < NU; ++u) request(u, tp[u % (PD + 1)]);
                if (PD == 0) request(0, tp[0]);
                f4 s, qq;
#pragma unroll
                for (int u = 0; u < NU; ++u) {
                    const int q = u / NSRC, i = u % NSRC;
                    if (PD > 0 ? (u + PD < NU) : (u > 0)) request(u + PD, tp[(u + PD) % (PD + 1)]);
                    if (i == 0) {
                        if (MODE == MODE_VARIANCE) { s = r[F16 ? 0 : q]; qq = s * s; }
                        else { s = (f4){0, 0, 0, 0}; qq = s; }
                    }
                    f4 (&c)[4] = tp[u % (PD + 1)];
                    // LDS returns in order: touching the last-requested tap first makes the compiler emit ONE
                    // s_waitcnt for the unit instead of one per tap
                    asm volatile("" : "+v"(c[3]));
                    f4 val = blend(c[0], c[1], c[2], c[3], t[i].nw, t[i].ne, t[i].sw, t[i].se);
                    accumulate(s, qq, pair_acc, val, q, i);
                    if (i == NSRC - 1 && MODE != MODE_PAIR) finalize_store(s, qq, ob, q);
                }
            } else {
                TapG t[NSRC];
#pragma unroll
                for (int i = 0; i < NSRC; ++i) {
                    float u, v;
                    project(ray[i], T0[i], T1[i], T2[i], dv, h, w, u, v);
                    t[i] = make_tap_glb(u, v, h, w);
                    if (i >= p.n_src) { t[i].nw = t[i].ne = t[i].sw = t[i].se = 0.0f; t[i].off = 0; t[i].dx = 0; t[i].dyw = 0; }
                }
#pragma unroll
                for (int q = 0; q < Q; ++q) {
                    f4 s, qq;
                    if (MODE == MODE_VARIANCE) { s = r[F16 ? 0 : q]; qq = s * s; }
                    else { s = (f4){0, 0, 0, 0}; qq = s; }
#pragma unroll
                    for (int i = 0; i < NSRC; ++i) {
                        const float* __restrict__ g = p.feats[min(i + 1, p.n_src)] + (size_t)(c0 + 4 * q) * plane + t[i].off;
                        f4 val;
#pragma unroll
                        for (int kk = 0; kk < 4; ++kk) {
                            const float* __restrict__ gk = g + (size_t)kk * plane;
                            val[kk] = fmaf(gk[t[i].dyw + t[i].dx], t[i].se,
                                           fmaf(gk[t[i].dyw], t[i].sw, fmaf(gk[t[i].dx], t[i].ne, gk[0] * t[i].nw)));
                        }
                        accumulate(s, qq, pair_acc, val, q, i);
                    }
                    if (MODE != MODE_PAIR) finalize_store(s, qq, ob, q);
                }
            }
            if (MODE == MODE_PAIR)
                store_sbase(uniform64(p.out + (size_t)d * plane), pixb, pair_acc / (float)CH);
    };
    for (int gi = 0; gi < a.ngroups; ++gi) {
    grp = gi;
    c0 = gi * CH;
    if (gi > 0) load_reference();   // (group 0: requested ahead of the planning)
    if (FREERUN && ring) {   // (see D3D_FREERUN)
        stage(0, wave, NWAVES);
        __syncthreads();   // start of the pass: window 0 is complete, the counters are clear
        int kcur = 0;
        load_step(0);
        for (;;) {
            int pg = 0;
            if (lane == 0) pg = __hip_atomic_fetch_add(ldsi + L::HDR + 28 + pw, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            pg = rfl(pg);
            if (pg >= nplanes) break;
            if (pg >= (kcur + 1) * SP) {   // first plane this wave takes of a later step
                while (pg >= (kcur + 1) * SP) ++kcur;
                ctr_wait(ldsi + L::CNT + MAXSTEPS + kcur, NLOADW);
                load_step(kcur);
            }
            sweep_plane(pg);
            ctr_add(ldsi + L::CNT + kcur, lane);   // (after this plane's tap reads, in program order)
        }
        __syncthreads();   // end of the pass (matches the loaders'): the rings may be overwritten
        if (wave == 0) {
            for (int i = lane; i < 2 * MAXSTEPS; i += 64) ldsi[L::CNT + i] = 0;
            if (lane < 2) ldsi[L::HDR + 28 + lane] = 0;
        }
        continue;
    }
    if (!ring) {
        // GATHER_ALL: taps straight from global memory, no windows, no steps, no barriers inside the pass: every wave of the
        // workgroup (the loader waves too) takes planes sub, sub + NW, ... of its pixel rows
        constexpr int NW = NWAVES / NPIXW;
        static_assert(NWAVES % NPIXW == 0, "the loader waves must come in whole sets of pixel waves");
        load_step(0);
        for (int dl_ = wave / NPIXW; dl_ < nplanes; dl_ += NW) sweep_plane(dl_);
        __syncthreads();   // end of the pass (uniform with the ring passes of other workgroups: one barrier per pass)
        continue;
    }
#if STAGE0_ALL_EFF
    if (ring) stage(0, wave, NCOMP + NLOADW);  // first window of the pass: every wave of the workgroup stages
#endif
    for (int k = 0; k < nsteps; ++k) {
        if (ring) {
            long long ta = 0;
            if (timing) ta = clock64();
            long long tw0 = 0;
            if (wtiming) tw0 = clock64();
            __syncthreads();  // barrier k: rings hold window(k)
            if (wtiming && k > 0) t_ww += clock64() - tw0;
            if (timing) { t_mark = clock64(); t_w += t_mark - ta; if (k == 0) t_w0 = t_mark - ta; }
        }
        load_step(k);

#if D3D_GRAB
        static_assert(NPIXW <= 2, "one plane counter per step and pixel wave: CNT holds two rows of MAXSTEPS");
        const int spk = min(SP, nplanes - k * SP);   // planes of this step
        for (;;) {
            int jg = 0;
            if (lane == 0) jg = __hip_atomic_fetch_add(ldsi + L::CNT + pw * MAXSTEPS + k, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            jg = rfl(jg);
            if (jg >= spk) break;
            const int dl_ = k * SP + jg;
#else
        for (int j = 0; sub + NSUB * j < SP; ++j) {
            const int dl_ = k * SP + sub + NSUB * j;
            if (dl_ >= nplanes) break;
#endif
            sweep_plane(dl_);
        }
        if (timing) t_c += clock64() - t_mark;
        // no loader waves (NLOAD_T = 0): every wave stages its share of the next step's delta once its own planes are done
        // (the delta's ring slots are not part of the window the other waves are still reading)
        if constexpr (NLOADW == 0) {
            if (ring && k + 1 < nsteps) stage(k + 1, wave, NWAVES);
        }
    }
    __syncthreads();  // end of the pass (matches the loaders')
#if D3D_GRAB
    // every wave has left the last step: clear the plane counters for the next pass (its first grab follows a barrier)
    if (wave == 0) for (int i = lane; i < 2 * MAXSTEPS; i += 64) ldsi[L::CNT + i] = 0;
#endif
    if (timing) t_mark = clock64();
    }
    if (wtiming && lane == 0) atomicAdd(a.tstats + 8 + wave, (unsigned long long)t_ww);
    if (timing && lane == 0) {
        atomicAdd(a.tstats + 1, (unsigned long long)t_w);
        atomicAdd(a.tstats + 5, (unsigned long long)t_w0);  // wait for the initial window (first barrier)
        atomicAdd(a.tstats + 3, (unsigned long long)t_c);
        atomicAdd(a.tstats + 4, (unsigned long long)(clock64() - t_start));
    }
}

// ---------------------------------------------------------------------------------------------
// Channel-last copy of the source maps for the loaders: [view][group][position][CH].  A loader lane then fetches the
// CH channels of a ring position with 16-byte loads instead of CH element loads from CH different planes (the
// staging waves are bound by the number of VMEM instructions they issue, not by bytes).  One extra read + write of
// the source maps per launch (0.33 GB next to 15.7 GB of output at config 2).  The copy lives in the CALLER's
// workspace (d3d_sweep_workspace_bytes); without one the fp32 loaders read the planar maps directly.
struct PackArgs {
    const void* src[D3D_MAX_VIEWS];
};
template <int CH, typename T>
__global__ __launch_bounds__(256) void pack_channel_last_kernel(PackArgs pa, int ngroups, long plane, float* __restrict__ out) {
    // thread = (position, 16-byte chunk): consecutive lanes write the contiguous bytes of one position;
    // for a fixed channel, every Q-th lane reads consecutive positions
    constexpr int CW = CH * (int)sizeof(T) / 4, Q = CW / 4, CPC = 16 / (int)sizeof(T);
    const long t = (long)blockIdx.x * 256 + threadIdx.x;
    const long pos = t / Q;
    const int q = (int)(t - pos * Q);
    if (pos >= plane) return;
    const int vg = blockIdx.y;  // view * ngroups + group
    const int v = vg / ngroups, g = vg - v * ngroups;
    const T* __restrict__ s = reinterpret_cast<const T*>(pa.src[v]) + ((size_t)g * CH + CPC * q) * plane + pos;
    f4 x;
    if (sizeof(T) == 4) {
        const float* __restrict__ sf = reinterpret_cast<const float*>(s);
        x[0] = sf[0]; x[1] = sf[plane]; x[2] = sf[2 * plane]; x[3] = sf[3 * plane];
    } else {
        const unsigned short* __restrict__ sh = reinterpret_cast<const unsigned short*>(s);
        unsigned wd[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) wd[k] = (unsigned)sh[(2 * k) * plane] | ((unsigned)sh[(2 * k + 1) * plane] << 16);
        x = __builtin_bit_cast(f4, (uint4){wd[0], wd[1], wd[2], wd[3]});
    }
    reinterpret_cast<f4*>(out + ((size_t)vg * plane + pos) * CW)[q] = x;
}

// The fp32 source maps of a sweep as [view][C / 8][h * w][8] in p.workspace (for the window kernel's gather path, planesweep_window.hip)
int pack_channel_last_g8(const SweepParams& p, hipStream_t stream) {
    PackArgs pa = {};
    for (int i = 0; i < p.n_src; ++i) pa.src[i] = p.feats[i + 1];
    const long plane = (long)p.h * p.w;
    const int ngroups = p.C / 8;
    hipLaunchKernelGGL((pack_channel_last_kernel<8, float>), dim3((unsigned)ceil_div(plane * 2, 256), p.n_src * ngroups), dim3(256), 0, stream, pa,
                       ngroups, plane, reinterpret_cast<float*>(p.workspace));
    D3D_LAUNCH_CHECK("pack_channel_last_kernel launch (window gather path)");
    return D3D_OK;
}

static int group_channels(int C, int n_src, int elem_bytes);

// depth planes per workgroup segment and the number of segments (shared by the launcher and the workspace query)
static void segments(int D, long tiles, int& dseg, int& nseg) {
    int dseg_cap = DSEG_MAX;
#ifdef D3D_EXPERIMENTS
    if (const char* e = getenv("D3D_TILED_DSEG")) dseg_cap = max(NSUB, min(atoi(e), DSEG_MAX));
#endif
    nseg = ceil_div(D, dseg_cap);
    // enough workgroups to fill 256 CUs a few times over; 128-plane segments measured best on config 2 (rings sized
    // per segment follow the depth-dependent window size; shorter segments pay the planning prologue more often)
    while (tiles * nseg < 4096 && nseg * 2 <= ceil_div(D, 32)) nseg *= 2;
    dseg = ceil_div(D, nseg);
    nseg = ceil_div(D, dseg);
}

// The channel-last copy pays off when the sweep is deep: it costs 2 x (n_src*C*h*w) elements of traffic.
#ifndef D3D_CL_MIN_PLANES
#define D3D_CL_MIN_PLANES 96
#endif
static bool wants_channel_last(int D, int elem_bytes) { return elem_bytes == 2 || D >= D3D_CL_MIN_PLANES; }

size_t tiled_workspace_bytes(int n_src, int C, int D, int h, int w, int elem_bytes) {
    if (C % 8 != 0 || n_src > 6 || n_src < 1 || !wants_channel_last(D, elem_bytes)) return 0;
    return (size_t)n_src * C * h * w * elem_bytes;
}

#ifndef D3D_SHALLOW_PLANES
#define D3D_SHALLOW_PLANES 16   // sweeps of at most this many planes take the 32 x 8-pixel patches (0: never)
#endif
// shallow form: 32 x 8-pixel patches (4 pixel waves x 2 depth sub-ranges + 4 loaders, still 12 waves and all of the LDS):
// the per-workgroup planning prologue and first-window latency are paid once per 256 pixels instead of once per 128
constexpr int SHALLOW_NSUB = 2, SHALLOW_NLOADW = D3D_NLOADW, SHALLOW_NPIXW = 4, SHALLOW_PLANES = D3D_SHALLOW_PLANES;

template <int MODE, int NSRC, int CH, typename T, bool OUTCL = false, bool SHALLOW = false>
static int launch_one(const SweepParams& p, hipStream_t stream) {
    constexpr int CW = CH * (int)sizeof(T) / 4;
    using L = Lds<CW, NSRC>;
    constexpr int LDS_BYTES = 160 * 1024;
    constexpr int NSUBK = SHALLOW ? SHALLOW_NSUB : NSUB, NLOADK = SHALLOW ? SHALLOW_NLOADW : NLOADW;
    constexpr int NPIXK = SHALLOW ? SHALLOW_NPIXW : NPIXW, THK = 2 * NPIXK;
    constexpr int THREADSK = 64 * (NPIXK * NSUBK + NLOADK);
    auto kern = sweep_tiled_kernel<MODE, NSRC, CH, T, OUTCL, NSUBK, NLOADK, NPIXK>;
    if (OUTCL && (size_t)p.h * p.w * p.C * 2 >= ((size_t)1 << 32)) return D3D_ERR_UNSUPPORTED;
    // per device and idempotent: set on every launch (no process-global "done" flag that a second GPU would miss)
    int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(kern), LDS_BYTES);
    if (rc != D3D_OK) return rc;
    TiledArgs a;
    a.ngroups = p.C / CH;
    a.tiles_x = ceil_div(p.w, TW);
    a.tiles_y = ceil_div(p.h, THK);
    a.cap_floats = LDS_BYTES / 4 - L::DATA;
    segments(p.D, (long)a.tiles_x * a.tiles_y * a.ngroups, a.dseg, a.nseg);
    const long nblk = (long)a.tiles_x * a.tiles_y * a.nseg;  // a workgroup sweeps every channel group of its patch
    if (nblk > 0x7fffffffL) return D3D_ERR_UNSUPPORTED;
    a.stats = nullptr;
    a.tstats = nullptr;
    a.cl = nullptr;
    {
        bool use_cl = wants_channel_last(p.D, (int)sizeof(T));
#ifdef D3D_EXPERIMENTS
        if (const char* e = getenv("D3D_TILED_CL")) use_cl = atoi(e) != 0;
#endif
        const size_t bytes = (size_t)p.n_src * p.C * p.h * p.w * sizeof(T);
        if (use_cl && p.workspace && p.workspace_bytes >= bytes) {
            PackArgs pa = {};
            for (int i = 0; i < p.n_src; ++i) pa.src[i] = p.feats[i + 1];
            const long plane = (long)p.h * p.w;
            hipLaunchKernelGGL((pack_channel_last_kernel<CH, T>), dim3((unsigned)ceil_div(plane * (CW / 4), 256), p.n_src * a.ngroups),
                               dim3(256), 0, stream, pa, a.ngroups, plane, reinterpret_cast<float*>(p.workspace));
            D3D_LAUNCH_CHECK("pack_channel_last_kernel launch");
            a.cl = reinterpret_cast<const float*>(p.workspace);
        }
        // the fp16 loaders only read the channel-last copy (2-byte planar gathers would be VMEM-issue bound)
        if (sizeof(T) == 2 && !a.cl) return D3D_ERR_UNSUPPORTED;
    }
#ifdef D3D_EXPERIMENTS
    if (getenv("D3D_TILED_STATS")) {  // debug only: synchronous, allocates
        (void)hipMalloc(&a.stats, 8 * sizeof(unsigned) + 24 * sizeof(unsigned long long));
        (void)hipMemset(a.stats, 0, 8 * sizeof(unsigned) + 24 * sizeof(unsigned long long));
        a.tstats = reinterpret_cast<unsigned long long*>(a.stats + 8);
    }
#endif
    hipLaunchKernelGGL(kern, dim3((unsigned)nblk), dim3(THREADSK), LDS_BYTES, stream, p, a);
    D3D_LAUNCH_CHECK("sweep_tiled_kernel launch");
#ifdef D3D_EXPERIMENTS
    if (a.stats) {
        unsigned hs[8];
        unsigned long long ht[24];
        (void)hipMemcpy(hs, a.stats, sizeof(hs), hipMemcpyDeviceToHost);
        (void)hipMemcpy(ht, a.tstats, sizeof(ht), hipMemcpyDeviceToHost);
        fprintf(stderr, "[d3d tiled timing] per-WG mean cycles: compute wave 0: prologue %.0f | barrier wait %.0f (initial window %.0f) | compute %.0f | total %.0f || loader 0: barrier wait %.0f | decode+issue %.0f | wait+write %.0f\n",
                ht[0] / (double)nblk, ht[1] / (double)nblk, ht[5] / (double)nblk, ht[3] / (double)nblk, ht[4] / (double)nblk,
                ht[2] / (double)nblk, ht[6] / (double)nblk, ht[7] / (double)nblk);
        fprintf(stderr, "[d3d tiled timing] prologue phases: depth range %.0f | candidate windows %.0f | plan tables %.0f (rest: per-lane constants)\n",
                ht[16] / (double)nblk, ht[17] / (double)nblk, ht[18] / (double)nblk);
        fprintf(stderr, "[d3d tiled timing] barrier wait after the first, per compute wave (pixel rows, depth sub-range):");
        for (int wv = 0; wv < NPIXK * NSUBK; ++wv) fprintf(stderr, " %.0f", ht[8 + wv] / (double)nblk);
        fprintf(stderr, "\n");
        fprintf(stderr, "[d3d tiled stats] CH=%d elem=%d wgs=%ld ring=%u fallback=%u mean_step_planes=%.2f mean_steps=%.2f mean_ring_positions=%.0f (cap %d) dseg=%d\n",
                CH, (int)sizeof(T), nblk, hs[0], hs[1], hs[2] / (double)nblk, hs[3] / (double)nblk, hs[0] ? hs[4] / (double)hs[0] : 0.0,
                a.cap_floats / L::STRIDE, a.dseg);
        (void)hipFree(a.stats);
    }
#endif
    return D3D_OK;
}

static int group_channels(int C, int n_src, int elem_bytes) {
    // channels per workgroup pass.  Six fp32 rings only fit LDS with 8-channel positions; fp16 cells take 16.
    int cg = (n_src > 4 && elem_bytes == 4) ? 8 : (C >= 16 ? 16 : C);
#ifdef D3D_EXPERIMENTS
    if (const char* e = getenv("D3D_TILED_CG")) cg = atoi(e);   // must divide C
#endif
    if (cg != 8 && cg != 16 && cg != 32) cg = 8;
    while (C % cg) cg >>= 1;
    return cg;
}

template <int MODE, int NSRC>
static int launch_ch(const SweepParams& p, hipStream_t stream) {
    const int cg = (MODE == MODE_PAIR) ? p.C : group_channels(p.C, p.n_src, 4);
    // shallow sweeps of 8-channel groups (the full-resolution cascade stage, 8 planes): 32 x 8-pixel patches (see SHALLOW_*).
    // (A first attempt, 6-wave workgroups on half the LDS so that two share a CU, helped the weighted mode on the bench
    // scene but fell back to gathering inside AdaMVS views, whose windows need more than half the LDS.)
#ifndef D3D_SHALLOW_CG16
#define D3D_SHALLOW_CG16 0      // A/B: the 16-channel groups too
#endif
    bool shallow = p.D <= SHALLOW_PLANES && (cg == 8 || (D3D_SHALLOW_CG16 && cg == 16));
#ifdef D3D_EXPERIMENTS
    if (const char* e = getenv("D3D_TILED_SHALLOW")) shallow = shallow && atoi(e) != 0;
#endif
    if constexpr (MODE == MODE_VARIANCE) {
        if (p.out_cl) {
            if constexpr (SHALLOW_PLANES > 0) {
                if (shallow && cg == 8) return launch_one<MODE, NSRC, 8, float, true, true>(p, stream);
                if (D3D_SHALLOW_CG16 && shallow && cg == 16) return launch_one<MODE, NSRC, 16, float, true, true>(p, stream);
            }
            switch (cg) {
                case 16: return launch_one<MODE, NSRC, 16, float, true>(p, stream);
                case 8: return launch_one<MODE, NSRC, 8, float, true>(p, stream);
            }
            return D3D_ERR_UNSUPPORTED;
        }
    }
    if (p.out_cl) return D3D_ERR_UNSUPPORTED;
    if constexpr (SHALLOW_PLANES > 0 && (MODE == MODE_VARIANCE || MODE == MODE_WEIGHTED)) {
        if (shallow && cg == 8) return launch_one<MODE, NSRC, 8, float, false, true>(p, stream);
        if (D3D_SHALLOW_CG16 && shallow && cg == 16) return launch_one<MODE, NSRC, 16, float, false, true>(p, stream);
    }
    (void)shallow;
    switch (cg) {
        case 32: return launch_one<MODE, NSRC, 32, float>(p, stream);
        case 16: return launch_one<MODE, NSRC, 16, float>(p, stream);
        case 8: return launch_one<MODE, NSRC, 8, float>(p, stream);
    }
    return D3D_ERR_UNSUPPORTED;
}

// fp16 storage (BASELINE config 5): 16-channel groups of 32-byte cells
template <int NSRC>
static int launch_f16(const SweepParams& p, hipStream_t stream) {
    if (p.C % 16 != 0) return D3D_ERR_UNSUPPORTED;
    return launch_one<MODE_VARIANCE, NSRC, 16, __half>(p, stream);
}

// Non-default compile-time knobs of this translation unit (d3d_build_flags): empty for the production build.  The
// timing-only experiments of rounds 1-3 (no stores / no staging / no taps / no blend / no barriers: results wrong by
// construction) are gone from the source; their numbers live in profiles/r03_ab_variants.txt.
const char* tiled_build_flags() {
    return ""
#ifdef D3D_EXPERIMENTS
           " D3D_EXPERIMENTS"
#endif
#ifdef D3D_DEV_ONLY_HEADLINE
           " D3D_DEV_ONLY_HEADLINE"
#endif
#if D3D_NSUB != 4
           " D3D_NSUB"
#endif
#if D3D_NLOADW != 4
           " D3D_NLOADW"
#endif
#if D3D_LDS_PIPE != 1
           " D3D_LDS_PIPE"
#endif
#if D3D_GRAB
           " D3D_GRAB"
#endif
#if D3D_FREERUN
           " D3D_FREERUN"
#endif
#if D3D_DECOUPLE
           " D3D_DECOUPLE"
#endif
#if STAGE0_ALL != 1
           " STAGE0_ALL"
#endif
#if D3D_SHALLOW_PLANES != 16
           " D3D_SHALLOW_PLANES"
#endif
#if D3D_SHALLOW_CG16
           " D3D_SHALLOW_CG16"
#endif
#if D3D_CL_MIN_PLANES != 96
           " D3D_CL_MIN_PLANES"
#endif
#ifdef D3D_YOUNG_PRIO
           " D3D_YOUNG_PRIO"
#endif
#ifdef D3D_RV_VGPR
           " D3D_RV_VGPR"
#endif
        ;
}

int launch_tiled(int mode, const SweepParams& p, hipStream_t stream) {
#ifdef D3D_DEV_ONLY_HEADLINE   // development builds (ISA inspection): instantiate the config-2 kernel alone
    return launch_one<MODE_VARIANCE, 4, 16, float>(p, stream);
#else
    if (p.C % 8 != 0 || (mode == MODE_PAIR && p.C != 32 && p.C != 16 && p.C != 8)) {
        set_error("tiled kernel unsupported: C=%d", p.C);
        return D3D_ERR_UNSUPPORTED;
    }
    if (p.n_src > 6) {
        set_error("tiled kernel unsupported: %d source views (max 6)", p.n_src);
        return D3D_ERR_UNSUPPORTED;
    }
    if (p.elem_bytes == 2) {
        if (mode != MODE_VARIANCE || p.out_cl) return D3D_ERR_UNSUPPORTED;
        if (p.n_src > 4) return launch_f16<6>(p, stream);
        return p.n_src <= 2 ? launch_f16<2>(p, stream) : launch_f16<4>(p, stream);
    }
    switch (mode) {
        case MODE_VARIANCE:
            if (p.n_src > 4) return launch_ch<MODE_VARIANCE, 6>(p, stream);  // BASELINE config 5: 7 views
            return p.n_src <= 2 ? launch_ch<MODE_VARIANCE, 2>(p, stream) : launch_ch<MODE_VARIANCE, 4>(p, stream);
        case MODE_WEIGHTED:
            if (p.n_src > 4) return launch_ch<MODE_WEIGHTED, 6>(p, stream);
            return p.n_src <= 2 ? launch_ch<MODE_WEIGHTED, 2>(p, stream) : launch_ch<MODE_WEIGHTED, 4>(p, stream);
        case MODE_PAIR: return launch_ch<MODE_PAIR, 1>(p, stream);
        case MODE_WARP: return launch_ch<MODE_WARP, 1>(p, stream);
    }
    set_error("internal: bad mode %d", mode);
    return D3D_ERR_INVALID_ARG;
#endif
}

}  // namespace d3d
