// Device helpers shared by the plane-sweep kernels that stage source windows in LDS (planesweep_tiled.hip: torus rings over deep
// sweeps; planesweep_window.hip: one window per patch over the cascades' shallow sweeps): wave reductions, bilinear tap
// descriptors, packed two-channel arithmetic, scalar-base streaming stores.  Not part of the public ABI.
#pragma once
#include <hip/hip_runtime.h>

#include "common.h"

namespace d3d {
namespace {

__device__ __forceinline__ float wave_min(float v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v = fminf(v, __shfl_xor(v, m));
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v = fmaxf(v, __shfl_xor(v, m));
    return v;
}
__device__ __forceinline__ int wave_maxi(int v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v = max(v, __shfl_xor(v, m));
    return v;
}
__device__ __forceinline__ int rfl(int v) { return __builtin_amdgcn_readfirstlane(v); }

// Per-view sample geometry against the ring.
struct TapL {
    int a0, a1;            // byte addresses of the north / south tap rows (west tap; east = +STRIDE*4)
    float nw, ne, sw, se;  // bilinear weights (taps outside the image are zeros in LDS)
};
// Global-memory variant (fallback workgroups).
struct TapG {
    int off, dx, dyw;
    float nw, ne, sw, se;
};

__device__ __forceinline__ TapG make_tap_glb(float u, float v, int h, int w) {
    TapG t;
    float fu = floorf(u), fv = floorf(v);
    int x0 = (int)fu, y0 = (int)fv;
    float ax = u - fu, ay = v - fv;
    float bx = (fu + 1.0f) - u, by = (fv + 1.0f) - v;
    bool vx0 = (x0 >= 0) && (x0 < w), vx1 = (x0 >= -1) && (x0 < w - 1);
    bool vy0 = (y0 >= 0) && (y0 < h), vy1 = (y0 >= -1) && (y0 < h - 1);
    t.nw = (vx0 && vy0) ? bx * by : 0.0f;
    t.ne = (vx1 && vy0) ? ax * by : 0.0f;
    t.sw = (vx0 && vy1) ? bx * ay : 0.0f;
    t.se = (vx1 && vy1) ? ax * ay : 0.0f;
    int x0c = min(max(x0, 0), w - 1), x1c = min(max(x0 + 1, 0), w - 1);
    int y0c = min(max(y0, 0), h - 1), y1c = min(max(y0 + 1, 0), h - 1);
    t.off = y0c * w + x0c;
    t.dx = x1c - x0c;
    t.dyw = (y1c - y0c) * w;
    return t;
}

typedef float f4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f4 lds_read4(const float* lds, int byte_addr) {
    return *reinterpret_cast<const f4*>(reinterpret_cast<const char*>(lds) + byte_addr);
}
// ds_read_b128 at an ABSOLUTE LDS byte address (the dynamic-LDS base is folded into RingView::base once, so the plane
// loop has no "base + offset" addition per tap address)
typedef __attribute__((address_space(3))) const f4* lds_f4_ptr;
__device__ __forceinline__ f4 lds_read4_abs(int byte_addr) { return *(lds_f4_ptr)(unsigned)byte_addr; }
__device__ __forceinline__ int lds_base_bytes(float* lds) { return (int)(unsigned)(size_t)(__attribute__((address_space(3))) float*)lds; }

typedef float f2 __attribute__((ext_vector_type(2)));
// Two channels per VALU instruction (v_pk_mul_f32 / v_pk_fma_f32 / v_pk_add_f32): each half is the IEEE operation,
// so results are bit-identical to the one-channel forms.  Why: beside the ds_read_b128 tap stream a SIMD's time goes
// with the NUMBER of VALU instructions it issues, not with their FLOPs (tools/issue_model.hip: 24 plain vs 12 packed
// instructions per unit = 302 vs 235 cycles per unit at 8 waves per CU).
__device__ __forceinline__ f2 pk_fma(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ f2 lo2(const f4& v) { return (f2){v[0], v[1]}; }
__device__ __forceinline__ f2 hi2(const f4& v) { return (f2){v[2], v[3]}; }
__device__ __forceinline__ f4 cat2(f2 a, f2 b) { return (f4){a[0], a[1], b[0], b[1]}; }

// same summation order as grid_sample: nw, ne, sw, se
__device__ __forceinline__ f4 blend(f4 t00, f4 t01, f4 t10, f4 t11, float nw, float ne, float sw, float se) {
    const f2 wnw = {nw, nw}, wne = {ne, ne}, wsw = {sw, sw}, wse = {se, se};
    f2 a = pk_fma(lo2(t11), wse, pk_fma(lo2(t10), wsw, pk_fma(lo2(t01), wne, lo2(t00) * wnw)));
    f2 b = pk_fma(hi2(t11), wse, pk_fma(hi2(t10), wsw, pk_fma(hi2(t01), wne, hi2(t00) * wnw)));
    return cat2(a, b);
}

// Store with a wave-uniform 64-bit base (SGPR pair) and a 32-bit per-lane byte offset: no 64-bit
// VALU address arithmetic per store.  Stores are never waited on inside the kernel.
__device__ __forceinline__ unsigned long long uniform64(const void* ptr) {  // force an SGPR pair
    const unsigned long long b = reinterpret_cast<unsigned long long>(ptr);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)b);
    const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(b >> 32));
    return ((unsigned long long)hi << 32) | lo;
}
__device__ __forceinline__ void store_sbase(unsigned long long sb, unsigned byte_off, float v) {
    // no "memory" clobber: nothing in the kernel reads the output, and a clobber would stop the
    // scheduler from hoisting the next unit's LDS reads above these stores
    // "nt": the cost volume is write-once streaming data; keep it from evicting the source windows
    // (re-read by neighbouring workgroups) out of L2 / Infinity Cache
    asm volatile("global_store_dword %0, %1, %2 nt" : : "v"(byte_off), "v"(v), "s"(sb));
}

// channel-last 16-bit output (the library's h16 format, common.h): four consecutive channels of the lane's voxel, RNE, as two
// dwords ...  The variance volume is the one operand of the regularisers whose magnitude the data decides (feature maps leave
// their nets without a normalisation), and so is the weighted correlation of the slice models: in the half format they SATURATE
// at the largest finite magnitude instead of becoming inf (one v_med3_f32 per value).
__device__ __forceinline__ unsigned long long pack_h16x4(const float __attribute__((ext_vector_type(4)))& v) {
#ifdef D3D_H16_BF16
    const unsigned a = pack_h16x2(v[0], v[1]), b = pack_h16x2(v[2], v[3]);
#else
    auto sat = [](float x) { return __builtin_amdgcn_fmed3f(x, -65504.0f, 65504.0f); };   // (a correlation can be negative)
    const unsigned a = pack_h16x2(sat(v[0]), sat(v[1])), b = pack_h16x2(sat(v[2]), sat(v[3]));
#endif
    return (unsigned long long)a | ((unsigned long long)b << 32);
}
// ... and EIGHT of them (two finished quads) as one 16-byte store: a lane's stores are 64 cells apart from its
// neighbours' (cell = C * 2 bytes), so every store instruction is 64 separate memory transactions whatever its width --
// 8-byte stores made the 16-channel groups of stages 1 / 2 twice as slow as the planar fp32 kernel
// PARTIAL (experiment, not used): with C > 8 the 16 bytes are only part of the voxel's cell -- the other channel groups arrive
// in later passes of the same workgroup -- and WRITE_SIZE shows every 16-byte store as a 32-byte write (stage 2, C = 16:
// 2.66 GB for a 1.31 GB volume, the planar volume's bytes).  Dropping `nt` so that the halves could merge in L2 / the
// Infinity Cache left WRITE_SIZE unchanged and moved the time by -5 % (window kernel, stage 2) to +20 % (ring kernel, stage 1).
template <bool PARTIAL = false>
__device__ __forceinline__ void store_sbase_h16x8(unsigned long long sb, unsigned byte_off, unsigned long long lo, unsigned long long hi) {
    typedef unsigned u4v __attribute__((ext_vector_type(4)));
    const u4v bits = {(unsigned)lo, (unsigned)(lo >> 32), (unsigned)hi, (unsigned)(hi >> 32)};
    // (a store of more than 8 bytes reads its data registers late: the next VALU write of one of them needs wait states, and
    //  the compiler's hazard recognizer does not see inside the string -- without the s_nop some lanes stored garbage)
    if constexpr (PARTIAL) asm volatile("global_store_dwordx4 %0, %1, %2\n\ts_nop 1" : : "v"(byte_off), "v"(bits), "s"(sb));
    else asm volatile("global_store_dwordx4 %0, %1, %2 nt\n\ts_nop 1" : : "v"(byte_off), "v"(bits), "s"(sb));
}
// one finished quad of a CL8 cell on its own: 8 bytes at byte_off (+ 8 for the cell's second quad).  The weighted-correlation
// instance of the window kernel has no registers left for holding the first quad until the second is finished (148 spills with
// the 16-byte form); a wave's two 8-byte stores still cover one contiguous kilobyte between them.
__device__ __forceinline__ void store_sbase_h16x4(unsigned long long sb, unsigned byte_off, unsigned long long quad) {
    typedef unsigned u2v __attribute__((ext_vector_type(2)));
    const u2v bits = {(unsigned)quad, (unsigned)(quad >> 32)};
    asm volatile("global_store_dwordx2 %0, %1, %2 nt" : : "v"(byte_off), "v"(bits), "s"(sb));
}
}  // namespace
}  // namespace d3d
