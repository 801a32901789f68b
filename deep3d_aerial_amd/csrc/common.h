// Shared host/device helpers for the gfx950 plane-sweep engine (not part of the public ABI).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>

#include "../../include/deep3d_planesweep.h"

namespace d3d {

void set_error(const char* fmt, ...);
int hip_status(hipError_t e, const char* what);
// hipFuncSetAttribute(MaxDynamicSharedMemorySize) once per (kernel, device, size reached): a launch path calls this before
// every launch, the runtime is asked only when the kernel has not been given that much on the current device yet
// (a CasMVSNet view made ~800 of these calls, ~1 us of host time each).
int ensure_dynamic_lds(const void* kernel, int bytes);

#define D3D_REQUIRE(cond, ...)                \
    do {                                      \
        if (!(cond)) {                        \
            ::d3d::set_error(__VA_ARGS__);    \
            return D3D_ERR_INVALID_ARG;       \
        }                                     \
    } while (0)

#define D3D_LAUNCH_CHECK(what)                                     \
    do {                                                           \
        int _rc = ::d3d::hip_status(hipGetLastError(), what);      \
        if (_rc != D3D_OK) return _rc;                             \
    } while (0)

static inline int ceil_div(long a, long b) { return (int)((a + b - 1) / b); }

// Two floats -> one register of two bf16 (RNE, NaN stays NaN) in ONE v_cvt_pk_bf16_f32.  Written as two scalar casts the
// compiler emits a conversion per value plus a shift and an or: four instructions -- on kernels bound by instruction issue
// (every epilogue of the channel-last layers) that is a tenth of the work.  Same values either way.
__device__ __forceinline__ unsigned pack_bf16x2(float a, float b) {
    typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
    typedef float f32x2_t __attribute__((ext_vector_type(2)));
    return __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2_t){a, b}, bf16x2_t));
}

// ---------------------------------------------------------------------------------------
// "h16": the 16-bit operand format of the fast mode (BASELINE config 3) -- one format for the whole library, fixed at build
// time and reported by d3d_h16_format().  IEEE half ("f16", 11 significand bits) by default; -DD3D_H16_BF16 builds the
// bfloat16 form (8 bits) rounds 2-4 shipped.  Why: on the arg-max-sensitive model fixtures EVERY rounding site of a regulariser
// (input volume, each layer's weights, each stored activation: 23 sites in a CostRegNet) carries 0.1 - 0.7 stage-3 depth
// intervals of error on its own in bf16 and they add in quadrature to 0.9 - 1.4 -- against the 0.25 the parity bar allows; no
// subset of sites kept in fp32 helps (profiles/r05_bf16_ablation.txt).  Half has the same bytes, the same matrix-core rate
// (v_mfma_f32_16x16x32_f16), single-instruction conversions both ways (v_cvt_pk_f16_f32, v_cvt_f32_f16) and an eighth of the
// rounding error.  Its range (6e-8 .. 65504) covers BN-normalised activations and weights; the one operand whose magnitude the
// data decides, the variance volume, saturates instead of overflowing (sweep_device.h).  The split-operand kernels (fp32 mode,
// "bf16x3") keep their three bf16 pieces: 24 significand bits need bf16's exponent range in the low pieces.
// ---------------------------------------------------------------------------------------
typedef float f32x4_t __attribute__((ext_vector_type(4)));
#ifdef D3D_H16_BF16
#define D3D_H16_FORMAT "bf16"
typedef __bf16 h16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ unsigned pack_h16x2(float a, float b) { return pack_bf16x2(a, b); }
__device__ __forceinline__ float h16_lo(unsigned u) { return __builtin_bit_cast(float, u << 16); }
__device__ __forceinline__ float h16_hi(unsigned u) { return __builtin_bit_cast(float, u & 0xffff0000u); }
__device__ __forceinline__ f32x4_t mfma_h16(h16x8 a, h16x8 b, f32x4_t c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
typedef __bf16 h16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x4_t mfma_h16_k16(h16x4 a, h16x4 b, f32x4_t c) { return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, b, c, 0, 0, 0); }
#else
#define D3D_H16_FORMAT "f16"
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ unsigned pack_h16x2(float a, float b) {   // one v_cvt_pk_f16_f32 (RNE)
    typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
    typedef float f32x2_t __attribute__((ext_vector_type(2)));
    return __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2_t){a, b}, f16x2_t));
}
__device__ __forceinline__ float h16_lo(unsigned u) {   // v_cvt_f32_f16
    typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
    return (float)__builtin_bit_cast(f16x2_t, u)[0];
}
__device__ __forceinline__ float h16_hi(unsigned u) {   // v_cvt_f32_f16 ... src0_sel:WORD_1
    typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
    return (float)__builtin_bit_cast(f16x2_t, u)[1];
}
__device__ __forceinline__ f32x4_t mfma_h16(h16x8 a, h16x8 b, f32x4_t c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }
typedef _Float16 h16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x4_t mfma_h16_k16(h16x4 a, h16x4 b, f32x4_t c) { return __builtin_amdgcn_mfma_f32_16x16x16f16(a, b, c, 0, 0, 0); }
#endif
__device__ __forceinline__ h16x4 cvt_h16x4(float a, float b, float c, float d) {   // two packed conversions
    typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
    return __builtin_bit_cast(h16x4, (u32x2_t){pack_h16x2(a, b), pack_h16x2(c, d)});
}
// one value rounded to the format and back (what a kernel that keeps fp32 values "as the matrix cores see them" applies)
__device__ __forceinline__ float round_h16(float v) { return h16_lo(pack_h16x2(v, 0.0f)); }

// sigmoid and tanh of the conv-GRU epilogues (adamvs.py:60-72 / module.py ConvGRUCell: torch.sigmoid, torch.tanh): one v_exp_f32
// and one v_rcp_f32 each, ~2e-7 of the exact value (relative for the sigmoid, absolute for the tanh) -- far inside what the
// state's own fp32 rounding moves.  The IEEE division and tanhf they replace expand to ~12 and ~35 instructions per value, a
// quarter of the fused cell's instruction stream.  Every kernel family uses these two, so fused and unfused cells agree bit for bit.
__device__ __forceinline__ float gru_sigmoid(float y) { return __builtin_amdgcn_rcpf(1.0f + __expf(-y)); }
__device__ __forceinline__ float gru_tanh(float y) { return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __expf(2.0f * y)); }
// The fast forms above belong to the instruction-bound h16 kernels (whose operands carry 2^-12 of rounding anyway).  The fp32-mode
// kernels -- exact fp32 / split-operand convolutions, the elementwise gate and update kernels -- take torch's own expressions
// (IEEE division, tanhf: no cancellation near 0), so fp32 mode reproduces torch.sigmoid / torch.tanh to their last bits and
// nothing compounds over the 48+ recurrent slices.
template <bool FAST>
__device__ __forceinline__ float gru_sigmoid_as(float y) { return FAST ? gru_sigmoid(y) : 1.0f / (1.0f + expf(-y)); }
template <bool FAST>
__device__ __forceinline__ float gru_tanh_as(float y) { return FAST ? gru_tanh(y) : tanhf(y); }

// Bytes of a channel-last bf16 cell of C channels in LDS, read as the A operand of v_mfma_f32_16x16x32_bf16 (one ds_read_b128 per
// lane: pixel lane & 15, K group lane >> 4).  gfx950 services that read in four NON-contiguous 16-lane groups -- {0-3, 12-15,
// 20-27}, {4-11, 16-19, 28-31}, + 32 -- so pixels {0-3, 12-15} of one K group meet pixels {4-11} of the NEXT K group on the
// banks.  With 16 or more channels two neighbouring K groups are the two 16-byte halves of one cell, and the 16 lanes of a group
// fall on 16 different 16-byte slots of the 256-byte bank row exactly when the cell pitch is 32 bytes mod 64: 32, 96, 160.  The
// "odd number of 16-byte slots" of rounds 2-4 (48, 80, 144 bytes) assumed contiguous lane groups and made every one of these
// reads two-way conflicted: 8 LDS cycles instead of 4 (tools/conv_bank_sim.py; SQ_LDS_BANK_CONFLICT was 42 % of the LDS cycles
// of the fused conv-GRU cell).  8-channel cells are 16 bytes: K groups are then taps, and neighbouring taps share their cell.
__host__ __device__ constexpr int bf16_cell_bytes(int C) { return C <= 8 ? 16 : ((2 * C - 32 + 63) / 64) * 64 + 32; }

// Tiles per workgroup along y.  A workgroup pays its prologue once (the weights into LDS, the per-lane task state): more tiles per
// workgroup amortise it and let the next tile's patch fly under the sweep -- but the launch should still cover the chip, and a
// grid a little above a whole number of rounds of resident workgroups leaves most of the last round's slots idle (a 2-D tile kernel at 688 x 464, 32
// channels: 1290 one-tile workgroups on 512 slots = three rounds, the last half empty; 430 three-tile workgroups = one).  Round 5
// picks the count that minimises  rounds x (tiles + prologue)  with the prologue priced in tiles from the bytes it moves;
// -DD3D_Z2_TPER_MODEL=0 is the rule of rounds 2-4 (halve from 8 until there are 1024 workgroups).
#ifndef D3D_Z2_TPER_MODEL
#define D3D_Z2_TPER_MODEL 1
#endif
inline int pick_tper(int gx, int nty, int lds_bytes, int weight_bytes, int patch_bytes, int max_per_cu = 2) {
#if D3D_Z2_TPER_MODEL
    const int per_cu = lds_bytes > 0 ? (160 * 1024 / lds_bytes < max_per_cu ? (160 * 1024 / lds_bytes > 0 ? 160 * 1024 / lds_bytes : 1) : max_per_cu) : max_per_cu;
    const long slots = 256L * per_cu;
#ifndef D3D_TPER_P0
#define D3D_TPER_P0 0.25
#endif
#ifndef D3D_TPER_P1
#define D3D_TPER_P1 0.5
#endif
    const double prologue = D3D_TPER_P0 + D3D_TPER_P1 * (double)weight_bytes / (double)(patch_bytes > 0 ? patch_bytes : 1);
    int best = 1;
    double best_cost = 1e30;
    for (int t = 1; t <= 8; ++t) {
        const long wgs = (long)gx * ceil_div(nty, t);
        const long rounds = (wgs + slots - 1) / slots;
        const double cost = (double)rounds * ((double)t + prologue);
        if (cost < best_cost * (1.0 - 1e-9)) { best_cost = cost; best = t; }
    }
    return best;
#else
    (void)lds_bytes; (void)weight_bytes; (void)patch_bytes; (void)max_per_cu;
    int tper = 8;   // every tile re-reads two halo rows of its neighbour; enough workgroups for 256 CUs come first
    while (tper > 1 && (long)gx * ceil_div(nty, tper) < 1024) tper >>= 1;
    return tper;
#endif
}

// The same question for the z-streaming 3-D kernels: into how many depth segments to cut a volume whose (x, y) tiles alone do not
// cover the chip.  A segment pays `halo` planes it shares with its neighbour plus a prologue; the count that minimises
// rounds x (planes per segment + halo + prologue) wins (rounds 2-4: double until there are 1024 workgroups).  Returns planes per segment.
inline int pick_zper(long tiles_xy, int D, int min_planes, int halo, int lds_bytes, int max_per_cu = 2, long old_target = 1024) {
#if D3D_Z2_TPER_MODEL
    const int fit = lds_bytes > 0 ? 160 * 1024 / lds_bytes : max_per_cu;
    const long slots = 256L * (fit < 1 ? 1 : (fit < max_per_cu ? fit : max_per_cu));
    int best = D;
    double best_cost = 1e30;
    for (int nz = 1; nz <= D; ++nz) {
        const int zper = ceil_div(D, nz);
        if (zper < min_planes && nz > 1) break;
        if (nz > 1 && zper == ceil_div(D, nz - 1)) continue;   // (the same segments as the previous count)
        const long wgs = tiles_xy * ceil_div(D, zper);
        const long rounds = (wgs + slots - 1) / slots;
        const double cost = (double)rounds * ((double)zper + (double)halo + 1.0);
        if (cost < best_cost * (1.0 - 1e-9)) { best_cost = cost; best = zper; }
    }
    return best;
#else
    (void)halo; (void)lds_bytes; (void)max_per_cu;
    int nz = 1;
    while (tiles_xy * nz < old_target && D / (nz * 2) >= min_planes) nz *= 2;
    return ceil_div(D, nz);
#endif
}

// gfx950: the data registers of a 12- / 16-byte buffer store are still being read when the NEXT vector instructions issue.  LLVM's
// hazard recognizer inserts the wait states only when the store's soffset is an immediate (GCNHazardRecognizer::createsVALUHazard:
// "this hazard only exists if the instruction is not using a register in the soffset field"); with an SGPR soffset nothing is
// inserted, and on this chip a store followed at once by an instruction that overwrites its data registers puts the NEW values of
// lanes 12-15 of each row of 16 (second dword) into memory -- found in round 5 as the "nondeterministic" stride-2 fused conv-GRU
// cell with 8-row tiles (tools/gru2_debug.py; the stride-1 cells had the same sequence, on lanes whose stores are dropped).  The
// empty asm keeps the stored registers alive behind two wait states; tools/store_hazard_scan.py (a CPU test runs it on the built
// library) looks for the pattern in every kernel.
typedef unsigned d3d_u4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void buffer_store_b128_guarded(d3d_u4 data, __amdgpu_buffer_rsrc_t rsrc, unsigned voffset, unsigned soffset) {
    __builtin_amdgcn_raw_buffer_store_b128(data, rsrc, voffset, soffset, 0);
    asm volatile("s_nop 1" : : "v"(data) : "memory");
}

// Workgroup barrier for data handed over through LDS only.  __syncthreads() is a workgroup-scope fence + s_barrier, and the
// fence waits for EVERY memory operation the wave has in flight (s_waitcnt vmcnt(0)): the global stores of the plane just
// finished and the global loads issued ahead for a later plane -- a z-streaming kernel then pays a store's round trip per
// plane and cannot prefetch across the barrier.  This one waits for the wave's LDS (and scalar) operations only; the
// "memory" clobber keeps the compiler from moving memory accesses across it.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// ---------------------------------------------------------------------------------------
// Geometry shared by every warp kernel.  module.py:532-546: p = (rot@[x,y,1])*d + trans,
// (u,v) = p.xy/p.z, sampled bilinearly with align_corners=True so (u,v) are pixel
// coordinates directly.  The multiply and add are kept as two roundings like the
// reference's `rot_depth_xyz + trans`; the divide is v_rcp_f32 plus one Newton step.
// ---------------------------------------------------------------------------------------
struct Ray {  // rot @ [x,y,1] for one (pixel, source view)
    float rx, ry, rz;
};

__device__ __forceinline__ Ray make_ray(const float* __restrict__ P, float x, float y) {
    Ray r;
    r.rx = fmaf(P[0], x, fmaf(P[1], y, P[2]));
    r.ry = fmaf(P[4], x, fmaf(P[5], y, P[6]));
    r.rz = fmaf(P[8], x, fmaf(P[9], y, P[10]));
    return r;
}

// Projects to source pixel coordinates.  Non-finite or far-outside results are moved to
// (-2,-2), where all four taps are outside the image (zero padding) -- SURVEY.md a1.
__device__ __forceinline__ void project(const Ray& r, float tx, float ty, float tz, float d, int h, int w,
                                        float& u, float& v) {
    float px = __fadd_rn(__fmul_rn(r.rx, d), tx);
    float py = __fadd_rn(__fmul_rn(r.ry, d), ty);
    float pz = __fadd_rn(__fmul_rn(r.rz, d), tz);
    // quotient = v_rcp_f32 estimate + one Newton correction: correctly rounded in practice, so
    // integer-aligned homographies (identity, whole-pixel shifts) reproduce exact copies as the
    // reference's IEEE divide does.
    float iz = __builtin_amdgcn_rcpf(pz);
    float u0 = px * iz, v0 = py * iz;
    u = fmaf(fmaf(-u0, pz, px), iz, u0);
    v = fmaf(fmaf(-v0, pz, py), iz, v0);
    bool ok = (u > -2.0f) && (u < (float)w + 1.0f) && (v > -2.0f) && (v < (float)h + 1.0f);
    u = ok ? u : -2.0f;
    v = ok ? v : -2.0f;
}

}  // namespace d3d
