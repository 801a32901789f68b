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
