// Microbenchmark (round 3): the ARITHMETIC of the sweep kernel's plane loop in isolation -- per plane and 16-channel group the
// geometry of four views (the instruction sequence of geo_ring in csrc/planesweep_tiled.hip) and sixteen (quad, view) units
// of blend + sum / sum-of-squares accumulation, taps taken from registers -- with no LDS, no stores, no barriers, so
// that what the vector units sustain on exactly this instruction mix can be read off at 1, 2, 3 and 4 waves per SIMD, with
// packed (v_pk_fma_f32) or plain (v_fma_f32) arithmetic.  Shader cycles from s_memtime (median over the waves).
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/plane_loop_rate.hip -o tools/plane_loop_rate
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>

typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f2 pk_fma(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ f2 lo2(const f4& v) { return (f2){v[0], v[1]}; }
__device__ __forceinline__ f2 hi2(const f4& v) { return (f2){v[2], v[3]}; }
__device__ __forceinline__ f4 cat2(f2 a, f2 b) { return (f4){a[0], a[1], b[0], b[1]}; }

struct Tap { int a0, a1; float nw, ne, sw, se; };
struct Ray { float rx, ry, rz; };

__device__ __forceinline__ Tap geo(const Ray& r, float tx, float ty, float tz, float d, float umax, float vmax, int kx, int ky, int RW, int RH,
                                   int rowb, int base) {
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
    Tap t;
    t.nw = bx * by; t.ne = ax * by; t.sw = bx * ay; t.se = ax * ay;
    unsigned c = (unsigned)((int)fu + kx), rr = (unsigned)((int)fv + ky);
    c = min(c, c - (unsigned)RW);
    rr = min(rr, rr - (unsigned)RH);
    t.a0 = base + (int)__umul24(rr, (unsigned)rowb) + (int)__umul24(c, 80u);
    t.a1 = t.a0 + rowb;
    return t;
}

template <int PK, int GEO, int NT>
__global__ __launch_bounds__(NT, 1) void plane_loop(unsigned long long* stamps, float* sink, const float* consts, int planes) {
    const int lane = threadIdx.x & 63;
    Ray ray[4];
    float T0[4], T1[4], T2[4];
    int kx[4], ky[4], RW[4], RH[4], rowb[4], base[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        ray[i].rx = consts[i * 3 + 0] + lane * 1e-3f; ray[i].ry = consts[i * 3 + 1] + lane * 2e-3f; ray[i].rz = consts[i * 3 + 2];
        T0[i] = consts[16 + i]; T1[i] = consts[20 + i]; T2[i] = consts[24 + i];
        kx[i] = __builtin_amdgcn_readfirstlane((int)consts[28 + i]); ky[i] = __builtin_amdgcn_readfirstlane((int)consts[32 + i]);
        RW[i] = __builtin_amdgcn_readfirstlane((int)consts[36 + i]); RH[i] = __builtin_amdgcn_readfirstlane((int)consts[40 + i]);
        rowb[i] = __builtin_amdgcn_readfirstlane((int)consts[44 + i]); base[i] = __builtin_amdgcn_readfirstlane((int)consts[48 + i]);
    }
    f4 r[4], tapv[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        r[q] = (f4){consts[52 + q] + lane, consts[53 + q], consts[54 + q], consts[55 + q]};
        tapv[q] = (f4){consts[60 + q] * lane, consts[61 + q], consts[62 + q], consts[63 + q]};
    }
    const float invV = 0.2f;
    float acc = 0.0f;
    unsigned long long t0, t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    float dv = consts[70];
    for (int pl = 0; pl < planes; ++pl) {
        dv += 1.04f;
        Tap t[4];
        if (GEO) {
#pragma unroll
            for (int i = 0; i < 4; ++i) t[i] = geo(ray[i], T0[i], T1[i], T2[i], dv, 464.0f, 688.0f, kx[i], ky[i], RW[i], RH[i], rowb[i], base[i]);
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) { t[i].nw = dv; t[i].ne = dv * 0.5f; t[i].sw = dv * 0.25f; t[i].se = 1.0f - dv; t[i].a0 = i; t[i].a1 = i; }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) asm volatile("" : : "v"(t[i].a0), "v"(t[i].a1));
        f4 s, qq;
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int q = u / 4, i = u % 4;
            if (i == 0) { s = r[q]; qq = s * s; }
            // taps: registers (they stand for the four ds_read_b128 results)
            f4 c0 = tapv[(q + 0) & 3], c1 = tapv[(q + 1) & 3], c2 = tapv[(q + 2) & 3], c3 = tapv[(q + 3) & 3];
            asm volatile("" : "+v"(c3));
            f4 val;
            if (PK) {
                const f2 wnw = {t[i].nw, t[i].nw}, wne = {t[i].ne, t[i].ne}, wsw = {t[i].sw, t[i].sw}, wse = {t[i].se, t[i].se};
                f2 a = pk_fma(lo2(c3), wse, pk_fma(lo2(c2), wsw, pk_fma(lo2(c1), wne, lo2(c0) * wnw)));
                f2 b = pk_fma(hi2(c3), wse, pk_fma(hi2(c2), wsw, pk_fma(hi2(c1), wne, hi2(c0) * wnw)));
                val = cat2(a, b);
                s = cat2(lo2(s) + a, hi2(s) + b);
                qq = cat2(pk_fma(a, a, lo2(qq)), pk_fma(b, b, hi2(qq)));
            } else {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    val[k] = fmaf(c3[k], t[i].se, fmaf(c2[k], t[i].sw, fmaf(c1[k], t[i].ne, c0[k] * t[i].nw)));
                    s[k] += val[k];
                    qq[k] = fmaf(val[k], val[k], qq[k]);
                }
            }
            if (i == 3) {
                f4 o;
                if (PK) {
                    const f2 iv = {invV, invV};
                    const f2 ml = lo2(s) * iv, mh = hi2(s) * iv;
                    o = cat2(pk_fma(lo2(qq), iv, -(ml * ml)), pk_fma(hi2(qq), iv, -(mh * mh)));
                } else {
#pragma unroll
                    for (int k = 0; k < 4; ++k) { const float m = s[k] * invV; o[k] = fmaf(qq[k], invV, -(m * m)); }
                }
                asm volatile("" : : "v"(o[0]), "v"(o[1]), "v"(o[2]), "v"(o[3]));   // (the four stores)
            }
        }
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    if (lane == 0) stamps[(blockIdx.x * NT + threadIdx.x) / 64] = t1 - t0;
    if (acc == 123.0f) sink[0] = acc;
}

template <int PK, int GEO, int NT>
static void run_nt(const char* name, unsigned long long* d_st, float* d_sink, float* d_c) {
    // ONE workgroup of NT threads per CU (the whole LDS is requested, so no second workgroup fits): NT / 256 waves per SIMD
    const int planes = 400, blocks = 256, wps = NT / 256;
    std::vector<unsigned long long> st(blocks * (NT / 64));
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(plane_loop<PK, GEO, NT>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL((plane_loop<PK, GEO, NT>), dim3(blocks), dim3(NT), 160 * 1024, 0, d_st, d_sink, d_c, planes);
        (void)hipDeviceSynchronize();
    }
    (void)hipMemcpy(st.data(), d_st, st.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    std::sort(st.begin(), st.end());
    const double c = (double)st[st.size() / 2] / planes;
    printf("  %-44s waves/SIMD %d : %7.0f cycles per plane-group per wave, %7.0f per SIMD\n", name, wps, c, c / wps);
}

int main() {
    unsigned long long* d_st;
    float *d_sink, *d_c;
    (void)hipMalloc(&d_st, 256 * 16 * sizeof(unsigned long long));
    (void)hipMalloc(&d_sink, 16);
    (void)hipMalloc(&d_c, 128 * sizeof(float));
    float hc[128];
    for (int i = 0; i < 128; ++i) hc[i] = 0.37f + 0.01f * i;
    for (int i = 0; i < 4; ++i) { hc[i * 3 + 2] = 1e-4f; hc[24 + i] = 1.0f; hc[28 + i] = 3; hc[32 + i] = 2; hc[36 + i] = 48; hc[40 + i] = 8; hc[44 + i] = 3920; hc[48 + i] = 4096 * i; }
    hc[70] = 400.0f;
    (void)hipMemcpy(d_c, hc, sizeof(hc), hipMemcpyHostToDevice);
    printf("plane loop arithmetic (4 views of geometry + 16 units of blend/accumulate + finalisation), no LDS, no stores, no barriers\n");
#define ALL(NT)                                                                   \
    run_nt<1, 1, NT>("packed arithmetic, with geometry", d_st, d_sink, d_c);      \
    run_nt<0, 1, NT>("plain arithmetic, with geometry", d_st, d_sink, d_c);       \
    run_nt<1, 0, NT>("packed arithmetic, no geometry", d_st, d_sink, d_c);        \
    run_nt<0, 0, NT>("plain arithmetic, no geometry", d_st, d_sink, d_c);
    ALL(256) ALL(512) ALL(768) ALL(1024)
    return 0;
}
