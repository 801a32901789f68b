// Sandbox (round 3): the plane loop of the sweep kernel rebuilt piece by piece -- per plane and 16-channel group the
// geometry of four views (the instruction sequence of geo_ring in csrc/planesweep_tiled.hip) and sixteen (quad, view) units
// of blend + sum / sum-of-squares accumulation, taps taken from registers -- with no LDS, no stores, no barriers, so
// that what the vector units sustain on exactly this instruction mix can be read off at 1, 2, 3 and 4 waves per SIMD, with
// packed (v_pk_fma_f32) or plain (v_fma_f32) arithmetic.  Shader cycles from s_memtime (median over the waves).
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/plane_loop_rate.hip -o tools/plane_loop_rate
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
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

// LDSR: the taps are read from LDS (ds_read_b128 at the computed ring addresses, one unit ahead) instead of standing in registers
// STORES: the sixteen results of a plane-group leave as global_store_dword ... nt (scalar base + lane offset), 490 MB apart
// OVH: per plane one LDS atomic (plane hand-out) and one LDS read (the plane's depth), as the kernel has them
template <int PK, int GEO, int NT, int LDSR, int STORES, int OVH, int PD = 1, int REFLDS = 0, int TAPW = 16>
__global__ __launch_bounds__(NT, 1) void plane_loop(unsigned long long* stamps, float* sink, const float* consts, int planes, float* vol) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    for (int i = threadIdx.x; i < 40 * 1024 - 16; i += NT) lds[i] = (float)(i & 1023) * 1e-3f;
    __syncthreads();
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
    int* ldsi = reinterpret_cast<int*>(lds);
    unsigned pixb = (unsigned)((blockIdx.x * NT + threadIdx.x) * 4) % (688u * 464u * 4u);
    if (STORES == 2) {   // two 128-byte row segments per instruction, rows 1856 bytes apart (w = 464: every other row starts mid-line)
        const unsigned wv = (blockIdx.x * (NT / 64) + (threadIdx.x >> 6));
        const unsigned tx = wv % 14u, ty = (wv / 14u) % 343u;
        pixb = ((2u * ty + ((threadIdx.x >> 5) & 1u)) * 464u + 32u * tx + (threadIdx.x & 31u)) * 4u;
    }
    for (int pl = 0; pl < planes; ++pl) {
        dv += 1.04f;
        if (OVH) {
            int jg = 0;
            if (lane == 0) jg = __hip_atomic_fetch_add(ldsi + 40 * 1024 - 8 + (threadIdx.x >> 8), 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            jg = __builtin_amdgcn_readfirstlane(jg);
            dv += lds[jg & 255] * 1e-6f;
        }
        unsigned long long ob = 0;
        if (STORES) {
            const unsigned long long b = reinterpret_cast<unsigned long long>(vol + (size_t)((blockIdx.x * 131 + pl) % 384) * (688 * 464));
            const unsigned blo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)b), bhi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(b >> 32));
            ob = ((unsigned long long)bhi << 32) | (unsigned long long)blo;
        }
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
        typedef __attribute__((address_space(3))) const f4* lptr;
        f4 tp[2][4];
        auto request = [&](int u, f4 (&dst)[4]) {
            const int q2 = u / 4, i2 = u % 4;
            const unsigned a0 = ((unsigned)t[i2].a0 & 0x1fff0u) % (150u * 1024u), a1 = ((unsigned)t[i2].a1 & 0x1fff0u) % (150u * 1024u);
            if (TAPW == 16) {
                dst[0] = *(lptr)(a0 + q2 * 16); dst[1] = *(lptr)(a0 + q2 * 16 + 80);
                dst[2] = *(lptr)(a1 + q2 * 16); dst[3] = *(lptr)(a1 + q2 * 16 + 80);
            } else {
                typedef __attribute__((address_space(3))) const f2* lptr2;
#pragma unroll
                for (int tt = 0; tt < 4; ++tt) {
                    const unsigned ad = ((tt & 2) ? a1 : a0) + q2 * 16 + ((tt & 1) ? 80 : 0);
                    const f2 lo = *(lptr2)(ad), hi = *(lptr2)(ad + 8);
                    dst[tt] = (f4){lo[0], lo[1], hi[0], hi[1]};
                }
            }
        };
        if (LDSR && PD) request(0, tp[0]);
        f4 rl[4];
        if (REFLDS) {
#pragma unroll
            for (int q = 0; q < 4; ++q) rl[q] = *(lptr)(((unsigned)lane * 80u + q * 16u + 120u * 1024u));
        }
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int q = u / 4, i = u % 4;
            if (i == 0) { s = REFLDS ? rl[q] : r[q]; qq = s * s; }
            f4 c0, c1, c2, c3;
            if (LDSR && PD) {
                if (u + 1 < 16) request(u + 1, tp[(u + 1) & 1]);
                c0 = tp[u & 1][0]; c1 = tp[u & 1][1]; c2 = tp[u & 1][2]; c3 = tp[u & 1][3];
            } else if (LDSR) {
                request(u, tp[0]);
                c0 = tp[0][0]; c1 = tp[0][1]; c2 = tp[0][2]; c3 = tp[0][3];
            } else {   // taps: registers (they stand for the four ds_read_b128 results)
                c0 = tapv[(q + 0) & 3]; c1 = tapv[(q + 1) & 3]; c2 = tapv[(q + 2) & 3]; c3 = tapv[(q + 3) & 3];
            }
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
                if (STORES) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        asm volatile("global_store_dword %0, %1, %2 nt" : : "v"(pixb), "v"(o[k]), "s"(ob));
                        ob += 384ull * 688 * 464 * 4;
                    }
                } else {
                    asm volatile("" : : "v"(o[0]), "v"(o[1]), "v"(o[2]), "v"(o[3]));   // (the four stores)
                }
            }
        }
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    if (lane == 0) stamps[(blockIdx.x * NT + threadIdx.x) / 64] = t1 - t0;
    if (acc == 123.0f) sink[0] = acc;
}

template <int PK, int GEO, int NT, int LDSR, int STORES, int OVH, int PD = 1, int REFLDS = 0, int TAPW = 16>
static void run_nt(const char* name, unsigned long long* d_st, float* d_sink, float* d_c, float* vol) {
    // ONE workgroup of NT threads per CU (the whole LDS is requested, so no second workgroup fits): NT / 256 waves per SIMD
    const int planes = 400, blocks = 256, wps = NT / 256;
    std::vector<unsigned long long> st(blocks * (NT / 64));
    auto kern = plane_loop<PK, GEO, NT, LDSR, STORES, OVH, PD, REFLDS, TAPW>;
    printf("  %-58s waves/SIMD %d : ", name, wps); fflush(stdout);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(NT), 160 * 1024, 0, d_st, d_sink, d_c, planes, vol);
        (void)hipDeviceSynchronize();
    }
    (void)hipMemcpy(st.data(), d_st, st.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    std::sort(st.begin(), st.end());
    const double c = (double)st[st.size() / 2] / planes;
    printf("%7.0f cycles per plane-group per wave, %7.0f per SIMD\n", c, c / wps); fflush(stdout);
}

int main(int argc, char** argv) {
    const int only = argc > 1 ? atoi(argv[1]) : 0;   // 0: all block sizes; else 256 / 512 / 768 / 1024
    unsigned long long* d_st;
    float *d_sink, *d_c, *vol;
    (void)hipMalloc(&d_st, 256 * 16 * sizeof(unsigned long long));
    (void)hipMalloc(&d_sink, 16);
    (void)hipMalloc(&d_c, 128 * sizeof(float));
    if (hipMalloc(&vol, (size_t)17 * 384 * 688 * 464 * 4) != hipSuccess) { printf("alloc failed\n"); return 1; }
    float hc[128];
    for (int i = 0; i < 128; ++i) hc[i] = 0.37f + 0.01f * i;
    for (int i = 0; i < 4; ++i) { hc[i * 3 + 2] = 1e-4f; hc[24 + i] = 1.0f; hc[28 + i] = 3; hc[32 + i] = 2; hc[36 + i] = 48; hc[40 + i] = 8; hc[44 + i] = 3920; hc[48 + i] = 4096 * i; }
    hc[70] = 400.0f;
    (void)hipMemcpy(d_c, hc, sizeof(hc), hipMemcpyHostToDevice);
    printf("plane loop of the sweep kernel in a sandbox: packed arithmetic + geometry, then + per-plane overhead, + LDS tap reads, + stores\n");
#define ALL(NT)                                                                                        \
    run_nt<1, 1, NT, 0, 0, 0>("arithmetic only", d_st, d_sink, d_c, vol);                               \
    run_nt<1, 1, NT, 0, 0, 1>("+ plane hand-out atomic and depth read", d_st, d_sink, d_c, vol);        \
    run_nt<1, 1, NT, 1, 0, 1>("+ 64 ds_read_b128 taps (one unit ahead)", d_st, d_sink, d_c, vol);       \
    run_nt<1, 1, NT, 0, 1, 1>("+ 16 global_store_dword nt (no LDS taps)", d_st, d_sink, d_c, vol);      \
    run_nt<1, 1, NT, 1, 1, 1>("+ taps + stores", d_st, d_sink, d_c, vol);                                \
    run_nt<1, 1, NT, 1, 1, 1, 0, 0>("+ taps + stores, taps NOT requested ahead", d_st, d_sink, d_c, vol); \
    run_nt<1, 1, NT, 1, 1, 1, 1, 1>("+ taps + stores, reference features from LDS", d_st, d_sink, d_c, vol); \
    run_nt<1, 1, NT, 1, 1, 1, 0, 1>("+ taps + stores, no request ahead, reference from LDS", d_st, d_sink, d_c, vol); \
    run_nt<1, 1, NT, 0, 2, 1>("+ stores in the kernel's shape (2 x 128 B rows, pitch 1856 B), no taps", d_st, d_sink, d_c, vol); \
    run_nt<1, 1, NT, 1, 2, 1>("+ taps + stores in the kernel's shape", d_st, d_sink, d_c, vol);
    if (!only || only == 256) { ALL(256) }
    if (!only || only == 512) { ALL(512) }
    if (!only || only == 768) { ALL(768) }
    if (!only || only == 1024) { ALL(1024) }
    return 0;
}
