// Sandbox 2 (round 4): VERDICT r03 items 1a / 1b priced before any port -- GROUPS = 2: the geometry of a plane paid ONCE for both
// 16-channel groups (32 units, 128 taps, 32 stores per plane); ILP = 2: two (quad, view) units of different quads blended
// interleaved (disjoint accumulators).  Everything else as tools/sweep_sandbox.hip:
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
template <int PK, int GEO, int NT, int LDSR, int STORES, int OVH, int GROUPS = 1, int ILP = 1>
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
    constexpr int PD = 1, REFLDS = 0, TAPW = 16, NQ = 4 * GROUPS, NU = 4 * NQ;
    f4 r[NQ], tapv[4];
#pragma unroll
    for (int q = 0; q < NQ; ++q) r[q] = (f4){consts[52 + (q & 3)] + lane + q, consts[53 + (q & 3)], consts[54 + (q & 3)], consts[55 + (q & 3)]};
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        tapv[q] = (f4){consts[60 + q] * lane, consts[61 + q], consts[62 + q], consts[63 + q]};
    }
    const float invV = 0.2f;
    float acc = 0.0f;
    unsigned long long t0, t1, rt0, rt1;
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(rt0)::"memory");
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
        typedef __attribute__((address_space(3))) const f4* lptr;
        // taps of unit u = (quad q2, view i2): four ds_read_b128 at the ring addresses (the second group's ring 40 KB further on)
        auto request = [&](int u, f4 (&dst)[4]) {
            const int q2 = u / 4, i2 = u % 4;
            const unsigned a0 = ((unsigned)t[i2].a0 & 0xfff0u), a1 = ((unsigned)t[i2].a1 & 0xfff0u);
            const unsigned go = (q2 & 3) * 16 + (q2 >> 2) * 70 * 1024;
            dst[0] = *(lptr)(a0 + go); dst[1] = *(lptr)(a0 + go + 80);
            dst[2] = *(lptr)(a1 + go); dst[3] = *(lptr)(a1 + go + 80);
        };
        auto unit = [&](int q, int i, f4& s, f4& qq, const f4& c0, const f4& c1, const f4& c2, const f4& c3) {
            const f2 wnw = {t[i].nw, t[i].nw}, wne = {t[i].ne, t[i].ne}, wsw = {t[i].sw, t[i].sw}, wse = {t[i].se, t[i].se};
            f2 a = pk_fma(lo2(c3), wse, pk_fma(lo2(c2), wsw, pk_fma(lo2(c1), wne, lo2(c0) * wnw)));
            f2 b = pk_fma(hi2(c3), wse, pk_fma(hi2(c2), wsw, pk_fma(hi2(c1), wne, hi2(c0) * wnw)));
            s = cat2(lo2(s) + a, hi2(s) + b);
            qq = cat2(pk_fma(a, a, lo2(qq)), pk_fma(b, b, hi2(qq)));
        };
        auto finish = [&](const f4& s, const f4& qq, unsigned long long& obq) {
            const f2 iv = {invV, invV};
            const f2 ml = lo2(s) * iv, mh = hi2(s) * iv;
            const f4 o = cat2(pk_fma(lo2(qq), iv, -(ml * ml)), pk_fma(hi2(qq), iv, -(mh * mh)));
            if (STORES) {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    asm volatile("global_store_dword %0, %1, %2 nt" : : "v"(pixb), "v"(o[k]), "s"(obq));
                    obq += 384ull * 688 * 464 * 4;
                }
            } else {
                asm volatile("" : : "v"(o[0]), "v"(o[1]), "v"(o[2]), "v"(o[3]));
            }
        };
        if (ILP == 1) {
            f4 s, qq;
            f4 tp[2][4];
            if (LDSR) request(0, tp[0]);
#pragma unroll
            for (int u = 0; u < NU; ++u) {
                const int q = u / 4, i = u % 4;
                if (i == 0) { s = r[q]; qq = s * s; }
                f4 c0, c1, c2, c3;
                if (LDSR) {
                    if (u + 1 < NU) request(u + 1, tp[(u + 1) & 1]);
                    c0 = tp[u & 1][0]; c1 = tp[u & 1][1]; c2 = tp[u & 1][2]; c3 = tp[u & 1][3];
                } else {
                    c0 = tapv[(q + 0) & 3]; c1 = tapv[(q + 1) & 3]; c2 = tapv[(q + 2) & 3]; c3 = tapv[(q + 3) & 3];
                }
                asm volatile("" : "+v"(c3));
                unit(q, i, s, qq, c0, c1, c2, c3);
                if (i == 3) finish(s, qq, ob);
            }
        } else {
            // two quads at a time: units (q, i) and (q + 1, i) written tap by tap side by side, accumulators disjoint
            f4 tpa[2][4], tpb[2][4];
            if (LDSR) { request(0, tpa[0]); request(4, tpb[0]); }
#pragma unroll
            for (int qp = 0; qp < NQ; qp += 2) {
                f4 sa = r[qp], sb = r[qp + 1];
                f4 qa = sa * sa, qb = sb * sb;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int n = (qp / 2) * 4 + i;   // pair index
                    f4 a0, a1, a2, a3, b0, b1, b2, b3;
                    if (LDSR) {
                        const int un = (i + 1 < 4) ? qp * 4 + i + 1 : (qp + 2) * 4;   // next pair's first unit
                        if (i + 1 < 4 || qp + 2 < NQ) { request(un, tpa[(n + 1) & 1]); request(un + 4, tpb[(n + 1) & 1]); }
                        a0 = tpa[n & 1][0]; a1 = tpa[n & 1][1]; a2 = tpa[n & 1][2]; a3 = tpa[n & 1][3];
                        b0 = tpb[n & 1][0]; b1 = tpb[n & 1][1]; b2 = tpb[n & 1][2]; b3 = tpb[n & 1][3];
                    } else {
                        a0 = tapv[0]; a1 = tapv[1]; a2 = tapv[2]; a3 = tapv[3]; b0 = tapv[1]; b1 = tapv[2]; b2 = tapv[3]; b3 = tapv[0];
                    }
                    asm volatile("" : "+v"(b3));
                    const f2 wnw = {t[i].nw, t[i].nw}, wne = {t[i].ne, t[i].ne}, wsw = {t[i].sw, t[i].sw}, wse = {t[i].se, t[i].se};
                    f2 xa = lo2(a0) * wnw, xb = lo2(b0) * wnw, ya = hi2(a0) * wnw, yb = hi2(b0) * wnw;
                    xa = pk_fma(lo2(a1), wne, xa); xb = pk_fma(lo2(b1), wne, xb); ya = pk_fma(hi2(a1), wne, ya); yb = pk_fma(hi2(b1), wne, yb);
                    xa = pk_fma(lo2(a2), wsw, xa); xb = pk_fma(lo2(b2), wsw, xb); ya = pk_fma(hi2(a2), wsw, ya); yb = pk_fma(hi2(b2), wsw, yb);
                    xa = pk_fma(lo2(a3), wse, xa); xb = pk_fma(lo2(b3), wse, xb); ya = pk_fma(hi2(a3), wse, ya); yb = pk_fma(hi2(b3), wse, yb);
                    sa = cat2(lo2(sa) + xa, hi2(sa) + ya); sb = cat2(lo2(sb) + xb, hi2(sb) + yb);
                    qa = cat2(pk_fma(xa, xa, lo2(qa)), pk_fma(ya, ya, hi2(qa))); qb = cat2(pk_fma(xb, xb, lo2(qb)), pk_fma(yb, yb, hi2(qb)));
                    __builtin_amdgcn_sched_barrier(0);
                }
                finish(sa, qa, ob);
                finish(sb, qb, ob);
            }
        }
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(rt1)::"memory");
    if (lane == 0) { stamps[(blockIdx.x * NT + threadIdx.x) / 64] = t1 - t0; stamps[256 * 16 + (blockIdx.x * NT + threadIdx.x) / 64] = rt1 - rt0; }
    if (acc == 123.0f) sink[0] = acc;
}

template <int NT, int LDSR, int STORES, int GROUPS, int ILP>
static void run_nt(const char* name, unsigned long long* d_st, float* d_sink, float* d_c, float* vol) {
    // ONE workgroup of NT threads per CU (the whole LDS is requested, so no second workgroup fits): NT / 256 waves per SIMD
    const int planes = 400 / GROUPS, blocks = 256, wps = NT / 256;
    std::vector<unsigned long long> st(blocks * (NT / 64));
    auto kern = plane_loop<1, 1, NT, LDSR, STORES, 1, GROUPS, ILP>;
    printf("  %-66s waves/SIMD %d : ", name, wps); fflush(stdout);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float ms = 0.0f;
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(e0, 0);
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(NT), 160 * 1024, 0, d_st, d_sink, d_c, planes, vol);
        (void)hipEventRecord(e1, 0);
        (void)hipDeviceSynchronize();
        (void)hipEventElapsedTime(&ms, e0, e1);
    }
    std::vector<unsigned long long> rt(st.size());
    (void)hipMemcpy(st.data(), d_st, st.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    (void)hipMemcpy(rt.data(), d_st + 256 * 16, rt.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    std::sort(st.begin(), st.end());
    std::sort(rt.begin(), rt.end());
    const double c = (double)st[st.size() / 2] / planes;   // cycles per plane (GROUPS 16-channel groups) per wave
    // config 2: 384 planes x 688 x 464 pixels / 64 = 1 915 392 wave-planes of 32 channels over 1024 SIMDs; wall time of the
    // replay itself gives the clock the chip held (cycles / time)
    const double per_simd_32ch = c / wps * (2.0 / GROUPS);
    const double ghz = (double)st[st.size() / 2] / ((double)rt[rt.size() / 2] * 10.0);   // s_memrealtime ticks at 100 MHz
    (void)ms;
    printf("%7.0f cyc per plane-pass per wave | %6.0f per SIMD and 32-channel plane | clock %.2f GHz | config 2 at that clock: %.2f ms\n",
           c, per_simd_32ch, ghz, per_simd_32ch * 1915392.0 / 1024.0 / (ghz * 1e6));
    fflush(stdout);
}

int main(int argc, char** argv) {
    const int only = argc > 1 ? atoi(argv[1]) : 0;   // 0: 512 and 768 threads; else 256 / 512 / 768 / 1024
    unsigned long long* d_st;
    float *d_sink, *d_c, *vol;
    (void)hipMalloc(&d_st, 2 * 256 * 16 * sizeof(unsigned long long));
    (void)hipMalloc(&d_sink, 16);
    (void)hipMalloc(&d_c, 128 * sizeof(float));
    if (hipMalloc(&vol, (size_t)33 * 384 * 688 * 464 * 4) != hipSuccess) { printf("alloc failed\n"); return 1; }
    float hc[128];
    for (int i = 0; i < 128; ++i) hc[i] = 0.37f + 0.01f * i;
    for (int i = 0; i < 4; ++i) { hc[i * 3 + 2] = 1e-4f; hc[24 + i] = 1.0f; hc[28 + i] = 3; hc[32 + i] = 2; hc[36 + i] = 48; hc[40 + i] = 8; hc[44 + i] = 3920; hc[48 + i] = 4096 * i; }
    hc[70] = 400.0f;
    (void)hipMemcpy(d_c, hc, sizeof(hc), hipMemcpyHostToDevice);
    printf("plane loop of the sweep kernel in a sandbox (geometry + packed blend + LDS taps + nt stores; no planner, staging, barriers)\n");
#define ALL(NT)                                                                                                       \
    run_nt<NT, 1, 1, 1, 1>("kernel's form: geometry per 16-channel group", d_st, d_sink, d_c, vol);                      \
    run_nt<NT, 1, 1, 2, 1>("(a) geometry once per 32 channels (two groups per plane)", d_st, d_sink, d_c, vol);          \
    run_nt<NT, 1, 1, 1, 2>("(b) two units of different quads interleaved", d_st, d_sink, d_c, vol);                      \
    run_nt<NT, 1, 1, 2, 2>("(a) + (b)", d_st, d_sink, d_c, vol);                                                         \
    run_nt<NT, 1, 0, 1, 1>("taps, no stores, kernel's form", d_st, d_sink, d_c, vol);                                    \
    run_nt<NT, 1, 0, 2, 1>("taps, no stores, (a)", d_st, d_sink, d_c, vol);                                              \
    run_nt<NT, 0, 1, 1, 1>("stores, no taps, kernel's form", d_st, d_sink, d_c, vol);                                    \
    run_nt<NT, 0, 1, 2, 1>("stores, no taps, (a)", d_st, d_sink, d_c, vol);                                              \
    run_nt<NT, 0, 0, 1, 1>("arithmetic only, kernel's form", d_st, d_sink, d_c, vol);                                    \
    run_nt<NT, 0, 0, 2, 1>("arithmetic only, (a)", d_st, d_sink, d_c, vol);                                              \
    run_nt<NT, 0, 0, 2, 2>("arithmetic only, (a) + (b)", d_st, d_sink, d_c, vol);
    if (only == 256) { ALL(256) }
    if (!only || only == 512) { ALL(512) }
    if (!only || only == 768) { ALL(768) }
    if (only == 1024) { ALL(1024) }
    return 0;
}
