/*
 * planesweep_oracle.c -- CPU restatement of the Deep3D_Aerial plane-sweep hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  This file is the parity checker for the HIP kernels in
 * deep3d_aerial_amd/csrc/.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load it.  The product path never calls it and has no CPU
 * fallback.
 *
 * Parity pinning: the reference ships no tests or golden vectors for this path
 * (SURVEY.md F4), so the oracle is pinned against outputs of the reference itself,
 * run on CPU in the build container by tests/golden/make_golden.py and committed as
 * tests/golden/ (.npz files) (checked by tests/test_oracle_golden.py).
 *
 * Every function cites the reference lines (relative to the reference checkout,
 * mvs/mvs_cas/models/...) whose arithmetic it restates.  Plain scalar fp32 C; the
 * only parallelism is an OpenMP "parallel for" over independent output elements,
 * which does not change any result.
 *
 * Layout everywhere: contiguous row-major, NCHW / NCDHW with the batch dimension
 * handled by the caller (the reference runs inference at batch 1, predict.py:49).
 */
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

#define API __attribute__((visibility("default")))

API int d3d_oracle_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

API void d3d_oracle_set_num_threads(int n) {
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}

/* ------------------------------------------------------------------------------------
 * module.py:528-530  proj = matmul(src_proj, inverse(ref_proj)); rot = proj[:3,:3];
 * trans = proj[:3,3:4].  fp32 Gauss-Jordan with partial pivoting stands in for
 * torch.inverse (LAPACK getrf/getri); differences are at fp32 rounding level and the
 * golden test for this function carries the tolerance.
 * out34: row-major 3x4 = [rot | trans].
 * Returns 0, or -1 if ref_proj is singular.
 * ---------------------------------------------------------------------------------- */
API int d3d_oracle_compose_proj(const float* src44, const float* ref44, float* out34) {
    float a[4][8];
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) {
            a[i][j] = ref44[i * 4 + j];
            a[i][4 + j] = (i == j) ? 1.0f : 0.0f;
        }
    for (int col = 0; col < 4; ++col) {
        int piv = col;
        float best = fabsf(a[col][col]);
        for (int r = col + 1; r < 4; ++r)
            if (fabsf(a[r][col]) > best) { best = fabsf(a[r][col]); piv = r; }
        if (best == 0.0f) return -1;
        if (piv != col)
            for (int j = 0; j < 8; ++j) { float t = a[col][j]; a[col][j] = a[piv][j]; a[piv][j] = t; }
        float inv = 1.0f / a[col][col];
        for (int j = 0; j < 8; ++j) a[col][j] *= inv;
        for (int r = 0; r < 4; ++r) {
            if (r == col) continue;
            float f = a[r][col];
            if (f == 0.0f) continue;
            for (int j = 0; j < 8; ++j) a[r][j] -= f * a[col][j];
        }
    }
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 4; ++j) {
            float s = 0.0f;
            for (int k = 0; k < 4; ++k) s += src44[i * 4 + k] * a[k][4 + j];
            out34[i * 4 + j] = s;
        }
    return 0;
}

/* ------------------------------------------------------------------------------------
 * One bilinear sample with per-tap zero padding, align_corners=True.
 * module.py:532-553: pixel (x,y), depth d:
 *   rot_xyz = rot @ [x,y,1]; p = rot_xyz*d + trans; (u,v) = p.xy / p.z;
 *   grid = (u/((W-1)/2) - 1, v/((H-1)/2) - 1); F.grid_sample(bilinear, zeros,
 *   align_corners=True) un-normalises with ((g+1)/2)*(size-1) and sums the four taps
 *   with weights (x_se-ix)(y_se-iy) ... , dropping taps outside the image.
 * The normalise / un-normalise round trip is kept so the fp32 rounding matches.
 * ---------------------------------------------------------------------------------- */
typedef struct {
    int x0, y0;          /* north-west tap */
    float nw, ne, sw, se; /* weights, already zeroed for out-of-image taps */
    int ok_nw, ok_ne, ok_sw, ok_se;
} tap_t;

static inline tap_t make_tap(const float* P, float x, float y, float d, int h, int w) {
    tap_t t;
    float rx = P[0] * x + P[1] * y + P[2];
    float ry = P[4] * x + P[5] * y + P[6];
    float rz = P[8] * x + P[9] * y + P[10];
    float px = rx * d + P[3];
    float py = ry * d + P[7];
    float pz = rz * d + P[11];
    float u = px / pz;
    float v = py / pz;
    float gx = u / ((float)(w - 1) / 2.0f) - 1.0f;
    float gy = v / ((float)(h - 1) / 2.0f) - 1.0f;
    float ix = (gx + 1.0f) * ((float)(w - 1) / 2.0f);
    float iy = (gy + 1.0f) * ((float)(h - 1) / 2.0f);
    t.ok_nw = t.ok_ne = t.ok_sw = t.ok_se = 0;
    t.nw = t.ne = t.sw = t.se = 0.0f;
    t.x0 = t.y0 = 0;
    /* non-finite or far-out coordinates sample nothing (output defined 0) */
    if (!(ix > -2.0f && ix < (float)w + 1.0f && iy > -2.0f && iy < (float)h + 1.0f)) return t;
    float fx = floorf(ix), fy = floorf(iy);
    int x0 = (int)fx, y0 = (int)fy;
    float x1f = fx + 1.0f, y1f = fy + 1.0f;
    t.x0 = x0; t.y0 = y0;
    t.nw = (x1f - ix) * (y1f - iy);
    t.ne = (ix - fx) * (y1f - iy);
    t.sw = (x1f - ix) * (iy - fy);
    t.se = (ix - fx) * (iy - fy);
    t.ok_nw = (x0 >= 0 && x0 < w && y0 >= 0 && y0 < h);
    t.ok_ne = (x0 + 1 >= 0 && x0 + 1 < w && y0 >= 0 && y0 < h);
    t.ok_sw = (x0 >= 0 && x0 < w && y0 + 1 >= 0 && y0 + 1 < h);
    t.ok_se = (x0 + 1 >= 0 && x0 + 1 < w && y0 + 1 >= 0 && y0 + 1 < h);
    return t;
}

static inline float sample_tap(const float* plane, int w, const tap_t* t) {
    float acc = 0.0f;
    if (t->ok_nw) acc += plane[(size_t)t->y0 * w + t->x0] * t->nw;
    if (t->ok_ne) acc += plane[(size_t)t->y0 * w + t->x0 + 1] * t->ne;
    if (t->ok_sw) acc += plane[(size_t)(t->y0 + 1) * w + t->x0] * t->sw;
    if (t->ok_se) acc += plane[(size_t)(t->y0 + 1) * w + t->x0 + 1] * t->se;
    return acc;
}

static inline float depth_at(const float* depth, int depth_is_map, int d, int y, int x, int h, int w) {
    return depth_is_map ? depth[((size_t)d * h + y) * w + x] : depth[d];
}

/* module.py:516-557 homo_warping_float.  src [C,h,w], proj34 from compose_proj,
 * depth [D] (depth_is_map=0) or [D,h,w] (depth_is_map=1), out [C,D,h,w]. */
API void d3d_oracle_homo_warp(const float* src, const float* proj34, const float* depth, int depth_is_map,
                              int C, int D, int h, int w, float* out) {
#pragma omp parallel for collapse(2) schedule(static)
    for (int d = 0; d < D; ++d)
        for (int y = 0; y < h; ++y)
            for (int x = 0; x < w; ++x) {
                float dv = depth_at(depth, depth_is_map, d, y, x, h, w);
                tap_t t = make_tap(proj34, (float)x, (float)y, dv, h, w);
                for (int c = 0; c < C; ++c)
                    out[(((size_t)c * D + d) * h + y) * w + x] = sample_tap(src + (size_t)c * h * w, w, &t);
            }
}

/* module.py:560-601 homo_warping_double.  src [C,h,w] fp32; ref44 / src44 row-major 4x4 DOUBLE; depth fp32.
 * proj = src44 @ inverse(ref44) in fp64 (:571), rot @ [x,y,1] (:581), * depth (:582-583), + trans (:584), the divide
 * (:585) and the normalisation x/((W-1)/2) - 1 (:586-587) in fp64; grid = .float() (:590); F.grid_sample
 * (bilinear, zeros, align_corners=True) then un-normalises in fp32 with ((g+1)/2)*(size-1). */
API int d3d_oracle_homo_warp_f64(const float* src, const double* ref44, const double* src44, const float* depth,
                                 int depth_is_map, int C, int D, int h, int w, float* out) {
    double a[4][8];
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) { a[i][j] = ref44[i * 4 + j]; a[i][4 + j] = (i == j) ? 1.0 : 0.0; }
    for (int col = 0; col < 4; ++col) {
        int piv = col;
        for (int r = col + 1; r < 4; ++r) if (fabs(a[r][col]) > fabs(a[piv][col])) piv = r;
        if (a[piv][col] == 0.0) return -1;
        for (int j = 0; j < 8; ++j) { double t = a[col][j]; a[col][j] = a[piv][j]; a[piv][j] = t; }
        double inv = 1.0 / a[col][col];
        for (int j = 0; j < 8; ++j) a[col][j] *= inv;
        for (int r = 0; r < 4; ++r) {
            if (r == col) continue;
            double f = a[r][col];
            for (int j = 0; j < 8; ++j) a[r][j] -= f * a[col][j];
        }
    }
    double P[12];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 4; ++j) {
            double s = 0.0;
            for (int k = 0; k < 4; ++k) s += src44[i * 4 + k] * a[k][4 + j];
            P[i * 4 + j] = s;
        }
#pragma omp parallel for collapse(2) schedule(static)
    for (int d = 0; d < D; ++d)
        for (int y = 0; y < h; ++y)
            for (int x = 0; x < w; ++x) {
                double dv = (double)depth_at(depth, depth_is_map, d, y, x, h, w);
                double rx = P[0] * x + P[1] * y + P[2], ry = P[4] * x + P[5] * y + P[6], rz = P[8] * x + P[9] * y + P[10];
                double px = rx * dv + P[3], py = ry * dv + P[7], pz = rz * dv + P[11];
                float gx = (float)((px / pz) / ((double)(w - 1) / 2.0) - 1.0);
                float gy = (float)((py / pz) / ((double)(h - 1) / 2.0) - 1.0);
                float ix = ((gx + 1.0f) / 2.0f) * (float)(w - 1);
                float iy = ((gy + 1.0f) / 2.0f) * (float)(h - 1);
                tap_t t;
                t.ok_nw = t.ok_ne = t.ok_sw = t.ok_se = 0;
                t.nw = t.ne = t.sw = t.se = 0.0f;
                t.x0 = t.y0 = 0;
                if (ix > -2.0f && ix < (float)w + 1.0f && iy > -2.0f && iy < (float)h + 1.0f) {
                    float fx = floorf(ix), fy = floorf(iy);
                    int x0 = (int)fx, y0 = (int)fy;
                    float x1f = fx + 1.0f, y1f = fy + 1.0f;
                    t.x0 = x0; t.y0 = y0;
                    t.nw = (x1f - ix) * (y1f - iy); t.ne = (ix - fx) * (y1f - iy);
                    t.sw = (x1f - ix) * (iy - fy); t.se = (ix - fx) * (iy - fy);
                    t.ok_nw = (x0 >= 0 && x0 < w && y0 >= 0 && y0 < h);
                    t.ok_ne = (x0 + 1 >= 0 && x0 + 1 < w && y0 >= 0 && y0 < h);
                    t.ok_sw = (x0 >= 0 && x0 < w && y0 + 1 >= 0 && y0 + 1 < h);
                    t.ok_se = (x0 + 1 >= 0 && x0 + 1 < w && y0 + 1 >= 0 && y0 + 1 < h);
                }
                for (int c = 0; c < C; ++c)
                    out[(((size_t)c * D + d) * h + y) * w + x] = sample_tap(src + (size_t)c * h * w, w, &t);
            }
    return 0;
}

/* cas_mvsnet.py:45-60 (same arithmetic: ucsnet.py:119-134, msrednet.py:217-230,400-414)
 * variance cost volume.  ref [C,h,w]; srcs [V-1,C,h,w]; projs [V-1,12]; out [C,D,h,w].
 *   sum = ref + sum_i warp_i ; sq = ref^2 + sum_i warp_i^2 ; var = sq/V - (sum/V)^2 */
API void d3d_oracle_variance_volume(const float* ref, const float* srcs, const float* projs, const float* depth,
                                    int depth_is_map, int V, int C, int D, int h, int w, float* out) {
    const size_t plane = (size_t)h * w;
#pragma omp parallel for collapse(2) schedule(static)
    for (int d = 0; d < D; ++d)
        for (int y = 0; y < h; ++y)
            for (int x = 0; x < w; ++x) {
                float dv = depth_at(depth, depth_is_map, d, y, x, h, w);
                tap_t taps[16];
                for (int i = 0; i < V - 1; ++i) taps[i] = make_tap(projs + 12 * i, (float)x, (float)y, dv, h, w);
                for (int c = 0; c < C; ++c) {
                    float r = ref[c * plane + (size_t)y * w + x];
                    float s = r, q = r * r;
                    for (int i = 0; i < V - 1; ++i) {
                        float wv = sample_tap(srcs + ((size_t)i * C + c) * plane, w, &taps[i]);
                        s += wv;
                        q += wv * wv;
                    }
                    float m = s / (float)V;
                    out[(((size_t)c * D + d) * h + y) * w + x] = q / (float)V - m * m;
                }
            }
}

/* adamvs.py:469-474 per-pair channel-mean correlation: out[d] = mean_c(ref[c]*warp_d[c]).
 * ref, src [C,h,w]; out [D,h,w]. */
API void d3d_oracle_pair_corr_mean(const float* ref, const float* src, const float* proj34, const float* depth,
                                   int depth_is_map, int C, int D, int h, int w, float* out) {
    const size_t plane = (size_t)h * w;
#pragma omp parallel for collapse(2) schedule(static)
    for (int d = 0; d < D; ++d)
        for (int y = 0; y < h; ++y)
            for (int x = 0; x < w; ++x) {
                float dv = depth_at(depth, depth_is_map, d, y, x, h, w);
                tap_t t = make_tap(proj34, (float)x, (float)y, dv, h, w);
                float acc = 0.0f;
                for (int c = 0; c < C; ++c)
                    acc += ref[c * plane + (size_t)y * w + x] * sample_tap(src + c * plane, w, &t);
                out[((size_t)d * h + y) * w + x] = acc / (float)C;
            }
}

/* adamvs.py:492-509 visibility-weighted correlation:
 *   sim[c] = sum_i (warp_i[c]*ref[c]) * vw_i / (1e-5 + sum_i vw_i)
 * weights [V-1,h,w] already at this stage's resolution.  out [C,D,h,w]. */
API void d3d_oracle_weighted_corr(const float* ref, const float* srcs, const float* projs, const float* weights,
                                  const float* depth, int depth_is_map, int V, int C, int D, int h, int w,
                                  float* out) {
    const size_t plane = (size_t)h * w;
#pragma omp parallel for collapse(2) schedule(static)
    for (int d = 0; d < D; ++d)
        for (int y = 0; y < h; ++y)
            for (int x = 0; x < w; ++x) {
                float dv = depth_at(depth, depth_is_map, d, y, x, h, w);
                tap_t taps[16];
                float den = 1e-5f;
                for (int i = 0; i < V - 1; ++i) {
                    taps[i] = make_tap(projs + 12 * i, (float)x, (float)y, dv, h, w);
                    den += weights[i * plane + (size_t)y * w + x];
                }
                for (int c = 0; c < C; ++c) {
                    float r = ref[c * plane + (size_t)y * w + x];
                    float num = 0.0f;
                    for (int i = 0; i < V - 1; ++i) {
                        float wv = sample_tap(srcs + ((size_t)i * C + c) * plane, w, &taps[i]);
                        num += (wv * r) * weights[i * plane + (size_t)y * w + x];
                    }
                    out[(((size_t)c * D + d) * h + y) * w + x] = num / den;
                }
            }
}

/* cas_mvsnet.py:69-76 + module.py:605-613: softmax over D, soft-argmin depth, and the
 * 4-plane-window confidence around trunc(sum_k p_k*k).
 * cost [D,h,w]; depth [D] or [D,h,w]; depth_out, conf_out [h,w]. */
API void d3d_oracle_softargmin_conf4(const float* cost, const float* depth, int depth_is_map, int D, int h, int w,
                                     float* depth_out, float* conf_out) {
    const size_t plane = (size_t)h * w;
#pragma omp parallel for schedule(static)
    for (long i = 0; i < (long)plane; ++i) {
        float mx = -INFINITY;
        for (int d = 0; d < D; ++d) mx = fmaxf(mx, cost[d * plane + i]);
        float den = 0.0f;
        for (int d = 0; d < D; ++d) den += expf(cost[d * plane + i] - mx);
        float dep = 0.0f, idxf = 0.0f;
        for (int d = 0; d < D; ++d) {
            float p = expf(cost[d * plane + i] - mx) / den;
            float dv = depth_is_map ? depth[d * plane + i] : depth[d];
            dep += p * dv;
            idxf += p * (float)d;
        }
        long k = (long)idxf; /* .long() truncation */
        if (k < 0) k = 0;
        if (k > D - 1) k = D - 1;
        float conf = 0.0f;
        for (long j = k - 1; j <= k + 2; ++j) {
            float p = (j >= 0 && j < D) ? expf(cost[j * plane + i] - mx) / den : 0.0f;
            conf += p;
        }
        depth_out[i] = dep;
        conf_out[i] = conf;
    }
}

/* F.interpolate(x, [2h,2w], mode='bilinear', align_corners=False) as used for the depth
 * plane in adamvs.py:519-520 (and the view-weight resize at adamvs.py:502 when ratios
 * are 2x).  General output size so it also covers 4x.  in [h,w] -> out [H,W]. */
API void d3d_oracle_resize_bilinear(const float* in, int h, int w, int H, int W, float* out) {
    const float sy = (float)h / (float)H, sx = (float)w / (float)W;
#pragma omp parallel for schedule(static)
    for (int Y = 0; Y < H; ++Y) {
        float fy = ((float)Y + 0.5f) * sy - 0.5f;
        if (fy < 0.0f) fy = 0.0f;
        int y0 = (int)fy;
        int y1 = y0 + (y0 < h - 1 ? 1 : 0);
        float ly = fy - (float)y0, hy = 1.0f - ly;
        for (int X = 0; X < W; ++X) {
            float fx = ((float)X + 0.5f) * sx - 0.5f;
            if (fx < 0.0f) fx = 0.0f;
            int x0 = (int)fx;
            int x1 = x0 + (x0 < w - 1 ? 1 : 0);
            float lx = fx - (float)x0, hx = 1.0f - lx;
            out[(size_t)Y * W + X] = hy * (hx * in[(size_t)y0 * w + x0] + lx * in[(size_t)y0 * w + x1]) +
                                     ly * (hx * in[(size_t)y1 * w + x0] + lx * in[(size_t)y1 * w + x1]);
        }
    }
}

/* adamvs.py:514-525 (same: msrednet.py:418-429) one plane of the online regression.
 * reg, dplane [n]; max_p, sum_d, sum_p [n] updated in place.
 *   p = exp(reg); max_p = p if max_p < p else max_p; sum_d += d*p; sum_p += p */
API void d3d_oracle_online_regress_update(const float* reg, const float* dplane, long n, float* max_p,
                                          float* sum_d, float* sum_p) {
#pragma omp parallel for schedule(static)
    for (long i = 0; i < n; ++i) {
        float p = expf(reg[i]);
        float flag = (max_p[i] < p) ? 1.0f : 0.0f;
        max_p[i] = flag * p + (1.0f - flag) * max_p[i];
        sum_d[i] = dplane[i] * p + sum_d[i];
        sum_p[i] = sum_p[i] + p;
    }
}

/* adamvs.py:527-529: depth = sum_d/(sum_p+1e-10); conf = max_p/(sum_p+1e-10) */
API void d3d_oracle_online_regress_finalize(const float* max_p, const float* sum_d, const float* sum_p, long n,
                                            float* depth_out, float* conf_out) {
#pragma omp parallel for schedule(static)
    for (long i = 0; i < n; ++i) {
        float e = sum_p[i] + 1e-10f;
        depth_out[i] = sum_d[i] / e;
        conf_out[i] = max_p[i] / e;
    }
}

/* module.py:616-650 depth hypotheses.
 * mode 0 (cur_depth is [2] = min,max; module.py:637-645): out[d] = min + d*(max-min)/(D-1), tiled.
 * mode 1 (cur_depth is [h,w]; module.py:616-630):
 *   lo = cur - D/2*interval ; hi = cur + D/2*interval ; out[d] = lo + d*((hi-lo)/(D-1))
 * out [D,h,w]. */
API void d3d_oracle_depth_range_samples(const float* cur_depth, int mode, int D, float interval, int h, int w,
                                        float* out) {
    const size_t plane = (size_t)h * w;
    if (mode == 0) {
        float lo = cur_depth[0], hi = cur_depth[1];
        float step = (hi - lo) / (float)(D - 1);
        for (int d = 0; d < D; ++d) {
            float v = lo + (float)d * step;
            for (size_t i = 0; i < plane; ++i) out[d * plane + i] = v;
        }
        return;
    }
    for (size_t i = 0; i < plane; ++i) {
        float lo = cur_depth[i] - (float)D / 2.0f * interval;
        float hi = cur_depth[i] + (float)D / 2.0f * interval;
        float step = (hi - lo) / (float)(D - 1);
        for (int d = 0; d < D; ++d) out[d * plane + i] = lo + (float)d * step;
    }
}

/* ------------------------------------------------------------------------------------
 * Convolution family used by the regularisers (naive direct form, fp32):
 *   nn.Conv3d k=3 pad=1 stride 1|2, bias optional           (module.py:297-304, cas_mvsnet.py:84-110)
 *   nn.ConvTranspose3d k=3 pad=1 out_pad=1 stride=2          (cas_mvsnet.py:94-108)
 *   eval-mode BatchNorm as a per-channel affine + optional ReLU, optional skip add
 * in [Ci,D,H,W], weight conv: [Co,Ci,3,3,3]; convT: [Ci,Co,3,3,3] (PyTorch layouts).
 * ---------------------------------------------------------------------------------- */
API void d3d_oracle_conv3d_k3(const float* in, const float* wt, const float* bias, int Ci, int Co, int D, int H,
                              int W, int stride, float* out) {
    const int Do = (D + 2 - 3) / stride + 1, Ho = (H + 2 - 3) / stride + 1, Wo = (W + 2 - 3) / stride + 1;
#pragma omp parallel for collapse(2) schedule(static)
    for (int co = 0; co < Co; ++co)
        for (int z = 0; z < Do; ++z)
            for (int y = 0; y < Ho; ++y)
                for (int x = 0; x < Wo; ++x) {
                    float acc = bias ? bias[co] : 0.0f;
                    for (int ci = 0; ci < Ci; ++ci)
                        for (int kz = 0; kz < 3; ++kz) {
                            int iz = z * stride - 1 + kz;
                            if (iz < 0 || iz >= D) continue;
                            for (int ky = 0; ky < 3; ++ky) {
                                int iy = y * stride - 1 + ky;
                                if (iy < 0 || iy >= H) continue;
                                for (int kx = 0; kx < 3; ++kx) {
                                    int ix = x * stride - 1 + kx;
                                    if (ix < 0 || ix >= W) continue;
                                    acc += in[(((size_t)ci * D + iz) * H + iy) * W + ix] *
                                           wt[((((size_t)co * Ci + ci) * 3 + kz) * 3 + ky) * 3 + kx];
                                }
                            }
                        }
                    out[(((size_t)co * Do + z) * Ho + y) * Wo + x] = acc;
                }
}

API void d3d_oracle_convtranspose3d_k3s2(const float* in, const float* wt, const float* bias, int Ci, int Co, int D,
                                         int H, int W, float* out) {
    /* k=3, stride=2, padding=1, output_padding=1 -> output dims exactly 2x */
    const int Do = 2 * D, Ho = 2 * H, Wo = 2 * W;
#pragma omp parallel for collapse(2) schedule(static)
    for (int co = 0; co < Co; ++co)
        for (int z = 0; z < Do; ++z)
            for (int y = 0; y < Ho; ++y)
                for (int x = 0; x < Wo; ++x) {
                    float acc = bias ? bias[co] : 0.0f;
                    for (int ci = 0; ci < Ci; ++ci)
                        for (int kz = 0; kz < 3; ++kz) {
                            int tz = z + 1 - kz;
                            if (tz < 0 || (tz & 1) || (tz >> 1) >= D) continue;
                            for (int ky = 0; ky < 3; ++ky) {
                                int ty = y + 1 - ky;
                                if (ty < 0 || (ty & 1) || (ty >> 1) >= H) continue;
                                for (int kx = 0; kx < 3; ++kx) {
                                    int tx = x + 1 - kx;
                                    if (tx < 0 || (tx & 1) || (tx >> 1) >= W) continue;
                                    acc += in[(((size_t)ci * D + (tz >> 1)) * H + (ty >> 1)) * W + (tx >> 1)] *
                                           wt[((((size_t)ci * Co + co) * 3 + kz) * 3 + ky) * 3 + kx];
                                }
                            }
                        }
                    out[(((size_t)co * Do + z) * Ho + y) * Wo + x] = acc;
                }
}

/* 2D: nn.Conv2d k=3 pad=1 stride 1|2 (module.py ConvReLU/ConvBnReLU, ConvGRUCell). */
API void d3d_oracle_conv2d_k3(const float* in, const float* wt, const float* bias, int Ci, int Co, int H, int W,
                              int stride, float* out) {
    const int Ho = (H + 2 - 3) / stride + 1, Wo = (W + 2 - 3) / stride + 1;
#pragma omp parallel for collapse(2) schedule(static)
    for (int co = 0; co < Co; ++co)
        for (int y = 0; y < Ho; ++y)
            for (int x = 0; x < Wo; ++x) {
                float acc = bias ? bias[co] : 0.0f;
                for (int ci = 0; ci < Ci; ++ci)
                    for (int ky = 0; ky < 3; ++ky) {
                        int iy = y * stride - 1 + ky;
                        if (iy < 0 || iy >= H) continue;
                        for (int kx = 0; kx < 3; ++kx) {
                            int ix = x * stride - 1 + kx;
                            if (ix < 0 || ix >= W) continue;
                            acc += in[((size_t)ci * H + iy) * W + ix] * wt[(((size_t)co * Ci + ci) * 3 + ky) * 3 + kx];
                        }
                    }
                out[((size_t)co * Ho + y) * Wo + x] = acc;
            }
}

/* nn.ConvTranspose2d k=3 stride=2 pad=1 out_pad=1 (adamvs.py:411-414). wt [Ci,Co,3,3]. */
API void d3d_oracle_convtranspose2d_k3s2(const float* in, const float* wt, const float* bias, int Ci, int Co, int H,
                                         int W, float* out) {
    const int Ho = 2 * H, Wo = 2 * W;
#pragma omp parallel for collapse(2) schedule(static)
    for (int co = 0; co < Co; ++co)
        for (int y = 0; y < Ho; ++y)
            for (int x = 0; x < Wo; ++x) {
                float acc = bias ? bias[co] : 0.0f;
                for (int ci = 0; ci < Ci; ++ci)
                    for (int ky = 0; ky < 3; ++ky) {
                        int ty = y + 1 - ky;
                        if (ty < 0 || (ty & 1) || (ty >> 1) >= H) continue;
                        for (int kx = 0; kx < 3; ++kx) {
                            int tx = x + 1 - kx;
                            if (tx < 0 || (tx & 1) || (tx >> 1) >= W) continue;
                            acc += in[((size_t)ci * H + (ty >> 1)) * W + (tx >> 1)] *
                                   wt[(((size_t)ci * Co + co) * 3 + ky) * 3 + kx];
                        }
                    }
                out[((size_t)co * Ho + y) * Wo + x] = acc;
            }
}

/* Eval-mode BatchNorm (y = (x-mean)/sqrt(var+eps)*gamma+beta), then optional ReLU, then
 * optional skip add AFTER the ReLU (cas_mvsnet.py:116-118: x = conv4 + conv7(x)).
 * x [C,n] in place. */
API void d3d_oracle_bn_relu_add(float* x, int C, long n, const float* gamma, const float* beta, const float* mean,
                                const float* var, float eps, int relu, const float* skip) {
#pragma omp parallel for schedule(static)
    for (int c = 0; c < C; ++c) {
        float inv = 1.0f / sqrtf(var[c] + eps);
        for (long i = 0; i < n; ++i) {
            float v = (x[c * n + i] - mean[c]) * inv * gamma[c] + beta[c];
            if (relu && v < 0.0f) v = 0.0f;
            if (skip) v = skip[c * n + i] + v;
            x[c * n + i] = v;
        }
    }
}

/* module.py:24-51 ConvGRUCell.forward given the two conv outputs is elementwise; the
 * convs themselves are d3d_oracle_conv2d_k3 on the concatenated input.  This helper does
 * the gate math:  r,u = sigmoid(gates[0:H]), sigmoid(gates[H:2H]);  (caller then convolves
 * cat(x, r*h))  and  h' = u*h + (1-u)*tanh(convc).
 */
API void d3d_oracle_sigmoid(float* x, long n) {
#pragma omp parallel for schedule(static)
    for (long i = 0; i < n; ++i) x[i] = 1.0f / (1.0f + expf(-x[i]));
}

API void d3d_oracle_gru_update(const float* u, const float* h, const float* convc, long n, float* out) {
#pragma omp parallel for schedule(static)
    for (long i = 0; i < n; ++i) out[i] = u[i] * h[i] + (1.0f - u[i]) * tanhf(convc[i]);
}

/* module.py:62-67,78-79,87 nn.GroupNorm(1, C, eps, affine): one group = statistics over all C*plane
 * elements (biased variance), then per-channel gamma/beta.  In place. */
API void d3d_oracle_groupnorm1(float* x, int C, long plane, const float* gamma, const float* beta, float eps) {
    const long n = (long)C * plane;
    double s = 0.0, q = 0.0;
    for (long i = 0; i < n; ++i) { s += x[i]; q += (double)x[i] * x[i]; }
    const double m = s / (double)n;
    double var = q / (double)n - m * m;
    if (var < 0.0) var = 0.0;
    const float mean = (float)m, rstd = (float)(1.0 / sqrt(var + (double)eps));
    for (int c = 0; c < C; ++c)
        for (long i = 0; i < plane; ++i) {
            float* p = x + (long)c * plane + i;
            *p = (*p - mean) * rstd * gamma[c] + beta[c];
        }
}
