/*
 * fusion_oracle.c -- CPU restatement of the reference's geometric consistency check and per-reference-view
 * fusion accumulators.
 *
 * TEST INFRASTRUCTURE ONLY: the parity checker for deep3d_aerial_amd/csrc/fusion.hip.  Only tests/ may load it.
 *
 * Parity pinning: the reference implements this path with CuPy (fuse/consistency_check_n.py:10), which is not
 * installed in the build container, and ships no fixtures for it.  tests/golden/make_golden.py therefore runs the
 * reference's ConsistencyChecker.check_cupy with the name `cupy` bound to a thin module that forwards to NumPy
 * (CuPy mirrors NumPy's API and promotion rules) on scenes whose reprojections stay inside the source image, and
 * the oracle is checked against those outputs (tests/test_oracle_golden.py).  What that cannot pin -- CuPy's
 * wrap-around of out-of-range indices (NumPy raises instead) and the float -> int conversion of non-finite
 * values -- is restated from CuPy's documented behaviour and is marked "unpinned" in DESIGN.md.
 *
 * One function per reference block; plain scalar C, one pixel at a time, every rounding point of the reference's
 * mixed float32 / float64 array expressions kept (see the comments).  Camera matrices arrive as doubles holding
 * float32 values: the caller (oracle/fusion.py) forms the inverses and the E_src @ inv(E_ref) product in float32
 * with NumPy exactly as the reference does.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define API __attribute__((visibility("default")))

/* layout of the camera block, 94 doubles (same as include/deep3d_planesweep.h) */
enum { KRI = 0, M1 = 9, KS = 21, KSI = 30, ESI = 39, ER = 55, KR = 67, RSI = 76, RRI = 85, NCAM = 94 };

/* ndarray.astype(int) of a float64 on x86-64: cvttsd2si -- truncation; NaN / inf / out of range give INT64_MIN */
static int64_t to_i64(double v) { return (fabs(v) < 9.2e18) ? (int64_t)v : INT64_MIN; }

/* CuPy integer-array indexing: out-of-range indices wrap around (Python modulo) */
static int64_t wrap(int64_t i, int64_t n) {
    int64_t r = i % n;
    return r < 0 ? r + n : r;
}

static void m3(const double* m, double a, double b, double c, double* o) {
    for (int r = 0; r < 3; ++r) o[r] = m[3 * r] * a + m[3 * r + 1] * b + m[3 * r + 2] * c;
}

static void m3f(const double* m, float a, float b, float c, float* o) {
    for (int r = 0; r < 3; ++r) o[r] = (float)m[3 * r] * a + (float)m[3 * r + 1] * b + (float)m[3 * r + 2] * c;
}

/*
 * fuse/consistency_check_n.py:29-138 (check_cupy).  Outputs as the reference returns them: mask [H,W] (u8),
 * depth_reprojected [H,W], depth_src_out [Hs,Ws] (copy of depth_src with the consistent samples zeroed),
 * xyz_world_src [3,H,W], angle_conf [3,H,W].
 */
API void d3d_oracle_consistency_check(const float* depth_ref, const float* normal_ref, const float* prob_ref,
                                      const float* depth_src, const float* normal_src, const double* cam, int H, int W,
                                      int Hs, int Ws, double pos_thr, float depth_thr, float normal_thr, float conf_thr,
                                      unsigned char* mask, float* depth_reprojected, float* depth_src_out,
                                      float* xyz_world_src, float* angle_conf) {
    const int64_t plane = (int64_t)H * W;
    memcpy(depth_src_out, depth_src, sizeof(float) * (size_t)Hs * Ws); /* :36 cp.array copy */
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            const int64_t i = (int64_t)y * W + x;
            const float dref = depth_ref[i];
            const double d = dref;
            double pr[3], ps[3], pk[3];
            /* :53-55 inv(K_ref) @ ([x, y, 1] * depth): int64 grid * float32 depth -> float64 */
            m3(cam + KRI, (double)x * d, (double)y * d, d, pr);
            /* :58-59 (E_src @ inv(E_ref)) @ [xyz_ref; 1] */
            for (int r = 0; r < 3; ++r)
                ps[r] = cam[M1 + 4 * r] * pr[0] + cam[M1 + 4 * r + 1] * pr[1] + cam[M1 + 4 * r + 2] * pr[2] + cam[M1 + 4 * r + 3];
            /* :64-65 K_src @ xyz_src, perspective divide */
            m3(cam + KS, ps[0], ps[1], ps[2], pk);
            /* :70-71 nearest pixel: (+0.5).astype(int) */
            const int64_t xi = to_i64(pk[0] / pk[2] + 0.5), yi = to_i64(pk[1] / pk[2] + 0.5);
            /* :72-73 depth_src[y_src, x_src], normal_src[y_src, x_src, :] */
            const int64_t si = wrap(yi, Hs) * Ws + wrap(xi, Ws);
            const float dsrc = depth_src[si];
            const float* ns = normal_src + 3 * si;
            /* :77-78 inv(K_src) @ ([x_src, y_src, 1] * sampled depth) */
            double pb[3], pw[4], pq[3], kq[3];
            const double ds = dsrc;
            m3(cam + KSI, (double)xi * ds, (double)yi * ds, ds, pb);
            /* :81-82 inv(E_src) @ [xyz; 1] */
            for (int r = 0; r < 4; ++r)
                pw[r] = cam[ESI + 4 * r] * pb[0] + cam[ESI + 4 * r + 1] * pb[1] + cam[ESI + 4 * r + 2] * pb[2] + cam[ESI + 4 * r + 3];
            /* :85 E_ref @ src_world */
            for (int r = 0; r < 3; ++r)
                pq[r] = cam[ER + 4 * r] * pw[0] + cam[ER + 4 * r + 1] * pw[1] + cam[ER + 4 * r + 2] * pw[2] + cam[ER + 4 * r + 3] * pw[3];
            const float drep = (float)pq[2]; /* :87 .astype(float32) */
            m3(cam + KR, pq[0], pq[1], pq[2], kq);
            const float xr = (float)(kq[0] / kq[2]), yr = (float)(kq[1] / kq[2]); /* :88-91 */
            /* :94 float32 - int64 grid -> float64 */
            const double ex = (double)xr - (double)x, ey = (double)yr - (double)y;
            const double dist = sqrt(ex * ex + ey * ey);
            /* :97-98 float32 */
            const float rel = fabsf(drep - dref) / dref;
            /* :101-111 world normals (float32 matmul), cosine */
            float sw[3], rw[3];
            m3f(cam + RSI, ns[0], ns[1], ns[2], sw);
            m3f(cam + RRI, normal_ref[3 * i], normal_ref[3 * i + 1], normal_ref[3 * i + 2], rw);
            float cs = rw[0] * sw[0] + rw[1] * sw[1] + rw[2] * sw[2];
            cs = cs / (sqrtf(rw[0] * rw[0] + rw[1] * rw[1] + rw[2] * rw[2]) * sqrtf(sw[0] * sw[0] + sw[1] * sw[1] + sw[2] * sw[2]));
            /* :116-119 */
            const int m = (dist < pos_thr) && (rel < depth_thr) && (prob_ref[i] > conf_thr) && (cs > normal_thr) && (dref > 0.0f);
            mask[i] = (unsigned char)m;
            depth_reprojected[i] = m ? drep : 0.0f; /* :121 */
            if (m) { /* :123-126: the integer indices pass through (+0.5).astype(int) again */
                const int64_t x2 = to_i64((double)xi + 0.5), y2 = to_i64((double)yi + 0.5);
                depth_src_out[wrap(y2, Hs) * Ws + wrap(x2, Ws)] = 0.0f;
            }
            for (int r = 0; r < 3; ++r) {
                xyz_world_src[r * plane + i] = m ? (float)pw[r] : 0.0f;          /* :128-131 */
                angle_conf[r * plane + i] = m ? (cs < 0.0f ? 0.0f : cs) : 0.0f;  /* :113, :133-136 */
            }
        }
}

/*
 * fuse/fusion_3d_normal.py:452-474: world points of the reference view's own depth map (float32), confidence 1,
 * consistency count 1, unit world normals.  cam slots used: inv(K_ref), ESI (holding inv(E_ref)), RRI.
 */
API void d3d_oracle_fusion_ref_init(const float* depth_ref, const float* normal_ref, const double* cam, int H, int W,
                                    float* all_xyz_world, float* conf_sum, int32_t* geo_mask_sum, float* normal_world) {
    const int64_t plane = (int64_t)H * W;
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            const int64_t i = (int64_t)y * W + x;
            const double d = depth_ref[i];
            double pr[3];
            m3(cam + KRI, (double)x * d, (double)y * d, d, pr); /* :457-458 */
            for (int r = 0; r < 3; ++r)                         /* :459-461 */
                all_xyz_world[r * plane + i] =
                    (float)(cam[ESI + 4 * r] * pr[0] + cam[ESI + 4 * r + 1] * pr[1] + cam[ESI + 4 * r + 2] * pr[2] + cam[ESI + 4 * r + 3]);
            conf_sum[i] = 1.0f;  /* :463 */
            geo_mask_sum[i] = 1; /* :474 */
            float rw[3];         /* :466-469 */
            m3f(cam + RRI, normal_ref[3 * i], normal_ref[3 * i + 1], normal_ref[3 * i + 2], rw);
            const float n = sqrtf(rw[0] * rw[0] + rw[1] * rw[1] + rw[2] * rw[2]);
            for (int r = 0; r < 3; ++r) normal_world[3 * i + r] = rw[r] / n;
        }
}

/* fuse/fusion_3d_normal.py:513-518: fold one pair's check outputs into the accumulators */
API void d3d_oracle_fusion_accumulate(const unsigned char* mask, const float* xyz_world_src, const float* angle_conf,
                                      int H, int W, int src_idx, int32_t* geo_mask_sum, float* all_xyz_world,
                                      float* conf_sum, int32_t* vis) {
    const int64_t plane = (int64_t)H * W;
    for (int64_t i = 0; i < plane; ++i) {
        geo_mask_sum[i] += mask[i];
        for (int r = 0; r < 3; ++r) all_xyz_world[r * plane + i] += angle_conf[r * plane + i] * xyz_world_src[r * plane + i];
        conf_sum[i] += angle_conf[i];
        vis[i] = mask[i] ? src_idx : 0;
    }
}

/* fuse/fusion_3d_normal.py:522-527 */
API void d3d_oracle_fusion_finalize(const float* all_xyz_world, const float* conf_sum, const int32_t* geo_mask_sum, int H,
                                    int W, int min_num, float* avg_xyz_world, unsigned char* final_mask) {
    const int64_t plane = (int64_t)H * W;
    for (int64_t i = 0; i < plane; ++i) {
        for (int r = 0; r < 3; ++r) avg_xyz_world[r * plane + i] = all_xyz_world[r * plane + i] / conf_sum[i];
        final_mask[i] = geo_mask_sum[i] >= min_num;
    }
}


/* fuse/fusion_3d_normal.py:545-570: vertices of one reference view.  valid = final_mask (row-major order, as boolean
 * indexing yields them, :546-552); every skip_line-th VALID point (:554) that lies strictly inside the block in x and y
 * (:558) becomes a vertex with views = sorted(vis[vis > 0] - 1) (:559-562), colour = (ref_img * 255).astype(int) (:551),
 * normal = ref_normal_est_world (:552).  `len(all_vis_infos[0]) > 1` (:555): a view with a single valid point emits nothing.
 * Returns the number of vertices; outputs must hold that many rows (call with NULL outputs to count). */
API int64_t d3d_oracle_fusion_points(const float* avg_xyz_world, const unsigned char* final_mask, const int32_t* const* vis,
                                     int n_vis, const float* color, const float* normal_world, int H, int W, int skip_line,
                                     const double* scene_range, float* out_xyz, int32_t* out_color, float* out_normal,
                                     int32_t* out_views, int32_t* out_nviews) {
    const int64_t plane = (int64_t)H * W;
    int64_t n_valid = 0;
    for (int64_t i = 0; i < plane; ++i) n_valid += final_mask[i] ? 1 : 0;
    if (n_valid <= 1) return 0;
    int64_t ord = 0, n = 0;
    for (int64_t i = 0; i < plane; ++i) {
        if (!final_mask[i]) continue;
        const int64_t k = ord++;
        if (k % skip_line != 0) continue;
        const double x = (double)avg_xyz_world[i], y = (double)avg_xyz_world[plane + i];
        if (!(scene_range[0] < x && x < scene_range[1] && scene_range[2] < y && y < scene_range[3])) continue;
        if (out_xyz) {
            for (int c = 0; c < 3; ++c) out_xyz[n * 3 + c] = avg_xyz_world[(int64_t)c * plane + i];
            if (color) for (int c = 0; c < 3; ++c) out_color[n * 3 + c] = (int32_t)(color[i * 3 + c] * 255.0f);
            if (normal_world) for (int c = 0; c < 3; ++c) out_normal[n * 3 + c] = normal_world[i * 3 + c];
            int nv = 0;
            int32_t* row = out_views + n * n_vis;
            for (int v = 0; v < n_vis; ++v) {
                const int32_t id = vis[v][i];
                if (id > 0) {
                    int j = nv++;
                    while (j > 0 && row[j - 1] > id - 1) { row[j] = row[j - 1]; --j; }
                    row[j] = id - 1;
                }
            }
            for (int v = nv; v < n_vis; ++v) row[v] = -1;
            out_nviews[n] = nv;
        }
        ++n;
    }
    return n;
}
