/*
 * deep3d_planesweep.h -- C ABI of the MI355X (gfx950) plane-sweep cost-volume engine.
 *
 * The reference (gpcv-liujin/Deep3D_Aerial) has no FFI or plugin API for this path
 * (SURVEY.md F3): its operators are Python functions and nn.Module bodies that call
 * PyTorch ATen kernels.  Each entry point below therefore replaces one such Python
 * operator (or one fused group of them) and cites it; the host-side mirror that keeps
 * the reference's names and argument meaning lives in the deep3d_aerial_amd Python package and binds
 * these symbols with ctypes (see INTEGRATION.md for the binding a maintainer would add
 * on the reference side).
 *
 * Conventions
 *   - Plain C: raw DEVICE pointers owned by the caller, explicit sizes, a HIP stream,
 *     int status.  No torch types, no exceptions across the boundary, no hidden
 *     synchronisation, no allocation: every call is stream-ordered and reentrant.  Scratch
 *     memory is the caller's: d3d_sweep_workspace_bytes() says how much a sweep can use and
 *     the sweep entry points take the buffer as (workspace, workspace_bytes).
 *   - All tensors are contiguous fp32, batch handled by the caller (the reference runs
 *     inference at batch 1, predict.py:49):
 *         features  [C,h,w]      cost volume [C,D,h,w]      maps [h,w] / [D,h,w]
 *   - Depth hypotheses are given either per plane (depth_mode = D3D_DEPTH_PER_PLANE,
 *     pointer to [D]) or per pixel (D3D_DEPTH_PER_PIXEL, pointer to [D,h,w]); both are
 *     accepted by the reference's homo_warping_float (module.py:520-521,539).
 *     D3D_DEPTH_AFFINE is the per-pixel form without the volume: the pointer is to [2,h,w] =
 *     (lo, step) maps and plane k of pixel (y,x) lies at lo[y,x] + k * step[y,x] (fp32, the
 *     product rounded, then the sum: exactly the statement of module.py:616-631, whose
 *     hypotheses are affine in the plane index).  A sweep in this mode reads two maps instead
 *     of D planes, and its results are bit-identical to the D3D_DEPTH_PER_PIXEL sweep over
 *     the volume those maps generate.  Taken by the aggregation and regression entry points
 *     (d3d_variance_volume*, d3d_weighted_corr, d3d_pair_corr_mean, d3d_softargmin_conf4*);
 *     d3d_homo_warp, d3d_homo_warp_f64coord and d3d_pair_softmax_max take modes 0 and 1 only.
 *   - proj34 is the composed homography of module.py:528-530,
 *     (src_proj @ inverse(ref_proj))[:3,:4] = [rot | trans], row-major 12 floats per
 *     source view, in DEVICE memory (d3d_compose_projections produces it).
 *   - Return value: D3D_OK, or a negative D3D_ERR_*; d3d_last_error() gives the text of
 *     the calling thread's last failure.
 *   - Sampling semantics everywhere: bilinear, per-tap zero padding,
 *     align_corners=True (module.py:548-553).  Samples whose projected
 *     coordinate is non-finite contribute 0.
 */
#ifndef DEEP3D_PLANESWEEP_H
#define DEEP3D_PLANESWEEP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define D3D_ABI_VERSION 10

#define D3D_OK 0
#define D3D_ERR_INVALID_ARG (-1)
#define D3D_ERR_UNSUPPORTED (-2)
#define D3D_ERR_HIP (-3)

#define D3D_DEPTH_PER_PLANE 0
#define D3D_DEPTH_PER_PIXEL 1
#define D3D_DEPTH_AFFINE 2

#define D3D_MAX_VIEWS 16

/* hipStream_t without dragging HIP headers into C callers. */
typedef void* d3d_stream_t;

/* ABI version of the loaded library (== D3D_ABI_VERSION it was built with). */
int d3d_version(void);

/* Text of the calling thread's last error ("" if none). Never NULL. */
const char* d3d_last_error(void);

/*
 * module.py:528-530 -- proj = matmul(src_proj, inverse(ref_proj)); rot, trans.
 * proj44: device [V,4,4] (index 0 = reference view).  out34: device [V-1,12].
 * The 4x4 inverse and product are evaluated in fp64 on the device and rounded once.
 */
int d3d_compose_projections(const float* proj44, int n_views, float* out34, d3d_stream_t stream);

/*
 * Test hook, process-wide: 0 = the dispatcher chooses (default), 1 = direct-gather kernel, 2 = LDS-ring kernel, 3 = window kernel
 * (D3D_ERR_UNSUPPORTED where that kernel does not take the shape).  The parity suite runs every sweep case on
 * all of them.  Not for production callers.
 */
int d3d_debug_force_path(int path);

/*
 * Test hook, process-wide: out4[1] / out4[2] / out4[3] = sweep calls (homo_warp, variance, weighted, pair) served so far by the
 * direct-gather / LDS-ring / window kernel; reset != 0 clears the counters.  The model-level parity tests use it to prove that
 * the production kernels, not a fallback, produced what they compare with the reference's outputs.
 */
int d3d_debug_dispatch_counts(unsigned long long* out4, int reset);

/*
 * Compile-time experiment knobs this library was built with that differ from the production defaults, as a space-separated
 * list ("" for the production build: tests/test_abi.py asserts that).  Several -D switches of the kernels change block
 * shapes or select measured-and-dropped variants; one of them set by accident in the Makefile must not ship silently.
 */
const char* d3d_build_flags(void);

/*
 * The 16-bit operand format of this build's fast mode ("h16": every entry point named *_h16 below, BASELINE config 3): "f16"
 * (IEEE half, 11 significand bits; the default) or "bf16" (a -DD3D_H16_BF16 build).  One format per library: "h16" volumes,
 * packed weight fragments and stored activations are in it, and a caller creates them accordingly (torch.float16 /
 * torch.bfloat16).  Accumulation is fp32 either way.  Half is the default because bfloat16's 8 significand bits put the
 * regressed depth 0.9 - 1.4 stage-3 intervals from the reference's on the arg-max-sensitive model fixtures against a bar of
 * 0.25 (north_star: 1e-3 relative L1), with every rounding site of a regulariser contributing; half at the same bytes and
 * matrix-core rate stays below 0.25 (profiles/r05_h16_ablation.txt).  The variance volume, the one operand whose
 * magnitude the data decides, saturates at 65504 in the half format.  The *_bf16x3 entry points (fp32 mode: an fp32 operand as
 * the exact sum of three bfloat16 pieces) are bfloat16 in every build.
 * Replaces nothing in the reference (fp32 throughout).  ABI 9.
 */
const char* d3d_h16_format(void);

/*
 * Scratch bytes the plane-sweep entry points below (d3d_homo_warp, d3d_variance_volume[_f16],
 * d3d_weighted_corr, d3d_pair_corr_mean) can use for a problem of n_views views (reference included) of
 * [C,h,w] elements of elem_bytes (4 = fp32, 2 = fp16) swept over D planes: room for a channel-last staging
 * copy of the source maps.  0 = the shape takes none.  The buffer is the caller's (device memory, 16-byte
 * aligned, private to the call until it completes on its stream); passing NULL / fewer bytes is valid and
 * selects a slower staging form (fp32) or the direct-gather kernel (fp16).  Sweeps of at most 48 planes (the window kernel) use it
 * only when the hypotheses are a [D,h,w] VOLUME (D3D_DEPTH_PER_PIXEL): patches whose planes no window bounds then gather their
 * taps from the channel-last copy instead of the planar maps (round 4, ABI 8); (lo, step) maps and per-plane depths leave it unused.
 * Replaces nothing in the reference: torch's caching allocator plays this role there.
 */
size_t d3d_sweep_workspace_bytes(int n_views, int C, int D, int h, int w, int elem_bytes);
/* The same for a call whose depth mode is known (ABI 9): (lo, step) maps and per-plane depths never use the window kernel's
 * channel-last copy (650 MB at the last cascade stage), so their figure is the ring kernel's alone. */
size_t d3d_sweep_workspace_bytes_for(int n_views, int C, int D, int h, int w, int elem_bytes, int depth_mode);

/*
 * module.py:516-557 homo_warping_float -- warp ONE source feature map onto D planes.
 * src [C,h,w] -> out [C,D,h,w].
 */
int d3d_homo_warp(const float* src, const float* proj34, const float* depth, int depth_mode, int C, int D, int h,
                  int w, float* out, void* workspace, size_t workspace_bytes, d3d_stream_t stream);

/*
 * module.py:560-601 homo_warping_double -- the warp with an fp64 coordinate chain: proj = src_proj @ inverse(ref_proj),
 * rot @ [x,y,1], * depth, + trans, the divide and the [-1,1] normalisation in double; the grid is rounded to fp32 and
 * un-normalised / sampled in fp32 as F.grid_sample does.  (The reference function only runs when handed fp64
 * projection matrices; no model of the reference calls it.)
 * d3d_compose_projections_f64: proj44 device double [V,4,4] -> out34 device double [V-1,12].
 * d3d_homo_warp_f64coord: src [C,h,w] fp32, proj34 device double [12], depth fp32 -> out [C,D,h,w] fp32.
 */
int d3d_compose_projections_f64(const double* proj44, int n_views, double* out34, d3d_stream_t stream);
int d3d_homo_warp_f64coord(const float* src, const double* proj34, const float* depth, int depth_mode, int C, int D,
                           int h, int w, float* out, d3d_stream_t stream);

/*
 * cas_mvsnet.py:45-60 (same arithmetic ucsnet.py:119-134, msrednet.py:217-230 and,
 * with D = 1 per call, msrednet.py:400-414) -- fused warp + variance cost volume:
 *     sum = ref + SUM_i warp_i ; sq = ref^2 + SUM_i warp_i^2 ; var = sq/V - (sum/V)^2
 * feats: HOST array of n_views device pointers, feats[0] = reference [C,h,w].
 * proj34: device [V-1,12].  out: device [C,D,h,w].
 * Never materialises the warped volumes, sum or sq.
 */
int d3d_variance_volume(const float* const* feats, const float* proj34, const float* depth, int depth_mode,
                        int n_views, int C, int D, int h, int w, float* out, void* workspace, size_t workspace_bytes,
                        d3d_stream_t stream);
/* d3d_variance_volume with plane d as one contiguous block: out [D,C,h,w] (round 4, ABI 8) -- the layout the slice-recurrent
 * regularisers read (msrednet.py:400-437), as d3d_weighted_corr(plane_major = 1) is for adamvs.py:492-512.  Same values. */
int d3d_variance_volume_planes(const float* const* feats, const float* proj34, const float* depth, int depth_mode,
                               int n_views, int C, int D, int h, int w, float* out, void* workspace, size_t workspace_bytes,
                               d3d_stream_t stream);

/* d3d_variance_volume with the result as a channel-last bf16 volume [D,h,w,C] (RNE at the store; fp32 features and fp32
 * arithmetic as above): the form conv0 of the 3-D regulariser takes in bf16 mode (d3d_conv3d_k3_cl_h16, in_cl = 1), which
 * rounds its input to bf16 anyway -- so the regularised result is bit-identical to the planar fp32 route, the volume is
 * written once at half the bytes and read with 16-byte loads.  C % 8 == 0, h*w*C < 2^31, at most 6 source views;
 * D3D_ERR_UNSUPPORTED otherwise (nothing is launched; callers use d3d_variance_volume + d3d_volume_planar_to_cl_h16). */
int d3d_variance_volume_cl_h16(const float* const* feats, const float* proj34, const float* depth, int depth_mode,
                                int n_views, int C, int D, int h, int w, void* out, void* workspace, size_t workspace_bytes,
                                d3d_stream_t stream);
/* The same volume in planes of 8-channel groups, "CL8": out [D, C/8, h, w, 8] bf16 -- the 16 bytes a lane stores per voxel
 * and group are then a WHOLE cell (with C > 8 the [D,h,w,C] form above makes every store a partial 32-byte write: the
 * write traffic of a C = 16 volume was that of the planar fp32 one).  d3d_conv3d_k3_cl_h16 / d3d_conv3d_k3_c1_cl_h16 take
 * it with in_cl = 2.  For C = 8 the two layouts coincide. */
int d3d_variance_volume_cl8_h16(const float* const* feats, const float* proj34, const float* depth, int depth_mode,
                                int n_views, int C, int D, int h, int w, void* out, void* workspace, size_t workspace_bytes,
                                d3d_stream_t stream);

/* The same volume with fp16 STORAGE (BASELINE config 5): feats[i] and out are IEEE half tensors of the shapes
 * above; projections, depth and every product / sum stay fp32, the result is rounded once (RNE).  With a
 * workspace of d3d_sweep_workspace_bytes(..., 2) bytes and C % 16 == 0, up to 6 source views run on the LDS-ring
 * kernel with fp16 ring cells; otherwise on the direct-gather kernel. */
int d3d_variance_volume_f16(const void* const* feats, const float* proj34, const float* depth, int depth_mode,
                            int n_views, int C, int D, int h, int w, void* out, void* workspace,
                            size_t workspace_bytes, d3d_stream_t stream);

/*
 * adamvs.py:469-474 -- per-pair channel-mean correlation for the visibility net:
 *     out[d] = mean_c( ref[c] * warp_d(src)[c] )          ref, src [C,h,w] -> out [D,h,w]
 */
int d3d_pair_corr_mean(const float* ref, const float* src, const float* proj34, const float* depth, int depth_mode,
                       int C, int D, int h, int w, float* out, void* workspace, size_t workspace_bytes,
                       d3d_stream_t stream);

/*
 * adamvs.py:492-509 -- visibility-weighted correlation:
 *     sim[c,d] = SUM_i (warp_i[c,d] * ref[c]) * vw_i / (1e-5 + SUM_i vw_i)
 * weights: device [V-1,h,w] at this stage's resolution.  out [C,D,h,w] (plane_major = 0, the reference's training-time
 * layout adamvs.py:292-301) or [D,C,h,w] (plane_major = 1: the slice loop of adamvs.py:492-512 then reads plane d as one
 * contiguous [C,h,w] block -- the reference never materialises the volume at inference).
 */
int d3d_weighted_corr(const float* const* feats, const float* proj34, const float* weights, const float* depth,
                      int depth_mode, int n_views, int C, int D, int h, int w, int plane_major, float* out, void* workspace,
                      size_t workspace_bytes, d3d_stream_t stream);
/* adamvs.py:492-509 with the volume leaving as 16-bit cells in planes of 8-channel groups, out [D, C/8, h, w, 8] in the library's h16
 * format (RNE of the fp32 value d3d_weighted_corr stores; saturating at +-65504 in the half format): what the fused conv-GRU cell
 * stages with 16-byte loads (d3d_gru_cell_fused_cl8_h16).  Window kernel only -- C % 8 == 0, at most 4 source views, D <= 48: every
 * stage of the cascades -- D3D_ERR_UNSUPPORTED otherwise (the caller then takes d3d_weighted_corr).  ABI 9. */
int d3d_weighted_corr_cl8_h16(const float* const* feats, const float* proj34, const float* weights, const float* depth,
                              int depth_mode, int n_views, int C, int D, int h, int w, void* out, void* workspace,
                              size_t workspace_bytes, d3d_stream_t stream);


/*
 * cas_mvsnet.py:69-76 + module.py:605-613 -- softmax over D, soft-argmin depth and the
 * 4-plane-window confidence around trunc(SUM_k p_k k):
 * cost [D,h,w], depth ([D] or [D,h,w]) -> depth_out [h,w], conf_out [h,w].
 */
int d3d_softargmin_conf4(const float* cost, const float* depth, int depth_mode, int D, int h, int w,
                         float* depth_out, float* conf_out, d3d_stream_t stream);

/*
 * adamvs.py:514-525 (same msrednet.py:418-429) -- one plane of the online regression:
 *     p = exp(reg); max_p = max(max_p, p); sum_d += d*p; sum_p += p
 * reg [H,W].  dplane [hd,wd]: the per-pixel depth of this plane; when (hd,wd) != (H,W)
 * it is resampled bilinearly with align_corners=False (adamvs.py:519-520, the x2 case).
 * max_p, sum_d, sum_p [H,W] are updated in place (zero them before the first plane).
 */
int d3d_online_regress_update(const float* reg, const float* dplane, int hd, int wd, int H, int W, float* max_p,
                              float* sum_d, float* sum_p, d3d_stream_t stream);

/* adamvs.py:527-529 -- depth = sum_d/(sum_p+1e-10); conf = max_p/(sum_p+1e-10). */
/* The head of a slice regulariser fused with d3d_online_regress_update (bf16 mode): reg = ConvTranspose2d(8, 1, k 3, s 2, p 1,
 * output_pad 1)(up) + bias (transposed != 0: adamvs.py:417, stages 1-2; accumulators [2h,2w]) or Conv2d(8, 1, 3, pad 1)(up) + bias
 * (stage 3; accumulators [h,w]) with operands rounded to bf16 as the matrix cores round them, then the update of adamvs.py:514-525
 * with dplane [hd,wd] resampled as d3d_online_regress_update resamples it.  up [8,h,w], weight 72 floats [c][k_y][k_x] (the
 * nn.ConvTranspose2d [8,1,3,3] / nn.Conv2d [1,8,3,3] tensor, ALREADY rounded to bf16 values), bias [1]; w % 2 == 0 (transposed) /
 * w % 4 == 0, else D3D_ERR_UNSUPPORTED.  `reg` never reaches memory. */
int d3d_slice_head_regress_h16(const float* up, const float* weight, const float* bias, int transposed, const float* dplane, int hd,
                                int wd, int h, int w, float* max_p, float* sum_d, float* sum_p, d3d_stream_t stream);
/* A 3 x 3 convolution of ConvGRUCell2 (module.py:71-99: gate_conv / output_conv over cat(x, h), bias, no activation) that also
 * accumulates the GroupNorm(1, C) statistics of its output (round 4, ABI 8; csrc/gn_stats.h): gn_stats [ngroups][2] fp64 =
 * (sum, sum of squares) per channel group, ZEROED by the caller before the launch (stream order); channels >= gn_split are the second
 * group (the update half of the gate convolution), gn_split = Co means one group.  The sums are those d3d_groupnorm_stats computes
 * from the stored tensor (same operands, fp64; the order of the additions differs) -- that launch and its pass over the tensor go.
 * _zs: the tile kernel of d3d_conv2d_k3_zs_h16 for C1 + C2 = 16 | 24 | 32 | 40 (24 | 40: Co <= 16; else Co <= 32), W % 4 == 0;
 * _wide: d3d_conv2d_k3_wide_h16's shapes.  D3D_ERR_UNSUPPORTED otherwise (nothing launched). */
int d3d_conv2d_k3_zs_h16_gn(const float* in, int C1, const float* in2, int C2, const void* wpacked, const float* shift, int Co, int H,
                             int W, float* out, double* gn_stats, int gn_split, d3d_stream_t stream);
int d3d_conv2d_k3_wide_h16_gn(const float* in, int C1, const float* in2, int C2, const void* wpacked, const float* shift, int Co, int H,
                               int W, float* out, double* gn_stats, int gn_split, d3d_stream_t stream);
/* module.py:287-294 at 64 input channels (msrednet.py:348 upconv3, ABI 10): out [Co,2H,2W] = act(convT3x3_s2(in) * scale + shift)
 * (+ skip, added last) as the stride-1 convolution of the zero-stuffed input with the flipped, transposed kernel (wpacked =
 * ops._pack_z2_bf16 of it); the stuffed image exists only in the kernel's staging.  in [64,H,W]; Co = 32 | 64; 16-bit operands. */
int d3d_convtranspose2d_k3s2_wide_h16(const float* in, const void* wpacked, const float* scale, const float* shift,
                                      const float* skip, int act, int Ci, int Co, int H, int W, float* out, d3d_stream_t stream);
/* conv0 of a feature trunk in ONE launch (round 4, ABI 7; csrc/conv2d_zs.hip, IMG3 form): out = act(scale * Conv3x3_8->Co(c) + shift)
 * with c = act0(scale0 * Conv3x3_3->8(img) + shift0) evaluated per tile from the staged image patch and never written (module.py:
 * 663-666: ConvBnReLU(3, 8) + ConvBnReLU(8, 8) at full resolution).  img [3,H,W]; w0packed [4][3][3][8] fp32 as
 * d3d_conv2d_k3_stream takes it (input channel 3 = zeros); wpacked = the split B operands of d3d_conv2d_k3_zs_bf16x3.
 * Bit-identical to d3d_conv2d_k3_stream followed by d3d_conv2d_k3_zs_bf16x3.  Co <= 16, W % 4 == 0, out 16-byte aligned;
 * D3D_ERR_UNSUPPORTED otherwise (nothing launched). */
int d3d_conv2d_k3_pair3_bf16x3(const float* img, const float* w0packed, const float* scale0, const float* shift0, int act0,
                               const void* wpacked, const float* scale, const float* shift, int act, int Co, int H, int W,
                               float* out, d3d_stream_t stream);
/* Tail of a depth slice at the stages whose head up-samples (adamvs.py:413-418, 423-425, 514-525) in ONE kernel (round 4, ABI 6;
 * csrc/regress.hip slice_tail_kernel): up = relu(ConvTranspose2d_16->8(state2) + bup + state1) stays in LDS,
 * reg = ConvTranspose2d_8->1(up) + bhead, and the online regression update of (max_p, sum_d, sum_p) [4h, 4w] at `dplane`.
 * state2 [16,h,w], state1 [8,2h,2w]; wup_packed = ops._pack_t2d_bf16, whead = the 72 head weights rounded to bf16 (fp32 values).
 * Bit-identical to d3d_convtranspose2d_k3s2_zs_h16 followed by d3d_slice_head_regress_h16(transposed = 1).  w % 4 == 0,
 * 16-byte aligned maps; D3D_ERR_UNSUPPORTED otherwise (nothing launched). */
int d3d_slice_tail_regress_h16(const float* state2, const void* wup_packed, const float* bup, const float* state1, const float* whead,
                                const float* bhead, const float* dplane, int hd, int wd, int h, int w, float* max_p, float* sum_d,
                                float* sum_p, d3d_stream_t stream);
/* The same tail with the head at `up`'s own resolution (ABI 10; adamvs.py:413-418 at the last stage, msrednet.py:361-363 + 418-437):
 * up = relu(ConvTranspose2d_16->8(state2) + bup + state1), or with skip_after_act relu(ConvTranspose2d_16->8(state2) + bup) + state1
 * (module.py:287-294 ConvTransReLU followed by the skip); reg = Conv2d(8, 1, 3, pad 1)(up) + bhead; the online regression update of
 * (max_p, sum_d, sum_p) [2h, 2w] at `dplane`.  `up` and `reg` stay in LDS.  bup may be null; whead = the [1,8,3,3] weights [c][k_y][k_x]
 * rounded to the 16-bit format (fp32 values).  Bit-identical to d3d_convtranspose2d_k3s2_zs_h16 followed by
 * d3d_slice_head_regress_h16(transposed = 0).  w % 4 == 0, 16-byte aligned maps; D3D_ERR_UNSUPPORTED otherwise (nothing launched). */
int d3d_slice_tail_regress_same_h16(const float* state2, const void* wup_packed, const float* bup, const float* state1, int skip_after_act,
                                    const float* whead, const float* bhead, const float* dplane, int hd, int wd, int h, int w,
                                    float* max_p, float* sum_d, float* sum_p, d3d_stream_t stream);
int d3d_online_regress_finalize(const float* max_p, const float* sum_d, const float* sum_p, int64_t n,
                                float* depth_out, float* conf_out, d3d_stream_t stream);

/*
 * module.py:616-650 get_depth_range_samples.
 * mode D3D_DEPTH_PER_PLANE: cur_depth is [2] = (min,max) -> out [D] = linspace.
 * mode D3D_DEPTH_PER_PIXEL: cur_depth is [h,w] -> out [D,h,w],
 *     lo = cur - D/2*interval, hi = cur + D/2*interval, out[k] = lo + k*(hi-lo)/(D-1).
 * mode D3D_DEPTH_AFFINE: cur_depth is [h,w] -> out [2,h,w] = (lo, (hi-lo)/(D-1)): the two maps the
 *     D planes of the per-pixel mode are generated from (the sweep, soft-argmin and resize entry
 *     points take them in place of the volume: cas_mvsnet.py:224-226 resamples the volume
 *     bilinearly plane by plane, which commutes with the affine form).
 */
int d3d_depth_range_samples(const float* cur_depth, int mode, int D, float interval, int h, int w, float* out,
                            d3d_stream_t stream);

/*
 * F.interpolate(mode='bilinear', align_corners=False) for a stack of n maps, as used for
 * the view weights (adamvs.py:502) and the inter-stage depth hand-off
 * (cas_mvsnet.py:211-213).  in [n,h,w] -> out [n,H,W].
 */
int d3d_resize_bilinear(const float* in, int n, int h, int w, int H, int W, float* out, d3d_stream_t stream);

/*
 * cas_mvsnet.py:224-226 -- trilinear (align_corners=False) resample of the full
 * resolution hypothesis volume [D,H,W] to the stage grid [D,h,w]; the depth axis keeps
 * its size, so this is a per-plane bilinear resize.
 * Provided as an alias of d3d_resize_bilinear with n = D.
 */

/*
 * module.py:297-304 ConvBnReLU3D / cas_mvsnet.py:84-110 -- 3x3x3 convolution, pad 1,
 * stride 1 or 2, with eval-mode BatchNorm folded to a per-channel affine
 * (scale, shift; NULL = identity), optional ReLU and optional skip tensor added AFTER
 * the activation (cas_mvsnet.py:116-118).
 * in [Ci,D,H,W]; weight [Co,Ci,3,3,3] (nn.Conv3d layout); out [Co,Do,Ho,Wo].
 */
int d3d_conv3d_k3(const float* in, const float* weight, const float* scale, const float* shift, const float* skip,
                  int relu, int Ci, int Co, int D, int H, int W, int stride, float* out, d3d_stream_t stream);

/* The same convolution for C_out = 8, stride 1 (conv0 of every CostRegNet, cas_mvsnet.py:84) on the fp32 vector units,
 * streaming the input volume through LDS plane by plane.  wpacked: the nn.Conv3d weight [8,Ci,3,3,3] re-laid out as
 * [Ci][ky][kx][kz][8] (host-side permutation, ops.conv3d_k3).  Ci % 8 == 0; returns D3D_ERR_UNSUPPORTED when
 * 7*D*H*W*4 bytes exceeds the 32-bit offsets of its staging loads. */
int d3d_conv3d_k3_co8(const float* in, const float* wpacked, const float* scale, const float* shift, const float* skip,
                      int relu, int Ci, int D, int H, int W, float* out, d3d_stream_t stream);

/* The same layer (3x3x3, stride 1, pad 1, C_out = 8; C_in = 8 | 16 | 32; W % 4 == 0) with bf16 OPERANDS on the matrix
 * cores (v_mfma_f32_16x16x32_bf16, fp32 accumulation; BASELINE config 3): tensors stay fp32 in memory, the input is
 * rounded to bf16 (RNE) while it is staged, each input plane is read once and feeds three output planes.
 * wpacked: the weight [8,Ci,3,3,3] rounded to bf16 and laid out in the instruction's B-operand order,
 * [k_z][K block of 32][lane 0..63][8 values], K = (k_y, k_x, c_in) (ops.conv3d_k3 packs it once per parameter version). */
int d3d_conv3d_k3_c8_h16(const float* in, const void* wpacked, const float* scale, const float* shift, const float* skip,
                          int relu, int Ci, int D, int H, int W, float* out, d3d_stream_t stream);
/* The same kernel for C_out <= 16 (C_in = 8 | 16 | 32) and 32 -> 32: conv2 / conv4 of CostRegNet (cas_mvsnet.py:87,90).
 * wpacked: [k_z][K block][N tile of 16 channels][lane][8 values]. */
int d3d_conv3d_k3_zs_h16(const float* in, const void* wpacked, const float* scale, const float* shift, const float* skip,
                          int relu, int Ci, int Co, int D, int H, int W, float* out, d3d_stream_t stream);
/* The same kernel with fp32 ACCURACY (the default precision of the regularisers): both operands as exact three-way bf16
 * splits (hi + mid + lo), six v_mfma_f32_16x16x32_bf16 products per K block accumulated in fp32 -- the 3-D form of
 * d3d_conv2d_k3_zs_bf16x3.  C_in = 8 | 16 | 32 with C_out <= 16; round 4: 32 -> 32 and 64 -> 64 (conv4 / conv6, fragments from
 * L2); W % 4 == 0.  wpacked: [hi | mid | lo] x the layout above
 * (ops._pack_c8_bf16x3). */
int d3d_conv3d_k3_zs_bf16x3(const float* in, const void* wpacked, const float* scale, const float* shift, const float* skip,
                            int relu, int Ci, int Co, int D, int H, int W, float* out, d3d_stream_t stream);
/* C_out = 1 (the probability layer) the same way: k_z folded into the columns of one tile (d3d_conv3d_k3_c1_cl_h16's form),
 * planar fp32 in [8,D,H,W] and out [D,H,W]; wpacked: [hi | mid | lo] x ops._pack_c8_kzfold_bf16. */
int d3d_conv3d_k3_c1_bf16x3(const float* in, const void* wpacked, const float* scale, const float* shift, const float* skip,
                            int relu, int Ci, int D, int H, int W, float* out, d3d_stream_t stream);


/* module.py:307-314 Deconv3d (+BN+ReLU) / cas_mvsnet.py:103,118 for C_out = 8 (conv11 of CostRegNet: 16 -> 8, then the
 * skip add) on the fp32 vector units, streaming the INPUT volume through LDS: k = 3, stride 2, pad 1, output_pad 1;
 * in [Ci,D,H,W] -> out [8,2D,2H,2W]; skip (same shape as out, may be NULL) added after the activation.
 * wpacked: the nn.ConvTranspose3d weight [Ci,8,3,3,3] re-laid out as [Ci][kz][ky][kx][8].  Ci % 8 == 0. */
int d3d_convtranspose3d_k3s2_co8(const float* in, const float* wpacked, const float* scale, const float* shift,
                                 const float* skip, int relu, int Ci, int D, int H, int W, float* out,
                                 d3d_stream_t stream);

/* module.py:307-314 Deconv3d (+BN+ReLU) / cas_mvsnet.py:97-103,116-118 with bf16 OPERANDS on the matrix cores
 * (v_mfma_f32_16x16x32_bf16, fp32 accumulation; BASELINE config 3): k = 3, stride 2, pad 1, output_pad 1;
 * in [Ci,D,H,W] fp32 -> out [Co,2D,2H,2W] fp32; skip (shape of out, may be NULL) added after the activation.  Taken channel
 * pairs: 16->8, 16->16, 32->16, 64->32 (D3D_ERR_UNSUPPORTED otherwise).  The layer runs as eight dense per-parity
 * convolutions over one staged input (no multiplications by inserted zeros), each input plane read once per tile.
 * wpacked: the weight [Ci,Co,3,3,3] rounded to bf16, per output parity class in B-operand lane order
 * (ops.convtranspose3d_k3s2 packs it once per parameter version). */
int d3d_convtranspose3d_k3s2_zs_h16(const float* in, const void* wpacked, const float* scale, const float* shift,
                                     const float* skip, int relu, int Ci, int Co, int D, int H, int W, float* out,
                                     d3d_stream_t stream);
/* ... and with fp32 accuracy from three-way bf16 splits of both operands (16 -> 8 | 16: conv11 of every CostRegNet in the
 * default precision; round 4: 32 -> 16 and 64 -> 32, conv9 / conv7, the latter with its fragments read from L2); wpacked: [hi | mid | lo] x the layout above (ops._pack_t2_bf16x3). */
int d3d_convtranspose3d_k3s2_zs_bf16x3(const float* in, const void* wpacked, const float* scale, const float* shift,
                                     const float* skip, int relu, int Ci, int Co, int D, int H, int W, float* out,
                                     d3d_stream_t stream);
/* Stride-2 3x3x3 convolution in fp32 accuracy on split bf16 operands (conv1 / conv3 / conv5 of a CostRegNet in fp32 mode,
 * cas_mvsnet.py:86,89,92; csrc/conv_s2x3.hip): planar fp32 in [Ci,D,H,W] -> out [Co,(D-1)/2+1,(H-1)/2+1,(W-1)/2+1];
 * wpacked = ops._pack_c8_bf16x3.  8 -> 16, 16 -> 32, 32 -> 64 with an output width that is a multiple of 4;
 * D3D_ERR_UNSUPPORTED otherwise (nothing launched). */
int d3d_conv3d_k3s2_zs_bf16x3(const float* in, const void* wpacked, const float* scale, const float* shift, const float* skip,
                              int relu, int Ci, int Co, int D, int H, int W, float* out, d3d_stream_t stream);

/* module.py:5-51 ConvGRUCell / adamvs.py:409-413 ConvReLU of the slice regularisers, bf16 mode: 3x3 stride-1 2-D convolution
 * over the channel concat of `in` [C1,H,W] and `in2` [C2,H,W] (may be NULL, C2 = 0) on v_mfma_f32_16x16x32_bf16, one 64 x 8
 * (32 x 8) tile of the image per step, planar fp32 tensors in HBM.  out [Co,H,W] = epilogue(conv * scale + shift):
 *   act 0 none | 1 ReLU, with `skip` [Co,H,W] (may be NULL) added before the activation or, skip_after_act, after it;
 *   act 2 GRU gates: sigmoid, channels < ep_split multiplied by h = skip [ep_split,H,W]      (-> [r*h | u]);
 *   act 3 GRU update: u*h + (1-u)*tanh(.), h = skip [Co,H,W], u = aux1 [Co,H,W].
 * C1 + C2 = 8 | 16 | 32 in groups of 8 with Co <= 32, or 48 with Co <= 48 (adamvs.py:198-238, the pair-visibility UNet);
 * W % 4 == 0; D3D_ERR_UNSUPPORTED otherwise (nothing launched).
 * wpacked: the weight [Co,C1+C2,3,3] rounded to bf16 in B-operand lane order, [K block][N tile][lane][8], K = (k_y,k_x,c_in)
 * (ops._pack_z2_bf16). */
int d3d_conv2d_k3_zs_h16(const float* in, int C1, const float* in2, int C2, const void* wpacked, const float* scale,
                          const float* shift, const float* skip, const float* aux1, int act, int ep_split,
                          int skip_after_act, int Co, int H, int W, float* out, d3d_stream_t stream);

/* The same layer in the models' default precision: exact fp32 operands on v_mfma_f32_16x16x4_f32 (weights fp32 in the same
 * [K block of 4][N tile][lane] order, ops._pack_z2_f32); arguments and shapes as d3d_conv2d_k3_zs_h16. */
int d3d_conv2d_k3_zs_f32(const float* in, int C1, const float* in2, int C2, const void* wpacked, const float* scale,
                         const float* shift, const float* skip, const float* aux1, int act, int ep_split,
                         int skip_after_act, int Co, int H, int W, float* out, d3d_stream_t stream);

/* The same layer with fp32 accuracy on the bf16 matrix cores: every fp32 operand is the exact sum of three bf16 numbers
 * (hi + mid + lo); the activations are split while a tile is staged, the weights on the host (wpacked: the three
 * d3d_conv2d_k3_zs_h16 packings of hi | mid | lo one after the other, ops._pack_z2_bf16x3), and a K block takes the six
 * products down to 2^-16 of the leading one on v_mfma_f32_16x16x32_bf16, accumulated in fp32 -- what is dropped is below
 * fp32's own rounding of a product.  Arguments and shapes as d3d_conv2d_k3_zs_f32 (C1 + C2 = 8 | 16 | 32, Co <= 32). */
int d3d_conv2d_k3_zs_bf16x3(const float* in, int C1, const float* in2, int C2, const void* wpacked, const float* scale,
                            const float* shift, const float* skip, const float* aux1, int act, int ep_split,
                            int skip_after_act, int Co, int H, int W, float* out, d3d_stream_t stream);

/* Pooled-context heads of the AdaMVS feature pyramid (adamvs.py:75-101, 116-151 of the reference:
 * out = head(cat(up(branch_4(f)), up(branch_8(f)), f)), up = bilinear resize with align_corners=False, head a 1x1 convolution
 * without bias).  d3d_avgpool2d_4_8: both AvgPool2d((4,4),4) and AvgPool2d((8,8),8) of in [C,H,W] in one read ->
 * out4 [C,H/4,W/4], out8 [C,H/8,W/8] (floor sizes; W % 4 == 0, else D3D_ERR_UNSUPPORTED).
 * d3d_conv1x1_context: out [Co,H,W] = weight [Co,Ci] . f [Ci,H,W] + resize(a [Co,Ha,Wa]) + resize(b [Co,Hb,Wb]) -- the head
 * applied to the branch outputs at THEIR resolution (a = W_a . branch_4 output, b = W_b . branch_8 output: a 1x1
 * convolution commutes with the resize), so neither the upsampled branches nor the concat reach HBM.
 * Ci == Co in 8 | 16 | 32, W % 4 == 0, 3 Wa <= W, 3 Wb <= W; D3D_ERR_UNSUPPORTED otherwise (nothing launched). */
int d3d_avgpool2d_4_8(const float* in, int C, int H, int W, float* out4, float* out8, d3d_stream_t stream);
/* module.py:677-679, 701-703 (the 1 x 1 output layers of the feature pyramids) as a streaming kernel in exact fp32 (ABI 10):
 * out [Co,H,W] = act(scale * Conv1x1(in) + shift) (+ skip, added last).  in [Ci,H,W], Ci = 8 | 16 | 32; wt = the nn.Conv2d weight
 * re-laid as [ceil(Co / 8)][Ci][8] (blocks of 8 output channels, zero-padded: ops._pack_k1); act 0 | 1 (ReLU); scale / shift / skip
 * may be null.  H * W a multiple of 4 and 16-byte aligned tensors, else D3D_ERR_UNSUPPORTED (nothing launched). */
int d3d_conv2d_k1_f32(const float* in, const float* wt, const float* scale, const float* shift, const float* skip, int act,
                      int Ci, int Co, int H, int W, float* out, d3d_stream_t stream);

/* out [Co,H,W] -= sum over the 3x3 taps whose source pixel lies OUTSIDE the image of taps[Co,3,3]: the border correction of a
 * bias that was added to the input of a zero-padded 3x3 convolution and folded into the layer's constant (module.fpn_output:
 * the lateral bias of the last FPN level, module.py:745-747 of the reference).  In place, border pixels only. */
int d3d_conv3x3_bias_border(float* out, const float* taps, int Co, int H, int W, d3d_stream_t stream);
int d3d_conv1x1_context(const float* f, int Ci, const float* weight, const float* a, int Ha, int Wa, const float* b, int Hb,
                        int Wb, int Co, int H, int W, float* out, d3d_stream_t stream);

/* The stride-2 and the transposed (k 3, stride 2, pad 1, output_pad 1) 2-D layers of the slice regularisers on the same tile
 * scheme (adamvs.py:411 ConvReLU(8,16,3,2,1); :413-417 upconv1 16->8 with the skip before the ReLU, upconv2d 8->1): planar fp32
 * in [Ci,H,W] -> out [Co,(H-1)/2+1,(W-1)/2+1] resp. [Co,2H,2W]; act 0 | 1 (ReLU); skip (shape of out, may be NULL) added before
 * the activation or, skip_after_act, after it.  Stride 2: C_in = 8 | 16, C_out <= 32, output width % 4 == 0; transposed:
 * C_in = 8 | 16 | 32, C_out <= 16, W % 4 == 0.  wpacked: ops._pack_z2_bf16 / ops._pack_t2d_bf16 (per output parity class, as the 3-D form). */
int d3d_conv2d_k3s2_zs_h16(const float* in, const void* wpacked, const float* scale, const float* shift, const float* skip,
                            int act, int skip_after_act, int Ci, int Co, int H, int W, float* out, d3d_stream_t stream);
/* The same layer over `nbatch` images in ONE launch (ABI 10; msrednet.py:352-356: the encoder's stride-2 ConvReLUs depend on the cost
 * slices only, so the three of them run for every depth slice of a stage before the recurrent loop): in [nbatch][Ci,H,W],
 * out [nbatch][Co,Ho,Wo] with the given element strides between items; no skip.  Per item bit for bit d3d_conv2d_k3s2_zs_h16. */
int d3d_conv2d_k3s2_zs_h16_batched(const float* in, const void* wpacked, const float* scale, const float* shift, int act, int Ci, int Co,
                                   int H, int W, int nbatch, int64_t in_bstride, int64_t out_bstride, float* out, d3d_stream_t stream);
int d3d_convtranspose2d_k3s2_zs_h16(const float* in, const void* wpacked, const float* scale, const float* shift,
                                     const float* skip, int act, int skip_after_act, int Ci, int Co, int H, int W, float* out,
                                     d3d_stream_t stream);
/* ... and in exact fp32 (v_mfma_f32_16x16x4_f32; weights ops._pack_z2_f32 / ops._pack_t2d_f32): stride 2 takes C_in = 8 only
 * (two 65 x 17 patches of fp32 cells must fit the LDS), transposed C_in = 8 | 16 | 32. */
int d3d_conv2d_k3s2_zs_f32(const float* in, const void* wpacked, const float* scale, const float* shift, const float* skip,
                           int act, int skip_after_act, int Ci, int Co, int H, int W, float* out, d3d_stream_t stream);
int d3d_convtranspose2d_k3s2_zs_f32(const float* in, const void* wpacked, const float* scale, const float* shift,
                                    const float* skip, int act, int skip_after_act, int Ci, int Co, int H, int W, float* out,
                                    d3d_stream_t stream);
/* The two layers above with fp32 accuracy from three-way bf16 splits of both operands (see d3d_conv2d_k3_zs_bf16x3; wpacked:
 * the three bf16 packings of hi | mid | lo one after the other, ops._pack_z2_bf16x3 / ops._pack_t2d_bf16x3); shapes as the
 * *_zs_bf16 forms (stride 2: Ci 8 | 16; transposed: Ci 8 | 16 | 32, Co <= 16). */
int d3d_conv2d_k3s2_zs_bf16x3(const float* in, const void* wpacked, const float* scale, const float* shift, const float* skip,
                              int act, int skip_after_act, int Ci, int Co, int H, int W, float* out, d3d_stream_t stream);
int d3d_convtranspose2d_k3s2_zs_bf16x3(const float* in, const void* wpacked, const float* scale, const float* shift,
                                       const float* skip, int act, int skip_after_act, int Ci, int Co, int H, int W,
                                       float* out, d3d_stream_t stream);

/* ConvTranspose2d(kernel 4, stride 2, padding 1): in [Ci,H,W] -> out [Co,2H,2W], on the transposed tile kernel with split
 * operands (every output parity class has 2 x 2 taps; the patch carries a halo on both sides).  conv3x3(nearest_x2(f)) -- the
 * input side of the last FPN level, module.py:745-747 of the reference -- is this layer with summed weights
 * (ops.upsampled_conv_weight), so the upsampled tensor is never formed.  wpacked: ops._pack_t2d_k4_bf16x3 (for Co <= 8 both
 * column parities of an output row share one 16-column operand tile: ops._pack_t2d_k4fold_bf16); act 0 | 1, skip
 * [Co,2H,2W] or NULL as in the k = 3 form; Ci 8 | 16 | 32, Co <= 16, W % 4 == 0; D3D_ERR_UNSUPPORTED otherwise. */
int d3d_convtranspose2d_k4s2_zs_bf16x3(const float* in, const void* wpacked, const float* scale, const float* shift,
                                       const float* skip, int act, int skip_after_act, int Ci, int Co, int H, int W,
                                       float* out, d3d_stream_t stream);

/* 3x3 stride-1 convolution over cat(in, in2) with WIDE channel counts on the bf16 matrix cores (csrc/conv2d_wide.hip; the coarse
 * conv-GRU levels of the RED-Net slice regulariser, msrednet.py:337-370 with module.py:53-99): C1 + C2 = 64 | 128 in parts of 32,
 * Co = 32 | 64 | 128; out [Co,H,W] = act(conv * scale + shift) (+ skip, added last); act 0 | 1 (ReLU).  K is walked in
 * chunks of 32 input channels (the tile kernels above keep a whole layer in LDS, which ends at 48 channels).  wpacked:
 * ops._pack_z2_bf16(weight [Co,C1+C2,3,3]).  D3D_ERR_UNSUPPORTED for other shapes. */
int d3d_conv2d_k3_wide_h16(const float* in, int C1, const float* in2, int C2, const void* wpacked, const float* scale,
                            const float* shift, const float* skip, int act, int Co, int H, int W, float* out, d3d_stream_t stream);

/* One conv-GRU cell of the slice regularisers in ONE launch (csrc/gru_fused.hip; replaces adamvs.py:409-412 conv1 + conv_gru1 resp.
 * conv2 + conv_gru2, module.py:5-51 ConvGRUCell, run as three d3d_conv2d_k3*_zs_bf16 launches before):
 *     x = relu(conv3x3_stride(cost));  r, u = sigmoid(conv3x3(cat(x, h)) + bg);  c = tanh(conv3x3(cat(x, r * h)) + bc);
 *     hout = u * h + (1 - u) * c
 * cost [CP,HI,WI], h / hout [HID,H,W] planar fp32, hout != h.  stride 1: HI, WI = H, W, CP = 8 | 16 | 32, HID = 8; stride 2:
 * H, W = (HI-1)/2+1, (WI-1)/2+1, CP = 8, HID = 16.  w1 / wg / wc: ops._pack_z2_bf16 of the three weights (bf16 matrix-core
 * operands, fp32 accumulation, the state stays fp32); bg [2 HID], bc [HID].  Bit-identical to the three-launch form.
 * W, WI multiples of 4, 16-byte aligned tensors.
 * D3D_ERR_UNSUPPORTED for other channel counts. */
int d3d_gru_cell_fused_h16(const float* cost, int CP, int HI, int WI, int stride, const float* h, int HID, int H, int W,
                            const void* w1, const void* wg, const float* bg, const void* wc, const float* bc, float* hout,
                            d3d_stream_t stream);
/* The stride-1 cell on a cost plane of channel-last 16-bit cells in 8-channel groups, cost_cl8 [CP / 8, H, W, 8] in the library's h16
 * format (one plane of d3d_weighted_corr_cl8_h16's volume): a staging task is one 16-byte load and one 16-byte LDS write, no
 * conversion, half the bytes.  Bit-identical to d3d_gru_cell_fused_h16 on the planar fp32 plane of the same values (the planar entry
 * applies the same rounding while it stages).  CP = 8 | 16 | 32, HID = 8.  ABI 9.  adamvs.py:409-410, module.py:24-51. */
int d3d_gru_cell_fused_cl8_h16(const void* cost_cl8, int CP, const float* h, int HID, int H, int W, const void* w1, const void* wg,
                               const float* bg, const void* wc, const float* bc, float* hout, d3d_stream_t stream);


/* Conv2d(kernel 5, stride 2, padding 2) -- the downsampling layers of the feature trunks (module.py:669, 675; adamvs.py:64, 70 of the reference) --
 * on the stride-2 tile kernel with split operands (fp32 accuracy): in [Ci,H,W] -> out [Co,(H-1)/2+1,(W-1)/2+1]; wpacked:
 * ops._pack_z2_bf16x3 of the weight [Co,Ci,5,5] (K = (k_y,k_x,c_in)); scale / shift / skip / act as d3d_conv2d_k3s2_zs_h16.
 * Ci 8 with Co <= 16 | Ci 16 with Co <= 32, output width % 4 == 0; D3D_ERR_UNSUPPORTED otherwise. */
int d3d_conv2d_k5s2_zs_bf16x3(const float* in, const void* wpacked, const float* scale, const float* shift, const float* skip,
                              int act, int skip_after_act, int Ci, int Co, int H, int W, float* out, d3d_stream_t stream);


/* ---- channel-last bf16 activations between the layers of a CostRegNet (bf16 mode, BASELINE config 3) ------------------
 * "CL" volume: bf16 [D][H][W][C].  The matrix-core kernels round their operands to bf16 when they stage them, so a layer
 * that hands its output on in this form loses nothing its consumer would have kept, the activation traffic halves, and a
 * staging task is one 16-byte load.  Same layers and weight packings as the *_zs_bf16 entry points above
 * (cas_mvsnet.py:84-118: conv0 planar -> CL, conv1 .. conv11 CL -> CL with CL skips, prob CL -> planar).
 *
 * d3d_conv3d_k3_cl_h16: stride 1; in_cl / out_cl select the format of `in` (also 2: CL8 [D,Ci/8,H,W,8], see
 * d3d_variance_volume_cl8_h16) and of `out` + `skip` (0: planar fp32
 *   [C,D,H,W], 1: CL).  C_in = 8 | 16 | 32, C_out <= 16 or 32 -> 32; C_out % 4 == 0 for CL output, W % 4 == 0 for planar.
 * d3d_conv3d_k3s2_cl_h16: stride 2, pad 1; CL in [D,H,W,Ci] -> CL out [(D-1)/2+1, (H-1)/2+1, (W-1)/2+1, Co];
 *   8->8, 8->16, 16->16, 16->32.
 * d3d_convtranspose3d_k3s2_cl_h16: channel_last = 1: CL in / skip / out ([2D,2H,2W,Co]); 0: the planar form above;
 *   2: CL with the x-folded weight packing (16 -> 8 only: both output-column parities in one GEMM, ops._pack_t2_fold_bf16).
 * d3d_volume_planar_to_cl_h16 / d3d_volume_cl_h16_to_planar: format conversion of a volume of n voxels, C % 8 == 0 (RNE;
 *   the way back is exact) -- for the layers that stay on the planar kernels (conv5 / conv6) and for tests.
 * D3D_ERR_UNSUPPORTED for other shapes; nothing is launched then. */
int d3d_conv3d_k3_cl_h16(const void* in, int in_cl, const void* wpacked, const float* scale, const float* shift,
                          const void* skip, int relu, int Ci, int Co, int D, int H, int W, void* out, int out_cl,
                          d3d_stream_t stream);
/* C_out = 1 (the probability layer, cas_mvsnet.py:110): in planar fp32 or CL (in_cl), out planar fp32 [D,H,W]; the three k_z
 * slices of the weight are columns 0..2 of ONE operand tile (ops._pack_c8_kzfold_bf16), a third of the matrix work of the
 * generic entry point.  C_in = 8 | 16 | 32, W % 4 == 0. */
int d3d_conv3d_k3_c1_cl_h16(const void* in, int in_cl, const void* wpacked, const float* scale, const float* shift,
                             const float* skip, int relu, int Ci, int D, int H, int W, float* out, d3d_stream_t stream);
int d3d_conv3d_k3s2_cl_h16(const void* in, const void* wpacked, const float* scale, const float* shift, const void* skip,
                            int relu, int Ci, int Co, int D, int H, int W, void* out, d3d_stream_t stream);
int d3d_convtranspose3d_k3s2_cl_h16(const void* in, const void* wpacked, const float* scale, const float* shift,
                                     const void* skip, int relu, int Ci, int Co, int D, int H, int W, void* out,
                                     int channel_last, d3d_stream_t stream);
/* conv11 + prob of a CostRegNet in one kernel (cas_mvsnet.py:103-105,118-119; csrc/conv_t2p.hip):
 *   y = skip + ReLU(scale * ConvTranspose3d_16->8(in) + shift) rounded to bf16, out = Conv3d_8->1(y) + prob_bias[0];
 * y lives in LDS only.  in CL [D,H,W,16], skip CL [2D,2H,2W,8] (or null), out planar fp32 [2D,2H,2W]; wt_folded as for
 * d3d_convtranspose3d_k3s2_cl_h16(channel_last = 2), wprob_kzfolded as for d3d_conv3d_k3_c1_cl_h16.  Bit-identical to
 * those two calls in sequence.  W even; D3D_ERR_UNSUPPORTED otherwise (nothing launched). */
int d3d_convtranspose3d_prob_cl_h16(const void* in, const void* wt_folded, const float* scale, const float* shift,
                                     const void* skip, int relu, const void* wprob_kzfolded, const float* prob_bias,
                                     int D, int H, int W, float* out, d3d_stream_t stream);
int d3d_volume_planar_to_cl_h16(const float* in, int C, size_t n, void* out, d3d_stream_t stream);
int d3d_volume_cl_h16_to_planar(const void* in, int C, size_t n, float* out, d3d_stream_t stream);


/* 3x3 stride-1 nn.Conv2d with C_out = 8 | 16 (the full- / half-resolution layers of the feature pyramids,
 * module.py:653-755, and the conv-GRU cells of the slice regularisers, module.py:5-51) on the fp32 vector units.
 * Input = cat(in0 [Ci0], in1 [Ci1]) along the channels (in1 may be NULL with Ci1 = 0; with two inputs Ci0 % 8 == 0).
 * act 0 | 1: same epilogue as d3d_conv2d_k3 (affine, ReLU, skip added after the activation); act 2 | 3: the ConvGRUCell
 * epilogues of d3d_conv_fold_f32 (2: sigmoid, channels < ep_split times h = skip; 3: u*h + (1-u)*tanh(y), u = aux1).
 * wpacked: weight [Co,Ci0+Ci1,3,3] re-laid out as [C_in rounded up to 8][ky][kx][Co], zero rows for the padding. */
int d3d_conv2d_k3_stream(const float* in0, int Ci0, const float* in1, int Ci1, const float* wpacked, const float* scale,
                         const float* shift, const float* skip, const float* aux1, int ep_split, int act, int Co, int H,
                         int W, float* out, d3d_stream_t stream);

/* module.py:736-747 (FeatureNet_mvsnet, "fpn"): out = conv1x1(in) + bias + nearest-x2 upsampling of `coarse`
 * (F.interpolate(..., scale_factor=2, mode="nearest") + self.inner(x)) without materialising the upsampled tensor.
 * in [Ci,H,W]; coarse [Co,H/2,W/2]; wpacked = weight [Co,Ci,1,1] transposed to [Ci][Co]; out [Co,H,W]; H, W even;
 * (Ci, Co) in {(8,32), (16,32)}, else D3D_ERR_UNSUPPORTED. */
int d3d_conv1x1_upskip(const float* in, int Ci, const float* wpacked, const float* bias, const float* coarse, int Co, int H,
                       int W, float* out, d3d_stream_t stream);

/*
 * cas_mvsnet.py:94-108 -- ConvTranspose3d k=3, stride 2, padding 1, output_padding 1
 * (output exactly 2x per axis) + folded BatchNorm + ReLU + skip add.
 * in [Ci,D,H,W]; weight [Ci,Co,3,3,3] (nn.ConvTranspose3d layout); out [Co,2D,2H,2W].
 */
int d3d_convtranspose3d_k3s2(const float* in, const float* weight, const float* scale, const float* shift,
                             const float* skip, int relu, int Ci, int Co, int D, int H, int W, float* out,
                             d3d_stream_t stream);

/*
 * 2D family used by the slice-recurrent regulariser (adamvs.py:403-427) and the pair
 * visibility net (adamvs.py:198-238): 3x3 conv pad 1 stride 1|2 over the channel-wise
 * concatenation of up to two inputs (torch.cat((x,h),1), module.py:30,41), with
 * per-channel affine (folded BN or bias), activation and optional skip add.
 * act: 0 none, 1 ReLU.  in1 may be NULL (then Ci1 = 0).
 * weight [Co,Ci0+Ci1,3,3].
 */
int d3d_conv2d_k3(const float* in0, int Ci0, const float* in1, int Ci1, const float* weight, const float* scale,
                  const float* shift, const float* skip, int act, int Co, int H, int W, int stride, float* out,
                  d3d_stream_t stream);

/* ConvTranspose2d k=3 stride 2 pad 1 out_pad 1 (adamvs.py:411-414). weight [Ci,Co,3,3].
 * skip (optional) is added BEFORE the activation here: adamvs.py:424,
 * relu(upconv1(x) + reg_cost1); set skip_after_act = 1 for the UNet form
 * (adamvs.py:233-235, conv4 + relu(bn(convT(x)))). */
int d3d_convtranspose2d_k3s2(const float* in, const float* weight, const float* scale, const float* shift,
                             const float* skip, int skip_after_act, int act, int Ci, int Co, int H, int W,
                             float* out, d3d_stream_t stream);

/*
 * The same convolution family on the matrix cores (v_mfma_f32_16x16x4_f32: fp32 in, fp32 accumulate,
 * numerically an fmaf chain, so parity with the fp32 reference is unchanged) as ONE implicit-GEMM
 * entry point driven by a tap list.  It serves module.py:297-304 / cas_mvsnet.py:84-121 (Conv3d,
 * ConvTranspose3d) and module.py:5-51 / adamvs.py:198-238,403-427 (Conv2d over a channel concat,
 * ConvTranspose2d); the Python host packs the weights and emits the tap lists (ops.conv_gemm).
 *   wpack  [ntaps*(Ci0+Ci1)][mpad]: packed weights, row (t*Ci + ci), mpad = 16*ceil(Co/16) (64 if 48),
 *          zero padded; value = W[co][ci][tap t] (conv) or W[ci][co][tap t] (transposed).
 *   taps_zyx [ntaps][3] (HOST, signed char): input offset of tap t; input index = g*istride + offset.
 *   The launch iterates an output grid Dg x Hg x Wg; output index = g*ostride + (oz,oy,ox) in a
 *   [Co,Do,Ho,Wo] tensor (ostride 2 = one output-parity class of a stride-2 transposed conv).
 *   scale/shift/skip/act/skip_after_act as for d3d_conv2d_k3.  2D tensors use D = Dg = Do = 1.
 */
int d3d_conv_gemm_f32(const float* in0, int Ci0, const float* in1, int Ci1, const float* wpack, int mpad,
                      const float* scale, const float* shift, const float* skip, int skip_after_act, int act,
                      int Co, int D, int H, int W, int Dg, int Hg, int Wg, int Do, int Ho, int Wo, int istride,
                      int ostride, int oz, int oy, int ox, int ntaps, const signed char* taps_zyx, float* out,
                      d3d_stream_t stream);

/*
 * z-streaming variant of the matrix-core convolution (conv_stream.hip): every input plane is staged into
 * LDS once per (x,y) tile and feeds up to three live output planes; all packed weights stay resident in
 * LDS; GEMM rows are (fold position, c_out), so narrow layers fill the 16-row MFMA tile with neighbouring
 * outputs and a stride-2 transposed convolution is ONE launch (fold = its 2x2(x2) output parities).
 * Serves the same reference modules as d3d_conv_gemm_f32; returns D3D_ERR_UNSUPPORTED when the resident
 * weights do not fit LDS (callers then use d3d_conv_gemm_f32).
 *   geom (HOST int[15]) = {Gz,Gy,Gx, cz,cy,cx, sz,sy,sx, bz,by,bx, fz,fy,fx}: column grid; per dimension
 *        input index = g*c + tap offset, output index = g*s + b + fold position (clipped to Do/Ho/Wo).
 *   M = Co*fz*fy*fx <= 64 GEMM rows, row m = ((fz_i*fy + fy_i)*fx + fx_i)*Co + co;  mpad = 16 | 32 | 64.
 *   act: 0 none, 1 ReLU, 2 = ConvGRUCell gates fused (module.py:24-38): y = sigmoid(y), and the reset-gate rows
 *        c_out < ep_split are multiplied by the state h passed in `skip` (out = [r*h | u]); 3 = ConvGRUCell state
 *        update fused (module.py:41-51): out = u*h + (1-u)*tanh(y) with h in `skip`, u in `aux1` (image kernels,
 *        <= 16 GEMM rows; otherwise D3D_ERR_UNSUPPORTED).  aux1 / ep_split are ignored for act 0 | 1.
 *   wpack [ntaps][Ci0+Ci1][mpad] (zero where a fold position does not use a tap);
 *   taps_zyx (HOST signed char[ntaps][3]) sorted by z offset, ntaps <= 128.
 */
int d3d_conv_fold_f32(const float* in0, int Ci0, const float* in1, int Ci1, const float* wpack, int mpad, int M,
                      const float* scale, const float* shift, const float* skip, int skip_after_act, int act,
                      const float* aux1, int ep_split, int Co, int D, int H, int W, int Do, int Ho, int Wo,
                      const int* geom, int ntaps, const signed char* taps_zyx, float* out, d3d_stream_t stream);

/* Same contract with bf16 MFMA operands (v_mfma_f32_16x16x16_bf16; inputs and weights rounded to nearest-even
 * bf16 as the operands are formed, fp32 accumulation, fp32 tensors in memory): the precision BASELINE.json's
 * config 3 asks for.  Depth stays within the 1e-3 relative-L1 budget of the fp32 reference (tests/test_parity_gpu.py). */
int d3d_conv_fold_h16(const float* in0, int Ci0, const float* in1, int Ci1, const float* wpack, int mpad, int M,
                       const float* scale, const float* shift, const float* skip, int skip_after_act, int act,
                       const float* aux1, int ep_split, int Co, int D, int H, int W, int Do, int Ho, int Wo,
                       const int* geom, int ntaps, const signed char* taps_zyx, float* out, d3d_stream_t stream);

/*
 * module.py:24-51 ConvGRUCell gate math, fused:
 *   phase 0: gates [2Hc,H,W] (pre-activation, bias already applied) ->
 *            r = sigmoid(gates[:Hc]); u = sigmoid(gates[Hc:]); rh = r*h; u stored.
 *   phase 1: h' = u*h + (1-u)*tanh(convc)   (in place on h allowed)
 */
int d3d_gru_gates(const float* gates, const float* h, int Hc, int64_t plane, float* rh, float* u,
                  d3d_stream_t stream);
int d3d_gru_update(const float* u, const float* h, const float* convc, int64_t n, float* h_out,
                   d3d_stream_t stream);

/*
 * module.py:53-99 ConvGRUCell2 (msrednet.py:337-371): like ConvGRUCell, but every convolution output passes
 * nn.GroupNorm(1, C) (one group: statistics over all C*H*W elements; per-channel gamma/beta) first.
 *   d3d_groupnorm_stats: x holds ngroups consecutive segments of n elements; stats[2g] = sum, stats[2g+1] = sum of
 *     squares of segment g (device doubles; zeroed by the call, stream-ordered).  The r and u halves of the gate
 *     tensor are two segments of one call.
 *   d3d_gru_gates_gn:  r = sigmoid(gn_r(gates[:Hc])); u = sigmoid(gn_u(gates[Hc:])); rh = r*h.
 *   d3d_gru_update_gn: h' = u*h + (1-u)*tanh(gn_o(o)).
 * fast (ABI 9): 0 = torch's own expressions (IEEE division, tanhf: the fp32 mode), 1 = the one-exp-one-rcp forms the h16 kernels
 * use (about 2e-7 from the exact value; these launches are ~15 us each, 2816 of them per RED-Net view).
 */
int d3d_groupnorm_stats(const float* x, int64_t n, int ngroups, double* stats, d3d_stream_t stream);
int d3d_gru_gates_gn(const float* gates, const double* stats_r, const double* stats_u, const float* gamma_r,
                     const float* beta_r, const float* gamma_u, const float* beta_u, const float* h, int Hc,
                     int64_t plane, float eps, int fast, float* rh, float* u, d3d_stream_t stream);
int d3d_gru_update_gn(const float* o, const double* stats_o, const float* gamma, const float* beta, const float* u,
                      const float* h, int Hc, int64_t plane, float eps, int fast, float* h_out, d3d_stream_t stream);
/* The same cell in two elementwise passes over 104 channel planes instead of 128 (ABI 10): the reset half alone, and the update
 * gate evaluated where it is used --
 *   d3d_gru_reset_gn:        rh = sigmoid(gn_r(gates[:Hc])) * h
 *   d3d_gru_update_gates_gn: h' = u*h + (1-u)*tanh(gn_o(o)),  u = sigmoid(gn_u(gates[Hc:]))   (gates: the whole [2Hc,plane] tensor)
 * Per element the operations of the pair above in their order: the same bits.  plane % 4 == 0 and 16-byte aligned tensors, else
 * D3D_ERR_UNSUPPORTED (nothing launched). */
int d3d_gru_reset_gn(const float* gates, const double* stats_r, const float* gamma_r, const float* beta_r, const float* h, int Hc,
                     int64_t plane, float eps, int fast, float* rh, d3d_stream_t stream);
int d3d_gru_update_gates_gn(const float* o, const double* stats_o, const float* gamma, const float* beta, const float* gates,
                            const double* stats_u, const float* gamma_u, const float* beta_u, const float* h, int Hc, int64_t plane,
                            float eps, int fast, float* h_out, d3d_stream_t stream);
/* One ConvGRUCell2 step as ONE call (ABI 10): the gate convolution with the GroupNorm statistics in its epilogue, d3d_gru_reset_gn, the
 * candidate convolution with its statistics and d3d_gru_update_gates_gn issued back to back -- the same kernels on the same operands as
 * the four entry points one by one; what goes away is the host's work between them (a RED-Net view is 352 cells).
 *   x [Cx,H,W], h [Hc,H,W] -> hout [Hc,H,W]; wg / wc = ops._pack_z2_bf16 of the gate [2Hc,Cx+Hc,3,3] / candidate [Hc,Cx+Hc,3,3] weights;
 *   stats_g [2][2], stats_o [2]: fp64, ZEROED by the caller; gates [2Hc,H,W], rh [Hc,H,W], o [Hc,H,W]: scratch.
 * Cx + Hc = 16 | 24 | 32 | 40 (W % 4 == 0) or 64 | 128 (parts of 32); H * W % 4 == 0; 16-byte aligned tensors.  D3D_ERR_UNSUPPORTED
 * otherwise, with nothing launched. */
int d3d_gru2_cell_gn_h16(const float* x, int Cx, const float* h, int Hc, int H, int W, const void* wg, const float* bg, const void* wc,
                         const float* bc, const float* gamma_r, const float* beta_r, const float* gamma_u, const float* beta_u,
                         const float* gamma_o, const float* beta_o, float eps, int fast, double* stats_g, double* stats_o, float* gates,
                         float* rh, float* o, float* hout, d3d_stream_t stream);

/* ucsnet.py:137-151 (compute_depth of UCS-Net): d3d_softargmin_conf4 plus the spread of the per-pixel distribution,
 * var_out = lamb * sqrt(sum_d softmax(cost)_d * (depth_d - depth_out)^2)  [h,w]. */
int d3d_softargmin_conf4_var(const float* cost, const float* depth, int depth_mode, int D, int h, int w, float lamb,
                             float* depth_out, float* conf_out, float* var_out, d3d_stream_t stream);

/* ucsnet.py:42-51 (uncertainty_aware_samples, stages after the first): out[d,y,x] = low + step * d + 1e-12 with
 * low = cur - var, step = ((cur + var) - low) / (D - 1); cur_depth, exp_var [h,w] -> out [D,h,w].  (The first stage's
 * uniform hypotheses, ucsnet.py:33-41, are d3d_depth_range_samples in per-plane mode: the same formula.) */
int d3d_uncertainty_samples(const float* cur_depth, const float* exp_var, int D, int h, int w, float* out,
                            d3d_stream_t stream);

/*
 * adamvs.py:478-486 -- per-pair softmax over D, view weight = max_D prob,
 * pair depth = SUM_D prob*d.  score [D,h,w], depth [D] or [D,h,w].
 */
int d3d_pair_softmax_max(const float* score, const float* depth, int depth_mode, int D, int h, int w,
                         float* view_weight, float* pair_depth, d3d_stream_t stream);

/*
 * SURVEY.md §8f row N1 -- the consumer of the depth / confidence maps: geometric consistency between a reference
 * and a source view, and the fusion accumulators of one reference view.
 *
 * d3d_consistency_check replaces ConsistencyChecker.check (fuse/consistency_check_n.py:141-147 -> check_cupy
 * :29-138): reference pixel -> source pixel (nearest, "+0.5 truncate"; out-of-range indices wrap around as CuPy's
 * integer-array indexing does), sampled source depth -> world -> reference pixel; a pixel is consistent when the
 * reprojection distance < position_threshold, |d_reproj - d_ref| / d_ref < depth_threshold, the reference
 * confidence > confidence_threshold, the cosine between the world normals > normal_cos_threshold
 * (= cos(radians(normal_threshold)), :22) and d_ref > 0.
 *   depth_ref, prob_ref [H,W]; normal_ref [H,W,3]; depth_src [Hs,Ws]; normal_src [Hs,Ws,3]   (device, fp32)
 *   cam: HOST array of D3D_FUSION_CAM_DOUBLES doubles, every matrix row-major and computed in float32 as the
 *        reference does (linalg.inv / matmul of float32 arrays), then widened:
 *          inv(K_ref)[9], (E_src @ inv(E_ref))[:3,:4][12], K_src[9], inv(K_src)[9], inv(E_src)[16],
 *          E_ref[:3,:4][12], K_ref[9], inv(E_src[:3,:3])[9], inv(E_ref[:3,:3])[9]
 *   outputs (device; any may be NULL): mask [H,W] u8; depth_reprojected [H,W] (0 where inconsistent);
 *        depth_src_out [Hs,Ws]: a COPY of depth_src made by the caller, in which the samples of consistent pixels
 *        are set to 0 (:123-126); xyz_world_src [3,H,W] and angle_conf [3,H,W] (cosine, clamped at 0; 0 where
 *        inconsistent).
 *
 * d3d_fusion_ref_init / _accumulate / _finalize are the body of Fuse_Depth_Map.fuse_depths for one reference view
 * (fuse/fusion_3d_normal.py:452-474, :476-518, :522-527) on resident accumulators: all_xyz_world [3,H,W],
 * conf_sum [H,W] (the reference's three identical planes kept once), geo_mask_sum [H,W] i32, vis [H,W] i32
 * (= mask * src_idx, :518).  _accumulate is the consistency check fused with :513-518 -- the pair outputs never
 * reach memory.  For _ref_init the cam slots inv(K_ref), inv(E_src) (holding inv(E_ref)) and inv(E_ref[:3,:3])
 * are read; normal_world [H,W,3] (unit world normals, :466-469) may be NULL.
 */
#define D3D_FUSION_CAM_DOUBLES 94
int d3d_consistency_check(const float* depth_ref, const float* normal_ref, const float* prob_ref,
                          const float* depth_src, const float* normal_src, const double* cam, int H, int W, int Hs,
                          int Ws, double position_threshold, float depth_threshold, float normal_cos_threshold,
                          float confidence_threshold, unsigned char* mask, float* depth_reprojected,
                          float* depth_src_out, float* xyz_world_src, float* angle_conf, d3d_stream_t stream);
int d3d_fusion_ref_init(const float* depth_ref, const float* normal_ref, const double* cam, int H, int W,
                        float* all_xyz_world, float* conf_sum, int* geo_mask_sum, float* normal_world,
                        d3d_stream_t stream);
int d3d_fusion_accumulate(const float* depth_ref, const float* normal_ref, const float* prob_ref,
                          const float* depth_src, const float* normal_src, const double* cam, int H, int W, int Hs,
                          int Ws, double position_threshold, float depth_threshold, float normal_cos_threshold,
                          float confidence_threshold, int src_idx, int* geo_mask_sum, float* all_xyz_world,
                          float* conf_sum, int* vis, float* depth_src_out, d3d_stream_t stream);
int d3d_fusion_finalize(const float* all_xyz_world, const float* conf_sum, const int* geo_mask_sum, int H, int W,
                        int min_geo_consist_num, float* avg_xyz_world, unsigned char* final_mask, d3d_stream_t stream);

/*
 * fuse/fusion_3d_normal.py:545-570 -- the confirmed pixels of a reference view as point-cloud vertices, replacing the
 * boolean-index compaction and the Python loop over points.  Two calls, because the caller sizes the outputs:
 *   d3d_fusion_mark_points: pixel i is KEPT iff final_mask[i], its ordinal among the valid pixels (row-major) is a
 *     multiple of skip_line, and scene_range_xy[0] < x < [1] and [2] < y < [3] (host array of 4 doubles; strict, NaN fails).
 *     keep [H*W] bytes; counts[0] = valid pixels, counts[1] = kept points (device, read them after the stream).
 *   d3d_fusion_gather_points: out_xyz [n,3], out_color [n,3] = trunc(color * 255) of color [H,W,3] in 0..1 (NULL: skipped),
 *     out_normal [n,3] of normal_world [H,W,3] (NULL: skipped), out_views [n,n_vis] = sorted(vis[vis > 0] - 1) padded
 *     with -1, out_nviews [n]; rows in the order of the reference's lists.  vis: HOST array of n_vis device pointers.
 *   scratch: d3d_fusion_points_scratch_bytes(H, W) bytes of device memory shared by both calls (caller-owned).
 */
size_t d3d_fusion_points_scratch_bytes(int H, int W);
int d3d_fusion_mark_points(const float* avg_xyz_world, const unsigned char* final_mask, int H, int W, int skip_line,
                           const double* scene_range_xy, void* scratch, unsigned char* keep, unsigned* counts,
                           d3d_stream_t stream);
int d3d_fusion_gather_points(const float* avg_xyz_world, const unsigned char* keep, const int* const* vis, int n_vis,
                             const float* color, const float* normal_world, int H, int W, void* scratch, float* out_xyz,
                             int* out_color, float* out_normal, int* out_views, int* out_nviews, d3d_stream_t stream);


/*
 * SURVEY.md §8f row N2 -- PFM payload order.  save_pfm_utf8 (mvs/mvs_cas/datasets/data_io.py:196-223) writes rows
 * bottom-up (np.flipud) and read_pfm / load_pfm (data_io.py:150-193, IO/pfm.py:19-60) flips them back.
 * d3d_flip_rows: out[k][H-1-y][x] = maps[k][y][x] for n <= 8 maps [H,W] (maps: HOST array of device pointers; out
 * [n,H,W] device, must not alias an input): one staging buffer in file order for a single D2H copy, or the
 * inverse after an upload.
 */
int d3d_flip_rows(const float* const* maps, int n, int H, int W, float* out, d3d_stream_t stream);

/*
 * SURVEY.md §8f row N3 -- input side of a view: crop window + per-image normalisation of a decoded 8-bit image, as
 * the dataset item builder does for every view (mvs/mvs_cas/datasets/preprocess.py:60-88 crop_input, :92-117
 * center_image; cas_normal_eval.py:112-147).
 *   img [h,w,channels] u8 interleaved (device); window rows y0..y0+H, columns x0..x0+W; out [channels,H,W] fp32.
 *   mode 0 'standard': x / 255;  mode 1 'mean': (x - mean_c) / (sqrt(var_c) + 1e-8) with the population mean and
 *   variance of channel c over the window (exact integer sums, evaluated in double, applied in float32);
 *   mode 2 'vit' (3 channels): (x - {123.675, 116.28, 103.53}_c) / ({58.395, 57.12, 57.375}_c + 1e-8).
 *   sums: device workspace of 8 uint64 (zeroed by the call, stream-ordered).
 */
int d3d_center_image_u8(const unsigned char* img, int h, int w, int channels, int y0, int x0, int H, int W, int mode,
                        unsigned long long* sums, float* out, d3d_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* DEEP3D_PLANESWEEP_H */
